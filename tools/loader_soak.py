"""Soak of the device data path: N steps of DeviceBatchLoader -> train_step on the HI-Small-shaped graph; reports the loss
range, peak memory growth between the first and the last hundred steps, and that every batch kept the sampler's contract."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "models-for-relational-multimodal-data_amd")]
import numpy as np, torch
import tabgnn_amd as T
from tabgnn_amd import synthetic as S, DeviceBatchLoader, DeviceNeighborSampler
from tabgnn_amd.frame import stype
from tabgnn_amd.sampler import ColumnStore
dev = torch.device("cuda:0")
steps, B = int(os.environ.get("STEPS", 300)), int(os.environ.get("B", 2048))
rs = np.random.RandomState(0)
N, E = 515_080, 5_078_345
ei = np.stack([rs.permutation(N)[S._zipf_choice(rs, N, E, 1.0)], rs.permutation(N)[S._zipf_choice(rs, N, E, 0.5)]])
num, cat, ts = S.edge_table(E, 0)
labels = torch.from_numpy((rs.rand(E) < 0.01).astype(np.int64))
store = ColumnStore({stype.numerical: torch.from_numpy(num), stype.categorical: torch.from_numpy(cat), stype.timestamp: torch.from_numpy(ts)},
                    S.EDGE_COLS, {stype.relation: torch.ones(N, 1)}, S.NODE_COLS, labels).to(dev)
torch.manual_seed(1)
cfg = S.make_config(128, 2, 4, B, compute_dtype=torch.bfloat16)
model = T.TABGNNFusedS(cfg).to(dev).train()
flat = T.FlatParams(model, shadow_dtype=torch.bfloat16); opt = T.FusedAdam(flat, lr=cfg["lr"])
lw = torch.tensor(cfg["loss_weights"], device=dev)
smp = DeviceNeighborSampler(ei, N, (100, 100), dev)
seeds = [rs.choice(E, B, replace=False) for _ in range(steps)]
losses, mem = [], []
t0 = time.perf_counter()
for i, batch in enumerate(DeviceBatchLoader(smp, store, seeds, mode="index", rng_seed=3)):
    eid = batch[2].row_ids
    assert torch.equal(eid[:B].cpu(), torch.from_numpy(seeds[i])), i          # seeds first, in order, of THIS batch
    loss, _ = T.train_step(model, flat, opt, batch, lw)
    losses.append(loss)
    if i % 50 == 49:
        torch.cuda.synchronize(); mem.append(torch.cuda.memory_allocated(dev) / 1e9)
torch.cuda.synchronize()
l = torch.stack(losses).float().cpu()
print(f"{steps} steps, {1e3 * (time.perf_counter() - t0) / steps:.2f} ms/step; loss first {l[:10].mean():.4f} last {l[-10:].mean():.4f} finite {bool(torch.isfinite(l).all())}; "
      f"allocated GB every 50 steps: {[round(m, 2) for m in mem]}; reserved {torch.cuda.memory_reserved(dev) / 1e9:.1f} GB")
