"""Sampling-inclusive train loop (SURVEY §8f ranks 1-2 feeding the hot path): native k-hop sampler on the host ->
raw columns resident in HBM, gathered per batch -> fused train step.  One prefetch thread samples batch i+1 while
the GPU runs batch i (the ctypes call releases the GIL).  Prints one JSON line; not the bench.py metric."""
import argparse, json, os, sys, threading, queue, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "models-for-relational-multimodal-data_amd"))
import tabgnn_amd as T
from tabgnn_amd import synthetic as S
from tabgnn_amd.frame import stype
from tabgnn_amd.sampler import ColumnStore, NeighborSampler

ap = argparse.ArgumentParser()
ap.add_argument("--batch-size", type=int, default=8192)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--warmup", type=int, default=3)
ap.add_argument("--threads", type=int, default=0)
ap.add_argument("--nodes", type=int, default=515_080)
ap.add_argument("--edges", type=int, default=5_078_345)
ap.add_argument("--no-prefetch", action="store_true")
a = ap.parse_args()
dev = torch.device("cuda:0")
rs = np.random.RandomState(0)
N, E = a.nodes, a.edges
ei = np.stack([rs.permutation(N)[S._zipf_choice(rs, N, E, 1.0)], rs.permutation(N)[S._zipf_choice(rs, N, E, 0.5)]])
num, cat, ts = S.edge_table(E, 0)
labels = torch.from_numpy((rs.rand(E) < 0.001).astype(np.int64))
store = ColumnStore({stype.numerical: torch.from_numpy(num), stype.categorical: torch.from_numpy(cat),
                     stype.timestamp: torch.from_numpy(ts)}, S.EDGE_COLS,
                    {stype.relation: torch.ones(N, 1)}, S.NODE_COLS, labels).to(dev)
sampler = NeighborSampler(ei, N, (100, 100), num_threads=a.threads)
cfg = S.make_config(128, 2, 4, a.batch_size, compute_dtype=torch.bfloat16)
model = T.TABGNNFusedS(cfg).to(dev).train()
flat = T.FlatParams(model, shadow_dtype=torch.bfloat16)
opt = T.FusedAdam(flat, lr=cfg["lr"])
loss_w = torch.tensor(cfg["loss_weights"], device=dev)
total = a.steps + a.warmup
seeds = [rs.choice(E, a.batch_size, replace=False) for _ in range(total)]
t_sample = []

def produce(i):
    t0 = time.time()
    out = sampler.sample(seeds[i], i)
    t_sample.append(time.time() - t0)
    return out

def batches():
    if a.no_prefetch:
        for i in range(total):
            yield i, produce(i)
        return
    q = queue.Queue(maxsize=2)
    def work():
        for i in range(total):
            q.put((i, produce(i)))
    threading.Thread(target=work, daemon=True).start()
    for _ in range(total):
        yield q.get()

edges = 0
for i, (eid, lei, nodes) in batches():
    if i == a.warmup:
        torch.cuda.synchronize(); t0 = time.time(); edges = 0
    eid_d, nodes_d = eid.to(dev, non_blocking=True), nodes.to(dev, non_blocking=True)
    from tabgnn_amd.frame import TensorFrame
    edge_tf = TensorFrame({k: v.index_select(0, eid_d) for k, v in store.edge_feats.items()}, store.edge_cols)
    node_tf = TensorFrame({k: v.index_select(0, nodes_d) for k, v in store.node_feats.items()}, store.node_cols)
    y = store.labels.index_select(0, eid_d[:a.batch_size])
    T.train_step(model, flat, opt, (node_tf, lei.to(dev, non_blocking=True), edge_tf, y), loss_w)
    edges += eid.numel()
torch.cuda.synchronize()
dt = time.time() - t0
print(json.dumps({"mode": "e2e sampler+gather+train", "batch_size": a.batch_size, "steps": a.steps,
                  "edges_per_step": edges / a.steps, "ms_per_step": 1e3 * dt / a.steps,
                  "edges_per_s": edges / dt, "sampler_ms_mean": 1e3 * float(np.mean(t_sample[a.warmup:])),
                  "sampler_threads": a.threads, "prefetch": not a.no_prefetch}))
