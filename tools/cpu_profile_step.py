"""cProfile of the host side of the train step in the launch-bound regime (default B=200)."""
import cProfile, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "models-for-relational-multimodal-data_amd"))
import torch, tabgnn_amd as T
from tabgnn_amd import synthetic as S
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda:0")
cfg = S.make_config(128, 2, 4, bs, compute_dtype=torch.bfloat16)
model = T.TABGNNFusedS(cfg).to(dev).train()
flat = T.FlatParams(model, shadow_dtype=torch.bfloat16); opt = T.FusedAdam(flat, lr=cfg["lr"])
lw = torch.tensor(cfg["loss_weights"], device=dev)
b = S.make_batch(bs, seed=1, device=dev)
from tabgnn_amd.sampler import batch_index
b = (b[0], batch_index(b[1].cpu(), b[0].num_rows, bs, dev), b[2], b[3])      # as bench.py: sampler-built CSRs
for _ in range(5):
    T.train_step(model, flat, opt, b, lw)
torch.cuda.synchronize()
torch.autograd.set_multithreading_enabled(False)      # backward on this thread, so cProfile sees it
pr = cProfile.Profile(); pr.enable()
for _ in range(20):
    T.train_step(model, flat, opt, b, lw)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(25)
st.sort_stats("cumtime").print_stats(r"tabgnn_amd", 45)
