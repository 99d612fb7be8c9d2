"""Where do the library copies / fills / adds of one train step come from?  Wraps the torch entry points that lower to
copyBuffer / fill / elementwise kernels and counts them per calling line of the package."""
import collections, os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "models-for-relational-multimodal-data_amd"))
import torch
import tabgnn_amd as T
from tabgnn_amd import synthetic as S
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
torch.manual_seed(1)
cfg = S.make_config(128, 2, 4, B, compute_dtype=torch.bfloat16)
model = T.TABGNNFusedS(cfg).to(dev).train()
flat = T.FlatParams(model, shadow_dtype=torch.bfloat16); opt = T.FusedAdam(flat, lr=cfg["lr"])
lw = torch.tensor(cfg["loss_weights"], device=dev)
from tabgnn_amd.sampler import batch_index
b = S.make_batch(B, seed=3, device=dev)
batch = (b[0], batch_index(b[1].cpu(), b[0].num_rows, B, dev), b[2], b[3])
for _ in range(2): T.train_step(model, flat, opt, batch, lw)
counts = collections.Counter()
def site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "tabgnn_amd" in fr.filename and "copy_sites" not in fr.filename:
            return f"{os.path.basename(fr.filename)}:{fr.lineno}"
    return "autograd/other"
def wrap(obj, name, pred=lambda self, out: True):
    orig = getattr(obj, name)
    def f(*a, **k):
        out = orig(*a, **k)
        try:
            if pred(a[0] if a else None, out): counts[(name, site())] += 1
        except Exception: pass
        return out
    setattr(obj, name, f)
wrap(torch.Tensor, "clone"); wrap(torch.Tensor, "copy_")
wrap(torch.Tensor, "contiguous", lambda s, o: o.data_ptr() != s.data_ptr())
wrap(torch.Tensor, "float", lambda s, o: o.data_ptr() != s.data_ptr())
wrap(torch.Tensor, "to", lambda s, o: isinstance(o, torch.Tensor) and o.data_ptr() != s.data_ptr())
wrap(torch.Tensor, "zero_"); wrap(torch.Tensor, "fill_"); wrap(torch.Tensor, "add_"); wrap(torch.Tensor, "sum")
wrap(torch.Tensor, "bfloat16", lambda s, o: o.data_ptr() != s.data_ptr())
for n in ("zeros", "zeros_like", "cat", "stack", "ones", "full"):
    wrap(torch, n)
T.train_step(model, flat, opt, batch, lw)
torch.cuda.synchronize()
for (n, s), c in sorted(counts.items(), key=lambda kv: -kv[1]):
    print(f"{c:4d}  {n:12s} {s}")
