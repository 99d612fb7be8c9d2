"""Which module's output first differs between repeated eval forwards of the same batch (race hunt)."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "models-for-relational-multimodal-data_amd")]
import torch
import tabgnn_amd as T
from tabgnn_amd import synthetic as S
DEV = "cuda:0"
B = int(os.environ.get("B", 256))
cfg = S.make_config(128, 2, 4, B, compute_dtype=torch.bfloat16)
torch.manual_seed(0)
model = T.TABGNNFusedS(cfg).to(DEV)
batch = S.make_batch(B, seed=3, device=DEV)
model.train()
T.ops.DropoutRNG.new_step(1)
model(batch[0], batch[1], batch[2]).float().sum().backward()
model.eval()
cur = {}
def hook(name):
    def f(mod, inp, out):
        outs = out if isinstance(out, (tuple, list)) else (out,)
        cur[name] = [o.detach().clone() for o in outs if isinstance(o, torch.Tensor)]
    return f
for name, m in model.named_modules():
    if name:
        m.register_forward_hook(hook(name))
first = None
stat = collections.Counter()
with torch.no_grad():
    for r in range(int(os.environ.get("REPS", 600))):
        cur.clear()
        out = model(batch[0], batch[1], batch[2])
        snap = dict(cur)
        if first is None:
            first = snap
            order = list(snap)
            continue
        for name in order:
            if any(not torch.equal(a, b) for a, b in zip(snap[name], first[name])):
                d = max((a.float() - b.float()).abs().max().item() for a, b in zip(snap[name], first[name]))
                nbad = sum(int((a != b).sum()) for a, b in zip(snap[name], first[name]))
                stat[name] += 1
                if stat[name] <= 3:
                    print(f"rep {r}: first differing module {name}: max |d| {d:.3e}, {nbad} elements")
                break
print(dict(stat))
