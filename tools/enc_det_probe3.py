"""Where inside the layer does a nondeterministic tile first go wrong?  z1 (pre-LN1 sum), z2 (pre-LN2 sum), out of repeated forwards."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "models-for-relational-multimodal-data_amd")]
import torch
import tabgnn_amd.encoder_layer as EL
from tabgnn_amd import ops
from tabgnn_amd.layers import ColumnTransformerLayer
dev = "cuda:0"
torch.manual_seed(0)
layer = ColumnTransformerLayer(128, 4, 128, dropout=0.5).to(dev)
tail = torch.nn.LayerNorm(128).to(dev)
sa = layer.self_attn
bf = lambda t: t.detach().to(torch.bfloat16).contiguous()
wpack, prm = EL.pack_layer(bf(sa.in_proj_weight), bf(sa.out_proj.weight), bf(layer.linear1.weight), bf(layer.linear2.weight),
                           sa.in_proj_bias, sa.out_proj.bias, layer.norm1.weight, layer.norm1.bias, layer.linear1.bias,
                           layer.linear2.bias, layer.norm2.weight, layer.norm2.bias, tail.weight, tail.bias)
R = int(os.environ.get("R", 60000))
x = torch.randn(R, 6, 128, device=dev).to(torch.bfloat16)
runs = []
for r in range(int(os.environ.get("REPS", 600))):
    out, z1, z2 = EL.fused_forward(x, 4, 0.0, True, 0.5, 0.5, wpack, prm, 7, [1, 2, 3, 4], True)
    runs.append((out.clone(), z1.clone(), z2.clone()))
ref = [torch.stack([runs[i][k] for i in range(5)]).float().median(0).values for k in range(3)]
n = 0
for r, (o, a, b) in enumerate(runs):
    d = [(t.float() - ref[k]).abs() for k, t in enumerate((o, a, b))]
    if max(float(t.max()) for t in d) == 0:
        continue
    n += 1
    if n > 6:
        continue
    dd = d[0] + d[1] + d[2]
    rows = (dd > 0).any(-1).any(-1).nonzero().flatten()
    t0 = int(rows[0])
    z = d[1][t0 - t0 % 5:t0 - t0 % 5 + 5].reshape(30, 128)          # z1 errors of the wave tile: token slot x channel
    toks = (z > 0).any(-1).nonzero().flatten().tolist()
    print(f"   z1 token slots with an error: {toks}; per-slot channels: {[(t, (z[t] > 0).nonzero().flatten().tolist()[:6]) for t in toks[:8]]}")
    def desc(t):
        blk = t[t0 - t0 % 5:t0 - t0 % 5 + 5]          # the 5 rows of the wave tile
        ch = (blk > 0).any(0).any(0).nonzero().flatten().tolist()
        return f"max {float(blk.max()):.4f}, {len(ch)} channels {ch[:12]}{'...' if len(ch) > 12 else ''}"
    print(f"rep {r}: tile {t0 // 5} (WG {(t0 // 5 // 4) % 512}, wave {(t0 // 5) % 4}, iteration {(t0 // 5 // 4) // 512})\n   z1: {desc(d[1])}\n   z2: {desc(d[2])}\n   out: {desc(d[0])}")
print("bad reps", n)
