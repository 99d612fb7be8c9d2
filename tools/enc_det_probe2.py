"""Detail of the nondeterministic tiles of the one-kernel forward (eval, R = 13000)."""
import os, sys, collections
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
for q in list(layer.parameters()) + list(tail.parameters()):
    q._lp = q.detach().to(torch.bfloat16)
    if q.dim() == 2:
        q._lp_t = q._lp.t().contiguous()
R = int(os.environ.get("R", 13000))
use_tail = os.environ.get("TAIL", "1") == "1"
x = torch.randn(R, 6, 128, device=dev).to(torch.bfloat16)
outs = []
with torch.no_grad():
    for r in range(300):
        outs.append(EL.encoder_layer(x, layer, 0.0, tail if use_tail else None, 0.5 if use_tail else 0.0, 0.5 if use_tail else 1.0).clone())
# majority vote as the reference
ref = torch.stack(outs[:9]).float().median(0).values
wgs = collections.Counter(); waves = collections.Counter(); chans = collections.Counter(); toks = collections.Counter()
nb = 0
for r, o in enumerate(outs):
    d = (o.float() - ref).abs()
    if d.max() == 0:
        continue
    nb += 1
    rows = (d > 0).any(-1).any(-1).nonzero().flatten()
    for t in sorted(set((rows // 5).tolist())):
        wgs[(t // 4) % 512] += 1; waves[t % 4] += 1
    bad = d[rows]
    if nb <= 4:
        per_tok = bad.amax(-1)
        print(f"rep {r}: tiles {sorted(set((rows // 5).tolist()))}, rows {rows.tolist()[:6]}, max per token {per_tok[0].tolist()}, differing channels of row0 tok0: {(bad[0,0]>0).nonzero().flatten().tolist()[:40]}")
print("bad reps", nb, "WG index histogram (mod 512):", sorted(wgs.items())[:40])
print("wave in WG:", dict(waves), " iteration>0 tiles:", sum(1 for k in wgs if False))
