"""Determinism of the one-kernel column-transformer forward: the same input many times, per R / mode."""
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
for q in list(layer.parameters()) + list(tail.parameters()):
    q._lp = q.detach().to(torch.bfloat16)
    if q.dim() == 2:
        q._lp_t = q._lp.t().contiguous()
reps = int(os.environ.get("REPS", 400))
for R in (13000, 60000):
    for mode in ("eval", "train-p0", "train-p0.5"):
        x = torch.randn(R, 6, 128, device=dev).to(torch.bfloat16)
        bad = 0
        first = None
        worst = 0.0
        for r in range(reps):
            ops.DropoutRNG.new_step(7)
            if mode == "eval":
                with torch.no_grad():
                    out = EL.encoder_layer(x, layer, 0.0, tail, 0.5, 0.5)
            else:
                xr = x.clone().requires_grad_(True)
                out = EL.encoder_layer(xr, layer, 0.0 if mode == "train-p0" else 0.5, tail, 0.5, 0.5).detach()
            if first is None:
                first = out.clone()
            elif not torch.equal(out, first):
                bad += 1
                rows = (out != first).any(-1).any(-1).nonzero().flatten()
                worst = max(worst, (out.float() - first.float()).abs().max().item())
                if bad <= 2:
                    print(f"   R={R} {mode} rep {r}: rows {rows[:8].tolist()} ... ({rows.numel()} rows), tiles {sorted(set((rows // 5).tolist()))[:6]}")
        print(f"R={R} {mode}: {bad} of {reps - 1} differ, worst {worst:.3e}")
