#!/usr/bin/env python
"""One-kernel layer, training forward + backward at the bench shape, a few iterations (for rocprofv3 / timing)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "models-for-relational-multimodal-data_amd")]
import torch
import tabgnn_amd.encoder_layer as EL
from tabgnn_amd import ops
from tabgnn_amd.layers import ColumnTransformerLayer
dev = "cuda:0"
R, S, H = int(os.environ.get("R", 430162)), int(os.environ.get("S", 6)), int(os.environ.get("H", 4))
p = float(os.environ.get("P", 0.5))
torch.manual_seed(0)
layer = ColumnTransformerLayer(128, H, 128, dropout=0.5).to(dev)
tail = torch.nn.LayerNorm(128).to(dev)
for q in list(layer.parameters()) + list(tail.parameters()):
    q._lp = q.detach().to(torch.bfloat16)
    if q.dim() == 2:
        q._lp_t = q._lp.t().contiguous()
x = torch.randn(R, S, 128, device=dev).to(torch.bfloat16).requires_grad_(True)
go = torch.randn(R, S, 128, device=dev).to(torch.bfloat16)
n = int(os.environ.get("N", 4))
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
tf = tb = 0.0
for i in range(n):
    ops.DropoutRNG.new_step(1)
    ev[0].record()
    out = EL.encoder_layer(x, layer, p, tail, 0.5, 0.5)
    ev[1].record()
    out.backward(go)
    ev[2].record()
    torch.cuda.synchronize()
    if i > 0:
        tf += ev[0].elapsed_time(ev[1]); tb += ev[1].elapsed_time(ev[2])
print(f"fwd {tf / (n - 1):.3f} ms  bwd {tb / (n - 1):.3f} ms")
