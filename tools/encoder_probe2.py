#!/usr/bin/env python
"""One-kernel layer forward only, a few launches (for rocprofv3)."""
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
x = torch.randn(R, S, 128, device=dev).to(torch.bfloat16)
for _ in range(int(os.environ.get("N", 5))):
    ops.DropoutRNG.new_step(1)
    with torch.no_grad():
        EL.encoder_layer(x, layer, p, tail, 0.5, 0.5)
torch.cuda.synchronize()
