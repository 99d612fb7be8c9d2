#!/usr/bin/env python
"""Time tg_encoder_fwd_bf16 alone (training form: z1 / z2 written) at the bench shape.  TABGNN_LIB_PATH selects an A/B
build (tools/encoder_ablate.sh).  env: R, S, H, P, N (launches)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "models-for-relational-multimodal-data_amd")]
import torch
import tabgnn_amd.encoder_layer as EL
from tabgnn_amd import ops
from tabgnn_amd.layers import ColumnTransformerLayer
dev = "cuda:0"
R, S, H = int(os.environ.get("R", 430162)), int(os.environ.get("S", 6)), int(os.environ.get("H", 4))
n = int(os.environ.get("N", 10))
torch.manual_seed(0)
layer = ColumnTransformerLayer(128, H, 128, dropout=0.5).to(dev)
tail = torch.nn.LayerNorm(128).to(dev)
sa = layer.self_attn
bf = lambda t: t.detach().to(torch.bfloat16)
wpack, prm = EL.pack_layer(bf(sa.in_proj_weight), bf(sa.out_proj.weight), bf(layer.linear1.weight), bf(layer.linear2.weight),
                           sa.in_proj_bias, sa.out_proj.bias, layer.norm1.weight, layer.norm1.bias, layer.linear1.bias,
                           layer.linear2.bias, layer.norm2.weight, layer.norm2.bias, tail.weight, tail.bias)
x = torch.randn(R, S, 128, device=dev).to(torch.bfloat16)
for p in [float(v) for v in os.environ.get("P", "0.5,0.0").split(",")]:
    run = lambda: EL.fused_forward(x, H, p, True, 0.5, 0.5, wpack, prm, 1234, [1, 2, 3, 4], True)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        run()
    e1.record(); torch.cuda.synchronize()
    print(f"lib={os.environ.get('TABGNN_LIB_PATH', 'default').split('/')[-2:-1]} H={H} S={S} p={p}: fwd {e0.elapsed_time(e1) / n:.3f} ms", flush=True)
