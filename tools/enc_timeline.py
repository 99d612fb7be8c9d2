#!/usr/bin/env python
"""Per-section share of a wave's time in the fused encoder forward (diagnostic build: tools/encoder_ablate.sh 1024, then
TABGNN_LIB_PATH=.../build/abl_1024/libtabgnn_hip.so python tools/enc_timeline.py).  s_memtime stamps at section edges,
summed per wave (csrc/encoder_fused.hip, EF_STAMP); read SHARES, not the length (stamps forbid overlaps)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "models-for-relational-multimodal-data_amd")]
import numpy as np
import torch
import tabgnn_amd.encoder_layer as EL
from tabgnn_amd import _lib as L
from tabgnn_amd.layers import ColumnTransformerLayer
dev = "cuda:0"
R, S, H = int(os.environ.get("R", 430162)), int(os.environ.get("S", 6)), int(os.environ.get("H", 4))
p = float(os.environ.get("P", 0.5))
torch.manual_seed(0)
layer = ColumnTransformerLayer(128, H, 128, dropout=0.5).to(dev)
tail = torch.nn.LayerNorm(128).to(dev)
sa = layer.self_attn
bf = lambda t: t.detach().to(torch.bfloat16)
wpack, prm = EL.pack_layer(bf(sa.in_proj_weight), bf(sa.out_proj.weight), bf(layer.linear1.weight), bf(layer.linear2.weight),
                           sa.in_proj_bias, sa.out_proj.bias, layer.norm1.weight, layer.norm1.bias, layer.linear1.bias,
                           layer.linear2.bias, layer.norm2.weight, layer.norm2.bias, tail.weight, tail.bias)
x = torch.randn(R, S, 128, device=dev).to(torch.bfloat16)
for _ in range(3):
    EL.fused_forward(x, H, p, True, 0.5, 0.5, wpack, prm, 1234, [1, 2, 3, 4], True)
torch.cuda.synchronize()
lib = ctypes.CDLL(L.LIB_PATH)
n_slots = 2048
buf = np.zeros((n_slots, 16), dtype=np.uint64)
rc = lib.tg_encoder_dbg_read(buf.ctypes.data_as(ctypes.c_void_p), n_slots)
assert rc == 0, rc
names = {0: "QKV chains + pack", 1: "softmax + PV (heads)", 2: "out-proj chains + epilogue", 3: "z1 store", 4: "LN1 stats/apply/park",
         5: "FFN1 chains + epilogue (+x prefetch issue)", 6: "FFN2 chains + epilogue (+x reload issue)", 7: "z2 store", 8: "LN2 + tail LN", 9: "out store",
         10: "tile top (geometry, x copy = wait for x)", 12: "boundary: vmcnt wait", 13: "boundary: barrier", 14: "boundary: DMA issue"}
tot = buf.sum(1).astype(np.float64)
live = tot > 0
print(f"waves with data: {live.sum()}, mean cycles per wave {tot[live].mean():.0f} (s_memtime ticks)")
mean = buf[live].astype(np.float64).mean(0)
for k in sorted(names):
    print(f"  {names[k]:48s} {100 * mean[k] / mean.sum():5.1f} %   {mean[k] / 42:8.0f} ticks per tile")
