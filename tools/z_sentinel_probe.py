"""Are the odd dwords of z1 / z2 WRONG or UNWRITTEN?  Output buffers pre-filled with a sentinel."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "models-for-relational-multimodal-data_amd")]
import torch
import tabgnn_amd.encoder_layer as EL
from tabgnn_amd import _lib as L, ops
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
R, S = 60000, 6
x = torch.randn(R, S, 128, device=dev).to(torch.bfloat16)
rs = (ctypes.c_uint32 * 4)(1, 2, 3, 4)
SENT = 0x7f7f
tot = {"out": 0, "z1": 0, "z2": 0}
for rep in range(20):
    bufs = [torch.full((R * S, 128), SENT, dtype=torch.int16, device=dev) for _ in range(3)]
    L.call("tg_encoder_fwd_bf16", L.ptr(x), L.ptr(bufs[0]), L.ptr(bufs[1]), L.ptr(bufs[2]), L.ptr(wpack), L.ptr(prm), R, S, 4, 1, 0.5, 0.5,
           1e-5, 0.0, 7, ctypes.addressof(rs), L.stream())
    torch.cuda.synchronize()
    for name, b in zip(tot, bufs):
        n = int((b == SENT).sum())
        tot[name] += n
        if n and rep < 3:
            pos = (b == SENT).nonzero()[:8].tolist()
            print(f"rep {rep} {name}: {n} elements still hold the sentinel, e.g. (token, channel) {pos}")
print("unwritten elements over 20 launches:", tot)

# what do the wrong dwords hold?  reference = the same launch on one workgroup per CU (never observed wrong)
def run():
    bufs = [torch.full((R * S, 128), SENT, dtype=torch.int16, device=dev) for _ in range(3)]
    L.call("tg_encoder_fwd_bf16", L.ptr(x), L.ptr(bufs[0]), L.ptr(bufs[1]), L.ptr(bufs[2]), L.ptr(wpack), L.ptr(prm), R, S, 4, 1, 0.5, 0.5,
           1e-5, 0.0, 7, ctypes.addressof(rs), L.stream())
    torch.cuda.synchronize()
    return bufs
ref = None
import numpy as np
cands = [run() for _ in range(6)]
z1s = torch.stack([c[1] for c in cands])
maj = z1s.float().median(0).values.to(torch.int16)            # (int16 view: majority bit pattern per element, good enough to spot outliers)
for k in range(6):
    bad = (z1s[k] != maj).any(-1).nonzero().flatten()
    if not bad.numel():
        continue
    t = int(bad[0]); tile = t // 30; slot = t % 30
    hexs = lambda v: " ".join(f"{int(q) & 0xffff:04x}" for q in v)
    print(f"launch {k}: token {t} tile {tile} slot {slot} (WG {(tile // 4) % 512} iteration {(tile // 4) // 512})")
    print("   this launch z1[96:112]:", hexs(z1s[k][t][96:112]))
    print("   majority    z1[96:112]:", hexs(maj[t][96:112]))
    zf = z1s[k].view(torch.bfloat16).float()[t - slot:t - slot + 30]      # the wave tile, this launch
    mu = zf.mean(-1, keepdim=True); var = zf.var(-1, unbiased=False, keepdim=True)
    x1 = ((zf - mu) / (var + 1e-5).sqrt() * layer.norm1.weight.float() + layer.norm1.bias.float()).to(torch.bfloat16)
    print("   x1 (LN1 of this tile's z1) slot", slot, "[96:112]:", hexs(x1[slot].view(torch.int16)[96:112]))
    print("   x  [96:112]:", hexs(x.reshape(-1, 128)[t].view(torch.int16)[96:112]))
    for other in (slot - 1, slot + 1, (slot + 2) % 30):
        print(f"   z1 of slot {other} [96:112]:", hexs(z1s[k][t - slot + other][96:112]))
    break
