"""Microbenchmarks of the hand-written kernels at the bench shape: achieved GB/s against their algorithmic bytes."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "models-for-relational-multimodal-data_amd"))
import torch
import tabgnn_amd as T
from tabgnn_amd import ops, _lib as L
from tabgnn_amd.encoder_layer import _ln_fwd, _ln_bwd
dev = "cuda:0"
dt = torch.bfloat16
b = 2

def timeit(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3

def report(name, secs, nbytes):
    print(f"{name:44s} {secs*1e6:9.1f} us  {nbytes/secs/1e9:8.0f} GB/s  ({nbytes/1e6:.0f} MB)", flush=True)

R, S, C, H = 430162, 6, 128, 4
Tk = R * S
x = torch.randn(Tk, C, device=dev, dtype=dt); y = torch.randn_like(x); g = torch.randn_like(x)
gam = torch.ones(C, device=dev); bet = torch.zeros(C, device=dev); bias = torch.zeros(C, device=dev)
t = Tk * C * b
# plain copy for reference
report("torch copy (read+write)", timeit(lambda: y.copy_(x)), 2 * t)
for p in (0.0, 0.5):
    report(f"ln_fwd a+b p={p}", timeit(lambda: _ln_fwd(x, y, bias, gam, bet, None, 0., 1., p, 1, 1)), 3 * t)
report("ln_fwd a only", timeit(lambda: _ln_fwd(x, None, None, gam, bet, None, 0., 1., 0., 1, 1)), 2 * t)
report("ln_fwd a+res", timeit(lambda: _ln_fwd(x, None, None, gam, bet, y, .5, .5, 0., 1, 1)), 3 * t)
out, st = _ln_fwd(x, y, bias, gam, bet, None, 0., 1., 0.5, 1, 1)
da = torch.empty_like(x)
for p in (0.0, 0.5):
    report(f"ln_bwd a+b p={p}", timeit(lambda: _ln_bwd(x, y, bias, gam, st, g, da, True, None, 0., 1., p, 1, 1, False)), 5 * t)
report("ln_bwd a only", timeit(lambda: _ln_bwd(x, None, None, gam, st, g, da, False, None, 0., 1., 0., 1, 1, False)), 3 * t)
h = torch.empty_like(x)
for p in (0.0, 0.5):
    report(f"act_dropout_fwd p={p}", timeit(lambda: L.call("tg_act_dropout_fwd", L.ptr(x), L.ptr(h), x.numel(), 1, p, 1, 1, 1, L.stream())), 2 * t)
qkv = torch.randn(R, S, 3 * C, device=dev, dtype=dt); o = torch.empty(R, S, C, device=dev, dtype=dt)
lse = torch.empty(R, H, S, device=dev); dqkv = torch.empty_like(qkv); go = torch.randn_like(o)
for p in (0.0, 0.5):
    report(f"attn_fwd H={H} p={p}", timeit(lambda: L.call("tg_attn_fwd", L.ptr(qkv), L.ptr(o), L.ptr(lse), R, S, C, H, p, 1, 1, 1, L.stream())), 4 * t)
    report(f"attn_bwd H={H} p={p}", timeit(lambda: L.call("tg_attn_bwd", L.ptr(qkv), L.ptr(o), L.ptr(go), L.ptr(lse), L.ptr(dqkv), R, S, C, H, p, 1, 1, 1, L.stream())), 8 * t)
report("axpby", timeit(lambda: L.call("tg_axpby", L.ptr(x), L.ptr(y), L.ptr(h), x.numel(), .5, .5, 1, L.stream())), 3 * t)
# gemm_tn
for (M, N) in ((128, 128), (384, 128), (128, 768)):
    rows = Tk if N != 768 else R
    G = torch.randn(rows, M, device=dev, dtype=dt); X = torch.randn(rows, N, device=dev, dtype=dt)
    report(f"gemm_tn R={rows} M={M} N={N}", timeit(lambda: ops.weight_grad(G, X, True)), rows * (M + N) * b)
    W = torch.randn(M, N, device=dev, dtype=dt)
    report(f"  torch fwd X@W^T  [{rows},{N}]x[{N},{M}]", timeit(lambda: X @ W.t()), rows * (M + N) * b)
    report(f"  torch dx  G@W    [{rows},{M}]x[{M},{N}]", timeit(lambda: G @ W), rows * (M + N) * b)
