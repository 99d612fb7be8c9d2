"""How close are the library (hipBLASLt via torch) forward / dX GEMMs of the step to the HBM stream rate?"""
import os, sys, torch, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "models-for-relational-multimodal-data_amd"))
from tabgnn_amd import _lib as L
dev = "cuda:0"
E, N, S = 430162, 524165, 6
shapes = [("qkv", E * S, 128, 384), ("out_proj/lin1/lin2", E * S, 128, 128), ("edge_emb", E, 768, 128),
          ("pna msg / edge-upd 1", E, 384, 128), ("pna post", N, 512, 384), ("pna x*Wx / lin", N, 128, 128),
          ("edge-upd 2", E, 128, 128)]
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
tot_f = tot_b = ideal = 0
for name, R, K, Nn in shapes:
    x = torch.randn(R, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(Nn, K, device=dev, dtype=torch.bfloat16) * 0.05
    b = torch.randn(Nn, device=dev, dtype=torch.bfloat16)
    g = torch.randn(R, Nn, device=dev, dtype=torch.bfloat16)
    tf = timeit(lambda: torch.addmm(b, x, w.t()))
    tb = timeit(lambda: g @ w)
    bf = b.float()
    y = torch.empty(R, Nn, device=dev, dtype=torch.bfloat16)
    wt = w.t().contiguous()                      # dX = G W  ==  NT GEMM with W^T [K, N] as the weight
    dx = torch.empty(R, K, device=dev, dtype=torch.bfloat16)
    mine_f = mine_b = float("nan")
    if Nn % 128 == 0 and K % 128 == 0:
        mine_f = timeit(lambda: L.call("tg_gemm_nt_bf16", L.ptr(x), L.ptr(w), L.ptr(bf), None, L.ptr(y), R, Nn, K, K, Nn, 0, 0.0, 0, 0, L.stream()))
        mine_b = timeit(lambda: L.call("tg_gemm_nt_bf16", L.ptr(g), L.ptr(wt), None, None, L.ptr(dx), R, K, Nn, Nn, K, 0, 0.0, 0, 0, L.stream()))
    by = (R * K + R * Nn) * 2
    tot_f += tf; tot_b += tb; ideal += by / 5.3e12
    print(f"{name:22s} R={R:8d} K={K:4d} N={Nn:4d}: fwd {tf*1e6:7.1f} us {by/tf/1e12:5.2f} TB/s {2*R*K*Nn/tf/1e12:6.1f} TFLOP/s | dX {tb*1e6:7.1f} us {by/tb/1e12:5.2f} TB/s || mine fwd {mine_f*1e6:7.1f} us {by/mine_f/1e12:5.2f} TB/s dX {mine_b*1e6:7.1f} us {by/mine_b/1e12:5.2f} TB/s")
print(f"sum fwd {tot_f*1e3:.2f} ms, sum dX {tot_b*1e3:.2f} ms, each at 5.3 TB/s: {ideal*1e3:.2f} ms")
