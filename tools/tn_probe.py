"""Weight-gradient GEMM (tg_gemm_tn_bf16) at the step's shapes: time, TB/s of the two operand streams."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "models-for-relational-multimodal-data_amd"))
from tabgnn_amd import _lib as L
dev = "cuda:0"
E, N, S = 430162, 524165, 6
shapes = [("encoder 128x128", E * S, 128, 128), ("encoder W_in", E * S, 384, 128), ("edge_emb", E, 128, 768),
          ("pna msg", E, 128, 384), ("edge-upd 2", E, 128, 128), ("post x", N, 128, 128), ("seed rows", 8192 * 6, 128, 128)]
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for name, R, M, Nn in shapes:
    g = torch.randn(R, M, device=dev, dtype=torch.bfloat16)
    x = torch.randn(R, Nn, device=dev, dtype=torch.bfloat16)
    out = torch.empty(M, Nn, device=dev, dtype=torch.float32)
    db = torch.empty(M, device=dev, dtype=torch.float32)
    ws = torch.empty(L.load().tg_gemm_tn_workspace_floats(R, M, Nn), device=dev, dtype=torch.float32)
    t = timeit(lambda: L.call("tg_gemm_tn_bf16", L.ptr(g), L.ptr(x), L.ptr(out), L.ptr(db), L.ptr(ws), R, M, Nn, M, Nn, 0, L.stream()))
    ref = (g[:65536].float().t() @ x[:65536].float()) if R > 65536 else None
    by = R * (M + Nn) * 2
    err = ((out - g.float().t() @ x.float()).abs().max() / (g.float().t() @ x.float()).abs().max()).item() if R < 3e5 else float("nan")
    print(f"{name:18s} R={R:8d} M={M:4d} N={Nn:4d}: {t*1e6:7.1f} us  {by/t/1e12:5.2f} TB/s  relerr {err:.1e}")
