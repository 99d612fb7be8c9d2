"""Weight-gradient GEMMs (tg_gemm_tn_bf16 / tg_gemm_tn_scaled_bf16) at the step's shapes: time, TB/s of the two operand
streams, error against an fp32 torch product (chunked) incl. the bias gradient."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "models-for-relational-multimodal-data_amd"))
from tabgnn_amd import _lib as L
dev = "cuda:0"
E, N, S = 430162, 524165, 6
shapes = [("seed rows", 8192 * 6, 128, 128), ("seed W_in", 8192 * 6, 384, 128), ("B rows 128", 8192, 128, 128), ("B rows 384x128", 8192, 384, 128), ("nodes 128", N, 128, 128)] if os.environ.get("TN_SMALL") else [("encoder 128x128", E * S, 128, 128), ("encoder W_in", E * S, 384, 128), ("edge_emb", E, 128, 768),
          ("pna msg", E, 128, 384), ("edge-upd 2", E, 128, 128), ("post x", N, 128, 128), ("seed rows", 8192 * 6, 128, 128),
          ("fuse 1", 8192, 1536, 384), ("fuse 2", 8192, 1536, 1536), ("fuse 3", 8192, 384, 1536), ("ragged", 100003, 384, 256),
          ("tiny", 300, 384, 128)]
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
def ref_tn(g, x, scale=None):
    out = torch.zeros(g.shape[1], x.shape[1], device=dev, dtype=torch.float32)
    for i in range(0, g.shape[0], 1 << 18):
        gg = g[i:i + (1 << 18)].float()
        if scale is not None:
            gg = (gg * scale[i:i + (1 << 18), None]).bfloat16().float()
        out += gg.t() @ x[i:i + (1 << 18)].float()
    return out
torch.manual_seed(0)
for name, R, M, Nn in shapes:
    g = torch.randn(R, M, device=dev, dtype=torch.bfloat16)
    x = torch.randn(R, Nn, device=dev, dtype=torch.bfloat16)
    out = torch.empty(M, Nn, device=dev, dtype=torch.float32)
    db = torch.empty(M, device=dev, dtype=torch.float32)
    ws = torch.empty(L.load().tg_gemm_tn_workspace_floats(R, M, Nn), device=dev, dtype=torch.float32)
    t = timeit(lambda: L.call("tg_gemm_tn_bf16", L.ptr(g), L.ptr(x), L.ptr(out), L.ptr(db), L.ptr(ws), R, M, Nn, M, Nn, 0, L.stream()))
    ref = ref_tn(g, x)
    err = ((out - ref).abs().max() / ref.abs().max()).item()
    rb = g.float().sum(0)
    eb = ((db - rb).abs().max() / rb.abs().max()).item()
    by = R * (M + Nn) * 2
    print(f"{name:18s} R={R:8d} M={M:4d} N={Nn:4d}: {t*1e6:7.1f} us  {by/t/1e12:5.2f} TB/s  {2*R*M*Nn/t/1e12:6.1f} TFLOP/s  relerr {err:.1e} bias {eb:.1e}", flush=True)
# scaled post projection weight gradient: out [3F, K] = [g | amp g | att g]^T agg
F, K = 128, 512
g = torch.randn(N, F, device=dev, dtype=torch.bfloat16)
agg = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
scales = (torch.rand(N, 2, device=dev) * 2 + 0.1).float()
dw = torch.empty(3 * F, K, device=dev, dtype=torch.float32)
ws = torch.empty(L.load().tg_gemm_tn_workspace_floats(N, 3 * F, K), device=dev, dtype=torch.float32)
t = timeit(lambda: L.call("tg_gemm_tn_scaled_bf16", L.ptr(g), L.ptr(agg), L.ptr(scales), L.ptr(dw), L.ptr(ws), N, F, K, F, K, 0, L.stream()))
ref = torch.cat([ref_tn(g, agg), ref_tn(g, agg, scales[:, 0]), ref_tn(g, agg, scales[:, 1])], 0)
err = ((dw - ref).abs().max() / ref.abs().max()).item()
print(f"scaled post dW     R={N:8d} M={3*F:4d} N={K:4d}: {t*1e6:7.1f} us  {N*(F+K)*2/t/1e12:5.2f} TB/s  {2*N*3*F*K/t/1e12:6.1f} TFLOP/s  relerr {err:.1e}")
