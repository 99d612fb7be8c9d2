"""A few launches of the wide weight-gradient kernels for counter passes (rocprofv3 --pmc ... -- python3 tools/tn_pmc_probe.py)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "models-for-relational-multimodal-data_amd"))
from tabgnn_amd import _lib as L
dev = "cuda:0"
E, N, S = 430162, 524165, 6
def tn(R, M, Nn):
    g = torch.randn(R, M, device=dev, dtype=torch.bfloat16); x = torch.randn(R, Nn, device=dev, dtype=torch.bfloat16)
    out = torch.empty(M, Nn, device=dev, dtype=torch.float32); db = torch.empty(M, device=dev, dtype=torch.float32)
    ws = torch.empty(L.load().tg_gemm_tn_workspace_floats(R, M, Nn), device=dev, dtype=torch.float32)
    for _ in range(3):
        L.call("tg_gemm_tn_bf16", L.ptr(g), L.ptr(x), L.ptr(out), L.ptr(db), L.ptr(ws), R, M, Nn, M, Nn, 0, L.stream())
    torch.cuda.synchronize()
tn(E * S, 384, 128)
tn(E, 128, 384)
F, K = 128, 512
g = torch.randn(N, F, device=dev, dtype=torch.bfloat16); agg = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
scales = (torch.rand(N, 2, device=dev) * 2 + 0.1).float()
dw = torch.empty(3 * F, K, device=dev, dtype=torch.float32)
ws = torch.empty(L.load().tg_gemm_tn_workspace_floats(N, 3 * F, K), device=dev, dtype=torch.float32)
for _ in range(3):
    L.call("tg_gemm_tn_scaled_bf16", L.ptr(g), L.ptr(agg), L.ptr(scales), L.ptr(dw), L.ptr(ws), N, F, K, F, K, 0, L.stream())
torch.cuda.synchronize()
