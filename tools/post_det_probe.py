"""Determinism of the PNA post projection kernels (forward and d agg) at the bench's node count."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "models-for-relational-multimodal-data_amd")]
import torch
from tabgnn_amd import _lib as L
dev = "cuda:0"
torch.manual_seed(0)
N, F, K = 515080, 128, 512
agg = torch.randn(N, K, device=dev).to(torch.bfloat16)
x = torch.randn(N, F, device=dev).to(torch.bfloat16)
g = torch.randn(N, F, device=dev).to(torch.bfloat16)
wcat = (torch.randn(F, 3 * K, device=dev) * 0.05).to(torch.bfloat16)
wt_cat = (torch.randn(K, 3 * F, device=dev) * 0.05).to(torch.bfloat16)
wx = (torch.randn(F, F, device=dev) * 0.05).to(torch.bfloat16)
bias = torch.randn(F, device=dev)
scales = torch.rand((N + 127) // 128 * 128, 2, device=dev)
first = [None, None]
bad = [0, 0]
reps = int(os.environ.get("REPS", 150))
for r in range(reps):
    out = torch.empty(N, F, dtype=torch.bfloat16, device=dev)
    L.call("tg_pna_post_fwd_bf16", L.ptr(agg), L.ptr(x), L.ptr(wcat), L.ptr(wx), L.ptr(bias), L.ptr(scales), L.ptr(out), N, K, K, F, F, L.stream())
    dagg = torch.empty(N, K, dtype=torch.bfloat16, device=dev)
    L.call("tg_pna_post_dagg_bf16", L.ptr(g), L.ptr(wt_cat), L.ptr(scales), L.ptr(dagg), N, K, F, K, L.stream())
    for i, t in enumerate((out, dagg)):
        if first[i] is None:
            first[i] = t.clone()
        elif not torch.equal(t, first[i]):
            bad[i] += 1
print(f"post forward: {bad[0]} of {reps - 1} differ; d agg: {bad[1]} of {reps - 1} differ")
