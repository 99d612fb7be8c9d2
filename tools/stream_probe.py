#!/usr/bin/env python
"""Streaming-rate comparison on 1 GiB bf16 buffers: torch.add (vectorised elementwise kernel, uncapped grid),
tensor.copy_ (hipMemcpy D2D) and the package's own k_axpby (grid-capped, grid-stride loop), k_ln_fwd, k_bn_apply."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "models-for-relational-multimodal-data_amd"))
from tabgnn_amd import _lib as L, ops
dev = "cuda:0"
def rate(fn, nbytes, it=20):
    for _ in range(3): fn()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(it): fn()
    t1.record(); torch.cuda.synchronize()
    return nbytes * it / (t0.elapsed_time(t1) * 1e-3) / 1e9
n = 512 * 1024 * 1024
a = torch.randn(n // 4, device=dev).bfloat16().repeat(4); b = torch.empty_like(a); c = torch.empty_like(a)
print("torch.add(a,1)->b      ", round(rate(lambda: torch.add(a, 1.0, out=b), 2 * n * 2)), "GB/s")
print("torch.add(a,b)->c      ", round(rate(lambda: torch.add(a, b, out=c), 3 * n * 2)), "GB/s")
print("b.copy_(a)             ", round(rate(lambda: b.copy_(a), 2 * n * 2)), "GB/s")
print("tg_axpby(a,b)->c       ", round(rate(lambda: L.call("tg_axpby", L.ptr(a), L.ptr(b), L.ptr(c), n, 0.5, 0.5, L.dt(a), L.stream()), 3 * n * 2)), "GB/s")
x = a.view(-1, 128)
g, be = torch.ones(128, device=dev), torch.zeros(128, device=dev)
print("ops.layer_norm(x)      ", round(rate(lambda: ops.layer_norm(x, g, be), 2 * n * 2)), "GB/s (read x, write out; + 8 B/row stats)")
print("ops.act_dropout relu   ", round(rate(lambda: ops.act_dropout(x, "relu", 0.0), 2 * n * 2)), "GB/s")
