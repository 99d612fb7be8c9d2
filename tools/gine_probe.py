#!/usr/bin/env python
"""Time tg_gine_aggregate_fwd / its backward at the bench subgraph shape (B=8192) and print achieved HBM GB/s."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "models-for-relational-multimodal-data_amd"))
from tabgnn_amd import ops, synthetic as S

dev = "cuda:0"
ei, N = S.sampled_subgraph(8192, 0)
ei = torch.from_numpy(ei)[:, 8192:].contiguous().to(dev)
E, F = ei.shape[1], 128
for flip in (False, True):
    g = ops.SubgraphIndex.build(ei, N)
    if flip:
        g = g.flip()
    x = torch.randn(N, F, device=dev).bfloat16().requires_grad_(True)
    le = torch.randn(E, F, device=dev).bfloat16().requires_grad_(True)
    co = torch.randn(N, F, device=dev).bfloat16()
    for name in ("fwd", "fwd+bwd"):
        def run():
            out = ops.gine_aggregate(x, le, g, 1.0)
            if name != "fwd":
                x.grad = le.grad = None
                out.backward(co)
        for _ in range(3):
            run()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(20):
            run()
        t1.record(); torch.cuda.synchronize()
        ms = t0.elapsed_time(t1) / 20
        fwd_bytes = E * (2 * F * 2 + 8) + 2 * N * F * 2
        print(f"flip={flip} {name}: {ms*1e3:.1f} us  (fwd algorithmic {fwd_bytes/1e6:.1f} MB -> {fwd_bytes/ms/1e6:.0f} GB/s if fwd only)  E={E} N={N}")
