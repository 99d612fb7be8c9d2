"""Aggregation on the flipped graph (reverse message passing): the destinations are then the heavy-tailed SOURCES."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "models-for-relational-multimodal-data_amd"))
import tabgnn_amd as T
from tabgnn_amd import ops, synthetic as S
dev = "cuda:0"
ei_np, N = S.sampled_subgraph(8192, 42)
ei = torch.from_numpy(ei_np[:, 8192:].copy()).to(dev)
E, F = ei.shape[1], 128
g = ops.SubgraphIndex.build(ei, N)
for name, gg in (("forward graph", g), ("flipped graph", g.flip())):
    deg = gg.by_dst[0][1:] - gg.by_dst[0][:-1]
    h = torch.randn(E, F, device=dev, dtype=torch.bfloat16).requires_grad_(True)
    go = torch.randn(N, 4 * F, device=dev, dtype=torch.bfloat16)
    for sorted_rows in (False, True):
        def run():
            out = ops.pna_aggregate(h, gg, sorted_rows=sorted_rows)
            return out
        for _ in range(2): run().backward(go)
        torch.cuda.synchronize()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record(); out = run(); e[1].record(); out.backward(go); e[2].record(); torch.cuda.synchronize()
        print(f"{name} max in-degree {int(deg.max())} sorted={sorted_rows}: fwd {e[0].elapsed_time(e[1])*1e3:.0f} us  bwd {e[1].elapsed_time(e[2])*1e3:.0f} us")
