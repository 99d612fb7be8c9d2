import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "models-for-relational-multimodal-data_amd"))
import torch, tabgnn_amd as T
from tabgnn_amd import ops, synthetic as S, _lib as L
dev = "cuda:0"
ei_np, N = S.sampled_subgraph(8192, 42)
ei = torch.from_numpy(ei_np[:, 8192:].copy()).to(dev)
E = ei.shape[1]; F = 128
g = ops.SubgraphIndex.build(ei, N)
for dt, b in ((torch.bfloat16, 2), (torch.float32, 4)):
    h = torch.randn(E, F, device=dev, dtype=dt)
    agg = torch.empty(N, 4 * F, device=dev, dtype=dt); dagg = torch.randn_like(agg); dh = torch.empty_like(h)
    rp, pm = g.by_dst
    def f(): L.call("tg_pna_aggregate_fwd", L.ptr(h), L.ptr(rp), L.ptr(pm), L.ptr(agg), N, F, E, None, L.dt(h), L.stream())
    def bw(): L.call("tg_pna_aggregate_bwd", L.ptr(h), L.ptr(agg), L.ptr(dagg), L.ptr(rp), L.ptr(pm), L.ptr(dh), N, F, None, L.dt(h), L.stream())
    def fs(): L.call("tg_pna_aggregate_fwd", L.ptr(h), L.ptr(rp), None, L.ptr(agg), N, F, E, None, L.dt(h), L.stream())
    def bws(): L.call("tg_pna_aggregate_bwd", L.ptr(h), L.ptr(agg), L.ptr(dagg), L.ptr(rp), None, L.ptr(dh), N, F, None, L.dt(h), L.stream())
    for name, fn, nbytes in (("fwd", f, E * (F * b + 4) + N * 4 * F * b), ("bwd", bw, E * (2 * F * b + 4) + 2 * N * 4 * F * b),
                             ("fwd SORTED", fs, E * (F * b + 4) + N * 4 * F * b), ("bwd SORTED", bws, E * (2 * F * b + 4) + 2 * N * 4 * F * b)):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 10 * 1e-3
        print(f"aggregate {name} {dt}: {t*1e6:7.1f} us  {nbytes/t/1e9:6.0f} GB/s = {nbytes/t/8e12*100:4.1f}% of 8 TB/s ({nbytes/1e6:.0f} MB)")

# write-only floor: no edges at all (every node takes the deg-0 path)
z = torch.zeros(N + 1, dtype=torch.int32, device=dev)
for dt, b in ((torch.bfloat16, 2), (torch.float32, 4)):
    h = torch.randn(E, F, device=dev, dtype=dt); agg = torch.empty(N, 4 * F, device=dev, dtype=dt)
    def f0(): L.call("tg_pna_aggregate_fwd", L.ptr(h), L.ptr(z), L.ptr(g.by_dst[1]), L.ptr(agg), N, F, E, None, L.dt(h), L.stream())
    for _ in range(3): f0()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f0()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 10 * 1e-3
    print(f"aggregate fwd EMPTY graph {dt}: {t*1e6:7.1f} us  write {N*4*F*b/t/1e9:6.0f} GB/s")
    def fz(): agg.zero_()
    for _ in range(3): fz()
    torch.cuda.synchronize(); e0.record()
    for _ in range(10): fz()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 10 * 1e-3
    print(f"aggregate torch zero_ same bytes {dt}: {t*1e6:7.1f} us  write {N*4*F*b/t/1e9:6.0f} GB/s")
