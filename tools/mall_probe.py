#!/usr/bin/env python
"""Does a re-read of recently written data beat HBM?  Copy src -> dst repeatedly for buffer sizes from 8 MiB to 2 GiB
and print the achieved read+write rate: sizes that fit the L2s (32 MiB) / the 256 MiB Infinity Cache show up as
plateaus above the HBM copy rate.  Second table: producer -> consumer chains (b = f(a); c = f(b)) at each size."""
import torch
dev = "cuda:0"
def rate(fn, nbytes, it=30):
    for _ in range(3): fn()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(it): fn()
    t1.record(); torch.cuda.synchronize()
    return nbytes * it / (t0.elapsed_time(t1) * 1e-3) / 1e9
print("size_MiB  copy_GBs(same buffers)  chain3_GBs")
for mib in (8, 16, 32, 64, 96, 128, 192, 256, 384, 512, 1024, 2048):
    n = mib * 1024 * 1024 // 2
    a = torch.randn(n, device=dev).bfloat16(); b = torch.empty_like(a); c = torch.empty_like(a); d = torch.empty_like(a)
    r1 = rate(lambda: b.copy_(a), 2 * n * 2)
    def chain():
        torch.add(a, 1.0, out=b); torch.add(b, 1.0, out=c); torch.add(c, 1.0, out=d)
    r2 = rate(chain, 3 * 2 * n * 2)
    print(f"{mib:8d}  {r1:10.0f}  {r2:10.0f}")
