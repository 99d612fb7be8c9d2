"""Time the native k-hop sampler on an HI-Small-shaped transaction graph (515 080 accounts, 5 078 345 edges)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "models-for-relational-multimodal-data_amd"))
from tabgnn_amd.sampler import NeighborSampler

N, E = 515_080, 5_078_345
rng = np.random.default_rng(0)
def zipf(n, size, alpha):
    w = 1.0 / np.arange(1, n + 1) ** alpha
    return rng.choice(n, size=size, p=w / w.sum())
ei = np.stack([rng.permutation(N)[zipf(N, E, 1.0)], rng.permutation(N)[zipf(N, E, 0.5)]])
t0 = time.time(); s1 = {t: NeighborSampler(ei, N, (100, 100), num_threads=t) for t in (1, 4, 8, 16)}
print(f"build {time.time()-t0:.2f}s x{len(s1)}")
for B in (200, 8192):
    seeds = rng.choice(E, size=B, replace=False)
    for t, s in s1.items():
        s.sample(seeds, 1)
        t0 = time.time()
        for r in range(3):
            eid, lei, nodes = s.sample(seeds, 2 + r)
        dt = (time.time() - t0) / 3
        print(f"B={B} threads={t}: E_out={len(eid)} N_out={len(nodes)} {dt*1e3:.1f} ms -> {len(eid)/dt/1e6:.2f} M edges/s")
