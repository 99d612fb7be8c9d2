"""Where a sampled B = 8192 batch spends its host time: draw / emit / index build, one thread (bench.py's end_to_end worker)."""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "models-for-relational-multimodal-data_amd"))
from tabgnn_amd import synthetic as S
from tabgnn_amd import sampler as SM

N, E, B = 515_080, 5_078_345, int(os.environ.get("B", 8192))
rs = np.random.RandomState(0)
ei = np.stack([rs.permutation(N)[S._zipf_choice(rs, N, E, 1.0)], rs.permutation(N)[S._zipf_choice(rs, N, E, 0.5)]])
sm = SM.NeighborSampler(ei, N, (100, 100), num_threads=1)
lib = SM._load()
p64 = SM._p64
fan = sm.fanout
t = {"gather seeds": [], "draw": [], "alloc": [], "emit": [], "index": []}
for i in range(10):
    seeds = np.ascontiguousarray(rs.choice(E, B, replace=False).astype(np.int64))
    t0 = time.perf_counter()
    s_src, s_dst = np.ascontiguousarray(sm._src[seeds]), np.ascontiguousarray(sm._dst[seeds])
    t1 = time.perf_counter()
    ne, nn = C.c_int64(0), C.c_int64(0)
    lib.tg_sampler_draw(sm._h, p64(s_src), p64(s_dst), p64(seeds), B, fan.ctypes.data_as(C.POINTER(C.c_int32)), 2, i, 1, E + B,
                        C.byref(ne), C.byref(nn))
    t2 = time.perf_counter()
    out_eid, out_ei, out_nodes = np.empty(ne.value, np.int64), np.empty((2, ne.value), np.int64), np.empty(nn.value, np.int64)
    t3 = time.perf_counter()
    lib.tg_sampler_emit(sm._h, 1, ne.value, p64(out_eid), p64(out_ei), p64(out_nodes))
    t4 = time.perf_counter()
    SM.host_batch_index(out_ei, nn.value, B)
    t5 = time.perf_counter()
    for k, v in zip(t, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
        t[k].append(v)
print(f"B={B} E_out={ne.value} N_out={nn.value}: " + "  ".join(f"{k} {1e3 * np.mean(v[2:]):.2f} ms" for k, v in t.items()))
