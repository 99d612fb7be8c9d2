#!/usr/bin/env python
"""bench.py — edges/sec of one full supervised training step of the fused tabular-transformer + PNA path
(forward + weighted CE + backward + [RCCL grad all-reduce] + Adam) on synthetic HI-Small-shaped batches.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
      bench.py --gpus N --steps K --warmup W

Prints ONE JSON line (rank 0).  Inputs are resident in HBM before the timed region; `value` = total sampled
edges processed by all ranks / max-over-ranks wall time.  `roofline` is the PNA multi-aggregation kernel
(algorithmic bytes / HIP-event time, vs 8 TB/s HBM).  `cpu_baseline` = the oracle's train step (kind "port")
on the host cores, on a bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "models-for-relational-multimodal-data_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch-size", type=int, default=8192, help="seed edges per step per GPU (reference default 200)")
    ap.add_argument("--hidden", type=int, default=128)
    ap.add_argument("--layers", type=int, default=2)
    ap.add_argument("--nhead", type=int, default=4)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--distinct-batches", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch-size", type=int, default=200, help="reference default --batch_size (utils.py:40-44)")
    ap.add_argument("--cpu-steps", type=int, default=10, help="timed oracle steps per thread count (median reported)")
    ap.add_argument("--no-extras", action="store_true", help="skip every extra object (reference_batch, eval, "
                    "other_workloads, roofline_gemm, cpu_baseline, end_to_end): the bare contract line, for profiling")
    ap.add_argument("--workload", default="aml-fused", choices=["aml-fused", "tabgnn-arxiv", "wide64-c256", "wide64-graph", "reference-batch-graph", "reference-sampled-loop"],
                    help="aml-fused = the headline (BASELINE configs[1]); the other two run ONE extra leg alone "
                         "(configs[3] / configs[4] shapes) and print its object — for profiling, never the headline")
    ap.add_argument("--reverse-mp", action="store_true", help="PNAConvHetero (forward + reverse message passing); "
                    "not the BASELINE configuration")
    ap.add_argument("--index", default="sampler", choices=["sampler", "device"],
                    help="who builds the batch's CSR-by-destination/by-source: 'sampler' = host counting sort next to "
                         "the sampler (tg_host_csr), uploaded with the batch (SURVEY 8f rank 1); 'device' = rebuilt from "
                         "edge_index inside every forward (tg_csr_build, +~0.3 ms/step)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the sampling-inclusive loop reported beside `value`")
    ap.add_argument("--e2e-steps", type=int, default=8)
    return ap.parse_args()


def host_cores():
    """CPU threads this process may really use: affinity mask capped by the cgroup quota (the GPU box exposes
    all host cores to os.cpu_count() but grants a share)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except Exception:
            pass
    return min(n, 16)


def copy_rate_gbs(dev):
    """Device-to-device copy rate of this box (read + write bytes / time), the second roofline denominator of
    SURVEY 8d next to the 8 TB/s vendor peak."""
    n = 1 << 29
    a = torch.empty(n, dtype=torch.uint8, device=dev); b = torch.empty_like(a)
    for _ in range(2):
        b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        b.copy_(a)
    e1.record(); torch.cuda.synchronize()
    return 2 * n * 5 / (e0.elapsed_time(e1) * 1e-3) / 1e9


def stream_rate_gbs(dev):
    """Best streaming rate seen on this box: the package's own elementwise kernel (y = a/2 + b/2 over 512 MiB bf16
    operands, one pass per workgroup) — higher than the hipMemcpy rate above, so the stricter denominator."""
    from tabgnn_amd import _lib as L
    n = 1 << 28
    a = torch.zeros(n, dtype=torch.bfloat16, device=dev); b = torch.zeros_like(a); c = torch.empty_like(a)
    run = lambda: L.call("tg_axpby", L.ptr(a), L.ptr(b), L.ptr(c), n, 0.5, 0.5, L.dt(a), L.stream())
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        run()
    e1.record(); torch.cuda.synchronize()
    return 3 * 2 * n * 5 / (e0.elapsed_time(e1) * 1e-3) / 1e9


def end_to_end(model, flat, opt, loss_w, batch_size, steps, dev):
    """Sampling-inclusive loop on rank 0 (SURVEY 8d "end-to-end time including sampling"; 8f ranks 1-2): native k-hop
    sampler on the host (prefetching one batch ahead) -> id upload -> row gather from the HBM-resident raw table ->
    the same train step.  HI-Small-shaped graph: 515 080 nodes, 5 078 345 edges.  Reported beside `value`, never in it."""
    import queue, threading
    import numpy as np
    import tabgnn_amd as T
    from tabgnn_amd import synthetic as S
    from tabgnn_amd.frame import stype
    from tabgnn_amd.sampler import ColumnStore, NeighborSampler, host_batch_index
    rs = np.random.RandomState(0)
    N, E = 515_080, 5_078_345
    ei = np.stack([rs.permutation(N)[S._zipf_choice(rs, N, E, 1.0)], rs.permutation(N)[S._zipf_choice(rs, N, E, 0.5)]])
    num, cat, ts = S.edge_table(E, 0)
    labels = torch.from_numpy((rs.rand(E) < 0.001).astype(np.int64))
    store = ColumnStore({stype.numerical: torch.from_numpy(num), stype.categorical: torch.from_numpy(cat),
                         stype.timestamp: torch.from_numpy(ts)}, S.EDGE_COLS,
                        {stype.relation: torch.ones(N, 1)}, S.NODE_COLS, labels).to(dev)
    n_workers = 2                               # one sampler handle per host thread (the C call releases the GIL)
    samplers = [NeighborSampler(ei, N, (100, 100), num_threads=1) for _ in range(n_workers)]
    warm = 2
    total = steps + warm
    seeds = [rs.choice(E, batch_size, replace=False) for _ in range(total)]
    t_sample = [0.0] * total
    slots = [queue.Queue(maxsize=1) for _ in range(total)]
    ahead = threading.Semaphore(2 * n_workers)                      # bounded prefetch depth

    def work(w):
        for i in range(w, total, n_workers):
            ahead.acquire()
            try:
                t0 = time.perf_counter()
                eid, lei, nodes = samplers[w].sample(seeds[i], i)
                pre = host_batch_index(lei, nodes.numel(), batch_size)      # the batch's CSRs, in the sampler thread
                t_sample[i] = time.perf_counter() - t0
            except BaseException as e:      # noqa: BLE001 - handed to the consumer, which re-raises
                slots[i].put(e)
                return
            slots[i].put((eid, lei, nodes, pre))

    for w in range(n_workers):
        threading.Thread(target=work, args=(w,), daemon=True).start()
    edges = n_nodes = 0
    timer = T.ops.KernelTimer(only=("tg_pna_aggregate_fwd",))      # 2 event pairs per step: the named kernel on THIS shape
    t_model = []
    for i in range(total):
        item = slots[i].get()
        if isinstance(item, BaseException):
            raise RuntimeError(f"sampler thread failed at batch {i}") from item
        eid, lei, nodes, pre = item
        ahead.release()
        if i == warm:
            torch.cuda.synchronize(); t0 = time.perf_counter(); edges = n_nodes = 0
            T.ops.KernelTimer.active = timer
        # ids only cross PCIe: raw columns are read by id from the HBM-resident table, the CSRs arrive with the batch
        batch = store.batch(eid, lei, nodes, batch_size, index=pre)
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
        T.train_step(model, flat, opt, batch, loss_w)
        ev[1].record()
        if i >= warm:
            t_model.append(ev)
        edges += eid.numel()
        n_nodes += nodes.numel()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    T.ops.KernelTimer.active = None
    # The same loop with the DEVICE sampler (csrc/sampler_gpu.hip): seeds -> k-hop draw + relabel on the GPU over the
    # HBM-resident CSC -> train step with the batch's CSRs built on the device; no host thread, no upload.
    from tabgnn_amd import DeviceBatchLoader, DeviceNeighborSampler
    dsmp = DeviceNeighborSampler(ei, N, (100, 100), dev)
    loader = DeviceBatchLoader(dsmp, store, [seeds[i] for i in range(total)], mode="index", rng_seed=0)
    d_edges, d_ev = 0, []
    for i, batch in enumerate(loader):
        if i == warm:
            torch.cuda.synchronize(); t1 = time.perf_counter(); d_edges = 0
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
        T.train_step(model, flat, opt, batch, loss_w)
        ev[1].record()
        if i >= warm:
            d_ev.append(ev)
        d_edges += batch[1].edge_index.shape[1]
    torch.cuda.synchronize()
    d_dt = time.perf_counter() - t1
    # the sampler alone, synchronously (what the side stream hides): draw + emit of one batch
    torch.cuda.synchronize(); ts0 = time.perf_counter()
    for i in range(5):
        dsmp.sample(seeds[i], 1000 + i)
    torch.cuda.synchronize()
    device_sampler = dict(value=d_edges / d_dt, unit="edges/s", ms_per_step=1e3 * d_dt / steps, steps=steps,
                          edges_per_step=d_edges / steps,
                          model_only_ms_per_step=sum(a.elapsed_time(b) for a, b in d_ev) / len(d_ev),
                          sampler_ms_per_batch=1e3 * (time.perf_counter() - ts0) / 5, sampler_threads=0,
                          what="DeviceBatchLoader: k-hop draw + relabel + the batch's CSRs on the GPU, one batch ahead on a side "
                               "stream (no size wait, no CSR kernel on the training stream) -> train step; no sampler threads")
    F, b_act = model.config["n_hidden"], 2
    agg_bytes = (edges / steps - batch_size) * (F * b_act + 4) + (n_nodes / steps) * 4 * F * b_act
    agg_ms = timer.mean_ms("tg_pna_aggregate_fwd")
    sampled = dict(kernel="k_pna_aggregate_fwd", bound="hbm", achieved=agg_bytes / (agg_ms * 1e-3) / 1e9, peak=HBM_PEAK_GBS,
                   unit="GB/s", frac=agg_bytes / (agg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, algorithmic_bytes_per_launch=agg_bytes,
                   avg_launch_ms=agg_ms, launches_timed=timer.count("tg_pna_aggregate_fwd"),
                   shape="the batch the native sampler draws from the HI-Small-shaped graph (mean over the timed steps): "
                         f"E={int(edges / steps)}, N={int(n_nodes / steps)}")
    return dict(value=edges / dt, unit="edges/s", ms_per_step=1e3 * dt / steps, steps=steps, roofline_sampled=sampled,
                model_only_ms_per_step=sum(a.elapsed_time(b) for a, b in t_model) / len(t_model),
                edges_per_step=edges / steps, nodes_per_step=n_nodes / steps, sampler_ms_per_batch=1e3 * float(np.mean(t_sample[warm:])),
                device_sampler=device_sampler,
                sampler=f"libtabgnn_sampler.so k-hop [100,100], {n_workers} host threads (one handle each), prefetch <= "
                        f"{2 * n_workers} batches; batch = ids + CSRs built in the sampler thread (tg_host_batch_index; no row gather, no "
                        f"device CSR build); sampler_ms_per_batch includes them",
                graph="synthetic HI-Small-shaped: 515080 nodes, 5078345 edges, raw columns resident in HBM")


def cpu_baseline(model_sd, nhead, bs, steps, lr, loss_w):
    """Oracle (CPU restatement) train step timed on the host cores, model-only (batches pre-built), at the reference's
    default batch (B=200): ``steps`` timed steps after 2 warm-ups, MEDIAN step time, once on all granted cores and once
    on 4 threads (``benchmark.py:50``).  A bounded sample (~10-30 s), a reported baseline, not the target."""
    from oracle import step as ostep
    from tabgnn_amd import synthetic as S
    cores = host_cores()
    batches = []
    for i in range(4):
        node_tf, ei, edge_tf, y = S.make_batch(bs, seed=900 + i)
        batches.append(({k.value: v for k, v in node_tf.feat_dict.items()}, ei,
                        {k.value: v for k, v in edge_tf.feat_dict.items()}, y))
    E = sum(b[1].shape[1] for b in batches) / len(batches)

    def run(threads):
        torch.set_num_threads(threads)
        sd = {k: v.detach().float().cpu().clone() for k, v in model_sd.items()}
        opt_state, times = {}, []
        for i in range(steps + 2):
            nf, ei, ef, y = batches[i % len(batches)]
            t0 = time.perf_counter()
            ostep.train_step(sd, opt_state, nhead, bs, nf, ei, ef, y, torch.tensor(loss_w), lr, p_backbone=0.5, p_head=0.083)
            if i >= 2:
                times.append(time.perf_counter() - t0)
        times.sort()
        med = times[len(times) // 2]
        return dict(value=E / med, ms_per_step_median=1e3 * med, ms_per_step_min=1e3 * times[0], threads=threads)

    full = run(cores)
    four = run(min(4, cores))
    torch.set_num_threads(cores)
    return dict(value=full["value"], unit="edges/s", cores=cores, kind="port", ms_per_step_median=full["ms_per_step_median"],
                threads4=four,
                sample=f"oracle/step.py train step (fp32, dropout on, model-only: batches pre-built), B={bs} seed edges "
                       f"(E~{int(E)} sampled edges/step, the reference's default batch), median of {steps} steps after "
                       f"2 warm-ups; `value` on {cores} threads, `threads4` on 4 (benchmark.py:50)")


def count_launches(fn):
    """Device kernels launched by ``fn()`` (one step), counted by torch's profiler; None if the profiler is unavailable
    (e.g. under rocprofv3)."""
    try:
        from torch.profiler import ProfilerActivity, profile
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            fn()
            torch.cuda.synchronize()
        n = 0
        for ev in prof.events():
            if getattr(ev, "device_type", None) is not None and "cuda" in str(ev.device_type).lower():
                n += 1
        return n or None
    except Exception:
        return None


def _child_env():
    """Environment of a plain single-process child run on this GPU (no process-group variables of the parent)."""
    return {k: v for k, v in os.environ.items()
            if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "TABGNN_FORCE_ALLREDUCE",
                         "TABGNN_DIST_BACKEND", "TABGNN_ONE_DEVICE", "TORCHELASTIC_RUN_ID", "GROUP_RANK", "ROLE_RANK")}


def fp32_twin(args):
    """The SAME step (same workload, batch size, model) in fp32 — the reference's arithmetic (no mixed
    precision anywhere in it) — beside the bf16 `value`: `python bench.py --dtype fp32 --steps 5 --warmup 2 --no-extras`
    in a child process after the timed region.  fp32 activations, library GEMMs, the op-by-op column transformer."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--dtype", "fp32", "--steps", "5", "--warmup", "2", "--no-extras",
           "--batch-size", str(args.batch_size), "--hidden", str(args.hidden), "--layers", str(args.layers),
           "--nhead", str(args.nhead), "--index", args.index]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=400, env=_child_env())
        g = json.loads(r.stdout.strip().splitlines()[-1])
        return {"dtype": "fp32", "value": g["value"], "unit": g["unit"], "ms_per_step": g["ms_per_step"], "steps": g["steps"],
                "peak_hbm_gb": g["config"].get("peak_hbm_gb"),
                "what": "same B, E, N, model and step definition as `value`, fp32 storage and arithmetic (parity mode)"}
    except Exception as e:      # noqa: BLE001 - reported, never fatal for the bench line
        return {"error": repr(e)[:300]}


def reference_batch(args, cdt, dev):
    """The reference's default batch (``utils.py:40-44``: --batch_size 200 -> E = 10 702 sampled edges): step time,
    edges/s and device launches per step of the same train step.  Outside the timed region of `value`."""
    import tabgnn_amd as T
    from tabgnn_amd import synthetic as S
    B = 200
    torch.manual_seed(1234)
    cfg = S.make_config(args.hidden, args.layers, args.nhead, B, compute_dtype=cdt)
    model = T.TABGNNFusedS(cfg).to(dev).train()
    flat = T.FlatParams(model, shadow_dtype=cdt)
    opt = T.FusedAdam(flat, lr=cfg["lr"])
    lw = torch.tensor(cfg["loss_weights"], device=dev)
    batches = [S.make_batch(B, seed=77 + i, device=dev) for i in range(4)]
    if args.index == "sampler":      # as in the timed region: the batch carries the sampler-built CSRs
        from tabgnn_amd.sampler import batch_index
        batches = [(b[0], batch_index(b[1].cpu(), b[0].num_rows, B, dev), b[2], b[3]) for b in batches]

    def timed(one, n=50):
        for i in range(10):
            one(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            one(i)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n

    dt_eager = timed(lambda i: T.train_step(model, flat, opt, batches[i % 4], lw))
    E = sum(b[1].shape[1] for b in batches) / 4
    launches = count_launches(lambda: T.train_step(model, flat, opt, batches[0], lw))
    out = dict(batch_size=B, edges_per_step=int(E), ms_per_step=1e3 * dt_eager, value=E / dt_eager, unit="edges/s",
               launches_per_step=launches, steps=50, mode="eager",
               note="reference default --batch_size 200 (utils.py:40-44); launch-bound regime")
    if os.environ.get("TABGNN_NO_GRAPH") == "1":
        return out
    # The same step as one HIP graph per shape bucket, measured in a CHILD process (a fault inside a replay would
    # take this process, and with it the bench line, down): `python bench.py --workload reference-batch-graph`.
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--workload", "reference-batch-graph", "--dtype", args.dtype,
           "--hidden", str(args.hidden), "--layers", str(args.layers), "--nhead", str(args.nhead)]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=_child_env())
        g = json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else None
    except Exception as e:      # noqa: BLE001 - reported, never fatal for the bench line
        r, g = None, None
        out["graph_error"] = repr(e)[:200]
    if g is None:
        out.setdefault("graph_error", (r.stderr[-300:] if r is not None else "") or "child failed")
        return out
    out.update(eager_ms_per_step=out["ms_per_step"], eager_value=out["value"])
    out.update(g)
    return out


def _child_leg(extra_args, timeout=420):
    """One leg in a child process -> its JSON object (or {"error": ...})."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__)] + list(extra_args)
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=_child_env())
        if r.returncode != 0:
            return {"error": (r.stderr or "child failed")[-300:]}
        return json.loads(r.stdout.strip().splitlines()[-1])
    except Exception as e:      # noqa: BLE001 - reported, never fatal for the bench line
        return {"error": repr(e)[:200]}


def reference_batch_sampled_loop(cfg, cdt, dev, B, steps=120, warm=60):
    """The reference's loop at its default batch END TO END as graph replays: native sampler threads ->
    ``graph_step.prepare_sample`` (pad to the bucket, all index parts, one pinned arena of ids) -> one upload -> replay;
    raw columns are read by id from the HBM-resident tables.  Real sampled batches fall into several (E_pad, N_pad)
    buckets: the first batch of a bucket pays its capture, so the loop is timed after ``warm`` steps."""
    import queue, threading
    import numpy as np
    import tabgnn_amd as T
    from tabgnn_amd import synthetic as S
    from tabgnn_amd import graph_step as G
    from tabgnn_amd.frame import stype
    from tabgnn_amd.sampler import ColumnStore, NeighborSampler
    rs = np.random.RandomState(0)
    N, E = 515_080, 5_078_345
    ei = np.stack([rs.permutation(N)[S._zipf_choice(rs, N, E, 1.0)], rs.permutation(N)[S._zipf_choice(rs, N, E, 0.5)]])
    num, cat, ts = S.edge_table(E, 0)
    labels = torch.from_numpy((rs.rand(E) < 0.001).astype(np.int64))
    store = ColumnStore({stype.numerical: torch.from_numpy(num), stype.categorical: torch.from_numpy(cat),
                         stype.timestamp: torch.from_numpy(ts)}, S.EDGE_COLS,
                        {stype.relation: torch.ones(N, 1)}, S.NODE_COLS, labels).to(dev)
    torch.manual_seed(4321)
    model = T.TABGNNFusedS(cfg).to(dev).train()
    flat = T.FlatParams(model, shadow_dtype=cdt)
    opt = T.FusedAdam(flat, lr=cfg["lr"])
    lw = torch.tensor(cfg["loss_weights"], device=dev)
    step = G.GraphedTrainStep(model, flat, opt, lw, B)
    ids0 = torch.zeros(1, dtype=torch.int64, device=dev)
    frames = (T.TensorFrame(store.node_feats, store.node_cols, None, ids0),
              T.TensorFrame(store.edge_feats, store.edge_cols, None, ids0))
    n_workers = 2
    samplers = [NeighborSampler(ei, N, (100, 100), num_threads=1) for _ in range(n_workers)]
    total = steps + warm
    seeds = [rs.choice(E, B, replace=False) for _ in range(total)]
    t_host = [0.0] * total
    slots = [queue.Queue(maxsize=1) for _ in range(total)]
    ahead = threading.Semaphore(4 * n_workers)

    def work(w):
        for i in range(w, total, n_workers):
            ahead.acquire()
            try:
                t0 = time.perf_counter()
                eid, lei, nodes = samplers[w].sample(seeds[i], i)
                prep = G.prepare_sample(eid, lei, nodes, labels[eid[:B]], B)
                t_host[i] = time.perf_counter() - t0
            except BaseException as e:      # noqa: BLE001 - handed to the consumer, which re-raises (never a silent hang on the queue)
                slots[i].put(e)
                return
            slots[i].put(prep)

    for w in range(n_workers):
        threading.Thread(target=work, args=(w,), daemon=True).start()
    edges = 0
    new_after_warm = 0
    kept = []
    for i in range(total):
        prep = slots[i].get()
        if isinstance(prep, BaseException):
            raise RuntimeError(f"sampler thread failed at batch {i}") from prep
        ahead.release()
        if i == warm:
            torch.cuda.synchronize(); t0 = time.perf_counter(); edges = 0
        nb = len(step.buckets)
        loss, _ = step(prep, frames)
        if i >= warm and len(step.buckets) > nb:
            new_after_warm += 1
        edges += prep.e_real
        kept.append(prep)
        if len(kept) > 8:
            kept.pop(0)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # the same loop body over the last eight batches again, already prepared: upload + replay alone, the like-for-like
    # GPU-side floor of `ms_per_step` (the plain reference_batch replay runs 10.7 k-edge batches, these are 34 k)
    for prep in kept:
        step(prep, frames)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(5):
        for prep in kept:
            loss, _ = step(prep, frames)
    torch.cuda.synchronize()
    replay_only = 1e3 * (time.perf_counter() - t1) / (5 * len(kept))
    # The same loop on the DEVICE sampler: seeds -> k-hop draw + relabel + padding + index parts on the GPU, one batch
    # ahead on a side stream (tabgnn_amd.DeviceBatchLoader, mode "bucket") -> device-to-device copy -> replay.  No sampler
    # thread, no upload, no size wait; the buckets captured above are reused (same model, same GraphedTrainStep).
    from tabgnn_amd import DeviceBatchLoader, DeviceNeighborSampler
    dsmp = DeviceNeighborSampler(ei, N, (100, 100), dev)
    # two passes over the SAME seed batches: the first captures every bucket these draws fall into (a capture costs ~0.1 s
    # once), the second is timed — replays only, as in a steady-state epoch
    d_seeds = [rs.choice(E, B, replace=False) for _ in range(steps)]
    for _ in DeviceBatchLoader(dsmp, store, d_seeds, mode="bucket", rng_seed=7):
        step(_, frames)
    torch.cuda.synchronize()
    d_edges = d_new = 0
    t2 = time.perf_counter()
    for prep in DeviceBatchLoader(dsmp, store, d_seeds, mode="bucket", rng_seed=7):
        nb = len(step.buckets)
        step(prep, frames)
        d_new += int(len(step.buckets) > nb)
        d_edges += prep.e_real
    torch.cuda.synchronize()
    d_dt = time.perf_counter() - t2
    device_loop = dict(ms_per_step=1e3 * d_dt / steps, value=d_edges / d_dt, unit="edges/s", steps=steps,
                       edges_per_step=d_edges / steps, sampler_threads=0, buckets_captured_inside_the_timed_steps=d_new,
                       what="DeviceBatchLoader(mode=bucket): draw + relabel + pad + index parts on the GPU one batch ahead on a "
                            "side stream -> device-to-device copy -> graph replay; no sampler threads, no upload")
    return dict(ms_per_step=1e3 * dt / steps, value=edges / dt, unit="edges/s", steps=steps, edges_per_step=edges / steps,
                device_sampler=device_loop,
                buckets=len(step.buckets), buckets_captured_inside_the_timed_steps=new_after_warm,
                host_ms_per_batch=1e3 * float(np.mean(t_host[warm:])), sampler_threads=n_workers, final_loss=float(loss),
                upload_and_replay_only_ms_per_step=replay_only,
                what="sampler -> prepare_sample (pad + index parts) -> one pinned upload -> graph replay; lazy frames over "
                     "the HBM-resident HI-Small-shaped tables (515 080 nodes, 5 078 345 edges), fan-out [100, 100]")


def reference_batch_graph(args, cdt, dev):
    """Child leg of ``reference_batch``: batches padded to their bucket on the host next to the sampler
    (``graph_step.prepare``), uploaded ahead of the step, copied into the bucket's static buffers, replayed.
    Edges/s counts the REAL edges of the batches, not the padding."""
    import tabgnn_amd as T
    from tabgnn_amd import synthetic as S
    from tabgnn_amd import graph_step as G
    B = 200
    torch.manual_seed(1234)
    cfg = S.make_config(args.hidden, args.layers, args.nhead, B, compute_dtype=cdt)
    model = T.TABGNNFusedS(cfg).to(dev).train()
    flat = T.FlatParams(model, shadow_dtype=cdt)
    opt = T.FusedAdam(flat, lr=cfg["lr"])
    lw = torch.tensor(cfg["loss_weights"], device=dev)
    plain = [S.make_batch(B, seed=77 + i, device=dev) for i in range(4)]
    preps = [G.prepare(b, B).to(dev) for b in plain]
    step = G.GraphedTrainStep(model, flat, opt, lw, B)
    frames = (plain[0][0], plain[0][2])
    for i in range(10):
        step(preps[i % 4], frames)
    torch.cuda.synchronize()
    n = 100
    t0 = time.perf_counter()
    for i in range(n):
        loss, _ = step(preps[i % 4], frames)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    E = sum(p.e_real for p in preps) / 4
    sampled = reference_batch_sampled_loop(cfg, cdt, dev, B)
    return dict(mode="hip-graph replay over shape buckets", ms_per_step=1e3 * dt, value=E / dt, steps=n,
                buckets=len(step.buckets), padded_edges_per_step=int(sum(p.key[0] for p in preps) / 4),
                final_loss=float(loss), sampled_loop=sampled,
                graph_note="one captured graph per (E_pad, N_pad) bucket; dropout seed, Adam step count and the "
                           "BatchNorm row count are read from device memory by the kernels")


def eval_throughput(model, batches, dev, steps=10):
    """Inference (main.py:104-155: eval mode under no_grad) on the bench batches: edges/s of the forward alone."""
    model.eval()
    with torch.no_grad():
        for i in range(3):
            model(*batches[i % len(batches)][:3])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        edges = 0
        for i in range(steps):
            b = batches[i % len(batches)]
            model(*b[:3])
            edges += b[1].shape[1]
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    model.train()
    return dict(value=edges / dt, unit="edges/s", ms_per_step=1e3 * dt / steps, steps=steps,
                what="TABGNNFusedS forward, eval mode, no_grad (main.py:104-155)")


# ----------------------------------------------------------------------------------------------- other BASELINE shapes


def arxiv_batch(B=200, fan=(15, 10), ncol=129, seed=0, device="cpu"):
    """ogbn-arxiv-shaped node-seeded 2-hop sample (BASELINE configs[3]): B seed nodes, fan-out 15 then 10 in-neighbours;
    node table = 129 numerical columns ~N(-0.1, 0.11) (data/ogbn-arxiv.ipynb cell 4), edge table = 1 relation column."""
    import numpy as np
    import tabgnn_amd as T
    st = T.stype
    rs = np.random.RandomState(seed)
    V = 169_343
    src, dst, frontier = [], [], rs.choice(V, B, replace=False)
    for f in fan:
        nb = rs.randint(0, V, size=(frontier.size, f))
        src.append(nb.reshape(-1)); dst.append(np.repeat(frontier, f))
        frontier = np.unique(nb)
    src, dst = np.concatenate(src), np.concatenate(dst)
    nodes, inv = np.unique(np.concatenate([src, dst]), return_inverse=True)
    ei = inv.reshape(2, -1).astype(np.int64)
    N, E = nodes.size, ei.shape[1]
    names_n = {st.numerical: [f"f_{i}" for i in range(ncol - 1)] + ["year"]}
    node_tf = T.TensorFrame({st.numerical: torch.from_numpy((rs.randn(N, ncol) * 0.11 - 0.1).astype(np.float32))}, names_n)
    edge_tf = T.TensorFrame({st.relation: torch.ones(E, 1)}, {st.relation: ["edge_attr"]})
    y = torch.from_numpy(rs.randint(0, 40, N))
    return node_tf.to(device), torch.from_numpy(ei).to(device), edge_tf.to(device), y.to(device)


def leg_tabgnn_arxiv(cdt, dev, steps=5, warmup=2):
    """BASELINE configs[3]: `tabgnn` path (src/nn/models/tabgnn.py:100-151), node classification, S = 130 column
    attention over every sampled node row.  d = 128, 8 heads (reference default), 2 layers, B = 200 seed nodes."""
    import tabgnn_amd as T
    st = T.stype
    C = 128
    batches = [arxiv_batch(seed=s, device=dev) for s in range(2)]
    node_tf, ei, edge_tf, y = batches[0]
    names_n, names_e = node_tf.col_names_dict, edge_tf.col_names_dict
    stats_n = {n: dict(mean=-0.1, std=0.11) for n in names_n[st.numerical]}
    torch.manual_seed(7)
    cfg = dict(model="tabgnn", task="node_classification", batch_size=200, n_hidden=C, n_gnn_layers=2, n_classes=40,
               dropout=0.083, backbone_dropout=0.5, nhead=8, num_node_features=129, num_edge_features=1,
               in_degrees=torch.bincount(ei[1].cpu(), minlength=node_tf.num_rows), reverse_mp=False,
               node_encoder=T.StypeWiseFeatureEncoder(C, stats_n, names_n, cdt),
               edge_encoder=T.StypeWiseFeatureEncoder(C, {}, names_e, cdt))
    model = T.TABGNNS(cfg).to(dev).train()
    flat = T.FlatParams(model, shadow_dtype=cdt)
    opt = T.FusedAdam(flat, lr=6e-4)

    def one(i):
        node_tf, ei, edge_tf, y = batches[i % 2]
        T.ops.DropoutRNG.new_step()
        flat.zero_grad()
        out = model(node_tf, ei, edge_tf)
        T.ops.weighted_cross_entropy(out, y).backward()
        opt.step()
        return ei.shape[1], node_tf.num_rows
    for i in range(warmup):
        one(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    edges = rows = 0
    for i in range(steps):
        e, n = one(i)
        edges += e; rows += n + e
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return dict(workload="configs[3] ogbn-arxiv-shaped, tabgnn path, node classification: 129 numerical node columns "
                         "(S=130), 1 relation edge column (S=2), d=128, 8 heads, 2 FT + 2 PNA layers, 200 seed nodes, "
                         "fan-out 15/10, Adam", value=edges / dt, unit="edges/s", rows_per_sec=rows / dt,
                ms_per_step=1e3 * dt / steps, steps=steps, edges_per_step=edges // steps,
                nodes_per_step=(rows - edges) // steps, attention_tokens_per_step=(130 * (rows - edges) + 2 * edges) // steps,
                dtype="bf16" if cdt == torch.bfloat16 else "fp32")


def leg_wide64(cdt, dev, steps=5, warmup=2, B=512):
    """BASELINE configs[4] per-GPU shape: fused model on a 64-column mixed table (32 categorical with cardinalities
    log-uniform 2..10^4, 24 numerical, 8 timestamp), d = 256, S = 65; HI-Small-shaped sampled subgraphs."""
    import numpy as np
    import tabgnn_amd as T
    from tabgnn_amd import synthetic as S
    st = T.stype
    C = 256
    rs = np.random.RandomState(5)
    cards = [int(c) for c in np.exp(rs.uniform(np.log(2), np.log(1e4), 32))]
    names = {st.numerical: [f"n{i}" for i in range(24)], st.categorical: [f"c{i}" for i in range(32)],
             st.timestamp: [f"t{i}" for i in range(8)]}
    stats = {**{f"n{i}": dict(mean=0.0, std=1.0) for i in range(24)},
             **{f"c{i}": dict(cardinality=cards[i]) for i in range(32)}, **{f"t{i}": dict(min_year=2015) for i in range(8)}}

    def batch(seed):
        ei, N = S.sampled_subgraph(B, seed)
        E = ei.shape[1]
        r = np.random.RandomState(seed + 1)
        cat = np.stack([r.randint(0, c, E) for c in cards], 1).astype(np.int64)
        num = r.randn(E, 24).astype(np.float32)
        ts = np.stack([r.randint(2015, 2024, (E, 8)), r.randint(0, 12, (E, 8)), r.randint(0, 28, (E, 8)),
                       r.randint(0, 7, (E, 8)), r.randint(0, 24, (E, 8)), r.randint(0, 60, (E, 8)),
                       r.randint(0, 60, (E, 8))], -1).astype(np.int64)
        etf = T.TensorFrame({st.numerical: torch.from_numpy(num), st.categorical: torch.from_numpy(cat),
                             st.timestamp: torch.from_numpy(ts)}, names)
        ntf = T.TensorFrame({st.relation: torch.ones(N, 1)}, S.NODE_COLS)
        y = torch.from_numpy((r.rand(B) < 0.05).astype(np.int64))
        return ntf.to(dev), torch.from_numpy(ei).to(dev), etf.to(dev), y.to(dev)
    batches = [batch(s) for s in range(2)]
    torch.manual_seed(9)
    cfg = dict(model="tabgnnfused", task="edge_classification", batch_size=B, n_hidden=C, n_gnn_layers=2, n_classes=2,
               dropout=0.083, backbone_dropout=0.5, nhead=8, num_node_features=1, num_edge_features=64,
               in_degrees=S.in_degrees_like(), reverse_mp=False, load_model=None, checkpoint=False,
               node_encoder=T.StypeWiseFeatureEncoder(C, {}, S.NODE_COLS, cdt),
               edge_encoder=T.StypeWiseFeatureEncoder(C, stats, names, cdt))
    model = T.TABGNNFusedS(cfg).to(dev).train()
    flat = T.FlatParams(model, shadow_dtype=cdt)
    opt = T.FusedAdam(flat, lr=6e-4)
    lw = torch.tensor([1.0, 9.23], device=dev)
    for i in range(warmup):
        T.train_step(model, flat, opt, batches[i % 2], lw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    edges = 0
    for i in range(steps):
        T.train_step(model, flat, opt, batches[i % 2], lw)
        edges += batches[i % 2][1].shape[1]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return dict(workload=f"configs[4] per-GPU shape: fused supervised on a 64-column mixed table (32 cat up to 10^4 "
                         f"categories, 24 num, 8 ts), d=256, S=65, 8 heads, 2 layers, B={B} seed edges, Adam",
                value=edges / dt, unit="edges/s", ms_per_step=1e3 * dt / steps, steps=steps, edges_per_step=edges // steps,
                attention_tokens_per_step=65 * (edges // steps), dtype="bf16" if cdt == torch.bfloat16 else "fp32")


def leg_wide64_graph(cdt, dev, steps=6, warmup=3, B=512, N=10_000_000, E=100_000_000, fanout=(10, 5)):
    """BASELINE configs[4] ON ITS GRAPH: the synthetic 10 M-node / 100 M-edge graph and its 64-column raw
    table (80 GB) generated and kept in ONE GPU's HBM (``ColumnStore``), the device sampler drawing B seed edges' 2-hop
    neighbourhoods from the CSC (0.8 GB) one batch ahead on a side stream (``DeviceBatchLoader``), the fused model at
    d = 256 training on those batches — ids only, the stype encoders read the raw rows from the resident table."""
    import tabgnn_amd as T
    from tabgnn_amd import synthetic as S
    from tabgnn_amd import DeviceBatchLoader, DeviceNeighborSampler
    t0 = time.perf_counter()
    ei = S.powerlaw_graph_on_device(N, E, dev)
    store = S.wide64_store_on_device(N, E, dev)
    deg = torch.bincount(ei[1], minlength=N).cpu()
    smp = DeviceNeighborSampler(ei, N, fanout, dev)
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t0
    resident_gb = torch.cuda.memory_allocated(dev) / 1e9
    torch.manual_seed(9)
    model = T.TABGNNFusedS(S.wide64_config(B, deg, cdt)).to(dev).train()
    flat = T.FlatParams(model, shadow_dtype=cdt)
    opt = T.FusedAdam(flat, lr=6e-4)
    lw = torch.tensor([1.0, 9.23], device=dev)
    g = torch.Generator(device="cpu"); g.manual_seed(3)
    total = steps + warmup
    seeds = [torch.randint(0, E, (B,), generator=g) for _ in range(total)]
    loader = DeviceBatchLoader(smp, store, seeds, mode="index", rng_seed=1)
    edges = nodes = 0
    for i, batch in enumerate(loader):
        if i == warmup:
            torch.cuda.synchronize(); t1 = time.perf_counter(); edges = nodes = 0
        loss, _ = T.train_step(model, flat, opt, batch, lw)
        edges += batch[1].edge_index.shape[1]
        nodes += batch[0].num_rows
    torch.cuda.synchronize()
    dt = time.perf_counter() - t1
    return dict(graph="10M/100M", nodes=N, edges=E, fanout=list(fanout), B=B, value=edges / dt, unit="edges/s",
                ms_per_step=1e3 * dt / steps, steps=steps, edges_per_step=edges // steps, nodes_per_step=nodes // steps,
                resident_table_and_graph_GB=resident_gb, peak_hbm_GB=torch.cuda.max_memory_allocated(dev) / 1e9,
                build_s=t_build, final_loss=float(loss), sampler_threads=0,
                what="configs[4] on its graph: 10 M nodes / 100 M edges, 64-column raw table (800 B/row) resident in one "
                     "GPU's HBM, DeviceBatchLoader -> fused d=256 train step")


# ----------------------------------------------------------------------------------------------- step roofline (8d)

MFMA_PEAK_TFLOPS = 2500.0      # bf16 dense (MI355X_MICROARCH.md)


def step_roofline(E, N, B, S, C, F, L, b, n_params, ms_per_step):
    """edges/s roofline of the whole step = E / sum_op max(bytes_op / BW, flops_op / MFMA), every operator
    reading its inputs and writing its outputs ONCE (forward and backward listed separately; backward of a Linear = two
    products).  b = bytes per activation element.  Shapes: R_e = E - B neighbour rows and B seed rows of S column tokens."""
    En, D = E - B, C + 2 * F
    ops_ = []

    def op(name, bytes_, flops=0.0):
        t_b, t_f = bytes_ / (HBM_PEAK_GBS * 1e9), flops / (MFMA_PEAK_TFLOPS * 1e12)
        ops_.append(dict(op=name, bytes=float(bytes_), flops=float(flops), us=1e6 * max(t_b, t_f),
                         bound="hbm" if t_b >= t_f else "mfma"))
    ncols = S - 1
    raw = 3 * 8 + 1 * 4 + 1 * 56
    op("stype encoders fwd (E rows)", E * (raw + ncols * C * b))
    op("stype encoders bwd", E * (raw + ncols * C * b))
    for rows, tag, times in ((En, "edge rows", 1), (B, "seed rows", 1 + L)):
        fl = rows * (12 * S * C * C + 4 * S * S * C)
        op(f"column attention layer fwd ({tag}) x{times}", times * 2 * rows * S * C * b, times * fl)
        op(f"column attention layer bwd ({tag}) x{times}", times * 3 * rows * S * C * b, times * 2 * fl)
    op("edge_emb fwd (E_n + B rows)", E * (S * C + F) * b, 2 * E * S * C * F)
    op("edge_emb bwd", E * (2 * S * C + F) * b, 4 * E * S * C * F)
    op("node_emb fwd+bwd", 3 * N * F * b, 0)
    for l in range(L):
        op(f"L{l} PNA message gather-GEMM fwd", En * (3 * F + F) * b + En * 8, En * 8 * F * F)
        op(f"L{l} PNA message bwd", En * (3 * F + F + 3 * F) * b + N * F * b, 2 * En * 8 * F * F)
        op(f"L{l} PNA aggregation fwd", En * (F * b + 4) + N * 4 * F * b)
        op(f"L{l} PNA aggregation bwd", N * 8 * F * b + En * (2 * F * b + 4))
        op(f"L{l} PNA post projection fwd", N * (5 * F + F) * b, N * 28 * F * F)
        op(f"L{l} PNA post projection bwd", N * (F + 5 * F + 5 * F) * b, 2 * N * 28 * F * F)
        op(f"L{l} BatchNorm+ReLU+residual fwd", N * 4 * F * b)
        op(f"L{l} BatchNorm bwd", N * 5 * F * b)
        op(f"L{l} edge update gather-MLP fwd", En * (3 * F + F + F) * b, En * 8 * F * F)
        op(f"L{l} edge update bwd", En * (3 * F + 2 * F + 3 * F) * b + N * F * b, 2 * En * 8 * F * F)
        op(f"L{l} fuse MLP fwd+bwd (B rows)", 3 * 36 * D * D * 2 + 12 * B * D * b, 3 * 48 * B * D * D)
        op(f"L{l} seed pooling fwd+bwd", 8 * B * F * b)
    op("head + loss fwd+bwd", 6 * B * 3 * F * b, 3 * 2 * B * (3 * F * 50 + 50 * 25 + 25 * 2))
    op("Adam (16 B read + 14 B written per parameter)", 30 * n_params)
    t = sum(o["us"] for o in ops_) * 1e-6
    return dict(edges_per_sec_bound=E / t, ms_per_step_bound=1e3 * t, frac=(t * 1e3) / ms_per_step,
                peaks=dict(hbm_GBs=HBM_PEAK_GBS, mfma_bf16_TFLOPs=MFMA_PEAK_TFLOPS),
                total_bytes=sum(o["bytes"] for o in ops_), total_flops=sum(o["flops"] for o in ops_), ops=ops_,
                definition="E / sum_op max(bytes_op/BW, flops_op/MFMA): each operator reads its inputs and writes its "
                           "outputs once, no traffic for intermediates inside an operator (SURVEY 8d)")


def kernel_source_sha():
    """sha256 over the kernel sources of this tree (the GPU box has no .git): what a committed PMC summary is checked
    against (tools/publish_profiles.py stamps the same hash into the file)."""
    import glob, hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(PKG, "csrc", "*"))):
        h.update(os.path.basename(f).encode()); h.update(open(f, "rb").read())
    return h.hexdigest()


def pmc_profile(args, E_mean):
    """Counter-derived HBM bytes per launch from the newest committed PMC summary that matches this workload
    (profiles/rNN_pmc_hbm_traffic.json; collected by separate rocprofv3 --pmc passes, never in this run).  A summary
    collected from OTHER kernel sources than this tree's is refused (returns its path and the reason; `traffic` stays
    null): VERDICT r03 found the r03 file one commit behind the kernels it was quoted for."""
    import glob
    sha = kernel_source_sha()
    stale = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm_traffic.json")), reverse=True):
        try:
            pmc = json.load(open(path))
            wl = pmc["workload"]
            if (wl["batch_size"], wl["F"], wl["dtype"]) == (args.batch_size, args.hidden, args.dtype) and abs(wl["E"] - E_mean) < 1:
                if pmc.get("kernel_source_sha256") != sha:
                    stale = stale or (os.path.relpath(path, ROOT) + ": collected from other kernel sources (" +
                                      str(pmc.get("commit", "no commit stamp"))[:12] + ") than this tree's: not quoted")
                    continue
                return os.path.relpath(path, ROOT), pmc, None
        except Exception:
            continue
    return None, None, stale


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal of the multi-rank control flow on ONE card (gpurun boxes have one GPU, RCCL refuses two ranks on one
    # device): TABGNN_DIST_BACKEND=gloo TABGNN_ONE_DEVICE=1 python -m torch.distributed.run --nproc-per-node 2 bench.py
    backend = os.environ.get("TABGNN_DIST_BACKEND", "nccl")
    if os.environ.get("TABGNN_ONE_DEVICE") == "1":
        local_rank = 0
    use_dist = world > 1 or os.environ.get("TABGNN_FORCE_ALLREDUCE") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist.init_process_group(backend)
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    torch.cuda.set_device(local_rank)
    dev = torch.device(f"cuda:{local_rank}")

    import tabgnn_amd as T
    from tabgnn_amd import ops
    from tabgnn_amd import synthetic as S
    from tabgnn_amd import _lib
    _lib.call("tg_device_check")

    cdt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    if args.workload == "reference-batch-graph":
        print(json.dumps(reference_batch_graph(args, cdt, dev)))
        return
    if args.workload == "reference-sampled-loop":
        cfg = S.make_config(args.hidden, args.layers, args.nhead, 200, compute_dtype=cdt)
        print(json.dumps(reference_batch_sampled_loop(cfg, cdt, dev, 200)))
        return
    if args.workload != "aml-fused":          # one extra leg alone (profiling aid): prints that leg's object
        leg = {"tabgnn-arxiv": leg_tabgnn_arxiv, "wide64-c256": leg_wide64, "wide64-graph": leg_wide64_graph}[args.workload]
        print(json.dumps(leg(cdt, dev, steps=args.steps, warmup=args.warmup)))
        return
    torch.manual_seed(1234)
    cfg = S.make_config(args.hidden, args.layers, args.nhead, args.batch_size, compute_dtype=cdt)
    cfg["reverse_mp"] = bool(args.reverse_mp)
    model = T.TABGNNFusedS(cfg).to(dev).train()
    flat = T.FlatParams(model, shadow_dtype=cdt)
    opt = T.FusedAdam(flat, lr=cfg["lr"])
    ddp = T.DataParallel(model, flat) if use_dist else None
    if ddp is not None and os.environ.get("TABGNN_NO_ALLREDUCE_OVERLAP") != "1":
        # bucket-ready exchange: the head's and the later layers' gradient ranges go out from autograd pre-hooks while the
        # earlier layers' backward still runs (train.DataParallel.enable_overlap); layer 0 + encoders after the backward
        ddp.enable_overlap(list(model.model.backbone) + [model.decoder])
    loss_w = torch.tensor(cfg["loss_weights"], device=dev)

    batches = [S.make_batch(args.batch_size, seed=42 + rank * 1000 + i, device=dev)
               for i in range(args.distinct_batches)]
    plain_batches = batches
    if args.index == "sampler":     # the batch as the sampler hands it over: ids + host-built CSRs (ops.BatchIndex)
        from tabgnn_amd.sampler import batch_index
        batches = [(b[0], batch_index(b[1].cpu(), b[0].num_rows, args.batch_size, dev), b[2], b[3]) for b in batches]
    E_mean = sum(b[1].shape[1] for b in batches) / len(batches)
    N_mean = sum(b[0].num_rows for b in batches) / len(batches)

    loss_log = [] if os.environ.get("TABGNN_BENCH_LOSSES") == "1" else None      # (tests: the loss trajectory, read AFTER the timed region)

    def run(n, first):
        edges = 0
        for i in range(n):
            b = batches[(first + i) % len(batches)]
            loss, _ = T.train_step(model, flat, opt, b, loss_w, ddp)
            if loss_log is not None:
                loss_log.append(loss)
            edges += b[1].shape[1]
        return edges

    run(args.warmup, 0)
    timer = ops.KernelTimer(only=("tg_pna_aggregate_fwd", "tg_pna_aggregate_bwd"))   # 4 event pairs per step
    ops.KernelTimer.active = timer
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    edges = run(args.steps, args.warmup)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ops.KernelTimer.active = None
    peak_hbm_gb = torch.cuda.max_memory_allocated(dev) / 1e9          # model + optimiser + one step's activations

    tot = torch.tensor([elapsed, float(edges)], dtype=torch.float64, device=dev)
    if use_dist:
        tmax = tot.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = tot.clone(); dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        elapsed, edges = float(tmax[0]), float(tsum[1])
    if rank != 0:
        dist.destroy_process_group()
        return

    # roofline of the PNA multi-aggregation forward: E_n*(F*b + 4) read + N*4F*b written per launch
    b_act = 2 if cdt == torch.bfloat16 else 4
    F = args.hidden
    En = E_mean - args.batch_size
    agg_bytes = En * (F * b_act + 4) + N_mean * 4 * F * b_act
    agg_ms = timer.mean_ms("tg_pna_aggregate_fwd")
    achieved = agg_bytes / (agg_ms * 1e-3) / 1e9
    rows = E_mean + args.layers * args.batch_size
    pmc_path, pmc, pmc_stale = pmc_profile(args, E_mean)
    traffic = None
    if pmc is not None:
        key = [k for k in pmc["kernels"] if "k_pna_aggregate_fwd" in k]
        traffic = pmc["kernels"][key[0]]["hbm_bytes_per_launch"] if key else None
    ms_per_step = elapsed / args.steps * 1e3
    out = {
        "metric": "edges/sec per training step, fused AML supervised (TABGNNFused fwd+CE+bwd+Adam)",
        "value": edges / elapsed, "unit": "edges/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        # (the driver keeps `config` whole but cuts every string at 120 characters: facts travel as short strings and numbers)
        "config": {"workload": f"configs[1] HI-Small-shaped AML, fused supervised, d={args.hidden} {args.nhead}-head + "
                               f"{args.layers}-layer PNA, B={args.batch_size}" + (", reverse_mp" if args.reverse_mp else ""),
                   "columns": "5 edge columns (3 cat, 1 num, 1 ts), 1 node column", "dropout": [0.5, 0.083], "optimizer": "Adam",
                   "hidden": args.hidden, "nhead": args.nhead, "layers": args.layers,
                   "batch_size": args.batch_size, "edges_per_step": int(E_mean), "nodes_per_step": int(N_mean),
                   "spmm_frac_synthetic": achieved / HBM_PEAK_GBS,
                   "rows_per_sec": rows * args.steps * world / elapsed, "parallelism": f"dp{world}",
                   "peak_hbm_gb": peak_hbm_gb,
                   "index": ("batch CSRs built by the sampler side (host), resident with the batch" if args.index == "sampler"
                             else "batch CSRs rebuilt on the device inside every forward (tg_csr_build)")},
        "roofline": {"kernel": "k_pna_aggregate_fwd", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "traffic_source": (f"{pmc_path}: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                                        "command on this workload (committed file, NOT measured in this run)")
                     if traffic is not None else pmc_stale,
                     "algorithmic_bytes_per_launch": agg_bytes, "avg_launch_ms": agg_ms,
                     "launches_timed": timer.count("tg_pna_aggregate_fwd"),
                     "bwd_avg_launch_ms": timer.mean_ms("tg_pna_aggregate_bwd")},
    }
    if loss_log is not None:
        out["losses"] = [float(v) for v in loss_log]
    if use_dist:
        out["collective"] = {"backend": backend, "world": world, "all_reduce_calls": ddp.calls,
                             "bytes_per_step": ddp.bytes / max(1, args.steps + args.warmup), "bucket_MiB": 32,
                             "overlapped": bool(getattr(ddp, "overlapped_bytes", 0)),
                             "overlapped_bytes_per_step": getattr(ddp, "overlapped_bytes", 0) / max(1, args.steps + args.warmup),
                             "what": "flat fp32 gradient buffer, bucketed async all_reduce(SUM), 1/world folded into Adam; "
                                     "later stages' ranges issued from autograd pre-hooks during the backward"}
    extras = not args.no_extras and world == 1
    if extras:
        out["roofline"].update(measured_copy_GBs=copy_rate_gbs(dev), measured_stream_GBs=stream_rate_gbs(dev))
        out["roofline"]["frac_of_measured_copy"] = achieved / out["roofline"]["measured_copy_GBs"]
        out["roofline"]["frac_of_measured_stream"] = achieved / out["roofline"]["measured_stream_GBs"]
        n_params = sum(p.numel() for p in model.parameters())
        out["step_roofline"] = step_roofline(E_mean, N_mean, args.batch_size, 6, args.hidden, args.hidden, args.layers,
                                             b_act, n_params, ms_per_step)
    # the kernels that dominate the step by TIME, measured the same way (HIP events on the launching stream, algorithmic
    # bytes per launch) in a few extra steps AFTER the timed region: ~80 more event pairs per step inside it would cost
    # the headline number ~0.5 ms
    if extras and args.dtype == "bf16":
        names = ("tg_gemm_nt_bf16", "tg_gemm_tn_bf16", "tg_gemm_nt_gather3_bf16", "tg_gemm_tn_gather3_bf16",
                 "tg_encoder_fwd_bf16", "tg_encoder_bwd_ffn_bf16", "tg_encoder_bwd_ffn_dw_bf16", "tg_encoder_bwd_attn_bf16")
        gt = ops.KernelTimer(only=names)
        ops.KernelTimer.active = gt
        run(3, args.warmup + args.steps)
        torch.cuda.synchronize()
        ops.KernelTimer.active = None
        gemms = {}
        for name in names:
            if gt.count(name):
                gemms[name] = {"launches_per_step": gt.count(name) / 3, "ms_per_step": gt.total_ms(name) / 3,
                               "achieved": gt.gbs(name), "frac": gt.gbs(name) / HBM_PEAK_GBS,
                               "algorithmic_bytes_per_step": gt.nbytes[name] / 3}
        # The column-transformer kernels are matrix-instruction-issue bound, not HBM bound: v_mfma_f32_32x32x16
        # instructions per 32-token wave tile (counted in the compiled kernels; the attention backward = 4 head blocks x 45
        # + 96 of the d_x epilogue), 32 cycles each, 1 024 SIMDs at the 2.4 GHz peak clock.
        mfma_per_tile = {"tg_encoder_fwd_bf16": 212, "tg_encoder_bwd_ffn_dw_bf16": 224, "tg_encoder_bwd_ffn_bf16": 96,
                         "tg_encoder_bwd_attn_bf16": 276}
        for name, n_mfma in mfma_per_tile.items():
            if name in gemms and gt.units.get(name):
                tiles = gt.units[name] / 3
                bound_ms = tiles * n_mfma * 32 / 1024 / 2.4e9 * 1e3
                gemms[name]["mfma_issue"] = {"mfma_per_tile": n_mfma, "tiles_per_step": tiles, "bound_ms_per_step": bound_ms,
                                             "frac": bound_ms / gemms[name]["ms_per_step"]}
        if pmc is not None:      # counter bytes / algorithmic bytes of the weight-gradient kernel (incl. its slab reduction)
            kk = pmc["kernels"]
            # kernels behind tg_gemm_tn_bf16: the 128x128-tile kernel and the two unscaled, ungathered wide variants; the
            # slab reduction is shared with the scaled / gathered entry points: its bytes are split by launch count
            tn = [kk[k] for k in kk if "k_gemm_tn_bf16<false>" in k or "k_gemm_tn_wide<false, true, false>" in k
                  or "k_gemm_tn_wide<false, false, false>" in k]
            oth = [kk[k] for k in kk if "k_gemm_tn_wide<true" in k or "k_gemm_tn_wide<false, false, true>" in k
                   or "k_gemm_tn_bf16<true>" in k]
            sl = [kk[k] for k in kk if "k_sum_slabs" in k]
            if tn and "tg_gemm_tn_bf16" in gemms:
                n_tn, n_oth = sum(t["launches"] for t in tn), sum(t["launches"] for t in oth)
                cnt = sum(t["hbm_bytes_per_launch"] * t["launches"] for t in tn)
                if sl:
                    cnt += sl[0]["hbm_bytes_per_launch"] * sl[0]["launches"] * n_tn / max(n_tn + n_oth, 1)
                steps_profiled = pmc.get("steps_profiled", 6)
                gemms["tg_gemm_tn_bf16"]["counter_over_algorithmic"] = cnt / steps_profiled / gemms["tg_gemm_tn_bf16"]["algorithmic_bytes_per_step"]
                gemms["tg_gemm_tn_bf16"]["counter_source"] = pmc_path
        if gemms:
            out["roofline_gemm"] = {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "kernels": gemms}
    if extras and args.index == "sampler":
        # the same step with the CSRs rebuilt from edge_index on the device inside every forward (tg_csr_build): what
        # `value` measured in round 1 and what the CPU oracle's step includes
        batches_s, batches = batches, plain_batches
        run(3, 0)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        e_dev = run(10, 3)
        torch.cuda.synchronize()
        d_dev = time.perf_counter() - t1
        batches = batches_s
        out["index_device"] = {"value": e_dev / d_dev, "unit": "edges/s", "ms_per_step": 1e3 * d_dev / 10, "steps": 10,
                               "what": "same workload with --index device: the batch's CSRs built inside the forward"}
    if extras:
        out["eval"] = eval_throughput(model, plain_batches, dev)
        out["reference_batch"] = reference_batch(args, cdt, dev)
        if args.dtype == "bf16":
            out["other_workloads"] = {"tabgnn-arxiv": leg_tabgnn_arxiv(cdt, dev), "wide64-c256": leg_wide64(cdt, dev)}
            # configs[4] on its 10 M-node / 100 M-edge graph with the 80 GB raw table resident: a CHILD process (its 85 GB
            # must not sit beside this process's allocations, and a failure there must not take the bench line down)
            out["other_workloads"]["wide64-c256"]["on_graph"] = _child_leg(["--workload", "wide64-graph", "--steps", "6", "--warmup", "3"])
    if extras and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(model.state_dict(), args.nhead, args.cpu_batch_size, args.cpu_steps,
                                           cfg["lr"], cfg["loss_weights"])
    if extras and not args.no_e2e and args.dtype == "bf16":
        out["end_to_end"] = end_to_end(model, flat, opt, loss_w, args.batch_size, args.e2e_steps, dev)
        out["roofline_sampled"] = out["end_to_end"].pop("roofline_sampled")
        # (the driver keeps `config` whole but only the key names of extra objects: the aggregation's fraction on the
        # batch the package's own sampler draws — fewer, denser destination rows — travels as a numeric config key)
        rsf = out["roofline_sampled"].get("frac")
        if rsf is not None:
            out["config"]["spmm_frac_sampled"] = rsf       # the named kernel on the batch the package's own sampler draws
    if extras and args.dtype == "bf16":
        out["fp32_twin"] = fp32_twin(args)
    print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
