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
    ap.add_argument("--cpu-batch-size", type=int, default=256)
    ap.add_argument("--cpu-steps", type=int, default=3)
    ap.add_argument("--reverse-mp", action="store_true", help="PNAConvHetero (forward + reverse message passing); "
                    "not the BASELINE configuration")
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
    from tabgnn_amd.sampler import ColumnStore, NeighborSampler
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
            t0 = time.perf_counter()
            out = samplers[w].sample(seeds[i], i)
            t_sample[i] = time.perf_counter() - t0
            slots[i].put(out)

    for w in range(n_workers):
        threading.Thread(target=work, args=(w,), daemon=True).start()
    edges = 0
    for i in range(total):
        eid, lei, nodes = slots[i].get()
        ahead.release()
        if i == warm:
            torch.cuda.synchronize(); t0 = time.perf_counter(); edges = 0
        eid_d, nodes_d = eid.to(dev, non_blocking=True), nodes.to(dev, non_blocking=True)
        edge_tf = T.frame.TensorFrame({k: v.index_select(0, eid_d) for k, v in store.edge_feats.items()}, store.edge_cols)
        node_tf = T.frame.TensorFrame({k: v.index_select(0, nodes_d) for k, v in store.node_feats.items()}, store.node_cols)
        y = store.labels.index_select(0, eid_d[:batch_size])
        T.train_step(model, flat, opt, (node_tf, lei.to(dev, non_blocking=True), edge_tf, y), loss_w)
        edges += eid.numel()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return dict(value=edges / dt, unit="edges/s", ms_per_step=1e3 * dt / steps, steps=steps,
                edges_per_step=edges / steps, sampler_ms_per_batch=1e3 * float(np.mean(t_sample[warm:])),
                sampler=f"libtabgnn_sampler.so k-hop [100,100], {n_workers} host threads (one handle each), prefetch <= "
                        f"{2 * n_workers} batches",
                graph="synthetic HI-Small-shaped: 515080 nodes, 5078345 edges, raw columns resident in HBM")


def cpu_baseline(model_sd, nhead, bs, steps, lr, loss_w):
    """Oracle (CPU restatement) train step timed on the host cores: bounded sample of the same workload."""
    from oracle import step as ostep
    from tabgnn_amd import synthetic as S
    sd = {k: v.detach().float().cpu().clone() for k, v in model_sd.items()}
    cores = host_cores()
    torch.set_num_threads(cores)
    opt_state = {}
    times, edges = [], 0
    for i in range(steps + 1):
        node_tf, ei, edge_tf, y = S.make_batch(bs, seed=900 + i)
        nf = {k.value: v for k, v in node_tf.feat_dict.items()}
        ef = {k.value: v for k, v in edge_tf.feat_dict.items()}
        t0 = time.perf_counter()
        ostep.train_step(sd, opt_state, nhead, bs, nf, ei, ef, y, torch.tensor(loss_w), lr, p_backbone=0.5,
                         p_head=0.083)
        dt = time.perf_counter() - t0
        if i > 0:                      # first step = warm-up
            times.append(dt)
            edges += ei.shape[1]
    return dict(value=edges / sum(times), unit="edges/s", cores=cores, kind="port",
                sample=f"{steps} oracle train steps (fp32, dropout on) at B={bs} seed edges "
                       f"(E~{int(edges / steps)} sampled edges/step), after 1 warm-up step")


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
    torch.manual_seed(1234)
    cfg = S.make_config(args.hidden, args.layers, args.nhead, args.batch_size, compute_dtype=cdt)
    cfg["reverse_mp"] = bool(args.reverse_mp)
    model = T.TABGNNFusedS(cfg).to(dev).train()
    flat = T.FlatParams(model, shadow_dtype=cdt)
    opt = T.FusedAdam(flat, lr=cfg["lr"])
    ddp = T.DataParallel(model, flat) if use_dist else None
    loss_w = torch.tensor(cfg["loss_weights"], device=dev)

    batches = [S.make_batch(args.batch_size, seed=42 + rank * 1000 + i, device=dev)
               for i in range(args.distinct_batches)]
    E_mean = sum(b[1].shape[1] for b in batches) / len(batches)
    N_mean = sum(b[0].num_rows for b in batches) / len(batches)

    def run(n, first):
        edges = 0
        for i in range(n):
            b = batches[(first + i) % len(batches)]
            T.train_step(model, flat, opt, b, loss_w, ddp)
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
    # HBM bytes per launch from the committed PMC passes (profiles/), valid only for the workload they were taken on
    traffic = None
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_hbm_traffic.json")))
        wl = pmc["workload"]
        if (wl["batch_size"], wl["F"], wl["dtype"]) == (args.batch_size, F, args.dtype) and abs(wl["E"] - E_mean) < 1:
            key = [k for k in pmc["kernels"] if "k_pna_aggregate_fwd" in k][0]
            traffic = pmc["kernels"][key]["hbm_bytes_per_launch"]
    except Exception:
        traffic = None
    copy_gbs = copy_rate_gbs(dev)
    stream_gbs = stream_rate_gbs(dev)
    out = {
        "metric": "edges/sec per training step, fused AML supervised (TABGNNFused fwd+CE+bwd+Adam)",
        "value": edges / elapsed, "unit": "edges/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"HI-Small-shaped AML sampled subgraphs, fused supervised (configs[1]): d={args.hidden}, "
                               f"{args.nhead}-head FT-Transformer + {args.layers}-layer PNA, B={args.batch_size} seed "
                               f"edges/step/GPU, E={int(E_mean)} sampled edges, N={int(N_mean)} nodes, 5 edge columns "
                               f"(3 cat, 1 num, 1 ts), dropout 0.5/0.083, Adam" + (", reverse_mp" if args.reverse_mp else ""),
                   "batch_size": args.batch_size, "edges_per_step": int(E_mean), "nodes_per_step": int(N_mean),
                   "rows_per_sec": rows * args.steps * world / elapsed, "parallelism": f"dp{world}"},
        "roofline": {"kernel": "k_pna_aggregate_fwd", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "measured_copy_GBs": copy_gbs, "frac_of_measured_copy": achieved / copy_gbs,
                     "measured_stream_GBs": stream_gbs, "frac_of_measured_stream": achieved / stream_gbs,
                     "algorithmic_bytes_per_launch": agg_bytes, "avg_launch_ms": agg_ms,
                     "launches_timed": timer.count("tg_pna_aggregate_fwd"),
                     "bwd_avg_launch_ms": timer.mean_ms("tg_pna_aggregate_bwd")},
    }
    # the kernels that dominate the step by TIME, measured the same way (HIP events on the launching stream,
    # algorithmic bytes R*(K+N)*2 (+ gate / accumulate reads) and R*(M+N)*2 per launch) in a few extra steps AFTER the
    # timed region: 75 more event pairs per step would cost the headline number ~0.5 ms
    gemms = {}
    if world == 1 and args.dtype == "bf16":
        gt = ops.KernelTimer(only=("tg_gemm_nt_bf16", "tg_gemm_tn_bf16"))
        ops.KernelTimer.active = gt
        run(3, args.warmup + args.steps)
        torch.cuda.synchronize()
        ops.KernelTimer.active = None
        for name in gt.only:
            if gt.count(name):
                gemms[name] = {"launches_per_step": gt.count(name) / 3, "ms_per_step": gt.total_ms(name) / 3,
                               "achieved": gt.gbs(name), "frac": gt.gbs(name) / HBM_PEAK_GBS}
    if gemms:
        out["roofline_gemm"] = {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "kernels": gemms}
    if not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline(model.state_dict(), args.nhead, args.cpu_batch_size, args.cpu_steps,
                                           cfg["lr"], cfg["loss_weights"])
    if not args.no_e2e and world == 1 and args.dtype == "bf16":
        out["end_to_end"] = end_to_end(model, flat, opt, loss_w, args.batch_size, args.e2e_steps, dev)
    print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
