"""Shape-bucketed HIP-graph replay of the train step (tabgnn_amd/graph_step.py; reference loop main.py:41-75).

1. padding changes nothing a real row sees: the padded batch's seed logits and every weight gradient equal the plain
   batch's (up to the summation order of the BatchNorm statistics),
2. N graph replays == N eager runs of the same body, bit for bit (weights, Adam moments, BatchNorm running statistics),
   with batches of different true size sharing one bucket and the dropout seed / step count on the device.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _model(B, dtype, seed=7, dropout=0.5):
    import tabgnn_amd as T
    from tabgnn_amd import synthetic as S
    torch.manual_seed(seed)
    cfg = S.make_config(128 if dtype == torch.bfloat16 else 32, 2, 4, B, backbone_dropout=dropout,
                        head_dropout=0.083 if dropout else 0.0, compute_dtype=dtype)
    model = T.TABGNNFusedS(cfg).to(DEV).train()
    flat = T.FlatParams(model, shadow_dtype=dtype)
    opt = T.FusedAdam(flat, lr=cfg["lr"])
    return model, flat, opt, torch.tensor(cfg["loss_weights"], device=DEV)


def _resize(batch, extra_nodes, drop_edges):
    """The batch with ``extra_nodes`` isolated nodes appended and without its last ``drop_edges`` neighbour edges."""
    import tabgnn_amd as T
    node_tf, ei, edge_tf, y = batch
    E = ei.shape[1] - drop_edges
    feats = {k: torch.cat([v, v[:1].expand(extra_nodes, *v.shape[1:])]) for k, v in node_tf.feat_dict.items()}
    return (T.TensorFrame(feats, node_tf.col_names_dict), ei[:, :E].contiguous(), edge_tf[slice(0, E)], y)


@pytest.mark.parametrize("dtype,B", [(torch.float32, 48), (torch.bfloat16, 48), (torch.bfloat16, 1024)])
def test_padded_batch_gives_the_plain_batch_gradients(dtype, B):
    """The tie between a bucket's padded body and the unpadded step: same logits on the seed rows, same BatchNorm running
    statistics, and EVERY parameter gradient (reported per parameter; B = 1024 is the size of the bf16 parity tests, where
    the wide GEMM forms, the hub passes and several workgroups per kernel are in play)."""
    import tabgnn_amd as T
    from tabgnn_amd import synthetic as S, graph_step as G, _lib as L
    model, flat, opt, lw = _model(B, dtype)
    batch = S.make_batch(B, seed=11, device=DEV)
    T.ops.DropoutRNG.new_step(1234)
    flat.zero_grad()
    logits = model(batch[0], batch[1], batch[2])
    T.ops.weighted_cross_entropy(logits[:B], batch[3].view(-1), lw).backward()
    want_logits = logits[:B].detach().float().clone()
    want = {k: p.grad.detach().float().clone() for k, p in model.named_parameters()}
    want_rm = {k: v.clone() for k, v in model.named_buffers() if "running" in k}

    for k, b in model.named_buffers():           # undo the running-statistics update of the first pass
        if "running_mean" in k:
            b.zero_()
        elif "running_var" in k:
            b.fill_(1.0)
    E, N = batch[1].shape[1], batch[0].num_rows
    prep = G.prepare(batch, B, key=(E + 37 + B, N + 5 + B // 8))
    bucket = G._Bucket(prep, (batch[0], batch[2]), torch.device(DEV))
    bucket.load(prep)
    T.ops.DropoutRNG.new_step(1234)
    flat.zero_grad()
    T.ops.StepContext.set_bn_row_limit(bucket.static["n_real"])     # handed to the BatchNorm launches as their row_limit argument
    try:
        logits = model(bucket.node_tf, bucket.index(B), bucket.edge_tf)
        T.ops.weighted_cross_entropy(logits[:B], bucket.static["y"], lw).backward()
    finally:
        T.ops.StepContext.set_bn_row_limit(None)
    assert logits.shape[0] == B
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    np.testing.assert_allclose(logits[:B].detach().float().cpu().numpy(), want_logits.cpu().numpy(), rtol=tol, atol=tol)
    gscale = max(float(v.norm()) for v in want.values())
    # (bf16 at B = 48 the two runs take different small-problem GEMM routes: 0.5 %; at B = 1024 they agree to 1e-4)
    rel, abs_ = (1e-4, 1e-6) if dtype == torch.float32 else ((3e-2, 1e-3) if B < 1024 else (2e-3, 1e-4))
    rows = []
    for k, p in model.named_parameters():
        err, den = float((p.grad.float() - want[k]).norm()), float(want[k].norm())
        rows.append((err / (rel * den + abs_ * gscale), err / max(den, 1e-30), den / gscale, k))
    rows.sort(reverse=True)
    print(f"padded vs plain batch, {dtype}, B={B}: worst 5 (gate ratio, rel. error, ||g||/max||g||, name)")
    for r in rows[:5]:
        print("   %.3f  %.5f  %.2e  %s" % r)
    assert len(rows) > 90 and rows[0][0] <= 1.0, rows[:4]
    for k, v in model.named_buffers():
        if "running" in k:
            np.testing.assert_allclose(v.cpu().numpy(), want_rm[k].cpu().numpy(), rtol=1e-4, atol=1e-5)


def test_graph_replays_equal_eager_steps_bit_for_bit():
    import tabgnn_amd as T
    from tabgnn_amd import synthetic as S, graph_step as G
    B, steps = 64, 6
    batches = [_resize(S.make_batch(B, seed=40 + i, device=DEV), 9 * i, 40 * i) for i in range(3)]
    key = (G.bucket_size(max(b[1].shape[1] for b in batches)), G.bucket_size(max(b[0].num_rows for b in batches) + 1))
    preps = [G.prepare(b, B, key=key) for b in batches]
    frames = (batches[0][0], batches[0][2])
    assert len({(p.e_real, p.n_real) for p in preps}) > 1        # different true sizes, one bucket

    runs = {}
    for mode in ("eager", "graph"):
        model, flat, opt, lw = _model(B, torch.bfloat16, seed=3)
        step = G.GraphedTrainStep(model, flat, opt, lw, B)
        losses = []
        for i in range(steps):
            p = preps[i % 3]
            loss, _ = step.run_eager(p, frames) if mode == "eager" else step(p, frames)
            losses.append(loss.clone())
        torch.cuda.synchronize()
        runs[mode] = (torch.stack(losses).float().cpu(), flat.flat.clone().cpu(), opt.m.clone().cpu(), opt.v.clone().cpu(),
                      {k: v.clone().cpu() for k, v in model.named_buffers()}, step.state.buf.clone().cpu(), opt.t)
    e, g = runs["eager"], runs["graph"]
    assert e[6] == g[6] == steps
    assert torch.equal(e[5], g[5])                       # device step record: seed word, t, bias corrections
    assert torch.isfinite(e[0]).all() and float(e[0][0]) != float(e[0][3])     # same batch, another step: new masks, new weights
    assert torch.equal(e[0], g[0]), (e[0], g[0])
    for i in (1, 2, 3):
        assert torch.equal(e[i], g[i]), i
    for k in e[4]:
        assert torch.equal(e[4][k], g[4][k]), k


def test_graph_step_with_data_parallel_keeps_allreduce_and_adam_eager():
    """With a DataParallel the graph ends after the backward; the (RCCL) all-reduce and the Adam launch run eagerly
    behind the replay and read the device-resident step record.  World size 1: the result must equal the all-in-graph
    step bit for bit."""
    import os
    import torch.distributed as dist
    import tabgnn_amd as T
    from tabgnn_amd import synthetic as S, graph_step as G, ops
    B, steps = 64, 4
    batches = [_resize(S.make_batch(B, seed=60 + i, device=DEV), 5 * i, 30 * i) for i in range(2)]
    key = (G.bucket_size(max(b[1].shape[1] for b in batches)), G.bucket_size(max(b[0].num_rows for b in batches) + 1))
    preps = [G.prepare(b, B, key=key) for b in batches]
    frames = (batches[0][0], batches[0][2])
    import socket
    with socket.socket() as sk:                    # a free port: other tests of the suite start process groups too
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    saved_env = {k: os.environ.get(k) for k in ("MASTER_ADDR", "MASTER_PORT")}
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ["TABGNN_FORCE_ALLREDUCE"] = "1"
    seed0 = ops.DropoutRNG.seed
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    try:
        out = {}
        for mode in ("plain", "ddp"):
            model, flat, opt, lw = _model(B, torch.bfloat16, seed=5)
            ddp = T.DataParallel(model, flat) if mode == "ddp" else None
            ops.DropoutRNG.seed = seed0                        # (DataParallel derives a per-rank seed: same for rank 0 here)
            if ddp is not None:
                assert ddp.active
                ops.DropoutRNG.seed = seed0
            step = G.GraphedTrainStep(model, flat, opt, lw, B, ddp=ddp)
            step.host_seed = seed0
            for i in range(steps):
                loss, _ = step(preps[i % 2], frames)
            torch.cuda.synchronize()
            out[mode] = (flat.flat.clone().cpu(), opt.m.clone().cpu(), opt.t, float(loss), None if ddp is None else ddp.calls)
        assert out["ddp"][4] >= steps and out["plain"][2] == out["ddp"][2] == steps
        assert torch.equal(out["plain"][0], out["ddp"][0]) and torch.equal(out["plain"][1], out["ddp"][1])
    finally:
        os.environ.pop("TABGNN_FORCE_ALLREDUCE", None)
        for k, v in saved_env.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        ops.DropoutRNG.seed = seed0
        dist.destroy_process_group()


def test_sampled_lazy_batches_replay_over_several_buckets():
    """The whole loop of the reference at its default batch (ibm_transactions_for_aml.py:159-180 -> main.py:41-75):
    native sampler -> prepare_sample (ids + index parts, padded to a bucket) -> one upload -> replay, with the frames
    reading the HBM-resident tables by id.  Batches fall into several buckets; replays equal the eager runs of the same
    body bit for bit."""
    import tabgnn_amd as T
    from tabgnn_amd import synthetic as S, graph_step as G
    from tabgnn_amd.frame import stype
    from tabgnn_amd.sampler import ColumnStore, NeighborSampler
    rs = np.random.RandomState(3)
    N, E, B, steps = 4000, 30000, 32, 8
    ei = np.stack([rs.randint(0, N, E), rs.randint(0, N, E)])
    num, cat, ts = S.edge_table(E, 0)
    labels = torch.from_numpy((rs.rand(E) < 0.05).astype(np.int64))
    store = ColumnStore({stype.numerical: torch.from_numpy(num), stype.categorical: torch.from_numpy(cat),
                         stype.timestamp: torch.from_numpy(ts)}, S.EDGE_COLS,
                        {stype.relation: torch.ones(N, 1)}, S.NODE_COLS, labels).to(DEV)
    sampler = NeighborSampler(ei, N, (6, 4), num_threads=1)
    preps = []
    for i in range(steps):
        seeds = rs.choice(E, B, replace=False)
        eid, lei, nodes = sampler.sample(seeds, i)
        preps.append(G.prepare_sample(eid, lei, nodes, labels[eid[:B]], B))
    assert preps[0].arena.is_pinned() and len({p.key for p in preps}) >= 2
    frames = (T.TensorFrame(store.node_feats, store.node_cols, None, torch.zeros(1, dtype=torch.int64, device=DEV)),
              T.TensorFrame(store.edge_feats, store.edge_cols, None, torch.zeros(1, dtype=torch.int64, device=DEV)))
    runs = {}
    for mode in ("eager", "graph"):
        model, flat, opt, lw = _model(B, torch.bfloat16, seed=9)
        step = G.GraphedTrainStep(model, flat, opt, lw, B)
        losses = [(step.run_eager(p, frames) if mode == "eager" else step(p, frames))[0].clone() for p in preps]
        torch.cuda.synchronize()
        runs[mode] = (torch.stack(losses).float().cpu(), flat.flat.clone().cpu(), len(step.buckets))
    assert runs["graph"][2] == len({p.key for p in preps})
    assert torch.isfinite(runs["eager"][0]).all()
    assert torch.equal(runs["eager"][0], runs["graph"][0]) and torch.equal(runs["eager"][1], runs["graph"][1])


def test_graph_step_for_a_wrapper_that_takes_the_plain_edge_index():
    """``GNN(config)`` (utils.py:111-233, --model pna) only takes the int64 edge_index: with ``index=False`` the bucket's
    static edge_index goes in and the CSR kernels run inside the graph.  Replays == eager runs, bit for bit."""
    import tabgnn_amd as T
    from tabgnn_amd import synthetic as S, graph_step as G
    B, steps = 48, 4
    batches = [_resize(S.make_batch(B, seed=80 + i, device=DEV), 4 * i, 25 * i) for i in range(2)]
    key = (G.bucket_size(max(b[1].shape[1] for b in batches)), G.bucket_size(max(b[0].num_rows for b in batches) + 1))
    preps = [G.prepare(b, B, key=key) for b in batches]
    frames = (batches[0][0], batches[0][2])
    runs = {}
    for mode in ("eager", "graph"):
        torch.manual_seed(13)
        cfg = S.make_config(128, 2, 4, B, head_dropout=0.083, compute_dtype=torch.bfloat16)
        cfg.update(model="pna", emlps=True)
        model = T.GNN(cfg).to(DEV).train()
        flat = T.FlatParams(model, shadow_dtype=torch.bfloat16)
        opt = T.FusedAdam(flat, lr=1e-3)
        lw = torch.tensor([1.0, 9.23], device=DEV)
        step = G.GraphedTrainStep(model, flat, opt, lw, B, index=False)
        losses = [(step.run_eager(preps[i % 2], frames) if mode == "eager" else step(preps[i % 2], frames))[0].clone()
                  for i in range(steps)]
        torch.cuda.synchronize()
        runs[mode] = (torch.stack(losses).float().cpu(), flat.flat.clone().cpu())
    assert torch.isfinite(runs["eager"][0]).all() and float(runs["eager"][0][0]) != float(runs["eager"][0][2])
    assert torch.equal(runs["eager"][0], runs["graph"][0]) and torch.equal(runs["eager"][1], runs["graph"][1])


def test_two_graphed_steps_alive_in_one_process_do_not_share_state():
    """ABI v6: the seed word and the BatchNorm row limit are ARGUMENTS of the launches (TG_SEED_DEVICE / row_limit), not
    library globals — two GraphedTrainSteps with their own StepState, stepped alternately (replays of one between the
    replays of the other, different true batch sizes in flight), end exactly where each ends when it runs alone."""
    import tabgnn_amd as T
    from tabgnn_amd import synthetic as S, graph_step as G
    B, steps = 48, 4
    batches = [_resize(S.make_batch(B, seed=90 + i, device=DEV), 7 * i, 33 * i) for i in range(2)]
    key = (G.bucket_size(max(b[1].shape[1] for b in batches)), G.bucket_size(max(b[0].num_rows for b in batches) + 1))
    preps = [G.prepare(b, B, key=key) for b in batches]
    frames = (batches[0][0], batches[0][2])

    def make(seed):
        model, flat, opt, lw = _model(B, torch.bfloat16, seed=seed)
        return G.GraphedTrainStep(model, flat, opt, lw, B, state=G.StepState(DEV, seed=1000 + seed)), flat

    alone = {}
    for seed in (21, 22):
        step, flat = make(seed)
        for i in range(steps):
            step(preps[(i + seed) % 2], frames)
        torch.cuda.synchronize()
        alone[seed] = (flat.flat.clone().cpu(), step.state.buf.clone().cpu())
    a, fa = make(21)
    b, fb = make(22)
    for i in range(steps):                         # interleaved; the two see batches of different true size at the same time
        a(preps[(i + 21) % 2], frames)
        b(preps[(i + 22) % 2], frames)
    torch.cuda.synchronize()
    assert torch.equal(fa.flat.cpu(), alone[21][0]) and torch.equal(a.state.buf.cpu(), alone[21][1])
    assert torch.equal(fb.flat.cpu(), alone[22][0]) and torch.equal(b.state.buf.cpu(), alone[22][1])
    assert not torch.equal(alone[21][1][:1], alone[22][1][:1])        # two seed words
    assert T.ops.DropoutRNG.seed < (1 << 63)                          # eager code after a step draws from host seeds again


def test_capture_of_a_new_bucket_while_a_sampler_thread_pins_memory():
    """A sampler thread keeps building pinned arenas (``prepare`` -> hipHostMalloc / event queries in the caching host
    allocator) while the main thread captures a bucket: with capture_error_mode='thread_local' neither side fails."""
    import threading
    import tabgnn_amd as T
    from tabgnn_amd import synthetic as S, graph_step as G
    B = 32
    model, flat, opt, lw = _model(B, torch.bfloat16, seed=4)
    step = G.GraphedTrainStep(model, flat, opt, lw, B)
    host_batches = [S.make_batch(B, seed=70 + i) for i in range(3)]
    errors, stop = [], threading.Event()

    def worker():
        k = 0
        try:
            while not stop.is_set():
                b = host_batches[k % 3]
                E, N = b[1].shape[1], b[0].num_rows
                G.prepare(b, B, key=(G.bucket_size(E) + 64 * (k % 17), G.bucket_size(N + 1) + 32 * (k % 13)))   # ever new arena sizes
                k += 1
        except Exception as e:                      # noqa: BLE001 - the test reports it
            errors.append(e)

    th = threading.Thread(target=worker)
    th.start()
    try:
        for i in range(3):                          # three buckets captured while the worker allocates
            batch = tuple(t.to(DEV) if hasattr(t, "to") else t for t in host_batches[i])
            E, N = batch[1].shape[1], batch[0].num_rows
            prep = G.prepare(batch, B, key=(G.bucket_size(E) + 128 * i, G.bucket_size(N + 1) + 64 * i))
            loss, _ = step(prep, (batch[0], batch[2]))
            assert torch.isfinite(loss).item()
    finally:
        stop.set()
        th.join()
    assert not errors, errors
    assert len(step.buckets) == 3


def test_graphed_step_refuses_what_its_graphs_cannot_follow():
    """Frozen or assumed values the caller changed: seed-row count of the batch, lr after capture, an eager optimiser step
    between replays."""
    import tabgnn_amd as T
    from tabgnn_amd import synthetic as S, graph_step as G
    B = 32
    model, flat, opt, lw = _model(B, torch.bfloat16, seed=6)
    step = G.GraphedTrainStep(model, flat, opt, lw, B)
    batch = S.make_batch(B, seed=5, device=DEV)
    prep = G.prepare(batch, B)
    frames = (batch[0], batch[2])
    step(prep, frames)
    short = S.make_batch(B - 8, seed=6, device=DEV)
    with pytest.raises(ValueError, match="seed rows"):
        step(G.prepare(short, B - 8), frames)
    opt.t += 1                                      # as an eager opt.step() would
    with pytest.raises(RuntimeError, match="device step record"):
        step(prep, frames)
    opt.t -= 1
    opt.lr *= 0.5
    with pytest.raises(RuntimeError, match="after capture"):
        step(prep, frames)
