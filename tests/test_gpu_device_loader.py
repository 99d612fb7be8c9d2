"""GPU: the device sampler as the DATA PATH of the training loop (round 5; SURVEY §8f rank 1, reference loop
``src/datasets/ibm_transactions_for_aml.py:61-112,159-180`` -> ``main.py:41-75``):
  * ``device_batch_index`` — the batch's 13 index parts built by the device kernels — equals ``host_batch_index`` bit for bit;
  * ``prepare_sample_device`` builds the same bucket arena as the host's ``prepare_sample``, part by part;
  * ``DeviceBatchLoader`` (draw / emit / index one step ahead on a side stream) hands over exactly the batches the plain
    sampler calls give, and a training loop fed by it follows the same loss trajectory bit for bit;
  * HIP-graph replays of device-prepared batches equal the eager runs of the same body bit for bit."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _graph(seed=3, N=4000, E=30000):
    rs = np.random.RandomState(seed)
    ei = np.stack([rs.randint(0, N, E), rs.randint(0, N, E)])
    return rs, ei, N, E


def _store(rs, N, E):
    from tabgnn_amd import synthetic as S
    from tabgnn_amd.frame import stype
    from tabgnn_amd.sampler import ColumnStore
    num, cat, ts = S.edge_table(E, 0)
    labels = torch.from_numpy((rs.rand(E) < 0.05).astype(np.int64))
    return ColumnStore({stype.numerical: torch.from_numpy(num), stype.categorical: torch.from_numpy(cat),
                        stype.timestamp: torch.from_numpy(ts)}, S.EDGE_COLS,
                       {stype.relation: torch.ones(N, 1)}, S.NODE_COLS, labels).to(DEV), labels


def _model(B, seed):
    import tabgnn_amd as T
    from tabgnn_amd import synthetic as S
    torch.manual_seed(seed)
    cfg = S.make_config(128, 2, 4, B, compute_dtype=torch.bfloat16)
    model = T.TABGNNFusedS(cfg).to(DEV).train()
    flat = T.FlatParams(model, shadow_dtype=torch.bfloat16)
    return model, flat, T.FusedAdam(flat, lr=1e-3), torch.tensor([1.0, 9.23], device=DEV)


@pytest.mark.parametrize("B,fan", [(32, (6, 4)), (200, (20, 10)), (1, (3,))])
def test_device_index_parts_equal_the_host_builder(B, fan):
    from tabgnn_amd import DeviceNeighborSampler, device_batch_index
    from tabgnn_amd.sampler import host_batch_index
    rs, ei, N, E = _graph()
    smp = DeviceNeighborSampler(ei, N, fan, DEV)
    eid, lei, nodes = smp.sample(rs.choice(E, B, replace=False), 5)
    n = int(nodes.shape[0])
    flat_h, off, _ = host_batch_index(lei.cpu().numpy(), n, B)
    idx = device_batch_index(lei, n, B)
    got = torch.cat([idx.graph.src, idx.graph.dst, idx.graph.by_dst[0], idx.graph.by_dst[1], idx.graph.by_src[0], idx.graph.by_src[1],
                     idx.seeds.tei, idx.seeds.rowptr, idx.seeds.perm, idx.graph._sorted["dst"], idx.graph._sorted["src"],
                     idx.graph._sorted["inv"], idx.graph._sorted["src_to_sorted"]]).cpu().numpy()
    En = int(lei.shape[1]) - B
    want = flat_h
    assert got.shape == want.shape
    names = "src dst rp_d pm_d rp_s pm_s tei rp_t pm_t dst_sorted src_sorted inv s2s".split()
    for i, name in enumerate(names):
        a, b = got[int(off[i]):int(off[i + 1])], want[int(off[i]):int(off[i + 1])]
        if name in ("pm_d", "pm_s") and En == 0:       # (the one-entry placeholders of empty permutations are not defined)
            continue
        assert np.array_equal(a, b), name
    assert int(idx.graph.err.item()) == 0


def test_device_prepared_bucket_equals_the_host_prepared_bucket():
    from tabgnn_amd import DeviceNeighborSampler, graph_step as G, prepare_sample_device
    rs, ei, N, E = _graph()
    store, labels = _store(rs, N, E)
    smp = DeviceNeighborSampler(ei, N, (6, 4), DEV)
    B = 32
    for i in range(3):
        eid, lei, nodes = smp.sample(rs.choice(E, B, replace=False), i)
        y = labels[eid[:B].cpu()]
        host = G.prepare_sample(eid.cpu(), lei.cpu(), nodes.cpu(), y, B)
        dev = prepare_sample_device(eid, lei, nodes, y.to(DEV), B)
        assert dev.key == host.key and dev.layout == host.layout and dev.arena.is_cuda
        assert (dev.e_real, dev.n_real) == (host.e_real, host.n_real)
        for name in ("flat", "ei", "n_real", "y", "node.ids", "edge.ids"):
            assert torch.equal(dev.tensors[name].cpu(), host.tensors[name]), name


def test_loader_hands_over_the_plain_sampler_batches_and_trains_the_same():
    import tabgnn_amd as T
    from tabgnn_amd import DeviceBatchLoader, DeviceNeighborSampler
    rs, ei, N, E = _graph(seed=5)
    store, labels = _store(rs, N, E)
    B, steps = 64, 6
    seeds = [rs.choice(E, B, replace=False) for _ in range(steps)]
    smp = DeviceNeighborSampler(ei, N, (8, 4), DEV)
    # plain: sample -> batch; the model builds the index structures inside the step
    model, flat, opt, lw = _model(B, seed=4)
    plain, shapes = [], []
    for i, s in enumerate(seeds):
        eid, lei, nodes = smp.sample(s, 100 + i)
        shapes.append((int(eid.shape[0]), int(nodes.shape[0])))
        plain.append(T.train_step(model, flat, opt, store.batch(eid, lei, nodes, B), lw, step_seed=i)[0].clone())
    w_plain = flat.flat.clone()
    # loader: the same seeds and rng seeds, batches prepared one step ahead on the side stream
    model, flat, opt, lw = _model(B, seed=4)
    smp2 = DeviceNeighborSampler(ei, N, (8, 4), DEV)
    loader = DeviceBatchLoader(smp2, store, seeds, mode="index", rng_seed=100)
    got, shapes2 = [], []
    for i, batch in enumerate(loader):
        shapes2.append((int(batch[1].edge_index.shape[1]), int(batch[0].num_rows)))
        got.append(T.train_step(model, flat, opt, batch, lw, step_seed=i)[0].clone())
    torch.cuda.synchronize()
    assert shapes2 == shapes and len(got) == steps
    assert torch.equal(torch.stack(got), torch.stack(plain)) and torch.equal(flat.flat, w_plain)


def test_device_prepared_batches_replay_equals_eager():
    import tabgnn_amd as T
    from tabgnn_amd import DeviceBatchLoader, DeviceNeighborSampler, graph_step as G
    rs, ei, N, E = _graph(seed=7)
    store, labels = _store(rs, N, E)
    B, steps = 32, 12
    seeds = [rs.choice(E, B, replace=False) for _ in range(steps)]
    frames = (T.TensorFrame(store.node_feats, store.node_cols, None, torch.zeros(1, dtype=torch.int64, device=DEV)),
              T.TensorFrame(store.edge_feats, store.edge_cols, None, torch.zeros(1, dtype=torch.int64, device=DEV)))
    runs = {}
    for mode in ("eager", "graph"):
        smp = DeviceNeighborSampler(ei, N, (10, 6), DEV)
        loader = DeviceBatchLoader(smp, store, seeds, mode="bucket", rng_seed=11)
        model, flat, opt, lw = _model(B, seed=9)
        step = G.GraphedTrainStep(model, flat, opt, lw, B)
        losses, keys = [], set()
        for prep in loader:
            assert prep.arena.is_cuda
            keys.add(prep.key)
            losses.append((step.run_eager(prep, frames) if mode == "eager" else step(prep, frames))[0].clone())
        torch.cuda.synchronize()
        runs[mode] = (torch.stack(losses).float().cpu(), flat.flat.clone().cpu(), len(keys))
    assert len(runs["eager"][0]) == steps and runs["eager"][2] >= 1 and torch.isfinite(runs["eager"][0]).all()
    print("buckets:", runs["eager"][2])
    assert torch.equal(runs["eager"][0], runs["graph"][0]) and torch.equal(runs["eager"][1], runs["graph"][1])


def test_configs4_graph_at_its_size_samples_and_trains():
    """BASELINE configs[4] on its real graph: 10 M nodes, 100 M edges, the 64-column raw table (80 GB)
    resident in HBM.  Properties of the sampler's output at that size — seed edges first and in order, edge_index = ranks
    of the endpoints in the sorted node list (relabel), every sampled edge a real in-edge of a frontier node — and one
    finite train step of the d = 256 fused model fed by ids."""
    import tabgnn_amd as T
    from tabgnn_amd import DeviceBatchLoader, DeviceNeighborSampler, synthetic as S
    free, _ = torch.cuda.mem_get_info()
    if free < 120e9:
        pytest.skip("needs ~100 GB of free HBM")
    N, E, B = 10_000_000, 100_000_000, 256
    ei = S.powerlaw_graph_on_device(N, E, DEV)
    store = S.wide64_store_on_device(N, E, DEV)
    smp = DeviceNeighborSampler(ei, N, (10, 5), DEV)
    g = torch.Generator(); g.manual_seed(1)
    seeds = torch.randint(0, E, (B,), generator=g)
    eid, lei, nodes = smp.sample(seeds, 3)
    assert torch.equal(eid[:B].cpu(), seeds)                                           # seeds first, in order
    assert eid.unique().numel() == eid.numel()                                          # no edge twice
    assert torch.equal(nodes, torch.sort(nodes).values) and nodes.unique().numel() == nodes.numel()
    assert torch.equal(nodes[lei[0]], ei[0, eid]) and torch.equal(nodes[lei[1]], ei[1, eid])   # relabel
    assert int(eid.max()) < E and int(nodes.max()) < N and eid.numel() > 4 * B
    # rows of the last table row range are reachable (64-bit row arithmetic of the stype encoders)
    assert int(eid.max()) > 2 ** 26
    deg = torch.bincount(ei[1], minlength=N).cpu()
    torch.manual_seed(9)
    model = T.TABGNNFusedS(S.wide64_config(B, deg, torch.bfloat16)).to(DEV).train()
    flat = T.FlatParams(model, shadow_dtype=torch.bfloat16)
    opt = T.FusedAdam(flat, lr=6e-4)
    lw = torch.tensor([1.0, 9.23], device=DEV)
    loader = DeviceBatchLoader(smp, store, [torch.randint(0, E, (B,), generator=g) for _ in range(3)], mode="index", rng_seed=5)
    losses = [float(T.train_step(model, flat, opt, b, lw)[0]) for b in loader]
    assert len(losses) == 3 and all(np.isfinite(v) for v in losses)
    assert all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)
