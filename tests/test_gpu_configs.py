"""GPU parity on the shapes of the other BASELINE configs (parity-test cases, not bench lines):
config 4 — ogbn-arxiv-shaped `tabgnn` path (129 numerical node columns -> S=130 attention, node classification);
config 5 — 64 mixed stype columns (32 categorical with cardinalities up to 10^4, 24 numerical, 8 timestamp), C=256."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _oracle_feats(tf):
    return {k.value: v for k, v in tf.feat_dict.items()}


def test_config4_tabgnn_path_s130_node_classification():
    import tabgnn_amd as T
    from oracle.step import tabgnn_wrapper_forward
    from detparams import rand_subgraph
    st = T.stype
    torch.manual_seed(4)
    V, E, C, ncol = 90, 260, 32, 129
    names_n = {st.numerical: [f"f_{i}" for i in range(128)] + ["year"]}
    stats_n = {n: dict(mean=-0.1, std=0.11) for n in names_n[st.numerical]}
    names_e = {st.relation: ["edge_attr"]}
    node_tf = T.TensorFrame({st.numerical: torch.randn(V, ncol) * 0.11 - 0.1}, names_n)
    edge_tf = T.TensorFrame({st.relation: torch.ones(E, 1)}, names_e)
    ei = torch.from_numpy(rand_subgraph(V, E, 8, 44))
    cfg = dict(model="tabgnn", task="node_classification", batch_size=16, n_hidden=C, n_gnn_layers=2, n_classes=40,
               dropout=0.0, backbone_dropout=0.0, nhead=8, num_node_features=ncol, num_edge_features=1,
               in_degrees=torch.bincount(ei[1], minlength=V), reverse_mp=False,
               node_encoder=T.StypeWiseFeatureEncoder(C, stats_n, names_n),
               edge_encoder=T.StypeWiseFeatureEncoder(C, {}, names_e))
    model = T.TABGNNS(cfg).eval()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    with torch.no_grad():
        want = tabgnn_wrapper_forward(sd, 8, 16, _oracle_feats(node_tf), ei, _oracle_feats(edge_tf),
                                      "node_classification")
        model.to(DEV)
        got = model(node_tf.to(DEV), ei.to(DEV), edge_tf.to(DEV))
    assert got.shape == (V, 40)
    np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), rtol=1e-4, atol=1e-4)
    # backward runs through S=130 attention
    model.train()
    out = model(node_tf.to(DEV), ei.to(DEV), edge_tf.to(DEV))
    T.ops.weighted_cross_entropy(out, (torch.arange(V) % 40).to(DEV)).backward()
    assert all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)


def test_config5_wide_mixed_table_c256():
    import tabgnn_amd as T
    from oracle.step import wrapper_forward, weighted_ce, trainable_keys
    from detparams import rand_subgraph
    st = T.stype
    rs = np.random.RandomState(5)
    C, B, N, E = 256, 12, 70, 150
    cards = [int(c) for c in np.exp(rs.uniform(np.log(2), np.log(1e4), 32))]
    names = {st.numerical: [f"n{i}" for i in range(24)], st.categorical: [f"c{i}" for i in range(32)],
             st.timestamp: [f"t{i}" for i in range(8)]}
    stats = {**{f"n{i}": dict(mean=0.0, std=1.0) for i in range(24)},
             **{f"c{i}": dict(cardinality=cards[i]) for i in range(32)},
             **{f"t{i}": dict(min_year=2015) for i in range(8)}}
    g = torch.Generator().manual_seed(55)
    cat = torch.stack([torch.randint(-1, c, (E,), generator=g) for c in cards], dim=1)
    ts = torch.stack([torch.randint(2015, 2024, (E, 8), generator=g), torch.randint(0, 12, (E, 8), generator=g),
                      torch.randint(0, 31, (E, 8), generator=g), torch.randint(0, 7, (E, 8), generator=g),
                      torch.randint(0, 24, (E, 8), generator=g), torch.randint(0, 60, (E, 8), generator=g),
                      torch.randint(0, 60, (E, 8), generator=g)], dim=2)
    edge_tf = T.TensorFrame({st.numerical: torch.randn(E, 24, generator=g), st.categorical: cat, st.timestamp: ts}, names)
    node_names = {st.relation: ["node_attr"]}
    node_tf = T.TensorFrame({st.relation: torch.ones(N, 1)}, node_names)
    ei = torch.from_numpy(rand_subgraph(N, E, B, 56))
    y = (torch.arange(B) % 2).long()
    torch.manual_seed(6)
    cfg = dict(model="tabgnnfused", task="edge_classification", batch_size=B, n_hidden=C, n_gnn_layers=1, n_classes=2,
               dropout=0.0, backbone_dropout=0.0, nhead=8, num_node_features=1, num_edge_features=64,
               in_degrees=torch.bincount(ei[1], minlength=N), reverse_mp=False, load_model=None, checkpoint=False,
               node_encoder=T.StypeWiseFeatureEncoder(C, {}, node_names),
               edge_encoder=T.StypeWiseFeatureEncoder(C, stats, names))
    model = T.TABGNNFusedS(cfg).train()
    with torch.no_grad():
        for e in model.edge_encoder.encoder_dict["categorical"].embs:
            e.weight.mul_(30.0)                 # make the embedding rows matter against the other 32 columns
            e.weight[0].zero_()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    keys = trainable_keys(sd)
    for k in keys:
        sd[k].requires_grad_(True)
    lw = torch.tensor([1.0, 9.23])
    want = wrapper_forward(sd, 8, B, _oracle_feats(node_tf), ei, _oracle_feats(edge_tf), training=True)
    loss = weighted_ce(want, y, lw)
    loss.backward()
    model.to(DEV)
    got = model(node_tf.to(DEV), ei.to(DEV), edge_tf.to(DEV))
    dl = T.ops.weighted_cross_entropy(got, y.to(DEV), lw.to(DEV))
    dl.backward()
    np.testing.assert_allclose(got.detach().cpu().numpy(), want.detach().numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(dl.item(), loss.item(), rtol=1e-5)
    for k, p in model.named_parameters():
        ref = sd[k].grad if sd[k].grad is not None else torch.zeros_like(sd[k])
        got_g = p.grad.cpu() if p.grad is not None else torch.zeros_like(ref)   # unused edge-update of the last layer
        err = (got_g - ref).abs().max().item()
        assert err <= 2e-3 * (ref.abs().max().item() + 1e-6) + 1e-7, (k, err, ref.abs().max().item())


def test_column_store_batches_by_id_equal_gathered_batches_bit_for_bit():
    """the HBM-resident raw table + a list of sampled edge / node ids (``TensorFrame.row_ids``) is the
    batch; the stype encoders read the raw columns BY ID (tg_enc_ptrs.row_ids).  Against the host-assembled batch
    (``tensor_frame[idx]``, ibm_transactions_for_aml.py:163,168 -> index_select): encoder outputs, seed/neighbour
    slices, logits and every parameter gradient are bit-for-bit equal (same kernels, same reduction order), and the
    index columns seen by the kernel are the gathered ones exactly."""
    import tabgnn_amd as T
    from tabgnn_amd import synthetic as S
    from tabgnn_amd.frame import stype
    from tabgnn_amd.sampler import ColumnStore, NeighborSampler
    rs = np.random.RandomState(3)
    N, E, B = 4000, 30000, 64
    ei = np.stack([rs.randint(0, N, E), rs.randint(0, N, E)])
    num, cat, ts = S.edge_table(E, 7)
    cat[rs.rand(E) < 0.01, 1] = -1                                     # missing categories (padding row)
    labels = torch.from_numpy((rs.rand(E) < 0.05).astype(np.int64))
    store = ColumnStore({stype.numerical: torch.from_numpy(num), stype.categorical: torch.from_numpy(cat),
                         stype.timestamp: torch.from_numpy(ts)}, S.EDGE_COLS,
                        {stype.relation: torch.ones(N, 1)}, S.NODE_COLS, labels).to(DEV)
    sampler = NeighborSampler(ei, N, (10, 10), num_threads=1)
    seeds = rs.choice(E, B, replace=False)
    eid, lei, nodes = sampler.sample(seeds, 5)
    lazy = store.batch(eid, lei, nodes, B, lazy=True, index=True)          # ids + host-built CSRs (BatchIndex, opt-in)
    full = store.batch(eid, lei, nodes, B, lazy=False)         # gathered rows + plain edge_index
    assert lazy[2].row_ids is not None and full[2].row_ids is None and lazy[2].num_rows == full[2].num_rows == eid.numel()
    mat = lazy[2].materialize()
    for k in full[2].feat_dict:                                        # index / raw columns: identical rows
        assert torch.equal(mat.feat_dict[k], full[2].feat_dict[k])
    torch.manual_seed(0)
    cfg = S.make_config(32, 1, 8, B, backbone_dropout=0.0, head_dropout=0.0)
    model = T.TABGNNFusedS(cfg).to(DEV).train()
    with torch.no_grad():
        a, _ = model.edge_encoder(lazy[2][B:, :])
        b, _ = model.edge_encoder(full[2][B:, :])
    assert torch.equal(a, b) and a.shape[0] == eid.numel() - B
    lw = torch.tensor(cfg["loss_weights"], device=DEV)
    grads = []
    for batch in (lazy, full):
        for p in model.parameters():
            p.grad = None
        out = model(batch[0], batch[1], batch[2])
        T.ops.weighted_cross_entropy(out[:B], batch[3], lw).backward()
        grads.append((out.detach().clone(), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}))
    assert grads[0][1].keys() == grads[1][1].keys() and len(grads[0][1]) > 20
    assert torch.equal(grads[0][0], grads[1][0])
    for k in grads[0][1]:
        assert torch.equal(grads[0][1][k], grads[1][1][k]), k
    # the sampler-side index structures (sampler.batch_index: host counting sorts, one upload) are the ones the device
    # builds from edge_index (tg_csr_build): same rowptr, same stable permutation
    bi = lazy[1]
    assert isinstance(bi, T.ops.BatchIndex)
    dev_graph = T.ops.SubgraphIndex.build(bi.edge_index[:, B:].contiguous(), nodes.numel())
    dev_seeds = T.ops.SeedIndex(bi.edge_index[:, :B].contiguous(), nodes.numel())
    En = eid.numel() - B
    for a_, b_ in ((bi.graph.by_dst, dev_graph.by_dst), (bi.graph.by_src, dev_graph.by_src)):
        assert torch.equal(a_[0], b_[0]) and torch.equal(a_[1][:En], b_[1][:En])
    assert torch.equal(bi.graph.src, dev_graph.src) and torch.equal(bi.graph.dst, dev_graph.dst)
    assert torch.equal(bi.seeds.rowptr, dev_seeds.rowptr) and torch.equal(bi.seeds.perm[:2 * B], dev_seeds.perm[:2 * B])
    assert torch.equal(bi.seeds.tei, dev_seeds.tei)


def test_gpu_store_graph_inputs_feed_every_wrapper_as_plain_edge_index():
    """``ColumnStore.graph_inputs`` on a GPU store is the drop-in for ``get_graph_inputs`` (main.py:48): its ``edge_index``
    is a plain int64 tensor, so the non-fused wrappers (``TABGNNS``, ``GNN``: utils.py:111-328) take the batch as it
    comes; the prebuilt ``ops.BatchIndex`` is opt-in (``index=True``) and only ``TABGNNFusedS`` unwraps it."""
    import tabgnn_amd as T
    from tabgnn_amd import synthetic as S
    from tabgnn_amd.frame import stype
    from tabgnn_amd.sampler import ColumnStore, NeighborSampler
    rs = np.random.RandomState(5)
    N, E, B = 3000, 20000, 32
    ei = np.stack([rs.randint(0, N, E), rs.randint(0, N, E)])
    num, cat, ts = S.edge_table(E, 11)
    labels = torch.from_numpy((rs.rand(E) < 0.05).astype(np.int64))
    store = ColumnStore({stype.numerical: torch.from_numpy(num), stype.categorical: torch.from_numpy(cat),
                         stype.timestamp: torch.from_numpy(ts)}, S.EDGE_COLS,
                        {stype.relation: torch.ones(N, 1)}, S.NODE_COLS, labels).to(DEV)
    sampler = NeighborSampler(ei, N, (8, 8), num_threads=1)
    seeds = rs.choice(E, B, replace=False)
    node_tf, edge_index, edge_tf, y = store.graph_inputs(sampler, seeds, rng_seed=2)
    assert isinstance(edge_index, torch.Tensor) and edge_index.dtype == torch.int64 and edge_index.is_cuda
    _, bi, _, _ = store.graph_inputs(sampler, seeds, rng_seed=2, index=True)
    assert isinstance(bi, T.ops.BatchIndex) and torch.equal(bi.edge_index, edge_index)
    torch.manual_seed(0)
    cfg = S.make_config(32, 2, 4, B, backbone_dropout=0.0, head_dropout=0.0)
    outs = {}
    for name, cls in (("tabgnnfused", T.TABGNNFusedS), ("tabgnn", T.TABGNNS), ("pna", T.GNN)):
        c = dict(cfg)
        c["model"] = name
        model = cls(c).to(DEV).train()
        out = model(node_tf, edge_index, edge_tf)
        assert out.shape[0] >= B and torch.isfinite(out.float()).all(), name
        outs[name] = out
    fused = T.TABGNNFusedS(dict(cfg)).to(DEV).eval()
    with torch.no_grad():
        assert torch.equal(fused(node_tf, edge_index, edge_tf), fused(node_tf, bi, edge_tf))
