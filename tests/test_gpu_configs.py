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
