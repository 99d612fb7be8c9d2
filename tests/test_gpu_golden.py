"""GPU parity of the whole drop-in path (TABGNNFused / TABGNN + heads, through the C ABI) against the golden
vectors produced by the reference's own files, and against the oracle on the same inputs."""
import numpy as np
import pytest
import torch

from golden_util import FUSED_CASES, build_state, fused_inputs, fused_train_loss, load_case
from detparams import det_tensor

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _build_fused(cfg, z):
    import tabgnn_amd as T
    C = cfg["C"]
    ei = torch.from_numpy(z["edge_index"].astype(np.int64))
    deg = torch.bincount(torch.bincount(ei[1], minlength=cfg["N"]))
    model = T.TABGNNFused(channels=C, num_layers=cfg["L"], deg=deg, node_dim=C, nhidden=C, edge_dim=cfg["ncols"] * C,
                          reverse_mp=cfg["reverse_mp"], nhead=cfg["H"], dropout=0.0)
    head = T.ClassifierHead(2, C, dropout=0.0)
    model.load_state_dict(build_state(cfg["keys"], z, cfg["seed"]))        # reference key names, strict
    head.load_state_dict(build_state(cfg["head_keys"], z, cfg["seed"] + 1))
    return model.to(DEV), head.to(DEV)


def _run(model, head, cfg, z):
    x, ei, ea = fused_inputs(cfg, z)
    B = cfg["B"]
    x, ei, ea = x.to(DEV), ei.to(DEV), ea.to(DEV)
    tei = ei[:, :B].contiguous()
    xg, e, t = model(x, ei[:, B:].contiguous(), ea[B:].contiguous(), tei, ea[:B].contiguous(), lp=cfg["lp"])
    return xg, e, t, head(xg, tei, t)


@pytest.mark.parametrize("name", FUSED_CASES)
def test_fused_eval_logits_within_1e4_of_reference(name):
    cfg, z = load_case(name)
    model, head = _build_fused(cfg, z)
    model.eval(); head.eval()
    with torch.no_grad():
        xg, e, t, lg = _run(model, head, cfg, z)
    rs = cfg["row_stride"]
    np.testing.assert_allclose(lg.cpu().numpy(), z["eval.logits"], rtol=1e-4, atol=1e-4)      # north_star tolerance
    np.testing.assert_allclose(t.cpu().numpy(), z["eval.target"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(xg[::rs].cpu().numpy(), z["eval.x_gnn"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(e[::rs].cpu().numpy(), z["eval.edge_attr"], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("name", FUSED_CASES)
def test_fused_train_grads_match_reference(name):
    cfg, z = load_case(name)
    model, head = _build_fused(cfg, z)
    model.train(); head.train()
    xg, e, t, lg = _run(model, head, cfg, z)
    loss = fused_train_loss(cfg, xg, e, lg, torch.from_numpy(z["y"]).to(DEV))
    loss.backward()
    np.testing.assert_allclose(loss.item(), float(z["train.loss"]), rtol=2e-5)
    np.testing.assert_allclose(lg.detach().cpu().numpy(), z["train.logits"], rtol=1e-4, atol=1e-4)
    for k, p in model.named_parameters():
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        ref = float(z["gradnorm." + k])
        assert abs(g.double().norm().item() - ref) <= 2e-3 * max(ref, 1e-3), (k, g.double().norm().item(), ref)
        if ("grad." + k) in z.files:
            np.testing.assert_allclose(g.cpu().numpy(), z["grad." + k], rtol=5e-3, atol=2e-5, err_msg=k)
    for k, v in model.state_dict().items():
        if "running" in k:
            np.testing.assert_allclose(v.cpu().numpy(), z["bn_after." + k], rtol=1e-4, atol=1e-5, err_msg=k)


def test_tabgnn_matches_reference():
    import tabgnn_amd as T
    cfg, z = load_case("tabgnn_c32_h8_l2")
    C, seed = cfg["C"], cfg["seed"]
    ei = torch.from_numpy(z["edge_index"])
    deg = torch.bincount(torch.bincount(ei[1], minlength=cfg["N"]))
    model = T.TABGNN(channels=C, num_layers=cfg["L"], deg=deg, node_dim=cfg["n_node_cols"] * C, nhidden=C,
                     edge_dim=cfg["n_edge_cols"] * C, nhead=cfg["H"], dropout=0.0)
    head = T.NodeClassificationHead(cfg["n_classes"], C, dropout=0.0)
    model.load_state_dict(build_state(cfg["keys"], z, seed)); head.load_state_dict(build_state(cfg["head_keys"], z, seed + 1))
    model.to(DEV).eval(); head.to(DEV).eval()
    x = det_tensor("in.x", (cfg["N"], cfg["n_node_cols"], C), seed).to(DEV)
    ea = det_tensor("in.edge_attr", (cfg["E"], cfg["n_edge_cols"], C), seed).to(DEV)
    with torch.no_grad():
        xv, e = model(x, ei.to(DEV), ea)
        lg = head(xv)
    np.testing.assert_allclose(xv.cpu().numpy(), z["eval.x"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(e.cpu().numpy(), z["eval.edge_attr"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(lg.cpu().numpy(), z["eval.logits"], rtol=1e-4, atol=1e-4)
    model.train(); head.train()
    xv, e = model(x, ei.to(DEV), ea)
    lg = head(xv)
    loss = torch.nn.functional.cross_entropy(lg, torch.from_numpy(z["y"]).to(DEV)) \
        + 0.01 * (e * det_tensor("co.e", e.shape, seed).to(DEV)).sum() / cfg["E"]
    loss.backward()
    np.testing.assert_allclose(loss.item(), float(z["train.loss"]), rtol=2e-5)
    for k, p in model.named_parameters():
        ref = float(z["gradnorm." + k])
        assert abs(p.grad.double().norm().item() - ref) <= 2e-3 * max(ref, 1e-3), k


def test_config1_tiny_csv_batch_real_column_values():
    """BASELINE configs[0]: the sampled subgraph of ``data/Over-Sampled_Tiny_Trans-c.csv`` with its real column values
    (tests/golden/tinycsv_c32_h8_l1.npz) through the product's wrapper — stype encoders, TABGNNFused(d=32, H=8, L=1),
    ClassifierHead — on the GPU, against values the reference's own fused.py / decoder.py produced: logits within
    1e-4 (north_star), loss, every parameter's gradient norm, BatchNorm statistics."""
    import tabgnn_amd as T
    from golden_util import tinycsv_state
    st = T.stype
    cfg, z = load_case("tinycsv_c32_h8_l1")
    sd, nf, ef = tinycsv_state(cfg, z)
    C, B = cfg["C"], cfg["B"]
    cols = {st.numerical: ["Amount Paid"], st.categorical: cfg["cat_cols"], st.timestamp: ["Timestamp"]}
    stats = {"Amount Paid": dict(mean=cfg["mean"], std=cfg["std"]), "Timestamp": dict(min_year=cfg["min_year"]),
             **{n: dict(cardinality=c) for n, c in zip(cfg["cat_cols"], cfg["cards"])}}
    ncols = {st.relation: ["node_attr"]}
    # in_degrees whose histogram is the fixture's (train-graph in-degree histogram, main.py:283-286)
    hist = torch.from_numpy(z["deg_hist"])
    in_deg = torch.repeat_interleave(torch.arange(hist.numel()), hist)
    wcfg = dict(model="tabgnnfused", task="edge_classification", batch_size=B, n_hidden=C, n_gnn_layers=cfg["L"],
                n_classes=2, dropout=0.0, backbone_dropout=0.0, nhead=cfg["H"], num_node_features=1, num_edge_features=5,
                in_degrees=in_deg, reverse_mp=False, load_model=None, checkpoint=False,
                node_encoder=T.StypeWiseFeatureEncoder(C, {}, ncols), edge_encoder=T.StypeWiseFeatureEncoder(C, stats, cols))
    model = T.TABGNNFusedS(wcfg)
    own = model.state_dict()
    missing = [k for k in sd if k not in own]
    assert not missing, missing
    for k in own:                                   # every parameter comes from the fixture; max_values is a constant
        assert k in sd or k.endswith("max_values"), k
    model.load_state_dict(sd, strict=False)
    np.testing.assert_allclose(own["edge_encoder.encoder_dict.numerical.std"].numpy(),
                               sd["edge_encoder.encoder_dict.numerical.std"].numpy(), rtol=1e-6)
    model.to(DEV)
    node_tf = T.TensorFrame({st.relation: nf["relation"]}, ncols).to(DEV)
    edge_tf = T.TensorFrame({st.numerical: ef["numerical"], st.categorical: ef["categorical"],
                             st.timestamp: ef["timestamp"]}, cols).to(DEV)
    ei = torch.from_numpy(z["edge_index"]).to(DEV)
    model.eval()
    with torch.no_grad():
        lg = model(node_tf, ei, edge_tf)
    np.testing.assert_allclose(lg.cpu().numpy(), z["eval.logits"], rtol=1e-4, atol=1e-4)
    model.train()
    lg = model(node_tf, ei, edge_tf)
    loss = T.ops.weighted_cross_entropy(lg, torch.from_numpy(z["y"]).to(DEV), torch.tensor([1.0, 9.23], device=DEV))
    loss.backward()
    np.testing.assert_allclose(loss.item(), float(z["train.loss"]), rtol=2e-5)
    np.testing.assert_allclose(lg.detach().cpu().numpy(), z["train.logits"], rtol=1e-4, atol=1e-4)
    checked = 0
    for k, p in model.named_parameters():
        ref = float(z["gradnorm." + k])
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        assert abs(g.double().norm().item() - ref) <= 2e-3 * max(ref, 1e-3), (k, g.double().norm().item(), ref)
        checked += 1
        if ("grad." + k) in z.files:
            np.testing.assert_allclose(g.cpu().numpy(), z["grad." + k], rtol=5e-3, atol=2e-6, err_msg=k)
    assert checked >= 60
    for k, v in model.state_dict().items():
        if "running" in k:
            np.testing.assert_allclose(v.cpu().numpy(), z["bn_after." + k], rtol=1e-4, atol=1e-5, err_msg=k)
