"""CPU: the oracle restatement against golden vectors produced by the reference's own files."""
import numpy as np
import pytest
import torch

from golden_util import FUSED_CASES, build_state, fused_inputs, fused_train_loss, load_case
from detparams import det_tensor
from oracle.fused_path import fused_forward
from oracle.heads import classifier_head, node_classification_head
from oracle.tabgnn_path import tabgnn_forward

TOL = dict(rtol=2e-5, atol=2e-5)


def _run_fused(cfg, z, sd, hsd, training):
    x, ei, ea = fused_inputs(cfg, z)
    B = cfg["B"]
    xg, e, t = fused_forward(sd, cfg["H"], x, ei[:, B:], ea[B:], ei[:, :B], ea[:B], lp=cfg["lp"],
                             p_drop=0.0, training=training)
    lg = classifier_head(xg, ei[:, :B], t, hsd)
    return xg, e, t, lg


@pytest.mark.parametrize("name", FUSED_CASES)
def test_fused_eval_matches_reference(name):
    cfg, z = load_case(name)
    sd = build_state(cfg["keys"], z, cfg["seed"])
    hsd = build_state(cfg["head_keys"], z, cfg["seed"] + 1)
    with torch.no_grad():
        xg, e, t, lg = _run_fused(cfg, z, sd, hsd, training=False)
    rs = cfg["row_stride"]
    np.testing.assert_allclose(lg.numpy(), z["eval.logits"], rtol=1e-4, atol=1e-4)   # north_star: logits within 1e-4
    np.testing.assert_allclose(t.numpy(), z["eval.target"], **TOL)
    np.testing.assert_allclose(xg[::rs].numpy(), z["eval.x_gnn"], **TOL)
    np.testing.assert_allclose(e[::rs].numpy(), z["eval.edge_attr"], **TOL)


@pytest.mark.parametrize("name", [c for c in FUSED_CASES if c != "fused_amlbatch_c32_h8_l1"] + ["fused_amlbatch_c32_h8_l1"])
def test_fused_train_grads_match_reference(name):
    cfg, z = load_case(name)
    sd = build_state(cfg["keys"], z, cfg["seed"])
    hsd = build_state(cfg["head_keys"], z, cfg["seed"] + 1)
    params = [k for k, v in sd.items() if v.is_floating_point() and "running" not in k and "avg_deg" not in k]
    for k in params:
        sd[k].requires_grad_(True)
    xg, e, t, lg = _run_fused(cfg, z, sd, hsd, training=True)
    y = torch.from_numpy(z["y"])
    loss = fused_train_loss(cfg, xg, e, lg, y)
    loss.backward()
    np.testing.assert_allclose(loss.item(), float(z["train.loss"]), rtol=1e-5)
    np.testing.assert_allclose(lg.detach().numpy(), z["train.logits"], rtol=1e-4, atol=1e-4)
    for k in params:
        g = sd[k].grad if sd[k].grad is not None else torch.zeros_like(sd[k])
        ref = float(z["gradnorm." + k])
        assert abs(g.double().norm().item() - ref) <= 1e-4 * max(ref, 1e-3), k
        if ("grad." + k) in z.files:
            np.testing.assert_allclose(g.numpy(), z["grad." + k], rtol=1e-3, atol=1e-5, err_msg=k)
    for k in sd:
        if "running" in k:
            np.testing.assert_allclose(sd[k].numpy(), z["bn_after." + k], rtol=1e-5, atol=1e-6, err_msg=k)


def test_tabgnn_matches_reference():
    cfg, z = load_case("tabgnn_c32_h8_l2")
    seed = cfg["seed"]
    sd = build_state(cfg["keys"], z, seed)
    hsd = build_state(cfg["head_keys"], z, seed + 1)
    x = det_tensor("in.x", (cfg["N"], cfg["n_node_cols"], cfg["C"]), seed)
    ea = det_tensor("in.edge_attr", (cfg["E"], cfg["n_edge_cols"], cfg["C"]), seed)
    ei = torch.from_numpy(z["edge_index"])
    with torch.no_grad():
        xv, e = tabgnn_forward(sd, cfg["H"], x, ei, ea)
        lg = node_classification_head(xv, hsd)
    np.testing.assert_allclose(xv.numpy(), z["eval.x"], **TOL)
    np.testing.assert_allclose(e.numpy(), z["eval.edge_attr"], **TOL)
    np.testing.assert_allclose(lg.numpy(), z["eval.logits"], rtol=1e-4, atol=1e-4)
    params = [k for k, v in sd.items() if v.is_floating_point() and "running" not in k and "avg_deg" not in k]
    for k in params:
        sd[k].requires_grad_(True)
    xv, e = tabgnn_forward(sd, cfg["H"], x, ei, ea, training=True)
    lg = node_classification_head(xv, hsd)
    loss = torch.nn.functional.cross_entropy(lg, torch.from_numpy(z["y"])) \
        + 0.01 * (e * det_tensor("co.e", e.shape, seed)).sum() / cfg["E"]
    loss.backward()
    np.testing.assert_allclose(loss.item(), float(z["train.loss"]), rtol=1e-5)
    for k in params:
        g = sd[k].grad if sd[k].grad is not None else torch.zeros_like(sd[k])
        ref = float(z["gradnorm." + k])
        assert abs(g.double().norm().item() - ref) <= 1e-4 * max(ref, 1e-3), k


def test_restated_pnaconv_inventory_matches_the_notebook_parameter_counts():
    """A pin the reference itself printed (benchmark.ipynb:230-231): ``model_params 7451904`` with the encoder held
    inside the model and ``encoder_params 5936384`` -> TABGNN(C=128, L=3, 6 edge columns) without its encoder has
    7 451 904 - 5 936 384 = 1 515 520 parameters.  886 272 of them are the three PyG ``PNAConv`` modules, i.e. this
    pins the restated PNAConv parameter inventory (edge_encoder F*F+F, pre_nn 3F*F+F, post_nn 13F*F+F, lin F*F+F)."""
    import os
    import tabgnn_amd as T
    from oracle import pyg_restate as R
    PIN = 7_451_904 - 5_936_384
    assert PIN == 1_515_520
    deg = torch.tensor([0, 5, 3, 1])
    kw = dict(aggregators=["mean", "max", "min", "std"], scalers=["identity", "amplification", "attenuation"], deg=deg,
              edge_dim=128, towers=1, pre_layers=1, post_layers=1, divide_input=False)
    conv = R.PNAConv(128, 128, **kw)
    n_conv = sum(p.numel() for p in conv.parameters())
    assert n_conv == 295_424
    model = T.TABGNN(channels=128, num_layers=3, edge_dim=6 * 128, node_dim=1, deg=deg)     # the product's mirror
    assert sum(p.numel() for p in model.parameters()) == PIN
    mine = {k: tuple(v.shape) for k, v in model.gnn_backbone[0].gnn_conv.state_dict().items()}
    assert mine == {k: tuple(v.shape) for k, v in conv.state_dict().items()}
    # the encoder count admits the restated inventory: 5 categorical tables of (cardinality + 1) x 128 rows plus one
    # numerical column's weight and bias ([1,128] each) -- a consistency check only, the cardinalities are not recorded
    assert (5_936_384 - 2 * 128) % 128 == 0
    if os.path.isdir("/root/reference"):       # this container only: the reference's own TABGNN around the restated conv
        from oracle.ref_shim import load_reference
        ref = load_reference()["TABGNN"](channels=128, num_layers=3, edge_dim=6 * 128, node_dim=1, deg=deg)
        assert sum(p.numel() for p in ref.parameters()) == PIN
        assert sum(p.numel() for n, p in ref.named_parameters() if "gnn_conv" in n) == 3 * n_conv


def test_config1_tiny_csv_batch_real_column_values():
    """BASELINE configs[0] (SURVEY 8c golden item ii): a sampled subgraph of ``data/Over-Sampled_Tiny_Trans-c.csv`` with
    its real currency / format / amount / timestamp values -> encoders -> TABGNNFused(d=32, H=8, L=1) -> ClassifierHead.
    Expected values come from the reference's own fused.py / decoder.py (tests/golden/make_golden.py:tiny_csv_case);
    here the oracle's whole wrapper restatement (oracle/step.py) is checked against them."""
    from golden_util import tinycsv_state
    from oracle.step import trainable_keys, weighted_ce, wrapper_forward
    cfg, z = load_case("tinycsv_c32_h8_l1")
    sd, nf, ef = tinycsv_state(cfg, z)
    ei = torch.from_numpy(z["edge_index"])
    B = cfg["B"]
    assert ef["categorical"].max() < 15 and ef["timestamp"][:, 0, 0].min() == 2022 and cfg["E"] == ei.shape[1]
    with torch.no_grad():
        lg = wrapper_forward(sd, cfg["H"], B, nf, ei, ef, training=False)
    np.testing.assert_allclose(lg.numpy(), z["eval.logits"], rtol=1e-4, atol=1e-4)
    keys = trainable_keys(sd)
    for k in keys:
        sd[k].requires_grad_(True)
    lg = wrapper_forward(sd, cfg["H"], B, nf, ei, ef, training=True)
    loss = weighted_ce(lg, torch.from_numpy(z["y"]), torch.tensor([1.0, 9.23]))
    loss.backward()
    np.testing.assert_allclose(loss.item(), float(z["train.loss"]), rtol=1e-5)
    np.testing.assert_allclose(lg.detach().numpy(), z["train.logits"], rtol=1e-4, atol=1e-4)
    checked = 0
    for k in keys:
        if ("gradnorm." + k) not in z.files:
            continue
        g = sd[k].grad if sd[k].grad is not None else torch.zeros_like(sd[k])
        ref = float(z["gradnorm." + k])
        assert abs(g.double().norm().item() - ref) <= 1e-4 * max(ref, 1e-3), k
        checked += 1
        if ("grad." + k) in z.files:
            np.testing.assert_allclose(g.numpy(), z["grad." + k], rtol=1e-3, atol=1e-6, err_msg=k)
    assert checked >= 60
    for k in sd:
        if "running" in k and k.startswith("model."):
            np.testing.assert_allclose(sd[k].numpy(), z["bn_after." + k], rtol=1e-5, atol=1e-6, err_msg=k)
