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
