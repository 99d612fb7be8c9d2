"""Sibling backbones on the same ops: TABGNNInterleaved (inteleaved.py) and PNAS (pna.py:48-97)
against goldens generated from the reference's own files.  CPU: the oracle restatement; GPU: the product modules."""
import numpy as np
import pytest
import torch

from detparams import det_tensor
from golden_util import build_state, load_case


def _scalar(cfg, a, b, N, E):
    seed = cfg["seed"]
    return (a.float() * det_tensor("co.x", a.shape, seed).to(a.device)).sum() / N \
        + (b.float() * det_tensor("co.e", b.shape, seed).to(b.device)).sum() / E


def _inputs(cfg, z, Cx):
    seed, N, E, nc = cfg["seed"], cfg["N"], cfg["E"], cfg["ncols"]
    return (det_tensor("in.x", (N, 1, Cx), seed), torch.from_numpy(z["edge_index"].astype(np.int64)),
            det_tensor("in.edge_attr", (E, nc, Cx), seed))


def test_oracle_interleaved_matches_reference():
    from oracle.siblings import interleaved_forward
    cfg, z = load_case("interleaved_c32_h4_l2")
    x, ei, ea = _inputs(cfg, z, cfg["C"])
    sd = build_state(cfg["keys"], z, cfg["seed"])
    with torch.no_grad():
        xg, xe = interleaved_forward({k: v.clone() for k, v in sd.items()}, cfg["H"], x, ei, ea)
    np.testing.assert_allclose(xg.numpy(), z["eval.x_gnn"], atol=2e-5)
    np.testing.assert_allclose(xe.numpy(), z["eval.x_edge"], atol=2e-5)
    for k in sd:
        if sd[k].is_floating_point() and "avg_deg" not in k and "running" not in k:
            sd[k].requires_grad_(True)
    xg, xe = interleaved_forward(sd, cfg["H"], x, ei, ea, training=True)
    loss = _scalar(cfg, xg, xe, cfg["N"], cfg["E"])
    loss.backward()
    assert abs(loss.item() - float(z["train.loss"])) < 1e-5
    for k in sd:
        if f"gradnorm.{k}" in z.files:
            g = sd[k].grad if sd[k].grad is not None else torch.zeros(())
            assert abs(g.double().norm().item() - float(z[f"gradnorm.{k}"])) < 1e-4 * max(1.0, float(z[f"gradnorm.{k}"])), k
        if f"bn.{k}" in z.files:
            np.testing.assert_allclose(sd[k].detach().numpy(), z[f"bn.{k}"], atol=1e-6)


@pytest.mark.parametrize("name", ["pnas_f32_l2", "pnas_f32_l2_rmp"])
def test_oracle_pnas_matches_reference(name):
    from oracle.siblings import pnas_forward
    cfg, z = load_case(name)
    x, ei, ea = _inputs(cfg, z, cfg["F"])
    sd = build_state(cfg["keys"], z, cfg["seed"])
    with torch.no_grad():
        xo, eo = pnas_forward({k: v.clone() for k, v in sd.items()}, x, ei, ea)
    np.testing.assert_allclose(xo.numpy(), z["eval.x"], atol=2e-5)
    np.testing.assert_allclose(eo.numpy(), z["eval.edge_attr"], atol=2e-5)
    for k in sd:
        if sd[k].is_floating_point() and "avg_deg" not in k and "running" not in k:
            sd[k].requires_grad_(True)
    xo, eo = pnas_forward(sd, x, ei, ea, training=True)
    loss = _scalar(cfg, xo, eo, cfg["N"], cfg["E"])
    loss.backward()
    assert abs(loss.item() - float(z["train.loss"])) < 1e-5
    for k in sd:
        if f"gradnorm.{k}" in z.files and sd[k].grad is not None:
            assert abs(sd[k].grad.double().norm().item() - float(z[f"gradnorm.{k}"])) < 1e-4 * max(1.0, float(z[f"gradnorm.{k}"])), k


@pytest.mark.gpu
def test_gpu_interleaved_matches_reference():
    import tabgnn_amd as T
    dev = "cuda:0"
    cfg, z = load_case("interleaved_c32_h4_l2")
    C, N, E = cfg["C"], cfg["N"], cfg["E"]
    x, ei, ea = _inputs(cfg, z, C)
    deg = torch.bincount(torch.bincount(ei[1], minlength=N))
    m = T.TABGNNInterleaved(channels=C, num_layers=cfg["L"], deg=deg, node_dim=C, nhidden=C, edge_dim=cfg["ncols"] * C,
                            nhead=cfg["H"], dropout=0.0)
    m.load_state_dict(build_state(cfg["keys"], z, cfg["seed"]))
    m.to(dev).eval()
    with torch.no_grad():
        xg, xe = m(x.to(dev), ei.to(dev), ea.to(dev))
    np.testing.assert_allclose(xg.cpu().numpy(), z["eval.x_gnn"], atol=1e-4)
    np.testing.assert_allclose(xe.cpu().numpy(), z["eval.x_edge"], atol=1e-4)
    m.load_state_dict(build_state(cfg["keys"], z, cfg["seed"]))
    m.train()
    xg, xe = m(x.to(dev), ei.to(dev), ea.to(dev))
    loss = _scalar(cfg, xg, xe, N, E)
    loss.backward()
    assert abs(loss.item() - float(z["train.loss"])) < 1e-4
    for k, p in m.named_parameters():
        ref = float(z[f"gradnorm.{k}"])
        got = 0.0 if p.grad is None else p.grad.double().norm().item()      # unused parameters (edge_emb) stay None
        assert abs(got - ref) < 2e-3 * max(1.0, ref), k
    for k, v in m.state_dict().items():
        if f"bn.{k}" in z.files:
            np.testing.assert_allclose(v.cpu().numpy(), z[f"bn.{k}"], atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["pnas_f32_l2", "pnas_f32_l2_rmp"])
def test_gpu_pnas_matches_reference(name):
    import tabgnn_amd as T
    dev = "cuda:0"
    cfg, z = load_case(name)
    Fh, N, E = cfg["F"], cfg["N"], cfg["E"]
    x, ei, ea = _inputs(cfg, z, Fh)
    deg = torch.bincount(torch.bincount(ei[1], minlength=N))
    m = T.PNAS(num_features=Fh, num_gnn_layers=cfg["L"], n_hidden=Fh, edge_updates=True, edge_dim=cfg["ncols"] * Fh,
               deg=deg, reverse_mp=cfg["reverse_mp"])
    m.load_state_dict(build_state(cfg["keys"], z, cfg["seed"]))
    m.to(dev).eval()
    with torch.no_grad():
        xo, eo = m(x.to(dev), ei.to(dev), ea.to(dev))
    np.testing.assert_allclose(xo.cpu().numpy(), z["eval.x"], atol=1e-4)
    np.testing.assert_allclose(eo.cpu().numpy(), z["eval.edge_attr"], atol=1e-4)
    m.load_state_dict(build_state(cfg["keys"], z, cfg["seed"]))
    m.train()
    xo, eo = m(x.to(dev), ei.to(dev), ea.to(dev))
    loss = _scalar(cfg, xo, eo, N, E)
    loss.backward()
    assert abs(loss.item() - float(z["train.loss"])) < 1e-4
    for k, p in m.named_parameters():
        ref = float(z[f"gradnorm.{k}"])
        got = 0.0 if p.grad is None else p.grad.double().norm().item()      # unused parameters (edge_emb) stay None
        assert abs(got - ref) < 2e-3 * max(1.0, ref), k


def test_oracle_cpna_matches_reference():
    from oracle.siblings import cpna_forward
    cfg, z = load_case("cpna_f32_l2")
    x, ei, ea = _inputs(cfg, z, cfg["F"])
    sd = build_state(cfg["keys"], z, cfg["seed"])
    with torch.no_grad():
        xo, eo = cpna_forward({k: v.clone() for k, v in sd.items()}, x, ei, ea)
    np.testing.assert_allclose(xo.numpy(), z["eval.x"], atol=2e-5)
    np.testing.assert_allclose(eo.numpy(), z["eval.edge_attr"], atol=2e-5)
    for k in sd:
        if sd[k].is_floating_point() and "avg_deg" not in k and "running" not in k:
            sd[k].requires_grad_(True)
    ea_in = ea.clone().requires_grad_(True)
    xo, eo = cpna_forward(sd, x, ei, ea_in, training=True)
    loss = _scalar(cfg, xo, eo, cfg["N"], cfg["E"])
    loss.backward()
    assert abs(loss.item() - float(z["train.loss"])) < 1e-5
    np.testing.assert_allclose(ea_in.grad.numpy(), z["grad.edge_attr"], atol=1e-6)
    for k in sd:
        if f"gradnorm.{k}" in z.files and sd[k].grad is not None:
            assert abs(sd[k].grad.double().norm().item() - float(z[f"gradnorm.{k}"])) < 1e-4 * max(1.0, float(z[f"gradnorm.{k}"])), k


@pytest.mark.gpu
def test_gpu_cpna_matches_reference():
    import tabgnn_amd as T
    dev = "cuda:0"
    cfg, z = load_case("cpna_f32_l2")
    Fh, N, E = cfg["F"], cfg["N"], cfg["E"]
    x, ei, ea = _inputs(cfg, z, Fh)
    deg = torch.bincount(torch.bincount(ei[1], minlength=N))
    m = T.CPNA(num_features=Fh, num_gnn_layers=cfg["L"], n_hidden=Fh, edge_updates=True, edge_dim=cfg["ncols"] * Fh, deg=deg)
    m.load_state_dict(build_state(cfg["keys"], z, cfg["seed"]))
    m.to(dev).eval()
    with torch.no_grad():
        xo, eo = m(x.to(dev), ei.to(dev), ea.to(dev))
    np.testing.assert_allclose(xo.cpu().numpy(), z["eval.x"], atol=1e-4)
    np.testing.assert_allclose(eo.cpu().numpy(), z["eval.edge_attr"], atol=1e-4)
    m.load_state_dict(build_state(cfg["keys"], z, cfg["seed"]))
    m.train()
    ea_in = ea.to(dev).requires_grad_(True)
    xo, eo = m(x.to(dev), ei.to(dev), ea_in)
    loss = _scalar(cfg, xo, eo, N, E)
    loss.backward()
    assert abs(loss.item() - float(z["train.loss"])) < 1e-4
    np.testing.assert_allclose(ea_in.grad.cpu().numpy(), z["grad.edge_attr"], atol=1e-5)
    for k, p in m.named_parameters():
        ref = float(z[f"gradnorm.{k}"])
        got = 0.0 if p.grad is None else p.grad.double().norm().item()
        assert abs(got - ref) < 2e-3 * max(1.0, ref), k
