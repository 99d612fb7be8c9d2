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


def _gine_inputs(cfg, z):
    seed, N, E, nc, Fh = cfg["seed"], cfg["N"], cfg["E"], cfg["ncols"], cfg["F"]
    return (det_tensor("in.x", (N, Fh), seed), torch.from_numpy(z["edge_index"].astype(np.int64)),
            det_tensor("in.edge_attr", (E, nc, Fh), seed))


def _gine_state(cfg, z):
    """The two directions of a GINEConvHetero share ONE network (gine.py:18-19): the state dict lists it under both
    prefixes and the value written last (``conv_back``) is the one both use; gradients add up on the shared tensor."""
    sd = build_state(cfg["keys"], z, cfg["seed"])
    for k in list(sd):
        if ".conv_forw.nn." in k:
            sd[k] = sd[k.replace(".conv_forw.nn.", ".conv_back.nn.")]
    return sd


@pytest.mark.parametrize("name", ["gine_f32_l2", "gine_f32_l2_rmp"])
def test_oracle_gine_matches_reference(name):
    from oracle.gine import gine_forward
    cfg, z = load_case(name)
    x, ei, ea = _gine_inputs(cfg, z)
    sd = _gine_state(cfg, z)
    with torch.no_grad():
        xo, eo = gine_forward({k: v.clone() for k, v in sd.items()}, x, ei, ea)
    np.testing.assert_allclose(xo.numpy(), z["eval.x"], atol=2e-5)
    np.testing.assert_allclose(eo.numpy(), z["eval.edge_attr"], atol=2e-5)
    for k in sd:
        if sd[k].is_floating_point() and "running" not in k and not k.endswith("eps"):
            sd[k].requires_grad_(True)
    xo, eo = gine_forward(sd, x, ei, ea, training=True)
    loss = _scalar(cfg, xo, eo, cfg["N"], cfg["E"])
    loss.backward()
    assert abs(loss.item() - float(z["train.loss"])) < 1e-5
    checked = 0
    for k in sd:
        if f"gradnorm.{k}" in z.files and sd[k].grad is not None:
            ref = float(z[f"gradnorm.{k}"])
            assert abs(sd[k].grad.double().norm().item() - ref) < 1e-4 * max(1.0, ref), k
            checked += 1
    assert checked >= 10


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["gine_f32_l2", "gine_f32_l2_rmp"])
def test_gpu_gine_matches_reference(name):
    import tabgnn_amd as T
    dev = "cuda:0"
    cfg, z = load_case(name)
    Fh, N, E = cfg["F"], cfg["N"], cfg["E"]
    x, ei, ea = _gine_inputs(cfg, z)
    m = T.GINe(num_features=Fh, num_gnn_layers=cfg["L"], n_hidden=Fh, edge_updates=True, edge_dim=cfg["ncols"] * Fh,
               reverse_mp=cfg["reverse_mp"])
    assert set(m.state_dict().keys()) == set(cfg["keys"].keys())
    m.load_state_dict(build_state(cfg["keys"], z, cfg["seed"]))
    m.to(dev).eval()
    with torch.no_grad():
        xo, eo = m(x.to(dev), ei.to(dev), ea.to(dev))
    np.testing.assert_allclose(xo.cpu().numpy(), z["eval.x"], atol=1e-4)
    np.testing.assert_allclose(eo.cpu().numpy(), z["eval.edge_attr"], atol=1e-4)
    m.load_state_dict(build_state(cfg["keys"], z, cfg["seed"]))
    m.train()
    xo, eo = m(x.to(dev), ei.to(dev), ea.to(dev))
    np.testing.assert_allclose(xo.detach().cpu().numpy(), z["train.x"], atol=1e-4)
    loss = _scalar(cfg, xo, eo, N, E)
    loss.backward()
    assert abs(loss.item() - float(z["train.loss"])) < 1e-4
    for k, p in m.named_parameters():
        ref = float(z[f"gradnorm.{k}"])
        got = 0.0 if p.grad is None else p.grad.double().norm().item()
        assert abs(got - ref) < 2e-3 * max(1.0, ref), k


@pytest.mark.gpu
def test_gpu_gine_aggregate_hubs_and_bf16():
    """tg_gine_aggregate_fwd / tg_gine_message_bwd against a torch restatement on a graph with hub destinations
    (> 256 in-edges: the block-per-hub pass) and hub sources, fp32 tight and bf16 at its rounding tolerance."""
    from tabgnn_amd import ops
    dev = "cuda:0"
    g = torch.Generator().manual_seed(5)
    N, E, F = 3000, 40000, 128
    src = torch.randint(0, N, (E,), generator=g)
    dst = torch.randint(0, N, (E,), generator=g)
    dst[:6000] = 7; dst[6000:6300] = 11; src[20000:29000] = 3          # hub destinations 7 (6000 rows), 11; hub source 3
    ei = torch.stack([src, dst]).to(dev)
    graph = ops.SubgraphIndex.build(ei, N)
    x32 = torch.randn(N, F, generator=g).to(dev)
    le32 = torch.randn(E, F, generator=g).to(dev)
    co = torch.randn(N, F, generator=g).to(dev)
    for dt, tol, gtol in ((torch.float32, 2e-3, 2e-3), (torch.bfloat16, 0.02, 0.02)):
        for scale in (1.25, 0.0):
            x = x32.to(dt).clone().requires_grad_(True); le = le32.to(dt).clone().requires_grad_(True)
            out = ops.gine_aggregate(x, le, graph, scale)
            (out.float() * co).sum().backward()
            xr = x.detach().float().requires_grad_(True); lr = le.detach().float().requires_grad_(True)
            ref = torch.zeros(N, F, device=dev).index_add_(0, ei[1], torch.relu(xr[ei[0]] + lr)) + scale * xr
            (ref * co).sum().backward()
            rel = lambda a, b: ((a.float() - b).norm() / b.norm().clamp_min(1e-12)).item()
            assert rel(out, ref.detach()) < tol, (dt, scale, rel(out, ref.detach()))
            assert rel(x.grad, xr.grad) < gtol and rel(le.grad, lr.grad) < gtol, (dt, scale)
    # determinism: same inputs, bit-identical outputs
    a = ops.gine_aggregate(x32, le32, graph, 1.0)
    b = ops.gine_aggregate(x32, le32, graph, 1.0)
    assert torch.equal(a, b)
