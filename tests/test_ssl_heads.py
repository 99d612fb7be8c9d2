"""LP / MCM pre-training heads and losses against tests/golden/ssl_heads_c32.npz, generated from
the reference's own LinkPredHead (decoder.py:34-72), MCMHead (self_supervised.py:134-171) and SSLoss (loss.py:5-72).
CPU: the oracle restatement.  GPU: the product modules through the C ABI."""
import numpy as np
import pytest
import torch

from golden_util import build_state, load_case
from detparams import det_tensor


def _inputs(cfg, z, dev="cpu"):
    seed, N, P, k, C, Fh = cfg["seed"], cfg["N"], cfg["P"], cfg["k"], cfg["C"], cfg["F"]
    x = det_tensor("in.x_gnn", (N, Fh), seed).to(dev).requires_grad_(True)
    xt = det_tensor("in.x_tab", (P, C), seed).to(dev).requires_grad_(True)
    pe = det_tensor("in.pos_attr", (P, Fh), seed).to(dev)
    ne = pe.repeat_interleave(k, 0)
    pos, neg = torch.from_numpy(z["pos"]).to(dev), torch.from_numpy(z["neg"]).to(dev)
    return x, xt, pos, pe, neg, ne, torch.from_numpy(z["y"]).to(dev)


def _check(z, pp, nprd, num_out, cat_out, l_lp, mcm, x, xt, tol):
    l_mcm, (cl, tc, acc), (nl, tn) = mcm
    np.testing.assert_allclose(pp.detach().cpu().numpy(), z["pos_pred"], atol=tol)
    np.testing.assert_allclose(nprd.detach().cpu().numpy(), z["neg_pred"], atol=tol)
    np.testing.assert_allclose(num_out.detach().cpu().numpy(), z["num_out"], atol=10 * tol)
    for i, c in enumerate(cat_out):
        np.testing.assert_allclose(c.detach().cpu().numpy(), z[f"cat_out.{i}"], atol=10 * tol)
    assert abs(float(l_lp.detach()) - float(z["lp_loss"])) < 10 * tol and abs(float(l_mcm.detach()) - float(z["mcm_loss"])) < 10 * tol
    assert (tc, tn) == (int(z["t_c"]), int(z["t_n"])) and float(acc) == float(z["acc"])      # counts: exact
    assert abs(float(cl.detach()) - float(z["cat_loss"])) < 1e-3 and abs(float(nl.detach()) - float(z["num_loss"])) < 1e-3
    (l_lp + l_mcm).backward()
    np.testing.assert_allclose(x.grad.cpu().numpy(), z["grad.x_gnn"], atol=tol)
    np.testing.assert_allclose(xt.grad.cpu().numpy(), z["grad.x_tab"], atol=tol)


def test_oracle_ssl_heads_and_losses_match_reference():
    from oracle import ssl as O
    cfg, z = load_case("ssl_heads_c32")
    lp_sd = build_state(cfg["lp_keys"], z, cfg["seed"])
    mcm_sd = build_state(cfg["mcm_keys"], z, cfg["seed"] + 1)
    x, xt, pos, pe, neg, ne, y = _inputs(cfg, z)
    pp, nprd = O.link_pred_head(x, pos, pe, neg, ne, lp_sd)
    num_out, cat_out = O.mcm_head(xt, mcm_sd, len(cfg["cards"]))
    _check(z, pp, nprd, num_out, cat_out, O.lp_loss(pp, nprd), O.mcm_loss(cat_out, num_out, y, cfg["n_num"]), x, xt, 1e-5)


@pytest.mark.gpu
def test_gpu_ssl_heads_and_losses_match_reference():
    import tabgnn_amd as T
    dev = "cuda:0"
    cfg, z = load_case("ssl_heads_c32")
    lp = T.LinkPredHead(1, cfg["F"], dropout=0.0)
    mcm = T.MCMHead(cfg["C"], cfg["n_num"], cfg["cards"])
    lp.load_state_dict(build_state(cfg["lp_keys"], z, cfg["seed"]))          # reference key names load as they are
    mcm.load_state_dict(build_state(cfg["mcm_keys"], z, cfg["seed"] + 1))
    lp.to(dev).train(); mcm.to(dev).train()
    x, xt, pos, pe, neg, ne, y = _inputs(cfg, z, dev)
    pp, nprd = lp(x, pos, pe, neg, ne)
    num_out, cat_out = mcm(xt)
    L = T.SSLoss(dev, cfg["n_num"])
    _check(z, pp, nprd, num_out, cat_out, L.lp_loss(pp, nprd), L.mcm_loss(cat_out, num_out, y), x, xt, 1e-4)
    for pfx, m in (("lp.", lp), ("mcm.", mcm)):
        for k, p in m.named_parameters():
            assert abs(p.grad.double().norm().item() - float(z[f"gradnorm.{pfx}{k}"])) < 1e-3 * max(1.0, float(z[f"gradnorm.{pfx}{k}"]))
