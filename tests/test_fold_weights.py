"""Host logic (pure torch ops, no GPU): the PNAConv weight folds as one autograd node (ops.fold_pna_weights) against
the op-by-op composition under plain autograd — outputs, every gradient, both aggregator orders (fused.py:198 and
pna.py:59), and the in-place accumulation into existing ``.grad`` buffers."""
import pytest
import torch


def _mk(F, Fe, params=False):
    g = torch.Generator().manual_seed(F + Fe)
    shapes = [(F, 3 * F), (F,), (F, Fe), (F,), (F, 13 * F), (F,), (F, F), (F,)]
    ts = [torch.randn(*s, generator=g, dtype=torch.float32) for s in shapes]
    return [torch.nn.Parameter(t) if params else t.requires_grad_(True) for t in ts]


def _reference(P, pb, We, be, Qw, qb, Lw, lb, order):
    F = P.shape[0]
    w_msg = torch.cat([P[:, :2 * F], P[:, 2 * F:] @ We], 1)
    b_msg = pb + P[:, 2 * F:] @ be
    w_eff = Lw @ Qw
    b_eff = Lw @ qb + lb
    blk = lambda sc, j: w_eff[:, F + (sc * 4 + j) * F:F + (sc * 4 + j + 1) * F]
    w_st = torch.cat([torch.cat([blk(sc, j) for j in order], 1) for sc in range(3)], 0)
    return w_msg, b_msg, w_eff[:, :F], b_eff, w_st


@pytest.mark.parametrize("order", [[0, 1, 2, 3], [0, 2, 1, 3]])
@pytest.mark.parametrize("with_grad_buffers", [False, True])
def test_fold_matches_autograd(order, with_grad_buffers):
    from tabgnn_amd import ops
    F, Fe = 16, 16
    a, b = _mk(F, Fe, params=with_grad_buffers), _mk(F, Fe)
    if with_grad_buffers:                      # FlatParams-style: fp32 gradient buffers exist before the backward
        for p in a:
            p.grad = torch.full_like(p, 0.5)
    outs = ops.fold_pna_weights(*a, order)
    refs = _reference(*b, order)
    cos = [torch.randn_like(o) for o in refs]
    for o, r in zip(outs, refs):
        torch.testing.assert_close(o, r, rtol=1e-5, atol=1e-5)
    sum((o * c).sum() for o, c in zip(outs, cos)).backward()
    sum((o * c).sum() for o, c in zip(refs, cos)).backward()
    for p, q in zip(a, b):
        want = q.grad + (0.5 if with_grad_buffers else 0.0)
        torch.testing.assert_close(p.grad, want, rtol=1e-4, atol=1e-4)
