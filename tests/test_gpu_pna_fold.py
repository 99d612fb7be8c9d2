"""GPU: the one-launch PNAConv weight folds (tg_pna_fold_fwd / tg_pna_fold_bwd) against the torch composition of the
same algebra (ops._FoldPNAWeights, itself pinned by the golden PNA fixtures): outputs, the bf16 operand layouts, and the
eight parameter gradients in both delivery modes (returned tensors / accumulated into existing .grad buffers)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _params(F, Fe, seed, as_param=False):
    g = torch.Generator().manual_seed(seed)
    shapes = [(F, 3 * F), (F,), (F, Fe), (F,), (F, 13 * F), (F,), (F, F), (F,)]
    ts = [(torch.randn(*s, generator=g) * 0.1).to(DEV) for s in shapes]
    if as_param:
        ts = [torch.nn.Parameter(t) for t in ts]
    else:
        for t in ts:
            t.requires_grad_(True)
    return ts


@pytest.mark.parametrize("F,Fe,order", [(128, 128, (0, 1, 2, 3)), (128, 128, (0, 2, 1, 3)), (64, 32, (3, 1, 0, 2)), (96, 64, (1, 0, 3, 2)),
                                        (32, 128, (0, 1, 2, 3))])
def test_fold_kernels_equal_the_torch_composition(F, Fe, order):
    from tabgnn_amd import ops
    ps_a, ps_b = _params(F, Fe, 3), _params(F, Fe, 3)
    outs_a = ops._FoldPNAWeightsHIP.apply(*ps_a, order, True)
    packs = ops._FoldPNAWeightsHIP.last_packs
    outs_b = ops._FoldPNAWeights.apply(*ps_b, order)
    names = ("w_msg", "b_msg", "w_x", "b_eff", "w_st")
    for n, a, b in zip(names, outs_a, outs_b):
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-6, msg=n)
    w_msg, _, w_x, _, w_st = outs_a
    K = 4 * F
    bf = lambda t: t.detach().to(torch.bfloat16)
    assert torch.equal(packs[0], bf(w_msg)) and torch.equal(packs[1], bf(w_msg).t().contiguous())
    assert torch.equal(packs[2], bf(w_x)) and torch.equal(packs[3], bf(w_x).t().contiguous())
    w_lp = bf(w_st).view(3, F, K)
    assert torch.equal(packs[4], w_lp.view(3, F, K // 128, 128).permute(1, 2, 0, 3).reshape(F, 3 * K))
    assert torch.equal(packs[5], w_lp.permute(2, 0, 1).reshape(K, 3 * F))
    g = torch.Generator().manual_seed(9)
    cots = [torch.randn(o.shape, generator=g).to(DEV) for o in outs_a]
    torch.autograd.backward(outs_a, cots)
    torch.autograd.backward(outs_b, cots)
    pn = ("P", "pb", "We", "be", "Qw", "qb", "Lw", "lb")
    for n, a, b in zip(pn, ps_a, ps_b):
        scale = b.grad.abs().max().item()
        assert (a.grad - b.grad).abs().max().item() <= 2e-5 * scale + 1e-7, n


def test_fold_backward_accumulates_into_existing_grad_buffers_and_takes_missing_cotangents():
    from tabgnn_amd import ops
    F = 128
    ps = _params(F, F, 5, as_param=True)
    ref = _params(F, F, 5)
    for p in ps:
        p.grad = torch.full_like(p, 0.25)
    outs = ops._FoldPNAWeightsHIP.apply(*ps, (0, 1, 2, 3), False)
    outs_r = ops._FoldPNAWeights.apply(*ref, (0, 1, 2, 3))
    # only three of the five folded tensors receive a gradient (e.g. a layer whose message path is detached)
    g = torch.Generator().manual_seed(2)
    pick = (2, 3, 4)
    cots = [torch.randn(outs[i].shape, generator=g).to(DEV) for i in pick]
    torch.autograd.backward([outs[i] for i in pick], cots)
    torch.autograd.backward([outs_r[i] for i in pick], cots)
    for p, r in zip(ps, ref):
        want = 0.25 + (r.grad if r.grad is not None else torch.zeros_like(r))
        assert (p.grad - want).abs().max().item() <= 2e-5 * want.abs().max().item() + 1e-7
