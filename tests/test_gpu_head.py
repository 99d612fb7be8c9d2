"""The fused readout MLP (csrc/head.hip: ClassifierHead / NodeClassificationHead, src/nn/gnn/decoder.py:5-32) against
the op-by-op composition of the already-pinned kernels (linear / act_dropout / linear / act_dropout / fp32 linear) on
the same dropout masks, and against plain torch fp32 with dropout off."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "models-for-relational-multimodal-data_amd")]

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _run(mlp, h, p, fused, g, seed=11):
    from tabgnn_amd import heads, ops
    for q in mlp.parameters():
        q.grad = None
    ops.DropoutRNG.new_step(seed)
    old = ops.FUSED_HEAD
    ops.FUSED_HEAD = fused
    try:
        x = h.clone().requires_grad_(True)
        out = heads._run_mlp(mlp, x, p)
        out.backward(g)
    finally:
        ops.FUSED_HEAD = old
    return [out.detach(), x.grad] + [q.grad.clone() for q in mlp.parameters()]


def _rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-12))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("p", [0.0, 0.3])
@pytest.mark.parametrize("B,D0,NC", [(1, 384, 2), (5, 384, 2), (200, 384, 2), (8195, 384, 2), (333, 128, 1), (64, 512, 10)])
def test_fused_head_equals_the_op_by_op_composition(dtype, p, B, D0, NC):
    from tabgnn_amd import heads, ops
    torch.manual_seed(B + D0)
    mlp = heads._mlp(D0, NC, p).to(DEV)
    h = torch.randn(B, D0, device=DEV).to(dtype)
    g = torch.randn(B, NC, device=DEV)
    assert ops.head_mlp_ok(mlp, h)
    got = _run(mlp, h, p, True, g)
    want = _run(mlp, h, p, False, g)
    names = ["logits", "d_h", "w1", "b1", "w2", "b2", "w3", "b3"]
    # bf16: both paths round the same quantities, but sum in different orders, so a pre-activation within one bf16 ulp
    # of zero can land on the other side of its ReLU gate: one flipped gate moves a 50-element bias gradient by ~2 %
    tol = 2e-5 if dtype == torch.float32 else 6e-2
    for n, a, b in zip(names, got, want):
        assert a.shape == b.shape and a.dtype == b.dtype, n
        assert _rel(a, b) <= tol, f"{n}: relative error {_rel(a, b):.3e} (B={B}, D0={D0}, {dtype}, p={p})"
    again = _run(mlp, h, p, True, g)
    for n, a, b in zip(names, got, again):
        assert torch.equal(a, b), f"{n} differs run to run"


def test_fused_head_fp32_against_torch_autograd():
    from tabgnn_amd import heads
    torch.manual_seed(0)
    mlp = heads._mlp(384, 2, 0.0).to(DEV)
    h = torch.randn(777, 384, device=DEV)
    g = torch.randn(777, 2, device=DEV)
    got = _run(mlp, h, 0.0, True, g)
    for q in mlp.parameters():
        q.grad = None
    x = h.clone().requires_grad_(True)
    mlp.eval()
    out = mlp(x)
    out.backward(g)
    want = [out.detach(), x.grad] + [q.grad for q in mlp.parameters()]
    for a, b in zip(got, want):
        assert _rel(a, b) <= 2e-5


def test_fused_head_accumulates_into_existing_gradient_buffers():
    """`.grad +=` semantics when every parameter owns an fp32 gradient buffer (FlatParams views): the kernel adds in place
    and autograd receives no gradient tensors."""
    from tabgnn_amd import heads, ops
    torch.manual_seed(1)
    mlp = heads._mlp(384, 2, 0.0).to(DEV)
    h = torch.randn(100, 384, device=DEV).to(torch.bfloat16)
    g = torch.randn(100, 2, device=DEV)
    base = _run(mlp, h, 0.0, True, g)
    for q in mlp.parameters():
        q.grad = torch.ones_like(q)
    ptrs = [q.grad.data_ptr() for q in mlp.parameters()]
    ops.DropoutRNG.new_step(11)
    x = h.clone().requires_grad_(True)
    heads._run_mlp(mlp, x, 0.0).backward(g)
    for q, ptr, b in zip(mlp.parameters(), ptrs, base[2:]):
        assert q.grad.data_ptr() == ptr
        assert torch.allclose(q.grad, b + 1.0, rtol=1e-5, atol=1e-5)
