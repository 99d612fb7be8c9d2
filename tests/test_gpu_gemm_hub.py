"""GPU: split-row weight-gradient GEMM (MFMA + transposed LDS reads) and hub-node handling of the segmented sums."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("R,M,N", [(4096, 128, 128), (5000, 384, 128), (70001, 128, 768), (9000, 8, 8),
                                   (12345, 136, 200), (300000, 128, 384), (8192, 1536, 384)])
def test_weight_grad_gemm_matches_fp32_reference(R, M, N):
    from tabgnn_amd import ops
    torch.manual_seed(R % 1000 + M + N)
    # integer-valued bf16 operands: products and sums are exact in fp32 -> the MFMA layout is checked bit for bit
    g = torch.randint(-3, 4, (R, M), device=DEV).to(torch.bfloat16)
    x = torch.randint(-3, 4, (R, N), device=DEV).to(torch.bfloat16)
    got, db = ops.weight_grad(g, x, True)
    want = g.double().t() @ x.double()
    assert got.dtype == torch.float32 and got.shape == (M, N)
    assert torch.equal(got.double(), want), (got.double() - want).abs().max().item()
    assert torch.equal(db.double(), g.double().sum(0))
    # real-valued operands: fp32 accumulation over R rows
    g = torch.randn(R, M, device=DEV).to(torch.bfloat16)
    x = torch.randn(R, N, device=DEV).to(torch.bfloat16)
    got, _ = ops.weight_grad(g, x)
    want = g.double().t() @ x.double()
    err = (got.double() - want).abs().max().item()
    assert err <= 2e-5 * R ** 0.5 * 4 + 1e-3, err


def test_weight_grad_gemm_strided_views():
    from tabgnn_amd import ops
    R = 10000
    big = torch.randint(-2, 3, (R, 384), device=DEV).to(torch.bfloat16)
    g = big[:, 256:]                       # row stride 384, 16-byte aligned view
    x = torch.randint(-2, 3, (R, 64), device=DEV).to(torch.bfloat16)
    got, db = ops.weight_grad(g, x, True)
    assert torch.equal(got.double(), g.double().t() @ x.double())
    assert torch.equal(db.double(), g.double().sum(0))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_edge_gather_backward_with_hub_nodes(dtype):
    """A node with thousands of incident edges takes the block-per-hub path; result equals index_add."""
    from tabgnn_amd import ops
    torch.manual_seed(7)
    N, E, F = 500, 20000, 64
    src = torch.randint(0, N, (E,))
    dst = torch.randint(0, N, (E,))
    src[:6000] = 3                           # out-degree hub
    dst[5000:9000] = 11                      # in-degree hub
    ei = torch.stack([src, dst]).to(DEV)
    g = ops.SubgraphIndex.build(ei, N)
    x = torch.randn(N, F, device=DEV, dtype=dtype).requires_grad_(True)
    e = torch.randn(E, F, device=DEV, dtype=dtype).requires_grad_(True)
    out = ops.edge_gather(x, e, g, "dst")
    assert torch.equal(out[:, :F], x.detach()[ei[1]]) and torch.equal(out[:, F:2 * F], x.detach()[ei[0]])
    go = torch.randint(-2, 3, out.shape, device=DEV).to(dtype)   # exact sums in either dtype
    out.backward(go)
    want = torch.zeros(N, F, device=DEV, dtype=torch.float64)
    want.index_add_(0, ei[1], go[:, :F].double())
    want.index_add_(0, ei[0], go[:, F:2 * F].double())
    assert torch.equal(x.grad.double(), want.to(dtype).double())
    assert torch.equal(e.grad, go[:, 2 * F:])


@pytest.mark.gpu
def test_weight_grad_accumulates_into_existing_grad_buffer():
    """.grad += semantics in the kernel: with a gradient buffer present the result is added to it (twice here) and
    (None, None) comes back, so autograd's AccumulateGrad add never runs."""
    import torch
    from tabgnn_amd import ops
    torch.manual_seed(3)
    R, M, N = 9000, 128, 384
    g = torch.randn(R, M, device="cuda").bfloat16()
    x = torch.randn(R, N, device="cuda").bfloat16()
    w = torch.nn.Parameter(torch.zeros(M, N, device="cuda"))
    b = torch.nn.Parameter(torch.zeros(M, device="cuda"))
    w.grad = torch.full((M, N), 0.5, device="cuda")
    b.grad = torch.full((M,), -1.0, device="cuda")
    for _ in range(2):
        assert ops.weight_grad(g, x, True, w, b) == (None, None)
    ref = g.float().t() @ x.float()
    assert torch.allclose(w.grad, 0.5 + 2 * ref, rtol=1e-3, atol=1e-2)
    assert torch.allclose(b.grad, -1.0 + 2 * g.float().sum(0), rtol=1e-3, atol=1e-2)
