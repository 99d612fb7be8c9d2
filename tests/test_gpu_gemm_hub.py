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


@pytest.mark.parametrize("R,N,K,flags", [(128, 128, 128, 0), (1000, 128, 128, 1), (4133, 384, 128, 0), (2500, 128, 384, 4),
                                         (777, 128, 768, 1), (3000, 384, 512, 0), (5000, 128, 128, 3),
                                         (3001, 768, 128, 3), (2000, 384, 128, 4), (49152, 128, 128, 0)])   # 12 MiB operands end on their mapping's last page
def test_gemm_nt_fused_epilogue_matches_fp32_reference(R, N, K, flags):
    """tg_gemm_nt_bf16: Y = epilogue(X W^T): bias, ReLU, dropout (same counter stream as tg_act_dropout_fwd, so the
    fused output equals act_dropout applied to the un-fused product), accumulation into an existing Y; ragged R."""
    from tabgnn_amd import _lib as L
    torch.manual_seed(R + N)
    x = (torch.randn(R, K, device=DEV) * 0.5).bfloat16()
    w = (torch.randn(N, K, device=DEV) * 0.1).bfloat16()
    b = torch.randn(N, device=DEV)
    y0 = torch.randn(R, N, device=DEV).bfloat16()
    y = y0.clone() if flags & 4 else torch.full((R, N), float("nan"), device=DEV).bfloat16()
    p, seed, rs = 0.5, 1234567, 7
    L.call("tg_gemm_nt_bf16", L.ptr(x), L.ptr(w), L.ptr(b), None, L.ptr(y), R, N, K, K, N, flags, p, seed, rs, L.stream())
    ref = x.float() @ w.float().t() + b
    if flags & 1:
        ref = ref.relu()
    if flags & 2:                                        # mask of the stand-alone kernel on the same (seed, stream)
        ones = torch.ones(R, N, device=DEV).bfloat16()
        m = torch.empty_like(ones)
        L.call("tg_act_dropout_fwd", L.ptr(ones), L.ptr(m), ones.numel(), 0, p, seed, rs, L.dt(ones), L.stream())
        ref = ref * m.float()
        frac = float((m == 0).float().mean())
        assert 0.45 < frac < 0.55
    if flags & 4:
        ref = ref + y0.float()
    err = (y.float() - ref).abs().max().item()
    assert torch.isfinite(y.float()).all() and err < 0.03 * max(1.0, ref.abs().max().item()), err


def test_gemm_nt_gate_epilogue_is_the_backward_of_relu_dropout():
    """flags & 8: out = (G W^T) * 1/(1-p) where the saved forward output h > 0, else 0 == tg_act_dropout_bwd of the
    un-fused product (h > 0 <=> ReLU active AND kept by the dropout)."""
    from tabgnn_amd import _lib as L, ops
    torch.manual_seed(5)
    R, N, K, p, seed, rs = 3000, 128, 128, 0.5, 99, 3
    pre = torch.randn(R, N, device=DEV).bfloat16()
    h = torch.empty_like(pre)
    L.call("tg_act_dropout_fwd", L.ptr(pre), L.ptr(h), pre.numel(), 1, p, seed, rs, L.dt(pre), L.stream())
    g = (torch.randn(R, K, device=DEV) * 0.5).bfloat16()
    w = (torch.randn(N, K, device=DEV) * 0.1).bfloat16()
    got = ops.gemm_nt(g, w, None, ops.NT_GATE, p, gate=h)
    d_h = (g.float() @ w.float().t()).bfloat16()
    want = torch.empty_like(d_h)
    L.call("tg_act_dropout_bwd", L.ptr(pre), L.ptr(d_h), L.ptr(want), pre.numel(), 1, p, seed, rs, L.dt(pre), L.stream())
    assert torch.equal(got == 0, want == 0)
    assert (got.float() - want.float()).abs().max().item() < 0.05


@pytest.mark.parametrize("R,K,p", [(1000, 128, 0.0), (4133, 128, 0.5), (777, 384, 0.5)])
def test_gemm_nt_ln_equals_gemm_then_layernorm(R, K, p):
    """tg_gemm_nt_ln_bf16 (GEMM + bias + dropout + residual + LayerNorm) against tg_gemm_nt_bf16 followed by tg_ln_fwd
    on the same dropout stream; its z output drives tg_ln_bwd's z mode to the same gradients as the (a, b) form."""
    from tabgnn_amd import _lib as L, ops
    from tabgnn_amd.encoder_layer import _ln_fwd, _ln_bwd
    torch.manual_seed(R)
    C, seed, rs = 128, 4242, 5
    x = (torch.randn(R, K, device=DEV) * 0.5).bfloat16()
    w = (torch.randn(C, K, device=DEV) * 0.1).bfloat16()
    b = torch.randn(C, device=DEV)
    res = torch.randn(R, C, device=DEV).bfloat16()
    gam, bet = torch.rand(C, device=DEV) + 0.5, torch.randn(C, device=DEV)
    z, out, st = ops.gemm_nt_ln(x, w, b, res, gam, bet, p, seed, rs)
    y = ops.gemm_nt(x, w)
    out2, st2 = _ln_fwd(res, y, b, gam, bet, None, 0.0, 1.0, p, seed, rs)
    assert (out.float() - out2.float()).abs().max().item() < 0.06
    assert (st - st2).abs().max().item() < 0.02 * max(1.0, st2.abs().max().item())
    g = torch.randn(R, C, device=DEV).bfloat16()
    da1, da2 = torch.empty_like(res), torch.empty_like(res)
    db1, dp1 = _ln_bwd(z, None, None, gam, st, g, da1, True, None, 0.0, 1.0, p, seed, rs, False)
    db2, dp2 = _ln_bwd(res, y, b, gam, st2, g, da2, True, None, 0.0, 1.0, p, seed, rs, False)
    assert (da1.float() - da2.float()).abs().max().item() < 0.06 * max(1.0, da2.float().abs().max().item())
    assert torch.equal(db1 == 0, db2 == 0) and (db1.float() - db2.float()).abs().max().item() < 0.06 * max(1.0, db2.float().abs().max().item())
    assert (dp1 - dp2).abs().max().item() < 0.02 * max(1.0, dp2.abs().max().item())


@pytest.mark.parametrize("N,E", [(1000, 3000), (4099, 20000), (128, 50), (515080, 438354)])      # (last: the bench graph)
def test_pna_post_projection_with_scalers_inside_the_gemms(N, E):
    """tg_gemm_nt_scaled_bf16 / tg_gemm_tn_scaled_bf16 (+ tg_pna_degree_scalers) against the unfused composition
    G = agg w_st^T, out = x w_x^T + b + G0 + amp*G1 + att*G2 in fp32: output and every gradient.  Isolated nodes (amp = 0),
    a hub destination, N not a multiple of the 128-row tile."""
    from tabgnn_amd import ops
    g = torch.Generator().manual_seed(N)
    F = 128
    src = torch.randint(0, N, (E,), generator=g)
    dst = torch.randint(0, max(N - 7, 1), (E,), generator=g)         # the last nodes stay isolated
    dst[: E // 4] = 3                                                # hub
    graph = ops.SubgraphIndex.build(torch.stack([src, dst]).to(DEV), N)
    avg_log = torch.tensor([1.3], device=DEV)
    x32 = torch.randn(N, F, generator=g).to(DEV)
    wx32 = (torch.randn(F, F, generator=g) / 11.0).to(DEV)
    b32 = torch.randn(F, generator=g).to(DEV)
    agg32 = torch.randn(N, 4 * F, generator=g).to(DEV)
    w32 = (torch.randn(3 * F, 4 * F, generator=g) / 22.0).to(DEV)
    co = torch.randn(N, F, generator=g).to(DEV)
    # fp32 reference on the bf16-rounded operands
    xr = x32.bfloat16().float().requires_grad_(True)
    wxr = wx32.bfloat16().float().requires_grad_(True)
    br = b32.clone().requires_grad_(True)
    ar = agg32.bfloat16().float().requires_grad_(True)
    wr = w32.bfloat16().float().requires_grad_(True)
    deg = torch.bincount(dst, minlength=N).float().to(DEV).view(-1, 1)
    amp = torch.log(deg + 1) / avg_log
    att = avg_log / torch.log(deg.clamp(min=1) + 1)
    G = ar @ wr.t()
    ref = xr @ wxr.t() + br + G[:, :F] + amp * G[:, F:2 * F] + att * G[:, 2 * F:]
    (ref * co).sum().backward()
    # fused
    x = x32.bfloat16().requires_grad_(True)
    wx = wx32.bfloat16().float().requires_grad_(True)
    b = b32.clone().requires_grad_(True)
    agg = agg32.bfloat16().requires_grad_(True)
    w = w32.bfloat16().float().requires_grad_(True)
    assert ops.post_scaled_ok(x, agg)
    out = ops.pna_post_scaled(x, wx, b, agg, w, graph, avg_log)
    (out.float() * co).sum().backward()
    rel = lambda a, b: ((a.float() - b).norm() / b.norm().clamp_min(1e-12)).item()
    assert rel(out, ref.detach()) < 0.01, rel(out, ref.detach())
    assert rel(x.grad, xr.grad) < 0.01
    assert rel(wx.grad, wxr.grad) < 0.01 and rel(b.grad, br.grad) < 0.01
    assert rel(agg.grad, ar.grad) < 0.02, rel(agg.grad, ar.grad)
    assert rel(w.grad, wr.grad) < 0.02, rel(w.grad, wr.grad)
    sc = ops.degree_scalers(graph, avg_log)
    assert sc.shape[0] % 128 == 0 and sc.shape[0] >= N
    torch.testing.assert_close(sc[:N, 0:1], amp, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(sc[:N, 1:2], att, rtol=1e-5, atol=1e-6)
