"""GPU: the small round-2 operators and modes through the C ABI — dropout masks in both hash modes (8 bits per element
when p is a multiple of 1/256, else 16), gradient delivery into shared buffers (tg_segment_sum2 accumulate,
tg_rows_add), and the gather-fused GEMMs at ragged / tiny row counts."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("p", [0.5, 0.25, 0.083, 0.1])
def test_dropout_masks_keep_rate_independence_and_backward_agreement(p):
    """keep rate within 4 sigma of 1-p (so the 8-bit mode is exact for p = k/256 and the 16-bit mode for the rest),
    neighbouring elements and neighbouring rows uncorrelated, a new stream gives a new mask, and the backward drops
    exactly the elements the forward dropped."""
    from tabgnn_amd import ops
    R, C = 4096, 512
    x = torch.ones(R, C, device=DEV, dtype=torch.bfloat16, requires_grad=True)
    ops.DropoutRNG.new_step(4242)
    y = ops.act_dropout(x, "none", p)
    y.sum().backward()
    keep = (y.detach().float() > 0)
    k = keep.float()
    n = R * C
    rate = k.mean().item()
    assert abs(rate - (1 - p)) < 4 * np.sqrt(p * (1 - p) / n), (rate, p)
    np.testing.assert_allclose(y.detach().float().max().item(), 1 / (1 - p), rtol=1e-2)
    assert torch.equal(x.grad.float() > 0, keep)                         # same mask recomputed in the backward
    c = k - rate
    var = c.var().item()
    for a, b in ((c[:, :-1], c[:, 1:]), (c[:-1], c[1:]), (c[:, :-3], c[:, 3:])):
        assert abs((a * b).mean().item() / var) < 5 / np.sqrt(n), "correlated neighbours"
    assert abs(k.mean(0).std().item() - np.sqrt(p * (1 - p) / R)) < 0.3 * np.sqrt(p * (1 - p) / R)
    y2 = ops.act_dropout(x, "none", p)                                   # next stream of the same step
    assert (y2.detach() > 0).ne(keep).float().mean().item() > 0.5 * 2 * p * (1 - p)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_segment_sum_accumulate_and_rows_add_deliver_into_a_shared_buffer(dtype):
    from tabgnn_amd import ops, _lib as L
    rs = np.random.RandomState(0)
    N, E, F = 3000, 20000, 128
    src, dst = rs.randint(0, N, E), rs.randint(0, N, E)
    dst[:5000] = 11                                                      # a hub (block-per-hub pass)
    dst[dst == 17] = 18                                                  # node 17 has no in-edge: its row is not touched
    g = ops.SubgraphIndex.build(torch.from_numpy(np.stack([src, dst])).to(DEV), N)
    grad = torch.randn(E, 3 * F, device=DEV).to(dtype)
    base = torch.randn(N, F, device=DEV).to(dtype)
    want = base.float().clone()
    want.index_add_(0, g.dst.long(), grad[:, :F].float())
    want.index_add_(0, g.src.long(), grad[:, F:2 * F].float())
    buf = base.clone()
    hub = torch.empty(L.load().tg_segment_hub_ints(2 * E), dtype=torch.int32, device=DEV)
    L.call("tg_segment_sum2", L.ptr(grad), 3 * F, 0, L.ptr(g.by_dst[0]), L.ptr(g.by_dst[1]), F, L.ptr(g.by_src[0]),
           L.ptr(g.by_src[1]), 0, None, L.ptr(buf), N, F, L.ptr(hub), 1, L.dt(grad), L.stream())
    tol = 1e-4 if dtype == torch.float32 else 4e-2
    assert (buf.float() - want).abs().max().item() <= tol * want.abs().max().item()
    lone = int(np.setdiff1d(np.arange(N), np.concatenate([src, dst]))[0]) if len(np.setdiff1d(np.arange(N), np.concatenate([src, dst]))) else None
    if lone is not None:
        assert torch.equal(buf[lone], base[lone])                        # rows with empty segments are not rewritten
    # the edge third, row-gathered, added to / written into a buffer
    idx = torch.from_numpy(rs.permutation(E).astype(np.int32)).to(DEV)
    ebase = torch.randn(E, F, device=DEV).to(dtype)
    for acc in (0, 1):
        out = ebase.clone()
        L.call("tg_rows_add", L.ptr(out), grad[:, 2 * F:].data_ptr(), L.ptr(idx), E, F, 3 * F, acc, L.dt(grad), L.stream())
        ref = grad[idx.long(), 2 * F:].float() + (ebase.float() if acc else 0)
        assert (out.float() - ref).abs().max().item() <= (1e-6 if dtype == torch.float32 else 2e-2) * ref.abs().max().item()
    out = torch.empty(E, F, device=DEV, dtype=dtype)
    L.call("tg_rows_add", L.ptr(out), grad[:, 2 * F:].data_ptr(), None, E, F, 3 * F, 0, L.dt(grad), L.stream())
    assert torch.equal(out, grad[:, 2 * F:].contiguous())


@pytest.mark.parametrize("E", [1, 63, 64, 65, 127, 129, 2049, 4097])
def test_gather_gemms_at_ragged_row_counts(E):
    """One row, one short of / one past the 64-row step and the 128-row tile, one past the 2048-row index table of a
    weight-gradient slab: forward against the materialised fp32 product, weight gradient and bias gradient against
    torch, accumulate mode on top of an existing gradient."""
    import ctypes as C
    from tabgnn_amd import _lib as L
    torch.manual_seed(E)
    N, F = 500, 128
    x = (torch.randn(N, F, device=DEV) * 0.5).to(torch.bfloat16)
    e = (torch.randn(E, F, device=DEV) * 0.5).to(torch.bfloat16)
    ia = torch.randint(0, N, (E,), device=DEV, dtype=torch.int32)
    ib = torch.randint(0, N, (E,), device=DEV, dtype=torch.int32)
    w = (torch.randn(F, 3 * F, device=DEV) * 0.1).to(torch.bfloat16)
    b = torch.randn(F, device=DEV)
    gs = L.Gather3()
    for c, (t, i) in enumerate(((x, ia), (x, ib), (e, None))):
        gs.src[c], gs.idx[c], gs.stride[c] = t.data_ptr(), (None if i is None else i.data_ptr()), t.stride(0)
    y = torch.empty(E, F, device=DEV, dtype=torch.bfloat16)
    L.call("tg_gemm_nt_gather3_bf16", C.byref(gs), L.ptr(w), L.ptr(b), L.ptr(y), E, F, F, 0, L.stream())
    cat = torch.cat([x[ia.long()], x[ib.long()], e], 1).float()
    ref = cat @ w.float().t() + b
    assert (y.float() - ref).abs().max().item() <= 2e-2 * max(ref.abs().max().item(), 1.0)
    g = (torch.randn(E, F, device=DEV)).to(torch.bfloat16)
    for acc in (0, 1):
        dw = torch.full((F, 3 * F), 0.5, device=DEV)
        db = torch.full((F,), -0.25, device=DEV)
        ws = torch.empty(L.load().tg_gemm_tn_gather3_workspace_floats(E, F), device=DEV)
        L.call("tg_gemm_tn_gather3_bf16", L.ptr(g), C.byref(gs), L.ptr(dw), L.ptr(db), L.ptr(ws), E, F, F, acc, L.stream())
        rw = g.float().t() @ cat + (0.5 if acc else 0.0)
        rb = g.float().sum(0) + (-0.25 if acc else 0.0)
        assert (dw - rw).abs().max().item() <= 1e-4 * max(rw.abs().max().item(), 1.0)
        assert (db - rb).abs().max().item() <= 1e-4 * max(rb.abs().max().item(), 1.0)


@pytest.mark.parametrize("R,lazy", [(1, False), (33, False), (5000, True)])
def test_timestamp_columns_as_gemms_agree_with_the_vector_kernels(R, lazy):
    """bf16, C = 128: the timestamp encoder through tg_encode_ts_features + the NT / TN GEMMs against the per-row
    kernels (k_encode_ts_fwd / k_encode_ts_bwd, themselves pinned by the oracle tests): outputs within bf16 rounding of
    the 56-term products, parameter gradients within 1 %; every other column bit-identical; ids into a larger table."""
    from tabgnn_amd import encoders, synthetic as S
    from tabgnn_amd.frame import TensorFrame, stype
    num, cat, ts = S.edge_table(max(R, 64) * (3 if lazy else 1), 5)
    feat = {stype.numerical: torch.from_numpy(num).to(DEV), stype.categorical: torch.from_numpy(cat).to(DEV),
            stype.timestamp: torch.from_numpy(ts).to(DEV)}
    if lazy:
        ids = torch.randperm(feat[stype.numerical].shape[0], device=DEV)[:R].contiguous()
        tf = TensorFrame(feat, S.EDGE_COLS, row_ids=ids)
    else:
        tf = TensorFrame({k: v[:R].contiguous() for k, v in feat.items()}, S.EDGE_COLS)
    torch.manual_seed(1)
    enc = encoders.StypeWiseFeatureEncoder(128, S.EDGE_STATS, S.EDGE_COLS, torch.bfloat16).to(DEV)
    with torch.no_grad():
        enc.encoder_dict["timestamp"].weight.normal_(0, 0.3)
        enc.encoder_dict["timestamp"].bias.normal_(0, 0.3)
    cot = torch.randn(R, 5, 128, device=DEV, generator=torch.Generator(DEV).manual_seed(2)).to(torch.bfloat16)
    res = []
    try:
        for on in (True, False):
            encoders._TS_GEMM = on
            for p in enc.parameters():
                p.grad = None
            x, _ = enc(tf)
            (x.float() * cot.float()).sum().backward()
            res.append((x.detach().float().clone(), {k: p.grad.clone() for k, p in enc.named_parameters()}))
    finally:
        encoders._TS_GEMM = True
    a, b = res
    ts_col = 4                                             # columns: 1 numerical, 3 categorical, 1 timestamp
    assert torch.equal(a[0][:, :ts_col], b[0][:, :ts_col])
    scale = b[0][:, ts_col].abs().max().item()
    assert (a[0][:, ts_col] - b[0][:, ts_col]).abs().max().item() <= 3e-2 * scale
    for k in b[1]:
        ga, gb = a[1][k], b[1][k]
        if "timestamp" in k:
            assert (ga - gb).norm().item() <= 1e-2 * gb.norm().item() + 1e-6, k
        else:
            assert torch.equal(ga, gb), k


def _ref_tn(g, x, scale=None):
    out = torch.zeros(g.shape[1], x.shape[1], device=DEV, dtype=torch.float32)
    for i in range(0, g.shape[0], 1 << 18):
        gg = g[i:i + (1 << 18)].float()
        if scale is not None:
            gg = (gg * scale[i:i + (1 << 18), None]).bfloat16().float()
        out += gg.t() @ x[i:i + (1 << 18)].float()
    return out


@pytest.mark.parametrize("R,M,N", [(2580972, 384, 128), (430162, 128, 768), (430162, 128, 384), (8192, 1536, 1536),
                                   (100003, 384, 256), (300, 384, 128), (129, 128, 384)])
def test_wide_weight_gradient_kernel_at_the_step_shapes(R, M, N):
    """tg_gemm_tn_bf16 on the shapes that take the 384 x 128-block kernel (wide on either side), at the bench's full row
    counts: weight and bias gradient against a chunked fp32 product, and accumulate mode."""
    from tabgnn_amd import _lib as L
    torch.manual_seed(R % 1000)
    g = torch.randn(R, M, device=DEV, dtype=torch.bfloat16)
    x = torch.randn(R, N, device=DEV, dtype=torch.bfloat16)
    ref, rb = _ref_tn(g, x), g.float().sum(0)
    ws = torch.empty(L.load().tg_gemm_tn_workspace_floats(R, M, N), device=DEV, dtype=torch.float32)
    for acc in (0, 1):
        out = torch.full((M, N), 2.0, device=DEV)
        db = torch.full((M,), -1.0, device=DEV)
        L.call("tg_gemm_tn_bf16", L.ptr(g), L.ptr(x), L.ptr(out), L.ptr(db), L.ptr(ws), R, M, N, M, N, acc, L.stream())
        want, wantb = ref + (2.0 if acc else 0.0), rb + (-1.0 if acc else 0.0)
        assert (out - want).abs().max().item() <= 2e-5 * want.abs().max().item()
        assert (db - wantb).abs().max().item() <= 2e-5 * max(wantb.abs().max().item(), 1.0)


def test_scaled_post_projection_weight_gradient_at_the_bench_size():
    from tabgnn_amd import _lib as L
    N, F, K = 524165, 128, 512
    torch.manual_seed(3)
    g = torch.randn(N, F, device=DEV, dtype=torch.bfloat16)
    agg = torch.randn(N, K, device=DEV, dtype=torch.bfloat16)
    scales = (torch.rand(N, 2, device=DEV) * 2 + 0.1).float()
    scales[::7, 0] = 0.0                                      # isolated nodes: amplification 0
    dw = torch.empty(3 * F, K, device=DEV, dtype=torch.float32)
    ws = torch.empty(L.load().tg_gemm_tn_workspace_floats(N, 3 * F, K), device=DEV, dtype=torch.float32)
    L.call("tg_gemm_tn_scaled_bf16", L.ptr(g), L.ptr(agg), L.ptr(scales), L.ptr(dw), L.ptr(ws), N, F, K, F, K, 0, L.stream())
    ref = torch.cat([_ref_tn(g, agg), _ref_tn(g, agg, scales[:, 0]), _ref_tn(g, agg, scales[:, 1])], 0)
    assert (dw - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()
