"""GPU: the one-kernel column-transformer layer AT THE SIZES THE BENCH RUNS (round 5).

Round 4 found the saved rows z1 / z2 of the benched kernels wrong on every launch at R >= 13 000 (p = 0 instantiations)
while every parity case stayed at or below 5 000 rows and the determinism test — a result wrong the same way every time
repeats bit for bit — could not see it.  These cases compare, once each, at R in {13 000, 60 000, 430 162}, S = 6,
4 and 8 heads, dropout 0 and 0.5:
  * out, z1, z2 of the fused forward with the op-by-op kernels on the same dropout streams,
  * d_x and every parameter gradient of the fused backward chain with the op-by-op backward,
  * a strided sample of table rows (rows are independent) with torch.nn.TransformerEncoderLayer in fp32 autograd:
    out, z1, z2, d_x (p = 0).
Reference module: src/nn/models/fused.py:83-92 (construction), :160-166, :249 (call sites)."""
import pytest
import torch

from test_gpu_encoder_fused import BF16_TOL, _grads, _layer, _relerr

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
SIZES = [13000, 60000, 430162]


def _op_by_op_forward(x, layer, tail, p, seed, alpha, beta_c):
    """(out, z1, z2) through the single-op kernels in the stream order of encoder_layer._EncoderLayerFn.forward."""
    from tabgnn_amd import _lib as L, ops
    from tabgnn_amd.encoder_layer import _ln_fwd
    R, S, C = x.shape
    T = R * S
    sa = layer.self_attn
    bf = lambda t: t.detach().to(torch.bfloat16).contiguous()
    ops.DropoutRNG.new_step(seed)
    sd = ops.DropoutRNG.seed
    rs = [ops.DropoutRNG.next_stream() for _ in range(4)]
    x2d = x.view(T, C)
    qkv = ops.gemm_nt(x2d, bf(sa.in_proj_weight), sa.in_proj_bias.detach())
    o = torch.empty(T, C, dtype=x.dtype, device=x.device)
    lse = torch.empty(R, layer.nhead, S, dtype=torch.float32, device=x.device)
    L.call("tg_attn_fwd", L.ptr(qkv), L.ptr(o), L.ptr(lse), R, S, C, layer.nhead, p, sd, rs[0], L.dt(x), L.stream())
    z1, x1, _ = ops.gemm_nt_ln(o, bf(sa.out_proj.weight), sa.out_proj.bias.detach(), x2d, layer.norm1.weight.detach(),
                               layer.norm1.bias.detach(), p, sd, rs[1])
    h = ops.gemm_nt(x1, bf(layer.linear1.weight), layer.linear1.bias.detach(), ops.NT_RELU | ops.NT_DROPOUT, p, sd, rs[2])
    z2, x2, _ = ops.gemm_nt_ln(h, bf(layer.linear2.weight), layer.linear2.bias.detach(), x1, layer.norm2.weight.detach(),
                               layer.norm2.bias.detach(), p, sd, rs[3])
    out, _ = _ln_fwd(x2, None, None, tail.weight, tail.bias, x2d if alpha != 0.0 else None, alpha, beta_c, 0.0, 0, 0)
    return out.view(R, S, C), z1.view(R, S, C), z2.view(R, S, C)


def _fused_forward(x, layer, tail, p, seed, alpha, beta_c):
    from tabgnn_amd import ops
    import tabgnn_amd.encoder_layer as EL
    sa = layer.self_attn
    bf = lambda t: t.detach().to(torch.bfloat16).contiguous()
    ops.DropoutRNG.new_step(seed)
    sd = ops.DropoutRNG.seed
    rs = [ops.DropoutRNG.next_stream() for _ in range(4)]
    wpack, prm = EL.pack_layer(bf(sa.in_proj_weight), bf(sa.out_proj.weight), bf(layer.linear1.weight), bf(layer.linear2.weight),
                               sa.in_proj_bias, sa.out_proj.bias, layer.norm1.weight, layer.norm1.bias, layer.linear1.bias,
                               layer.linear2.bias, layer.norm2.weight, layer.norm2.bias, tail.weight, tail.bias)
    return EL.fused_forward(x, layer.nhead, p, True, alpha, beta_c, wpack, prm, sd, rs, True)


def _torch_rows(layer, tail, x_rows, co_rows, H, alpha, beta_c):
    """fp32 torch module on a few table rows: out, z1, z2 and d(sum(out * co)) / dx."""
    ref = torch.nn.TransformerEncoderLayer(128, H, 128, 0.0, "relu", batch_first=True).to(x_rows.device)
    ref.load_state_dict(layer.state_dict())
    ref.eval()
    xr = x_rows.float().requires_grad_(True)
    sa_out = ref.self_attn(xr, xr, xr, need_weights=False)[0]
    z1 = xr + sa_out
    x1 = ref.norm1(z1)
    z2 = x1 + ref.linear2(torch.relu(ref.linear1(x1)))
    x2 = ref.norm2(z2)
    y = alpha * xr + beta_c * torch.nn.functional.layer_norm(x2, (128,), tail.weight, tail.bias, 1e-5)
    (y * co_rows).sum().backward()
    return y.detach(), z1.detach(), z2.detach(), xr.grad


@pytest.mark.parametrize("p", [0.0, 0.5])
@pytest.mark.parametrize("H", [4, 8])
@pytest.mark.parametrize("R", SIZES)
def test_forward_out_z1_z2_against_op_by_op_kernels_at_bench_sizes(R, H, p):
    layer, tail = _layer(H, seed=21)
    layer.to(DEV); tail.to(DEV)
    x = (torch.randn(R, 6, 128, device=DEV) * 1.2).to(torch.bfloat16)
    with torch.no_grad():
        got = _fused_forward(x, layer, tail, p, 777, 0.5, 0.5)
        want = _op_by_op_forward(x, layer, tail, p, 777, 0.5, 0.5)
    for name, a, b in zip(("out", "z1", "z2"), got, want):
        d = (a.float() - b.float()).abs()
        # per TOKEN (a wrong 32-token tile or a handful of wrong dwords must not drown in 330 M elements)
        tok = d.view(-1, 128).max(dim=1).values
        scale = b.float().abs().view(-1, 128).max(dim=1).values.clamp_min(1.0)
        worst = (tok / scale).max().item()
        assert worst <= 0.08, (name, worst, int((tok / scale).argmax()))
        assert d.mean().item() <= 0.004, (name, d.mean().item())
        assert _relerr(a.float(), b.float()) <= 0.01, name


@pytest.mark.parametrize("H", [4, 8])
@pytest.mark.parametrize("R", SIZES)
def test_forward_and_dx_rows_against_torch_fp32_at_bench_sizes(R, H):
    """p = 0.  Strided table rows + the last rows (partial tile) + a whole run of 40 consecutive rows (eight wave tiles)."""
    import tabgnn_amd.encoder_layer as EL
    layer, tail = _layer(H, seed=22)
    layer.to(DEV); tail.to(DEV)
    x = (torch.randn(R, 6, 128, device=DEV) * 1.2).to(torch.bfloat16)
    co = torch.randn(R, 6, 128, device=DEV)
    idx = torch.cat([torch.arange(0, R, max(R // 50, 1), device=DEV), torch.arange(R - 7, R, device=DEV),
                     torch.arange(R // 2, R // 2 + 40, device=DEV)]).unique()
    with torch.no_grad():
        out, z1, z2 = _fused_forward(x, layer, tail, 0.0, 5, 0.5, 0.5)
    n0 = dict(EL.STATS)
    _, g = _grads(layer, tail, x, 0.0, True, 0.5, 0.5, co, seed=5)
    assert EL.STATS["fused_bwd"] == n0["fused_bwd"] + 1 and EL.STATS["fused_bwd_attn"] == n0["fused_bwd_attn"] + 1
    y_r, z1_r, z2_r, dx_r = _torch_rows(layer, tail, x[idx], co[idx], H, 0.5, 0.5)
    assert (out[idx].float() - y_r).abs().max().item() <= BF16_TOL
    for name, a, b in (("z1", z1[idx], z1_r), ("z2", z2[idx], z2_r)):
        rel = ((a.float() - b).abs() / b.abs().clamp_min(1.0)).max().item()
        assert rel <= 0.03, (name, rel)                    # bf16 storage of an O(1..4) sum
    assert _relerr(g["x"][idx], dx_r) <= 0.03
    # per row: no single row is off (a wrong tile is 5 rows of 430 k)
    per_row = (g["x"][idx] - dx_r).flatten(1).norm(dim=1) / dx_r.flatten(1).norm(dim=1).clamp_min(1e-6)
    assert per_row.max().item() <= 0.08, per_row.max().item()


@pytest.mark.parametrize("p", [0.0, 0.5])
@pytest.mark.parametrize("H", [4, 8])
@pytest.mark.parametrize("R", SIZES)
def test_every_gradient_against_op_by_op_kernels_at_bench_sizes(R, H, p):
    import tabgnn_amd.encoder_layer as EL
    layer, tail = _layer(H, seed=23)
    layer.to(DEV); tail.to(DEV)
    x = (torch.randn(R, 6, 128, device=DEV) * 1.2).to(torch.bfloat16)
    co = torch.randn(R, 6, 128, device=DEV)
    n0 = dict(EL.STATS)
    EL._FUSED_TRAIN = True
    out_f, g_f = _grads(layer, tail, x, p, True, 0.5, 0.5, co)
    assert EL.STATS["fused_fwd"] == n0["fused_fwd"] + 1 and EL.STATS["fused_bwd"] == n0["fused_bwd"] + 1
    assert EL.STATS["fused_bwd_attn"] == n0["fused_bwd_attn"] + 1
    EL._FUSED_TRAIN = False
    try:
        out_u, g_u = _grads(layer, tail, x, p, True, 0.5, 0.5, co)
    finally:
        EL._FUSED_TRAIN = True
    assert _relerr(out_f, out_u) <= 0.01
    # d_x per table row as well as in norm
    per_row = (g_f["x"] - g_u["x"]).flatten(1).norm(dim=1) / g_u["x"].flatten(1).norm(dim=1).clamp_min(1e-6)
    assert per_row.max().item() <= 0.15, (per_row.max().item(), int(per_row.argmax()))
    worst = sorted(((_relerr(g_f[k], g_u[k]), k) for k in g_u if g_u[k] is not None), reverse=True)
    assert all(r <= (0.06 if k.startswith("linear1") else 0.03) for r, k in worst), worst[:4]
