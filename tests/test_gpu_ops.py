"""GPU parity of the individual HIP operators (through the C ABI) against the oracle / plain torch fp32."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def T():
    import tabgnn_amd
    return tabgnn_amd


DEV = "cuda:0"


def close(got, want, tol, msg=""):
    """Scale-aware comparison for reductions: max |got - want| <= tol * max |want| (sums of many terms cancel)."""
    got = got.detach().cpu().double() if torch.is_tensor(got) else torch.as_tensor(got).double()
    want = want.detach().cpu().double() if torch.is_tensor(want) else torch.as_tensor(want).double()
    err = (got - want).abs().max().item()
    scale = want.abs().max().item() + 1e-12
    assert err <= tol * scale, f"{msg}: max err {err:.3e} vs scale {scale:.3e} (tol {tol})"


def test_device_is_gfx950(T):
    from tabgnn_amd import _lib
    _lib.call("tg_device_check")


@pytest.mark.parametrize("M,N", [(0, 5), (1, 1), (1000, 37), (20000, 5000), (300000, 1000)])
def test_csr_build_is_stable_counting_sort(T, M, N):
    g = torch.Generator().manual_seed(M + N)
    keys = torch.randint(0, N, (M,), generator=g)
    if M > 100:
        keys[: M // 3] = keys[0]                                   # one hub
    k32 = keys.to(torch.int32).to(DEV)
    rowptr, perm = T.ops.SubgraphIndex.csr(k32, N)
    ref_perm = np.argsort(keys.numpy(), kind="stable")
    ref_ptr = np.concatenate([[0], np.cumsum(np.bincount(keys.numpy(), minlength=N))])
    assert np.array_equal(rowptr.cpu().numpy(), ref_ptr)
    assert np.array_equal(perm.cpu().numpy()[:M], ref_perm)


def test_ids_out_of_range_are_flagged_not_faulted(T):
    ids = torch.tensor([[0, 5, 9], [1, 2, -3]], dtype=torch.int64, device=DEV)
    out, err = T.ops.SubgraphIndex.ids32(ids, 6)
    assert int(err.item()) == 1 and out.max().item() <= 5 and out.min().item() >= 0
    with pytest.raises(RuntimeError):
        T.ops.SubgraphIndex.build(ids, 6, check=True)


@pytest.mark.parametrize("S,C,H", [(6, 32, 8), (6, 128, 4), (6, 128, 8), (8, 128, 8), (2, 32, 8), (33, 64, 4)])
def test_attention_core_matches_torch(T, S, C, H):
    torch.manual_seed(S * C + H)
    R = 37
    qkv = torch.randn(R, S, 3 * C)
    go = torch.randn(R, S, C)
    d = C // H
    ref_in = qkv.clone().requires_grad_(True)
    q, k, v = ref_in.split(C, dim=-1)
    sh = lambda t: t.reshape(R, S, H, d).transpose(1, 2)
    p = torch.softmax(sh(q) @ sh(k).transpose(-1, -2) / math.sqrt(d), dim=-1)
    ref = (p @ sh(v)).transpose(1, 2).reshape(R, S, C)
    ref.backward(go)
    x = qkv.to(DEV).requires_grad_(True)
    out = T.ops.attention_core(x, H, 0.0)
    out.backward(go.to(DEV))
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(x.grad.cpu().numpy(), ref_in.grad.numpy(), rtol=2e-4, atol=2e-5)


@pytest.mark.parametrize("C,rows", [(32, 50), (128, 333), (96, 17), (384, 40), (768, 9)])
def test_layer_norm_fused_matches_torch(T, C, rows):
    torch.manual_seed(C)
    a, b, res, go = (torch.randn(rows, C) for _ in range(4))
    bias, gamma, beta = torch.randn(C) * 0.1, 1 + 0.1 * torch.randn(C), 0.1 * torch.randn(C)
    leaf = [t.clone().requires_grad_(True) for t in (a, b, bias, gamma, beta, res)]
    ref = 0.5 * leaf[5] + 0.25 * torch.nn.functional.layer_norm(leaf[0] + leaf[1] + leaf[2], (C,), leaf[3], leaf[4])
    ref.backward(go)
    dl = [t.clone().to(DEV).requires_grad_(True) for t in (a, b, bias, gamma, beta, res)]
    out = T.ops.layer_norm(dl[0], dl[3], dl[4], b=dl[1], bias_b=dl[2], res=dl[5], alpha=0.5, beta_c=0.25)
    out.backward(go.to(DEV))
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), rtol=1e-5, atol=1e-5)
    for got, want, name in zip(dl, leaf, "a b bias gamma beta res".split()):
        np.testing.assert_allclose(got.grad.cpu().numpy(), want.grad.numpy(), rtol=2e-4, atol=2e-5, err_msg=name)


@pytest.mark.parametrize("training", [True, False])
def test_batch_norm_relu_residual_matches_torch(T, training):
    torch.manual_seed(3)
    N, F = 515, 32
    x, res, go = torch.randn(N, F) * 2 + 0.5, torch.randn(N, F), torch.randn(N, F)
    bn = torch.nn.BatchNorm1d(F)
    with torch.no_grad():
        bn.weight.normal_(1, 0.2); bn.bias.normal_(0, 0.2); bn.running_mean.normal_(0, 0.3); bn.running_var.uniform_(0.5, 2)
    mine = T.BatchNorm(F)
    mine.module.load_state_dict(bn.state_dict())
    mine.to(DEV).train(training); bn.train(training)
    xr, rr = x.clone().requires_grad_(True), res.clone().requires_grad_(True)
    ref = (rr + torch.relu(bn(xr))) / 2
    ref.backward(go)
    xd, rd = x.to(DEV).requires_grad_(True), res.to(DEV).requires_grad_(True)
    out = mine(xd, res=rd, relu=True, alpha=0.5, beta_c=0.5)
    out.backward(go.to(DEV))
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(xd.grad.cpu().numpy(), xr.grad.numpy(), rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(rd.grad.cpu().numpy(), rr.grad.numpy(), rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(mine.module.weight.grad.cpu().numpy(), bn.weight.grad.numpy(), rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(mine.module.bias.grad.cpu().numpy(), bn.bias.grad.numpy(), rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(mine.module.running_mean.cpu().numpy(), bn.running_mean.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(mine.module.running_var.cpu().numpy(), bn.running_var.numpy(), rtol=1e-5, atol=1e-6)
    assert int(mine.module.num_batches_tracked) == int(bn.num_batches_tracked)


def _graph(N, E, seed, hub=True):
    from detparams import rand_subgraph
    return torch.from_numpy(rand_subgraph(N, E, 8, seed))


@pytest.mark.parametrize("N,E,F", [(50, 200, 32), (300, 900, 128), (64, 64, 8)])
def test_pna_aggregate_matches_oracle(T, N, E, F):
    from oracle.pna import multi_aggregate
    torch.manual_seed(N)
    ei = _graph(N, E, N)
    h = torch.randn(E, F)
    h[5] = h[4]                                   # exact tie on a shared destination
    ei[1, 5] = ei[1, 4]
    go = torch.randn(N, 4 * F)
    hr = h.clone().requires_grad_(True)
    ref, cnt = multi_aggregate(hr, ei[1], N)
    ref.backward(go)
    g = T.ops.SubgraphIndex.build(ei.to(DEV), N)
    hd = h.to(DEV).requires_grad_(True)
    out = T.ops.pna_aggregate(hd, g)
    out.backward(go.to(DEV))
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), rtol=1e-5, atol=1e-5)
    close(hd.grad, hr.grad, 2e-4, 'dh')
    # empty destinations aggregate to exactly zero
    empty = (cnt == 0).nonzero().flatten()
    assert empty.numel() > 0 and float(out.detach()[empty.to(DEV)].abs().max()) == 0.0


@pytest.mark.parametrize("N,E,F,hub", [(50, 200, 32, 0), (1000, 3000, 128, 0), (333, 5000, 128, 700), (64, 64, 8, 0),
                                       (4097, 9000, 64, 2000)])
def test_pna_aggregate_sorted_bf16_staged_path_equals_indexed_path(T, N, E, F, hub):
    """Destination-sorted bf16 messages go through the LDS-staged kernel (blocks of 32 destinations; ranges that do
    not fit the tile — hub destinations — read HBM directly): same rows summed in the same CSR order as the
    perm-indexed kernel, so forward and backward must agree bit for bit; and both agree with the fp32 oracle."""
    from oracle.pna import multi_aggregate
    torch.manual_seed(N + hub)
    ei = _graph(N, E, N + 1)
    if hub:
        ei[1, :hub] = N // 2                         # one destination with `hub` in-edges: overflows the 32 KB tile
    g = T.ops.SubgraphIndex.build(ei.to(DEV), N)
    h = torch.randn(E, F).to(DEV).bfloat16()
    perm = g.by_dst[1].long()
    go = torch.randn(N, 4 * F, device=DEV).bfloat16()
    h1 = h.clone().requires_grad_(True)
    a1 = T.ops.pna_aggregate(h1, g)
    a1.backward(go)
    h2 = h[perm].contiguous().requires_grad_(True)
    a2 = T.ops.pna_aggregate(h2, g, sorted_rows=True)
    a2.backward(go)
    assert torch.equal(a1, a2)
    assert torch.equal(h1.grad[perm], h2.grad)
    ref, _ = multi_aggregate(h.float().cpu(), ei[1], N)
    np.testing.assert_allclose(a2.detach().float().cpu().numpy(), ref.numpy(), rtol=1e-2, atol=1e-2)


def test_pna_conv_matches_oracle(T):
    from oracle.pna import pna_conv
    from detparams import fill_state_dict
    N, E, F = 90, 400, 32
    ei = _graph(N, E, 5)
    deg = torch.bincount(torch.bincount(ei[1], minlength=N))
    conv = T.PNAConv(F, F, ["mean", "max", "min", "std"], ["identity", "amplification", "attenuation"], deg, edge_dim=F)
    fill_state_dict(conv, 11)
    sd = {k: v.clone() for k, v in conv.state_dict().items()}
    x, e = torch.randn(N, F), torch.randn(E, F)
    go = torch.randn(N, F)
    for k in sd:
        if sd[k].is_floating_point() and "avg_deg" not in k:
            sd[k].requires_grad_(True)
    xr, er = x.clone().requires_grad_(True), e.clone().requires_grad_(True)
    ref = pna_conv(xr, ei, er, sd, "")
    ref.backward(go)
    conv.to(DEV)
    xd, ed = x.to(DEV).requires_grad_(True), e.to(DEV).requires_grad_(True)
    out = conv(xd, ei.to(DEV), ed)
    out.backward(go.to(DEV))
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(xd.grad.cpu().numpy(), xr.grad.numpy(), rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(ed.grad.cpu().numpy(), er.grad.numpy(), rtol=1e-3, atol=1e-4)
    for k, p in conv.named_parameters():
        close(p.grad, sd[k].grad, 1e-4, k)


def _frames(T, R, seed, with_nan=True):
    g = torch.Generator().manual_seed(seed)
    st = T.stype
    num = torch.rand(R, 2, generator=g)
    cat = torch.stack([torch.randint(0, 15, (R,), generator=g), torch.randint(0, 7, (R,), generator=g),
                       torch.randint(0, 15, (R,), generator=g)], dim=1)
    if with_nan:
        num[3, 0] = float("nan")
        cat[4, 1] = -1
    ts = torch.stack([torch.randint(2019, 2024, (R,), generator=g), torch.randint(0, 12, (R,), generator=g),
                      torch.randint(0, 31, (R,), generator=g), torch.randint(0, 7, (R,), generator=g),
                      torch.randint(0, 24, (R,), generator=g), torch.randint(0, 60, (R,), generator=g),
                      torch.randint(0, 60, (R,), generator=g)], dim=1).view(R, 1, 7)
    names = {st.numerical: ["Amount Paid", "in_port"], st.categorical: ["Payment Currency", "Payment Format", "Receiving Currency"],
             st.timestamp: ["Timestamp"]}
    stats = {"Amount Paid": dict(mean=0.4, std=0.3), "in_port": dict(mean=0.1, std=1.5),
             "Payment Currency": dict(cardinality=15), "Payment Format": dict(cardinality=7),
             "Receiving Currency": dict(cardinality=15), "Timestamp": dict(min_year=2019)}
    tf = T.TensorFrame({st.numerical: num, st.categorical: cat, st.timestamp: ts}, names)
    return tf, stats, names


@pytest.mark.parametrize("C", [32, 128])
def test_stype_encoder_matches_oracle(T, C):
    from oracle.encoders import stypewise_encode
    from detparams import fill_state_dict
    R = 333
    tf, stats, names = _frames(T, R, C, with_nan=False)
    tf.feat_dict[T.stype.categorical][4, 1] = -1          # missing category -> padding row 0, no gradient
    enc = T.StypeWiseFeatureEncoder(C, stats, names)
    fill_state_dict(enc, 21)
    with torch.no_grad():
        for e in enc.encoder_dict["categorical"].embs:
            e.weight[0].zero_()
    sd = {k: v.clone() for k, v in enc.state_dict().items()}
    for k in sd:
        if k.endswith(("weight", "bias")):
            sd[k].requires_grad_(True)
    feats = {s.value: t for s, t in tf.feat_dict.items()}
    ref = stypewise_encode(feats, sd, "")
    go = torch.randn(R, 6, C)
    ref.backward(go)
    enc.to(DEV)
    out, cols = enc(tf.to(DEV))
    out.backward(go.to(DEV))
    assert cols == names[T.stype.numerical] + names[T.stype.categorical] + names[T.stype.timestamp]
    # gathered embedding rows are bit-exact; affine / timestamp columns within fp tolerance
    assert torch.equal(out[:, 2:5].detach().cpu(), ref[:, 2:5].detach())
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), rtol=1e-5, atol=1e-6)
    for k, p in enc.named_parameters():
        close(p.grad, sd[k].grad, 2e-5, k)


@pytest.mark.parametrize("C,flat,lazy,dtype", [(128, False, False, torch.float32), (128, True, True, torch.bfloat16),
                                                (256, True, False, torch.bfloat16), (32, True, True, torch.float32)])
def test_big_embedding_tables_gradient_in_a_fixed_order(T, C, flat, lazy, dtype):
    """Embedding tables above tg_encode_small_table_rows (64) — configs[4] has 19 of them — reduced by a counting sort of the
    (column, category) pairs + one wave per bucket (tg_embed_grad_sorted) instead of float atomicAdd: equal to torch's
    embedding autograd on the same upstream gradient, with and without FlatParams gradient buffers and row ids, missing
    categories (-1 -> padding row) included, and bit-identical when repeated."""
    st = T.stype
    g = torch.Generator().manual_seed(C)
    Rtab, R = 5000, 3001
    cards = [5, 100, 3000, 70]
    cat = torch.stack([torch.randint(0, c, (Rtab,), generator=g) for c in cards], dim=1)
    cat[7, 2] = -1
    cat[8, 1] = -1
    num = torch.rand(Rtab, 1, generator=g)
    names = {st.numerical: ["n0"], st.categorical: [f"c{i}" for i in range(4)]}
    stats = {"n0": dict(mean=0.4, std=0.3), **{f"c{i}": dict(cardinality=c) for i, c in enumerate(cards)}}
    enc = T.StypeWiseFeatureEncoder(C, stats, names, dtype).to(DEV)
    assert len(enc._plan["big"]) == 3
    if flat:
        fp = T.FlatParams(enc, shadow_dtype=None if dtype == torch.float32 else dtype)
    ids = torch.randperm(Rtab, generator=g)[:R]
    ids[:20] = torch.tensor([7, 8] * 10)                        # the rows with missing categories, several times
    feats = {st.numerical: num.to(DEV), st.categorical: cat.to(DEV)}
    if lazy:
        tf = T.TensorFrame(feats, names, None, ids.to(DEV))
    else:
        tf = T.TensorFrame({k: v[ids.to(DEV)] for k, v in feats.items()}, names)
    go = torch.randn(R, 5, C, generator=g).to(dtype)
    grads = []
    for rep in range(2):
        for p_ in enc.parameters():
            if flat:
                p_.grad.zero_()
            else:
                p_.grad = None
        out, _ = enc(tf)
        assert out.dtype == dtype
        out.backward(go.to(DEV))
        grads.append([e.weight.grad.clone() for e in enc.encoder_dict["categorical"].embs])
    for a, b in zip(*grads):
        assert torch.equal(a, b)                                 # fixed order: repeats bit for bit
    sel = cat[ids]
    for j, c in enumerate(cards):
        w = torch.zeros(c + 1, C, requires_grad=True)
        idx = (sel[:, j] + 1).clamp(min=0)
        y = torch.nn.functional.embedding(idx, w, padding_idx=0)
        y.backward(go[:, 1 + j].float())
        got = grads[0][j].cpu()
        assert got.shape == w.grad.shape
        tol = 1e-5 if dtype == torch.float32 else 1e-5
        assert (got - w.grad).abs().max().item() <= tol * max(1.0, w.grad.abs().max().item()), (j, c)
        assert float(got[0].abs().max()) == 0.0                  # the padding row takes no gradient


def test_stype_encoder_nan_inputs_become_zero(T):
    from oracle.encoders import stypewise_encode
    from detparams import fill_state_dict
    tf, stats, names = _frames(T, 40, 5, with_nan=True)
    enc = T.StypeWiseFeatureEncoder(32, stats, names)
    fill_state_dict(enc, 3)
    sd = {k: v.clone() for k, v in enc.state_dict().items()}
    ref = stypewise_encode({s.value: t for s, t in tf.feat_dict.items()}, sd, "")
    with torch.no_grad():
        out, _ = enc.to(DEV)(tf.to(DEV))
    assert float(out[3, 0].abs().max()) == 0.0                 # NaN amount -> nan_to_num -> 0
    np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=1e-5, atol=1e-6)


def test_weighted_ce_and_adam_match_torch(T):
    torch.manual_seed(0)
    B, K = 200, 2
    lg = torch.randn(B, K)
    y = (torch.rand(B) < 0.2).long()
    w = torch.tensor([1.0, 9.23])
    lr_ = lg.clone().requires_grad_(True)
    ref = torch.nn.functional.cross_entropy(lr_, y, weight=w)
    ref.backward()
    ld = lg.to(DEV).requires_grad_(True)
    loss = T.ops.weighted_cross_entropy(ld, y.to(DEV), w.to(DEV))
    loss.backward()
    np.testing.assert_allclose(loss.item(), ref.item(), rtol=1e-6)
    np.testing.assert_allclose(ld.grad.cpu().numpy(), lr_.grad.numpy(), rtol=1e-5, atol=1e-7)
    # Adam: 3 steps against torch.optim.Adam
    lin = torch.nn.Linear(37, 11)
    ref_lin = torch.nn.Linear(37, 11); ref_lin.load_state_dict(lin.state_dict())
    lin.to(DEV)
    flat = T.FlatParams(lin)
    opt = T.FusedAdam(flat, lr=6.1e-4)
    ropt = torch.optim.Adam(ref_lin.parameters(), lr=6.1e-4)
    for i in range(3):
        xin = torch.randn(5, 37)
        flat.zero_grad(); ropt.zero_grad()
        torch.nn.functional.linear(xin.to(DEV), lin.weight, lin.bias).pow(2).sum().backward()
        ref_lin(xin).pow(2).sum().backward()
        opt.step(); ropt.step()
    np.testing.assert_allclose(lin.weight.detach().cpu().numpy(), ref_lin.weight.detach().numpy(), rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_seed_pool_inplace_equals_out_of_place(T, dtype):
    """fused.py:261-268 updates x_gnn in place; the in-place kernel touches only the seed rows and must agree bit for
    bit (forward and gradients) with the copying kernel; duplicate seed endpoints share one mean."""
    torch.manual_seed(0)
    N, F, C, B = 500, 32, 16, 40
    tei = torch.randint(0, N, (2, B))
    tei[0, :6] = 7; tei[1, 3:9] = 7                                  # a node that is many seeds' endpoint
    seeds = T.ops.SeedIndex(tei.to(DEV), N)
    x0 = torch.randn(N, F).to(DEV).to(dtype)
    xf0 = torch.randn(B, C + 2 * F).to(DEV).to(dtype)
    go = torch.randn(N, F).to(DEV).to(dtype)
    outs = []
    for inplace in (False, True):
        xl, xfl = x0.clone().requires_grad_(True), xf0.clone().requires_grad_(True)
        y = T.ops.seed_pool(xl * 1.0, xfl, seeds, C, inplace=inplace)
        y.backward(go)
        outs.append((y.detach(), xl.grad, xfl.grad))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    touched = torch.zeros(N, dtype=torch.bool); touched[tei.flatten()] = True
    assert torch.equal(outs[1][0][~touched.to(DEV)], x0[~touched.to(DEV)])          # other rows untouched


@pytest.mark.parametrize("C,H,dtype,p", [(128, 4, torch.bfloat16, 0.3), (128, 4, torch.float32, 0.3), (32, 8, torch.float32, 0.0)])
def test_encoder_layer_composite_equals_op_by_op_composition(T, C, H, dtype, p):
    """The one-node encoder layer (hand-scheduled backward; in bf16 at C = 128: MFMA NT GEMMs with fused ReLU+dropout
    and gate epilogues, in-kernel weight-gradient accumulation) against the op-by-op composition of the single
    operators — same dropout streams (same site order), so forward, input gradient and parameter gradients agree."""
    torch.manual_seed(1)
    R, S = 700, 6
    layer = T.ColumnTransformerLayer(C, H, dropout=p).to(DEV).train()
    x0 = torch.randn(R, S, C, device=DEV).to(dtype)
    go = torch.randn(R, S, C, device=DEV).to(dtype)
    res = []
    for fused in (True, False):
        for q in layer.parameters():      # existing gradient buffers on the fused run: the in-kernel "+=" paths are taken
            q.grad = torch.zeros_like(q) if fused else None
        if dtype == torch.bfloat16:
            for q in layer.parameters():
                q._lp = q.detach().to(dtype)
        T.ops.DropoutRNG.new_step(77)
        x = x0.clone().requires_grad_(True)
        y = layer(x) if fused else layer.forward_unfused(x)
        y.backward(go)
        res.append((y.detach().float(), x.grad.float(), {k: q.grad.float().clone() for k, q in layer.named_parameters()}))
    tol = 0.06 if dtype == torch.bfloat16 else 2e-4
    (y1, g1, p1), (y2, g2, p2) = res
    # (the composite is the one-kernel layer when the shapes allow: its bf16 rounding points differ from the op-by-op
    # kernels', so the bound scales with the magnitude: one bf16 ulp at |y| ~ 8 is 0.06)
    assert ((y1 - y2).abs() <= tol * (1.0 + y2.abs())).all() and (g1 - g2).abs().max().item() < tol * max(1.0, g2.abs().max().item())
    for k in p1:      # bf16: sums over R*S rows of rounded products -> compare in the Frobenius norm
        rel = (p1[k] - p2[k]).norm().item() / max(p2[k].norm().item(), 1e-6)
        assert rel < ((0.08 if k.startswith("linear1") else 0.05) if dtype == torch.bfloat16 else 2e-4), (k, rel)


@pytest.mark.parametrize("F,hubdeg", [(32, 3000), (128, 1500)])
def test_pna_aggregate_hub_destinations(T, F, hubdeg):
    """Destinations with more than 512 in-edges (reverse message passing makes the heavy-tailed sources destinations)
    are reduced by a whole workgroup: forward and backward against the fp32 oracle, and bf16 sorted == indexed."""
    from oracle.pna import multi_aggregate
    torch.manual_seed(F)
    N, E = 400, 6000
    ei = _graph(N, E, 7)
    ei[1, :hubdeg] = 11                                # one hub destination
    ei[1, hubdeg:hubdeg + 600] = 12                    # and one just above the threshold
    h = torch.randn(E, F)
    h[3] = h[2]                                        # a tie on the hub (shared max/min gradient)
    go = torch.randn(N, 4 * F)
    hr = h.clone().requires_grad_(True)
    ref, _ = multi_aggregate(hr, ei[1], N)
    ref.backward(go)
    g = T.ops.SubgraphIndex.build(ei.to(DEV), N)
    hd = h.to(DEV).requires_grad_(True)
    out = T.ops.pna_aggregate(hd, g)
    out.backward(go.to(DEV))
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), rtol=2e-4, atol=2e-4)
    close(hd.grad, hr.grad, 5e-4, 'dh')
    hb = h.to(DEV).bfloat16()
    perm = g.by_dst[1].long()
    gob = go.to(DEV).bfloat16()
    h1 = hb.clone().requires_grad_(True); a1 = T.ops.pna_aggregate(h1, g); a1.backward(gob)
    h2 = hb[perm].contiguous().requires_grad_(True); a2 = T.ops.pna_aggregate(h2, g, sorted_rows=True); a2.backward(gob)
    assert torch.equal(a1, a2) and torch.equal(h1.grad[perm], h2.grad)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("alpha,p", [(1.0, 0.3), (0.0, 0.0)])
def test_tail_and_norm2_backward_in_one_kernel(dtype, alpha, p):
    """tg_ln_tail_ln_bwd against the two tg_ln_bwd launches it replaces (tail LayerNorm, then norm2 in z mode):
    dres, d_x1, d_y2 and the five parameter-gradient sums, written and accumulated."""
    import ctypes
    from tabgnn_amd import _lib as L
    M, C = 4099, 128
    g = torch.Generator().manual_seed(17)
    r = lambda *s: torch.randn(*s, generator=g).to(DEV)
    x2, z2, dout = r(M, C).to(dtype), r(M, C).to(dtype), r(M, C).to(dtype)
    gt, g2 = 1 + 0.1 * r(C), 1 + 0.1 * r(C)
    st = lambda x: torch.stack([x.float().mean(1), (x.float().var(1, unbiased=False) + 1e-5).rsqrt()], 1).contiguous()
    st3, st2 = st(x2), st(z2)
    beta_c, seed, rs = 0.5, 1234, 7
    e = lambda: torch.empty(M, C, dtype=dtype, device=DEV)
    part = lambda n: torch.empty(2048 * n * C, dtype=torch.float32, device=DEV)
    # reference: two launches
    d_x2, dres_r, d_x1_r = e(), e(), e()
    dp_t, dp_2 = torch.empty(3, C, device=DEV), torch.empty(3, C, device=DEV)
    L.call("tg_ln_bwd", L.ptr(x2), None, None, L.ptr(gt), L.ptr(st3), L.ptr(dout), L.ptr(d_x2), None,
           L.ptr(dres_r) if alpha else None, L.ptr(dp_t), L.ptr(part(3)), M, C, alpha, beta_c, 0.0, 0, 0, 0, None, None,
           None, L.dt(x2), L.stream())
    d_y2_r = e()
    L.call("tg_ln_bwd", L.ptr(z2), None, None, L.ptr(g2), L.ptr(st2), L.ptr(d_x2), L.ptr(d_x1_r), L.ptr(d_y2_r), None,
           L.ptr(dp_2), L.ptr(part(3)), M, C, 0.0, 1.0, p, seed, rs, 0, None, None, None, L.dt(x2), L.stream())
    # fused, written
    dres, d_x1, d_y2, dp = e(), e(), e(), torch.empty(5, C, device=DEV)
    L.call("tg_ln_tail_ln_bwd", L.ptr(x2), L.ptr(z2), L.ptr(gt), L.ptr(st3), L.ptr(g2), L.ptr(st2), L.ptr(dout),
           L.ptr(dres) if alpha else None, L.ptr(d_x1), L.ptr(d_y2), L.ptr(dp), L.ptr(part(5)), M, C, alpha, beta_c, p,
           seed, rs, None, L.dt(x2), L.stream())
    tol = 1e-5 if dtype == torch.float32 else 2e-2        # bf16: the unfused path rounds d_x2 to bf16 in between
    rel = lambda a, b: ((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-12)).item()
    if alpha:
        assert torch.equal(dres, dres_r)
    assert rel(d_x1, d_x1_r) < tol and rel(d_y2, d_y2_r) < tol
    assert (d_y2 == 0).float().mean().item() == pytest.approx((d_y2_r == 0).float().mean().item(), abs=1e-3)
    for got, want in ((dp[0], dp_t[0]), (dp[1], dp_t[1]), (dp[2], dp_2[0]), (dp[3], dp_2[1]), (dp[4], dp_2[2])):
        assert rel(got, want) < max(tol, 1e-4)
    # fused, accumulated into existing buffers (one target skipped)
    acc_t = [torch.full((C,), 2.0, device=DEV) for _ in range(5)]
    ptrs = (ctypes.c_void_p * 5)(acc_t[0].data_ptr(), acc_t[1].data_ptr(), acc_t[2].data_ptr(), None, acc_t[4].data_ptr())
    L.call("tg_ln_tail_ln_bwd", L.ptr(x2), L.ptr(z2), L.ptr(gt), L.ptr(st3), L.ptr(g2), L.ptr(st2), L.ptr(dout),
           L.ptr(dres) if alpha else None, L.ptr(d_x1), L.ptr(d_y2), None, L.ptr(part(5)), M, C, alpha, beta_c, p, seed,
           rs, ptrs, L.dt(x2), L.stream())
    for i in (0, 1, 2, 4):
        torch.testing.assert_close(acc_t[i], dp[i] + 2.0, rtol=1e-5, atol=1e-4)
    assert torch.all(acc_t[3] == 2.0)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("R,S,C", [(1, 6, 128), (37, 6, 128), (50000, 6, 128), (4097, 3, 384), (0, 6, 128)])
def test_col_sum_of_the_cls_column_matches_torch_and_repeats(T, dtype, R, S, C):
    """`tg_col_sum`: the shared CLS vector's gradient = column 0 of the [R, S, C] row gradient summed over rows
    (src/nn/models/fused.py:158-159 via autograd).  fp32 accumulation in a fixed order: equal to a float64 sum of the
    same (rounded) inputs within fp32 summation error, and bit-identical run to run."""
    from tabgnn_amd import _lib as L
    from tabgnn_amd.models import _PrependCLS
    torch.manual_seed(R + C)
    g = torch.randn(R, S, C, device="cuda").to(dtype)
    ref = g[:, 0, :].double().sum(0)
    outs = []
    for _ in range(2):
        out = torch.full((C,), 7.0, device="cuda")
        ws = torch.empty(L.load().tg_col_sum_workspace_floats(R, C), dtype=torch.float32, device="cuda")
        L.call("tg_col_sum", L.ptr(g) if R else L.ptr(torch.empty(8, device="cuda", dtype=dtype)), R, C, S * C, L.ptr(out),
               L.ptr(ws), 0, L.dt(g), L.stream())
        outs.append(out)
    assert torch.equal(outs[0], outs[1])
    assert (outs[0].double() - ref).abs().max().item() <= 1e-5 * max(1.0, (R ** 0.5) * 4)
    acc = outs[0].clone()
    L.call("tg_col_sum", L.ptr(g) if R else L.ptr(torch.empty(8, device="cuda", dtype=dtype)), R, C, S * C, L.ptr(acc),
           L.ptr(ws), 1, L.dt(g), L.stream())
    assert torch.allclose(acc, 2 * outs[0], rtol=1e-6, atol=1e-6)
    if R:
        class Ctx:
            cls = None
        dcls, rest = _PrependCLS.backward(Ctx(), g)
        assert torch.equal(dcls, outs[0]) and rest.data_ptr() == g[:, 1:, :].data_ptr()
        par = torch.nn.Parameter(torch.zeros(C, device="cuda"))          # a parameter that owns a gradient buffer: added in place
        par.grad = torch.full((C,), 7.0, device="cuda")
        ctx = Ctx()
        ctx.cls = par
        dcls2, _ = _PrependCLS.backward(ctx, g)
        assert dcls2 is None and torch.allclose(par.grad, outs[0] + 7.0, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_axpby2_is_the_two_axpby_passes_in_one(T, dtype):
    from tabgnn_amd import _lib as L
    torch.manual_seed(0)
    n = 8 * 12345
    a = torch.randn(n, device="cuda").to(dtype)
    b = torch.randn(n, device="cuda").to(dtype)
    for with_a in (True, False):
        y1, y2 = torch.empty_like(b), torch.empty_like(b)
        L.call("tg_axpby2", L.ptr(a) if with_a else None, L.ptr(b), L.ptr(y1), L.ptr(y2), n, 1.0, 0.5, 0.25, L.dt(b), L.stream())
        r1 = torch.empty_like(b)
        L.call("tg_axpby", L.ptr(a if with_a else b), L.ptr(b), L.ptr(r1), n, 1.0 if with_a else 0.0, 0.5, L.dt(b), L.stream())
        assert torch.equal(y1, r1)
        assert torch.equal(y2, (b.float() * 0.25).to(dtype))
    # in place on the first operand, as the gradient-buffer accumulation uses it
    buf = a.clone()
    gb = torch.empty_like(b)
    L.call("tg_axpby2", L.ptr(buf), L.ptr(b), L.ptr(buf), L.ptr(gb), n, 1.0, 0.5, 0.5, L.dt(b), L.stream())
    assert torch.equal(buf, (a.float() + 0.5 * b.float()).to(dtype))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_row_head_scale_matches_the_torch_slicing_it_replaces(T, dtype):
    """`tg_row_head_scale`: the row-wise pieces of the CLS-merge backward (fused.py:259-260) and of the seed gathers'
    backward (fused.py:257, decoder.py:18-19) against the clone / slice-multiply / zeros / slice-copy they replace."""
    from tabgnn_amd import ops
    torch.manual_seed(0)
    B, S, C, F = 37, 6, 128, 128
    D = C + 2 * F
    g = torch.randn(B, S, C, device="cuda").to(dtype)
    # g with token 0 halved
    want = g.clone(); want[:, 0, :] *= 0.5
    got = torch.empty_like(g)
    ops._row_head_scale(g, S * C, S * C, got, B, S * C, C, 0.5, 1.0)
    assert torch.equal(got, want)
    # [g[:, 0] / 2 | 0] as a [B, D] row
    want = torch.zeros(B, D, dtype=dtype, device="cuda"); want[:, :C] = g[:, 0, :] * 0.5
    got = torch.full((B, D), 7.0, dtype=dtype, device="cuda")
    ops._row_head_scale(g, S * C, C, got, B, D, C, 0.5, 0.0)
    assert torch.equal(got, want)
    # the CLS slice of a [B, D] gradient padded to a [B, S, C] row; a tail block made contiguous
    gd = torch.randn(B, D, device="cuda").to(dtype)
    want = torch.zeros(B, S, C, dtype=dtype, device="cuda"); want[:, 0, :] = gd[:, :C]
    got = torch.full((B, S, C), 7.0, dtype=dtype, device="cuda")
    ops._row_head_scale(gd, D, C, got, B, S * C, C, 1.0, 0.0)
    assert torch.equal(got, want)
    got = torch.empty(B, C, dtype=dtype, device="cuda")
    ops._row_head_scale(gd, D, C, got, B, C, C, 1.0, 1.0, offset=2 * F)
    assert torch.equal(got, gd[:, 2 * F:].contiguous())
    # B = 0 is a no-op
    ops._row_head_scale(gd[:0], D, C, got[:0], 0, C, C, 1.0, 1.0)
