"""GPU: the whole wrapper (stype encoders -> TABGNNFused -> ClassifierHead, utils.py:353-362) through the C ABI
against the oracle on identical synthetic batches; bf16 tolerance; size-independent properties at the bench size."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _setup(B, hidden=32, layers=1, nhead=8, dtype=torch.float32, seed=0, dropout=0.0):
    import tabgnn_amd as T
    from tabgnn_amd import synthetic as S
    torch.manual_seed(seed)
    cfg = S.make_config(hidden, layers, nhead, B, backbone_dropout=dropout, head_dropout=dropout, compute_dtype=dtype)
    model = T.TABGNNFusedS(cfg)
    with torch.no_grad():                      # non-trivial BatchNorm statistics / LN gains
        for k, v in model.state_dict().items():
            if "running_var" in k:
                v.uniform_(0.5, 1.5)
            elif "running_mean" in k:
                v.normal_(0, 0.2)
    batch = S.make_batch(B, seed=seed + 5)
    return T, cfg, model, batch


def _oracle_logits(model, cfg, batch, training):
    from oracle.step import wrapper_forward
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    node_tf, ei, edge_tf, y = batch
    nf = {k.value: v for k, v in node_tf.feat_dict.items()}
    ef = {k.value: v for k, v in edge_tf.feat_dict.items()}
    return sd, wrapper_forward(sd, cfg["nhead"], cfg["batch_size"], nf, ei, ef, training=training)


@pytest.mark.parametrize("B,hidden,layers,nhead", [(64, 32, 1, 8), (48, 128, 2, 4)])
def test_wrapper_eval_logits_within_1e4_of_oracle(B, hidden, layers, nhead):
    T, cfg, model, batch = _setup(B, hidden, layers, nhead)
    model.eval()
    with torch.no_grad():
        _, want = _oracle_logits(model, cfg, batch, training=False)
        model.to(DEV)
        got = model(batch[0].to(DEV), batch[1].to(DEV), batch[2].to(DEV))
    assert got.dtype == torch.float32 and got.shape == (B, 2)
    np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), rtol=1e-4, atol=1e-4)     # north_star tolerance


def test_train_step_gradients_match_oracle():
    from oracle.step import trainable_keys, weighted_ce
    T, cfg, model, batch = _setup(64, 32, 1, 8, seed=3)
    model.train()
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    # oracle gradients (dropout 0): the running statistics it updates are compared too
    keys = trainable_keys(sd)
    for k in keys:
        sd[k].requires_grad_(True)
    from oracle.step import wrapper_forward
    node_tf, ei, edge_tf, y = batch
    nf = {k.value: v for k, v in node_tf.feat_dict.items()}
    ef = {k.value: v for k, v in edge_tf.feat_dict.items()}
    lw = torch.tensor(cfg["loss_weights"])
    logits = wrapper_forward(sd, cfg["nhead"], 64, nf, ei, ef, training=True)
    loss = weighted_ce(logits[:64], y.view(-1), lw)
    loss.backward()
    model.to(DEV)
    flat = T.FlatParams(model)
    flat.zero_grad()
    out = model(node_tf.to(DEV), ei.to(DEV), edge_tf.to(DEV))
    dl = T.ops.weighted_cross_entropy(out[:64], y.to(DEV), lw.to(DEV))
    dl.backward()
    np.testing.assert_allclose(dl.item(), loss.item(), rtol=1e-5)
    np.testing.assert_allclose(out.detach().cpu().numpy(), logits.detach().numpy(), rtol=1e-4, atol=1e-4)
    for k, p in model.named_parameters():
        want = sd[k].grad if sd[k].grad is not None else torch.zeros_like(sd[k])
        got = p.grad.cpu()
        err = (got - want).abs().max().item()
        assert err <= 1e-3 * (want.abs().max().item() + 1e-6) + 1e-7, (k, err, want.abs().max().item())
    for k, v in model.state_dict().items():
        if "running" in k:
            np.testing.assert_allclose(v.cpu().numpy(), sd[k].detach().numpy(), rtol=1e-4, atol=1e-5, err_msg=k)


def test_fused_adam_training_reduces_loss_and_matches_oracle_trajectory():
    """Three full train steps (FlatParams + FusedAdam) track the oracle's train_step losses (dropout 0)."""
    from oracle import step as ostep
    T, cfg, model, batch = _setup(64, 32, 1, 8, seed=11)
    model.train()
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    node_tf, ei, edge_tf, y = batch
    nf = {k.value: v for k, v in node_tf.feat_dict.items()}
    ef = {k.value: v for k, v in edge_tf.feat_dict.items()}
    lw = torch.tensor(cfg["loss_weights"])
    opt_state, want = {}, []
    for _ in range(3):
        l, _ = ostep.train_step(sd, opt_state, cfg["nhead"], 64, nf, ei, ef, y, lw, cfg["lr"])
        want.append(l)
    model.to(DEV)
    flat = T.FlatParams(model)
    opt = T.FusedAdam(flat, lr=cfg["lr"])
    dbatch = (node_tf.to(DEV), ei.to(DEV), edge_tf.to(DEV), y.to(DEV))
    got = [T.train_step(model, flat, opt, dbatch, lw.to(DEV))[0].item() for _ in range(3)]
    np.testing.assert_allclose(got, want, rtol=2e-3)
    assert got[2] < got[0]


def test_bf16_path_within_stated_tolerance():
    """bf16 activations / fp32 accumulation and master weights: |logit - fp32 oracle| <= 0.06 (stated bf16
    tolerance; the 1e-4 bar applies to the fp32 path)."""
    T, cfg, model, batch = _setup(64, 128, 2, 4, dtype=torch.bfloat16, seed=2)
    model.eval()
    with torch.no_grad():
        _, want = _oracle_logits(model, cfg, batch, training=False)
        model.to(DEV)
        flat = T.FlatParams(model, shadow_dtype=torch.bfloat16)
        got = model(batch[0].to(DEV), batch[1].to(DEV), batch[2].to(DEV))
    err = (got.cpu() - want).abs().max().item()
    assert err <= 0.06, err


def test_dropout_is_counter_based_and_recomputed_in_backward():
    from tabgnn_amd import ops
    x = torch.randn(4096, 128, device=DEV).requires_grad_(True)
    ops.DropoutRNG.new_step(1234)
    y1 = ops.act_dropout(x, "none", 0.5)
    y1.sum().backward()
    g1 = x.grad.clone()
    ops.DropoutRNG.new_step(1234)
    y2 = ops.act_dropout(x, "none", 0.5)
    assert torch.equal(y1, y2)                                   # same (seed, site) -> same mask
    keep = (y1 != 0).float().mean().item()
    assert abs(keep - 0.5) < 0.01
    assert torch.equal(g1 != 0, y1.detach() != 0)                # backward used the same mask
    ops.DropoutRNG.new_step(99)
    assert not torch.equal(ops.act_dropout(x, "none", 0.5), y1)


def test_full_size_properties_of_the_aggregation_path():
    """BASELINE size (B=8192: E~438k, N~524k): sortedness/stability of the CSR, aggregation of constant and
    of integer messages (exact), empty destinations, linearity of the gather backward."""
    from tabgnn_amd import ops
    from tabgnn_amd import synthetic as S
    ei_np, N = S.sampled_subgraph(8192, seed=77)
    ei = torch.from_numpy(ei_np).to(DEV)
    E = ei.shape[1]
    g = ops.SubgraphIndex.build(ei, N, check=True)
    rowptr, perm = g.by_dst
    dst_sorted = ei[1][perm.long()]
    assert bool((dst_sorted[1:] >= dst_sorted[:-1]).all())                        # sorted by destination
    same = dst_sorted[1:] == dst_sorted[:-1]
    assert bool((perm[1:][same] > perm[:-1][same]).all())                         # stable inside a segment
    deg = torch.bincount(ei[1], minlength=N)
    assert torch.equal((rowptr[1:] - rowptr[:-1]).long(), deg)
    F = 128
    for dtype in (torch.bfloat16, torch.float32):
        h = torch.full((E, F), 3.0, device=DEV, dtype=dtype)
        agg = ops.pna_aggregate(h, g).float()
        has = deg > 0
        assert bool((agg[has][:, :3 * F] == 3.0).all()) and bool((agg[has][:, 3 * F:] == 0).all())
        assert bool((agg[~has] == 0).all())
        hi = torch.randint(-4, 5, (E, F), device=DEV).to(dtype)
        agg = ops.pna_aggregate(hi, g).float()
        ref_max = torch.zeros(N, F, device=DEV).scatter_reduce(0, ei[1].view(-1, 1).expand(E, F), hi.float(),
                                                                reduce="amax", include_self=False)
        assert torch.equal(agg[:, F:2 * F], ref_max)                              # max is exact
        ref_sum = torch.zeros(N, F, device=DEV).index_add_(0, ei[1], hi.float())
        got_sum = agg[:, :F] * deg.clamp(min=1).view(-1, 1)
        tol = 0.0 if dtype == torch.float32 else 0.51                            # bf16 rounding of the stored mean
        assert float((got_sum - ref_sum).abs().max()) <= tol * float(deg.max()) + 1e-3
    x = torch.zeros(N, F, device=DEV, requires_grad=True)
    e = torch.zeros(E, F, device=DEV, requires_grad=True)
    out = ops.edge_gather(x, e, g, "src")
    g1 = torch.randint(-3, 4, out.shape, device=DEV).float()
    g2 = torch.randint(-3, 4, out.shape, device=DEV).float()
    (d1,) = torch.autograd.grad(out, x, g1, retain_graph=True)
    (d2,) = torch.autograd.grad(out, x, g2, retain_graph=True)
    (d12,) = torch.autograd.grad(out, x, g1 + g2)
    assert torch.equal(d1 + d2, d12)                                              # linearity, exact on integers


def test_bf16_eval_forward_is_deterministic_and_leaves_inputs_untouched():
    """Inference (`main.py:104-155`: eval mode under no_grad) in bf16: two calls agree bit for bit, nothing in the
    batch is modified (the in-place seed pooling works on an internal tensor), logits are finite."""
    import copy
    import tabgnn_amd as T
    from tabgnn_amd import synthetic as S
    cfg = S.make_config(128, 2, 4, 256, compute_dtype=torch.bfloat16)
    torch.manual_seed(0)
    model = T.TABGNNFusedS(cfg).to(DEV)
    batch = S.make_batch(256, seed=3, device=DEV)
    keep = copy.deepcopy([batch[0].feat_dict, batch[1], batch[2].feat_dict])
    model.train()
    T.ops.DropoutRNG.new_step(1)
    model(batch[0], batch[1], batch[2]).float().sum().backward()          # one training pass first (BN statistics)
    model.eval()
    with torch.no_grad():
        a = model(batch[0], batch[1], batch[2])
        b = model(batch[0], batch[1], batch[2])
    assert torch.isfinite(a.float()).all() and torch.equal(a, b) and a.shape == (256, 2)
    assert all(torch.equal(v, keep[0][k]) for k, v in batch[0].feat_dict.items())
    assert torch.equal(batch[1], keep[1]) and all(torch.equal(v, keep[2][k]) for k, v in batch[2].feat_dict.items())


def test_transposed_shadows_track_the_optimiser():
    """FlatParams keeps W^T bf16 copies for the input-gradient GEMMs: after every FusedAdam step they must equal the
    transposes of the (just refreshed) bf16 shadows, for every 2-D parameter."""
    import tabgnn_amd as T
    from tabgnn_amd import synthetic as S
    cfg = S.make_config(128, 1, 4, 128, compute_dtype=torch.bfloat16)
    torch.manual_seed(0)
    model = T.TABGNNFusedS(cfg).to(DEV).train()
    flat = T.FlatParams(model, shadow_dtype=torch.bfloat16)
    opt = T.FusedAdam(flat, lr=1e-2)
    lw = torch.tensor(cfg["loss_weights"], device=DEV)
    batch = S.make_batch(128, seed=1, device=DEV)

    def check():
        n = 0
        for p in model.parameters():
            assert torch.equal(p._lp.float(), p.detach().to(torch.bfloat16).float())
            if p.dim() == 2:
                assert torch.equal(p._lp_t, p._lp.t()); n += 1
        assert n > 20
    check()
    for _ in range(2):
        T.train_step(model, flat, opt, batch, lw)
        check()


@pytest.mark.parametrize("name", ["gin", "pna", "cpna"])
def test_gnn_wrapper_routes_match_oracle(name):
    """``GNN(config)`` (utils.py:111-233) for --model gin / pna / cpna: fp32 logits within 1e-4 of the oracle's
    composition on the same synthetic batch, then one bf16 d=128 training step (MFMA paths) with finite gradients."""
    import tabgnn_amd as T
    from tabgnn_amd import synthetic as S
    from oracle.step import gnn_wrapper_forward
    B = 48
    torch.manual_seed(3)
    cfg = S.make_config(32, 2, 4, B, head_dropout=0.0)
    cfg.update(model=name, emlps=True)
    model = T.GNN(cfg)
    batch = S.make_batch(B, seed=11)
    node_tf, ei, edge_tf, y = batch
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    nf = {k.value: v for k, v in node_tf.feat_dict.items()}
    ef = {k.value: v for k, v in edge_tf.feat_dict.items()}
    with torch.no_grad():
        ref = gnn_wrapper_forward(sd, name, B, nf, ei, ef)
    model.to(DEV).eval()
    with torch.no_grad():
        got = model(node_tf.to(DEV), ei.to(DEV), edge_tf.to(DEV))
    assert tuple(got.shape) == (B, 2)
    np.testing.assert_allclose(got.float().cpu().numpy(), ref.numpy(), atol=1e-4)
    # bf16, d=128
    cfg = S.make_config(128, 2, 4, B, head_dropout=0.0, compute_dtype=torch.bfloat16)
    cfg.update(model=name, emlps=True)
    model = T.GNN(cfg).to(DEV).train()
    flat = T.FlatParams(model, shadow_dtype=torch.bfloat16)
    opt = T.FusedAdam(flat, lr=1e-3)
    w = torch.tensor([1.0, 9.23], device=DEV)
    b = (node_tf.to(DEV), ei.to(DEV), edge_tf.to(DEV), y.to(DEV))
    l0, _ = T.train_step(model, flat, opt, b, w)
    for _ in range(5):
        l1, _ = T.train_step(model, flat, opt, b, w)
    assert np.isfinite(float(l0)) and np.isfinite(float(l1)) and float(l1) < float(l0)


# Tolerance of bf16 gradients against the fp32 oracle: tests/test_gpu_bf16_parity.py states it and where the rounding
# noise enters (per parameter ||g - g_ref|| <= 0.17 ||g_ref|| + 2e-3 max_k ||g_ref_k||; measured worst 0.137 on the
# first PNA layer's post-projection weight, whose gradient norm is 24 % of the model's largest — not a small tensor —,
# median 0.04); the fp32 twin below pins every term of the same step to 1e-3.
BF16_GRAD_REL_FRO_MEDIAN = 0.04
BF16_LOGIT_ABS = 0.06


def test_bf16_train_step_every_gradient_against_fp32_oracle():
    """The BENCHED path (bf16, C=128, H=4, L=2, through FlatParams): one full training step at B=1024 — large enough
    for the wide-QKV form, the GEMM+LayerNorm kernels, the scaled post projection, the fused tail-LayerNorm backward,
    in-place weight-gradient accumulation and the hub pass of the segmented sums — dropout 0, against oracle/step.py
    in fp32: logits, loss and EVERY parameter gradient (relative Frobenius norm), then a 3-step loss trajectory."""
    from oracle import step as ostep
    B = 1024
    T, cfg, model, batch = _setup(B, 128, 2, 4, dtype=torch.bfloat16, seed=21)
    model.train()
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    node_tf, ei, edge_tf, y = batch
    assert torch.bincount(ei[0]).max() > 256                   # a hub source: the segmented sums' hub pass runs
    nf = {k.value: v for k, v in node_tf.feat_dict.items()}
    ef = {k.value: v for k, v in edge_tf.feat_dict.items()}
    lw = torch.tensor(cfg["loss_weights"])
    torch.set_num_threads(max(1, min(16, len(__import__("os").sched_getaffinity(0)))))
    keys = ostep.trainable_keys(sd)
    for k in keys:
        sd[k].requires_grad_(True)
    logits = ostep.wrapper_forward(sd, cfg["nhead"], B, nf, ei, ef, training=True)
    loss = ostep.weighted_ce(logits[:B], y.view(-1), lw)
    loss.backward()
    want = {k: (sd[k].grad.clone() if sd[k].grad is not None else torch.zeros_like(sd[k])) for k in keys}
    model.to(DEV)
    flat = T.FlatParams(model, shadow_dtype=torch.bfloat16)
    flat.zero_grad()
    dbatch = (node_tf.to(DEV), ei.to(DEV), edge_tf.to(DEV), y.to(DEV))
    out = model(*dbatch[:3])
    dl = T.ops.weighted_cross_entropy(out[:B], dbatch[3], lw.to(DEV))
    dl.backward()
    assert (out.detach().float().cpu() - logits.detach()).abs().max().item() <= BF16_LOGIT_ABS
    assert abs(dl.item() - loss.item()) <= 2e-2 * abs(loss.item())
    from test_gpu_bf16_parity import compare_gradients
    rows = compare_gradients(model, want, flat, 0.17, 2e-3, min_tensors=90, label="configs[1] H=4 bf16")
    rels = sorted(r[1] for r in rows if r[2] >= 1e-3)
    assert rels[len(rels) // 2] <= BF16_GRAD_REL_FRO_MEDIAN, rels[len(rels) // 2]
    # 3-step trajectory: the oracle's Adam against FusedAdam on the flat buffer
    sd2 = {k: v.detach().cpu().clone() for k, v in T.TABGNNFusedS(cfg).state_dict().items()}
    for k in sd2:
        sd2[k] = sd[k].detach().clone()
    opt_state, traj = {}, []
    for _ in range(3):
        l, _ = ostep.train_step(sd2, opt_state, cfg["nhead"], B, nf, ei, ef, y, lw, cfg["lr"])
        traj.append(l)
    opt = T.FusedAdam(flat, lr=cfg["lr"])
    got = [T.train_step(model, flat, opt, dbatch, lw.to(DEV))[0].item() for _ in range(3)]
    np.testing.assert_allclose(got, traj, rtol=3e-2)


def test_fp32_train_step_at_hub_size_every_gradient_within_1e3_of_oracle():
    """The fp32 path of the same kernels at B=1024 (hub sources, multi-block reductions, the 55 k-edge CSR): logits within
    1e-4, every parameter gradient within 1e-3 (relative Frobenius) of oracle/step.py — what pins the kernels' logic at
    the size where the bf16 test above can only state a rounding tolerance."""
    from oracle import step as ostep
    B = 1024
    T, cfg, model, batch = _setup(B, 128, 2, 4, dtype=torch.float32, seed=21)
    model.train()
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    node_tf, ei, edge_tf, y = batch
    nf = {k.value: v for k, v in node_tf.feat_dict.items()}
    ef = {k.value: v for k, v in edge_tf.feat_dict.items()}
    lw = torch.tensor(cfg["loss_weights"])
    torch.set_num_threads(max(1, min(16, len(__import__("os").sched_getaffinity(0)))))
    keys = ostep.trainable_keys(sd)
    for k in keys:
        sd[k].requires_grad_(True)
    logits = ostep.wrapper_forward(sd, cfg["nhead"], B, nf, ei, ef, training=True)
    loss = ostep.weighted_ce(logits[:B], y.view(-1), lw)
    loss.backward()
    model.to(DEV)
    flat = T.FlatParams(model)
    flat.zero_grad()
    out = model(node_tf.to(DEV), ei.to(DEV), edge_tf.to(DEV))
    T.ops.weighted_cross_entropy(out[:B], y.to(DEV), lw.to(DEV)).backward()
    np.testing.assert_allclose(out.detach().cpu().numpy(), logits.detach().numpy(), rtol=1e-4, atol=1e-4)
    gscale = max(sd[k].grad.double().norm().item() for k in keys if sd[k].grad is not None)
    n = 0
    for k, p in model.named_parameters():
        ref = sd[k].grad if sd[k].grad is not None else torch.zeros_like(sd[k])
        err = (p.grad.detach().cpu().double() - ref.double()).norm().item()
        den = ref.double().norm().item()
        assert err <= 1e-3 * den + 1e-6 * gscale, (k, err, den)
        n += 1
    assert n >= 90


def test_load_state_dict_after_flatparams_refreshes_the_bf16_shadows():
    """A checkpoint restore after FlatParams (main.py:271-274 resume) must not leave the GEMMs on stale bf16 weights."""
    T, cfg, model, batch = _setup(64, 128, 1, 4, dtype=torch.bfloat16, seed=5)
    model.to(DEV).eval()
    flat = T.FlatParams(model, shadow_dtype=torch.bfloat16)
    dbatch = (batch[0].to(DEV), batch[1].to(DEV), batch[2].to(DEV))
    with torch.no_grad():
        a = model(*dbatch)
        other = {k: (v + 0.05 * torch.randn_like(v) if v.is_floating_point() and "running" not in k and "avg_deg" not in k
                     and not k.endswith((".mean", ".std", "min_year", "max_values")) else v)
                 for k, v in model.state_dict().items()}
        model.load_state_dict(other)
        b = model(*dbatch)
        _, _, fresh, _ = _setup(64, 128, 1, 4, dtype=torch.bfloat16, seed=5)      # its own encoder modules
        fresh.load_state_dict({k: v.cpu() for k, v in other.items()})
        fresh.to(DEV).eval()
        T.FlatParams(fresh, shadow_dtype=torch.bfloat16)
        c = fresh(*dbatch)
    assert not torch.equal(a, b)
    assert torch.equal(b, c)
    for p in model.parameters():
        assert p.data_ptr() >= flat.flat.data_ptr() and p.data_ptr() < flat.flat.data_ptr() + 4 * flat.flat.numel()
        assert torch.equal(p._lp.float(), p.detach().to(torch.bfloat16).float())


def test_out_of_range_node_ids_raise_after_the_step():
    """The reference raises IndexError at ``x_gnn[src]`` (fused.py:252); here ids are clamped so no kernel faults and
    the flag reaches the host without a per-call synchronisation: the next step (or IndexGuard.check(wait=True)) raises."""
    T, cfg, model, batch = _setup(64, 32, 1, 8, seed=9)
    model.to(DEV).train()
    flat = T.FlatParams(model)
    opt = T.FusedAdam(flat, lr=1e-3)
    lw = torch.tensor(cfg["loss_weights"], device=DEV)
    node_tf, ei, edge_tf, y = (batch[0].to(DEV), batch[1].to(DEV), batch[2].to(DEV), batch[3].to(DEV))
    T.IndexGuard.check(wait=True)
    T.train_step(model, flat, opt, (node_tf, ei, edge_tf, y), lw)
    T.IndexGuard.check(wait=True)                               # clean batch: nothing raised
    bad = ei.clone()
    bad[0, 100] = node_tf.num_rows + 7
    T.train_step(model, flat, opt, (node_tf, bad, edge_tf, y), lw)
    with pytest.raises(RuntimeError, match="outside"):
        T.IndexGuard.check(wait=True)
    T.IndexGuard.check(wait=True)                               # the flag was consumed


def test_gradient_sinks_give_the_gradients_of_plain_autograd_accumulation():
    """ops.GradSink (every consumer of x / edge_attr adds its part into one shared buffer in its own backward kernel)
    against autograd's pairwise accumulation of per-consumer gradients: same forward bit for bit, gradients equal up to
    the rounding of the bf16 partial sums (the sink adds in fp32 before rounding once)."""
    from tabgnn_amd import ops
    T, cfg, model, batch = _setup(96, 128, 2, 4, dtype=torch.bfloat16, seed=11)
    model.to(DEV).train()
    flat = T.FlatParams(model, shadow_dtype=torch.bfloat16)
    lw = torch.tensor(cfg["loss_weights"], device=DEV)
    dbatch = (batch[0].to(DEV), batch[1].to(DEV), batch[2].to(DEV))
    y = batch[3].to(DEV)
    res = []
    assert ops.GRAD_SINKS
    try:
        for on in (True, False):
            ops.GRAD_SINKS = on
            ops.DropoutRNG.new_step(1234)
            flat.zero_grad()
            out = model(*dbatch)
            T.ops.weighted_cross_entropy(out[:96], y.view(-1), lw).backward()
            res.append((out.detach().clone(), flat.grad.clone()))
    finally:
        ops.GRAD_SINKS = True
    assert torch.equal(res[0][0], res[1][0])
    a, b = res[0][1], res[1][1]
    assert torch.isfinite(a).all()
    rel = ((a - b).norm() / b.norm()).item()
    assert rel < 2e-2, rel
    off = 0                                            # per parameter: none lost a contribution
    for name, p in model.named_parameters():
        n = p.numel()
        sa, sb = a[off:off + n], b[off:off + n]
        # structurally zero gradients (biases in front of BatchNorm) are rounding noise in both runs: absolute floor
        assert (sa - sb).norm().item() <= 0.08 * sb.norm().item() + 1e-3 * b.norm().item(), name
        off += n


def test_inplace_seed_pool_with_row_restore_equals_pooling_into_a_copy():
    """The fused layer pools the seed-endpoint rows of x in place (fused.py:268) although the gather-fused edge update
    re-reads x in its backward: the rows are stashed and put back first (ops._MLPReluGather).  Against pooling into a
    copy (TABGNN_NO_POOL_RESTORE): same forward, and — the kernels and their inputs being the same — the same gradients
    bit for bit, over three layers so that two of them back-propagate through their edge update."""
    from tabgnn_amd import ops
    T, cfg, model, batch = _setup(80, 128, 3, 4, dtype=torch.bfloat16, seed=21)
    model.to(DEV).train()
    flat = T.FlatParams(model, shadow_dtype=torch.bfloat16)
    lw = torch.tensor(cfg["loss_weights"], device=DEV)
    dbatch = (batch[0].to(DEV), batch[1].to(DEV), batch[2].to(DEV))
    y = batch[3].to(DEV)
    res = []
    assert ops.POOL_RESTORE
    try:
        for on in (True, False):
            ops.POOL_RESTORE = on
            ops.DropoutRNG.new_step(77)
            flat.zero_grad()
            out = model(*dbatch)
            T.ops.weighted_cross_entropy(out[:80], y.view(-1), lw).backward()
            res.append((out.detach().clone(), flat.grad.clone()))
    finally:
        ops.POOL_RESTORE = True
    assert torch.equal(res[0][0], res[1][0])
    assert torch.isfinite(res[0][1]).all() and res[0][1].abs().max() > 0
    assert torch.equal(res[0][1], res[1][1])
