"""GPU: the bf16 paths that ``bench.py`` times, against the fp32 oracle (oracle/step.py) — every parameter gradient.

VERDICT r02 item 3: (a) BASELINE configs[1] with the reference's default 8 heads (``fused.py:61``), (b) the ``tabgnn``
path at S = 130 column tokens (configs[3], ``src/nn/models/tabgnn.py:127-129``) on >= 2 000 node rows, (c) 64 mixed
columns at C = 256 (configs[4]) with B >= 256 seed edges.  Dropout 0 in both (masks are a separate matter: the
mask-equality tests of test_gpu_encoder_fused.py / test_gpu_wrapper.py).

Every configuration runs TWICE: in fp32 (the same operator graph on the fp32 kernels; gate 2e-3 relative — this is what
pins the LOGIC of every gradient term at these shapes) and in bf16 (the benched precision; the gate below states the
rounding noise measured for it).  Per parameter

    ||g - g_ref||_F  <=  REL * ||g_ref||_F  +  ABS * max_k ||g_ref_k||_F

Measured bf16 noise (round 3; dropout 0, random init): the error grows along the backward chain —
configs[1]: decoder 0.0001 / 0.002 / 0.006 / 0.02, fuse MLP 0.03 -> 0.09, first PNA layer 0.10-0.14 (its post / message
projection weights, whose gradient norms are 10-25 % of the model's largest: NOT small tensors), median 0.04;
configs[3] (S = 130, loss over all 2 100 sampled nodes): 0.005 at the last decoder layer, 0.05-0.08 two layers up,
0.17-0.23 in the column transformer.  Cause: a weight gradient is a sum over rows of (output gradient) x (input
activation); at random init the rows (nodes / edges) carry nearly the same activation, so the sum is the small
covariance of two factors whose common part cancels, while the bf16 rounding of the stored activation (2^-9 of its FULL
magnitude) does not cancel: relative noise ~ 2^-9 * mean / spread, and every LayerNorm / BatchNorm backward (which
subtracts the common part again) passes it on.  The fp32 twins are within 2e-3 on the same shapes, so a wrong or
dropped term cannot hide in these gates.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GATES = {"bf16": dict(c1=(0.17, 2e-3), c3=(0.28, 2e-3), c4=(0.17, 2e-3)), "fp32": dict(c1=(2e-3, 1e-5), c3=(2e-3, 1e-5), c4=(2e-3, 1e-5))}
REL, ABS = 0.17, 2e-3
SPREAD_GATE = 0.15          # bf16 vs fp32 on a state with spread node inputs: measured 0.07-0.115 on the first PNA layer's weights
TWIN_GATE = 0.05            # benched (fused) bf16 kernels vs the plain bf16 composition of single-op kernels, same masks
SUM_GATE = 0.05             # gradient sum over 16 batches, benched bf16 vs fp32 kernels (single batch: 0.07-0.15)
LOGIT_ABS = {"bf16": 0.06, "fp32": 1e-4}
DT = {"bf16": torch.bfloat16, "fp32": torch.float32}


def _feats(tf):
    return {k.value: v for k, v in tf.feat_dict.items()}


def compare_gradients(model, want, flat=None, rel=REL, abs_=ABS, min_tensors=20, label="", got=None):
    """Every parameter gradient of ``model`` (after backward; or the saved copies ``got``) against ``want`` (dict key ->
    reference gradient)."""
    gscale = max(v.double().norm().item() for v in want.values())
    rows = []
    for k, p in model.named_parameters():
        ref = want[k]
        if got is not None:
            g = got[k]
        else:
            g = p.grad.detach().float().cpu() if p.grad is not None else torch.zeros_like(ref)
        if flat is not None and p.grad is not None:
            assert flat.grad.data_ptr() <= p.grad.data_ptr() < flat.grad.data_ptr() + 4 * flat.grad.numel(), k
        den = ref.double().norm().item()
        err = (g.double() - ref.double()).norm().item()
        rows.append((err / max(rel * den + abs_ * gscale, 1e-30), err / max(den, 1e-30), den / gscale, k))
    rows.sort(reverse=True)
    print(f"{label} gradients vs fp32 oracle — worst 6 (gate ratio, rel. Frobenius error, ||g||/max||g||, name):")
    for r in rows[:6]:
        print("   %.3f  %.4f  %.2e  %s" % r)
    big = sorted(r[1] for r in rows if r[2] >= 0.01)
    if big:
        print(f"   tensors with ||g|| >= 1% of the largest: {len(big)}, median rel. error {big[len(big) // 2]:.4f}, worst {big[-1]:.4f}")
    assert len(rows) >= min_tensors
    bad = [r for r in rows if r[0] > 1.0]
    assert not bad, bad[:5]
    return rows


def _oracle_grads(sd, forward):
    from oracle import step as ostep
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    keys = ostep.trainable_keys(sd)
    for k in keys:
        sd[k].requires_grad_(True)
    logits, loss = forward(sd)
    loss.backward()
    want = {k: (sd[k].grad.clone() if sd[k].grad is not None else torch.zeros_like(sd[k])) for k in keys}
    return logits.detach(), float(loss.detach()), want


@pytest.mark.parametrize("dt", ["fp32", "bf16"])
def test_configs1_eight_heads_every_gradient(dt):
    """(a) d = 128, L = 2, **H = 8** (reference default ``nhead``, fused.py:61): the fused column-transformer kernels with
    head dim 16 forward and backward, B = 1024 (wide QKV form, hub pass of the segmented sums, scaled post projection)."""
    import tabgnn_amd as T
    import tabgnn_amd.encoder_layer as EL
    from oracle import step as ostep
    from tabgnn_amd import synthetic as S
    B = 1024
    torch.manual_seed(31)
    cfg = S.make_config(128, 2, 8, B, backbone_dropout=0.0, head_dropout=0.0, compute_dtype=DT[dt])
    model = T.TABGNNFusedS(cfg).train()
    batch = S.make_batch(B, seed=36)
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    node_tf, ei, edge_tf, y = batch
    lw = torch.tensor(cfg["loss_weights"])

    def fwd(sd_):
        lg = ostep.wrapper_forward(sd_, 8, B, _feats(node_tf), ei, _feats(edge_tf), training=True)
        return lg, ostep.weighted_ce(lg[:B], y.view(-1), lw)
    logits, loss, want = _oracle_grads(sd, fwd)
    model.to(DEV)
    flat = T.FlatParams(model, shadow_dtype=DT[dt])
    flat.zero_grad()
    n0 = dict(EL.STATS)
    out = model(node_tf.to(DEV), ei.to(DEV), edge_tf.to(DEV))
    dl = T.ops.weighted_cross_entropy(out[:B], y.to(DEV), lw.to(DEV))
    dl.backward()
    if dt == "bf16":
        assert EL.STATS["fused_fwd"] > n0["fused_fwd"] and EL.STATS["fused_bwd"] > n0["fused_bwd"]      # the benched kernels ran
    assert (out.detach().float().cpu() - logits).abs().max().item() <= LOGIT_ABS[dt]
    assert abs(dl.item() - loss) <= (2e-2 if dt == "bf16" else 1e-4) * abs(loss)
    compare_gradients(model, want, flat, *GATES[dt]["c1"], min_tensors=90, label=f"configs[1] H=8 {dt}")


def _arxiv_like(V, fan, B, ncol, seed):
    rs = np.random.RandomState(seed)
    src, dst, frontier = [], [], rs.choice(V, B, replace=False)
    for f in fan:
        nb = rs.randint(0, V, size=(frontier.size, f))
        src.append(nb.reshape(-1)); dst.append(np.repeat(frontier, f))
        frontier = np.unique(nb)
    src, dst = np.concatenate(src), np.concatenate(dst)
    nodes, inv = np.unique(np.concatenate([src, dst]), return_inverse=True)
    return inv.reshape(2, -1).astype(np.int64), nodes.size


@pytest.mark.parametrize("dt", ["fp32", "bf16"])
def test_configs3_tabgnn_s130_every_gradient(dt):
    """(b) ``TABGNNS`` (utils.py:235-328 -> tabgnn.py:100-151), node classification: S = 130 column attention over >= 2 000
    sampled node rows (129 numerical columns + CLS), S = 2 over the edge rows, d = 128, 8 heads, 2 FT + 2 PNA layers."""
    import tabgnn_amd as T
    from oracle import step as ostep
    st = T.stype
    C, ncol = 128, 129
    ei_np, N = _arxiv_like(40_000, (15, 10), 24, ncol, 3)
    assert N >= 2000
    E = ei_np.shape[1]
    g = torch.Generator().manual_seed(8)
    names_n = {st.numerical: [f"f_{i}" for i in range(ncol - 1)] + ["year"]}
    stats_n = {n: dict(mean=-0.1, std=0.11) for n in names_n[st.numerical]}
    names_e = {st.relation: ["edge_attr"]}
    node_tf = T.TensorFrame({st.numerical: torch.randn(N, ncol, generator=g) * 0.11 - 0.1}, names_n)
    edge_tf = T.TensorFrame({st.relation: torch.ones(E, 1)}, names_e)
    ei = torch.from_numpy(ei_np)
    y = torch.randint(0, 40, (N,), generator=g)
    torch.manual_seed(9)
    cfg = dict(model="tabgnn", task="node_classification", batch_size=24, n_hidden=C, n_gnn_layers=2, n_classes=40,
               dropout=0.0, backbone_dropout=0.0, nhead=8, num_node_features=ncol, num_edge_features=1,
               in_degrees=torch.bincount(ei[1], minlength=N), reverse_mp=False,
               node_encoder=T.StypeWiseFeatureEncoder(C, stats_n, names_n, DT[dt]),
               edge_encoder=T.StypeWiseFeatureEncoder(C, {}, names_e, DT[dt]))
    model = T.TABGNNS(cfg).train()
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}

    def fwd(sd_):
        lg = ostep.tabgnn_wrapper_forward(sd_, 8, 24, _feats(node_tf), ei, _feats(edge_tf), "node_classification",
                                          training=True)
        return lg, torch.nn.functional.cross_entropy(lg, y)
    logits, loss, want = _oracle_grads(sd, fwd)
    model.to(DEV)
    flat = T.FlatParams(model, shadow_dtype=DT[dt])
    flat.zero_grad()
    out = model(node_tf.to(DEV), ei.to(DEV), edge_tf.to(DEV))
    assert out.shape == (N, 40)
    dl = T.ops.weighted_cross_entropy(out, y.to(DEV))
    dl.backward()
    err = (out.detach().float().cpu() - logits).abs()
    print(f"configs[3] {dt} logits: max abs err %.5f, mean %.6f" % (err.max().item(), err.mean().item()))
    assert err.max().item() <= LOGIT_ABS[dt] and err.mean().item() <= (0.01 if dt == "bf16" else 1e-5)
    assert abs(dl.item() - loss) <= (2e-2 if dt == "bf16" else 1e-4) * abs(loss)
    compare_gradients(model, want, flat, *GATES[dt]["c3"], min_tensors=50, label=f"configs[3] S=130 {dt}")


@pytest.mark.parametrize("dt", ["fp32", "bf16"])
def test_configs4_wide64_c256_every_gradient(dt):
    """(c) the fused model on 64 mixed stype columns (32 categorical with up to 10^4 categories, 24 numerical, 8
    timestamp) at C = 256, S = 65, 8 heads, 2 layers, B = 256 seed edges, bf16 (the shape of bench.py's wide64-c256 leg)."""
    import tabgnn_amd as T
    from detparams import rand_subgraph
    from oracle import step as ostep
    st = T.stype
    rs = np.random.RandomState(5)
    C, B, N, E = 256, 256, 900, 2200
    cards = [int(c) for c in np.exp(rs.uniform(np.log(2), np.log(1e4), 32))]
    names = {st.numerical: [f"n{i}" for i in range(24)], st.categorical: [f"c{i}" for i in range(32)],
             st.timestamp: [f"t{i}" for i in range(8)]}
    stats = {**{f"n{i}": dict(mean=0.0, std=1.0) for i in range(24)},
             **{f"c{i}": dict(cardinality=cards[i]) for i in range(32)},
             **{f"t{i}": dict(min_year=2015) for i in range(8)}}
    g = torch.Generator().manual_seed(55)
    cat = torch.stack([torch.randint(-1, c, (E,), generator=g) for c in cards], dim=1)
    ts = torch.stack([torch.randint(2015, 2024, (E, 8), generator=g), torch.randint(0, 12, (E, 8), generator=g),
                      torch.randint(0, 31, (E, 8), generator=g), torch.randint(0, 7, (E, 8), generator=g),
                      torch.randint(0, 24, (E, 8), generator=g), torch.randint(0, 60, (E, 8), generator=g),
                      torch.randint(0, 60, (E, 8), generator=g)], dim=2)
    edge_tf = T.TensorFrame({st.numerical: torch.randn(E, 24, generator=g), st.categorical: cat, st.timestamp: ts}, names)
    node_names = {st.relation: ["node_attr"]}
    node_tf = T.TensorFrame({st.relation: torch.ones(N, 1)}, node_names)
    ei = torch.from_numpy(rand_subgraph(N, E, B, 56))
    y = (torch.arange(B) % 2).long()
    torch.manual_seed(6)
    cfg = dict(model="tabgnnfused", task="edge_classification", batch_size=B, n_hidden=C, n_gnn_layers=2, n_classes=2,
               dropout=0.0, backbone_dropout=0.0, nhead=8, num_node_features=1, num_edge_features=64,
               in_degrees=torch.bincount(ei[1], minlength=N), reverse_mp=False, load_model=None, checkpoint=False,
               node_encoder=T.StypeWiseFeatureEncoder(C, {}, node_names, DT[dt]),
               edge_encoder=T.StypeWiseFeatureEncoder(C, stats, names, DT[dt]))
    model = T.TABGNNFusedS(cfg).train()
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    lw = torch.tensor([1.0, 9.23])

    def fwd(sd_):
        lg = ostep.wrapper_forward(sd_, 8, B, _feats(node_tf), ei, _feats(edge_tf), training=True)
        return lg, ostep.weighted_ce(lg, y, lw)
    logits, loss, want = _oracle_grads(sd, fwd)
    model.to(DEV)
    flat = T.FlatParams(model, shadow_dtype=DT[dt])
    flat.zero_grad()
    out = model(node_tf.to(DEV), ei.to(DEV), edge_tf.to(DEV))
    dl = T.ops.weighted_cross_entropy(out, y.to(DEV), lw.to(DEV))
    dl.backward()
    assert (out.detach().float().cpu() - logits).abs().max().item() <= LOGIT_ABS[dt]
    assert abs(dl.item() - loss) <= (2e-2 if dt == "bf16" else 1e-4) * abs(loss)
    compare_gradients(model, want, flat, *GATES[dt]["c4"], min_tensors=120, label=f"configs[4] C=256 {dt}")


@pytest.mark.parametrize("H", [4])
def test_configs1_spread_state_bf16_gradients(H):
    """the SAME benched bf16 kernels (fused column-transformer chain incl. the in-kernel feed-forward
    weight gradients, gather-fused PNA projections, one-kernel post projection, MLP chain) on a NON-degenerate state: the
    node ``relation`` input is drawn per node (the reference feeds a constant 1, ``ibm_transactions_for_aml.py:296-319``,
    so at random init every node embedding in front of the first BatchNorm is the same row and bf16 rounding of the
    common part was thought to dominate the gradient's small covariance).  MEASURED (round 4): it does not — with spread
    node inputs the worst parameters are the same first-layer PNA weights at 0.07-0.115 (H = 4 / 8) as on the constant
    input (0.07-0.094): the single-batch bf16 noise is flip noise (see the many-batch tests below), so this state keeps a
    gate of SPREAD_GATE and the tight bound comes from the gradient sums.  H = 4 is the benched configuration."""
    import tabgnn_amd as T
    import tabgnn_amd.encoder_layer as EL
    from oracle import step as ostep
    from tabgnn_amd import synthetic as S
    B = 1024
    torch.manual_seed(41 + H)
    cfg = S.make_config(128, 2, H, B, backbone_dropout=0.0, head_dropout=0.0, compute_dtype=torch.bfloat16)
    model = T.TABGNNFusedS(cfg).train()
    node_tf, ei, edge_tf, y = S.make_batch(B, seed=37, p_pos=0.3)
    g = torch.Generator().manual_seed(5)
    node_tf = T.TensorFrame({T.stype.relation: torch.randn(node_tf.num_rows, 1, generator=g) * 1.5 + 0.5}, node_tf.col_names_dict)
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    lw = torch.tensor(cfg["loss_weights"])

    def fwd(sd_):
        lg = ostep.wrapper_forward(sd_, H, B, _feats(node_tf), ei, _feats(edge_tf), training=True)
        return lg, ostep.weighted_ce(lg[:B], y.view(-1), lw)
    logits, loss, want = _oracle_grads(sd, fwd)
    model.to(DEV)
    flat = T.FlatParams(model, shadow_dtype=torch.bfloat16)
    flat.zero_grad()
    n0 = dict(EL.STATS)
    out = model(node_tf.to(DEV), ei.to(DEV), edge_tf.to(DEV))
    dl = T.ops.weighted_cross_entropy(out[:B], y.to(DEV), lw.to(DEV))
    dl.backward()
    assert EL.STATS["fused_fwd"] > n0["fused_fwd"] and EL.STATS.get("fused_bwd_dw", 0) > n0.get("fused_bwd_dw", 0)
    assert (out.detach().float().cpu() - logits).abs().max().item() <= LOGIT_ABS["bf16"]
    compare_gradients(model, want, flat, SPREAD_GATE, 2e-3, min_tensors=90, label=f"configs[1] H={H} bf16, spread node inputs")


class _plain_bf16_composition:
    """Every fusion of the bf16 path switched off (the A/B switches of the package, flipped as module attributes): one-kernel
    column-transformer layer and its chained backward, in-kernel weight gradients, fused tail LayerNorm, gradient sinks,
    gather-fused GEMMs, one-kernel / scaled post projection, weight-fold kernel, MLP chain, timestamp GEMMs.  What is left
    is the composition of single-op kernels that tests/test_gpu_ops.py, test_gpu_round2_ops.py pin one by one."""
    SW = (("encoder_layer", "_FUSED_LAYER", False), ("encoder_layer", "_FUSED_TRAIN", False), ("encoder_layer", "_FUSED_TAIL", False),
          ("encoder_layer", "_DW_FFN", False), ("ops", "GRAD_SINKS", False), ("ops", "_GATHER_GEMM", False),
          ("ops", "_FUSED_POST", False), ("ops", "_POST_FWD_KERNEL", False), ("ops", "_FOLD_HIP", False),
          ("ops", "MLP_CHAIN_MIN_ROWS", 1 << 40), ("encoders", "_TS_GEMM", False))

    def __enter__(self):
        import importlib
        self.saved = []
        for mod, name, val in self.SW:
            m = importlib.import_module("tabgnn_amd." + mod)
            self.saved.append((m, name, getattr(m, name)))
            setattr(m, name, val)

    def __exit__(self, *exc):
        for m, name, val in self.saved:
            setattr(m, name, val)


def _accumulated(cfg_dtype, H, p_drop, batches, B, plain=False):
    """Sum of the parameter gradients over ``batches`` (no zero_grad in between: .grad accumulation) for a model built from
    torch seed 61 + H in compute dtype ``cfg_dtype``; returns also the gradients after the first batch alone."""
    import contextlib
    import tabgnn_amd as T
    from tabgnn_amd import synthetic as S
    torch.manual_seed(61 + H)
    cfg = S.make_config(128, 2, H, B, backbone_dropout=p_drop, head_dropout=0.083 if p_drop else 0.0, compute_dtype=cfg_dtype)
    model = T.TABGNNFusedS(cfg).to(DEV).train()
    flat = T.FlatParams(model, shadow_dtype=cfg_dtype)
    lw = torch.tensor(cfg["loss_weights"], device=DEV)
    flat.zero_grad()
    first = None
    with (_plain_bf16_composition() if plain else contextlib.nullcontext()):
        for i, (node_tf, ei, edge_tf, y) in enumerate(batches):
            T.ops.DropoutRNG.new_step(900 + i)
            out = model(node_tf, ei, edge_tf)
            T.ops.weighted_cross_entropy(out[:B], y, lw).backward()
            if i == 0:
                first = {k: p.grad.detach().float().cpu().clone() for k, p in model.named_parameters()}
    return model, flat, first, {k: p.grad.detach().float().cpu().clone() for k, p in model.named_parameters()}


@pytest.mark.parametrize("H,p_drop", [(4, 0.0), (8, 0.0)])
def test_benched_bf16_gradients_are_unbiased_over_many_batches(H, p_drop):
    """VERDICT r03 weak 1b / item 4 — a step-level bound on any WRONG TERM in the benched bf16 kernels that is tighter than
    the single-batch bf16 gate.  Measured this round: a single batch disagrees by 0.07-0.15 per parameter not only with
    fp32 but also between two bf16 paths (benched kernels vs the plain composition of single-op bf16 kernels, identical
    masks), and drawing the node inputs per node does not change that — the noise is discrete decisions flipping under
    bf16 rounding (ReLU / LeakyReLU gates, max / min argmax routing of the PNA aggregators), each a 100 % local change, not
    the degenerate BatchNorm input.  Such flips are independent from batch to batch while a wrong or mis-scaled term is
    not: over K = 16 batches the gradient SUM (.grad accumulation, what an optimiser with gradient accumulation consumes)
    of the benched bf16 path must agree with the fp32 kernels' sum (pinned to the oracle at 2e-3 by the fp32 twins above)
    within SUM_GATE per parameter — well below the single-batch noise, which the test prints beside it.  Dropout 0 here (the
    fp32 path allocates its dropout streams in another order, so its masks differ); the dropout-on twin is the next test."""
    from tabgnn_amd import synthetic as S
    import tabgnn_amd.encoder_layer as EL
    B, K = 1024, 16
    batches = [S.make_batch(B, seed=300 + i, device=DEV, p_pos=0.3) for i in range(K)]
    n0 = dict(EL.STATS)
    model, flat, first_b, sum_b = _accumulated(torch.bfloat16, H, p_drop, batches, B)
    assert EL.STATS["fused_fwd"] > n0["fused_fwd"] and EL.STATS.get("fused_bwd_dw", 0) > n0.get("fused_bwd_dw", 0)
    _, _, first_f, sum_f = _accumulated(torch.float32, H, p_drop, batches, B)
    single = compare_gradients(model, first_f, None, 1.0, 1e-3, min_tensors=90, label=f"ONE batch, benched bf16 vs fp32 kernels, H={H} p={p_drop}",
                               got=first_b)
    rows = compare_gradients(model, sum_f, None, SUM_GATE, 1e-3, min_tensors=90,
                             label=f"SUM over {K} batches, benched bf16 vs fp32 kernels, H={H} p={p_drop}", got=sum_b)
    w1 = max(r[1] for r in single if r[2] >= 0.01)
    wk = max(r[1] for r in rows if r[2] >= 0.01)
    print(f"   worst large-tensor error: one batch {w1:.4f}, sum of {K} {wk:.4f}")
    assert wk < 0.75 * w1                              # the disagreement averages out: noise, not bias


@pytest.mark.parametrize("H", [4, 8])
def test_benched_bf16_kernels_against_the_plain_bf16_composition(H):
    """The dropout-ON twin (reference defaults 0.5 backbone / 0.083 head): the benched path against the plain composition of
    single-op bf16 kernels — same weights, same batches, the SAME counter-based masks (every fused kernel draws the masks of
    the ops it replaces; a mask mismatch shows as an error of order 1, as the fp32 path with its own stream order does).
    Both carry bf16 rounding at slightly different points (the fused kernels keep q/k/v, the hidden activation and the PNA
    concatenations in fp32 registers), so ONE batch differs by the same flip noise as above; the sum over K = 16 batches
    must agree within TWIN_GATE per parameter."""
    from tabgnn_amd import synthetic as S
    import tabgnn_amd.encoder_layer as EL
    B, K = 1024, 16
    batches = [S.make_batch(B, seed=400 + i, device=DEV, p_pos=0.3) for i in range(K)]
    n0 = dict(EL.STATS)
    model, flat, first_b, sum_b = _accumulated(torch.bfloat16, H, 0.5, batches, B)
    assert EL.STATS["fused_fwd"] > n0["fused_fwd"] and EL.STATS.get("fused_bwd_dw", 0) > n0.get("fused_bwd_dw", 0)
    n1 = dict(EL.STATS)
    _, _, first_p, sum_p = _accumulated(torch.bfloat16, H, 0.5, batches, B, plain=True)
    assert EL.STATS["fused_fwd"] == n1["fused_fwd"]                       # the plain run took no fused kernel
    single = compare_gradients(model, first_p, None, 0.4, 1e-3, min_tensors=90, label=f"ONE batch, benched vs plain bf16, H={H} p=0.5",
                               got=first_b)
    rows = compare_gradients(model, sum_p, None, TWIN_GATE, 1e-3, min_tensors=90,
                             label=f"SUM over {K} batches, benched vs plain bf16, H={H} p=0.5", got=sum_b)
    w1 = max(r[1] for r in single if r[2] >= 0.01)
    wk = max(r[1] for r in rows if r[2] >= 0.01)
    print(f"   worst large-tensor error: one batch {w1:.4f}, sum of {K} {wk:.4f}")
