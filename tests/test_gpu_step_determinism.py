"""Run-to-run determinism of the WHOLE benched train step at bench size (B = 8192 seeds, ~438 k edges, bf16, dropout on):
the same state, batch and step seed twice -> logits, every gradient, the updated weights and Adam moments bit for bit.
Every reduction on the path has a fixed order (block partials summed in block order, segmented sums in CSR order, no
float atomics for tables of <= 64 rows: DESIGN 3), so anything that differs between two runs is a defect — this is the
test that would have caught round 4's two column-transformer defects at the size where they showed."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "models-for-relational-multimodal-data_amd")]

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("p_backbone", [0.5, 0.0])
def test_bench_size_train_step_repeats_bit_for_bit(p_backbone):
    import tabgnn_amd as T
    from tabgnn_amd import synthetic as S
    from tabgnn_amd.sampler import batch_index
    dev = torch.device("cuda:0")
    B = 8192
    cfg = S.make_config(128, 2, 4, B, compute_dtype=torch.bfloat16)
    if p_backbone == 0.0:
        cfg["backbone_dropout"] = 0.0
        cfg["dropout"] = 0.0
    b = S.make_batch(B, seed=3, device=dev)
    batch = (b[0], batch_index(b[1].cpu(), b[0].num_rows, B, dev), b[2], b[3])
    lw = torch.tensor(cfg["loss_weights"], device=dev)
    runs = []
    torch.manual_seed(7)
    sd = {k: v.clone() for k, v in T.TABGNNFusedS(cfg).to(dev).state_dict().items()}
    for _ in range(2):
        model = T.TABGNNFusedS(cfg).to(dev).train()
        model.load_state_dict(sd)
        flat = T.FlatParams(model, shadow_dtype=torch.bfloat16)
        opt = T.FusedAdam(flat, lr=cfg["lr"])
        outs = []
        for step in range(3):
            loss, logits = T.train_step(model, flat, opt, batch, lw, step_seed=100 + step)
            outs += [logits.clone(), flat.grad.clone(), flat.flat.clone(), opt.m.clone(), opt.v.clone()]
        runs.append(outs)
        del model, flat, opt
    names = ["logits", "gradients", "weights", "adam m", "adam v"]
    for i, (u, v) in enumerate(zip(*runs)):
        if not torch.equal(u, v):
            d = (u.float() - v.float()).abs()
            raise AssertionError(f"step {i // 5} {names[i % 5]}: {int((d > 0).sum())} of {d.numel()} elements differ, max {float(d.max()):.3e}")


def test_wide64_c256_train_step_repeats_bit_for_bit():
    """BASELINE configs[4]'s batch shape (64 mixed columns, 19 embedding tables of more than 64 rows, d = 256): since
    round 5 the big tables' gradients are reduced in a fixed order too (tg_embed_grad_sorted: counting sort of the batch's
    (column, category) pairs, one wave per bucket) — before, float atomicAdd made them differ from run to run."""
    import numpy as np
    import tabgnn_amd as T
    from tabgnn_amd import synthetic as S
    dev = torch.device("cuda:0")
    B = 256
    names, stats, cards = S.wide64_columns()
    assert sum(c + 1 > 64 for c in cards) >= 10
    ei, N = S.sampled_subgraph(B, 1)
    E = ei.shape[1]
    r = np.random.RandomState(2)
    cat = np.stack([r.randint(0, c, E) for c in cards], 1).astype(np.int64)
    num = r.randn(E, 24).astype(np.float32)
    ts = np.stack([r.randint(2015, 2024, (E, 8)), r.randint(0, 12, (E, 8)), r.randint(0, 28, (E, 8)), r.randint(0, 7, (E, 8)),
                   r.randint(0, 24, (E, 8)), r.randint(0, 60, (E, 8)), r.randint(0, 60, (E, 8))], -1).astype(np.int64)
    st = T.stype
    etf = T.TensorFrame({st.numerical: torch.from_numpy(num), st.categorical: torch.from_numpy(cat),
                         st.timestamp: torch.from_numpy(ts)}, names).to(dev)
    ntf = T.TensorFrame({st.relation: torch.ones(N, 1)}, S.NODE_COLS).to(dev)
    y = torch.from_numpy((r.rand(B) < 0.05).astype(np.int64)).to(dev)
    batch = (ntf, torch.from_numpy(ei).to(dev), etf, y)
    lw = torch.tensor([1.0, 9.23], device=dev)
    torch.manual_seed(7)
    sd = None
    runs = []
    for _ in range(2):
        model = T.TABGNNFusedS(S.wide64_config(B, S.in_degrees_like(), torch.bfloat16)).to(dev).train()
        if sd is None:
            sd = {k: v.clone() for k, v in model.state_dict().items()}
        model.load_state_dict(sd)
        flat = T.FlatParams(model, shadow_dtype=torch.bfloat16)
        opt = T.FusedAdam(flat, lr=6e-4)
        outs = []
        for step in range(2):
            loss, logits = T.train_step(model, flat, opt, batch, lw, step_seed=50 + step)
            outs += [logits.clone(), flat.grad.clone(), flat.flat.clone()]
        runs.append(outs)
        del model, flat, opt
    emb_grad_nonzero = False
    for i, (u, v) in enumerate(zip(*runs)):
        if not torch.equal(u, v):
            d = (u.float() - v.float()).abs()
            raise AssertionError(f"step {i // 3} {['logits', 'gradients', 'weights'][i % 3]}: {int((d > 0).sum())} of {d.numel()} "
                                 f"elements differ, max {float(d.max()):.3e}")
    assert all(torch.isfinite(t.float()).all() for t in runs[0])
