"""Run-to-run determinism of the column-transformer kernels.

Rounds 3-4 shipped two defects that showed only as run-to-run differences (LDS-DMA hazard): type-punned LDS
accesses of the row restage reordered under strict aliasing (z1 / z2 rows with dwords of later temporaries; closed by
-fno-strict-aliasing), and one wave tile of the output wrong at 2-10 % of the launches with two workgroups per CU: the
last ``ds_read_b128`` of a weight unit were still in the LDS queue when the wave crossed the raw ``s_barrier`` that hands
the unit's buffer to the next LDS-DMA, and that DMA could land first.  Round 5 closed it by construction (``s_waitcnt
vmcnt(K) lgkmcnt(0)`` in front of every unit barrier; tests/test_cabi_and_host.py audits the compiled assembly) — this
repeat test stays as the end-to-end guard: tools/enc_det_probe4.py, same library with and without that wait,
2 000 launches each at R = 13 000: 208 / 192 / 46 launches differ without it (eval / train p = 0 / p = 0.5), 0 / 0 / 0 with.
Row counts: fewer workgroups than CUs (R = 1 200), the size the defect was found at (13 000: two workgroups on every
CU, one or two tiles each) and the bench's edge table (430 162)."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "models-for-relational-multimodal-data_amd")]

pytestmark = pytest.mark.gpu

S, C = 6, 128
CASES = [(1200, 300), (13000, 600), (430162, 40)]        # (table rows, launches per case)


def _layer(dev):
    from tabgnn_amd.layers import ColumnTransformerLayer
    torch.manual_seed(0)
    layer = ColumnTransformerLayer(C, 4, C, dropout=0.5).to(dev)
    tail = torch.nn.LayerNorm(C).to(dev)
    for q in list(layer.parameters()) + list(tail.parameters()):
        q._lp = q.detach().to(torch.bfloat16)
        if q.dim() == 2:
            q._lp_t = q._lp.t().contiguous()
    return layer, tail


@pytest.mark.parametrize("R,REPS", CASES)
@pytest.mark.parametrize("p", [0.0, 0.5])
def test_fused_forward_outputs_and_saved_rows_repeat_bit_exactly(p, R, REPS):
    import tabgnn_amd.encoder_layer as EL
    dev = "cuda:0"
    layer, tail = _layer(dev)
    sa = layer.self_attn
    bf = lambda t: t.detach().to(torch.bfloat16).contiguous()
    wpack, prm = EL.pack_layer(bf(sa.in_proj_weight), bf(sa.out_proj.weight), bf(layer.linear1.weight), bf(layer.linear2.weight),
                               sa.in_proj_bias, sa.out_proj.bias, layer.norm1.weight, layer.norm1.bias, layer.linear1.bias,
                               layer.linear2.bias, layer.norm2.weight, layer.norm2.bias, tail.weight, tail.bias)
    x = torch.randn(R, S, C, device=dev).to(torch.bfloat16)
    first = None
    for r in range(REPS):
        got = EL.fused_forward(x, 4, p, True, 0.5, 0.5, wpack, prm, 7, [1, 2, 3, 4], True)
        if first is None:
            first = [t.clone() for t in got]
            continue
        for name, u, v in zip(("out", "z1", "z2"), got, first):
            assert torch.equal(u, v), f"launch {r}: {name} differs in {int((u != v).any(-1).sum())} token rows"


@pytest.mark.parametrize("R,REPS", CASES)
@pytest.mark.parametrize("p", [0.0, 0.5])
def test_layer_forward_and_backward_repeat_bit_exactly(p, R, REPS):
    import tabgnn_amd.encoder_layer as EL
    from tabgnn_amd import ops
    dev = "cuda:0"
    layer, tail = _layer(dev)
    x = torch.randn(R, S, C, device=dev).to(torch.bfloat16)
    g = torch.randn(R, S, C, device=dev).to(torch.bfloat16)
    params = [q for q in list(layer.parameters()) + list(tail.parameters())]
    first = None
    for r in range(REPS // 2):
        ops.DropoutRNG.new_step(7)
        for q in params:
            q.grad = None
        xr = x.clone().requires_grad_(True)
        out = EL.encoder_layer(xr, layer, p, tail, 0.5, 0.5)
        out.backward(g)
        got = [out.detach(), xr.grad] + [q.grad for q in params]
        if first is None:
            first = [t.clone() for t in got]
            continue
        names = ["out", "d_x"] + [f"grad[{i}] {tuple(q.shape)}" for i, q in enumerate(params)]
        for name, u, v in zip(names, got, first):
            assert torch.equal(u, v), f"pass {r}: {name} differs, max |diff| {float((u.float() - v.float()).abs().max()):.3e}"
