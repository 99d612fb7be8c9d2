"""Run-to-run determinism of the column-transformer kernels at a size that fills every CU with two workgroups.

Round 4 found two defects that only showed as run-to-run differences: type-punned LDS accesses of the row
restage reordered under strict aliasing (z1 / z2 rows with dwords of later temporaries), and one wave tile of the output
wrong at up to 13 % of the launches whenever a wave had two LDS-DMA instructions in flight beside a co-resident
workgroup.  Both needed >= 2 workgroups per CU and >= 512 workgroups, which no parity test reaches (the oracle sizes are
far smaller), so the guard is this repeat test; tools/enc_det_probe4.py measures the RATE over thousands of launches
(0 of 22 500 with the shipped DMA helper)."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "models-for-relational-multimodal-data_amd")]

pytestmark = pytest.mark.gpu

R, S, C, REPS = 13000, 6, 128, 600        # launches per case: a per-launch failure rate of 0.3 % is caught with probability 0.8


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


@pytest.mark.parametrize("p", [0.0, 0.5])
def test_fused_forward_outputs_and_saved_rows_repeat_bit_exactly(p):
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


@pytest.mark.parametrize("p", [0.0, 0.5])
def test_layer_forward_and_backward_repeat_bit_exactly(p):
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
