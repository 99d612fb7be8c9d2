"""GPU: the gather-fused projections (tg_gemm_nt_gather3_bf16 / tg_gemm_tn_gather3_bf16 behind ops.edge_linear and
ops.edge_mlp_relu) against the materialised composition edge_gather -> linear / mlp_relu: the same MFMA products over
the same rows, so outputs and every gradient agree to accumulation-order rounding; sizes with ragged tiles, hub nodes
and the destination-sorted layout."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _graph(N, E, seed, hub=False):
    from tabgnn_amd import ops
    rs = np.random.RandomState(seed)
    src, dst = rs.randint(0, N, E), rs.randint(0, N, E)
    if hub:
        dst[: E // 3] = 7                      # one destination with a third of the edges
    ei = torch.from_numpy(np.stack([src, dst])).to(DEV)
    return ops.SubgraphIndex.build(ei, N)


@pytest.mark.parametrize("N,E,first,hub", [(300, 1000, "src", False), (5000, 70001, "dst_sorted", False),
                                           (2000, 40000, "dst_sorted", True), (64, 129, "dst", False),
                                           (20000, 300000, "src", False)])
def test_gather_gemms_equal_the_materialised_composition(N, E, first, hub):
    from tabgnn_amd import ops
    torch.manual_seed(N + E)
    g = _graph(N, E, 1, hub)
    F = 128
    mk = lambda *s: (torch.randn(*s, device=DEV) * 0.5).to(torch.bfloat16).requires_grad_(True)
    lin0, lin2 = torch.nn.Linear(3 * F, F).to(DEV), torch.nn.Linear(F, F).to(DEV)
    res = []
    for fused in (True, False):
        ops._GATHER_GEMM = fused
        try:
            x, e = mk(N, F), mk(E, F)
            torch.manual_seed(5)
            with torch.no_grad():
                x.copy_(torch.randn(N, F, device=DEV) * 0.5); e.copy_(torch.randn(E, F, device=DEV) * 0.5)
            for p in list(lin0.parameters()) + list(lin2.parameters()):
                p.grad = None
            h = ops.edge_linear(x, e, g, first, lin0.weight, lin0.bias)
            u = ops.edge_mlp_relu(x, e, g, first, lin0, lin2)
            cot = torch.randn(E, F, device=DEV, generator=torch.Generator(DEV).manual_seed(3)).to(torch.bfloat16)
            ((h.float() * cot.float()).sum() + (u.float() * cot.float().flip(0)).sum()).backward()
            res.append((h.detach().float(), u.detach().float(), x.grad.float(), e.grad.float(),
                        [p.grad.clone() for p in list(lin0.parameters()) + list(lin2.parameters())]))
        finally:
            ops._GATHER_GEMM = True
    a, b = res
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])          # same products, same order within a row
    for i in (2, 3):
        scale = b[i].abs().max().item()
        assert (a[i] - b[i]).abs().max().item() <= 2e-2 * scale, i      # bf16 partial sums in different orders
    for pa, pb in zip(a[4], b[4]):
        assert (pa - pb).abs().max().item() <= 2e-3 * pb.abs().max().item() + 1e-6


def test_edge_mlp_relu_refuses_an_unstashed_inplace_change_of_its_node_input():
    """ops._MLPReluGather re-reads its node input in the backward by plain reference (outside autograd's saved-tensor
    version check).  An in-place ``seed_pool`` on that tensor WITHOUT ``stash`` must raise there instead of giving a
    silently wrong dW0; with ``stash`` the rows are put back and the gradients equal those of pooling into a copy."""
    from tabgnn_amd import ops
    torch.manual_seed(0)
    N, E, B, F, C = 500, 3000, 40, 128, 128
    g = _graph(N, E, 2)
    rs = np.random.RandomState(9)
    tei = torch.from_numpy(np.stack([rs.randint(0, N, B), rs.randint(0, N, B)])).to(DEV)
    seeds = ops.SeedIndex(tei, N)
    lin0, lin2 = torch.nn.Linear(3 * F, F).to(DEV), torch.nn.Linear(F, F).to(DEV)
    x0 = (torch.randn(N, F, device=DEV) * 0.5).to(torch.bfloat16)
    e0 = (torch.randn(E, F, device=DEV) * 0.5).to(torch.bfloat16)
    xf0 = (torch.randn(B, C + 2 * F, device=DEV) * 0.5).to(torch.bfloat16)

    def run(inplace, stash):
        for p in list(lin0.parameters()) + list(lin2.parameters()):
            p.grad = None
        xin = x0.clone().requires_grad_(True)
        x = xin * 1.0                                   # an intermediate, as x_gnn in the fused layer
        e = e0.clone().requires_grad_(True)
        assert ops.edge_mlp_rereads_x(x, e, lin0, lin2)
        u = ops.edge_mlp_relu(x, e, g, "src", lin0, lin2)
        pooled = ops.seed_pool(x, xf0.clone().requires_grad_(True), seeds, C, inplace=inplace, stash=stash)
        (u.float().sum() + pooled.float().sum()).backward()
        return xin.grad.float().clone(), [p.grad.clone() for p in list(lin0.parameters()) + list(lin2.parameters())]

    with pytest.raises(RuntimeError, match="modified in place"):
        run(True, False)
    gx_c, gp_c = run(False, False)
    gx_s, gp_s = run(True, True)
    assert torch.equal(gx_c, gx_s)
    for a, b in zip(gp_c, gp_s):
        assert torch.equal(a, b)
