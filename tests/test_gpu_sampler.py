"""The device k-hop sampler + relabel (csrc/sampler_gpu.hip) against the semantics of
sample_neighbors + get_graph_inputs (src/datasets/ibm_transactions_for_aml.py:61-112,159-180) and against the host
sampler where the two must agree exactly (everything that does not depend on a random draw).  The reference's sampler is
unseeded, so parity is structural, as for the host sampler (tests/test_sampler.py)."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "models-for-relational-multimodal-data_amd")]

pytestmark = pytest.mark.gpu


def _graph(N, E, seed, hubs=0):
    rs = np.random.RandomState(seed)
    src = rs.randint(0, N, E)
    dst = rs.randint(0, N, E)
    if hubs:                                   # a few destinations with far more in-edges than any fan-out
        m = rs.rand(E) < 0.3
        dst[m] = rs.randint(0, hubs, int(m.sum()))
    return np.stack([src, dst]).astype(np.int64)


def _check(ei, N, fan, seeds, eid, lei, nodes):
    E = ei.shape[1]
    B = len(seeds)
    eid, lei, nodes = eid.cpu().numpy(), lei.cpu().numpy(), nodes.cpu().numpy()
    assert (eid[:B] == seeds).all()                                          # seed edges first, in order
    assert len(np.unique(eid)) == len(eid)                                   # no edge twice
    assert (np.diff(nodes) > 0).all()                                        # sorted unique nodes
    assert (nodes[lei[0]] == ei[0, eid]).all() and (nodes[lei[1]] == ei[1, eid]).all()      # relabel = rank in nodes
    assert set(nodes.tolist()) == set(ei[:, eid].reshape(-1).tolist())      # nodes = exactly the endpoints
    deg = np.bincount(ei[1], minlength=N)
    seed_set = set(seeds.tolist())
    # hop 1 expands the seed endpoints: each of them has min(deg, fan) in-edges in the sample (seed edges count as drawn
    # when the draw hit them, so "at most fan sampled + whatever seeds" and "everything when deg <= fan")
    sampled_by_dst = {}
    for e in eid[B:]:
        sampled_by_dst.setdefault(int(ei[1, e]), []).append(int(e))
    in_edges = {}
    for v in set(ei[:, seeds].reshape(-1).tolist()):
        in_v = np.nonzero(ei[1] == v)[0]
        got = set(sampled_by_dst.get(v, []))
        assert got <= set(in_v.tolist())
        if fan[0] < 0 or deg[v] <= fan[0]:
            assert got == set(in_v.tolist()) - seed_set, f"node {v}: every in-edge expected"
        else:
            assert len(got) <= fan[0] and len(got) >= fan[0] - len(seed_set & set(in_v.tolist()))
    for v, es in sampled_by_dst.items():                                     # nobody is expanded twice
        f = max(f_ if f_ >= 0 else E for f_ in fan)
        assert len(es) <= f


@pytest.mark.parametrize("N,E,B,fan,hubs", [(50, 400, 4, (3, 2), 0), (2000, 30000, 64, (10, 5), 0), (5000, 200000, 200, (100, 100), 7),
                                           (300, 5000, 16, (-1, 4), 0), (100, 50, 8, (5, 5), 0), (1000, 20000, 32, (128, 1), 3)])
def test_device_sampler_structure(N, E, B, fan, hubs):
    from tabgnn_amd.device_sampler import DeviceNeighborSampler
    ei = _graph(N, E, N + E, hubs)
    smp = DeviceNeighborSampler(ei, N, fan)
    rs = np.random.RandomState(1)
    seeds = rs.choice(E, B, replace=False)
    out = smp.sample(seeds, rng_seed=5)
    _check(ei, N, fan, seeds, *out)
    again = smp.sample(seeds, rng_seed=5)                                    # a pure function of its inputs
    for a, b in zip(out, again):
        assert torch.equal(a, b)
    assert int(smp._seedbit.sum()) == 0                                      # the seed bitmap is clean between calls
    other = smp.sample(seeds, rng_seed=6)
    _check(ei, N, fan, seeds, *other)


def test_device_sampler_equals_host_sampler_where_no_draw_decides():
    """Fan-out >= every in-degree: nothing is random, so the sampled EDGE SET, the node list and the relabelling must be
    the host sampler's (the order of the sampled edges may differ: frontier order vs sampling order)."""
    from tabgnn_amd.device_sampler import DeviceNeighborSampler
    from tabgnn_amd.sampler import NeighborSampler
    N, E, B = 3000, 20000, 50
    ei = _graph(N, E, 3)
    assert np.bincount(ei[1], minlength=N).max() <= 100
    seeds = np.random.RandomState(2).choice(E, B, replace=False)
    d_eid, d_ei, d_nodes = DeviceNeighborSampler(ei, N, (100, 100)).sample(seeds, 9)
    h_eid, h_ei, h_nodes = NeighborSampler(ei, N, (100, 100)).sample(seeds, 9)
    assert torch.equal(d_nodes.cpu(), h_nodes)
    assert torch.equal(d_eid[:B].cpu(), h_eid[:B])
    assert set(d_eid.cpu().tolist()) == set(h_eid.tolist())
    order_d, order_h = torch.argsort(d_eid.cpu()), torch.argsort(h_eid)
    assert torch.equal(d_ei.cpu()[:, order_d], h_ei[:, order_h])


def test_device_sampler_draws_are_uniform_over_a_hub():
    """Floyd's draw on the device: every in-edge of a hub is taken with frequency fan / deg (chi-square, loose gate)."""
    from tabgnn_amd.device_sampler import DeviceNeighborSampler
    N, deg, fan = 40, 400, 20
    ei = np.stack([np.random.RandomState(0).randint(1, N, deg), np.zeros(deg, dtype=np.int64)]).astype(np.int64)
    ei = np.concatenate([ei, np.array([[0], [1]], dtype=np.int64)], axis=1)          # the seed edge 0 -> 1; node 0 is the hub
    smp = DeviceNeighborSampler(ei, N, (fan, 0))
    hits = np.zeros(deg)
    T = 600
    for t in range(T):
        eid, _, _ = smp.sample(np.array([deg]), rng_seed=1000 + t)
        got = eid.cpu().numpy()[1:]
        got = got[ei[1, got] == 0]
        assert len(got) == fan and len(set(got.tolist())) == fan
        hits[got] += 1
    exp = T * fan / deg
    chi2 = ((hits - exp) ** 2 / exp).sum()
    assert chi2 < deg + 6 * np.sqrt(2 * deg), f"chi2 {chi2:.1f} for {deg} cells"


def test_device_sampler_rejects_bad_seeds():
    from tabgnn_amd.device_sampler import DeviceNeighborSampler
    ei = _graph(100, 500, 0)
    smp = DeviceNeighborSampler(ei, 100, (5, 5))
    with pytest.raises(ValueError):
        smp.sample(np.array([3, 500]))
    out = smp.sample(np.array([3, 4]))                                       # the handle stays usable
    _check(ei, 100, (5, 5), np.array([3, 4]), *out)
