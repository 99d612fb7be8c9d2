"""Port numbering and ego ids (src/datasets/util/graph.py:68-145): oracle restatement and the native
``tg_edge_ports`` against a golden produced by the reference's own functions, plus randomised oracle-vs-native cases
(ties, parallel edges, self loops, isolated nodes, empty graph).  Host-side code: runs without a GPU."""
import os

import numpy as np
import pytest
import torch

from golden_util import GOLDEN


def _golden():
    z = np.load(os.path.join(GOLDEN, "ports_n40.npz"))
    return z["edge_index"], z["timestamps"], int(z["num_nodes"]), z["in_ports"], z["out_ports"]


def test_oracle_ports_match_reference():
    from oracle.ports import edge_ports
    ei, ts, n, pin, pout = _golden()
    a, b = edge_ports(ei, ts, n)
    np.testing.assert_array_equal(a, pin)
    np.testing.assert_array_equal(b, pout)


def test_native_ports_match_reference():
    from tabgnn_amd.sampler import edge_ports
    ei, ts, n, pin, pout = _golden()
    for threads in (1, 4):
        a, b = edge_ports(torch.from_numpy(ei), torch.from_numpy(ts), n, num_threads=threads)
        assert a.dtype == torch.float32 and tuple(a.shape) == (ei.shape[1], 1)
        np.testing.assert_array_equal(a.numpy(), pin)
        np.testing.assert_array_equal(b.numpy(), pout)


@pytest.mark.parametrize("seed,n,e,tmax", [(0, 30, 400, 10), (1, 200, 3000, 50), (2, 5, 60, 2), (3, 50, 200, 10 ** 9)])
def test_native_ports_match_oracle_with_ties(seed, n, e, tmax):
    from oracle.ports import edge_ports as oracle_ports
    from tabgnn_amd.sampler import edge_ports
    rs = np.random.RandomState(seed)
    ei = np.stack([rs.randint(0, n - 2, size=e), rs.randint(0, n - 2, size=e)]).astype(np.int64)
    ei[:, :5] = ei[0, :5]                                   # self loops
    ts = rs.randint(0, tmax, size=e).astype(np.int64)       # many equal timestamps
    a, b = edge_ports(ei, ts, n)
    oa, ob = oracle_ports(ei, ts, n)
    np.testing.assert_array_equal(a.numpy(), oa)
    np.testing.assert_array_equal(b.numpy(), ob)
    a0, b0 = edge_ports(ei, None, n)                        # no timestamps: all zero (graph.py:70)
    oa0, ob0 = oracle_ports(ei, None, n)
    np.testing.assert_array_equal(a0.numpy(), oa0)
    np.testing.assert_array_equal(b0.numpy(), ob0)


def test_ports_properties_large_and_edges():
    from tabgnn_amd.sampler import edge_ports
    rs = np.random.RandomState(7)
    n, e = 20000, 400000
    ei = np.stack([rs.randint(0, n, size=e), rs.zipf(1.7, size=e) % n]).astype(np.int64)
    ts = rs.randint(0, 10 ** 6, size=e).astype(np.int64)
    a, b = edge_ports(ei, ts, n, num_threads=4)
    a1, b1 = edge_ports(ei, ts, n, num_threads=1)
    assert torch.equal(a, a1) and torch.equal(b, b1)        # thread count does not change the result
    a = a.numpy().astype(np.int64).ravel()
    # per destination the in-ports are exactly 0..(#distinct sources - 1), and equal pairs share a port
    pair = ei[0] * n + ei[1]
    order = np.argsort(pair, kind="stable")
    same = pair[order][1:] == pair[order][:-1]
    assert np.all(a[order][1:][same] == a[order][:-1][same])
    distinct = np.zeros(n, dtype=np.int64)
    np.add.at(distinct, ei[1][order][np.r_[True, ~same]], 1)
    mx = np.full(n, -1, dtype=np.int64)
    np.maximum.at(mx, ei[1], a)
    assert np.array_equal(mx[distinct > 0], distinct[distinct > 0] - 1)
    # empty graph and bad ids
    z, _ = edge_ports(np.zeros((2, 0), dtype=np.int64), None, 3)
    assert tuple(z.shape) == (0, 1)
    with pytest.raises(ValueError):
        edge_ports(np.array([[0], [5]]), None, 3)


def test_ego_ids():
    from tabgnn_amd.frame import TensorFrame, stype
    from tabgnn_amd.sampler import add_ego_ids, add_ego_ids_from_nodes
    rel = torch.full((10, 2), 7.0)
    tf = TensorFrame({stype.relation: rel}, {stype.relation: ["node", "EgoID"]})
    add_ego_ids(tf, torch.tensor([[1, 4, 4], [2, 9, 1]]))
    assert tf.feat_dict[stype.relation][:, 1].tolist() == [0, 1, 1, 0, 1, 0, 0, 0, 0, 1]
    assert torch.all(tf.feat_dict[stype.relation][:, 0] == 7.0)
    add_ego_ids_from_nodes(tf, 3)
    assert tf.feat_dict[stype.relation][:, 1].tolist() == [1, 1, 1, 0, 0, 0, 0, 0, 0, 0]
