"""CPU: native neighbour sampler + relabel (libtabgnn_sampler.so) — exact k-hop neighbourhood when the fan-out covers
every in-edge, contract properties under real sampling (seed edges first and in order, no seed edge repeated, fan-out
respected per expanded node, relabel = rank among sorted unique endpoints), determinism across thread counts."""
import os

import numpy as np
import pytest
import torch

from tabgnn_amd.sampler import ColumnStore, NeighborSampler
from tabgnn_amd import synthetic as S
from tabgnn_amd.frame import stype


def _graph(n=300, e=4000, seed=0):
    rs = np.random.RandomState(seed)
    src = rs.randint(0, n, e)
    dst = (rs.zipf(1.6, e) - 1) % n            # heavy-tailed in-degree
    return np.stack([src, dst]).astype(np.int64), n


def _khop_reference(ei, seeds, hops):
    """Exact directed k-hop in-neighbourhood (every in-edge taken), in the sampler's output order."""
    src, dst = ei
    frontier = np.unique(np.concatenate([src[seeds], dst[seeds]]))
    visited = set(frontier.tolist())
    seed_set = set(seeds.tolist())
    out = list(seeds)
    order = np.argsort(dst, kind="stable")
    ptr = np.concatenate([[0], np.cumsum(np.bincount(dst, minlength=ei.max() + 1))])
    for _ in range(hops):
        nxt = []
        for v in frontier:
            for e in order[ptr[v]:ptr[v + 1]]:
                if e not in seed_set:
                    out.append(e)
                if src[e] not in visited:
                    visited.add(src[e]); nxt.append(src[e])
        frontier = np.array(nxt, dtype=np.int64)
    return np.array(out, dtype=np.int64)


def test_full_fanout_equals_exact_khop_neighbourhood():
    ei, n = _graph()
    s = NeighborSampler(ei, n, num_neighbors=(-1, -1))
    seeds = np.array([5, 17, 900, 901, 3999, 17 + 1], dtype=np.int64)
    eid, edge_index, nodes = s.sample(seeds)
    want = _khop_reference(ei, seeds, 2)
    assert np.array_equal(eid.numpy(), want)
    glob = nodes.numpy()[edge_index.numpy()]                       # local -> global round trip
    assert np.array_equal(glob[0], ei[0][want]) and np.array_equal(glob[1], ei[1][want])
    assert np.array_equal(nodes.numpy(), np.unique(np.concatenate([ei[0][want], ei[1][want]])))


def test_sampling_contract_and_thread_invariance():
    ei, n = _graph(2000, 60000, seed=3)
    seeds = np.random.RandomState(1).choice(60000, 200, replace=False).astype(np.int64)
    outs = []
    for threads in (1, 4):
        s = NeighborSampler(ei, n, num_neighbors=(10, 5), num_threads=threads)
        outs.append(s.sample(seeds, rng_seed=42))
    eid, edge_index, nodes = outs[0]
    assert all(torch.equal(a, b) for a, b in zip(outs[0], outs[1]))             # same result for any thread count
    eid_np = eid.numpy()
    assert np.array_equal(eid_np[:200], seeds)                                   # seed edges first, in order
    assert len(np.unique(eid_np)) == len(eid_np)                                 # no edge twice (seed edges dropped)
    sampled = eid_np[200:]
    # every sampled edge enters a node that was expanded, at most max(fanout) in-edges kept per destination
    per_dst = np.bincount(ei[1][sampled], minlength=n)
    assert per_dst.max() <= 10
    glob = nodes.numpy()[edge_index.numpy()]
    assert np.array_equal(glob[0], ei[0][eid_np]) and np.array_equal(glob[1], ei[1][eid_np])
    assert np.all(np.diff(nodes.numpy()) > 0)
    other = NeighborSampler(ei, n, num_neighbors=(10, 5)).sample(seeds, rng_seed=43)
    assert not torch.equal(other[0], eid)                                        # the seed matters


def test_sampler_rejects_bad_input():
    ei, n = _graph()
    with pytest.raises(ValueError):
        NeighborSampler(np.array([[0, 1], [1, n]]), n)
    s = NeighborSampler(ei, n)
    with pytest.raises(ValueError):
        s.sample(np.array([ei.shape[1]]))


def test_column_store_assembles_the_batch_contract():
    """(node_tf, edge_index, edge_tf, y) with seed rows first — what main.py:48 receives."""
    ei, n = _graph(500, 8000, seed=5)
    num, cat, ts = S.edge_table(8000, seed=1)
    labels = torch.from_numpy((np.arange(8000) % 7 == 0).astype(np.int64))
    store = ColumnStore({stype.numerical: torch.from_numpy(num), stype.categorical: torch.from_numpy(cat),
                         stype.timestamp: torch.from_numpy(ts)}, S.EDGE_COLS,
                        {stype.relation: torch.ones(n, 1)}, S.NODE_COLS, labels)
    sampler = NeighborSampler(ei, n, num_neighbors=(20, 10))
    seeds = np.arange(100, 164, dtype=np.int64)
    node_tf, edge_index, edge_tf, y = store.graph_inputs(sampler, seeds, rng_seed=9)
    assert torch.equal(y, labels[100:164]) and edge_index.shape[0] == 2
    assert edge_tf.num_rows == edge_index.shape[1] and node_tf.num_rows == int(edge_index.max()) + 1
    assert torch.equal(edge_tf.feat_dict[stype.categorical][:64], torch.from_numpy(cat[100:164]))


def test_sampler_library_exports_every_declared_symbol():
    import ctypes, re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "include", "tabgnn_sampler.h")).read()
    names = sorted(set(re.findall(r"\b(tg_sampler_\w+)\s*\(", text)))
    assert len(names) >= 6
    from tabgnn_amd import sampler as S
    lib = ctypes.CDLL(S._LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/tabgnn_sampler.h but not exported"


def test_sharded_seed_loader_is_rank_disjoint_and_full_batches():
    from tabgnn_amd.sampler import ShardedSeedLoader
    ids = np.arange(1000, 2003)
    per_rank = [np.concatenate(list(ShardedSeedLoader(ids, 50, rank=r, world=4, seed=7))) for r in range(4)]
    assert all(len(p) == 250 for p in per_rank)                                  # 5 full batches of 50 each
    allv = np.concatenate(per_rank)
    assert len(np.unique(allv)) == 1000 and np.isin(allv, ids).all()             # disjoint across ranks
    again = np.concatenate(list(ShardedSeedLoader(ids, 50, rank=1, world=4, seed=7)))
    assert np.array_equal(again, per_rank[1])                                    # deterministic per (seed, epoch)
    ld = ShardedSeedLoader(ids, 50, rank=1, world=4, seed=7); ld.set_epoch(1)
    assert not np.array_equal(np.concatenate(list(ld)), per_rank[1])             # reshuffled each epoch


def test_lp_inputs_assembles_positive_and_negative_targets():
    """lp_inputs (batch_processing.py:104-147): targets = B positives then the negatives; attribute rows of the
    negatives are their positive's row repeated num_neg_samples times."""
    ei, n = _graph(500, 8000, seed=6)
    num, cat, ts = S.edge_table(8000, seed=2)
    store = ColumnStore({stype.numerical: torch.from_numpy(num), stype.categorical: torch.from_numpy(cat),
                         stype.timestamp: torch.from_numpy(ts)}, S.EDGE_COLS,
                        {stype.relation: torch.ones(n, 1)}, S.NODE_COLS, torch.zeros(8000, dtype=torch.long))
    sampler = NeighborSampler(ei, n, num_neighbors=(5, 5))
    seeds = np.arange(40, 56, dtype=np.int64)
    node_tf, edge_index, edge_tf, nei, ntf, tei, ttf = store.lp_inputs(sampler, seeds, num_neg_samples=6, rng_seed=3)
    B, k = 16, 6
    assert tei.shape == (2, B + B * k) and torch.equal(tei[:, :B], edge_index[:, :B])
    assert nei.shape[1] == edge_index.shape[1] - B and ntf.num_rows == nei.shape[1]
    assert ttf.num_rows == B + B * k
    c = ttf.feat_dict[stype.categorical]
    assert torch.equal(c[:B], torch.from_numpy(cat[40:56]))
    assert torch.equal(c[B:], torch.from_numpy(cat[40:56]).repeat_interleave(k, 0))
    negs = tei[:, B:].reshape(2, B, 2, k // 2)
    assert torch.equal(negs[0, :, 0, :], tei[0, :B, None].expand(B, k // 2))      # first half keeps the source
    assert torch.equal(negs[1, :, 1, :], tei[1, :B, None].expand(B, k // 2))      # second half the destination


def test_host_csr_is_a_stable_counting_sort():
    """tg_host_csr (the CSR the sampler hands to the aggregation kernels): rowptr = prefix sums of the key histogram,
    perm = input positions ascending inside every segment; bad keys are refused."""
    from tabgnn_amd.sampler import host_csr
    rs = np.random.RandomState(0)
    N, M = 50, 400
    keys = rs.randint(0, N - 5, M).astype(np.int64)                 # nodes N-5.. have no rows
    rowptr, perm = host_csr(keys, N)
    assert rowptr[0] == 0 and rowptr[-1] == M and (np.diff(rowptr) == np.bincount(keys, minlength=N)).all()
    assert (keys[perm[:M]] == np.sort(keys, kind="stable")).all()
    assert (perm[:M] == np.argsort(keys, kind="stable")).all()
    rp0, _ = host_csr(np.zeros(0, dtype=np.int64), 3)
    assert rp0.tolist() == [0, 0, 0, 0]
    with pytest.raises(ValueError):
        host_csr(np.array([0, 7]), 7)
