"""Host side of the shape-bucket design (tabgnn_amd/graph_step.py): bucket sizes, padding edges, and the index
structures of a padded batch — no GPU."""
import numpy as np
import pytest
import torch


def test_bucket_sizes_are_monotone_and_pad_at_most_an_eighth():
    from tabgnn_amd.graph_step import bucket_size
    prev = 0
    for n in list(range(1, 3000)) + [10_702, 438_354, 524_165, 5_078_345]:
        b = bucket_size(n)
        assert b >= n and b >= prev or n > 3000
        if n >= 64:
            assert b <= n * 1.125 + 1, (n, b)
            assert bucket_size(b) == b
        prev = b if n < 3000 else prev
    assert len({bucket_size(n) for n in range(9000, 12000)}) <= 4         # a sampled batch size's spread: few graphs


def test_padding_edges_are_self_loops_on_padding_nodes():
    from tabgnn_amd.graph_step import pad_edges
    ei = np.array([[0, 1, 2, 2], [1, 2, 0, 1]], dtype=np.int64)
    out = pad_edges(ei, 3, 9, 5)
    assert out.shape == (2, 9) and (out[:, :4] == ei).all()
    assert (out[0, 4:] == out[1, 4:]).all() and set(out[0, 4:]) == {3, 4}
    assert pad_edges(ei, 3, 4, 3) is not None
    with pytest.raises(ValueError):
        pad_edges(ei, 3, 6, 3)            # padding edges but no padding node
    with pytest.raises(ValueError):
        pad_edges(ei, 3, 3, 5)


def test_prepare_keeps_the_real_rows_and_their_csr():
    import tabgnn_amd as T
    from tabgnn_amd import synthetic as S, graph_step as G
    B = 16
    batch = S.make_batch(B, seed=5)
    E, N = batch[1].shape[1], batch[0].num_rows
    p = G.prepare(batch, B)
    e_pad, n_pad = p.key
    assert e_pad >= E and n_pad > N and (p.e_real, p.n_real) == (E, N)
    assert int(p.tensors["n_real"][0]) == N
    ei = p.tensors["ei"]
    assert torch.equal(ei[:, :E], batch[1]) and bool((ei[:, E:] >= N).all())
    for st, v in batch[2].feat_dict.items():
        got = p.tensors[f"edge.{st.value}"]
        assert got.shape[0] == e_pad and torch.equal(got[:E], v) and torch.equal(got[E:], v[:1].expand_as(got[E:]))
    # by-destination CSR of the padded neighbour graph: a real node's row is what the plain batch gives it
    off, flat = p.off, p.tensors["flat"].numpy()
    rp_d = flat[off[2]:off[3]]
    assert rp_d.shape[0] == n_pad + 1
    want = np.bincount(batch[1][1, B:].numpy(), minlength=N)
    assert (np.diff(rp_d)[:N] == want).all() and rp_d[-1] == e_pad - B
    # a second batch padded to the same key has the same part offsets (one static buffer serves the bucket)
    b2 = S.make_batch(B, seed=6)
    if b2[1].shape[1] <= e_pad and b2[0].num_rows < n_pad:
        p2 = G.prepare(b2, B, key=p.key)
        assert (p2.off == p.off).all() and p2.tensors["flat"].shape == p.tensors["flat"].shape


def test_prepare_sample_builds_a_lazy_bucket_batch_from_sampler_output():
    from tabgnn_amd import graph_step as G
    from tabgnn_amd.sampler import NeighborSampler
    rs = np.random.RandomState(3)
    N, E, B = 3000, 20000, 24
    ei = np.stack([rs.randint(0, N, E), rs.randint(0, N, E)])
    sampler = NeighborSampler(ei, N, (5, 3), num_threads=1)
    labels = torch.from_numpy((rs.rand(E) < 0.1).astype(np.int64))
    keys, layouts = set(), {}
    for i in range(6):
        eid, lei, nodes = sampler.sample(rs.choice(E, B, replace=False), i)
        p = G.prepare_sample(eid, lei, nodes, labels[eid[:B]], B)
        assert p.lazy and (p.e_real, p.n_real) == (eid.numel(), nodes.numel())
        e_pad, n_pad = p.key
        t = p.tensors
        assert t["edge.ids"].shape == (e_pad,) and t["node.ids"].shape == (n_pad,) and t["y"].shape == (B,)
        assert torch.equal(t["edge.ids"][:p.e_real], eid) and torch.equal(t["node.ids"][:p.n_real], nodes)
        assert bool((t["edge.ids"][p.e_real:] == eid[0]).all())            # padding rows repeat a valid id
        assert torch.equal(t["ei"][:, :p.e_real], lei) and int(t["n_real"][0]) == p.n_real
        assert all(off % 256 == 0 for _, _, _, off in p.layout)            # every part 256-byte aligned in the arena
        keys.add(p.key)
        layouts.setdefault(p.key, p.layout)
        assert layouts[p.key] == p.layout                                  # one layout per bucket: one static arena
    assert 1 <= len(keys) <= 4
