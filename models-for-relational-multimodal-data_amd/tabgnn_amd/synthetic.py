"""Synthetic HI-Small-shaped mini-batches: the batch contract that
``get_graph_inputs`` produces (``src/datasets/ibm_transactions_for_aml.py:159-180``) — node TensorFrame,
``edge_index`` int64 [2,E] with the B seed edges first, edge TensorFrame (3 categorical (15,15,7), 1 numerical,
1 timestamp), labels for the seed edges — at the per-step shape of the reference's bundled real sample
(B=200 -> E=10 702, N=12 797; ``src/primitives/negative_sampling/edge_index.json``), scaled linearly with B.
"""
from __future__ import annotations

import numpy as np
import torch

from .frame import TensorFrame, stype

EDGES_PER_SEED = 10702 / 200.0
NODES_PER_SEED = 12797 / 200.0
GRAPH_NODES = 515_080    # HI-Small's node table: a sampled subgraph cannot hold more nodes than the graph
CARDS = (15, 7, 15)   # 'Payment Currency', 'Payment Format', 'Receiving Currency' (sorted names)

EDGE_COLS = {stype.numerical: ["Amount Paid"],
             stype.categorical: ["Payment Currency", "Payment Format", "Receiving Currency"],
             stype.timestamp: ["Timestamp"]}
NODE_COLS = {stype.relation: ["node_attr"]}
EDGE_STATS = {"Amount Paid": dict(mean=0.5, std=0.29), "Payment Currency": dict(cardinality=15),
              "Payment Format": dict(cardinality=7), "Receiving Currency": dict(cardinality=15),
              "Timestamp": dict(min_year=2022)}


def _zipf_choice(rs, n, size, alpha=1.2):
    w = 1.0 / np.arange(1, n + 1) ** alpha
    return rs.choice(n, size=size, p=w / w.sum())


def sampled_subgraph(batch_size, seed=0):
    """edge_index [2,E]: every node id in [0,N) present; heavy-tailed sources (a hub holds ~3-4 % of the edges)
    and mildly skewed destinations, calibrated on the reference's bundled batch (in-degree max 14, out-degree
    max 345, 43 % of nodes without in-edges at B=200)."""
    rs = np.random.RandomState(seed)
    E = int(round(EDGES_PER_SEED * batch_size))
    N = min(int(round(NODES_PER_SEED * batch_size)), 2 * E, GRAPH_NODES)
    flat = np.empty(2 * E, dtype=np.int64)
    cover = rs.permutation(2 * E)[:N]
    flat[cover] = rs.permutation(N)
    free = np.ones(2 * E, dtype=bool)
    free[cover] = False
    order = rs.permutation(N)
    rest_src = np.nonzero(free[:E])[0]
    rest_dst = np.nonzero(free[E:])[0] + E
    flat[rest_src] = order[_zipf_choice(rs, N, rest_src.size, 1.0)]
    flat[rest_dst] = order[::-1][_zipf_choice(rs, N, rest_dst.size, 0.5)]
    return flat.reshape(2, E), N


def edge_table(E, seed=0):
    rs = np.random.RandomState(seed + 1)
    num = rs.rand(E, 1).astype(np.float32)
    cat = np.stack([_zipf_choice(rs, c, E) for c in CARDS], axis=1).astype(np.int64)
    t = rs.randint(0, 18 * 86400, size=E)
    day = t // 86400
    ts = np.stack([np.full(E, 2022), np.full(E, 8), day, (day + 3) % 7, (t // 3600) % 24, (t // 60) % 60, t % 60],
                  axis=1).astype(np.int64).reshape(E, 1, 7)
    return num, cat, ts


def make_batch(batch_size, seed=0, device="cpu", p_pos=0.05):
    ei, N = sampled_subgraph(batch_size, seed)
    E = ei.shape[1]
    num, cat, ts = edge_table(E, seed)
    rs = np.random.RandomState(seed + 2)
    y = (rs.rand(batch_size) < p_pos).astype(np.int64)
    edge_tf = TensorFrame({stype.numerical: torch.from_numpy(num), stype.categorical: torch.from_numpy(cat),
                           stype.timestamp: torch.from_numpy(ts)}, EDGE_COLS)
    node_tf = TensorFrame({stype.relation: torch.ones(N, 1)}, NODE_COLS)
    return (node_tf.to(device), torch.from_numpy(ei).to(device), edge_tf.to(device), torch.from_numpy(y).to(device))


def in_degrees_like(batch_size=2000, seed=123):
    """In-degree sample standing in for the train graph's (main.py:283-286) to size the PNA scalers."""
    ei, N = sampled_subgraph(batch_size, seed)
    return torch.from_numpy(np.bincount(ei[1], minlength=N))


def make_config(n_hidden=128, n_layers=2, nhead=4, batch_size=200, backbone_dropout=0.5, head_dropout=0.083,
                compute_dtype=torch.float32):
    """The ``config`` dict of main.py:161-190 for ``--model tabgnnfused --task edge_classification``."""
    from .encoders import StypeWiseFeatureEncoder
    return dict(model="tabgnnfused", task="edge_classification", batch_size=batch_size, n_hidden=n_hidden,
                n_gnn_layers=n_layers, n_classes=2, dropout=head_dropout, backbone_dropout=backbone_dropout,
                nhead=nhead, num_node_features=1, num_edge_features=5, in_degrees=in_degrees_like(),
                reverse_mp=False, load_model=None, checkpoint=False, loss_weights=[1.0, 9.23], lr=0.0006116418,
                node_encoder=StypeWiseFeatureEncoder(n_hidden, {}, NODE_COLS, compute_dtype),
                edge_encoder=StypeWiseFeatureEncoder(n_hidden, EDGE_STATS, EDGE_COLS, compute_dtype))


# ---- BASELINE configs[4]: synthetic 10 M-node / 100 M-edge table + graph, 64 mixed stype columns -----------
WIDE_CARDS_SEED = 5


def wide64_columns():
    """(names, stats, cardinalities) of the 64-column edge table: 32 categorical (cardinalities log-uniform 2 .. 10^4),
    24 numerical, 8 timestamp."""
    rs = np.random.RandomState(WIDE_CARDS_SEED)
    cards = [int(c) for c in np.exp(rs.uniform(np.log(2), np.log(1e4), 32))]
    names = {stype.numerical: [f"n{i}" for i in range(24)], stype.categorical: [f"c{i}" for i in range(32)],
             stype.timestamp: [f"t{i}" for i in range(8)]}
    stats = {**{f"n{i}": dict(mean=0.0, std=1.0) for i in range(24)},
             **{f"c{i}": dict(cardinality=cards[i]) for i in range(32)}, **{f"t{i}": dict(min_year=2015) for i in range(8)}}
    return names, stats, cards


def wide64_config(batch_size, in_degrees, compute_dtype, n_hidden=256):
    from .encoders import StypeWiseFeatureEncoder
    names, stats, _ = wide64_columns()
    return dict(model="tabgnnfused", task="edge_classification", batch_size=batch_size, n_hidden=n_hidden, n_gnn_layers=2,
                n_classes=2, dropout=0.083, backbone_dropout=0.5, nhead=8, num_node_features=1, num_edge_features=64,
                in_degrees=in_degrees, reverse_mp=False, load_model=None, checkpoint=False,
                node_encoder=StypeWiseFeatureEncoder(n_hidden, {}, NODE_COLS, compute_dtype),
                edge_encoder=StypeWiseFeatureEncoder(n_hidden, stats, names, compute_dtype))


def powerlaw_graph_on_device(num_nodes, num_edges, device, seed=11):
    """edge_index int64 [2, E] generated ON the GPU: uniform sources, destinations ~ N * u^2 (density ~ x^-1/2: node 0
    collects ~E / sqrt(N) in-edges, the median node E / N / 2)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    ei = torch.empty(2, num_edges, dtype=torch.int64, device=device)
    chunk = 1 << 24
    for a in range(0, num_edges, chunk):
        b = min(a + chunk, num_edges)
        ei[0, a:b] = torch.randint(0, num_nodes, (b - a,), device=device, generator=g)
        u = torch.rand(b - a, device=device, generator=g)
        ei[1, a:b] = (u * u * num_nodes).long().clamp_(max=num_nodes - 1)
    return ei


def wide64_store_on_device(num_nodes, num_edges, device, seed=12, p_pos=0.05):
    """The raw 64-column edge table (800 B per row: 32 int64 categories, 24 fp32 values, 8 x 7 int64 calendar fields) and
    its labels generated ON the GPU in chunks, as a ``ColumnStore`` (the table stays resident in HBM; a batch is a list
    of row ids).  100 M rows = 80 GB."""
    from .sampler import ColumnStore
    names, _, cards = wide64_columns()
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    E = int(num_edges)
    cat = torch.empty(E, 32, dtype=torch.int64, device=device)
    num = torch.empty(E, 24, dtype=torch.float32, device=device)
    ts = torch.empty(E, 8, 7, dtype=torch.int64, device=device)
    labels = torch.empty(E, dtype=torch.int64, device=device)
    hi = (2024, 12, 28, 7, 24, 60, 60)
    chunk = 1 << 22
    for a in range(0, E, chunk):
        b = min(a + chunk, E)
        n = b - a
        for j, c in enumerate(cards):
            cat[a:b, j] = torch.randint(0, c, (n,), device=device, generator=g)
        num[a:b] = torch.randn(n, 24, device=device, generator=g)
        for k in range(7):
            ts[a:b, :, k] = torch.randint(2015 if k == 0 else 0, hi[k], (n, 8), device=device, generator=g)
        labels[a:b] = (torch.rand(n, device=device, generator=g) < p_pos).long()
    return ColumnStore({stype.numerical: num, stype.categorical: cat, stype.timestamp: ts}, names,
                       {stype.relation: torch.ones(int(num_nodes), 1, device=device)}, NODE_COLS, labels)
