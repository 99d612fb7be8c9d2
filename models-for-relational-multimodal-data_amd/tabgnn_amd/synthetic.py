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
