"""Oracle (test infrastructure): sibling backbones built from the same ops.

Restates ``TABGNNInterleaved.forward`` / ``FTTransformerPNAInterleavedLayer.forward``
(``src/nn/models/inteleaved.py:140-163,216-227``) and ``PNAS.forward`` (``src/nn/gnn/pna.py:48-97``; note its
aggregator order ['mean','min','max','std'], :59).  Pinned by ``tests/golden/interleaved_c32_h4_l2.npz`` and
``tests/golden/pnas_f32_l2*.npz`` generated from the reference's own files.
"""
from __future__ import annotations

import torch

from .fused_path import prepend_cls
from .pna import batch_norm, gnn_conv
from .transformer import encoder_layer, layer_norm

PNAS_AGGREGATORS = ("mean", "min", "max", "std")


def _edge_mlp(x, edge_index, e, sd, pfx):
    src, dst = edge_index
    m = torch.cat([x[src], x[dst], e], dim=-1)
    m = torch.relu(m @ sd[pfx + "0.weight"].t() + sd[pfx + "0.bias"])
    return m @ sd[pfx + "2.weight"].t() + sd[pfx + "2.bias"]


def interleaved_layer(x_gnn, edge_index, edge_attr, sd, pfx, nhead, p_drop, training):
    """inteleaved.py:216-227: column attention on EVERY edge row each layer; the CLS token is the edge embedding the
    PNA layer consumes and updates."""
    t = encoder_layer(edge_attr, sd, pfx + "tab_conv.", nhead, p_drop, training)
    edge_attr = edge_attr + layer_norm(t, sd[pfx + "tab_norm.weight"], sd[pfx + "tab_norm.bias"]) / 2   # sic (:217)
    cls, feat = edge_attr[:, 0, :], edge_attr[:, 1:, :]
    conv = gnn_conv(x_gnn, edge_index, cls, sd, pfx + "gnn_conv.")
    x_gnn = (x_gnn + torch.relu(batch_norm(conv, sd, pfx + "gnn_norm.module.", training))) / 2
    cls = (cls + _edge_mlp(x_gnn, edge_index, cls, sd, pfx + "gnn_edge_update.")) / 2
    return x_gnn, torch.cat([cls.unsqueeze(1), feat], dim=1)


def interleaved_forward(sd, nhead, x, edge_index, edge_attr, p_drop=0.0, training=False):
    """x [N, n_node_feats, C], edge_attr [E, ncols, C] -> (x_gnn [N,F], x_edge [E,C])  (inteleaved.py:140-163)."""
    node_dim = sd["node_emb.weight"].shape[1]
    x_gnn = x.reshape(-1, node_dim) @ sd["node_emb.weight"].t() + sd["node_emb.bias"]
    e = prepend_cls(sd["cls_embedding"], edge_attr)
    t = encoder_layer(e, sd, "tab_conv.", nhead, p_drop, training)
    e = (e + layer_norm(t, sd["tab_norm.weight"], sd["tab_norm.bias"])) / 2
    cur, i = e, 0
    while f"backbone.{i}.tab_norm.weight" in sd:
        x_gnn, cur = interleaved_layer(x_gnn, edge_index, cur, sd, f"backbone.{i}.", nhead, p_drop, training)
        i += 1
    e = (cur + e) / 2
    return x_gnn, e[:, 0, :]


def pnas_forward(sd, x, edge_index, edge_attr, training=False, edge_updates=True):
    """pna.py:88-97: x [N, ...] -> node_emb, edge_attr [E, ...] -> edge_emb, L x {(x + relu(BN(PNA)))/2; e + MLP/2}."""
    x = x.reshape(x.shape[0], -1) @ sd["node_emb.weight"].t() + sd["node_emb.bias"]
    e = edge_attr.reshape(edge_attr.shape[0], -1) @ sd["edge_emb.weight"].t() + sd["edge_emb.bias"]
    i = 0
    while f"batch_norms.{i}.module.weight" in sd:
        conv = gnn_conv(x, edge_index, e, sd, f"convs.{i}.", PNAS_AGGREGATORS)
        x = (x + torch.relu(batch_norm(conv, sd, f"batch_norms.{i}.module.", training))) / 2
        if edge_updates:
            e = e + _edge_mlp(x, edge_index, e, sd, f"emlps.{i}.") / 2
        i += 1
    return x, e


def cpna_forward(sd, x, edge_index, edge_attr, training=False, edge_updates=True):
    """``CPNA.forward`` (src/nn/gnn/pna.py:205-219): one stack of L PNA layers PER edge-table column, all sharing the
    evolving node state x; column c's embeddings are updated by its own edge MLPs and written back.
    edge_attr [E, ncols, F] -> (x [N,F], edge_attr [E, ncols, F])."""
    x = x.reshape(x.shape[0], -1) @ sd["node_emb.weight"].t() + sd["node_emb.bias"]
    cols = []
    c = 0
    while f"col_batch_norms.{c}.0.module.weight" in sd:
        col = edge_attr[:, c, :]
        i = 0
        while f"col_batch_norms.{c}.{i}.module.weight" in sd:
            conv = gnn_conv(x, edge_index, col, sd, f"col_convs.{c}.{i}.", PNAS_AGGREGATORS)
            x = (x + torch.relu(batch_norm(conv, sd, f"col_batch_norms.{c}.{i}.module.", training))) / 2
            if edge_updates:
                col = col + _edge_mlp(x, edge_index, col, sd, f"col_emlps.{c}.{i}.") / 2
            i += 1
        cols.append(col)
        c += 1
    return x, torch.stack(cols, dim=1)
