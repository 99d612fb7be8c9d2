"""Oracle (test infrastructure): the sequential (non-fused) TABGNN backbone, BASELINE config 4.

Restates ``TABGNN.forward`` (``src/nn/models/tabgnn.py:100-151``), ``PNALayer.forward``
(``:187-191``) and ``FTTransformerLayer.forward`` (``:218-219``).  Pinned by goldens
generated from the reference file.
"""
from __future__ import annotations

import torch

from .fused_path import prepend_cls
from .pna import batch_norm, gnn_conv
from .transformer import encoder_layer, layer_norm


def ft_layer(x, sd, pfx, nhead, p_drop, training):
    """tabgnn.py:218-219: (x + LN(enc(x))) / 2."""
    t = encoder_layer(x, sd, pfx + "tab_conv.", nhead, p_drop, training)
    return (x + layer_norm(t, sd[pfx + "tab_norm.weight"], sd[pfx + "tab_norm.bias"])) / 2


def pna_layer(x, edge_index, edge_attr, sd, pfx, training):
    """tabgnn.py:187-191 (note ``e + MLP/2``, not ``(e+MLP)/2``)."""
    conv = gnn_conv(x, edge_index, edge_attr, sd, pfx + "gnn_conv.")
    x = (x + torch.relu(batch_norm(conv, sd, pfx + "gnn_norm.module.", training))) / 2
    src, dst = edge_index
    m = torch.cat([x[src], x[dst], edge_attr], dim=-1)
    m = torch.relu(m @ sd[pfx + "gnn_edge_update.0.weight"].t() + sd[pfx + "gnn_edge_update.0.bias"])
    m = m @ sd[pfx + "gnn_edge_update.2.weight"].t() + sd[pfx + "gnn_edge_update.2.bias"]
    return x, edge_attr + m / 2


def tabgnn_forward(sd, nhead, x, edge_index, edge_attr, p_drop=0.0, training=False):
    """x [V, n_node_cols, C], edge_attr [E, n_edge_cols, C] -> (x [V,F], edge_attr [E,F])."""
    cls = sd["cls_embedding"]
    x = prepend_cls(cls, x)
    e = prepend_cls(cls, edge_attr)
    tx, te = x, e
    L = 0
    while f"tabular_backbone.{L}.tab_norm.weight" in sd:
        pfx = f"tabular_backbone.{L}."
        tx = ft_layer(tx, sd, pfx, nhead, p_drop, training)     # the SAME layer serves nodes and edges (:127-129)
        te = ft_layer(te, sd, pfx, nhead, p_drop, training)
        L += 1
    x = (x + tx) / 2
    e = (e + te) / 2
    x = x.reshape(x.shape[0], -1) @ sd["node_emb.weight"].t() + sd["node_emb.bias"]
    e = e.reshape(e.shape[0], -1) @ sd["edge_emb.weight"].t() + sd["edge_emb.bias"]
    for i in range(L):
        x, e = pna_layer(x, edge_index, e, sd, f"gnn_backbone.{i}.", training)
    return x, e
