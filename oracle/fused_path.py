"""Oracle (test infrastructure): the fused FT-Transformer + PNA backbone.

Functional restatement of ``TABGNNFused.forward`` (``src/nn/models/fused.py:144-175``)
and ``FTTransformerPNAFusedLayer.forward`` (``src/nn/models/fused.py:248-269``) over a
flat state dict carrying the reference's parameter names.  Pinned by golden vectors
produced from the reference file itself (``tests/golden/make_golden.py``).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from .pna import batch_norm, gnn_conv
from .transformer import encoder_layer, layer_norm


def prepend_cls(cls, rows):
    """fused.py:158-159,162-163: CLS token as column 0 of every row."""
    return torch.cat([cls.view(1, 1, -1).expand(rows.shape[0], 1, -1), rows], dim=1)


def pool_seed_nodes(x_gnn, target_edge_index, x, C, Fh):
    """fused.py:261-268: mean of the fused src/dst embeddings over duplicate seed endpoints,
    averaged into ``x_gnn`` at the unique endpoint ids (in place in the reference)."""
    index = target_edge_index.flatten()
    emb = torch.cat([x[:, C:C + Fh], x[:, C + Fh:]], dim=0)
    uniq, inv = torch.unique(index, return_inverse=True)
    summed = torch.zeros(uniq.numel(), emb.shape[1], dtype=torch.float).index_add_(0, inv, emb)
    counts = torch.bincount(inv)
    pooled = summed / counts.unsqueeze(1).float()
    x_gnn = x_gnn.clone()
    x_gnn[uniq] = (x_gnn[uniq] + pooled) / 2
    return x_gnn


def fuse_mlp(x, sd, pfx, p_drop, training):
    """fused.py:224-231: LN -> Linear(D,4D) -> LeakyReLU -> Drop -> Linear(4D,4D) -> LeakyReLU -> Drop -> Linear(4D,D)."""
    h = layer_norm(x, sd[pfx + "0.weight"], sd[pfx + "0.bias"])
    h = F.leaky_relu(h @ sd[pfx + "1.weight"].t() + sd[pfx + "1.bias"])
    h = F.dropout(h, p_drop, training)
    h = F.leaky_relu(h @ sd[pfx + "4.weight"].t() + sd[pfx + "4.bias"])
    h = F.dropout(h, p_drop, training)
    return h @ sd[pfx + "7.weight"].t() + sd[pfx + "7.bias"]


def fused_layer(x_tab, x_gnn, edge_index, edge_attr, target_edge_index, sd, pfx, nhead,
                p_drop=0.0, training=False, lp=False):
    """fused.py:248-269."""
    C = x_tab.shape[-1]
    Fh = x_gnn.shape[-1]
    t = encoder_layer(x_tab, sd, pfx + "tab_conv.", nhead, p_drop, training)
    x_tab = x_tab + layer_norm(t, sd[pfx + "tab_norm.weight"], sd[pfx + "tab_norm.bias"]) / 2   # sic, :249
    cls_tab, feat_tab = x_tab[:, 0, :], x_tab[:, 1:, :]

    conv = gnn_conv(x_gnn, edge_index, edge_attr, sd, pfx + "gnn_conv.")
    x_gnn = (x_gnn + torch.relu(batch_norm(conv, sd, pfx + "gnn_norm.module.", training))) / 2   # :252
    src, dst = edge_index
    m = torch.cat([x_gnn[src], x_gnn[dst], edge_attr], dim=-1)
    m = torch.relu(m @ sd[pfx + "gnn_edge_update.0.weight"].t() + sd[pfx + "gnn_edge_update.0.bias"])
    m = m @ sd[pfx + "gnn_edge_update.2.weight"].t() + sd[pfx + "gnn_edge_update.2.bias"]
    edge_attr = (edge_attr + m) / 2                                                              # :254

    if not lp:
        x = torch.cat([cls_tab, x_gnn[target_edge_index[0]], x_gnn[target_edge_index[1]]], dim=-1)
        f = fuse_mlp(x, sd, pfx + "fuse.", p_drop, training)
        x = (x + layer_norm(f, sd[pfx + "fuse_norm.weight"], sd[pfx + "fuse_norm.bias"])) / 2     # :258
        cls_tab = (cls_tab + x[:, :C]) / 2
        x_tab = torch.cat([cls_tab.unsqueeze(1), feat_tab], dim=1)
        x_gnn = pool_seed_nodes(x_gnn, target_edge_index, x, C, Fh)
    return x_tab, x_gnn, edge_attr


def num_layers_of(sd, stem="backbone."):
    n = 0
    while f"{stem}{n}.tab_norm.weight" in sd or f"{stem}{n}.gnn_norm.module.weight" in sd:
        n += 1
    return n


def fused_forward(sd, nhead, x, edge_index, edge_attr, target_edge_index, target_edge_attr,
                  lp=False, p_drop=0.0, training=False):
    """``TABGNNFused.forward`` (fused.py:144-175).

    x [N, n_node_feats, C]; edge_index int64 [2,E_n]; edge_attr [E_n, ncols, C];
    target_edge_index int64 [2,B]; target_edge_attr [B, ncols, C]
    -> (x_gnn [N,F], edge_attr [E_n,F], target_edge_attr [B,F]).
    """
    cls = sd["cls_embedding"]
    node_dim = sd["node_emb.weight"].shape[1]
    edge_dim = sd["edge_emb.weight"].shape[1]
    ln_w, ln_b = sd["tab_norm.weight"], sd["tab_norm.bias"]

    x_gnn = x.reshape(-1, node_dim) @ sd["node_emb.weight"].t() + sd["node_emb.bias"]

    t = prepend_cls(cls, target_edge_attr)
    t = layer_norm(encoder_layer(t, sd, "tab_conv.", nhead, p_drop, training), ln_w, ln_b)        # :160 (no residual)

    e = prepend_cls(cls, edge_attr)
    e = (e + layer_norm(encoder_layer(e, sd, "tab_conv.", nhead, p_drop, training), ln_w, ln_b)) / 2
    e = e.reshape(-1, edge_dim) @ sd["edge_emb.weight"].t() + sd["edge_emb.bias"]                  # :165-166

    x_tab = t
    for i in range(num_layers_of(sd)):
        x_tab, x_gnn, e = fused_layer(x_tab, x_gnn, edge_index, e, target_edge_index, sd,
                                      f"backbone.{i}.", nhead, p_drop, training, lp)

    t_out = ((x_tab + t) / 2).reshape(-1, edge_dim)
    t_out = t_out @ sd["edge_emb.weight"].t() + sd["edge_emb.bias"]                                # :172-174
    return x_gnn, e, t_out
