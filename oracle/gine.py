"""Oracle (test infrastructure): GINEConv / GINEConvHetero / GINe.

``GINEConv`` itself lives in ``torch_geometric==2.5.3`` (``nn/conv/gin_conv.py``; ``environment.yml:336``), which is
neither under ``/root/reference`` nor installed here — **parity unpinned** for that one class; this file restates its
published algorithm: ``forward(x, edge_index, edge_attr)``: ``x -> (x, x)`` if a tensor; ``out = sum_{j->i}
relu(x_j + lin(e_ji))`` (``aggr='add'``, ``message``); ``if x_r is not None: out = out + (1 + eps) * x_r``;
``return nn(out)``.  State-dict: ``nn.*``, ``lin.{weight,bias}`` (``Linear(edge_dim, nn[0].in_features)``) and the
``eps`` buffer (``train_eps=False``, initial 0).

``GINEConvHetero.forward`` (``src/nn/gnn/gine.py:27-35``, calls with ``(x, None)`` so there is NO self term, and both
directions share one network object, :18-19) and ``GINe.forward`` (``gine.py:82-94``) are the reference's own code,
pinned by ``tests/golden/gine_f32_l2*.npz`` generated from that file.
"""
from __future__ import annotations

import torch

from .pna import batch_norm
from .siblings import _edge_mlp


def gine_conv(x, edge_index, edge_attr, sd, pfx, self_term=True):
    src, dst = edge_index[0], edge_index[1]
    le = edge_attr @ sd[pfx + "lin.weight"].t() + sd[pfx + "lin.bias"]
    m = torch.relu(x[src] + le)
    out = torch.zeros_like(x).index_add_(0, dst, m)
    if self_term:
        out = out + (1 + sd[pfx + "eps"]) * x
    h = torch.relu(out @ sd[pfx + "nn.0.weight"].t() + sd[pfx + "nn.0.bias"])
    return h @ sd[pfx + "nn.2.weight"].t() + sd[pfx + "nn.2.bias"]


def gine_conv_hetero(x, edge_index, edge_attr, sd, pfx):
    a_in = gine_conv(x, edge_index, edge_attr, sd, pfx + "conv_forw.", self_term=False)
    a_out = gine_conv(x, edge_index.flipud(), edge_attr, sd, pfx + "conv_back.", self_term=False)
    return torch.cat([x, a_in, a_out], dim=1) @ sd[pfx + "lin.weight"].t() + sd[pfx + "lin.bias"]


def gine_forward(sd, x, edge_index, edge_attr, training=False, edge_updates=True):
    """gine.py:82-94: x [N, num_features], edge_attr [E, ...] -> (x [N,F], edge_attr [E,F])."""
    x = x @ sd["node_emb.weight"].t() + sd["node_emb.bias"]
    e = edge_attr.reshape(edge_attr.shape[0], -1) @ sd["edge_emb.weight"].t() + sd["edge_emb.bias"]
    i = 0
    while f"batch_norms.{i}.module.weight" in sd:
        pfx = f"convs.{i}."
        conv = (gine_conv_hetero(x, edge_index, e, sd, pfx) if (pfx + "conv_forw.lin.weight") in sd
                else gine_conv(x, edge_index, e, sd, pfx))
        x = (x + torch.relu(batch_norm(conv, sd, f"batch_norms.{i}.module.", training))) / 2
        if edge_updates:
            e = e + _edge_mlp(x, edge_index, e, sd, f"emlps.{i}.") / 2
        i += 1
    return x, e
