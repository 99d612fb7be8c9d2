"""Oracle (test infrastructure): readout heads (``src/nn/gnn/decoder.py:5-32``).  Pinned by goldens."""
from __future__ import annotations

import torch
import torch.nn.functional as F


def _mlp(x, sd, pfx, p_drop, training):
    """decoder.py:14-15 / :28-29: Linear -> ReLU -> Drop -> Linear -> ReLU -> Drop -> Linear (mlp.0, mlp.3, mlp.6)."""
    x = torch.relu(x @ sd[pfx + "mlp.0.weight"].t() + sd[pfx + "mlp.0.bias"])
    x = F.dropout(x, p_drop, training)
    x = torch.relu(x @ sd[pfx + "mlp.3.weight"].t() + sd[pfx + "mlp.3.bias"])
    x = F.dropout(x, p_drop, training)
    return x @ sd[pfx + "mlp.6.weight"].t() + sd[pfx + "mlp.6.bias"]


def classifier_head(x, edge_index, edge_attr, sd, pfx="", p_drop=0.0, training=False):
    """``ClassifierHead.forward`` decoder.py:17-21: relu(x[src]), relu(x[dst]), edge_attr -> logits [B,n_classes]."""
    Fh = x.shape[1]
    h = x[edge_index.t()].reshape(-1, 2 * Fh).relu()
    h = torch.cat((h, edge_attr.view(-1, edge_attr.shape[1])), 1)
    return _mlp(h, sd, pfx, p_drop, training)


def node_classification_head(x, sd, pfx="", p_drop=0.0, training=False):
    """``NodeClassificationHead.forward`` decoder.py:31-32."""
    return _mlp(x, sd, pfx, p_drop, training)
