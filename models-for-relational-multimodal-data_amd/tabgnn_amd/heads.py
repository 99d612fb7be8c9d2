"""Readout heads (``src/nn/gnn/decoder.py:5-32``) with the reference's ``mlp.{0,3,6}`` parameter names."""
from __future__ import annotations

import torch
from torch import nn

from . import ops


def _mlp(n_in, n_classes, dropout):
    return nn.Sequential(nn.Linear(n_in, 50), nn.ReLU(), nn.Dropout(dropout), nn.Linear(50, 25), nn.ReLU(),
                         nn.Dropout(dropout), nn.Linear(25, n_classes))


def _run_mlp(mlp, h, p):
    if ops.head_mlp_ok(mlp, h):                        # one forward / one backward kernel (csrc/head.hip)
        return ops.head_mlp(h, mlp, p)
    h = ops.act_dropout(ops.linear(h, mlp[0].weight, mlp[0].bias), "relu", p)
    h = ops.act_dropout(ops.linear(h, mlp[3].weight, mlp[3].bias), "relu", p)
    return ops.linear(h.float(), mlp[6].weight, mlp[6].bias)       # logits in fp32


class ClassifierHead(nn.Module):
    def __init__(self, n_classes=1, n_hidden=128, dropout=0.5, e_hidden=None):
        super().__init__()
        self.n_hidden = n_hidden
        self.e_hidden = n_hidden if e_hidden is None else e_hidden
        self.p = dropout
        self.mlp = _mlp(n_hidden * 2 + self.e_hidden, n_classes, dropout)

    def forward(self, x, edge_index, edge_attr):
        """[relu(x[src]), relu(x[dst]), edge_attr] -> logits [B, n_classes]   (decoder.py:17-21)"""
        seeds = ops.SeedIndex(edge_index, x.shape[0])
        h = ops.seed_gather(x, edge_attr.reshape(-1, edge_attr.shape[1]), seeds, "head")
        return _run_mlp(self.mlp, h, self.p if self.training else 0.0)


class NodeClassificationHead(nn.Module):
    def __init__(self, n_classes=1, n_hidden=128, dropout=0.5):
        super().__init__()
        self.n_hidden, self.p = n_hidden, dropout
        self.mlp = _mlp(n_hidden, n_classes, dropout)

    def forward(self, x):
        return _run_mlp(self.mlp, x, self.p if self.training else 0.0)


class LinkPredHead(nn.Module):
    """``LinkPredHead`` (``src/nn/gnn/decoder.py:34-72``): sigmoid link scores of positive and negative edges from
    [relu(x[src]), relu(x[dst]), edge_attr]; parameter names ``mlp.{0,3,6}`` as the reference."""

    def __init__(self, n_classes=1, n_hidden=128, dropout=0.5):
        super().__init__()
        self.n_hidden, self.n_classes, self.p = n_hidden, n_classes, dropout
        self.mlp = nn.Sequential(nn.Linear(n_hidden * 3, n_hidden), nn.ReLU(), nn.Dropout(dropout),
                                 nn.Linear(n_hidden, 25), nn.ReLU(), nn.Dropout(dropout), nn.Linear(25, n_classes))
        self.reset_parameters()

    def reset_parameters(self):
        for p in self.mlp.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)

    def _score(self, x, edge_index, edge_attr):
        seeds = ops.SeedIndex(edge_index, x.shape[0])
        h = ops.seed_gather(x, edge_attr.reshape(-1, edge_attr.shape[1]), seeds, "head")
        return torch.sigmoid(_run_mlp(self.mlp, h, self.p if self.training else 0.0))

    def forward(self, x, pos_edge_index, pos_edge_attr, neg_edge_index, neg_edge_attr):
        return self._score(x, pos_edge_index, pos_edge_attr), self._score(x, neg_edge_index, neg_edge_attr)


class MCMHead(nn.Module):
    """Masked-cell-modelling decoders (``src/nn/decoder/self_supervised.py:134-171``; ``SelfSupervisedHead`` :6-43 is
    the ``w=1`` case): per target LayerNorm -> ReLU -> Linear; keys ``num_decoder.{0,2}``, ``cat_decoder.{i}.{0,2}``."""

    def __init__(self, channels, num_numerical, num_categorical, w=1):
        super().__init__()
        d = w * channels
        mk = lambda n_out: nn.Sequential(nn.LayerNorm(d), nn.ReLU(), nn.Linear(d, n_out))
        self.num_decoder = mk(num_numerical)
        self.cat_decoder = nn.ModuleList([mk(c) for c in num_categorical])

    def reset_parameters(self):
        for seq in [self.num_decoder, *self.cat_decoder]:
            seq[0].reset_parameters(); seq[2].reset_parameters()

    @staticmethod
    def _decode(seq, x):
        h = ops.act_dropout(ops.layer_norm(x, seq[0].weight, seq[0].bias), "relu", 0.0)
        return ops.linear(h.float(), seq[2].weight, seq[2].bias)           # B-scale logits in fp32

    def forward(self, x):
        return self._decode(self.num_decoder, x), [self._decode(d, x) for d in self.cat_decoder]


class SelfSupervisedHead(MCMHead):
    def __init__(self, channels, num_numerical, num_categorical):
        super().__init__(channels, num_numerical, num_categorical, 1)
