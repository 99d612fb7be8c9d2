"""Readout heads (``src/nn/gnn/decoder.py:5-32``) with the reference's ``mlp.{0,3,6}`` parameter names."""
from __future__ import annotations

import torch
from torch import nn

from . import ops


def _mlp(n_in, n_classes, dropout):
    return nn.Sequential(nn.Linear(n_in, 50), nn.ReLU(), nn.Dropout(dropout), nn.Linear(50, 25), nn.ReLU(),
                         nn.Dropout(dropout), nn.Linear(25, n_classes))


def _run_mlp(mlp, h, p):
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
