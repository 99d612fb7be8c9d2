"""Oracle (test infrastructure): ``nn.Module`` faces of the restated PyG pieces.

Used for two things only: (1) as the ``torch_geometric.nn.{PNAConv,BatchNorm,Linear}`` the
reference's own files resolve when ``tests/golden/make_golden.py`` imports them in this
container (torch_geometric is not installed), so the reference's
``fused.py`` / ``tabgnn.py`` / ``pna.py`` / ``decoder.py`` code runs unmodified around them;
(2) to produce state dicts with PyG 2.5.3's parameter names.  The arithmetic is
``oracle.pna`` — **parity unpinned** for these three classes (see ``oracle/pna.py``).
"""
from __future__ import annotations

import torch
from torch import nn

from . import gine as G
from . import pna as P


class Linear(nn.Linear):
    """``torch_geometric.nn.Linear`` with explicit in_channels: same parameters/initialisation as ``nn.Linear``."""

    def __init__(self, in_channels, out_channels, bias=True, **kw):
        super().__init__(in_channels, out_channels, bias=bias)


class _AggrModule(nn.Module):
    def __init__(self, deg):
        super().__init__()
        lin, log = P.avg_degree_stats(deg)
        self.register_buffer("avg_deg_lin", torch.tensor([lin]))
        self.register_buffer("avg_deg_log", torch.tensor([log]))


class PNAConv(nn.Module):
    def __init__(self, in_channels, out_channels, aggregators, scalers, deg, edge_dim=None, towers=1,
                 pre_layers=1, post_layers=1, divide_input=False, **kw):
        super().__init__()
        assert towers == 1 and pre_layers == 1 and post_layers == 1 and not divide_input
        assert sorted(aggregators) == sorted(P.AGGREGATORS) and tuple(scalers) == P.SCALERS
        self.aggregators = tuple(aggregators)
        assert edge_dim is not None and in_channels == out_channels
        F_ = in_channels
        self.aggr_module = _AggrModule(deg)
        self.edge_encoder = Linear(edge_dim, F_)
        self.pre_nns = nn.ModuleList([nn.Sequential(Linear(3 * F_, F_))])
        self.post_nns = nn.ModuleList([nn.Sequential(Linear(13 * F_, out_channels))])
        self.lin = Linear(out_channels, out_channels)

    def reset_parameters(self):
        for m in self.modules():
            if isinstance(m, nn.Linear):
                m.reset_parameters()

    def forward(self, x, edge_index, edge_attr=None):
        return P.pna_conv(x, edge_index, edge_attr, dict(self.state_dict(keep_vars=True)), "", self.aggregators)


class BatchNorm(nn.Module):
    def __init__(self, in_channels, eps=1e-5, momentum=0.1, **kw):
        super().__init__()
        self.module = nn.BatchNorm1d(in_channels, eps, momentum)

    def reset_parameters(self):
        self.module.reset_parameters()

    def forward(self, x):
        return self.module(x)


class GINEConv(nn.Module):
    """``torch_geometric.nn.GINEConv(nn, eps=0., train_eps=False, edge_dim=...)`` face over ``oracle.gine.gine_conv``."""

    def __init__(self, nn_, eps=0.0, train_eps=False, edge_dim=None, **kw):
        super().__init__()
        assert not train_eps and edge_dim is not None
        self.nn = nn_
        self.initial_eps = eps
        self.register_buffer("eps", torch.full((1,), float(eps)))
        first = nn_[0] if isinstance(nn_, nn.Sequential) else nn_
        self.lin = Linear(edge_dim, first.in_features)

    def reset_parameters(self):
        for m in self.modules():
            if isinstance(m, nn.Linear):
                m.reset_parameters()
        self.eps.fill_(self.initial_eps)

    def forward(self, x, edge_index, edge_attr=None):
        self_term = True
        if isinstance(x, tuple):
            x, x_r = x
            self_term = x_r is not None
        return G.gine_conv(x, edge_index, edge_attr, dict(self.state_dict(keep_vars=True)), "", self_term)
