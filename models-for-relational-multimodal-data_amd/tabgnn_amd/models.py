"""Backbones with the reference's constructor arguments, forward signatures and state-dict names.

* ``TABGNNFused`` / ``FTTransformerPNAFusedLayer`` — ``src/nn/models/fused.py:31-269``
* ``TABGNN`` / ``PNALayer`` / ``FTTransformerLayer`` — ``src/nn/models/tabgnn.py:27-219``
Drop-in: ``from tabgnn_amd import TABGNNFused`` replaces ``from src.nn.models import TABGNNFused``
(``utils.py:16``, ``fused.py:20``); ``load_state_dict`` accepts the reference's checkpoints.
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import nn

from . import ops
from .layers import BatchNorm, ColumnTransformerLayer, GINEConv, GINEConvHetero, PNAConv, PNAConvHetero

_AGGR = ["mean", "max", "min", "std"]
_SCAL = ["identity", "amplification", "attenuation"]


def _make_conv(nhidden, deg, reverse_mp):
    kw = dict(in_channels=nhidden, out_channels=nhidden, aggregators=_AGGR, scalers=_SCAL, deg=deg, edge_dim=nhidden,
              towers=1, pre_layers=1, post_layers=1, divide_input=False)
    return PNAConvHetero(n_hidden=nhidden, **kw) if reverse_mp else PNAConv(**kw)


def _xavier_matrices(module):
    for p in module.parameters():
        if p.dim() > 1:
            nn.init.xavier_uniform_(p)


class _PrependCLS(torch.autograd.Function):
    """[cls | row] per table row (fused.py:158-159,162-163).  Rows that come from the stype encoder sit in columns 1..
    of a [R, ncols+1, C] buffer (``encoders._Encode``): the CLS vector is then written into column 0 and the buffer
    itself is the result — no copy; otherwise one gather-concat launch builds the tensor."""

    @staticmethod
    def forward(ctx, cls, rows):
        R, ncols, C = rows.shape
        ctx.C = C
        ctx.cls = cls if isinstance(cls, nn.Parameter) else None
        base = getattr(rows, "_cls_base", None)
        if (base is not None and base.shape == (R, ncols + 1, C) and base.is_contiguous() and base.dtype == rows.dtype
                and rows.data_ptr() == base.data_ptr() + C * rows.element_size()):
            base[:, 0, :] = cls.detach().to(rows.dtype)
            rows._cls_base = None                      # the slot is taken: a second prepend of the same rows must copy
            return base
        rows = rows.contiguous()
        c = cls.detach().to(rows.dtype).contiguous().view(1, C)
        flat = rows.view(R, ncols * C)
        out = torch.empty(R, (ncols + 1) * C, dtype=rows.dtype, device=rows.device)
        from . import _lib as L
        L.call("tg_gather_concat3", c.data_ptr(), None, 0, C, 0, flat.data_ptr(), None, ncols * C, ncols * C, 0,
               flat.data_ptr(), None, ncols * C, 0, L.ptr(out), R, L.dt(out), L.stream())
        return out.view(R, ncols + 1, C)

    @staticmethod
    def backward(ctx, g):
        R, S, C = g.shape
        if g.is_cuda and g.stride(2) == 1 and g.stride(1) == C and C % 8 == 0 and g.dtype in (torch.bfloat16, torch.float32):
            from . import _lib as L
            tgt = ops._grad_target(ctx.cls)          # the parameter's own fp32 gradient buffer: added in place
            if tgt is not None and tgt.shape != (C,):
                tgt = None
            dcls = tgt if tgt is not None else torch.empty(C, dtype=torch.float32, device=g.device)
            ws = ops._workspace(L.load().tg_col_sum_workspace_floats(R, C), g.device)
            L.call("tg_col_sum", L.ptr(g), R, C, g.stride(0), L.ptr(dcls), L.ptr(ws), 1 if tgt is not None else 0, L.dt(g),
                   L.stream())
            if tgt is not None:
                dcls = None
        else:
            dcls = g[:, 0, :].sum(0, dtype=torch.float32)
        return dcls, g[:, 1:, :]           # a view: the encoder's backward reads it in place (no copy of [R,ncols,C])


def prepend_cls(cls, rows):
    return _PrependCLS.apply(cls, rows)


class FTTransformerPNAFusedLayer(nn.Module):
    def __init__(self, channels: int, nhead: int, feedforward_channels: Optional[int] = None, dropout: float = 0.5,
                 activation: str = "relu", nhidden: int = 128, deg=None, reverse_mp: bool = False) -> None:
        super().__init__()
        self.channels, self.nhidden, self.p = channels, nhidden, dropout
        D = channels + 2 * nhidden
        self.tab_conv = ColumnTransformerLayer(channels, nhead, feedforward_channels, dropout, activation)
        self.tab_norm = nn.LayerNorm(channels)
        self.gnn_conv = _make_conv(nhidden, deg, reverse_mp)
        self.gnn_norm = BatchNorm(nhidden)
        self.gnn_edge_update = nn.Sequential(nn.Linear(3 * nhidden, nhidden), nn.ReLU(), nn.Linear(nhidden, nhidden))
        self.fuse = nn.Sequential(nn.LayerNorm(D), nn.Linear(D, 4 * D), nn.LeakyReLU(), nn.Dropout(dropout),
                                  nn.Linear(4 * D, 4 * D), nn.LeakyReLU(), nn.Dropout(dropout), nn.Linear(4 * D, D))
        self.fuse_norm = nn.LayerNorm(D)
        self.reset_parameters()

    def reset_parameters(self):
        _xavier_matrices(self.tab_conv)
        self.tab_norm.reset_parameters()
        self.gnn_conv.reset_parameters()
        self.gnn_norm.reset_parameters()
        _xavier_matrices(self.gnn_edge_update)
        _xavier_matrices(self.fuse)
        self.fuse_norm.reset_parameters()

    def forward(self, x_tab, x_gnn, edge_index, edge_attr, target_edge_index, lp=False):
        N = x_gnn.shape[0]
        g = ops.SubgraphIndex.build(edge_index, N)
        p = self.p if self.training else 0.0
        # x_tab + LN(enc(x_tab)) / 2   (sic, fused.py:249)
        x_tab = self.tab_conv(x_tab, self.tab_norm, 1.0, 0.5)
        # Shared gradient buffers (ops.GradSink) of the three tensors that fan out inside the layer — the layer input
        # x (message gather, post projection, residual), the edge embedding (two gathers, the edge update) and the new
        # x (edge-update gather, fuse gather, seed pooling): every consumer adds its part in its own backward kernel,
        # autograd never runs a [N,F] / [E,F] accumulation pass.  Only when ALL their consumers are sink-aware.
        from .layers import PNAConv
        sinks = (isinstance(self.gnn_conv, PNAConv) and x_gnn.is_contiguous() and edge_attr.is_contiguous()
                 and self.gnn_conv.sinks_ok(x_gnn) and torch.is_grad_enabled())
        sx, se, sn = ops.new_sink(sinks), ops.new_sink(sinks), ops.new_sink(sinks and not lp)
        # (x_gnn + relu(BN(PNA))) / 2   (fused.py:252)
        conv = self.gnn_conv(x_gnn, g, edge_attr, sx, se) if sx is not None else self.gnn_conv(x_gnn, g, edge_attr)
        x_gnn = self.gnn_norm(conv, res=x_gnn, relu=True, alpha=0.5, beta_c=0.5, sink_res=sx)
        # (e + MLP([x[src], x[dst], e])) / 2   (fused.py:253-254)
        reread = (ops.edge_mlp_rereads_x(x_gnn, edge_attr, self.gnn_edge_update[0], self.gnn_edge_update[2])
                  and torch.is_grad_enabled())
        upd = ops.edge_mlp_relu(x_gnn, edge_attr, g, "src", self.gnn_edge_update[0], self.gnn_edge_update[2], sn, se)
        edge_attr = ops.axpby(edge_attr, upd, 0.5, 0.5, sink_a=se)
        if not lp:
            seeds = ops.SeedIndex(target_edge_index, N)
            f = self.fuse
            xf0 = ops.seed_gather(x_gnn, x_tab, seeds, "fuse", sn)                  # [cls_tab, x[src], x[dst]]
            h = ops.layer_norm(xf0, f[0].weight, f[0].bias)
            h = ops.mlp_chain(h, (f[1], f[4], f[7]), "leaky_relu", p)
            xf = ops.layer_norm(h, self.fuse_norm.weight, self.fuse_norm.bias, res=xf0, alpha=0.5, beta_c=0.5)
            x_tab = ops.cls_merge(x_tab, xf)                                        # fused.py:259-260
            # in place, as fused.py:268.  The gather-fused edge update above re-reads x_gnn in its backward (x[src], x[dst]
            # for dW instead of a saved [E,384] concatenation): the pool stashes the <= 2B rows it overwrites and that
            # backward puts them back first (ops._MLPReluGather); TABGNN_NO_POOL_RESTORE=1 pools into a copy instead.
            # (``reread`` = edge_mlp_relu's own route predicate, taken above on the tensors it saw.)
            x_gnn = ops.seed_pool(x_gnn, xf, seeds, self.channels, inplace=not reread or ops.POOL_RESTORE, sink_x=sn,
                                  stash=reread and ops.POOL_RESTORE)                 # fused.py:261-268
        return x_tab, x_gnn, edge_attr


class TABGNNFused(nn.Module):
    def __init__(self, channels: int, num_layers: int, encoder=None, deg=None, node_dim: int = 1, nhidden: int = 128,
                 edge_dim: int = None, reverse_mp: bool = False, feedforward_channels: Optional[int] = None,
                 nhead: int = 8, dropout: float = 0.5, activation: str = "relu") -> None:
        super().__init__()
        if num_layers <= 0:
            raise ValueError(f"num_layers must be a positive integer (got {num_layers})")
        self.channels, self.nhidden, self.node_dim = channels, nhidden, node_dim
        self.edge_dim = edge_dim + channels
        self.encoder, self.reverse_mp = encoder, reverse_mp
        self.cls_embedding = nn.Parameter(torch.empty(channels))
        self.node_emb = nn.Linear(node_dim, nhidden)
        self.edge_emb = nn.Linear(self.edge_dim, nhidden)
        self.tab_conv = ColumnTransformerLayer(channels, nhead, feedforward_channels, dropout, activation)
        self.tab_norm = nn.LayerNorm(channels)
        self.backbone = nn.ModuleList([
            FTTransformerPNAFusedLayer(channels, nhead, feedforward_channels, dropout, activation, nhidden, deg,
                                       reverse_mp) for _ in range(num_layers)])
        self.reset_parameters()

    def reset_parameters(self) -> None:
        nn.init.normal_(self.cls_embedding, std=0.01)
        self.node_emb.reset_parameters()
        self.edge_emb.reset_parameters()
        self.tab_norm.reset_parameters()
        _xavier_matrices(self.tab_conv)
        for layer in self.backbone:
            layer.reset_parameters()

    def get_shared_params(self):
        groups = [self.encoder.parameters() if self.encoder is not None else [], self.tab_conv.parameters(),
                  self.tab_norm.parameters(), [self.cls_embedding], self.node_emb.parameters(),
                  self.edge_emb.parameters(), self.backbone.parameters()]
        return [p for grp in groups for p in grp]

    def zero_grad_shared_params(self):
        for p in self.get_shared_params():
            if p.grad is not None:
                p.grad.data.zero_()

    def forward(self, x, edge_index, edge_attr, target_edge_index, target_edge_attr, lp=False):
        """x [N, n_node_feats, C]; edge_index int64 [2,E_n]; edge_attr [E_n, ncols, C]; target_edge_index int64
        [2,B]; target_edge_attr [B, ncols, C] -> (x_gnn [N,F], edge_attr [E_n,F], target_edge_attr [B,F])."""
        N = x.shape[0]
        g = ops.SubgraphIndex.build(edge_index, N)
        seeds = ops.SeedIndex(target_edge_index, N) if not lp else target_edge_index
        tn = self.tab_norm
        x_gnn = ops.linear(x.reshape(-1, self.node_dim), self.node_emb.weight, self.node_emb.bias)

        t0 = prepend_cls(self.cls_embedding, target_edge_attr)
        target = self.tab_conv(t0, tn, 0.0, 1.0)                                             # fused.py:160
        e0 = prepend_cls(self.cls_embedding, edge_attr)
        e = self.tab_conv(e0, tn, 0.5, 0.5)                                                  # :164
        e = ops.linear(e.reshape(-1, self.edge_dim), self.edge_emb.weight, self.edge_emb.bias)     # :165-166

        x_tab = target
        for layer in self.backbone:
            x_tab, x_gnn, e = layer(x_tab, x_gnn, g, e, seeds, lp)

        t_out = ops.axpby(x_tab, target, 0.5, 0.5).reshape(-1, self.edge_dim)                # :172-173
        t_out = ops.linear(t_out, self.edge_emb.weight, self.edge_emb.bias)
        return x_gnn, e, t_out


# --------------------------------------------------------------------------------------- non-fused TABGNN


class FTTransformerLayer(nn.Module):
    def __init__(self, channels, nhead, feedforward_channels=None, dropout=0.5, activation="relu", nhidden=128):
        super().__init__()
        self.tab_conv = ColumnTransformerLayer(channels, nhead, feedforward_channels, dropout, activation)
        self.tab_norm = nn.LayerNorm(channels)
        self.reset_parameters()

    def reset_parameters(self):
        _xavier_matrices(self.tab_conv)
        self.tab_norm.reset_parameters()

    def forward(self, x_tab):                                                               # tabgnn.py:218-219
        return self.tab_conv(x_tab, self.tab_norm, 0.5, 0.5)


class PNALayer(nn.Module):
    def __init__(self, channels, nhidden=128, deg=None, reverse_mp=False):
        super().__init__()
        self.gnn_conv = _make_conv(nhidden, deg, reverse_mp)
        self.gnn_norm = BatchNorm(nhidden)
        self.gnn_edge_update = nn.Sequential(nn.Linear(3 * nhidden, nhidden), nn.ReLU(), nn.Linear(nhidden, nhidden))
        self.reset_parameters()

    def reset_parameters(self):
        self.gnn_conv.reset_parameters()
        self.gnn_norm.reset_parameters()
        _xavier_matrices(self.gnn_edge_update)

    def forward(self, x_gnn, edge_index, edge_attr):                                        # tabgnn.py:187-191
        g = ops.SubgraphIndex.build(edge_index, x_gnn.shape[0])
        x_gnn = self.gnn_norm(self.gnn_conv(x_gnn, g, edge_attr), res=x_gnn, relu=True, alpha=0.5, beta_c=0.5)
        upd = ops.edge_mlp_relu(x_gnn, edge_attr, g, "src", self.gnn_edge_update[0], self.gnn_edge_update[2])
        return x_gnn, ops.axpby(edge_attr, upd, 1.0, 0.5)  # e + MLP/2 (sic)


class TABGNN(nn.Module):
    def __init__(self, channels: int, num_layers: int, deg=None, node_dim: int = 1, nhidden: int = 128,
                 edge_dim: int = None, reverse_mp: bool = False, feedforward_channels: Optional[int] = None,
                 nhead: int = 8, dropout: float = 0.5, activation: str = "relu") -> None:
        super().__init__()
        if num_layers <= 0:
            raise ValueError(f"num_layers must be a positive integer (got {num_layers})")
        self.channels, self.nhidden = channels, nhidden
        self.node_dim = node_dim + channels
        self.edge_dim = edge_dim + channels
        self.cls_embedding = nn.Parameter(torch.empty(channels))
        self.node_emb = nn.Linear(self.node_dim, nhidden)
        self.edge_emb = nn.Linear(self.edge_dim, nhidden)
        self.tabular_backbone = nn.ModuleList(
            [FTTransformerLayer(channels, nhead, feedforward_channels, dropout, activation, nhidden)
             for _ in range(num_layers)])
        self.gnn_backbone = nn.ModuleList([PNALayer(channels, nhidden, deg, reverse_mp) for _ in range(num_layers)])
        nn.init.normal_(self.cls_embedding, std=0.01)

    def get_shared_params(self):
        groups = [[self.cls_embedding], self.node_emb.parameters(), self.edge_emb.parameters(),
                  self.tabular_backbone.parameters(), self.gnn_backbone.parameters()]
        return [p for grp in groups for p in grp]

    def zero_grad_shared_params(self):
        for p in self.get_shared_params():
            if p.grad is not None:
                p.grad.data.zero_()

    def forward(self, x, edge_index, edge_attr):                                             # tabgnn.py:100-151
        g = ops.SubgraphIndex.build(edge_index, x.shape[0])
        x = prepend_cls(self.cls_embedding, x)
        e = prepend_cls(self.cls_embedding, edge_attr)
        tx, te = x, e
        for layer in self.tabular_backbone:          # the same layer serves node rows and edge rows
            tx = layer(tx)
            te = layer(te)
        x = ops.axpby(x, tx, 0.5, 0.5)
        e = ops.axpby(e, te, 0.5, 0.5)
        x = ops.linear(x.reshape(-1, self.node_dim), self.node_emb.weight, self.node_emb.bias)
        e = ops.linear(e.reshape(-1, self.edge_dim), self.edge_emb.weight, self.edge_emb.bias)
        for layer in self.gnn_backbone:
            x, e = layer(x, g, e)
        return x, e


# --------------------------------------------------------------------------------------- sibling backbones (8f rank 4)


class FTTransformerPNAInterleavedLayer(nn.Module):
    """``src/nn/models/inteleaved.py:165-227``: column attention over EVERY edge row, then a PNA layer on the CLS token
    of each edge row (the edge embedding), which is written back into the row."""

    def __init__(self, channels, nhead, feedforward_channels=None, dropout=0.5, activation="relu", nhidden=128, deg=None,
                 reverse_mp=False):
        super().__init__()
        self.channels, self.nhidden = channels, nhidden
        self.tab_conv = ColumnTransformerLayer(channels, nhead, feedforward_channels, dropout, activation)
        self.tab_norm = nn.LayerNorm(channels)
        self.gnn_conv = _make_conv(nhidden, deg, reverse_mp)
        self.gnn_norm = BatchNorm(nhidden)
        self.gnn_edge_update = nn.Sequential(nn.Linear(3 * nhidden, nhidden), nn.ReLU(), nn.Linear(nhidden, nhidden))
        self.reset_parameters()

    def reset_parameters(self):
        _xavier_matrices(self.tab_conv)
        self.tab_norm.reset_parameters()
        self.gnn_conv.reset_parameters()
        self.gnn_norm.reset_parameters()
        _xavier_matrices(self.gnn_edge_update)

    def forward(self, x_gnn, edge_index, edge_attr):
        g = ops.SubgraphIndex.build(edge_index, x_gnn.shape[0])
        edge_attr = self.tab_conv(edge_attr, self.tab_norm, 1.0, 0.5)          # e + LN(enc(e)) / 2   (sic, :217)
        cls = edge_attr[:, 0, :].contiguous()
        x_gnn = self.gnn_norm(self.gnn_conv(x_gnn, g, cls), res=x_gnn, relu=True, alpha=0.5, beta_c=0.5)
        upd = ops.edge_mlp_relu(x_gnn, cls, g, "src", self.gnn_edge_update[0], self.gnn_edge_update[2])
        cls = ops.axpby(cls, upd, 0.5, 0.5)
        return x_gnn, torch.cat([cls.unsqueeze(1), edge_attr[:, 1:, :]], dim=1)


class TABGNNInterleaved(nn.Module):
    """``TABGNNInterleaved`` (``src/nn/models/inteleaved.py:27-163``), ``--model tabgnninterleaved``
    (``utils.py:306-320``): same parameter names as the reference (``cls_embedding, node_emb, edge_emb, tab_conv,
    tab_norm, backbone.{i}.*``; ``edge_emb`` is constructed but unused there too)."""

    def __init__(self, channels: int, num_layers: int, encoder=None, deg=None, node_dim: int = 1, nhidden: int = 128,
                 edge_dim: int = None, reverse_mp: bool = False, feedforward_channels: Optional[int] = None,
                 nhead: int = 8, dropout: float = 0.5, activation: str = "relu") -> None:
        super().__init__()
        if num_layers <= 0:
            raise ValueError(f"num_layers must be a positive integer (got {num_layers})")
        self.channels, self.nhidden, self.node_dim = channels, nhidden, node_dim
        self.edge_dim = edge_dim + channels
        self.encoder, self.reverse_mp = encoder, reverse_mp
        self.cls_embedding = nn.Parameter(torch.empty(channels))
        self.node_emb = nn.Linear(node_dim, nhidden)
        self.edge_emb = nn.Linear(self.edge_dim, nhidden)
        self.tab_conv = ColumnTransformerLayer(channels, nhead, feedforward_channels, dropout, activation)
        self.tab_norm = nn.LayerNorm(channels)
        self.backbone = nn.ModuleList([
            FTTransformerPNAInterleavedLayer(channels, nhead, feedforward_channels, dropout, activation, nhidden, deg,
                                             reverse_mp) for _ in range(num_layers)])
        self.reset_parameters()

    def reset_parameters(self) -> None:
        nn.init.normal_(self.cls_embedding, std=0.01)
        self.node_emb.reset_parameters()
        self.edge_emb.reset_parameters()
        self.tab_norm.reset_parameters()
        _xavier_matrices(self.tab_conv)
        for layer in self.backbone:
            layer.reset_parameters()

    def get_shared_params(self):
        groups = [self.encoder.parameters() if self.encoder is not None else [], self.tab_conv.parameters(),
                  self.tab_norm.parameters(), [self.cls_embedding], self.node_emb.parameters(),
                  self.edge_emb.parameters(), self.backbone.parameters()]
        return [p for grp in groups for p in grp]

    def zero_grad_shared_params(self):
        for p in self.get_shared_params():
            if p.grad is not None:
                p.grad.data.zero_()

    def forward(self, x, edge_index, edge_attr):
        """x [N, n_node_feats, C]; edge_attr [E, ncols, C] -> (x_gnn [N,F], x_edge [E,C])   (inteleaved.py:140-163)."""
        g = ops.SubgraphIndex.build(edge_index, x.shape[0])
        x_gnn = ops.linear(x.reshape(-1, self.node_dim), self.node_emb.weight, self.node_emb.bias)
        e0 = self.tab_conv(prepend_cls(self.cls_embedding, edge_attr), self.tab_norm, 0.5, 0.5)
        e = e0
        for layer in self.backbone:
            x_gnn, e = layer(x_gnn, g, e)
        x_edge = ops.axpby(e[:, 0, :].contiguous(), e0[:, 0, :].contiguous(), 0.5, 0.5)
        return x_gnn, x_edge


class PNAS(nn.Module):
    """``PNAS`` (``src/nn/gnn/pna.py:48-97``), ``--model pna``: node/edge embeddings then L x {PNA + BatchNorm + residual
    average; edge update ``e + MLP/2``}.  Parameter names ``node_emb, edge_emb, convs.{i}, emlps.{i}, batch_norms.{i}``;
    note the aggregator order ['mean','min','max','std'] (:59), honoured through the post-projection columns."""

    def __init__(self, num_features, num_gnn_layers, n_classes=2, n_hidden=128, edge_updates=True, edge_dim=None,
                 dropout=0.0, final_dropout=0.5, deg=None, reverse_mp=False):
        super().__init__()
        self.n_hidden, self.num_gnn_layers, self.edge_updates = n_hidden, num_gnn_layers, edge_updates
        self.final_dropout, self.reverse_mp = final_dropout, reverse_mp
        kw = dict(in_channels=n_hidden, out_channels=n_hidden, aggregators=["mean", "min", "max", "std"], scalers=_SCAL,
                  deg=deg, edge_dim=n_hidden, towers=1, pre_layers=1, post_layers=1, divide_input=False)
        self.node_emb = nn.Linear(num_features, n_hidden)
        self.edge_emb = nn.Linear(edge_dim, n_hidden)
        self.convs, self.emlps, self.batch_norms = nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
        for _ in range(num_gnn_layers):
            self.convs.append(PNAConvHetero(n_hidden=n_hidden, **kw) if reverse_mp else PNAConv(**kw))
            if edge_updates:
                self.emlps.append(nn.Sequential(nn.Linear(3 * n_hidden, n_hidden), nn.ReLU(), nn.Linear(n_hidden, n_hidden)))
            self.batch_norms.append(BatchNorm(n_hidden))

    def forward(self, x, edge_index, edge_attr):
        g = ops.SubgraphIndex.build(edge_index, x.shape[0])
        x = ops.linear(x.reshape(x.shape[0], -1), self.node_emb.weight, self.node_emb.bias)
        e = ops.linear(edge_attr.reshape(edge_attr.shape[0], -1), self.edge_emb.weight, self.edge_emb.bias)
        for i in range(self.num_gnn_layers):
            x = self.batch_norms[i](self.convs[i](x, g, e), res=x, relu=True, alpha=0.5, beta_c=0.5)
            if self.edge_updates:
                e = ops.axpby(e, ops.edge_mlp_relu(x, e, g, "src", self.emlps[i][0], self.emlps[i][2]), 1.0, 0.5)
        return x, e


class CPNA(nn.Module):
    """``CPNA`` (``src/nn/gnn/pna.py:150-219``), ``--model cpna``: one stack of L PNA layers per edge-table column, all
    reading and updating the same node state; column c's embedding is updated by its own edge MLPs.  Parameter names
    ``node_emb, col_convs.{c}.{i}, col_emlps.{c}.{i}, col_batch_norms.{c}.{i}``.  Returns a new ``[E, ncols, F]``
    tensor where the reference writes the columns back into its argument."""

    def __init__(self, num_features, num_gnn_layers, n_classes=2, n_hidden=128, edge_updates=True, edge_dim=None,
                 dropout=0.0, final_dropout=0.5, deg=None, reverse_mp=False):
        super().__init__()
        self.n_hidden, self.num_gnn_layers, self.edge_updates = n_hidden, num_gnn_layers, edge_updates
        self.final_dropout, self.reverse_mp = final_dropout, reverse_mp
        self.num_cols = edge_dim // n_hidden
        kw = dict(in_channels=n_hidden, out_channels=n_hidden, aggregators=["mean", "min", "max", "std"], scalers=_SCAL,
                  deg=deg, edge_dim=n_hidden, towers=1, pre_layers=1, post_layers=1, divide_input=False)
        self.node_emb = nn.Linear(num_features, n_hidden)
        self.col_convs, self.col_emlps, self.col_batch_norms = nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
        for _ in range(self.num_cols):
            convs, emlps, bns = nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
            for _ in range(num_gnn_layers):
                convs.append(PNAConvHetero(n_hidden=n_hidden, **kw) if reverse_mp else PNAConv(**kw))
                if edge_updates:
                    emlps.append(nn.Sequential(nn.Linear(3 * n_hidden, n_hidden), nn.ReLU(), nn.Linear(n_hidden, n_hidden)))
                bns.append(BatchNorm(n_hidden))
            self.col_convs.append(convs); self.col_emlps.append(emlps); self.col_batch_norms.append(bns)

    def forward(self, x, edge_index, edge_attr):
        g = ops.SubgraphIndex.build(edge_index, x.shape[0])
        x = ops.linear(x.reshape(x.shape[0], -1), self.node_emb.weight, self.node_emb.bias)
        cols = []
        for c in range(self.num_cols):
            col = edge_attr[:, c, :].contiguous()
            for i in range(self.num_gnn_layers):
                x = self.col_batch_norms[c][i](self.col_convs[c][i](x, g, col), res=x, relu=True, alpha=0.5, beta_c=0.5)
                if self.edge_updates:
                    mlp = self.col_emlps[c][i]
                    col = ops.axpby(col, ops.edge_mlp_relu(x, col, g, "src", mlp[0], mlp[2]), 1.0, 0.5)
            cols.append(col)
        return x, torch.stack(cols, dim=1)


class GINe(nn.Module):
    """``GINe`` (``src/nn/gnn/gine.py:37-115``), ``--model gin``: node/edge embeddings then L x {GINEConv (sum of
    ``relu(x_j + lin(e))`` messages, Linear-ReLU-Linear network) + BatchNorm + residual average; edge update
    ``e + MLP/2``}.  Parameter names ``node_emb, edge_emb, convs.{i}, emlps.{i}, batch_norms.{i}``; with ``reverse_mp``
    the two directions of a layer share ONE network object as the reference does (gine.py:18-19)."""

    def __init__(self, num_features=1, num_gnn_layers=2, n_hidden=100, edge_updates=False, edge_dim=None,
                 reverse_mp=False):
        super().__init__()
        self.n_hidden, self.num_gnn_layers, self.edge_updates, self.reverse_mp = n_hidden, num_gnn_layers, edge_updates, reverse_mp
        self.node_emb = nn.Linear(num_features, n_hidden)
        self.edge_emb = nn.Linear(edge_dim, n_hidden)
        self.convs, self.emlps, self.batch_norms = nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
        for _ in range(num_gnn_layers):
            net = nn.Sequential(nn.Linear(n_hidden, n_hidden), nn.ReLU(), nn.Linear(n_hidden, n_hidden))
            conv = GINEConvHetero(net, n_hidden=n_hidden) if reverse_mp else GINEConv(net, edge_dim=n_hidden)
            if edge_updates:
                self.emlps.append(nn.Sequential(nn.Linear(3 * n_hidden, n_hidden), nn.ReLU(), nn.Linear(n_hidden, n_hidden)))
            self.convs.append(conv)
            self.batch_norms.append(BatchNorm(n_hidden))

    def forward(self, x, edge_index, edge_attr):
        g = ops.SubgraphIndex.build(edge_index, x.shape[0])
        x = ops.linear(x.reshape(x.shape[0], -1), self.node_emb.weight, self.node_emb.bias)
        e = ops.linear(edge_attr.reshape(edge_attr.shape[0], -1), self.edge_emb.weight, self.edge_emb.bias)
        for i in range(self.num_gnn_layers):
            x = self.batch_norms[i](self.convs[i](x, g, e), res=x, relu=True, alpha=0.5, beta_c=0.5)
            if self.edge_updates:
                e = ops.axpby(e, ops.edge_mlp_relu(x, e, g, "src", self.emlps[i][0], self.emlps[i][2]), 1.0, 0.5)
        return x, e
