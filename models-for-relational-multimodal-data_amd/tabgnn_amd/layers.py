"""Building blocks with the reference's module/parameter names, running on the HIP operators.

* ``ColumnTransformerLayer`` — state-dict compatible with ``torch.nn.TransformerEncoderLayer(d_model=C, nhead,
  dim_feedforward, dropout, 'relu', batch_first=True)`` as built at ``src/nn/models/fused.py:83-92,187-196``.
* ``PNAConv`` / ``BatchNorm`` — state-dict compatible with torch_geometric 2.5.3 (``fused.py:204-214``).
* ``PNAConvHetero`` — ``src/nn/gnn/pna.py:17-46``.
* ``GINEConv`` / ``GINEConvHetero`` — torch_geometric 2.5.3 ``GINEConv(nn, edge_dim=...)`` and ``src/nn/gnn/gine.py:15-35``.
"""
from __future__ import annotations


import torch
from torch import nn

from . import ops


class Linear(nn.Linear):
    """nn.Linear parameters, HIP-path forward (also stands in for ``torch_geometric.nn.Linear``)."""

    def forward(self, x):
        return ops.linear(x, self.weight, self.bias)


class LayerNorm(nn.LayerNorm):
    def forward(self, x, res=None, alpha=0.0, beta_c=1.0):
        return ops.layer_norm(x, self.weight, self.bias, res=res, eps=self.eps, alpha=alpha, beta_c=beta_c)


class _SelfAttention(nn.Module):
    """Parameter holder named like ``nn.MultiheadAttention`` (packed in-projection + ``out_proj``)."""

    def __init__(self, channels, nhead):
        super().__init__()
        self.num_heads = nhead
        self.in_proj_weight = nn.Parameter(torch.empty(3 * channels, channels))
        self.in_proj_bias = nn.Parameter(torch.empty(3 * channels))
        self.out_proj = nn.Linear(channels, channels)
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.zeros_(self.in_proj_bias)
        self.out_proj.reset_parameters()
        nn.init.zeros_(self.out_proj.bias)


class ColumnTransformerLayer(nn.Module):
    """Post-norm encoder layer over the S column tokens of each row: [R,S,C] -> [R,S,C]."""

    def __init__(self, channels, nhead, dim_feedforward=None, dropout=0.5, activation="relu"):
        super().__init__()
        if activation != "relu":
            raise ValueError("only the reference's activation='relu' is implemented")
        if channels % nhead != 0 or (channels // nhead) not in (4, 8, 16, 32, 64):
            raise ValueError(f"head dim {channels}/{nhead} unsupported (need 4/8/16/32/64)")
        ff = dim_feedforward or channels
        self.nhead, self.p = nhead, dropout
        self.self_attn = _SelfAttention(channels, nhead)
        self.linear1 = nn.Linear(channels, ff)
        self.linear2 = nn.Linear(ff, channels)
        self.norm1 = nn.LayerNorm(channels, eps=1e-5)
        self.norm2 = nn.LayerNorm(channels, eps=1e-5)

    def forward(self, x, tail_norm=None, alpha=0.0, beta_c=1.0):
        """enc(x), or alpha*x + beta_c*tail_norm(enc(x)) — one autograd node (encoder_layer.py)."""
        from .encoder_layer import encoder_layer
        return encoder_layer(x, self, self.p if self.training else 0.0, tail_norm, alpha, beta_c)

    def forward_unfused(self, x):
        """Op-by-op composition of the same kernels (kept for the parity tests of the single operators)."""
        p = self.p if self.training else 0.0
        sa = self.self_attn
        qkv = ops.linear(x, sa.in_proj_weight, sa.in_proj_bias)
        o = ops.attention_core(qkv, self.nhead, p)
        y = ops.linear(o, sa.out_proj.weight, None)
        x1 = ops.layer_norm(x, self.norm1.weight, self.norm1.bias, b=y, bias_b=sa.out_proj.bias, p_drop=p)
        h = ops.act_dropout(ops.linear(x1, self.linear1.weight, self.linear1.bias), "relu", p)
        y2 = ops.linear(h, self.linear2.weight, None)
        return ops.layer_norm(x1, self.norm2.weight, self.norm2.bias, b=y2, bias_b=self.linear2.bias, p_drop=p)


class BatchNorm(nn.Module):
    """torch_geometric ``BatchNorm``: wraps ``BatchNorm1d`` as ``module`` (state-dict ``module.*``)."""

    def __init__(self, in_channels, eps=1e-5, momentum=0.1):
        super().__init__()
        self.module = nn.BatchNorm1d(in_channels, eps, momentum)
        self.sync_group = None      # set by train.DataParallel(sync_batchnorm=True): batch statistics over all ranks

    def reset_parameters(self):
        self.module.reset_parameters()

    def forward(self, x, res=None, relu=False, alpha=0.0, beta_c=1.0, sink_res=None):
        m = self.module
        training = self.training and x.shape[0] > 1
        out = ops.batch_norm_act_res(x, m.weight, m.bias, m.running_mean, m.running_var, training, res=res,
                                     momentum=m.momentum, eps=m.eps, relu=relu, alpha=alpha, beta_c=beta_c,
                                     group=self.sync_group if training else None, sink_res=sink_res)
        if training:
            m.num_batches_tracked += 1
        return out


class _DegreeScalerBuffers(nn.Module):
    def __init__(self, deg):
        super().__init__()
        h = deg.to(torch.float)
        n = float(h.sum())
        bins = torch.arange(h.numel(), dtype=torch.float)
        self.register_buffer("avg_deg_lin", torch.tensor([float((bins * h).sum()) / n]))
        self.register_buffer("avg_deg_log", torch.tensor([float(((bins + 1).log() * h).sum()) / n]))


class PNAConv(nn.Module):
    """PNAConv(F, F, ['mean','max','min','std'], ['identity','amplification','attenuation'], deg, edge_dim=F,
    towers=1, pre_layers=1, post_layers=1, divide_input=False).

    Linear maps with nothing between them are folded on the (tiny) weights each call, so no E- or N-scale
    intermediate exists for them: edge_encoder into the message projection, ``lin`` into the post projection;
    the degree scalers are applied after the post GEMM (``[N,12F]`` never materialises)."""

    def __init__(self, in_channels, out_channels, aggregators, scalers, deg, edge_dim=None, towers=1, pre_layers=1,
                 post_layers=1, divide_input=False, **kw):
        super().__init__()
        if (sorted(aggregators) != ["max", "mean", "min", "std"]
                or list(scalers) != ["identity", "amplification", "attenuation"] or towers != 1 or pre_layers != 1
                or post_layers != 1 or divide_input or edge_dim is None or in_channels != out_channels):
            raise ValueError("PNAConv: only the reference's configurations (fused.py:200-207, pna.py:59-72: the four "
                             "aggregators in either order, three scalers, one tower) are implemented")
        F = in_channels
        self.F = F
        # the aggregation kernel writes [mean|max|min|std]; agg_order[k] = position of that aggregator in the
        # module's own list (pna.py uses ['mean','min','max','std']), i.e. which column block of post_nn it owns
        self.agg_order = [list(aggregators).index(a) for a in ("mean", "max", "min", "std")]
        self.aggr_module = _DegreeScalerBuffers(deg)
        self.edge_encoder = nn.Linear(edge_dim, F)
        self.pre_nns = nn.ModuleList([nn.Sequential(nn.Linear(3 * F, F))])
        self.post_nns = nn.ModuleList([nn.Sequential(nn.Linear(13 * F, out_channels))])
        self.lin = nn.Linear(out_channels, out_channels)

    def reset_parameters(self):
        for m in (self.edge_encoder, self.pre_nns[0][0], self.post_nns[0][0], self.lin):
            m.reset_parameters()

    def sinks_ok(self, x):
        """May the caller hand this convolution the shared gradient buffers (ops.GradSink) of x and edge_attr?  Only on
        the path where every consumer of x inside it adds into the sink (the scaled post projection)."""
        return ops.post_scaled_ok(x, None, 4 * self.F)

    def forward(self, x, edge_index, edge_attr, sink_x=None, sink_e=None):
        F = self.F
        g = ops.SubgraphIndex.build(edge_index, x.shape[0])
        pre, post = self.pre_nns[0][0], self.post_nns[0][0]
        # message: W_pre [x_i, x_j, W_e e + b_e] + b_pre;  lin(post([x, agg, amp*agg, att*agg])) = x Wx^T + b +
        # (agg Wid^T) + amp (agg Wamp^T) + att (agg Watt^T): both folds on the (tiny) weights, one autograd node
        w_msg, b_msg, w_x, b_eff, w_st = ops.fold_pna_weights(pre.weight, pre.bias, self.edge_encoder.weight,
                                                              self.edge_encoder.bias, post.weight, post.bias,
                                                              self.lin.weight, self.lin.bias, self.agg_order,
                                                              lp_dtype=x.dtype)
        # messages are produced directly in destination-sorted order: the aggregation then streams contiguous rows
        if sink_x is not None and not self.sinks_ok(x):
            raise RuntimeError("PNAConv: gradient sinks need the scaled post projection path (ask sinks_ok first)")
        h = ops.edge_linear(x, edge_attr, g, "dst_sorted", w_msg, b_msg, sink_x, sink_e)
        agg = ops.pna_aggregate(h, g, sorted_rows=True)                 # [N,4F]
        if ops.post_scaled_ok(x, agg):          # scalers inside the GEMMs: G [N,3F] and its gradient never exist
            return ops.pna_post_scaled(x, w_x, b_eff, agg, w_st, g, self.aggr_module.avg_deg_log, sink_x)
        xw = ops.linear(x, w_x, b_eff)
        G = ops.linear(agg, w_st, None)
        return ops.pna_scale_combine(xw, G, g, self.aggr_module.avg_deg_log)


class PNAConvHetero(nn.Module):
    """Forward + reverse message passing (src/nn/gnn/pna.py:17-46)."""

    def __init__(self, n_hidden, in_channels, out_channels, aggregators, scalers, deg, edge_dim, towers=1,
                 pre_layers=1, post_layers=1, divide_input=False):
        super().__init__()
        kw = dict(in_channels=in_channels, out_channels=out_channels, aggregators=aggregators, scalers=scalers, deg=deg,
                  edge_dim=edge_dim, towers=towers, pre_layers=pre_layers, post_layers=post_layers,
                  divide_input=divide_input)
        self.conv_forw = PNAConv(**kw)
        self.conv_back = PNAConv(**kw)
        self.lin = Linear(n_hidden * 3, n_hidden)

    def reset_parameters(self):
        self.conv_forw.reset_parameters()
        self.conv_back.reset_parameters()
        self.lin.reset_parameters()

    def forward(self, x, edge_index, edge_attr):
        g = ops.SubgraphIndex.build(edge_index, x.shape[0])
        a_in = self.conv_forw(x, g, edge_attr)
        a_out = self.conv_back(x, g.flip(), edge_attr)
        return self.lin(torch.cat([x, a_in, a_out], dim=1))


class GINEConv(nn.Module):
    """``GINEConv(nn, edge_dim=F)`` (torch_geometric 2.5.3; built at ``src/nn/gnn/gine.py:62-72``):
    ``nn((1+eps)*x_i + sum_j relu(x_j + lin(e_ji)))``, state-dict ``nn.*``, ``lin.*`` and the ``eps`` buffer (train_eps=False).
    ``self_term=False`` is the ``(x, None)`` call of ``GINEConvHetero`` (gine.py:31-32): no ``(1+eps)*x_i`` term."""

    def __init__(self, nn_seq, eps=0.0, train_eps=False, edge_dim=None):
        super().__init__()
        if (train_eps or edge_dim is None or not isinstance(nn_seq, nn.Sequential) or len(nn_seq) != 3
                or not isinstance(nn_seq[0], nn.Linear) or not isinstance(nn_seq[1], nn.ReLU)
                or not isinstance(nn_seq[2], nn.Linear)):
            raise ValueError("GINEConv: only the reference's configuration (Linear-ReLU-Linear network, edge_dim given, "
                             "fixed eps; gine.py:62-72) is implemented")
        self.nn = nn_seq
        self.initial_eps = eps
        self.register_buffer("eps", torch.full((1,), float(eps)))
        self.lin = nn.Linear(edge_dim, nn_seq[0].in_features)

    def reset_parameters(self):
        for m in (self.nn[0], self.nn[2], self.lin):
            m.reset_parameters()
        self.eps.fill_(self.initial_eps)

    def _eps_value(self):
        """Host copy of the ``eps`` buffer, re-read only when the buffer object or its version changes (a checkpoint
        may carry a different eps; one device read per load, none per step)."""
        key = (id(self.eps), self.eps._version)
        if getattr(self, "_eps_key", None) != key:
            self._eps_key, self._eps_host = key, float(self.eps)
        return self._eps_host

    def forward(self, x, edge_index, edge_attr, self_term=True):
        g = ops.SubgraphIndex.build(edge_index, x.shape[0])
        le = ops.linear(edge_attr, self.lin.weight, self.lin.bias)
        out = ops.gine_aggregate(x, le, g, 1.0 + self._eps_value() if self_term else 0.0)
        return ops.mlp_relu(out, self.nn[0], self.nn[2])


class GINEConvHetero(nn.Module):
    """Forward + reverse message passing (src/nn/gnn/gine.py:15-35)."""

    def __init__(self, network, n_hidden):
        super().__init__()
        self.conv_forw = GINEConv(network, edge_dim=n_hidden)
        self.conv_back = GINEConv(network, edge_dim=n_hidden)      # the SAME network object, as gine.py:18-19: shared weights
        self.lin = Linear(n_hidden * 3, n_hidden)

    def reset_parameters(self):
        self.conv_forw.reset_parameters()
        self.conv_back.reset_parameters()
        self.lin.reset_parameters()

    def forward(self, x, edge_index, edge_attr):
        g = ops.SubgraphIndex.build(edge_index, x.shape[0])
        a_in = self.conv_forw(x, g, edge_attr, self_term=False)
        a_out = self.conv_back(x, g.flip(), edge_attr, self_term=False)
        return self.lin(torch.cat([x, a_in, a_out], dim=1))
