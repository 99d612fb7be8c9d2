"""Oracle (test infrastructure): PNA convolution, degree-scaled multi-aggregation, BatchNorm.

**Parity unpinned.**  The arithmetic lives in ``torch_geometric==2.5.3``
(``environment.yml:336``: ``nn/conv/pna_conv.py``, ``nn/aggr/{scaler,multi,basic}.py``,
``nn/norm/batch_norm.py``), whose source is not under ``/root/reference`` and which is
not installed in this image.  This file restates the published 2.5.3 algorithm, anchored
on the reference's call sites:

* ``src/nn/models/fused.py:200-214,252`` — ``PNAConv(F, F, aggregators=['mean','max','min','std'],
  scalers=['identity','amplification','attenuation'], deg, edge_dim=F, towers=1, pre_layers=1,
  post_layers=1, divide_input=False)`` followed by ``BatchNorm(F)``;
* ``src/nn/models/tabgnn.py:163-172,188`` — same configuration;
* ``src/nn/gnn/pna.py:17-46`` — ``PNAConvHetero`` (in the reference itself; pinned by the goldens).

State-dict names follow PyG 2.5.3: ``edge_encoder.{weight,bias}``, ``pre_nns.0.0.*``,
``post_nns.0.0.*``, ``lin.*``, buffers ``aggr_module.avg_deg_{lin,log}``;
``BatchNorm`` wraps ``BatchNorm1d`` as ``module``.
"""
from __future__ import annotations

import math

import torch

AGGREGATORS = ("mean", "max", "min", "std")            # fused.py:198 / tabgnn.py:160 order
SCALERS = ("identity", "amplification", "attenuation")  # fused.py:199


def avg_degree_stats(deg_hist: torch.Tensor):
    """``DegreeScalerAggregation.__init__``: avg_deg_lin / avg_deg_log from the in-degree histogram
    (histogram built at ``utils.py:383-389`` from ``main.py:283-286``)."""
    h = deg_hist.to(torch.float)
    n = int(h.sum())
    bins = torch.arange(h.numel(), dtype=torch.float)
    lin = float((bins * h).sum()) / n
    log = float(((bins + 1).log() * h).sum()) / n
    return lin, log


def multi_aggregate(h, dst, num_nodes, order=AGGREGATORS):
    """mean/max/min/std of messages ``h [E,F]`` per destination -> ``[N,4F]`` and ``deg [N]``.

    mean = sum / max(cnt,1); max/min over incoming (empty -> 0, ``include_self=False``);
    std = sqrt(clamp(mean(h^2) - mean(h)^2, 1e-5)) zeroed where <= sqrt(1e-5).
    """
    E, F_ = h.shape
    idx = dst.view(-1, 1).expand(E, F_)
    cnt = torch.zeros(num_nodes, dtype=h.dtype).index_add_(0, dst, torch.ones(E, dtype=h.dtype))
    denom = cnt.clamp(min=1).unsqueeze(1)
    s1 = torch.zeros(num_nodes, F_, dtype=h.dtype).index_add_(0, dst, h)
    s2 = torch.zeros(num_nodes, F_, dtype=h.dtype).index_add_(0, dst, h * h)
    mean = s1 / denom
    mx = torch.zeros(num_nodes, F_, dtype=h.dtype).scatter_reduce(0, idx, h, reduce="amax", include_self=False)
    mn = torch.zeros(num_nodes, F_, dtype=h.dtype).scatter_reduce(0, idx, h, reduce="amin", include_self=False)
    var = s2 / denom - mean * mean
    std = var.clamp(min=1e-5).sqrt()
    std = std.masked_fill(std <= math.sqrt(1e-5), 0.0)
    by_name = {"mean": mean, "max": mx, "min": mn, "std": std}
    return torch.cat([by_name[a] for a in order], dim=1), cnt   # PyG concatenates in the module's aggregator order


def degree_scale(agg, deg, avg_deg_log):
    """identity | * log(deg+1)/avg_log | * avg_log/log(max(deg,1)+1)  ->  [N, 3*4F]."""
    d = deg.view(-1, 1)
    amp = torch.log(d + 1) / avg_deg_log
    att = avg_deg_log / torch.log(d.clamp(min=1) + 1)
    return torch.cat([agg, agg * amp, agg * att], dim=1)


def pna_conv(x, edge_index, edge_attr, sd, pfx, aggregators=AGGREGATORS):
    """One ``PNAConv.forward``: x [N,F], edge_index int64 [2,E] (row 0 = source j, row 1 = target i),
    edge_attr [E,F] -> [N,F]."""
    N = x.shape[0]
    src, dst = edge_index[0], edge_index[1]
    e = edge_attr @ sd[pfx + "edge_encoder.weight"].t() + sd[pfx + "edge_encoder.bias"]
    h = torch.cat([x[dst], x[src], e], dim=-1)                       # [x_i, x_j, e]
    h = h @ sd[pfx + "pre_nns.0.0.weight"].t() + sd[pfx + "pre_nns.0.0.bias"]
    agg, deg = multi_aggregate(h, dst, N, aggregators)
    out = degree_scale(agg, deg, sd[pfx + "aggr_module.avg_deg_log"])
    out = torch.cat([x, out], dim=-1)                                 # [N,13F]
    out = out @ sd[pfx + "post_nns.0.0.weight"].t() + sd[pfx + "post_nns.0.0.bias"]
    return out @ sd[pfx + "lin.weight"].t() + sd[pfx + "lin.bias"]


def pna_conv_hetero(x, edge_index, edge_attr, sd, pfx, aggregators=AGGREGATORS):
    """``PNAConvHetero.forward`` (src/nn/gnn/pna.py:38-46): forward conv + conv on flipped edges,
    then ``lin([x, a_in, a_out])``."""
    a_in = pna_conv(x, edge_index, edge_attr, sd, pfx + "conv_forw.", aggregators)
    a_out = pna_conv(x, edge_index.flipud(), edge_attr, sd, pfx + "conv_back.", aggregators)
    return torch.cat([x, a_in, a_out], dim=1) @ sd[pfx + "lin.weight"].t() + sd[pfx + "lin.bias"]


def gnn_conv(x, edge_index, edge_attr, sd, pfx, aggregators=AGGREGATORS):
    if (pfx + "conv_forw.lin.weight") in sd:
        return pna_conv_hetero(x, edge_index, edge_attr, sd, pfx, aggregators)
    return pna_conv(x, edge_index, edge_attr, sd, pfx, aggregators)


def batch_norm(x, sd, pfx, training, momentum=0.1, eps=1e-5, update_stats=True):
    """PyG ``BatchNorm`` -> ``BatchNorm1d``: train = batch mean / biased var (running stats updated with the
    unbiased var), eval = running stats.  ``pfx`` ends with ``module.``."""
    w, b = sd[pfx + "weight"], sd[pfx + "bias"]
    if training:
        mu = x.mean(dim=0)
        var = x.var(dim=0, unbiased=False)
        if update_stats:
            with torch.no_grad():
                n = x.shape[0]
                sd[pfx + "running_mean"].mul_(1 - momentum).add_(momentum * mu.detach())
                sd[pfx + "running_var"].mul_(1 - momentum).add_(momentum * var.detach() * (n / max(n - 1, 1)))
                sd[pfx + "num_batches_tracked"].add_(1)
    else:
        mu, var = sd[pfx + "running_mean"], sd[pfx + "running_var"]
    return (x - mu) * torch.rsqrt(var + eps) * w + b
