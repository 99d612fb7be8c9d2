"""Oracle (test infrastructure): post-norm transformer encoder layer over table columns.

Restates ``torch.nn.TransformerEncoderLayer(d_model=C, nhead=H, dim_feedforward=C,
dropout=p, activation='relu', batch_first=True)`` exactly as the reference configures it
(``src/nn/models/fused.py:83-92`` top-level ``tab_conv`` and ``:187-196`` per-layer
``tab_conv``; ``src/nn/models/tabgnn.py:199-208``), norm_first=False, eps=1e-5.

Pinned: ``tests/test_oracle_golden.py`` compares this against torch's own module
through the golden vectors produced by the shim-imported reference files.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def layer_norm(x, w, b, eps=1e-5):
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
    return (x - mu) * torch.rsqrt(var + eps) * w + b


def column_self_attention(x, sd, pfx, nhead, p_drop=0.0, training=False):
    """Multi-head self-attention over the S column tokens of every row.

    x: [R, S, C].  Packed ``in_proj_weight [3C, C]`` / ``in_proj_bias [3C]``,
    heads of size C/H, softmax(QK^T / sqrt(C/H)) V, attention dropout on P,
    ``out_proj`` (torch ``nn.MultiheadAttention``; used by fused.py:160,164,249).
    """
    R, S, C = x.shape
    d = C // nhead
    qkv = x @ sd[pfx + "self_attn.in_proj_weight"].t() + sd[pfx + "self_attn.in_proj_bias"]
    q, k, v = qkv.split(C, dim=-1)
    q = q.reshape(R, S, nhead, d).transpose(1, 2)          # [R,H,S,d]
    k = k.reshape(R, S, nhead, d).transpose(1, 2)
    v = v.reshape(R, S, nhead, d).transpose(1, 2)
    scores = (q @ k.transpose(-1, -2)) / math.sqrt(d)      # [R,H,S,S]
    p = torch.softmax(scores, dim=-1)
    p = F.dropout(p, p_drop, training)
    o = (p @ v).transpose(1, 2).reshape(R, S, C)
    return o @ sd[pfx + "self_attn.out_proj.weight"].t() + sd[pfx + "self_attn.out_proj.bias"]


def encoder_layer(x, sd, pfx, nhead, p_drop=0.0, training=False):
    """x = LN1(x + drop(MHA(x))); x = LN2(x + drop(W2 drop(relu(W1 x)))).  [R,S,C] -> [R,S,C]."""
    a = column_self_attention(x, sd, pfx, nhead, p_drop, training)
    x = layer_norm(x + F.dropout(a, p_drop, training), sd[pfx + "norm1.weight"], sd[pfx + "norm1.bias"])
    h = torch.relu(x @ sd[pfx + "linear1.weight"].t() + sd[pfx + "linear1.bias"])
    h = F.dropout(h, p_drop, training)
    h = h @ sd[pfx + "linear2.weight"].t() + sd[pfx + "linear2.bias"]
    x = layer_norm(x + F.dropout(h, p_drop, training), sd[pfx + "norm2.weight"], sd[pfx + "norm2.bias"])
    return x
