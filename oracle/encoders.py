"""Oracle (test infrastructure): per-stype column encoders, TensorFrame -> float [rows, ncols, C].

**Parity unpinned.**  The reference calls ``torch_frame.nn`` encoders from the
``Atahanak/pytorch-frame`` fork (``.gitmodules:1-3``; the submodule directory is empty and no
commit is recorded).  Constructed at ``src/datasets/ibm_transactions_for_aml.py:283-294``
(edges: ``EmbeddingEncoder``, ``LinearEncoder``, ``TimestampEncoder``) and ``:313-319`` (nodes:
fork-only ``ProjectionEncoder`` over the ``relation`` stype); called at ``utils.py:357-359``.
Restated from upstream pytorch-frame 0.2.x:

* categorical: ``Embedding(card+1, C, padding_idx=0)(idx + 1)`` per column (NaN index -1 -> row 0);
* numerical:   ``((v - mean_c) / std_c) * w_c + b_c`` (std buffer already carries the +1e-6);
* timestamp:   7 calendar fields; year -> sinusoidal positional encoding of ``year - min_year``,
  the other six / [12,31,7,24,60,60] -> cyclic encoding ``[sin(pi k v), cos(2 pi k v)]``, ``out_size=8``,
  contracted with ``weight[col,7,8,C]`` (``einsum('ijkl,jklm->ijm')``) + bias;
* every stype: ``nan_to_num(nan=0)``; stypes concatenated on dim 1 in the canonical stype order
  (numerical, categorical, timestamp, relation);
* relation (``ProjectionEncoder``, semantics unknown — fork source absent): restated as a per-column
  learned affine of the raw value, ``v * w_c + b_c``.

State-dict names mirror upstream: ``encoder_dict.<stype>.…``.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

STYPE_ORDER = ("numerical", "categorical", "timestamp", "relation")
TS_OUT_SIZE = 8
TS_CYCLIC_DIV = (12.0, 31.0, 7.0, 24.0, 60.0, 60.0)


def encode_numerical(v, sd, pfx):
    z = (v - sd[pfx + "mean"]) / sd[pfx + "std"]
    return z.unsqueeze(-1) * sd[pfx + "weight"] + sd[pfx + "bias"]


def encode_categorical(idx, sd, pfx):
    cols = []
    for c in range(idx.shape[1]):
        table = sd[f"{pfx}embs.{c}.weight"]
        cols.append(F.embedding(idx[:, c] + 1, table, padding_idx=0))
    return torch.stack(cols, dim=1)


def timestamp_features(ts, min_year):
    """ts int64 [R, ncol, 7] -> [R, ncol, 7, 8]."""
    f = ts.to(torch.float32)
    year = f[..., :1] - min_year.view(1, -1, 1).to(torch.float32)
    rest = f[..., 1:] / torch.tensor(TS_CYCLIC_DIV).view(1, 1, -1)
    half = TS_OUT_SIZE // 2
    pos_mult = torch.pow(1 / 10000.0, torch.arange(0, TS_OUT_SIZE, 2) / TS_OUT_SIZE)
    a = year.unsqueeze(-1) * pos_mult.view(1, 1, 1, -1)
    pos = torch.cat([torch.sin(a), torch.cos(a)], dim=-1)
    k = torch.arange(1, half + 1, dtype=torch.float32).view(1, 1, 1, -1)
    b = rest.unsqueeze(-1) * k
    cyc = torch.cat([torch.sin(b * math.pi), torch.cos(b * 2 * math.pi)], dim=-1)
    return torch.cat([pos, cyc], dim=2)


def encode_timestamp(ts, sd, pfx):
    feats = timestamp_features(ts, sd[pfx + "min_year"])
    return torch.einsum("ijkl,jklm->ijm", feats, sd[pfx + "weight"]) + sd[pfx + "bias"]


def encode_relation(v, sd, pfx):
    return v.to(torch.float32).unsqueeze(-1) * sd[pfx + "weight"] + sd[pfx + "bias"]


_ENC = {"numerical": encode_numerical, "categorical": encode_categorical,
        "timestamp": encode_timestamp, "relation": encode_relation}


def stypewise_encode(feat_dict, sd, pfx=""):
    """``StypeWiseFeatureEncoder.forward``: feat_dict maps stype name -> raw tensor."""
    xs = []
    for st in STYPE_ORDER:
        if st in feat_dict:
            x = _ENC[st](feat_dict[st], sd, f"{pfx}encoder_dict.{st}.")
            xs.append(torch.nan_to_num(x, nan=0.0))
    return torch.cat(xs, dim=1)
