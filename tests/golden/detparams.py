"""Deterministic tensors for golden vectors: a value depends only on (name, shape, seed).

Fixtures then need to hold inputs' *indices* and the expected outputs, not 33 MB of weights:
``make_golden.py`` (run where /root/reference exists) and the parity tests (run anywhere)
both regenerate parameters and float inputs from this file.
"""
from __future__ import annotations

import zlib

import numpy as np
import torch

_BUFFER_TAILS = ("avg_deg_lin", "avg_deg_log")


def det_array(name: str, shape, seed: int = 0, scale: float = 1.0) -> np.ndarray:
    rs = np.random.RandomState((zlib.crc32(name.encode()) + 7919 * seed) % (2 ** 31 - 1))
    return (rs.standard_normal(tuple(shape)) * scale).astype(np.float32)


def det_tensor(name, shape, seed=0, scale=1.0) -> torch.Tensor:
    return torch.from_numpy(det_array(name, shape, seed, scale))


def det_param(key: str, like: torch.Tensor, seed: int = 0) -> torch.Tensor:
    """Value for one state-dict entry (shape/dtype of ``like``)."""
    shape = tuple(like.shape)
    if key.endswith("num_batches_tracked"):
        return torch.tensor(3, dtype=like.dtype)
    if key.endswith("running_var"):
        return det_tensor(key, shape, seed).abs() * 0.5 + 0.6
    if key.endswith("running_mean"):
        return det_tensor(key, shape, seed, 0.2)
    if key.endswith("cls_embedding"):
        return det_tensor(key, shape, seed, 0.5)
    if key.endswith(".weight") and len(shape) == 1:          # LayerNorm / BatchNorm gains
        return 1.0 + det_tensor(key, shape, seed, 0.1)
    if key.endswith("bias") or len(shape) == 1:
        return det_tensor(key, shape, seed, 0.1)
    fan_in = int(np.prod(shape[1:]))
    return det_tensor(key, shape, seed, 1.0 / np.sqrt(fan_in))


def fill_state_dict(module: torch.nn.Module, seed: int = 0) -> None:
    """Overwrite every float parameter/buffer of ``module`` in place (degree buffers excepted)."""
    with torch.no_grad():
        for k, v in module.state_dict().items():
            if k.endswith(_BUFFER_TAILS):
                continue
            v.copy_(det_param(k, v, seed))


def rand_subgraph(n_nodes, n_edges, n_seed, seed=0, n_isolated=5, dup_edges=6, dup_seed=3):
    """Small sampled-subgraph stand-in: seed edges first, duplicate edges, repeated seed endpoints,
    isolated nodes (ids >= n_nodes - n_isolated never appear)."""
    rs = np.random.RandomState(1000 + seed)
    live = n_nodes - n_isolated
    # heavy-tailed destinations
    w = 1.0 / np.arange(1, live + 1) ** 0.8
    w /= w.sum()
    src = rs.randint(0, live, size=n_edges)
    dst = rs.choice(live, size=n_edges, p=w)
    for k in range(dup_edges):                         # exact duplicate edges
        src[-1 - k], dst[-1 - k] = src[n_seed + k], dst[n_seed + k]
    for k in range(dup_seed):                          # repeated seed endpoints
        src[k + 1] = src[0]
        dst[n_seed - 1 - k] = src[0]
    return np.stack([src, dst]).astype(np.int64)
