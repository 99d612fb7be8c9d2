"""Shared helpers for parity tests: load a golden case and rebuild its parameters / float inputs."""
from __future__ import annotations

import json
import os

import numpy as np
import torch

from detparams import det_param, det_tensor

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

FUSED_CASES = ["fused_c32_h8_l1", "fused_c32_h8_l1_lp", "fused_c32_h4_l2_rmp", "fused_c128_h4_l2",
               "fused_c128_h8_l2_cols7", "fused_amlbatch_c32_h8_l1"]


def load_case(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    cfg = json.loads(str(z["cfg"]))
    return cfg, z


def build_state(keys, z, seed, dtype_override=None):
    sd = {}
    for k, shape in keys.items():
        if ("buf." + k) in z.files:
            sd[k] = torch.from_numpy(z["buf." + k].copy())
            continue
        like = torch.empty(shape, dtype=torch.long if k.endswith("num_batches_tracked") else torch.float32)
        sd[k] = det_param(k, like, seed).to(like.dtype).reshape(shape).clone()
    return sd


def fused_inputs(cfg, z):
    N, E, B, C, nc, seed = cfg["N"], cfg["E"], cfg["B"], cfg["C"], cfg["ncols"], cfg["seed"]
    ei = torch.from_numpy(z["edge_index"].astype(np.int64))
    x = det_tensor("in.x", (N, 1, C), seed)
    ea = det_tensor("in.edge_attr", (E, nc, C), seed)
    return x, ei, ea


def fused_train_loss(cfg, xg, e, lg, y):
    """The scalar the golden gradients were taken of (make_golden.fused_case)."""
    seed = cfg["seed"]
    w = torch.tensor([1.0, 9.23], device=lg.device)
    return torch.nn.functional.cross_entropy(lg.float(), y, weight=w) \
        + 0.01 * (e.float() * det_tensor("co.e", e.shape, seed).to(e.device)).sum() / e.shape[0] \
        + 0.01 * (xg.float() * det_tensor("co.x", xg.shape, seed).to(xg.device)).sum() / xg.shape[0]


def tinycsv_state(cfg, z):
    """Flat wrapper state (``node_encoder. / edge_encoder. / model. / decoder.``) and raw feature dicts of the
    ``tinycsv_c32_h8_l1`` fixture (make_golden.tiny_csv_case): parameters regenerated from detparams, column statistics
    from the fixture's config."""
    seed = cfg["seed"]
    sd = {"model." + k: v for k, v in build_state(cfg["keys"], z, seed).items()}
    sd.update({"decoder." + k: v for k, v in build_state(cfg["head_keys"], z, seed + 1).items()})
    for k, shape in cfg["enc_keys"].items():
        sd[k] = det_param(k, torch.empty(shape), seed + 2).reshape(shape).clone()
    for i in range(3):
        sd[f"edge_encoder.encoder_dict.categorical.embs.{i}.weight"][0].zero_()            # padding row
    sd["edge_encoder.encoder_dict.numerical.mean"] = torch.tensor([cfg["mean"]], dtype=torch.float32)
    sd["edge_encoder.encoder_dict.numerical.std"] = torch.tensor([cfg["std"]], dtype=torch.float32) + 1e-6
    sd["edge_encoder.encoder_dict.timestamp.min_year"] = torch.tensor([float(cfg["min_year"])])
    ef = {"numerical": torch.from_numpy(z["num"].copy()), "categorical": torch.from_numpy(z["cat"].copy()),
          "timestamp": torch.from_numpy(z["ts"].copy())}
    nf = {"relation": torch.ones(cfg["N"], 1)}
    return sd, nf, ef
