"""Minimal stand-ins for the pytorch-frame (Atahanak fork) data containers the path consumes.

The reference feeds ``torch_frame.TensorFrame`` objects into the model wrapper
(``utils.py:353-359``; built by ``src/datasets/ibm_transactions_for_aml.py:159-180``).  torch_frame is
not part of this build, so the same surface is kept here: ``feat_dict`` (stype -> raw tensor),
``col_names_dict``, row slicing and ``.to(device)`` — exactly what ``TABGNNFusedS.forward`` touches.
"""
from __future__ import annotations

import enum
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import torch


class stype(enum.Enum):
    """Semantic column types, in pytorch-frame's canonical order (numerical first); ``relation`` is fork-only."""
    numerical = "numerical"
    categorical = "categorical"
    timestamp = "timestamp"
    relation = "relation"


STYPE_ORDER = [stype.numerical, stype.categorical, stype.timestamp, stype.relation]


@dataclass
class TensorFrame:
    """feat_dict: numerical float32 [R,nn]; categorical int64 [R,nc] (-1 = missing);
    timestamp int64 [R,nt,7] (year, month, day, dayofweek, hour, minute, second); relation float32 [R,nr]."""
    feat_dict: Dict[stype, torch.Tensor]
    col_names_dict: Dict[stype, List[str]]
    y: Optional[torch.Tensor] = None

    @property
    def stypes(self):
        return [s for s in STYPE_ORDER if s in self.feat_dict]

    @property
    def num_rows(self):
        return next(iter(self.feat_dict.values())).shape[0]

    @property
    def num_cols(self):
        return sum(len(v) for v in self.col_names_dict.values())

    def __len__(self):
        return self.num_rows

    def __getitem__(self, index):
        if isinstance(index, tuple):          # tf[a:b, :] as utils.py:355 writes it
            index = index[0]
        return TensorFrame({k: v[index] for k, v in self.feat_dict.items()}, self.col_names_dict,
                           None if self.y is None else self.y[index])

    def to(self, device):
        return TensorFrame({k: v.to(device) for k, v in self.feat_dict.items()}, self.col_names_dict,
                           None if self.y is None else self.y.to(device))
