"""Model wrappers with the reference's ``config`` contract (``utils.py:235-404``).

``TABGNNFusedS(config).forward(x: TensorFrame, edge_index, edge_attr: TensorFrame) -> logits``: the first
``batch_size`` edges are the seed edges (``ibm_transactions_for_aml.py:64-66``), encoders -> backbone -> head.
"""
from __future__ import annotations

import torch
from torch import nn

from . import ops
from .heads import ClassifierHead, NodeClassificationHead
from .models import CPNA, PNAS, TABGNN, GINe, TABGNNFused, TABGNNInterleaved


def degree_histogram(in_degrees):
    """utils.py:383-389: histogram of the train-graph in-degrees (main.py:283-286)."""
    in_degrees = in_degrees.to(torch.long)
    hist = torch.zeros(int(in_degrees.max()) + 1, dtype=torch.long)
    hist += torch.bincount(in_degrees, minlength=hist.numel()).cpu()
    return hist


class TABGNNFusedS(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.batch_size = config["batch_size"]
        self.node_encoder = config["node_encoder"]
        self.edge_encoder = config["edge_encoder"]
        self.model = self.get_model(config)
        if config["task"] == "edge_classification":
            self.decoder = ClassifierHead(config["n_classes"], config["n_hidden"], dropout=config["dropout"])
        elif config["task"] == "node_classification":
            self.decoder = NodeClassificationHead(config["n_classes"], config["n_hidden"], dropout=config["dropout"])
        else:
            raise ValueError(f"task {config['task']} is outside the supervised hot path")
        if config.get("load_model") is not None and config.get("checkpoint"):
            self.decoder.load_state_dict(torch.load(config["load_model"] + "decoder"))

    def forward(self, x, edge_index, edge_attr):
        bs = self.batch_size
        edge_attr, target_edge_attr = edge_attr[bs:, :], edge_attr[:bs, :]
        prebuilt = edge_index if isinstance(edge_index, ops.BatchIndex) else None
        if prebuilt is None:
            edge_index, target_edge_index = edge_index[:, bs:].contiguous(), edge_index[:, :bs].contiguous()
        x, _ = self.node_encoder(x)
        edge_attr, _ = self.edge_encoder(edge_attr)
        target_edge_attr, _ = self.edge_encoder(target_edge_attr)
        if prebuilt is not None:      # the sampler built the CSRs with the batch (sampler.batch_index): no index kernels here
            if prebuilt.seeds.B != bs or prebuilt.graph.N != x.shape[0]:
                raise RuntimeError("BatchIndex does not match this batch (seed count / node count)")
            edge_index, seeds = prebuilt.graph, prebuilt.seeds
        else:
            seeds = ops.SeedIndex(target_edge_index, x.shape[0])      # one CSR of the seed endpoints for backbone AND head
        x, edge_attr, target_edge_attr = self.model(x, edge_index, edge_attr, seeds, target_edge_attr)
        if self.config["task"] == "edge_classification":
            return self.decoder(x, seeds, target_edge_attr)
        return self.decoder(x)

    def get_model(self, config):
        n_dim = config["num_node_features"] * config["n_hidden"]
        e_dim = config["num_edge_features"] * config["n_hidden"]
        if config["model"] != "tabgnnfused":
            raise ValueError("Invalid model name!")
        if config.get("in_degrees") is None:
            raise ValueError("In degrees are not provided for PNA model!")
        kw = {k: config[k] for k in ("nhead", "backbone_dropout") if k in config}
        model = TABGNNFused(node_dim=n_dim, nhidden=config["n_hidden"], channels=config["n_hidden"],
                            num_layers=config["n_gnn_layers"], edge_dim=e_dim,
                            deg=degree_histogram(config["in_degrees"]), reverse_mp=config.get("reverse_mp", False),
                            nhead=kw.get("nhead", 8), dropout=kw.get("backbone_dropout", 0.5))
        if config.get("load_model") is not None:
            model.load_state_dict(torch.load(config["load_model"] + "model"))
        return model


class TABGNNS(nn.Module):
    """``utils.py:235-328``: sequential FT-Transformer -> PNA model (BASELINE config 4, ``--model tabgnn``)."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.batch_size = config["batch_size"]
        self.node_encoder = config["node_encoder"]
        self.edge_encoder = config["edge_encoder"]
        n_dim = config["num_node_features"] * config["n_hidden"]
        e_dim = config["num_edge_features"] * config["n_hidden"]
        if config.get("in_degrees") is None:
            raise ValueError("In degrees are not provided for PNA model!")
        name = config.get("model", "tabgnn")
        if name not in ("tabgnn", "tabgnninterleaved"):
            raise ValueError("Invalid model name!")                                            # utils.py:321-322
        cls = TABGNN if name == "tabgnn" else TABGNNInterleaved                              # utils.py:292-320
        self.model = cls(node_dim=n_dim, nhidden=config["n_hidden"], channels=config["n_hidden"],
                         num_layers=config["n_gnn_layers"], edge_dim=e_dim,
                         deg=degree_histogram(config["in_degrees"]), reverse_mp=config.get("reverse_mp", False),
                         nhead=config.get("nhead", 8), dropout=config.get("backbone_dropout", 0.5))
        if config["task"] == "edge_classification":
            self.decoder = ClassifierHead(config["n_classes"], config["n_hidden"], dropout=config["dropout"])
        else:
            self.decoder = NodeClassificationHead(config["n_classes"], config["n_hidden"], dropout=config["dropout"])

    def forward(self, x, edge_index, edge_attr):
        x, _ = self.node_encoder(x)
        edge_attr, _ = self.edge_encoder(edge_attr)
        x, edge_attr = self.model(x, edge_index, edge_attr)
        if getattr(self, "cpna", False):
            edge_attr = edge_attr.reshape(edge_attr.shape[0], -1)                            # utils.py:142-144
        if self.config["task"] == "edge_classification":
            bs = self.batch_size
            return self.decoder(x, edge_index[:, :bs].contiguous(), edge_attr[:bs].contiguous())
        return self.decoder(x)



class GNN(nn.Module):
    """``utils.py:111-233`` for ``--model gin`` (``GINe``), ``--model pna`` (``PNAS``) and ``--model cpna`` (``CPNA``: the
    head then sees all ``num_edge_features`` column embeddings of a seed edge, ``e_hidden = ncols * n_hidden``,
    :120-122,142-144): encoders -> backbone -> head on the first ``batch_size`` (seed) edges, or on the nodes.  Not
    built: cpnatab (whose reference forward returns nothing, pna.py:285-302)."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.batch_size = config["batch_size"]
        self.node_encoder = config["node_encoder"]
        self.edge_encoder = config["edge_encoder"]
        name = config.get("model", "pna")
        if name not in ("gin", "pna", "cpna"):
            raise ValueError("Invalid model name!")
        n_dim = config["num_node_features"] * config["n_hidden"]
        e_dim = config["num_edge_features"] * config["n_hidden"]
        if name == "gin":      # utils.py:170-175 passes num_features=n_feats, which cannot multiply the encoders'
            # [N, n_feats*n_hidden] node rows; the flattened width is used here so the route runs
            self.model = GINe(num_features=n_dim, num_gnn_layers=config["n_gnn_layers"], n_hidden=config["n_hidden"],
                              edge_updates=config.get("emlps", True), edge_dim=e_dim,
                              reverse_mp=config.get("reverse_mp", False))
        else:
            if config.get("in_degrees") is None:
                raise ValueError("In degrees are not provided for PNA model!")
            cls = PNAS if name == "pna" else CPNA
            self.model = cls(num_features=n_dim, n_hidden=config["n_hidden"], num_gnn_layers=config["n_gnn_layers"],
                             edge_dim=e_dim, deg=degree_histogram(config["in_degrees"]),
                             edge_updates=config.get("emlps", True), reverse_mp=config.get("reverse_mp", False))
        self.cpna = name == "cpna"
        if config["task"] == "edge_classification":
            e_hidden = config["num_edge_features"] * config["n_hidden"] if self.cpna else None
            self.decoder = ClassifierHead(config["n_classes"], config["n_hidden"], dropout=config["dropout"],
                                          e_hidden=e_hidden)
        elif self.cpna:     # utils.py:126-127 sizes the node head by the edge width; the node state itself is n_hidden
            self.decoder = NodeClassificationHead(config["n_classes"], config["num_edge_features"] * config["n_hidden"],
                                                  dropout=config["dropout"])
        else:
            self.decoder = NodeClassificationHead(config["n_classes"], config["n_hidden"], dropout=config["dropout"])

    def forward(self, x, edge_index, edge_attr):
        x, _ = self.node_encoder(x)
        edge_attr, _ = self.edge_encoder(edge_attr)
        x, edge_attr = self.model(x, edge_index, edge_attr)
        if getattr(self, "cpna", False):
            edge_attr = edge_attr.reshape(edge_attr.shape[0], -1)                            # utils.py:142-144
        if self.config["task"] == "edge_classification":
            bs = self.batch_size
            return self.decoder(x, edge_index[:, :bs].contiguous(), edge_attr[:bs].contiguous())
        return self.decoder(x)
