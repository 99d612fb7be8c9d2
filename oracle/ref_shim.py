"""Oracle (test infrastructure): import the reference's own model files in THIS container.

Only ``tests/golden/make_golden.py`` uses this, and only where ``/root/reference`` exists
(never on the GPU box).  Technique recorded in SURVEY.md §8c: the package ``__init__`` files
import datasets/LLM code that cannot load here, so ``src``, ``src.nn``, ``src.nn.models`` and
``src.nn.gnn`` are pre-registered as bare namespace modules, and the two absent third-party
packages are represented by ``oracle.pyg_restate`` (``PNAConv``/``GINEConv``/``BatchNorm``/``Linear`` — parity
unpinned) plus a type-annotation-only ``StypeWiseFeatureEncoder`` name.  The reference's
``fused.py``, ``tabgnn.py``, ``pna.py``, ``gine.py``, ``decoder.py``, ``src/nn/decoder/self_supervised.py`` and
``src/utils/loss.py`` then execute unmodified.
"""
from __future__ import annotations

import importlib
import importlib.util
import os
import sys
import types

REF = os.environ.get("TABGNN_REFERENCE_ROOT", "/root/reference")


def _ns(name, path=None):
    m = types.ModuleType(name)
    if path is not None:
        m.__path__ = [path]
    sys.modules[name] = m
    return m


def load_reference():
    """Returns a dict of the reference classes on the path (real reference code objects)."""
    if not os.path.isdir(REF):
        raise FileNotFoundError(f"{REF} not present (reference never travels to the GPU box)")
    from . import pyg_restate as R

    tg = _ns("torch_geometric"); tgn = _ns("torch_geometric.nn")
    tgn.PNAConv, tgn.BatchNorm, tgn.Linear, tgn.GINEConv = R.PNAConv, R.BatchNorm, R.Linear, R.GINEConv
    tg.nn = tgn
    dense = _ns("torch_geometric.nn.dense"); lin = _ns("torch_geometric.nn.dense.linear"); lin.Linear = R.Linear
    tgn.dense = dense; dense.linear = lin
    inits = _ns("torch_geometric.nn.inits")
    inits.reset = lambda m: [c.reset_parameters() for c in m.modules() if hasattr(c, "reset_parameters") and c is not m]
    typing_ = _ns("torch_geometric.typing")
    for n in ("Adj", "OptPairTensor", "OptTensor", "Size"):
        setattr(typing_, n, object)
    _ns("torch_frame"); _ns("torch_frame.nn"); _ns("torch_frame.nn.encoder")
    se = _ns("torch_frame.nn.encoder.stypewise_encoder"); se.StypeWiseFeatureEncoder = object

    _ns("src", f"{REF}/src"); _ns("src.nn", f"{REF}/src/nn")
    _ns("src.nn.models", f"{REF}/src/nn/models"); _ns("src.nn.gnn", f"{REF}/src/nn/gnn")
    fused = importlib.import_module("src.nn.models.fused")
    tabgnn = importlib.import_module("src.nn.models.tabgnn")
    pna = importlib.import_module("src.nn.gnn.pna")
    dec = importlib.import_module("src.nn.gnn.decoder")
    _ns("src.nn.decoder", f"{REF}/src/nn/decoder")
    ssl = importlib.import_module("src.nn.decoder.self_supervised")
    spec = importlib.util.spec_from_file_location("tabgnn_ref_loss", f"{REF}/src/utils/loss.py")   # pure torch file
    loss = importlib.util.module_from_spec(spec); spec.loader.exec_module(loss)
    inter = importlib.import_module("src.nn.models.inteleaved")
    gine = importlib.import_module("src.nn.gnn.gine")
    extra = {"GINe": gine.GINe, "TABGNNInterleaved": inter.TABGNNInterleaved, "PNAS": pna.PNAS, "CPNA": pna.CPNA, "LinkPredHead": dec.LinkPredHead, "MCMHead": ssl.MCMHead, "SelfSupervisedHead": ssl.SelfSupervisedHead,
             "SSLoss": loss.SSLoss}
    return {**extra, "TABGNNFused": fused.TABGNNFused, "FTTransformerPNAFusedLayer": fused.FTTransformerPNAFusedLayer,
            "TABGNN": tabgnn.TABGNN, "PNAConvHetero": pna.PNAConvHetero,
            "ClassifierHead": dec.ClassifierHead, "NodeClassificationHead": dec.NodeClassificationHead}


def load_reference_graph_util():
    """The reference's ``src/datasets/util/graph.py`` (ports / ego-id preprocessing).  Its module-level imports of
    torch_geometric / torch_frame are only names here (``NeighborSampler``, ``stype``)."""
    if not os.path.isdir(REF):
        raise FileNotFoundError(f"{REF} not present (reference never travels to the GPU box)")
    tg = sys.modules.get("torch_geometric") or _ns("torch_geometric")
    smp = _ns("torch_geometric.sampler"); smp.NeighborSampler = object
    tg.sampler = smp
    tf = sys.modules.get("torch_frame") or _ns("torch_frame")
    if not hasattr(tf, "stype"):
        tf.stype = object
    spec = importlib.util.spec_from_file_location("tabgnn_ref_graph_util", f"{REF}/src/datasets/util/graph.py")
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    return mod
