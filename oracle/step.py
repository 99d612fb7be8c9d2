"""Oracle (test infrastructure): model wrapper forward, loss and optimiser step.

* ``wrapper_forward``  — ``TABGNNFusedS.forward`` (``utils.py:353-362``): seed edges are the first
  ``batch_size`` columns/rows; encoders; backbone; ``ClassifierHead``.
* ``tabgnn_wrapper_forward`` — ``TABGNNS.forward`` (``utils.py:257-282``).
* ``weighted_ce``      — ``torch.nn.CrossEntropyLoss(weight=w)`` (``main.py:335``), weighted mean.
* ``adam_step``        — ``torch.optim.Adam(lr)`` defaults (``main.py:336``): betas (0.9,0.999), eps 1e-8, no decay.
* ``train_step``       — ``main.py:41-75``: zero_grad, forward, loss on the first batch_size rows, backward, step.

State is ONE flat dict with the wrapper's four sub-modules as prefixes
(``node_encoder.``, ``edge_encoder.``, ``model.``, ``decoder.`` — ``utils.py:336-341``).
"""
from __future__ import annotations

import torch

from .encoders import stypewise_encode
from .fused_path import fused_forward
from .heads import classifier_head, node_classification_head
from .tabgnn_path import tabgnn_forward


def _sub(sd, pfx):
    n = len(pfx)
    return {k[n:]: v for k, v in sd.items() if k.startswith(pfx)}


def wrapper_forward(sd, nhead, batch_size, node_feats, edge_index, edge_feats,
                    p_backbone=0.0, p_head=0.0, training=False):
    """node_feats / edge_feats: dict stype -> raw tensor (rows: nodes / edges, seed edges first)."""
    tgt = {k: v[:batch_size] for k, v in edge_feats.items()}
    nbr = {k: v[batch_size:] for k, v in edge_feats.items()}
    ei, tei = edge_index[:, batch_size:], edge_index[:, :batch_size]
    x = stypewise_encode(node_feats, sd, "node_encoder.")
    e = stypewise_encode(nbr, sd, "edge_encoder.")
    t = stypewise_encode(tgt, sd, "edge_encoder.")
    x, e, t = fused_forward(_sub(sd, "model."), nhead, x, ei, e, tei, t,
                            lp=False, p_drop=p_backbone, training=training)
    return classifier_head(x, tei, t, sd, "decoder.", p_head, training)


def tabgnn_wrapper_forward(sd, nhead, batch_size, node_feats, edge_index, edge_feats, task,
                           p_backbone=0.0, p_head=0.0, training=False):
    x = stypewise_encode(node_feats, sd, "node_encoder.")
    e = stypewise_encode(edge_feats, sd, "edge_encoder.")
    x, e = tabgnn_forward(_sub(sd, "model."), nhead, x, edge_index, e, p_backbone, training)
    if task == "edge_classification":
        return classifier_head(x, edge_index[:, :batch_size], e[:batch_size], sd, "decoder.", p_head, training)
    return node_classification_head(x, sd, "decoder.", p_head, training)


def gnn_wrapper_forward(sd, model_name, batch_size, node_feats, edge_index, edge_feats, p_head=0.0, training=False):
    """``GNN.forward`` (``utils.py:137-160``), edge classification: encoders on ALL rows, backbone on the whole
    subgraph, head on the first ``batch_size`` (seed) edges.  ``model_name``: gin | pna | cpna."""
    from .gine import gine_forward
    from .siblings import cpna_forward, pnas_forward
    x = stypewise_encode(node_feats, sd, "node_encoder.")
    e = stypewise_encode(edge_feats, sd, "edge_encoder.")
    x = x.reshape(x.shape[0], -1)
    fwd = {"gin": gine_forward, "pna": pnas_forward, "cpna": cpna_forward}[model_name]
    x, e = fwd(_sub(sd, "model."), x, edge_index, e, training=training)
    e = e.reshape(e.shape[0], -1)                                                   # utils.py:141-143
    return classifier_head(x, edge_index[:, :batch_size], e[:batch_size], sd, "decoder.", p_head, training)


def weighted_ce(logits, y, w):
    logp = torch.log_softmax(logits, dim=-1)
    picked = -logp.gather(1, y.view(-1, 1)).squeeze(1)
    wy = w[y]
    return (picked * wy).sum() / wy.sum()


def adam_step(params, grads, state, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    state["t"] = state.get("t", 0) + 1
    t = state["t"]
    with torch.no_grad():
        for k, p in params.items():
            g = grads.get(k)
            if g is None:
                continue
            m = state.setdefault("m." + k, torch.zeros_like(p))
            v = state.setdefault("v." + k, torch.zeros_like(p))
            m.mul_(beta1).add_(g, alpha=1 - beta1)
            v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
            denom = (v.sqrt() / (1 - beta2 ** t) ** 0.5).add_(eps)
            p.addcdiv_(m, denom, value=-lr / (1 - beta1 ** t))


def trainable_keys(sd):
    skip = ("running_mean", "running_var", "num_batches_tracked", "avg_deg_lin", "avg_deg_log",
            ".mean", ".std", "min_year", "max_values", "mult_term")
    return [k for k, v in sd.items() if v.is_floating_point() and not k.endswith(skip)]


def train_step(sd, opt_state, nhead, batch_size, node_feats, edge_index, edge_feats, y, loss_w, lr,
               p_backbone=0.0, p_head=0.0):
    keys = trainable_keys(sd)
    for k in keys:
        sd[k].requires_grad_(True)
        sd[k].grad = None
    logits = wrapper_forward(sd, nhead, batch_size, node_feats, edge_index, edge_feats,
                             p_backbone, p_head, training=True)
    loss = weighted_ce(logits[:batch_size], y.view(-1).long(), loss_w)
    loss.backward()
    grads = {k: sd[k].grad for k in keys if sd[k].grad is not None}
    adam_step({k: sd[k] for k in keys}, grads, opt_state, lr)
    return float(loss.detach()), logits.detach()
