"""Oracle (test infrastructure): self-supervised heads and losses of the LP / MCM pre-training path.
Pinned by ``tests/golden/ssl_heads_c32.npz`` (generated from the reference's own ``decoder.py``,
``self_supervised.py`` and ``loss.py``)."""
from __future__ import annotations

import torch
import torch.nn.functional as F


def _lp_mlp(h, sd, pfx, p_drop, training):
    """LinkPredHead.mlp (src/nn/gnn/decoder.py:48-55): Linear(3h,h) ReLU Drop Linear(h,25) ReLU Drop Linear(25,k)."""
    h = F.dropout(torch.relu(h @ sd[pfx + "mlp.0.weight"].t() + sd[pfx + "mlp.0.bias"]), p_drop, training)
    h = F.dropout(torch.relu(h @ sd[pfx + "mlp.3.weight"].t() + sd[pfx + "mlp.3.bias"]), p_drop, training)
    return h @ sd[pfx + "mlp.6.weight"].t() + sd[pfx + "mlp.6.bias"]


def link_pred_head(x, pos_edge_index, pos_edge_attr, neg_edge_index, neg_edge_attr, sd, pfx="", p_drop=0.0,
                   training=False):
    """``LinkPredHead.forward`` decoder.py:63-72 -> (sigmoid scores of positives [P,k], of negatives [Q,k])."""
    Fh = x.shape[1]

    def score(ei, ea):
        h = x[ei.t()].reshape(-1, 2 * Fh).relu()
        h = torch.cat((h, ea.view(-1, ea.shape[1])), 1)
        return torch.sigmoid(_lp_mlp(h, sd, pfx, p_drop, training))

    return score(pos_edge_index, pos_edge_attr), score(neg_edge_index, neg_edge_attr)


def _decoder(x, sd, pfx):
    """Sequential(LayerNorm, ReLU, Linear)  (self_supervised.py:137-146: keys .0.* and .2.*)."""
    h = F.layer_norm(x, (x.shape[-1],), sd[pfx + "0.weight"], sd[pfx + "0.bias"], 1e-5)
    return torch.relu(h) @ sd[pfx + "2.weight"].t() + sd[pfx + "2.bias"]


def mcm_head(x, sd, n_categorical, pfx=""):
    """``MCMHead.forward`` / ``SelfSupervisedHead.forward`` (self_supervised.py:8-43,134-171)."""
    return _decoder(x, sd, pfx + "num_decoder."), [_decoder(x, sd, f"{pfx}cat_decoder.{i}.") for i in range(n_categorical)]


def lp_loss(pos_pred, neg_pred):
    """``SSLoss.lp_loss`` src/utils/loss.py:10-12."""
    return -torch.log(pos_pred + 1e-12).mean() - torch.log(1 - neg_pred + 1e-12).mean()


def mcm_loss(cat_out, num_out, y, num_numerical):
    """``SSLoss.mcm_loss`` src/utils/loss.py:41-72, the per-sample Python loop written as per-column masked sums.
    y[:,0] = masked value (class id or number), y[:,1] = masked column (numerical columns first)."""
    y_val, y_idx = y[:, 0], y[:, 1].long()
    cat_mask = y_idx >= num_numerical
    cat_loss = torch.zeros((), dtype=torch.float32)
    acc = torch.zeros((), dtype=torch.float32)
    for c, logits in enumerate(cat_out):
        rows = torch.where(cat_mask & (y_idx - num_numerical == c))[0]
        if rows.numel():
            t = y_val[rows].long()
            cat_loss = cat_loss + F.cross_entropy(logits[rows], t, reduction="sum")
            acc = acc + (logits[rows].argmax(1) == t).sum()
    num_rows = torch.where(~cat_mask)[0]
    num_loss = ((num_out[num_rows, y_idx[num_rows]] - y_val[num_rows]) ** 2).sum()
    t_c, t_n = int(cat_mask.sum()), int((~cat_mask).sum())
    if t_c == 0:
        total = torch.sqrt(num_loss / t_n)
    elif t_n == 0:
        total = cat_loss / t_c
    else:
        total = cat_loss / t_c + torch.sqrt(num_loss / t_n)
    return total, (cat_loss, t_c, acc), (num_loss, t_n)


def mv_loss(mv_out, y):
    """``SSLoss.mv_loss`` loss.py:74-78."""
    return F.cross_entropy(mv_out, y[:, 1].long())
