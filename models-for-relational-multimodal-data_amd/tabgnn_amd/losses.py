"""Self-supervised pre-training losses (``src/utils/loss.py:5-78``) on device tensors.  B-scale reductions: plain
device torch ops, no per-sample Python loop and no ``.cpu()`` round trip (the reference moves predictions to the
host and loops over samples, ``fused.py:289-290``, ``loss.py:54-56``)."""
from __future__ import annotations

import torch
import torch.nn.functional as F


class SSLoss:
    def __init__(self, device, num_numerical=None):
        self.device, self.num_numerical = device, num_numerical

    @staticmethod
    def lp_loss(pos_pred, neg_pred):
        return -torch.log(pos_pred + 1e-12).mean() - torch.log(1 - neg_pred + 1e-12).mean()

    def mcm_loss(self, cat_out, num_out, y):
        """-> (loss, (cat_loss_sum, t_c, n_correct), (num_sq_err_sum, t_n)), as the reference.  y[:,0] masked value,
        y[:,1] masked column index (numerical columns first)."""
        nn_ = self.num_numerical
        y_val, y_idx = y[:, 0], y[:, 1].long()
        cat_mask = y_idx >= nn_
        zero = torch.zeros((), dtype=torch.float32, device=y.device)
        cat_loss, acc = zero, zero
        for c, logits in enumerate(cat_out):                       # loop over COLUMNS (a handful), not samples
            sel = cat_mask & (y_idx - nn_ == c)
            t = torch.where(sel, y_val, torch.zeros_like(y_val)).long().clamp_(0, logits.shape[1] - 1)
            nll = F.cross_entropy(logits.float(), t, reduction="none")
            cat_loss = cat_loss + (nll * sel).sum()
            acc = acc + ((logits.argmax(1) == t) & sel).sum()
        num_sel = ~cat_mask
        pred = num_out.float().gather(1, y_idx.clamp(0, num_out.shape[1] - 1).unsqueeze(1)).squeeze(1)
        num_loss = (((pred - y_val) ** 2) * num_sel).sum()
        t_c, t_n = int(cat_mask.sum().item()), int(num_sel.sum().item())
        if t_c == 0:
            total = torch.sqrt(num_loss / t_n)
        elif t_n == 0:
            total = cat_loss / t_c
        else:
            total = cat_loss / t_c + torch.sqrt(num_loss / t_n)
        return total, (cat_loss, t_c, acc), (num_loss, t_n)

    @staticmethod
    def mv_loss(mv_out, y):
        return F.cross_entropy(mv_out.float(), y[:, 1].long())
