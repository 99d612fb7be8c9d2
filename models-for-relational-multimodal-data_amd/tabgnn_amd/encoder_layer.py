"""The post-norm column-transformer layer as ONE autograd node with a hand-scheduled backward.

Same kernels as the op-by-op composition in ``layers.ColumnTransformerLayer`` (torch ``nn.TransformerEncoderLayer``
as configured at ``src/nn/models/fused.py:83-92``), optionally followed by the ``tab_norm`` LayerNorm and the
residual combine every call site applies (``fused.py:160,164,249``; ``tabgnn.py:219``):

    out = alpha * x + beta_c * LN_tab(encoder_layer(x))            (tail)   |   out = encoder_layer(x)

Why one node: the layer input ``x`` and the intermediate ``x1`` each feed several consumers.  Scheduled by hand,
their gradient branches accumulate in place (LayerNorm backward with ``accum_da``, GEMMs with beta = 1) instead
of through separate elementwise add kernels over [rows*S, C] tensors, and intermediates are freed as soon as the
schedule is past them.
"""
from __future__ import annotations

import ctypes
import os

import torch

from . import _lib as L
from . import ops

_FUSED_TAIL = os.environ.get("TABGNN_NO_FUSED_TAIL") != "1"      # same-box A/B switch
_FUSED_LAYER = os.environ.get("TABGNN_NO_FUSED_ENCODER") != "1"   # same-box A/B switch: the one-kernel layer (encoder_fused.hip)
_FUSED_TRAIN = os.environ.get("TABGNN_NO_FUSED_ENCODER_TRAIN") != "1"   # ... in training (fused backward kernels)
_DW_FFN = os.environ.get("TABGNN_NO_DW_FFN") != "1"      # A/B: feed-forward weight gradients inside the chained backward kernel
_DX_FOLD = os.environ.get("TABGNN_NO_DX_FOLD") != "1"    # A/B: d_x += d_qkv W_in inside the attention-half backward kernel
# opt-in: layers wider than 128 channels (C = 256, configs[4]) with their projections on tg_gemm_nt_bf16 (ReLU + dropout / gate
# epilogues) instead of hipBLASLt + separate activation kernels.  Measured in round 5, same box: 35.7-37.3 ms/step against
# 34.4-35.1 on the library — its 256 x 256 tiles beat the 128 x 128-tile kernel on K = 256 products by more than the fused
# epilogues give back — so the default stays with the library until a 256-wide tile exists.
_NT_WIDE = os.environ.get("TABGNN_WIDE_NT") == "1"
_LONG_FFN = os.environ.get("TABGNN_NO_LONG_ROW_FFN") != "1"   # A/B: fused feed-forward-half backward for rows of more than 32 tokens


def token_group(S):
    """Rows of more than 32 tokens (S = 65, 130: the tabgnn path and the 64-column table) cannot take the one-kernel layer
    — its attention runs on a wave's 32 token slots — but everything behind the attention is token-wise.  The fused
    feed-forward-half backward then runs on the flat token stream cut into pseudo rows of d tokens, d a divisor of S that
    fills the 32 slots best (S = 130 -> 2: 16 pseudo rows per tile; S = 65 -> 5: 30 of 32 slots).  None: no such divisor."""
    best = None
    for d in range(2, 33):
        if S % d == 0 and (best is None or (32 // d) * d > (32 // best) * best):
            best = d
    return best



def wave_tiles(R, S):
    """32-token wave tiles of a [R, S, C] launch of the fused kernels (a tile holds floor(32 / S) whole table rows)."""
    per = max(32 // S, 1)
    return (R + per - 1) // per

STATS = {"fused_fwd": 0, "fused_bwd": 0, "fused_bwd_attn": 0}       # launches of the one-kernel layer (tests assert the path under test ran)


def fused_ok(x, nhead, w1):
    """Shapes the one-kernel layer takes: bf16 rows of 2 <= S <= 32 column tokens, d_model = feed-forward = 128, 4 or 8 heads."""
    return (_FUSED_LAYER and x.dtype == torch.bfloat16 and x.is_cuda and x.dim() == 3 and x.shape[0] > 0
            and bool(L.load().tg_encoder_fused_supported(x.shape[1], x.shape[2], nhead, w1.shape[0])))


def pack_layer(lw_in, lw_o, lw1, lw2, b_in, b_o, g1, be1, b1, b2, g2, be2, gt, bt):
    """LDS weight images + fp32 parameter block of one call (tg_encoder_pack): bf16 weights, fp32 vectors."""
    lib = L.load()
    dev = lw_in.device
    wpack = torch.empty(lib.tg_encoder_pack_bytes(), dtype=torch.uint8, device=dev)
    prm = torch.empty(lib.tg_encoder_prm_floats(), dtype=torch.float32, device=dev)
    keep = []            # converted copies stay alive until the launch is queued (their blocks must not be reused before)

    def f(t):
        if t is None:
            return None
        t = t.detach()
        if t.dtype != torch.float32 or not t.is_contiguous():
            t = t.float().contiguous()
        keep.append(t)
        return L.ptr(t)

    weights = [w.contiguous() for w in (lw_in, lw_o, lw1, lw2)]
    L.call("tg_encoder_pack", *(L.ptr(w) for w in weights), f(b_in), f(b_o), f(g1), f(be1), f(b1), f(b2), f(g2), f(be2),
           f(gt), f(bt), L.ptr(wpack), L.ptr(prm), L.stream())
    del keep, weights    # (stream-ordered allocator: freeing after the launch call is safe)
    return wpack, prm


def fused_forward(x, nhead, p, tail, alpha, beta_c, wpack, prm, seed, rs, want_z):
    """(out, z1, z2) of the one-kernel layer; z1 / z2 only when ``want_z`` (training)."""
    R, S, C = x.shape
    out = torch.empty_like(x)
    z1 = torch.empty_like(x) if want_z else None
    z2 = torch.empty_like(x) if want_z else None
    rs_arr = (ctypes.c_uint32 * 4)(*rs)
    ops._launch("tg_encoder_fwd_bf16", L.ptr(x), L.ptr(out), L.ptr(z1), L.ptr(z2), L.ptr(wpack), L.ptr(prm), R, S, nhead,
                int(tail), float(alpha), float(beta_c), 1e-5, float(p), int(seed), ctypes.addressof(rs_arr), L.stream(),
                nbytes=2 * x.numel() * (2 + 2 * int(want_z)), units=wave_tiles(R, S))
    return out, z1, z2


def _ln_fwd(a, b, bias_b, gamma, beta, res, alpha, beta_c, p, seed, rs, eps=1e-5):
    C = a.shape[-1]
    M = a.numel() // C
    out = torch.empty_like(a)
    stats = torch.empty(M, 2, dtype=torch.float32, device=a.device)
    L.call("tg_ln_fwd", L.ptr(a), L.ptr(b), L.ptr(bias_b), L.ptr(gamma), L.ptr(beta), L.ptr(res), L.ptr(out),
           L.ptr(stats), M, C, eps, alpha, beta_c, p, seed, rs, L.dt(a), L.stream())
    return out, stats


def _ln_bwd(a, b, bias_b, gamma, stats, g, da, want_db, dres, alpha, beta_c, p, seed, rs, accum, targets=None):
    """Returns (db, dparams[3,C]); writes (or accumulates into) ``da`` and writes ``dres`` when given.  ``targets`` =
    (dgamma, dbeta, dbias) gradient-buffer pointers (ops.ln_grad_targets): the parameter gradients are then added
    there by the kernel and ``dparams`` comes back as (None, None, None)."""
    C = a.shape[-1]
    M = a.numel() // C
    db = torch.empty_like(a) if want_db else None
    dparams = None if targets else torch.empty(3, C, dtype=torch.float32, device=a.device)
    partials = torch.empty(L.load().tg_ln_partials_floats(M, C), dtype=torch.float32, device=a.device)
    L.call("tg_ln_bwd", L.ptr(a), L.ptr(b), L.ptr(bias_b), L.ptr(gamma), L.ptr(stats), L.ptr(g), L.ptr(da), L.ptr(db),
           L.ptr(dres), L.ptr(dparams), L.ptr(partials), M, C, alpha, beta_c, p, seed, rs, int(accum),
           *(targets or (None, None, None)), L.dt(a), L.stream())
    return db, (dparams if dparams is not None else (None, None, None))


class _EncoderLayerFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, nhead, p, tail, alpha, beta_c, w_in, b_in, w_o, b_o, w1, b1, w2, b2, g1, be1, g2, be2, gt, bt,
                needs_grad=True):
        x = x.contiguous()
        R, S, C = x.shape
        T = R * S
        dt = x.dtype
        sh = lambda t: t if dt == torch.float32 else ops.shadow(t, dt)
        lw_in, lb_in, lw_o, lw1, lb1, lw2 = sh(w_in), sh(b_in), sh(w_o), sh(w1), sh(b1), sh(w2)
        seed = ops.DropoutRNG.seed
        rs = [ops.DropoutRNG.next_stream() for _ in range(4)]          # attention, norm1, ffn, norm2
        x2d = x.view(T, C)
        if not needs_grad and fused_ok(x, nhead, w1):
            # inference (main.py:104-155): the whole layer in one kernel, nothing saved
            wpack, prm = pack_layer(lw_in, lw_o, lw1, lw2, b_in, b_o, g1, be1, b1, b2, g2, be2, gt if tail else None,
                                    bt if tail else None)
            out, _, _ = fused_forward(x, nhead, p, tail, alpha, beta_c, wpack, prm, seed, rs, False)
            STATS["fused_fwd"] += 1
            return out
        if needs_grad and _FUSED_TRAIN and fused_ok(x, nhead, w1):
            # training: one kernel, only the two pre-LayerNorm sums are kept; the backward recomputes the rest
            wpack, prm = pack_layer(lw_in, lw_o, lw1, lw2, b_in, b_o, g1, be1, b1, b2, g2, be2, gt if tail else None,
                                    bt if tail else None)
            out, z1, z2 = fused_forward(x, nhead, p, tail, alpha, beta_c, wpack, prm, seed, rs, True)
            STATS["fused_fwd"] += 1
            ctx.save_for_backward(x2d, z1.view(T, C), z2.view(T, C), prm, lw_in, lw_o, lw1, lw2, b_in, b_o, g1, be1, g2, be2, gt)
            ctx.cfg = (R, S, C, nhead, p, tail, alpha, beta_c, seed, rs)
            ctx.fused = True
            isp = lambda t: t if isinstance(t, torch.nn.Parameter) else None
            ctx.params = (isp(w_in), isp(b_in), isp(w_o), isp(w1), isp(b1), isp(w2), isp(b2), isp(b_o))
            ctx.ln_params = ((g1, be1, b_o), (g2, be2, b2), (gt, bt, None))
            return out.view(R, S, C)
        ctx.fused = False
        # projections on the hand-written MFMA kernel when the shapes allow (bf16, d_model = feed-forward = 128);
        # its epilogue applies bias and, for linear1, ReLU + dropout, so the pre-activation never exists
        nt = C == 128 and lw1.shape[0] == 128 and ops.nt_ok(x2d, 3 * C, C)     # every GEMM of the layer qualifies
        # wider layers (C = 256: the 64-column table of configs[4]): the GEMMs alone on the hand-written kernel, with linear1's
        # ReLU + dropout and the gate of its backward in the epilogues; the GEMM + LayerNorm kernel is built for 128 channels
        ntg = (not nt) and _NT_WIDE and dt == torch.bfloat16 and x2d.is_cuda and C % 128 == 0 and lw1.shape[0] % 128 == 0 \
            and T > 0 and x2d.data_ptr() % 16 == 0
        qkv = ops.gemm_nt(x2d, lw_in, b_in.detach()) if (nt or ntg) else torch.addmm(lb_in, x2d, lw_in.t())
        o = torch.empty(T, C, dtype=dt, device=x.device)
        lse = torch.empty(R, nhead, S, dtype=torch.float32, device=x.device)
        L.call("tg_attn_fwd", L.ptr(qkv), L.ptr(o), L.ptr(lse), R, S, C, nhead, p, seed, rs[0], L.dt(x), L.stream())
        grp = token_group(S) if (_LONG_FFN and _DW_FFN and nt and S > 32 and dt == torch.bfloat16
                                 and lw1.shape == (128, 128)) else None
        if grp is not None:
            # rows of more than 32 tokens: everything behind the attention — out-proj + LN1, FFN, LN2, tail — in the one-kernel
            # layer's attention-free variant (tg_encoder_ffn_fwd_bf16) on the token stream as pseudo rows of `grp` tokens;
            # the backward (feed-forward half: chained kernel; attention half: op by op) needs z1, z2 and the LN1 statistics
            wpack, prm = pack_layer(lw_in, lw_o, lw1, lw2, b_in, b_o, g1, be1, b1, b2, g2, be2, gt if tail else None,
                                    bt if tail else None)
            out = torch.empty_like(x2d)
            y = torch.empty_like(x2d) if needs_grad else None
            y2 = torch.empty_like(x2d) if needs_grad else None
            st1 = torch.empty(T, 2, dtype=torch.float32, device=x.device) if needs_grad else None
            rs_arr = (ctypes.c_uint32 * 4)(*rs)
            ops._launch("tg_encoder_ffn_fwd_bf16", L.ptr(x2d), L.ptr(o), L.ptr(out), L.ptr(y), L.ptr(y2), L.ptr(st1), L.ptr(wpack),
                        L.ptr(prm), T // grp, grp, int(tail), float(alpha), float(beta_c), 1e-5, float(p), int(seed),
                        ctypes.addressof(rs_arr), L.stream(), nbytes=2 * T * C * (3 + 2 * int(needs_grad)), units=wave_tiles(T // grp, grp))
            STATS["fused_ffn_fwd"] = STATS.get("fused_ffn_fwd", 0) + 1
            ctx.save_for_backward(x2d, qkv, o, lse, y, None, st1, None, None, y2, None, None, None, lw_in, lw_o, lw1, lw2, b_o, b2,
                                  g1, g2, gt)
            ctx.cfg = (R, S, C, nhead, p, tail, alpha, beta_c, seed, rs)
            ctx.nt, ctx.ntg = nt, False
            ctx.prm_args, ctx.prm = None, prm
            isp = lambda t: t if isinstance(t, torch.nn.Parameter) else None
            ctx.params = (isp(w_in), isp(b_in), isp(w_o), isp(w1), isp(b1), isp(w2), isp(b2))
            ctx.ln_params = ((g1, be1, b_o), (g2, be2, b2), (gt, bt, None))
            return out.view(R, S, C)
        if nt:      # out_proj + bias + dropout + residual + LayerNorm in one kernel; y := z1 (the pre-norm sum) is kept
            y, x1, st1 = ops.gemm_nt_ln(o, lw_o, b_o.detach(), x2d, g1.detach(), be1.detach(), p, seed, rs[1])
        else:
            y = ops.gemm_nt(o, lw_o) if ntg else o @ lw_o.t()
            x1, st1 = _ln_fwd(x2d, y, b_o, g1, be1, None, 0.0, 1.0, p, seed, rs[1])
        if nt or ntg:      # h = drop(relu(x1 W1^T + b1)) in one pass; the backward gates on h > 0 (kept AND active)
            h = ops.gemm_nt(x1, lw1, b1.detach(), ops.NT_RELU | ops.NT_DROPOUT, p, seed, rs[2])
            hpre = h
        else:
            hpre = torch.addmm(lb1, x1, lw1.t())
            h = torch.empty_like(hpre)
            L.call("tg_act_dropout_fwd", L.ptr(hpre), L.ptr(h), hpre.numel(), 1, p, seed, rs[2], L.dt(x), L.stream())
        if nt:
            y2, x2, st2 = ops.gemm_nt_ln(h, lw2, b2.detach(), x1, g2.detach(), be2.detach(), p, seed, rs[3])
        else:
            y2 = ops.gemm_nt(h, lw2) if ntg else h @ lw2.t()
            x2, st2 = _ln_fwd(x1, y2, b2, g2, be2, None, 0.0, 1.0, p, seed, rs[3])
        if tail:
            out, st3 = _ln_fwd(x2, None, None, gt, bt, x2d if alpha != 0.0 else None, alpha, beta_c, 0.0, 0, 0)
        else:
            out, st3 = x2, None
        ctx.save_for_backward(x2d, qkv, o, lse, y, x1, st1, hpre, h, y2, x2, st2, st3, lw_in, lw_o, lw1, lw2, b_o, b2,
                              g1, g2, gt)
        ctx.cfg = (R, S, C, nhead, p, tail, alpha, beta_c, seed, rs)
        ctx.nt, ctx.ntg = nt, ntg
        ctx.prm_args, ctx.prm = (lw_in, lw_o, b_in, b_o, g1, be1, b1, b2, g2, be2, gt if tail else None, bt if tail else None), None
        isp = lambda t: t if isinstance(t, torch.nn.Parameter) else None
        ctx.params = (isp(w_in), isp(b_in), isp(w_o), isp(w1), isp(b1), isp(w2), isp(b2))   # weight gradients accumulate in place
        ctx.ln_params = ((g1, be1, b_o), (g2, be2, b2), (gt, bt, None))               # LayerNorm gradients too
        return out.view(R, S, C)

    @staticmethod
    def backward(ctx, g):
        if ctx.fused:
            return _fused_backward(ctx, g)
        (x2d, qkv, o, lse, y, x1, st1, hpre, h, y2, x2, st2, st3, lw_in, lw_o, lw1, lw2, b_o, b2, g1, g2,
         gt) = ctx.saved_tensors
        R, S, C, H, p, tail, alpha, beta_c, seed, rs = ctx.cfg
        p_in, pb_in, p_o, p_1, pb_1, p_2, pb_2 = ctx.params
        T = R * S
        g = g.contiguous().view(T, C)
        dgt = dbt = None
        d_x = None
        tg1, tg2 = ops.ln_grad_targets(*ctx.ln_params[0]), ops.ln_grad_targets(*ctx.ln_params[1])
        grp = token_group(S) if ((_LONG_FFN or x1 is None) and _DW_FFN and ctx.nt and S > 32 and g.dtype == torch.bfloat16
                                 and lw1.shape == (128, 128)) else None
        if grp is not None:
            # long rows: the whole feed-forward half (tail LN, LN2, FFN, its weight / bias / LayerNorm gradients) in the
            # chained kernel of the one-kernel layer, on the token stream as pseudo rows of `grp` tokens; recomputes x1 and
            # h from (z1 = y, z2 = y2) on the forward's dropout streams (element index = token * 128 + channel in both paths)
            d_x1, (dw1, db1, dw2, db2), (dg2, dbe2, dgt, dbt) = _ffn_half_backward(
                g, y, y2, (lw1, lw2), (p_1, pb_1, p_2, pb_2), ctx.ln_params, T // grp, grp, tail, beta_c, p, seed, rs,
                ctx.prm_args, ctx.prm)
            dp2 = (dg2, dbe2, db2)
            if tail and alpha != 0.0:
                d_x = g * alpha                      # the residual branch of out = alpha x + beta_c LN_t(.)
            return _attention_half_backward(ctx, d_x, d_x1, dw1, db1, dw2, dp2, dgt, dbt)
        d_x1 = torch.empty_like(x1)
        if tail and ctx.nt and _FUSED_TAIL:
            # out = alpha*x + beta_c*LN_t(x2), x2 = LN2(z2): both LayerNorm backwards in one kernel, the gradient of x2
            # stays in registers (6 row streams instead of 8)
            d_x = torch.empty_like(x2) if alpha != 0.0 else None
            d_y2 = torch.empty_like(y2)
            tgt = ops.ln_grad_targets(*ctx.ln_params[2])
            acc = dparams = None
            if tgt is not None and tg2 is not None:
                acc = (ctypes.c_void_p * 5)(tgt[0], tgt[1], tg2[0], tg2[1], tg2[2])
            else:
                dparams = torch.empty(5, C, dtype=torch.float32, device=g.device)
            partials = torch.empty(2048 * 5 * C, dtype=torch.float32, device=g.device)
            L.call("tg_ln_tail_ln_bwd", L.ptr(x2), L.ptr(y2), L.ptr(gt), L.ptr(st3), L.ptr(g2), L.ptr(st2), L.ptr(g),
                   L.ptr(d_x), L.ptr(d_x1), L.ptr(d_y2), L.ptr(dparams), L.ptr(partials), T, C, alpha, beta_c, p, seed,
                   rs[3], acc, L.dt(x2), L.stream())
            if dparams is not None:
                dgt, dbt = dparams[0], dparams[1]
                dp2 = (dparams[2], dparams[3], dparams[4])
            else:
                dp2 = (None, None, None)
            d_x2 = None
        else:
            if tail:                                         # out = alpha*x + beta_c*LN_t(x2)
                d_x2 = torch.empty_like(x2)
                d_x = torch.empty_like(x2) if alpha != 0.0 else None
                _, dp = _ln_bwd(x2, None, None, gt, st3, g, d_x2, False, d_x, alpha, beta_c, 0.0, 0, 0, False,
                                ops.ln_grad_targets(*ctx.ln_params[2]))
                dgt, dbt = dp[0], dp[1]
            else:
                d_x2 = g
            # x2 = LN2(x1 + drop(y2 + b2))
            if ctx.nt:  # y2 holds z2 = x1 + drop(h W2^T + b2): the LayerNorm backward's "z mode"
                d_y2, dp2 = _ln_bwd(y2, None, None, g2, st2, d_x2, d_x1, True, None, 0.0, 1.0, p, seed, rs[3], False, tg2)
            else:
                d_y2, dp2 = _ln_bwd(x1, y2, b2, g2, st2, d_x2, d_x1, True, None, 0.0, 1.0, p, seed, rs[3], False, tg2)
        del d_x2
        dw2, _ = ops.weight_grad(d_y2, h, False, p_2)
        nt = ctx.nt or ctx.ntg
        if nt:      # dX GEMM with the backward of drop(relu(.)) in its epilogue: d_h never exists
            d_hpre = ops.gemm_nt(d_y2, ops.wt(lw2, p_2), None, ops.NT_GATE, p, gate=h)
            del d_y2
        else:
            d_h = d_y2 @ lw2
            del d_y2
            d_hpre = torch.empty_like(d_h)
            L.call("tg_act_dropout_bwd", L.ptr(hpre), L.ptr(d_h), L.ptr(d_hpre), hpre.numel(), 1, p, seed, rs[2],
                   L.dt(hpre), L.stream())
            del d_h
        dw1, db1 = ops.weight_grad(d_hpre, x1, True, p_1, pb_1)
        if db1 is None and dw1 is not None:
            db1 = d_hpre.sum(0, dtype=torch.float32)
        if nt:                                           # second consumer of x1: accumulated by the GEMM (beta = 1)
            ops.gemm_nt(d_hpre, ops.wt(lw1, p_1), None, ops.NT_ACCUM, out=d_x1)
        else:
            d_x1.addmm_(d_hpre, lw1)
        del d_hpre
        return _attention_half_backward(ctx, d_x, d_x1, dw1, db1, dw2, dp2, dgt, dbt)


def _attention_half_backward(ctx, d_x, d_x1, dw1, db1, dw2, dp2, dgt, dbt):
    """Op-by-op backward from d_x1 (gradient of the LN1 output) to d_x: LN1, out-proj, attention, in-proj."""
    (x2d, qkv, o, lse, y, x1, st1, hpre, h, y2, x2, st2, st3, lw_in, lw_o, lw1, lw2, b_o, b2, g1, g2,
     gt) = ctx.saved_tensors
    R, S, C, H, p, tail, alpha, beta_c, seed, rs = ctx.cfg
    p_in, pb_in, p_o = ctx.params[:3]
    T = R * S
    nt = ctx.nt or ctx.ntg
    tg1 = ops.ln_grad_targets(*ctx.ln_params[0])
    # x1 = LN1(x + drop(y + b_o))
    acc_dx = d_x is not None
    if d_x is None:
        d_x = torch.empty_like(y)
    if ctx.nt:
        d_y, dp1 = _ln_bwd(y, None, None, g1, st1, d_x1, d_x, True, None, 0.0, 1.0, p, seed, rs[1], acc_dx, tg1)
    else:
        d_y, dp1 = _ln_bwd(x2d, y, b_o, g1, st1, d_x1, d_x, True, None, 0.0, 1.0, p, seed, rs[1], acc_dx, tg1)
    del d_x1
    dwo, _ = ops.weight_grad(d_y, o, False, p_o)
    d_o = ops.gemm_nt(d_y, ops.wt(lw_o, p_o)) if nt else d_y @ lw_o
    del d_y
    d_qkv = torch.empty_like(qkv)
    L.call("tg_attn_bwd", L.ptr(qkv), L.ptr(o), L.ptr(d_o), L.ptr(lse), L.ptr(d_qkv), R, S, C, H, p, seed, rs[0],
           L.dt(qkv), L.stream())
    del d_o
    dwin, dbin = ops.weight_grad(d_qkv, x2d, True, p_in, pb_in)
    if dbin is None and dwin is not None:
        dbin = d_qkv.sum(0, dtype=torch.float32)
    if nt:                                           # third consumer of x
        ops.gemm_nt(d_qkv, ops.wt(lw_in, p_in), None, ops.NT_ACCUM, out=d_x)
    else:
        d_x.addmm_(d_qkv, lw_in)
    return (d_x.view(R, S, C), None, None, None, None, None, dwin, dbin, dwo, dp1[2], dw1, db1, dw2, dp2[2],
            dp1[0], dp1[1], dp2[0], dp2[1], dgt, dbt, None)


def _grad_ptrs(params):
    """ctypes array of the parameters' gradient-buffer pointers (FlatParams views) or None when one has none."""
    tg = [ops._grad_target(p) if isinstance(p, torch.nn.Parameter) else None for p in params]
    return tg if all(t is not None for t in tg) else None


def _ln_reduce(lnp, nblk, params):
    """Sum the per-workgroup partial LayerNorm parameter gradients the chained backward kernels left in ``lnp``
    (tg_encoder_ln_reduce).  ``params`` = up to four parameters in the kernel's row order (None = row not wanted).
    Adds into their gradient buffers when every wanted one has one (returns Nones), else returns fp32 tensors."""
    dev = lnp.device
    live = [p for p in params if p is not None]
    tg = _grad_ptrs(live)
    outs = iter(tg if tg is not None else [torch.empty(128, dtype=torch.float32, device=dev) for _ in live])
    res, ptrs = [], []
    for p in params:
        if p is None:
            ptrs.append(None); res.append(None)
        else:
            o = next(outs)
            ptrs.append(o.data_ptr())
            res.append(None if tg is not None else o)
    arr = (ctypes.c_void_p * 4)(*(ptrs + [None] * (4 - len(ptrs))))
    L.call("tg_encoder_ln_reduce", L.ptr(lnp), nblk, ctypes.addressof(arr), int(tg is not None), L.stream())
    return res


def _dw_reduce(dwp, dbp, nblk, pairs):
    """Sum the per-workgroup partial weight / bias gradients of the DW backward kernels (tg_encoder_dw_reduce).  ``pairs`` =
    ((weight, bias), ...) in the kernel's order (Parameters or None).  When every parameter owns a gradient buffer the sums
    are ADDED there (returns Nones), else fresh fp32 tensors are returned: (dW_0, db_0, dW_1, db_1, ...)."""
    dev = dwp.device
    nw = len(pairs)
    tw = [ops._grad_target(w) if isinstance(w, torch.nn.Parameter) else None for w, _ in pairs]
    tb = [ops._grad_target(b) if isinstance(b, torch.nn.Parameter) else None for _, b in pairs]
    acc = all(t is not None and t.shape == (128, 128) for t in tw) and all(t is not None for t in tb)
    if not acc:
        tw = [torch.empty(128, 128, dtype=torch.float32, device=dev) for _ in pairs]
        tb = [torch.empty(128, dtype=torch.float32, device=dev) for _ in pairs]
    aw = (ctypes.c_void_p * nw)(*[t.data_ptr() for t in tw])
    ab = (ctypes.c_void_p * nw)(*[t.data_ptr() for t in tb])
    L.call("tg_encoder_dw_reduce", L.ptr(dwp), L.ptr(dbp), nblk, nw, ctypes.addressof(aw), ctypes.addressof(ab), int(acc),
           L.stream())
    out = []
    for w, b in zip(tw, tb):
        out += [None, None] if acc else [w, b]
    return out


def _ffn_half_backward(g, z1, z2, lws, params, ln_params, R, S, tail, beta_c, p, seed, rs, prm_args=None, prm=None):
    """tg_encoder_bwd_ffn_dw_bf16 + its two ordered reductions: -> d_x1, (dW1, db1, dW2, db2), (dg2, dbe2, dgt, dbt);
    gradients of parameters that own a gradient buffer are ADDED there and come back as None."""
    lw1, lw2 = lws
    p_1, pb_1, p_2, pb_2 = params
    T, C = g.shape
    dev = g.device
    lib = L.load()
    if prm is None:
        lw_in, lw_o, b_in, b_o, g1, be1, b1, b2, g2, be2, gt, bt = prm_args
        _, prm = pack_layer(lw_in, lw_o, lw1, lw2, b_in, b_o, g1, be1, b1, b2, g2, be2, gt, bt)
    rs_arr = (ctypes.c_uint32 * 4)(*rs)
    stage = lib.tg_encoder_stage_bytes()
    tiles = [lw1.contiguous(), ops.wt(lw2, p_2).contiguous(), ops.wt(lw1, p_1).contiguous()]      # W1, W2^T, W1^T
    wpack_b = torch.empty(3 * stage, dtype=torch.uint8, device=dev)
    tp = (ctypes.c_void_p * 3)(*[t.data_ptr() for t in tiles])
    ld = (ctypes.c_int32 * 3)(*[t.stride(0) for t in tiles])
    L.call("tg_encoder_pack_tiles", ctypes.addressof(tp), ctypes.addressof(ld), 3, L.ptr(wpack_b), L.stream())
    gg2, gb2, ggt, gbt = ln_params[1][0], ln_params[1][1], ln_params[2][0], ln_params[2][1]
    nblk = lib.tg_encoder_dw_blocks(R, S)
    d_x1 = torch.empty(T, C, dtype=g.dtype, device=dev)
    lnp = torch.empty(nblk * 512, dtype=torch.float32, device=dev)
    dwp = torch.empty(nblk * 2 * 128 * 128, dtype=torch.float32, device=dev)
    dbp = torch.empty(nblk * 2 * 128, dtype=torch.float32, device=dev)
    ops._launch("tg_encoder_bwd_ffn_dw_bf16", L.ptr(g), L.ptr(z1), L.ptr(z2), L.ptr(d_x1), L.ptr(wpack_b), L.ptr(prm), R, S,
                int(tail), float(beta_c), 1e-5, float(p), int(seed), ctypes.addressof(rs_arr), L.ptr(lnp), L.ptr(dwp),
                L.ptr(dbp), L.stream(), nbytes=2 * T * C * 4, units=wave_tiles(R, S))
    STATS["fused_bwd"] += 1
    STATS["fused_bwd_dw"] = STATS.get("fused_bwd_dw", 0) + 1
    ln = _ln_reduce(lnp, nblk, (gg2, gb2, ggt if tail else None, gbt if tail else None))
    dw = _dw_reduce(dwp, dbp, nblk, ((p_1, pb_1), (p_2, pb_2)))
    return d_x1, tuple(dw), tuple(ln)


def _fused_backward(ctx, g):
    """Backward of the one-kernel layer: everything is recomputed from (x, z1, z2) by two chained kernels — the
    feed-forward half (tg_encoder_bwd_ffn_bf16) and the attention half (tg_encoder_bwd_attn_bf16, 4 or 8 heads) — which
    also leave the LayerNorm parameter gradients as per-workgroup partial sums; the weight-gradient GEMMs take the
    operand pairs the kernels wrote."""
    (x2d, z1, z2, prm, lw_in, lw_o, lw1, lw2, b_in, b_o, g1, be1, g2, be2, gt) = ctx.saved_tensors
    R, S, C, H, p, tail, alpha, beta_c, seed, rs = ctx.cfg
    p_in, pb_in, p_o, p_1, pb_1, p_2, pb_2 = ctx.params[:7]
    T = R * S
    dev = g.device
    lib = L.load()
    g = g.contiguous().view(T, C)
    rs_arr = (ctypes.c_uint32 * 4)(*rs)
    stage = lib.tg_encoder_stage_bytes()
    nblk = lib.tg_encoder_ln_partial_blocks(R, S)
    # ---- feed-forward half
    tiles = [lw1.contiguous(), ops.wt(lw2, p_2).contiguous(), ops.wt(lw1, p_1).contiguous()]      # W1, W2^T, W1^T
    wpack_b = torch.empty(3 * stage, dtype=torch.uint8, device=dev)
    tp = (ctypes.c_void_p * 3)(*[t.data_ptr() for t in tiles])
    ld = (ctypes.c_int32 * 3)(*[t.stride(0) for t in tiles])
    L.call("tg_encoder_pack_tiles", ctypes.addressof(tp), ctypes.addressof(ld), 3, L.ptr(wpack_b), L.stream())
    gg2, gb2, ggt, gbt = ctx.ln_params[1][0], ctx.ln_params[1][1], ctx.ln_params[2][0], ctx.ln_params[2][1]
    if _DW_FFN:
        # weight gradients inside the kernel: d_y2, h, d_hpre, x1 never reach HBM; per-workgroup partials summed in block order
        nblk = lib.tg_encoder_dw_blocks(R, S)
        d_x1 = torch.empty(T, C, dtype=g.dtype, device=dev)
        lnp = torch.empty(nblk * 512, dtype=torch.float32, device=dev)
        dwp = torch.empty(nblk * 2 * 128 * 128, dtype=torch.float32, device=dev)
        dbp = torch.empty(nblk * 2 * 128, dtype=torch.float32, device=dev)
        ops._launch("tg_encoder_bwd_ffn_dw_bf16", L.ptr(g), L.ptr(z1), L.ptr(z2), L.ptr(d_x1), L.ptr(wpack_b), L.ptr(prm), R, S,
                    int(tail), float(beta_c), 1e-5, float(p), int(seed), ctypes.addressof(rs_arr), L.ptr(lnp), L.ptr(dwp),
                    L.ptr(dbp), L.stream(), nbytes=2 * T * C * 4, units=wave_tiles(R, S))
        STATS["fused_bwd"] += 1
        STATS["fused_bwd_dw"] = STATS.get("fused_bwd_dw", 0) + 1
        dg2, dbe2, dgt, dbt = _ln_reduce(lnp, nblk, (gg2, gb2, ggt if tail else None, gbt if tail else None))
        dw1, db1, dw2, db2 = _dw_reduce(dwp, dbp, nblk, ((p_1, pb_1), (p_2, pb_2)))
        del dwp, dbp
    else:
        d_x1, d_y2, h, d_hpre, x1 = (torch.empty(T, C, dtype=g.dtype, device=dev) for _ in range(5))
        lnp = torch.empty(nblk * 512, dtype=torch.float32, device=dev)
        ops._launch("tg_encoder_bwd_ffn_bf16", L.ptr(g), L.ptr(z1), L.ptr(z2), L.ptr(d_x1), L.ptr(d_y2), L.ptr(h), L.ptr(d_hpre),
                    L.ptr(x1), L.ptr(wpack_b), L.ptr(prm), R, S, int(tail), float(beta_c), 1e-5, float(p), int(seed),
                    ctypes.addressof(rs_arr), L.ptr(lnp), L.stream(), nbytes=2 * T * C * 8)
        STATS["fused_bwd"] += 1
        dg2, dbe2, dgt, dbt = _ln_reduce(lnp, nblk, (gg2, gb2, ggt if tail else None, gbt if tail else None))
        dw2, db2 = ops.weight_grad(d_y2, h, True, p_2, pb_2)
        if db2 is None and dw2 is not None:
            db2 = d_y2.sum(0, dtype=torch.float32)
        del d_y2, h
        dw1, db1 = ops.weight_grad(d_hpre, x1, True, p_1, pb_1)
        if db1 is None and dw1 is not None:
            db1 = d_hpre.sum(0, dtype=torch.float32)
        del d_hpre, x1
    nblk = lib.tg_encoder_ln_partial_blocks(R, S)
    # ---- attention half: LayerNorm-1 backward, output-projection backward, attention backward on recomputed q/k/v/P
    p_bo = ctx.params[7]
    wo_t = ops.wt(lw_o, p_o)
    if not wo_t.is_contiguous():
        wo_t = wo_t.contiguous()
    # d_x += d_qkv W_in inside the kernel (W_in^T as three more weight stages); _DX_FOLD off: the separate accumulating GEMM
    win_t = ops.wt(lw_in, p_in) if _DX_FOLD else None
    if win_t is not None and win_t.stride(1) != 1:
        win_t = win_t.contiguous()
    wpack_a = torch.empty((7 if _DX_FOLD else 4) * stage, dtype=torch.uint8, device=dev)
    d_x, d_y, o = (torch.empty(T, C, dtype=g.dtype, device=dev) for _ in range(3))
    d_qkv = torch.empty(T, 3 * C, dtype=g.dtype, device=dev)
    with_g = tail and alpha != 0.0
    lnp1 = torch.empty(nblk * 512, dtype=torch.float32, device=dev)
    ops._launch("tg_encoder_bwd_attn_bf16", L.ptr(d_x1), L.ptr(z1), L.ptr(x2d), L.ptr(g) if with_g else None, L.ptr(d_x),
                L.ptr(d_y), L.ptr(o), L.ptr(d_qkv), L.ptr(lw_in.contiguous()), L.ptr(wo_t), wo_t.stride(0),
                win_t.data_ptr() if win_t is not None else None, win_t.stride(0) if win_t is not None else 0,
                L.ptr(wpack_a), L.ptr(prm), R, S, H, float(alpha) if with_g else 0.0, 1e-5, float(p), int(seed),
                ctypes.addressof(rs_arr), L.ptr(lnp1), L.stream(), nbytes=2 * T * C * (9 + int(with_g)), units=wave_tiles(R, S))
    STATS["fused_bwd_attn"] += 1
    dg1, dbe1 = _ln_reduce(lnp1, nblk, (ctx.ln_params[0][0], ctx.ln_params[0][1]))[:2]
    del d_x1
    dwo, dbo = ops.weight_grad(d_y, o, True, p_o, p_bo)
    if dbo is None and dwo is not None:
        dbo = d_y.sum(0, dtype=torch.float32)
    del d_y, o
    dwin, dbin = ops.weight_grad(d_qkv, x2d, True, p_in, pb_in)
    if dbin is None and dwin is not None:
        dbin = d_qkv.sum(0, dtype=torch.float32)
    if not _DX_FOLD:
        ops.gemm_nt(d_qkv, ops.wt(lw_in, p_in), None, ops.NT_ACCUM, out=d_x)
    return (d_x.view(R, S, C), None, None, None, None, None, dwin, dbin, dwo, dbo, dw1, db1, dw2, db2,
            dg1, dbe1, dg2, dbe2, dgt, dbt, None)


def encoder_layer(x, layer, p, tail_norm=None, alpha=0.0, beta_c=1.0):
    """``layer``: a ColumnTransformerLayer (parameter holder); ``tail_norm``: the LayerNorm applied after it."""
    sa = layer.self_attn
    tail = tail_norm is not None
    gt = tail_norm.weight if tail else None
    bt = tail_norm.bias if tail else None
    params = (sa.in_proj_weight, sa.in_proj_bias, sa.out_proj.weight, sa.out_proj.bias, layer.linear1.weight,
              layer.linear1.bias, layer.linear2.weight, layer.linear2.bias, layer.norm1.weight, layer.norm1.bias,
              layer.norm2.weight, layer.norm2.bias, gt, bt)
    # (inside Function.forward grad mode is always off and needs_input_grad ignores it: decide here)
    needs_grad = torch.is_grad_enabled() and (x.requires_grad or any(t is not None and t.requires_grad for t in params))
    return _EncoderLayerFn.apply(x, layer.nhead, float(p), tail, float(alpha), float(beta_c), *params, needs_grad)
