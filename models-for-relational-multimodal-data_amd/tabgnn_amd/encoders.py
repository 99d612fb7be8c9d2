"""Stype-wise column encoders with the pytorch-frame module/state-dict layout, executed by one HIP kernel.

Mirrors ``torch_frame.nn`` ``StypeWiseFeatureEncoder`` + ``EmbeddingEncoder`` / ``LinearEncoder`` /
``TimestampEncoder`` / (fork) ``ProjectionEncoder`` as the reference configures them
(``src/datasets/ibm_transactions_for_aml.py:283-294,313-319``) and calls them
(``utils.py:357-359``: ``encoder(tf) -> (Tensor[R, ncols, C], col_names)``).
Parameter names follow upstream: ``encoder_dict.<stype>.…``.  Semantics restated in ``oracle/encoders.py``.
"""
from __future__ import annotations

import ctypes as C
import os

import torch
from torch import nn

from . import _lib as L
from .frame import STYPE_ORDER, TensorFrame, stype

KIND = {stype.numerical: 0, stype.categorical: 1, stype.timestamp: 2, stype.relation: 3}
TS_FIELDS, TS_OUT = 7, 8


class LinearEncoder(nn.Module):
    """numerical: ((v - mean_c) / std_c) * w_c + b_c."""

    def __init__(self, out_channels, means, stds):
        super().__init__()
        n = len(means)
        self.register_buffer("mean", torch.tensor(means, dtype=torch.float32))
        self.register_buffer("std", torch.tensor(stds, dtype=torch.float32) + 1e-6)
        self.weight = nn.Parameter(torch.empty(n, out_channels))
        self.bias = nn.Parameter(torch.empty(n, out_channels))
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.normal_(self.weight, std=0.01)
        nn.init.zeros_(self.bias)


class EmbeddingEncoder(nn.Module):
    """categorical: Embedding(card+1, C, padding_idx=0)(idx + 1) per column (``embs.<i>.weight``)."""

    def __init__(self, out_channels, cardinalities):
        super().__init__()
        self.embs = nn.ModuleList([nn.Embedding(c + 1, out_channels, padding_idx=0) for c in cardinalities])

    def reset_parameters(self):
        for e in self.embs:
            e.reset_parameters()


class TimestampEncoder(nn.Module):
    """timestamp: positional (year - min_year) / cyclic (month..second) features, out_size 8, contracted with
    ``weight [ncol, 7, 8, C]`` + ``bias``."""

    def __init__(self, out_channels, min_years):
        super().__init__()
        n = len(min_years)
        self.register_buffer("min_year", torch.tensor(min_years, dtype=torch.float32))
        self.register_buffer("max_values", torch.tensor([12., 31., 7., 24., 60., 60.]))
        self.weight = nn.Parameter(torch.empty(n, TS_FIELDS, TS_OUT, out_channels))
        self.bias = nn.Parameter(torch.empty(n, out_channels))
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.normal_(self.weight, std=0.01)
        nn.init.zeros_(self.bias)


class ProjectionEncoder(nn.Module):
    """relation (fork-only, semantics restated): v * w_c + b_c."""

    def __init__(self, out_channels, ncols):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(ncols, out_channels))
        self.bias = nn.Parameter(torch.empty(ncols, out_channels))
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.normal_(self.weight, std=0.01)
        nn.init.zeros_(self.bias)


_TS_GEMM = os.environ.get("TABGNN_NO_TS_GEMM") != "1"


class _Encode(torch.autograd.Function):
    @staticmethod
    def forward(ctx, enc, feats, out_dtype, row_ids, *params):
        # params order: num_w, num_b, cat_table, ts_w, ts_b, rel_w, rel_b (None where the stype is absent)
        plan = enc._plan
        R = row_ids.shape[0] if row_ids is not None else next(iter(feats.values())).shape[0]
        dev = next(iter(feats.values())).device
        Cc = enc.out_channels
        # The rows are written into columns 1.. of a [R, ncols+1, C] buffer and returned as that view: the backbones
        # prepend a CLS token to every encoded row (fused.py:158-163, tabgnn.py:120-125), which then is a write into
        # the free column 0 instead of a copy of the whole tensor (models.prepend_cls); same contract
        # ``[R, ncols, C]`` for every other consumer (a strided view).
        S = plan["ncols"] + 1
        buf = torch.empty(R, S, Cc, dtype=out_dtype, device=dev)
        out = buf[:, 1:, :]
        ptrs = enc._ptrs(feats, params, row_ids)
        # bf16, C = 128: the timestamp columns run as GEMMs over a [R,128] feature operand (tg_encode_ts_features) — the
        # generic kernels get the descriptor without them
        ts_gemm = _TS_GEMM and out_dtype == torch.bfloat16 and Cc == 128 and R > 0 and any(n[2] for n in plan["nots"])
        ts_feats = {}
        if ts_gemm:
            ts_w, ts_b = params[3], params[4]
            ts_raw = feats[stype.timestamp]
            for dn, _, ts_cols in plan["nots"]:
                if dn.ncol > 0:
                    L.call("tg_encode_fwd", C.addressof(dn), C.addressof(ptrs), out.data_ptr(), R, S, Cc, L.dt(out), L.stream())
                for src_col, out_col, _ in ts_cols:
                    f = torch.empty(R, 128, dtype=torch.bfloat16, device=dev)
                    wext = torch.empty(Cc, 128, dtype=torch.bfloat16, device=dev)          # [W^T | b | 0], packed by the launch
                    L.call("tg_encode_ts_features", L.ptr(ts_raw), ts_raw.shape[1], src_col, L.ptr(enc.encoder_dict["timestamp"].min_year),
                           L.ptr(row_ids), L.ptr(f), R, ts_w[src_col].data_ptr(), ts_b[src_col].data_ptr(), L.ptr(wext), Cc,
                           L.stream())
                    L.call("tg_gemm_nt_bf16", L.ptr(f), L.ptr(wext), None, None, out.data_ptr() + 2 * out_col * Cc, R, Cc, 128,
                           128, S * Cc, 0, 0.0, 0, 0, L.stream())
                    ts_feats[src_col] = f
        else:
            for desc in plan["descs"]:
                L.call("tg_encode_fwd", C.addressof(desc), C.addressof(ptrs), out.data_ptr(), R, S, Cc, L.dt(out), L.stream())
        ctx.ts_feats = ts_feats if ts_gemm else None
        ctx.enc, ctx.feats, ctx.params, ctx.R, ctx.row_ids = enc, feats, params, R, row_ids
        out._cls_base = buf
        return out

    @staticmethod
    def backward(ctx, g):
        enc, feats, params, R = ctx.enc, ctx.feats, ctx.params, ctx.R
        plan = enc._plan
        Cc = enc.out_channels
        # the gradient of a CLS-prepended tensor arrives as the [:, 1:, :] view of a [R, ncols+1, C] tensor: read in place
        row_cols = plan["ncols"]
        if g.dim() == 3 and g.stride() == ((row_cols + 1) * Cc, Cc, 1) and g.data_ptr() % 16 == 0:
            row_cols += 1
        else:
            g = g.contiguous()
        dev = g.device
        ptrs = enc._ptrs(feats, params, ctx.row_ids)
        num_w, num_b, cat_table, ts_w, ts_b, rel_w, rel_b = params
        nblk = L.load().tg_encode_bwd_blocks()
        ts_feats = ctx.ts_feats

        def run_group(gi, desc, acc_floats, dflat, partials, big_grad):
            """the reduced gradient vector of column group gi into dflat (layout of plan['segments'][gi])"""
            if ts_feats is None:
                L.call("tg_encode_bwd", C.addressof(desc), C.addressof(ptrs), g.data_ptr(), R, row_cols, Cc, acc_floats,
                       L.ptr(dflat), L.ptr(partials), big_grad, L.dt(g), L.stream())
                return
            dn, acc_gen, ts_cols = plan["nots"][gi]
            if dn.ncol > 0:
                L.call("tg_encode_bwd", C.addressof(dn), C.addressof(ptrs), g.data_ptr(), R, row_cols, Cc, acc_gen,
                       L.ptr(dflat), L.ptr(partials), big_grad, L.dt(g), L.stream())
            for src_col, out_col, off in ts_cols:       # [W | b] gradient = g_col^T feats  (column 56 of feats is 1)
                f = ts_feats[src_col]
                dw = torch.empty(Cc, 128, dtype=torch.float32, device=dev)
                ws = torch.empty(max(L.load().tg_gemm_tn_workspace_floats(R, Cc, 128), 1), dtype=torch.float32, device=dev)
                L.call("tg_gemm_tn_bf16", g.data_ptr() + 2 * out_col * Cc, L.ptr(f), L.ptr(dw), None, L.ptr(ws), R, Cc, 128,
                       row_cols * Cc, 128, 0, L.stream())
                dflat[off:off + 57 * Cc].view(57, Cc).copy_(dw[:, :57].t())

        big = plan["big"]                 # categorical columns whose table exceeds the LDS accumulators

        def big_tables(dst_of):
            """Their gradients in a fixed order (tg_embed_grad_sorted): counting sort of the batch's (column, category)
            pairs, one wave per bucket.  ``dst_of(src_col)`` = fp32 gradient tensor [rows, C] of that column's table."""
            if not big or R == 0:
                return
            cat = feats[stype.categorical]
            rows_sel = cat if ctx.row_ids is None else cat.index_select(0, ctx.row_ids)
            cols_t = torch.tensor([c["src_col"] for c in big], dtype=torch.int64, device=dev)
            lim = torch.tensor([c["rows"] - 1 for c in big], dtype=torch.int64, device=dev)
            base = torch.tensor([c["base"] for c in big], dtype=torch.int64, device=dev)
            idx = (rows_sel.index_select(1, cols_t) + 1).clamp_(min=0)
            keys = (torch.minimum(idx, lim) + base).t().contiguous().to(torch.int32).reshape(-1)       # [nbig * R], slot-major
            nb = big[-1]["base"] + big[-1]["rows"]
            rowptr = torch.empty(nb + 1, dtype=torch.int32, device=dev)
            perm = torch.empty(keys.numel(), dtype=torch.int32, device=dev)
            work = torch.empty(L.load().tg_csr_workspace_ints(keys.numel(), nb), dtype=torch.int32, device=dev)
            L.call("tg_csr_build", L.ptr(keys), keys.numel(), nb, L.ptr(rowptr), L.ptr(perm), L.ptr(work), L.stream())
            dsts = [dst_of(c["src_col"]) for c in big]
            table = torch.tensor([[c["base"], c["out_col"], d.data_ptr(), c["rows"]] for c, d in zip(big, dsts)],
                                 dtype=torch.int64, device=dev)
            L.call("tg_embed_grad_sorted", g.data_ptr(), row_cols * Cc, L.ptr(rowptr), L.ptr(perm), R, L.ptr(table), len(big),
                   max(c["rows"] for c in big), Cc, 1, L.dt(g), L.stream())

        seg_tables = enc._grad_segment_tables(dev)
        if seg_tables is not None:
            # every encoder parameter already owns a gradient buffer: the reduced vector of each column group is
            # added into them by one scatter-add launch (no zero-fill / slice copy / autograd add per column)
            for gi, (desc, acc_floats, (table, nseg, max_len)) in enumerate(zip(plan["descs"], plan["acc_floats"], seg_tables)):
                dflat = torch.empty(max(acc_floats, 1), dtype=torch.float32, device=dev)
                partials = torch.empty(nblk * max(acc_floats, 1), dtype=torch.float32, device=dev)
                run_group(gi, desc, acc_floats, dflat, partials, None)
                L.call("tg_scatter_add_segments", L.ptr(dflat), L.ptr(table), nseg, max_len, L.stream())
            embs = enc.encoder_dict["categorical"].embs if big else None
            big_tables(lambda j: embs[j].weight.grad)          # added straight into the parameters' gradient buffers
            return (None,) * (4 + len(params))
        grads = [None if p is None else torch.zeros_like(p) for p in params]
        for gi, (desc, acc_floats, segs) in enumerate(zip(plan["descs"], plan["acc_floats"], plan["segments"])):
            dflat = torch.empty(max(acc_floats, 1), dtype=torch.float32, device=dev)
            partials = torch.empty(nblk * max(acc_floats, 1), dtype=torch.float32, device=dev)
            run_group(gi, desc, acc_floats, dflat, partials, None)
            for kind, src_col, off, rows, tab_off in segs:
                if kind == 0:
                    grads[0][src_col] = dflat[off:off + Cc]; grads[1][src_col] = dflat[off + Cc:off + 2 * Cc]
                elif kind == 3:
                    grads[5][src_col] = dflat[off:off + Cc]; grads[6][src_col] = dflat[off + Cc:off + 2 * Cc]
                elif kind == 2:
                    grads[3][src_col] = dflat[off:off + 56 * Cc].view(TS_FIELDS, TS_OUT, Cc)
                    grads[4][src_col] = dflat[off + 56 * Cc:off + 57 * Cc]
                elif off >= 0:
                    grads[2][tab_off:tab_off + rows] = dflat[off:off + rows * Cc].view(rows, Cc)
        tab = {c["src_col"]: c["tab_off"] for c in big}
        rws = {c["src_col"]: c["rows"] for c in big}
        big_tables(lambda j: grads[2][tab[j]:tab[j] + rws[j]])     # (grads[2] starts zeroed: contiguous row slices of it)
        return (None, None, None, None, *grads)


class StypeWiseFeatureEncoder(nn.Module):
    """``encoder(tf) -> (x [R, ncols, C], col_names)``; columns ordered numerical, categorical, timestamp, relation."""

    def __init__(self, out_channels, col_stats, col_names_dict, compute_dtype=torch.float32):
        """col_stats[name]: numerical {'mean','std'}; categorical {'cardinality'}; timestamp {'min_year'}."""
        super().__init__()
        self.out_channels = out_channels
        self.col_names_dict = {s: list(col_names_dict[s]) for s in STYPE_ORDER if s in col_names_dict}
        self.compute_dtype = compute_dtype
        self.encoder_dict = nn.ModuleDict()
        for s, names in self.col_names_dict.items():
            if s == stype.numerical:
                m = LinearEncoder(out_channels, [col_stats[n]["mean"] for n in names], [col_stats[n]["std"] for n in names])
            elif s == stype.categorical:
                m = EmbeddingEncoder(out_channels, [col_stats[n]["cardinality"] for n in names])
            elif s == stype.timestamp:
                m = TimestampEncoder(out_channels, [col_stats[n]["min_year"] for n in names])
            else:
                m = ProjectionEncoder(out_channels, len(names))
            self.encoder_dict[s.value] = m
        self._plan = self._make_plan()

    def reset_parameters(self):
        for m in self.encoder_dict.values():
            m.reset_parameters()

    # ------------------------------------------------------------------ launch plan (host structs)
    def _make_plan(self):
        lib = L.load()
        max_cols, small = lib.tg_encode_max_cols(), lib.tg_encode_small_table_rows()
        Cc = self.out_channels
        cols, out_col, tab_off = [], 0, 0
        for s, names in self.col_names_dict.items():
            for i, _ in enumerate(names):
                rows = 0
                toff = 0
                if s == stype.categorical:
                    rows = self.encoder_dict["categorical"].embs[i].num_embeddings
                    toff = tab_off
                    tab_off += rows
                cols.append(dict(kind=KIND[s], out_col=out_col, src_col=i, rows=rows, tab_off=toff))
                out_col += 1
        budget = 100 * 1024 // 4          # floats of LDS accumulators per launch (feats + acc <= 150 KiB)
        groups, cur, cur_f = [], [], 0
        for c in cols:
            need = {0: 2 * Cc, 3: 2 * Cc, 2: 57 * Cc}.get(c["kind"], c["rows"] * Cc if c["rows"] <= small else 0)
            if cur and (len(cur) == max_cols or cur_f + need > budget):
                groups.append(cur); cur, cur_f = [], 0
            cur.append(c); cur_f += need
        if cur:
            groups.append(cur)
        descs, accs, segments, nots = [], [], [], []
        for grp in groups:
            grp = sorted(grp, key=lambda c: c["kind"] == 2)      # timestamp columns last: tg_encode_bwd's generic kernel
            d = L.EncDesc()                                       # then skips their accumulator slots and work items
            d.ncol = len(grp)
            off, nts, segs = 0, 0, []
            for j, c in enumerate(grp):
                e = d.col[j]
                e.kind, e.out_col, e.src_col, e.rows, e.tab_off = c["kind"], c["out_col"], c["src_col"], c["rows"], c["tab_off"]
                e.ts_slot = 0
                if c["kind"] == 2:
                    e.ts_slot = nts; nts += 1
                    need = 57 * Cc
                elif c["kind"] == 1:
                    need = c["rows"] * Cc if c["rows"] <= small else 0
                else:
                    need = 2 * Cc
                e.acc_off = off if need > 0 else -1
                segs.append((c["kind"], c["src_col"], e.acc_off, c["rows"], c["tab_off"]))
                off += need
            d.nts = nts
            descs.append(d); accs.append(off); segments.append(segs)
            # the same group without its (trailing) timestamp columns, for the GEMM route of those columns (bf16, C = 128):
            # (descriptor, floats of the reduced vector the generic kernel owns, [(src_col, out_col, acc_off)] of the rest)
            dn = L.EncDesc()
            C.memmove(C.addressof(dn), C.addressof(d), C.sizeof(d))
            dn.ncol, dn.nts = len(grp) - nts, 0
            ts_cols = [(c["src_col"], c["out_col"], d.col[j].acc_off) for j, c in enumerate(grp) if c["kind"] == 2]
            nots.append((dn, min([t[2] for t in ts_cols], default=off), ts_cols))
        big, base = [], 0
        for c in cols:
            if c["kind"] == 1 and c["rows"] > small:
                big.append(dict(src_col=c["src_col"], out_col=c["out_col"], rows=c["rows"], tab_off=c["tab_off"], base=base))
                base += c["rows"]
        return dict(ncols=out_col, descs=descs, acc_floats=accs, segments=segments, nots=nots, big=big)

    def _grad_segment_tables(self, dev):
        """Per column group: (device int64 table [nseg,3] = (gradient-buffer address, offset in the group's reduced
        vector, length), nseg, longest segment) — or None when some parameter has no gradient buffer yet.  Categorical
        tables too large for the LDS accumulators are not listed: ``tg_embed_grad_sorted`` adds their gradients."""
        ed, Cc = self.encoder_dict, self.out_channels

        def grad_of(p):
            g = p.grad
            return g if (g is not None and g.dtype == torch.float32 and g.is_contiguous() and g.device == dev) else None
        src = {}
        if "numerical" in ed:
            src[0] = (grad_of(ed["numerical"].weight), grad_of(ed["numerical"].bias))
        if "relation" in ed:
            src[3] = (grad_of(ed["relation"].weight), grad_of(ed["relation"].bias))
        if "timestamp" in ed:
            src[2] = (grad_of(ed["timestamp"].weight), grad_of(ed["timestamp"].bias))
        embs = [grad_of(e.weight) for e in ed["categorical"].embs] if "categorical" in ed else []
        if any(g is None for pair in src.values() for g in pair) or any(g is None for g in embs):
            return None
        key = tuple(g.data_ptr() for pair in src.values() for g in pair) + tuple(g.data_ptr() for g in embs)
        cache = self.__dict__.setdefault("_seg_cache", {})
        if key in cache:
            return cache[key]
        tables = []
        for segs in self._plan["segments"]:
            rows_ = []
            for kind, src_col, off, rows, tab_off in segs:
                if kind in (0, 3):
                    w, b = src[kind]
                    rows_ += [(w[src_col].data_ptr(), off, Cc), (b[src_col].data_ptr(), off + Cc, Cc)]
                elif kind == 2:
                    w, b = src[2]
                    rows_ += [(w[src_col].data_ptr(), off, 56 * Cc), (b[src_col].data_ptr(), off + 56 * Cc, Cc)]
                elif off >= 0:
                    rows_.append((embs[src_col].data_ptr(), off, rows * Cc))
                # (off < 0: a table too large for the LDS accumulators — reduced by tg_embed_grad_sorted, not listed here)
            table = torch.tensor(rows_, dtype=torch.int64, device=dev).reshape(-1, 3) if rows_ else \
                torch.zeros(0, 3, dtype=torch.int64, device=dev)
            tables.append((table, len(rows_), max([r[2] for r in rows_], default=1)))
        cache[key] = tables
        return tables

    def _cat_table(self):
        embs = self.encoder_dict["categorical"].embs
        return torch.cat([e.weight for e in embs], dim=0) if len(embs) > 1 else embs[0].weight

    def _ptrs(self, feats, params, row_ids=None):
        num_w, num_b, cat_table, ts_w, ts_b, rel_w, rel_b = params
        p = L.EncPtrs()
        if row_ids is not None:
            if row_ids.dtype != torch.int64 or not row_ids.is_contiguous():
                raise RuntimeError("TensorFrame.row_ids must be a contiguous int64 tensor")
            p.row_ids = L.ptr(row_ids)
        ed = self.encoder_dict

        def raw(s, dtype):
            if s not in feats:
                return None, 0
            t = feats[s]
            if t.dtype != dtype or not t.is_contiguous():
                t = t.to(dtype).contiguous()
                feats[s] = t
            return L.ptr(t), t.shape[1]
        p.num, p.nn = raw(stype.numerical, torch.float32)
        p.cat, p.nc = raw(stype.categorical, torch.int64)
        p.ts, p.nt = raw(stype.timestamp, torch.int64)
        p.rel, p.nr = raw(stype.relation, torch.float32)
        if "numerical" in ed:
            p.num_mean, p.num_std = L.ptr(ed["numerical"].mean), L.ptr(ed["numerical"].std)
            p.num_w, p.num_b = L.ptr(num_w), L.ptr(num_b)
        if "categorical" in ed:
            p.cat_table = L.ptr(cat_table)
        if "timestamp" in ed:
            p.ts_min_year, p.ts_w, p.ts_b = L.ptr(ed["timestamp"].min_year), L.ptr(ts_w), L.ptr(ts_b)
        if "relation" in ed:
            p.rel_w, p.rel_b = L.ptr(rel_w), L.ptr(rel_b)
        return p

    def forward(self, tf: TensorFrame):
        ed = self.encoder_dict
        feats = {s: tf.feat_dict[s] for s in self.col_names_dict}
        params = [
            ed["numerical"].weight if "numerical" in ed else None,
            ed["numerical"].bias if "numerical" in ed else None,
            self._cat_table() if "categorical" in ed else None,
            ed["timestamp"].weight if "timestamp" in ed else None,
            ed["timestamp"].bias if "timestamp" in ed else None,
            ed["relation"].weight if "relation" in ed else None,
            ed["relation"].bias if "relation" in ed else None,
        ]
        x = _Encode.apply(self, feats, self.compute_dtype, tf.row_ids, *params)
        names = [n for s in self.col_names_dict for n in self.col_names_dict[s]]
        return x, names
