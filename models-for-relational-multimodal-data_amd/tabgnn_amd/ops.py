"""Autograd operators of the hot path: thin torch.autograd.Function shells around the C-ABI kernels.

Every operator runs on the current HIP stream through ``_lib.call`` and raises if the library or a GPU
is missing — there is no eager/CPU fallback.  Dense projections of the 128-wide shapes run on the hand-written MFMA GEMMs (gemm_nt / gemm_tn), the rest on torch's GEMM (hipBLASLt);
everything else (gather, attention core, norms, aggregation, pooling, loss, optimiser) is a HIP kernel.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

import torch

from . import _lib as L

# --------------------------------------------------------------------------- dropout RNG bookkeeping


class DropoutRNG:
    """Counter-based dropout: a mask is a pure function of (seed, stream id, element index), so the
    backward recomputes it.  ``seed`` advances once per training step, ``stream`` once per dropout site."""
    BASE_SEED = 0x5EED
    seed = 0x5EED           # < 2^63: bit 63 of a seed argument marks a device seed word (include/tabgnn_hip.h:TG_SEED_DEVICE)
    _stream = 0

    @classmethod
    def next_stream(cls):
        cls._stream = (cls._stream + 1) & 0x7FFFFFFF
        return cls._stream

    @classmethod
    def new_step(cls, seed=None):
        cls.seed = (cls.seed * 6364136223846793005 + 1442695040888963407) & ((1 << 63) - 1) if seed is None else seed
        cls._stream = 0


class StepContext:
    """Per-step device-side inputs that a captured graph reads at run time, handed to the entry points as explicit
    arguments (ABI v6: the library holds no such state): ``bn_row_limit`` = device int32 with the real row count of a padded
    batch (BatchNorm statistics, ``tg_bn_act_res_fwd/bwd``) or None.  Thread-local: ``graph_step`` sets it around its body."""
    _tl = threading.local()

    @classmethod
    def bn_row_limit(cls):
        return getattr(cls._tl, "bn_row_limit", None)

    @classmethod
    def set_bn_row_limit(cls, t):
        cls._tl.bn_row_limit = t


_index_errors = []          # device error words of the index conversions since the last take_index_errors()


def take_index_errors():
    """The out-of-range flags (int32 [1] device tensors, nonzero = some id was clamped) of every SubgraphIndex /
    SeedIndex built since the last call; train.IndexGuard turns them into a RuntimeError without a per-call sync."""
    global _index_errors
    out, _index_errors = _index_errors, []
    return out


def restore_index_errors(flags):
    """Put flags taken by ``take_index_errors`` back in front of the pending list (a caller that only wanted to look)."""
    global _index_errors
    _index_errors = list(flags) + _index_errors


class KernelTimer:
    """HIP-event timing of selected launches on the stream they run on (bench.py's live roofline measurement)."""
    active = None

    def __init__(self, only=None):
        self.spans = {}
        self.nbytes = {}
        self.only = None if only is None else set(only)      # time just these entry points (None = every _launch)
        self.units = {}                                      # work units of the timed launches (wave tiles of the encoder kernels)

    def timed(self, name, fn, nbytes=0, units=0):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        self.spans.setdefault(name, []).append((e0, e1))
        self.nbytes[name] = self.nbytes.get(name, 0) + int(nbytes)
        self.units[name] = self.units.get(name, 0) + int(units)

    def total_ms(self, name):
        return sum(a.elapsed_time(b) for a, b in self.spans.get(name, []))

    def gbs(self, name):
        """Algorithmic bytes of all timed launches of ``name`` over their summed duration."""
        t = self.total_ms(name)
        return self.nbytes.get(name, 0) / (t * 1e-3) / 1e9 if t > 0 else float("nan")

    def mean_ms(self, name):
        sp = self.spans.get(name, [])
        return sum(a.elapsed_time(b) for a, b in sp) / len(sp) if sp else float("nan")

    def count(self, name):
        return len(self.spans.get(name, []))


def _launch(name, *args, nbytes=0, units=0):
    t = KernelTimer.active
    if t is None or (t.only is not None and name not in t.only):
        L.call(name, *args)
    else:
        t.timed(name, lambda: L.call(name, *args), nbytes, units)


def _workspace(n_floats, device):
    return torch.empty(max(int(n_floats), 1), dtype=torch.float32, device=device)


# --------------------------------------------------------------------------- index structure


class SubgraphIndex:
    """int32 endpoints + stable CSR by destination and by source of one sampled subgraph
    (``edge_index`` of ``TABGNNFused.forward``, src/nn/models/fused.py:144)."""

    def __init__(self, src, dst, by_dst, by_src, num_nodes):
        self.src, self.dst, self.by_dst, self.by_src, self.N = src, dst, by_dst, by_src, num_nodes
        self.E = src.numel()

    @staticmethod
    def ids32(ids, num_nodes):
        ids = ids.contiguous()
        out = torch.empty(ids.shape, dtype=torch.int32, device=ids.device)
        err = torch.zeros(1, dtype=torch.int32, device=ids.device)
        L.call("tg_ids_to_i32", L.ptr(ids), ids.numel(), num_nodes, L.ptr(out), L.ptr(err), L.stream())
        if len(_index_errors) < 64:          # bounded: callers that never drain the list (plain inference) keep the newest few
            _index_errors.append(err)
        return out, err

    @staticmethod
    def csr(keys32, num_nodes):
        M = keys32.numel()
        dev = keys32.device
        rowptr = torch.empty(num_nodes + 1, dtype=torch.int32, device=dev)
        perm = torch.empty(max(M, 1), dtype=torch.int32, device=dev)
        work = torch.empty(L.load().tg_csr_workspace_ints(M, num_nodes), dtype=torch.int32, device=dev)
        L.call("tg_csr_build", L.ptr(keys32), M, num_nodes, L.ptr(rowptr), L.ptr(perm), L.ptr(work), L.stream())
        return rowptr, perm

    @classmethod
    def build(cls, edge_index, num_nodes, check=False):
        if isinstance(edge_index, SubgraphIndex):
            return edge_index
        if edge_index.dim() != 2 or edge_index.shape[0] != 2 or edge_index.dtype != torch.int64:
            raise RuntimeError("edge_index must be int64 [2, E]")
        ei32, err = cls.ids32(edge_index, num_nodes)
        if check and int(err.item()) != 0:
            raise RuntimeError("edge_index holds node ids outside [0, num_nodes)")
        src, dst = ei32[0], ei32[1]
        g = cls(src, dst, cls.csr(dst, num_nodes), cls.csr(src, num_nodes), num_nodes)
        g.err = err
        return g

    def flip(self):
        """The graph with every edge reversed (``edge_index.flipud()``, src/nn/gnn/pna.py:40)."""
        return SubgraphIndex(self.dst, self.src, self.by_src, self.by_dst, self.N)

    def sorted_view(self):
        """Index arrays for laying the messages out in destination-sorted (CSR) order, so the aggregation streams
        contiguous rows with no indirection: row k of the sorted layout is edge ``perm[k]``."""
        sv = getattr(self, "_sorted", None)
        if sv is None:
            perm = self.by_dst[1][:self.E].long()
            inv = torch.empty(self.E, dtype=torch.int32, device=perm.device)
            inv[perm] = torch.arange(self.E, dtype=torch.int32, device=perm.device)
            sv = dict(perm=self.by_dst[1], dst=self.dst[perm].contiguous(), src=self.src[perm].contiguous(), inv=inv,
                      src_to_sorted=inv[self.by_src[1][:self.E].long()].contiguous())
            self._sorted = sv
        return sv


class SeedIndex:
    """Seed edges ``target_edge_index [2,B]``: int32 endpoints and the CSR over the 2B endpoint slots
    (replaces ``torch.unique(..., return_inverse=True)`` + ``bincount``, fused.py:261-266)."""

    def __init__(self, target_edge_index, num_nodes):
        if isinstance(target_edge_index, SeedIndex):
            self.__dict__.update(target_edge_index.__dict__)
            return
        t32, self.err = SubgraphIndex.ids32(target_edge_index, num_nodes)
        self.tei = t32.reshape(-1)               # [2B]: sources then destinations
        self.src, self.dst = t32[0], t32[1]
        self.B = t32.shape[1]
        self.N = num_nodes
        self.rowptr, self.perm = SubgraphIndex.csr(self.tei, num_nodes)

    @classmethod
    def from_parts(cls, tei32, rowptr, perm, B, num_nodes):
        """From structures built elsewhere (the sampler's host thread, sampler.batch_index)."""
        self = cls.__new__(cls)
        self.tei, self.src, self.dst, self.B, self.N = tei32, tei32[:B], tei32[B:], B, num_nodes
        self.rowptr, self.perm, self.err = rowptr, perm, None
        return self


class BatchIndex:
    """A sampled batch's ``edge_index`` together with its prebuilt index structures: ``graph`` (SubgraphIndex of the
    neighbour edges, columns B: ) and ``seeds`` (SeedIndex of the first B columns).  Accepted wherever the wrappers take
    ``edge_index``; ``.edge_index`` is the plain int64 [2,E] tensor."""

    def __init__(self, graph, seeds, edge_index):
        self.graph, self.seeds, self.edge_index = graph, seeds, edge_index

    @property
    def shape(self):
        return self.edge_index.shape


class GradSink:
    """One gradient buffer shared by ALL consumers of a tensor that fans out to several of the operators below (the
    layer input x feeds the message gather, the post projection and the residual; the edge embedding feeds two gathers
    and the edge update).  Autograd would materialise one gradient per consumer and add them pairwise (a read-read-write
    pass over [N,F] or [E,F] each); with a sink every consumer's backward kernel adds its part into the same buffer
    (first user writes, later ones accumulate in their epilogue) and only the first user hands the buffer to autograd,
    the others return None.  Contract: EVERY autograd consumer of the tensor must be given the sink (a consumer that
    returns its own tensor would make autograd sum a snapshot), one backward per forward.  ``TABGNN_NO_GRAD_SINK=1``
    turns the mechanism off (the layers then pass no sinks)."""

    __slots__ = ("buf",)

    def __init__(self):
        self.buf = None

    def target(self, shape, dtype, device):
        """(buffer, accumulate flag, hand the buffer to autograd?)"""
        if self.buf is None:
            self.buf = torch.empty(shape, dtype=dtype, device=device)
            return self.buf, 0, True
        if self.buf.shape != torch.Size(shape) or self.buf.dtype != dtype:
            raise RuntimeError("GradSink used for tensors of different shapes")
        return self.buf, 1, False


GRAD_SINKS = os.environ.get("TABGNN_NO_GRAD_SINK") != "1"


def new_sink(enabled=True):
    return GradSink() if (GRAD_SINKS and enabled) else None


def _sink_target(sink, shape, dtype, device):
    if sink is None:
        return torch.empty(shape, dtype=dtype, device=device), 0, True
    return sink.target(shape, dtype, device)


# --------------------------------------------------------------------------- dense projection


NT_RELU, NT_DROPOUT, NT_ACCUM, NT_GATE, NT_LEAKY = 1, 2, 4, 8, 16


# A/B: every N, K multiple of 128 on the hand-written kernel.  Measured in round 5 on configs[4]'s batch shape (same box, after
# post_scaled_ok stopped leaning on this policy): 34.3-34.4 ms/step against 33.5-33.8 with N, K > 128 on hipBLASLt's 256 x 256
# tiles — and the large C = 256 products (QKV, out-proj, FFN of the op-by-op column transformer) do not come through nt_ok
# at all (encoder_layer takes its hand-written route at C = 128 only), so the default stays with the faster library there.
_NT_ANY = os.environ.get("TABGNN_NT_ANY") == "1"


def nt_ok(x2, N, K):
    """Shapes the hand-written MFMA GEMM (tg_gemm_nt_bf16) takes: bf16 rows, N and K multiples of 128; measured at or
    above the library GEMM for N == 128 (any K) and for K == 128 (any N: the QKV projection, edge_emb's dX); 2 % behind it
    on configs[4]'s step for N, K = 256 .. 768 (TABGNN_NT_ANY=1 routes those here too)."""
    return (x2.dtype == torch.bfloat16 and x2.is_cuda and N % 128 == 0 and K % 128 == 0 and x2.shape[0] > 0
            and (N == 128 or K == 128 or _NT_ANY) and x2.stride(1) == 1 and x2.stride(0) % 8 == 0
            and x2.data_ptr() % 16 == 0)


def gemm_nt(x2, w, bias=None, flags=0, p=0.0, seed=0, rs=0, out=None, gate=None):
    """out[R,N] = epilogue(x2[R,K] w[N,K]^T)  (bf16; bias fp32 [N]; flags NT_RELU | NT_DROPOUT | NT_ACCUM | NT_GATE;
    ``gate`` [R,N]: the saved output of a drop(relu(.)) whose backward the epilogue applies)."""
    R, K = x2.shape
    N = w.shape[0]
    w = w.contiguous()
    if out is None:
        out = torch.empty(R, N, dtype=x2.dtype, device=x2.device)
    if bias is not None and bias.dtype != torch.float32:
        bias = bias.float()
    # algorithmic bytes (bench.py's live roofline of the GEMMs): X in, Y out, + the gate / the old Y when read
    nb = 2 * R * (K + N + (N if flags & NT_GATE else 0) + (N if flags & NT_ACCUM else 0))
    _launch("tg_gemm_nt_bf16", x2.data_ptr(), L.ptr(w), L.ptr(bias), L.ptr(gate), L.ptr(out), R, N, K, x2.stride(0),
            out.stride(0), int(flags), float(p), int(seed), int(rs), L.stream(), nbytes=nb)
    return out


def wt(lp, param=None):
    """W^T [in, out] as a row-major tensor for the input-gradient GEMM: the transposed shadow FlatParams keeps up to
    date (one batched transpose per optimiser step) when the parameter has one, else a transposed view (gemm_nt then
    makes it contiguous with a small kernel)."""
    t = getattr(param, "_lp_t", None) if param is not None else None
    if t is not None and t.dtype == lp.dtype and t.shape == (lp.shape[1], lp.shape[0]) and t.device == lp.device:
        _fresh(param)
        return t
    return lp.t()


def _fresh(p):
    """Refresh the FlatParams shadows when ``p`` was written through torch since they were taken (checkpoint restore
    on a sub-module, ``p.copy_()``, a torch optimiser): the masters' version counter moved, the shadows' stamp did not."""
    if p._version != getattr(p, "_lp_ver", p._version):
        flat = p._flat_ref() if getattr(p, "_flat_ref", None) is not None else None
        if flat is not None:
            flat.refresh_shadow()


def gemm_nt_ln(x2, w, bias, res, gamma, beta, p=0.0, seed=0, rs=0, eps=1e-5):
    """(z, out, stats): z = res + drop(x2 w^T + bias), out = LayerNorm(z) * gamma + beta  (bf16, d_model = 128)."""
    R, K = x2.shape
    w = w.contiguous()
    z = torch.empty(R, 128, dtype=x2.dtype, device=x2.device)
    out = torch.empty_like(z)
    stats = torch.empty(R, 2, dtype=torch.float32, device=x2.device)
    L.call("tg_gemm_nt_ln_bf16", x2.data_ptr(), L.ptr(w), L.ptr(bias), L.ptr(res), L.ptr(gamma), L.ptr(beta), L.ptr(z),
           L.ptr(out), L.ptr(stats), R, K, x2.stride(0), eps, float(p), int(seed), int(rs), L.stream())
    return z, out, stats


class _Linear(torch.autograd.Function):
    """y = x W^T + b.  bf16 problems of the shapes ``nt_ok`` lists run on the hand-written MFMA kernel
    (tg_gemm_nt_bf16), the rest on torch's library GEMM; W/b are fp32 masters, ``w_lp``/``b_lp`` their compute-dtype
    shadows (bf16 mode).  Weight gradients are returned in fp32 (or accumulated in place, see weight_grad)."""

    @staticmethod
    def forward(ctx, x, weight, bias, w_lp, b_lp):
        w = weight if w_lp is None else w_lp
        x2 = x.reshape(-1, x.shape[-1])
        if nt_ok(x2, w.shape[0], w.shape[1]) and w.dtype == x2.dtype:
            y = gemm_nt(x2, w, bias.detach() if bias is not None else None)      # fp32 bias in the epilogue
        else:
            b = bias if bias is None or w_lp is None else (b_lp if b_lp is not None else shadow(bias, x2.dtype))
            y = torch.addmm(b, x2, w.t()) if b is not None else x2 @ w.t()
        ctx.save_for_backward(x2, w)
        ctx.has_bias = bias is not None
        ctx.xshape = x.shape
        ctx.params = (weight, bias)
        return y.reshape(*x.shape[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, g):
        x2, w = ctx.saved_tensors
        g2 = g.reshape(-1, g.shape[-1])
        if not ctx.needs_input_grad[0]:
            dx = None
        elif g2.is_contiguous() and nt_ok(g2, w.shape[1], w.shape[0]) and w.dtype == g2.dtype:
            dx = gemm_nt(g2, wt(w, ctx.params[0])).reshape(ctx.xshape)   # dX = G W = G (W^T)^T: W^T is the NT weight
        else:
            dx = (g2 @ w).reshape(ctx.xshape)
        want_b = ctx.has_bias and ctx.needs_input_grad[2]
        wparam, bparam = ctx.params
        if ctx.needs_input_grad[1]:
            dw, db = weight_grad(g2, x2, want_b, wparam if isinstance(wparam, torch.nn.Parameter) else None,
                                 bparam if isinstance(bparam, torch.nn.Parameter) else None)
            if want_b and db is None and dw is not None:   # (None, None) = both accumulated in place
                db = g2.sum(0, dtype=torch.float32)
        else:
            dw, db = None, (g2.sum(0, dtype=torch.float32) if want_b else None)
        return dx, dw, db, None, None


def _grad_target(p):
    """The parameter's existing fp32 gradient buffer (a FlatParams view) when a kernel may accumulate into it."""
    if p is None:
        return None
    g = p.grad
    if g is None or g.dtype != torch.float32 or not g.is_contiguous() or g.data_ptr() % 16 != 0:
        return None
    return g


def ln_grad_targets(gamma, beta, bias):
    """(dgamma, dbeta, dbias) device pointers of the parameters' existing gradient buffers when ALL the LayerNorm's
    parameters own one (FlatParams views) — the backward kernel then adds into them and returns no gradient tensors —
    else None.  ``bias`` may be None (no bias on this path)."""
    is_p = lambda t: isinstance(t, torch.nn.Parameter)
    if not (is_p(gamma) and is_p(beta)) or (bias is not None and not is_p(bias)):
        return None
    tg = [_grad_target(gamma), _grad_target(beta), _grad_target(bias) if bias is not None else None]
    if tg[0] is None or tg[1] is None or (bias is not None and tg[2] is None):
        return None
    return tuple(None if t is None else t.data_ptr() for t in tg)


def weight_grad(g2, x2, want_bias=False, wparam=None, bparam=None):
    """(dW [M,N] fp32 = g2[R,M]^T x2[R,N], db [M] fp32 = column sums of g2 or None).  Tall-skinny bf16 problems go
    to the split-row MFMA kernel (tg_gemm_tn_bf16, bias gradient from the same LDS tiles); small or fp32 ones to
    torch's GEMM.  When ``wparam`` (``bparam``) already owns a gradient buffer, the result is ACCUMULATED into it by
    the kernel (``.grad +=`` semantics, no autograd AccumulateGrad add afterwards) and None is returned in its place."""
    R, M = g2.shape
    N = x2.shape[1]
    wg = _grad_target(wparam)
    bg = _grad_target(bparam) if want_bias else None
    if wg is not None and wg.shape != (M, N):
        wg = None
    if (g2.dtype == torch.bfloat16 and R >= 64 and M % 8 == 0 and N % 8 == 0 and g2.stride(1) == 1
            and x2.stride(1) == 1 and g2.stride(0) % 8 == 0 and x2.stride(0) % 8 == 0
            and g2.data_ptr() % 16 == 0 and x2.data_ptr() % 16 == 0):
        acc = wg is not None and (not want_bias or bg is not None)
        out = wg if acc else torch.empty(M, N, dtype=torch.float32, device=g2.device)
        db = (bg if acc else torch.empty(M, dtype=torch.float32, device=g2.device)) if want_bias else None
        ws = _workspace(L.load().tg_gemm_tn_workspace_floats(R, M, N), g2.device)
        _launch("tg_gemm_tn_bf16", g2.data_ptr(), x2.data_ptr(), L.ptr(out), L.ptr(db), L.ptr(ws), R, M, N,
                g2.stride(0), x2.stride(0), int(acc), L.stream(), nbytes=2 * R * (M + N))
        return (None, None) if acc else (out, db)
    if wg is not None and wg.dtype == g2.dtype:          # fp32 parity mode: one GEMM with beta = 1
        wg.addmm_(g2.t(), x2)
        if not want_bias:
            return None, None
        if bg is not None:
            bg.add_(g2.sum(0))
            return None, None
        return None, g2.sum(0, dtype=torch.float32)
    return (g2.t() @ x2).float(), None


def shadow(p, dtype):
    """Compute-dtype copy of a master parameter (flat bf16 shadow maintained by the optimiser when present)."""
    if p is None or dtype == torch.float32:
        return None
    lp = getattr(p, "_lp", None)
    if lp is not None and lp.dtype == dtype:
        _fresh(p)
        return lp
    return p.detach().to(dtype)


def linear(x, weight, bias=None):
    dtype = x.dtype
    # a bias without a maintained shadow is cast only if the library path needs it (the MFMA epilogue adds it in fp32)
    b_lp = shadow(bias, dtype) if bias is not None and getattr(bias, "_lp", None) is not None else None
    return _Linear.apply(x, weight, bias, shadow(weight, dtype), b_lp)


class _MLPRelu(torch.autograd.Function):
    """Linear -> ReLU -> Linear (the edge-update MLP ``gnn_edge_update``, fused.py:216-220 / tabgnn.py:174-178) as one
    node on the MFMA GEMMs: ReLU in the first GEMM's epilogue (the pre-activation never exists), its backward as the
    gate epilogue of the second layer's input-gradient GEMM, weight gradients accumulated in place."""

    @staticmethod
    def forward(ctx, x, w0, b0, w2, b2, lw0, lw2):
        x2 = x.reshape(-1, x.shape[-1])
        m = gemm_nt(x2, lw0, b0.detach(), NT_RELU)
        y = gemm_nt(m, lw2, b2.detach())
        ctx.save_for_backward(x2, m, lw0, lw2)
        ctx.params = (w0, b0, w2, b2)
        ctx.xshape = x.shape
        return y.reshape(*x.shape[:-1], lw2.shape[0])

    @staticmethod
    def backward(ctx, g):
        x2, m, lw0, lw2 = ctx.saved_tensors
        w0, b0, w2, b2 = ctx.params
        isp = lambda t: t if isinstance(t, torch.nn.Parameter) else None
        g2 = g.reshape(-1, g.shape[-1]).contiguous()
        dw2, db2 = weight_grad(g2, m, True, isp(w2), isp(b2))
        if db2 is None and dw2 is not None:
            db2 = g2.sum(0, dtype=torch.float32)
        d_pre = gemm_nt(g2, wt(lw2, w2), None, NT_GATE, 0.0, gate=m)    # (g W2) where relu was active
        dw0, db0 = weight_grad(d_pre, x2, True, isp(w0), isp(b0))
        if db0 is None and dw0 is not None:
            db0 = d_pre.sum(0, dtype=torch.float32)
        dx = gemm_nt(d_pre, wt(lw0, w0)).reshape(ctx.xshape) if ctx.needs_input_grad[0] else None
        return dx, dw0, db0, dw2, db2, None, None


def mlp_relu(x, lin0, lin2):
    """``lin2(relu(lin0(x)))`` for two ``nn.Linear`` modules; one fused node when the MFMA kernels take the shapes
    (bf16, widths multiples of 128 with a 128-wide hidden layer), the op-by-op composition otherwise."""
    x2 = x.reshape(-1, x.shape[-1])
    H, K = lin0.weight.shape
    N = lin2.weight.shape[0]
    if (lin0.bias is not None and lin2.bias is not None and H == 128 and N == 128 and nt_ok(x2, H, K)
            and x2.is_contiguous()):
        dt = x.dtype
        return _MLPRelu.apply(x, lin0.weight, lin0.bias, lin2.weight, lin2.bias, shadow(lin0.weight, dt),
                              shadow(lin2.weight, dt))
    return linear(act_dropout(linear(x, lin0.weight, lin0.bias), "relu", 0.0), lin2.weight, lin2.bias)


FUSED_HEAD = os.environ.get("TABGNN_NO_FUSED_HEAD", "0") != "1"


class _HeadMLP(torch.autograd.Function):
    """The readout MLP of ``ClassifierHead`` / ``NodeClassificationHead`` (``src/nn/gnn/decoder.py:5-32``):
    Linear(D0, 50) -> ReLU -> Dropout -> Linear(50, 25) -> ReLU -> Dropout -> Linear(25, n_classes) as one forward and one
    backward kernel (``csrc/head.hip``); same arithmetic, masks and dropout-stream order as the op-by-op composition
    ``linear / act_dropout / linear / act_dropout / linear(h.float())``.  Parameter gradients are added in place when
    every parameter owns a gradient buffer (FlatParams views), returned otherwise."""

    @staticmethod
    def forward(ctx, h, w1, b1, w2, b2, w3, b3, lw1, lw2, p):
        B, D0 = h.shape
        H1, H2, NC = lw1.shape[0], lw2.shape[0], w3.shape[0]
        z1 = torch.empty(B, H1, dtype=h.dtype, device=h.device)
        z2 = torch.empty(B, H2, dtype=h.dtype, device=h.device)
        logits = torch.empty(B, NC, dtype=torch.float32, device=h.device)
        ctx.seed, ctx.rs = DropoutRNG.seed, (DropoutRNG.next_stream(), DropoutRNG.next_stream())
        ctx.p = p
        L.call("tg_head_mlp_fwd", L.ptr(h), L.ptr(lw1), L.ptr(b1), L.ptr(lw2), L.ptr(b2), L.ptr(w3), L.ptr(b3), L.ptr(z1),
               L.ptr(z2), L.ptr(logits), B, D0, H1, H2, NC, p, ctx.seed, ctx.rs[0], ctx.rs[1], L.dt(h), L.stream())
        ctx.save_for_backward(h, z1, z2, lw1, lw2, w3)
        ctx.params = (w1, b1, w2, b2, w3, b3)
        return logits

    @staticmethod
    def backward(ctx, g):
        h, z1, z2, lw1, lw2, w3 = ctx.saved_tensors
        B, D0 = h.shape
        H1, H2, NC = lw1.shape[0], lw2.shape[0], w3.shape[0]
        g = g.contiguous().float()
        params = ctx.params
        targets = [_grad_target(q) if isinstance(q, torch.nn.Parameter) else None for q in params]
        inplace = all(t is not None and t.shape == q.shape for t, q in zip(targets, params))
        outs = targets if inplace else [torch.empty(q.shape, dtype=torch.float32, device=h.device) for q in params]
        dh = torch.empty_like(h)
        ws = _workspace(L.load().tg_head_mlp_partial_floats(D0, H1, H2, NC), h.device)
        L.call("tg_head_mlp_bwd", L.ptr(g), L.ptr(h), L.ptr(z1), L.ptr(z2), L.ptr(lw1), L.ptr(lw2), L.ptr(w3), L.ptr(dh),
               L.ptr(ws), *[L.ptr(t) for t in outs], 1 if inplace else 0, B, D0, H1, H2, NC, ctx.p, ctx.seed, ctx.rs[0],
               ctx.rs[1], L.dt(h), L.stream())
        grads = [None] * 6 if inplace else outs
        return (dh if ctx.needs_input_grad[0] else None, *grads, None, None, None)


def head_mlp_ok(mlp, h):
    """Does the fused readout kernel take this MLP (``nn.Sequential`` with Linear modules at 0, 3, 6) on this input?"""
    if not (FUSED_HEAD and h.is_cuda and h.dim() == 2 and h.is_contiguous() and h.dtype in (torch.float32, torch.bfloat16)):
        return False
    l0, l3, l6 = mlp[0], mlp[3], mlp[6]
    if l0.bias is None or l3.bias is None or l6.bias is None or h.data_ptr() % 16:
        return False
    # the kernels read W3 and the three biases as fp32 and W1 / W2 as fp32 masters (their shadows carry h's type): a head
    # cast with .half() / .to(bfloat16), or one holding non-contiguous views, takes the op-by-op path instead
    for t in (l0.weight, l3.weight, l6.weight, l0.bias, l3.bias, l6.bias):
        if t.dtype != torch.float32 or not t.is_contiguous() or not t.is_cuda:
            return False
    return bool(L.load().tg_head_mlp_supported(h.shape[1], l0.out_features, l3.out_features, l6.out_features))


def head_mlp(h, mlp, p):
    l0, l3, l6 = mlp[0], mlp[3], mlp[6]
    lw1 = shadow(l0.weight, h.dtype)
    lw2 = shadow(l3.weight, h.dtype)
    lw1 = l0.weight.detach() if lw1 is None else lw1
    lw2 = l3.weight.detach() if lw2 is None else lw2
    if lw1.data_ptr() % 16 or not lw1.is_contiguous() or not lw2.is_contiguous():
        lw1, lw2 = lw1.contiguous().clone(), lw2.contiguous()
    return _HeadMLP.apply(h, l0.weight, l0.bias, l3.weight, l3.bias, l6.weight, l6.bias, lw1, lw2, float(p))


class _MLPChain(torch.autograd.Function):
    """Linear -> act -> dropout -> ... -> Linear as ONE node on the MFMA GEMMs (the fuse MLP of the fused layer,
    fused.py:199-202: 3D -> 4*3D -> 4*3D -> 3D with LeakyReLU + Dropout; any chain whose widths are multiples of 128).
    Activation and dropout run in the producing GEMM's epilogue (the pre-activations never exist); the backward of
    act+dropout of layer i is the gate epilogue of layer i+1's input-gradient GEMM, read from the saved OUTPUT of
    layer i; weight and bias gradients are accumulated in place by the split-row kernel."""

    @staticmethod
    def forward(ctx, x, act, p, n, *wb):
        ws, bs, lws = wb[:n], wb[n:2 * n], wb[2 * n:]
        x2 = x.reshape(-1, x.shape[-1])
        aflag = NT_LEAKY if act == "leaky_relu" else NT_RELU
        ctx.seed, ctx.rs = DropoutRNG.seed, []
        hs = [x2]
        for i in range(n):
            last = i == n - 1
            rs = 0 if last else DropoutRNG.next_stream()       # one stream id per act+dropout site, as act_dropout takes
            ctx.rs.append(rs)
            flags = 0 if last else aflag | (NT_DROPOUT if p > 0.0 else 0)
            hs.append(gemm_nt(hs[-1], lws[i], bs[i].detach(), flags, 0.0 if last else p, ctx.seed, rs))
        ctx.save_for_backward(*hs[:-1], *lws)
        ctx.cfg = (aflag, p, n, x.shape)
        ctx.params = (ws, bs)
        return hs[-1].reshape(*x.shape[:-1], lws[-1].shape[0])

    @staticmethod
    def backward(ctx, g):
        aflag, p, n, xshape = ctx.cfg
        saved = ctx.saved_tensors
        hs, lws = saved[:n], saved[n:]
        ws, bs = ctx.params
        isp = lambda t: t if isinstance(t, torch.nn.Parameter) else None
        d = g.reshape(-1, g.shape[-1]).contiguous()
        dws, dbs = [None] * n, [None] * n
        for i in range(n - 1, -1, -1):
            dw, db = weight_grad(d, hs[i], True, isp(ws[i]), isp(bs[i]))
            if db is None and dw is not None:
                db = d.sum(0, dtype=torch.float32)
            dws[i], dbs[i] = dw, db
            if i > 0:            # (d W_i) through act+dropout of layer i-1, from that layer's saved output
                d = gemm_nt(d, wt(lws[i], ws[i]), None, NT_GATE | (aflag & NT_LEAKY), p, gate=hs[i])
            elif ctx.needs_input_grad[0]:
                d = gemm_nt(d, wt(lws[0], ws[0])).reshape(xshape)
            else:
                d = None
        return (d, None, None, None, *dws, *dbs, *([None] * n))


# Below ~1 k rows the 128 x 128-tile kernel leaves the chip empty (24 workgroups at the reference's 200 seed rows) and
# the library's small-tile GEMMs win: 3.26 vs 3.38 ms per replayed step at B = 200; equal at B = 8192 (18.2 ms/step).
MLP_CHAIN_MIN_ROWS = int(os.environ.get("TABGNN_MLP_CHAIN_MIN_ROWS", "1024"))


def mlp_chain(x, linears, act="leaky_relu", p_drop=0.0):
    """``lin_k(drop(act(... drop(act(lin_0(x))))))`` for ``nn.Linear`` modules: one fused node when every width is a
    multiple of 128 (bf16 on the GPU) and there are enough rows to fill the chip, the op-by-op composition otherwise."""
    x2 = x.reshape(-1, x.shape[-1])
    ok = (x2.dtype == torch.bfloat16 and x2.is_cuda and x2.is_contiguous() and x2.shape[0] > 0
          and x2.shape[0] >= MLP_CHAIN_MIN_ROWS and x2.data_ptr() % 16 == 0
          and all(l.bias is not None and l.weight.shape[0] % 128 == 0 and l.weight.shape[1] % 128 == 0 for l in linears))
    if ok:
        dt = x.dtype
        n = len(linears)
        return _MLPChain.apply(x, act, float(p_drop), n, *[l.weight for l in linears], *[l.bias for l in linears],
                               *[shadow(l.weight, dt) for l in linears])
    h = x
    for i, l in enumerate(linears):
        h = linear(h, l.weight, l.bias)
        if i + 1 < len(linears):
            h = act_dropout(h, act, p_drop)
    return h


# --------------------------------------------------------------------------- attention core


class _AttnCore(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, nhead, p_drop):
        R, S, C3 = qkv.shape
        C = C3 // 3
        qkv = qkv.contiguous()
        out = torch.empty(R, S, C, dtype=qkv.dtype, device=qkv.device)
        lse = torch.empty(R, nhead, S, dtype=torch.float32, device=qkv.device)
        ctx.seed, ctx.rs = DropoutRNG.seed, DropoutRNG.next_stream()
        L.call("tg_attn_fwd", L.ptr(qkv), L.ptr(out), L.ptr(lse), R, S, C, nhead, p_drop, ctx.seed, ctx.rs,
               L.dt(qkv), L.stream())
        ctx.save_for_backward(qkv, out, lse)
        ctx.cfg = (R, S, C, nhead, p_drop)
        return out

    @staticmethod
    def backward(ctx, g):
        qkv, out, lse = ctx.saved_tensors
        R, S, C, H, p = ctx.cfg
        dqkv = torch.empty_like(qkv)
        L.call("tg_attn_bwd", L.ptr(qkv), L.ptr(out), L.ptr(g.contiguous()), L.ptr(lse), L.ptr(dqkv), R, S, C, H, p,
               ctx.seed, ctx.rs, L.dt(qkv), L.stream())
        return dqkv, None, None


def attention_core(qkv, nhead, p_drop=0.0):
    return _AttnCore.apply(qkv, nhead, float(p_drop))


# --------------------------------------------------------------------------- LayerNorm (+pre-add, +combine)


class _LayerNorm(torch.autograd.Function):
    """out = alpha*res + beta_c*LN(a + dropout(b + bias_b))"""

    @staticmethod
    def forward(ctx, a, b, bias_b, gamma, beta, res, eps, alpha, beta_c, p_drop):
        C = a.shape[-1]
        a = a.contiguous()
        M = a.numel() // C
        b = b.contiguous() if b is not None else None
        res = res.contiguous() if res is not None else None
        out = torch.empty_like(a)
        stats = torch.empty(M, 2, dtype=torch.float32, device=a.device)
        ctx.seed, ctx.rs = DropoutRNG.seed, DropoutRNG.next_stream()
        L.call("tg_ln_fwd", L.ptr(a), L.ptr(b), L.ptr(bias_b), L.ptr(gamma), L.ptr(beta), L.ptr(res), L.ptr(out),
               L.ptr(stats), M, C, eps, alpha, beta_c, p_drop, ctx.seed, ctx.rs, L.dt(a), L.stream())
        ctx.save_for_backward(a, b, bias_b, gamma, stats)
        ctx.cfg = (M, C, alpha, beta_c, p_drop, res is not None)
        ctx.params = (gamma, beta, bias_b)
        return out

    @staticmethod
    def backward(ctx, g):
        a, b, bias_b, gamma, stats = ctx.saved_tensors
        M, C, alpha, beta_c, p_drop, has_res = ctx.cfg
        g = g.contiguous()
        da = torch.empty_like(a)
        db = torch.empty_like(a) if b is not None else None
        dres = torch.empty_like(a) if has_res else None
        partials = _workspace(L.load().tg_ln_partials_floats(M, C), a.device)
        tg = ln_grad_targets(*ctx.params)
        dparams = None if tg else torch.empty(3, C, dtype=torch.float32, device=a.device)
        L.call("tg_ln_bwd", L.ptr(a), L.ptr(b), L.ptr(bias_b), L.ptr(gamma), L.ptr(stats), L.ptr(g), L.ptr(da),
               L.ptr(db), L.ptr(dres), L.ptr(dparams), L.ptr(partials), M, C, alpha, beta_c, p_drop, ctx.seed, ctx.rs,
               0, *(tg or (None, None, None)), L.dt(a), L.stream())
        if tg:
            return da, db, None, None, None, dres, None, None, None, None
        dbias = dparams[2] if bias_b is not None else None
        return da, db, dbias, dparams[0], dparams[1], dres, None, None, None, None


def layer_norm(a, gamma, beta, b=None, bias_b=None, res=None, eps=1e-5, alpha=0.0, beta_c=1.0, p_drop=0.0):
    return _LayerNorm.apply(a, b, bias_b, gamma, beta, res, eps, float(alpha), float(beta_c), float(p_drop))


# --------------------------------------------------------------------------- BatchNorm + ReLU + residual


class _BatchNormActRes(torch.autograd.Function):
    """out = alpha*res + beta_c*relu(BN(x)); updates running statistics in training mode.  ``group`` (a
    torch.distributed process group, training only) synchronises the batch statistics across its ranks: the
    [2F]+1 vector (sum x, sum x^2, rows) is all-reduced between the statistics and the apply kernels, and the
    backward does the same with (sum dz, sum dz*xhat)  (optional SyncBatchNorm)."""

    @staticmethod
    def forward(ctx, x, res, gamma, beta, running_mean, running_var, training, momentum, eps, relu, alpha, beta_c, group,
                sink_res=None):
        ctx.sink_res = sink_res
        x = x.contiguous()
        N, F = x.shape
        res = res.contiguous() if res is not None else None
        out = torch.empty_like(x)
        mean = torch.empty(F, dtype=torch.float32, device=x.device)
        rstd = torch.empty(F, dtype=torch.float32, device=x.device)
        partials = _workspace(L.load().tg_bn_partials_floats(N, F) + 8, x.device)
        args = (L.ptr(x), L.ptr(res), L.ptr(gamma), L.ptr(beta), L.ptr(running_mean), L.ptr(running_var), L.ptr(mean),
                L.ptr(rstd), L.ptr(out), L.ptr(partials), N, F, int(training), momentum, eps, int(relu), alpha, beta_c)
        sync = bool(training) and group is not None
        n_stat = N
        if sync:
            import torch.distributed as dist
            L.call("tg_bn_act_res_fwd", *args, 0, 1, None, L.dt(x), L.stream())
            vec = partials[512 * 2 * F:512 * 2 * F + 2 * F + 1]      # (sum x, sum x^2) + one slot for the row count
            vec[2 * F] = float(N)
            dist.all_reduce(vec, group=group)
            n_stat = int(round(float(vec[2 * F])))
            L.call("tg_bn_act_res_fwd", *args, n_stat, 2, None, L.dt(x), L.stream())
        else:
            # padded batch of a shape bucket: the statistics stop at the device-side real row count (forward AND backward)
            ctx.row_limit = StepContext.bn_row_limit() if training else None
            L.call("tg_bn_act_res_fwd", *args, 0, 0, L.ptr(ctx.row_limit), L.dt(x), L.stream())
        ctx.save_for_backward(x, gamma, beta, mean, rstd)
        ctx.cfg = (N, F, int(training), int(relu), alpha, beta_c, res is not None, group if sync else None, n_stat)
        return out

    @staticmethod
    def backward(ctx, g):
        x, gamma, beta, mean, rstd = ctx.saved_tensors
        N, F, training, relu, alpha, beta_c, has_res, group, n_stat = ctx.cfg
        g = g.contiguous()
        dx = torch.empty_like(x)
        dres, ret_res, add_res = None, True, None
        if has_res:
            dres, acc, ret_res = _sink_target(ctx.sink_res, x.shape, x.dtype, x.device)
            if acc:                      # not the sink's first user (never the case in the fused layer): add afterwards
                add_res, dres = dres, torch.empty_like(x)
        dparams = torch.empty(2, F, dtype=torch.float32, device=x.device)
        partials = _workspace(L.load().tg_bn_partials_floats(N, F), x.device)
        head = (L.ptr(x), L.ptr(g), L.ptr(gamma), L.ptr(beta), L.ptr(mean), L.ptr(rstd), L.ptr(dx), L.ptr(dres))
        tail = (L.ptr(partials), N, F, training, relu, alpha, beta_c)
        if group is not None:
            import torch.distributed as dist
            L.call("tg_bn_act_res_bwd", *head, L.ptr(dparams), *tail, 0, 1, None, L.dt(x), L.stream())
            glob = dparams.clone()                 # parameter gradients stay LOCAL sums (the DP all-reduce averages them)
            dist.all_reduce(glob, group=group)
            L.call("tg_bn_act_res_bwd", *head, L.ptr(glob), *tail, n_stat, 2, None, L.dt(x), L.stream())
        else:
            L.call("tg_bn_act_res_bwd", *head, L.ptr(dparams), *tail, 0, 0, L.ptr(ctx.row_limit), L.dt(x), L.stream())
        if add_res is not None:
            add_res.add_(dres)
        return (dx, dres if ret_res else None, dparams[1], dparams[0], None, None, None, None, None, None, None, None, None,
                None)


def batch_norm_act_res(x, gamma, beta, running_mean, running_var, training, res=None, momentum=0.1, eps=1e-5,
                       relu=True, alpha=0.0, beta_c=1.0, group=None, sink_res=None):
    return _BatchNormActRes.apply(x, res, gamma, beta, running_mean, running_var, bool(training), float(momentum),
                                  float(eps), bool(relu), float(alpha), float(beta_c), group, sink_res)


# --------------------------------------------------------------------------- activation + dropout, axpby


class _ActDropout(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, act, p_drop):
        x = x.contiguous()
        y = torch.empty_like(x)
        ctx.seed, ctx.rs = DropoutRNG.seed, DropoutRNG.next_stream()
        L.call("tg_act_dropout_fwd", L.ptr(x), L.ptr(y), x.numel(), act, p_drop, ctx.seed, ctx.rs, L.dt(x), L.stream())
        ctx.save_for_backward(x)
        ctx.cfg = (act, p_drop)
        return y

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        act, p = ctx.cfg
        dx = torch.empty_like(x)
        L.call("tg_act_dropout_bwd", L.ptr(x), L.ptr(g.contiguous()), L.ptr(dx), x.numel(), act, p, ctx.seed, ctx.rs,
               L.dt(x), L.stream())
        return dx, None, None


ACT = {"none": 0, "relu": 1, "leaky_relu": 2}


def act_dropout(x, act="relu", p_drop=0.0):
    if act == "none" and p_drop == 0.0:
        return x
    return _ActDropout.apply(x, ACT[act], float(p_drop))


class _Axpby(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, alpha, beta, sink_a):
        a, b = a.contiguous(), b.contiguous()
        y = torch.empty_like(a)
        L.call("tg_axpby", L.ptr(a), L.ptr(b), L.ptr(y), a.numel(), alpha, beta, L.dt(a), L.stream())
        ctx.cfg = (alpha, beta)
        ctx.sink_a = sink_a
        return y

    @staticmethod
    def backward(ctx, g):
        alpha, beta = ctx.cfg
        if ctx.sink_a is not None and ctx.needs_input_grad[0]:
            g = g.contiguous()
            buf, acc, ret = ctx.sink_a.target(g.shape, g.dtype, g.device)
            # buf (+)= alpha * g   (first user: alpha*g + 0*g, so the uninitialised buffer is never read)
            if beta == 1.0 or not ctx.needs_input_grad[1]:
                L.call("tg_axpby", L.ptr(buf if acc else g), L.ptr(g), L.ptr(buf), g.numel(), 1.0 if acc else 0.0, alpha,
                       L.dt(g), L.stream())
                gb = g
            else:                                            # buf (+)= alpha * g and gb = beta * g in one pass over g
                gb = torch.empty_like(g)
                L.call("tg_axpby2", L.ptr(buf) if acc else None, L.ptr(g), L.ptr(buf), L.ptr(gb), g.numel(), 1.0, alpha, beta,
                       L.dt(g), L.stream())
            return (buf if ret else None), gb, None, None, None
        ga = g if alpha == 1.0 else g * alpha                 # one scaled copy serves both inputs when alpha == beta
        gb = ga if beta == alpha else (g if beta == 1.0 else g * beta)
        return ga, gb, None, None, None


def axpby(a, b, alpha, beta, sink_a=None):
    """alpha*a + beta*b.  ``sink_a``: the GradSink of ``a`` (see GradSink)."""
    return _Axpby.apply(a, b, float(alpha), float(beta), sink_a)


# --------------------------------------------------------------------------- gather-concat and its backward


def _gather3(parts, rows, dtype, device):
    """parts: three (tensor, index|None, width, relu) tuples; tensor rows of stride ``tensor.stride(0)``."""
    W = sum(p[2] for p in parts)
    out = torch.empty(rows, W, dtype=dtype, device=device)
    args = []
    for i, (t, idx, w, relu) in enumerate(parts):
        args += [t.data_ptr(), L.ptr(idx), t.stride(0), w]
        if i < 2:
            args.append(int(relu))
    L.call("tg_gather_concat3", *args, L.ptr(out), rows, L.dt(out), L.stream())
    return out


class _EdgeGather(torch.autograd.Function):
    """[x[ia] | x[ib] | e]  for all edges (PNAConv.message input / edge update, fused.py:254)."""

    @staticmethod
    def forward(ctx, x, e, graph, first, sink_x=None, sink_e=None):
        # first == "dst": [x[dst], x[src], e] (message, x_i = target);  first == "src": [x[src], x[dst], e]
        # first == "dst_sorted": as "dst" but row k is edge perm[k] (destination-sorted layout for the aggregation)
        x, e = x.contiguous(), e.contiguous()
        F = x.shape[1]
        if first == "dst_sorted":
            sv = graph.sorted_view()
            parts = [(x, sv["dst"], F, 0), (x, sv["src"], F, 0), (e, sv["perm"], e.shape[1], 0)]
        else:
            ia, ib = (graph.dst, graph.src) if first == "dst" else (graph.src, graph.dst)
            parts = [(x, ia, F, 0), (x, ib, F, 0), (e, None, e.shape[1], 0)]
        out = _gather3(parts, e.shape[0], x.dtype, x.device)
        ctx.graph, ctx.first, ctx.F, ctx.We = graph, first, F, e.shape[1]
        ctx.sinks = (sink_x, sink_e)
        return out

    @staticmethod
    def backward(ctx, g):
        dx, de = _edge_gather_backward(g.contiguous(), ctx.graph, ctx.first, ctx.F, ctx.We, ctx.sinks)
        return dx, de, None, None, None, None


def _edge_gather_backward(g, graph, first, F, We, sinks):
    """(dx, de) from g = d[x[ia] | x[ib] | e[ic]] ([E, 2F+We], contiguous): the node part by a deterministic segmented
    sum over the two CSRs, the edge part as a column block (row-gathered back to edge order for the destination-sorted
    layout).  With sinks (GradSink) the parts are added into the shared buffers and None is returned for a buffer that
    another consumer already handed to autograd."""
    if first == "dst_sorted":
        sv = graph.sorted_view()
        csr_a, csr_b = (graph.by_dst[0], None), (graph.by_src[0], sv["src_to_sorted"])   # A: rows already in CSR order
    else:
        csr_a, csr_b = (graph.by_dst, graph.by_src) if first == "dst" else (graph.by_src, graph.by_dst)
    sink_x, sink_e = sinks
    dx, acc_x, ret_x = _sink_target(sink_x, (graph.N, F), g.dtype, g.device)
    hub = torch.empty(L.load().tg_segment_hub_ints(2 * graph.E), dtype=torch.int32, device=g.device)
    L.call("tg_segment_sum2", L.ptr(g), g.shape[1], 0, L.ptr(csr_a[0]), L.ptr(csr_a[1]), F, L.ptr(csr_b[0]),
           L.ptr(csr_b[1]), 0, None, L.ptr(dx), graph.N, F, L.ptr(hub), acc_x, L.dt(g), L.stream())
    tail = g[:, 2 * F:]
    if sink_e is not None and We % 8 == 0:     # the edge third goes straight into e's shared gradient buffer
        de, acc_e, ret_e = sink_e.target((g.shape[0], We), g.dtype, g.device)
        idx = sv["inv"] if first == "dst_sorted" else None
        L.call("tg_rows_add", L.ptr(de), tail.data_ptr(), L.ptr(idx), g.shape[0], We, g.stride(0), acc_e, L.dt(g),
               L.stream())
        de = de if ret_e else None
    elif first == "dst_sorted":        # de[edge] = g[inv[edge], 2F:]: one row gather back to edge order
        de = _gather3([(tail, sv["inv"], We, 0), (tail, None, 0, 0), (tail, None, 0, 0)], g.shape[0], g.dtype, g.device)
    else:
        de = tail                       # a view: autograd's accumulation reads it strided, no [E,F] copy
    return (dx if ret_x else None), de


def edge_gather(x, e, graph, first, sink_x=None, sink_e=None):
    return _EdgeGather.apply(x, e, graph, first, sink_x, sink_e)


_GATHER_GEMM = os.environ.get("TABGNN_NO_GATHER_GEMM") != "1"


def gather_gemm_ok(x, e, n_out, k_w):
    """The shapes tg_gemm_nt_gather3_bf16 / tg_gemm_tn_gather3_bf16 take: bf16 rows on the MI355X, node and edge width
    128 (three 128-column sources, K = 384), output width a multiple of 128."""
    return (_GATHER_GEMM and x.is_cuda and x.dtype == torch.bfloat16 and e.dtype == torch.bfloat16 and x.dim() == 2
            and e.dim() == 2 and x.shape[1] == 128 and e.shape[1] == 128 and k_w == 384 and n_out % 128 == 0
            and e.shape[0] > 0)


def _gather_spec(x, e, graph, first):
    """tg_gather3 of [x[ia] | x[ib] | e[ic]] (see _EdgeGather.forward for ``first``); x, e contiguous bf16."""
    if first == "dst_sorted":
        sv = graph.sorted_view()
        idx = (sv["dst"], sv["src"], sv["perm"])
    else:
        ia, ib = (graph.dst, graph.src) if first == "dst" else (graph.src, graph.dst)
        idx = (ia, ib, None)
    gs = L.Gather3()
    for c, (t, i) in enumerate(((x, idx[0]), (x, idx[1]), (e, idx[2]))):
        gs.src[c] = t.data_ptr()
        gs.idx[c] = None if i is None else i.data_ptr()
        gs.stride[c] = t.stride(0)
    return gs


def _gather_weight_grad(g2, gs, E, wparam, bparam):
    """(dW [M,384], db [M]) = (g2^T [x[ia]|x[ib]|e[ic]], column sums of g2), accumulated into the parameters' gradient
    buffers when they own one (then (None, None))."""
    M = g2.shape[1]
    wg, bg = _grad_target(wparam), _grad_target(bparam)
    acc = wg is not None and bg is not None and wg.shape == (M, 384)
    out = wg if acc else torch.empty(M, 384, dtype=torch.float32, device=g2.device)
    db = bg if acc else torch.empty(M, dtype=torch.float32, device=g2.device)
    ws = _workspace(L.load().tg_gemm_tn_gather3_workspace_floats(E, M), g2.device)
    _launch("tg_gemm_tn_gather3_bf16", g2.data_ptr(), C.byref(gs), L.ptr(out), L.ptr(db), L.ptr(ws), E, M, g2.stride(0),
            int(acc), L.stream(), nbytes=2 * E * (M + 384))
    return (None, None) if acc else (out, db)


class _GatherLinear(torch.autograd.Function):
    """y = [x[ia] | x[ib] | e[ic]] W^T + b for all edges with the concatenation gathered inside the GEMMs
    (tg_gemm_nt_gather3_bf16 forward, tg_gemm_tn_gather3_bf16 for dW): the [E,384] operand of PNAConv.message never
    exists, forward or backward; only its gradient does (the segmented sums read it)."""

    @staticmethod
    def forward(ctx, x, e, weight, bias, graph, first, w_lp, sink_x, sink_e):
        x, e = x.contiguous(), e.contiguous()
        E, N = e.shape[0], w_lp.shape[0]
        gs = _gather_spec(x, e, graph, first)
        y = torch.empty(E, N, dtype=x.dtype, device=x.device)
        _launch("tg_gemm_nt_gather3_bf16", C.byref(gs), L.ptr(w_lp.contiguous()), L.ptr(bias.detach().float()), L.ptr(y), E, N,
                N, 0, L.stream(), nbytes=2 * E * (384 + N))
        ctx.save_for_backward(x, e, w_lp)
        ctx.cfg = (graph, first, (sink_x, sink_e))
        ctx.params = (weight, bias)
        return y

    @staticmethod
    def backward(ctx, g):
        x, e, w_lp = ctx.saved_tensors
        graph, first, sinks = ctx.cfg
        weight, bias = ctx.params
        isp = lambda t: t if isinstance(t, torch.nn.Parameter) else None
        g = g.contiguous()
        dw, db = _gather_weight_grad(g, _gather_spec(x, e, graph, first), e.shape[0], isp(weight), isp(bias))
        d_cat = gemm_nt(g, wt(w_lp, weight))                                   # [E,384]
        dx, de = _edge_gather_backward(d_cat, graph, first, x.shape[1], e.shape[1], sinks)
        return dx, de, dw, db, None, None, None, None, None


def edge_linear(x, e, graph, first, weight, bias, sink_x=None, sink_e=None):
    """``linear([x[ia] | x[ib] | e[ic]], weight, bias)`` for all edges of ``graph`` (``first`` as in edge_gather)."""
    if bias is not None and gather_gemm_ok(x, e, weight.shape[0], weight.shape[1]):
        return _GatherLinear.apply(x, e, weight, bias, graph, first, shadow(weight, x.dtype), sink_x, sink_e)
    return linear(edge_gather(x, e, graph, first, sink_x, sink_e), weight, bias)


class _MLPReluGather(torch.autograd.Function):
    """lin2(relu(lin0([x[ia] | x[ib] | e[ic]]))) — the edge update (fused.py:253-254) — with the concatenation gathered
    inside the first layer's GEMMs (see _GatherLinear) and the rest as _MLPRelu."""

    @staticmethod
    def forward(ctx, x, e, w0, b0, w2, b2, graph, first, lw0, lw2, sink_x, sink_e):
        x, e = x.contiguous(), e.contiguous()
        E, H = e.shape[0], lw0.shape[0]
        gs = _gather_spec(x, e, graph, first)
        m = torch.empty(E, H, dtype=x.dtype, device=x.device)
        _launch("tg_gemm_nt_gather3_bf16", C.byref(gs), L.ptr(lw0.contiguous()), L.ptr(b0.detach().float()), L.ptr(m), E, H,
                H, NT_RELU, L.stream(), nbytes=2 * E * (384 + H))
        y = gemm_nt(m, lw2, b2.detach())
        # x is held by plain reference, not save_for_backward: the fused layer pools the seed-endpoint rows of this very
        # tensor IN PLACE afterwards (fused.py:268); the backward below first puts the pre-pool rows back (_SeedPool
        # stashed them on the tensor), which is safe because every reader of the pooled values — the next layer's
        # nodes — is upstream of this node in the backward graph and has run by then.
        # Because x is outside save_for_backward, autograd's version check does not cover it: the check is done by
        # hand in the backward against the version recorded here.
        ctx.save_for_backward(e, m, lw0, lw2)
        ctx.x_ref = x
        ctx.x_ver = x._version
        ctx.cfg = (graph, first, (sink_x, sink_e))
        ctx.params = (w0, b0, w2, b2)
        return y

    @staticmethod
    def backward(ctx, g):
        e, m, lw0, lw2 = ctx.saved_tensors
        x = ctx.x_ref
        patch = getattr(x, "_pool_patch", None)
        if x._version != ctx.x_ver:
            # the only in-place change this node can undo: ONE seed_pool(..., inplace=True, stash=True) on the tensor it
            # read (the stash records the version it saw and the version it left)
            if patch is None or patch[2] != ctx.x_ver or patch[3] != x._version:
                raise RuntimeError(
                    "edge_mlp_relu: its node input was modified in place after the forward (version "
                    f"{ctx.x_ver} -> {x._version}) and the pre-modification rows were not stashed: call "
                    "seed_pool(..., inplace=True, stash=True) exactly once on that tensor, or use inplace=False")
            with torch.no_grad():
                x.index_copy_(0, patch[0], patch[1])
            x._pool_patch = None
        graph, first, sinks = ctx.cfg
        w0, b0, w2, b2 = ctx.params
        isp = lambda t: t if isinstance(t, torch.nn.Parameter) else None
        g2 = g.contiguous()
        dw2, db2 = weight_grad(g2, m, True, isp(w2), isp(b2))
        if db2 is None and dw2 is not None:
            db2 = g2.sum(0, dtype=torch.float32)
        d_pre = gemm_nt(g2, wt(lw2, w2), None, NT_GATE, 0.0, gate=m)            # (g W2) where relu was active
        dw0, db0 = _gather_weight_grad(d_pre, _gather_spec(x, e, graph, first), e.shape[0], isp(w0), isp(b0))
        d_cat = gemm_nt(d_pre, wt(lw0, w0))                                     # [E,384]
        dx, de = _edge_gather_backward(d_cat, graph, first, x.shape[1], e.shape[1], sinks)
        return dx, de, dw0, db0, dw2, db2, None, None, None, None, None, None


def edge_mlp_rereads_x(x, e, lin0, lin2):
    """True when ``edge_mlp_relu`` takes the gather-fused route, whose backward re-reads ``x`` (the ONE predicate both
    ``edge_mlp_relu`` and the caller that pools ``x`` in place afterwards use: models.FTTransformerPNAFusedLayer)."""
    H, K = lin0.weight.shape
    return bool(lin0.bias is not None and lin2.bias is not None and H == 128 and lin2.weight.shape[0] == 128
                and gather_gemm_ok(x, e, H, K))


def edge_mlp_relu(x, e, graph, first, lin0, lin2, sink_x=None, sink_e=None):
    """``lin2(relu(lin0([x[ia] | x[ib] | e[ic]])))`` for all edges (two ``nn.Linear`` modules)."""
    if edge_mlp_rereads_x(x, e, lin0, lin2):
        dt = x.dtype
        return _MLPReluGather.apply(x, e, lin0.weight, lin0.bias, lin2.weight, lin2.bias, graph, first,
                                    shadow(lin0.weight, dt), shadow(lin2.weight, dt), sink_x, sink_e)
    return mlp_relu(edge_gather(x, e, graph, first, sink_x, sink_e), lin0, lin2)


class _SeedGather(torch.autograd.Function):
    """Seed-edge rows: [lead | x[t_src] | x[t_dst]] (fuse input, fused.py:257) or
    [relu(x[t_src]) | relu(x[t_dst]) | tail] (ClassifierHead input, decoder.py:18-19)."""

    @staticmethod
    def forward(ctx, x, other, seeds, mode, sink_x=None):
        x = x.contiguous()
        F = x.shape[1]
        B = seeds.B
        ctx.sink_x = sink_x
        if mode == "fuse":      # other = x_tab [B,S,C]; lead = its CLS token (row stride S*C)
            C = other.shape[-1]
            other = other.contiguous()
            lead = other.reshape(B, -1)
            out = _gather3([(lead, None, C, 0), (x, seeds.src, F, 0), (x, seeds.dst, F, 0)], B, x.dtype, x.device)
            ctx.offs = (C, C + F)
        else:                   # head: other = target edge embedding [B, Fe]
            other = other.contiguous()
            out = _gather3([(x, seeds.src, F, 1), (x, seeds.dst, F, 1), (other, None, other.shape[1], 0)], B, x.dtype,
                           x.device)
            ctx.offs = (0, F)
            ctx.save_for_backward(x)
        ctx.seeds, ctx.mode, ctx.F, ctx.oshape = seeds, mode, F, other.shape
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        seeds, F = ctx.seeds, ctx.F
        relu_src = ctx.saved_tensors[0] if ctx.mode == "head" else None
        # with a sink that already holds a gradient only the <= 2B seed-endpoint rows are touched
        dx, acc_x, ret_x = _sink_target(ctx.sink_x, (seeds.N, F), g.dtype, g.device)
        hub = torch.empty(L.load().tg_segment_hub_ints(2 * seeds.B), dtype=torch.int32, device=g.device)
        L.call("tg_segment_sum2", L.ptr(g), g.shape[1], ctx.offs[0], L.ptr(seeds.rowptr), L.ptr(seeds.perm),
               ctx.offs[1], None, None, seeds.B, L.ptr(relu_src), L.ptr(dx), seeds.N, F, L.ptr(hub), acc_x, L.dt(g),
               L.stream())
        if ctx.mode == "fuse":                                        # [g[:, :C] | 0] as the [B, S, C] gradient of x_tab
            C = ctx.oshape[-1]
            dother = torch.empty(ctx.oshape, dtype=g.dtype, device=g.device)
            _row_head_scale(g, g.shape[1], C, dother, seeds.B, dother[0].numel(), C, 1.0, 0.0)
        else:                                                         # the tail block made contiguous
            Fe = g.shape[1] - 2 * F
            dother = torch.empty(g.shape[0], Fe, dtype=g.dtype, device=g.device)
            _row_head_scale(g, g.shape[1], Fe, dother, g.shape[0], Fe, Fe, 1.0, 1.0, offset=2 * F)
        return (dx if ret_x else None), dother, None, None, None


def seed_gather(x, other, seeds, mode, sink_x=None):
    return _SeedGather.apply(x, other, seeds, mode, sink_x)


# --------------------------------------------------------------------------- PNA aggregation + scalers


class _PNAAggregate(torch.autograd.Function):
    """messages h [E,F] -> [N,4F] = mean | max | min | std per destination."""

    @staticmethod
    def forward(ctx, h, graph, sorted_rows):
        h = h.contiguous()
        E, F = h.shape
        rowptr, perm = graph.by_dst
        if sorted_rows:          # h rows already in CSR order: no indirection (perm = NULL selects the streaming kernel)
            perm = None
        ctx.sorted_rows = sorted_rows
        agg = torch.empty(graph.N, 4 * F, dtype=h.dtype, device=h.device)
        hub = torch.zeros(L.load().tg_segment_hub_ints(max(E, 1)), dtype=torch.int32, device=h.device)
        _launch("tg_pna_aggregate_fwd", L.ptr(h), L.ptr(rowptr), L.ptr(perm), L.ptr(agg), graph.N, F, E, L.ptr(hub),
                L.dt(h), L.stream())
        L.call("tg_pna_aggregate_hubs", L.ptr(h), L.ptr(agg), None, L.ptr(rowptr), L.ptr(perm), None, F, L.ptr(hub),
               L.dt(h), L.stream())                     # destinations with > 512 rows (listed by the launch above)
        ctx.save_for_backward(h, agg)
        ctx.graph = graph
        return agg

    @staticmethod
    def backward(ctx, g):
        h, agg = ctx.saved_tensors
        rowptr, perm = ctx.graph.by_dst
        if ctx.sorted_rows:
            perm = None
        dh = torch.empty_like(h)
        hub = torch.zeros(L.load().tg_segment_hub_ints(max(h.shape[0], 1)), dtype=torch.int32, device=h.device)
        g = g.contiguous()
        _launch("tg_pna_aggregate_bwd", L.ptr(h), L.ptr(agg), L.ptr(g), L.ptr(rowptr), L.ptr(perm),
                L.ptr(dh), ctx.graph.N, h.shape[1], L.ptr(hub), L.dt(h), L.stream())
        L.call("tg_pna_aggregate_hubs", L.ptr(h), L.ptr(agg), L.ptr(g), L.ptr(rowptr), L.ptr(perm), L.ptr(dh),
               h.shape[1], L.ptr(hub), L.dt(h), L.stream())
        return dh, None, None


def pna_aggregate(h, graph, sorted_rows=False):
    """h [E,F] -> [N,4F].  ``sorted_rows``: h is laid out in destination-sorted order (``edge_gather(..., "dst_sorted")``)."""
    return _PNAAggregate.apply(h, graph, bool(sorted_rows))


class _ScaleCombine(torch.autograd.Function):
    """out = xw + G[:, :F] + amp*G[:, F:2F] + att*G[:, 2F:]  (degree scalers after the post projection)."""

    @staticmethod
    def forward(ctx, xw, G, graph, avg_log):
        xw, G = xw.contiguous(), G.contiguous()
        N, F = xw.shape
        out = torch.empty_like(xw)
        rowptr = graph.by_dst[0]
        L.call("tg_pna_scale_combine_fwd", L.ptr(xw), L.ptr(G), L.ptr(rowptr), L.ptr(avg_log), L.ptr(out), N, F,
               L.dt(xw), L.stream())
        ctx.graph, ctx.avg_log = graph, avg_log
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        N, F = g.shape
        dG = torch.empty(N, 3 * F, dtype=g.dtype, device=g.device)
        L.call("tg_pna_scale_combine_bwd", L.ptr(g), L.ptr(ctx.graph.by_dst[0]), L.ptr(ctx.avg_log), L.ptr(dG), N, F,
               L.dt(g), L.stream())
        return g, dG, None, None


def pna_scale_combine(xw, G, graph, avg_log):
    return _ScaleCombine.apply(xw, G, graph, avg_log)


class _FoldPNAWeights(torch.autograd.Function):
    """The two weight folds of PNAConv as ONE autograd node with a hand-written backward (the op-by-op version cost
    ~40 five-microsecond launches per convolution: slice/cat backward fills and copies, AccumulateGrad adds):

        w_msg = [P[:, :2F] | P[:, 2F:] We],  b_msg = pb + P[:, 2F:] be          (edge_encoder into pre_nn)
        w_eff = Lw Qw,  b_eff = Lw qb + lb;  w_x = w_eff[:, :F]                  (lin into post_nn)
        w_st[s] = [w_eff block (s, agg_order[k])  for k in mean, max, min, std]  ([3F,4F], the kernel's layout)

    Parameter gradients are added straight into existing ``.grad`` buffers (FlatParams views) with addmm_/addmv_."""

    @staticmethod
    def forward(ctx, P, pb, We, be, Qw, qb, Lw, lb, agg_order):
        F = P.shape[0]
        P3 = P[:, 2 * F:]
        w_msg = torch.cat([P[:, :2 * F], P3 @ We], dim=1)
        b_msg = torch.addmv(pb, P3, be)
        w_eff = Lw @ Qw                                                         # [F,13F]
        b_eff = torch.addmv(lb, Lw, qb)
        w4 = w_eff[:, F:].view(F, 3, 4, F)
        if list(agg_order) != [0, 1, 2, 3]:
            w4 = w4[:, :, list(agg_order), :]
        w_st = w4.permute(1, 0, 2, 3).reshape(3 * F, 4 * F)
        w_x = w_eff[:, :F].contiguous()
        ctx.save_for_backward(P, We, be, Qw, qb, Lw)
        ctx.agg_order = list(agg_order)
        ctx.params = (P, pb, We, be, Qw, qb, Lw, lb)
        return w_msg, b_msg, w_x, b_eff, w_st

    @staticmethod
    def backward(ctx, dw_msg, db_msg, dw_x, db_eff, dw_st):
        P, We, be, Qw, qb, Lw = ctx.saved_tensors
        F = P.shape[0]
        z = lambda t, *shape: P.new_zeros(*shape) if t is None else t
        dw_msg, db_msg = z(dw_msg, F, 2 * F + We.shape[1]), z(db_msg, F)
        dw_x, db_eff, dw_st = z(dw_x, F, F), z(db_eff, F), z(dw_st, 3 * F, 4 * F)
        P3 = P[:, 2 * F:]
        d3 = dw_msg[:, 2 * F:]
        dP = torch.cat([dw_msg[:, :2 * F], torch.addmm(torch.outer(db_msg, be), d3, We.t())], dim=1)
        dWe = P3.t() @ d3
        dbe = P3.t() @ db_msg
        d4 = dw_st.view(3, F, 4, F).permute(1, 0, 2, 3)                          # [F,3,4,F] in the kernel's order
        if ctx.agg_order != [0, 1, 2, 3]:
            inv = [ctx.agg_order.index(j) for j in range(4)]
            d4 = d4[:, :, inv, :]
        dw_eff = torch.cat([dw_x, d4.reshape(F, 12 * F)], dim=1)                # [F,13F]
        grads = [dP, db_msg, dWe, dbe, None, None, None, db_eff]
        out = []
        # Qw, qb, Lw: GEMM-shaped gradients; add into the flat gradient buffer in the same launch when there is one
        tq, tqb, tl = (_grad_target(p) if isinstance(p, torch.nn.Parameter) else None for p in ctx.params[4:7])
        if tq is not None:
            tq.addmm_(Lw.t(), dw_eff)
        else:
            grads[4] = Lw.t() @ dw_eff
        if tqb is not None:
            tqb.addmv_(Lw.t(), db_eff)
        else:
            grads[5] = Lw.t() @ db_eff
        if tl is not None:
            tl.addmm_(dw_eff, Qw.t())
            tl.addr_(db_eff, qb)
        else:
            grads[6] = torch.addmm(torch.outer(db_eff, qb), dw_eff, Qw.t())
        return (*grads, None)


class _FoldPNAWeightsHIP(torch.autograd.Function):
    """The same folds as one launch (tg_pna_fold_fwd) that also writes every bf16 operand layout the step's GEMMs read
    (attached to the fp32 outputs: ``_lp`` / ``_lp_t`` as FlatParams does for parameters, ``_packs`` on ``w_st`` for the
    scaled post projection), and one backward call (tg_pna_fold_bwd) that adds the eight parameter gradients into their
    ``.grad`` buffers.  Replaces ~45 library launches per convolution and step."""

    @staticmethod
    def forward(ctx, P, pb, We, be, Qw, qb, Lw, lb, agg_order, lp):
        F, Fe = P.shape[0], We.shape[1]
        dev = P.device
        f32 = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)
        w_msg, b_msg, w_x, b_eff, w_st = f32(F, 2 * F + Fe), f32(F), f32(F, F), f32(F), f32(3 * F, 4 * F)
        out = L.FoldOut(*(t.data_ptr() for t in (w_msg, b_msg, w_x, b_eff, w_st)))
        packs = None
        if lp:      # one allocation for the six bf16 layouts
            sizes = [F * (2 * F + Fe), F * (2 * F + Fe), F * F, F * F, 12 * F * F, 12 * F * F]
            flat = torch.empty(sum(sizes), dtype=torch.bfloat16, device=dev)
            views, off = [], 0
            for n in sizes:
                views.append(flat[off:off + n]); off += n
            packs = (views[0].view(F, 2 * F + Fe), views[1].view(2 * F + Fe, F), views[2].view(F, F), views[3].view(F, F),
                     views[4].view(F, 12 * F), views[5].view(4 * F, 3 * F))
            (out.w_msg_lp, out.w_msg_lp_t, out.w_x_lp, out.w_x_lp_t, out.w_cat, out.wt_cat) = (t.data_ptr() for t in packs)
        params = L.FoldParams(*(t.data_ptr() for t in (P, pb, We, be, Qw, qb, Lw, lb)))
        order = (C.c_int32 * 4)(*agg_order)
        L.call("tg_pna_fold_fwd", C.byref(params), C.byref(out), F, Fe, order, L.stream())
        ctx.save_for_backward(P, pb, We, be, Qw, qb, Lw, lb)
        ctx.agg_order = tuple(agg_order)
        _FoldPNAWeightsHIP.last_packs = packs       # picked up by fold_pna_weights right after apply()
        return w_msg, b_msg, w_x, b_eff, w_st

    @staticmethod
    def backward(ctx, dw_msg, db_msg, dw_x, db_eff, dw_st):
        saved = ctx.saved_tensors
        P, pb, We, be, Qw, qb, Lw, lb = saved
        F, Fe = P.shape[0], We.shape[1]
        cont = lambda t: None if t is None else t.contiguous().float()
        gin = [cont(t) for t in (dw_msg, db_msg, dw_x, db_eff, dw_st)]
        grads = L.FoldGrads(*(None if t is None else t.data_ptr() for t in gin))
        outs, ptrs, acc = [], [], 0
        for i, p in enumerate(saved):
            tgt = _grad_target(p) if isinstance(p, torch.nn.Parameter) else None
            if not ctx.needs_input_grad[i]:
                outs.append(None); ptrs.append(None)
            elif tgt is not None:                  # add into the flat gradient buffer; autograd gets no tensor
                outs.append(None); ptrs.append(tgt.data_ptr()); acc |= 1 << i
            else:
                t = torch.empty_like(p, dtype=torch.float32)
                outs.append(t); ptrs.append(t.data_ptr())
        ws = _workspace(L.load().tg_pna_fold_ws_floats(F), P.device)
        dpar = L.FoldDParams(*ptrs, ws.data_ptr(), acc)
        params = L.FoldParams(*(t.data_ptr() for t in saved))
        order = (C.c_int32 * 4)(*ctx.agg_order)
        L.call("tg_pna_fold_bwd", C.byref(params), C.byref(grads), C.byref(dpar), F, Fe, order, L.stream())
        return (*outs, None, None)


def fold_pna_weights(P, pb, We, be, Qw, qb, Lw, lb, agg_order, lp_dtype=None):
    """(w_msg, b_msg, w_x, b_eff, w_st) of a PNAConv.  fp32 parameters on the MI355X take the one-launch HIP fold
    (``lp_dtype == torch.bfloat16`` also asks for the bf16 operand layouts); anything else the torch composition."""
    ts = (P, pb, We, be, Qw, qb, Lw, lb)
    if (_FOLD_HIP and all(t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() for t in ts)
            and P.shape[0] % 32 == 0 and We.shape[1] % 32 == 0):
        lp = lp_dtype == torch.bfloat16
        w_msg, b_msg, w_x, b_eff, w_st = _FoldPNAWeightsHIP.apply(*ts, tuple(agg_order), lp)
        if lp:
            packs, _FoldPNAWeightsHIP.last_packs = _FoldPNAWeightsHIP.last_packs, None
            w_msg._lp, w_msg._lp_t, w_x._lp, w_x._lp_t = packs[0], packs[1], packs[2], packs[3]
            w_st._packs = (packs[4], packs[5])
        return w_msg, b_msg, w_x, b_eff, w_st
    return _FoldPNAWeights.apply(P, pb, We, be, Qw, qb, Lw, lb, tuple(agg_order))


_FOLD_HIP = os.environ.get("TABGNN_NO_FOLD_KERNEL") != "1"


def degree_scalers(graph, avg_log):
    """fp32 (amp, att) per node of ``graph`` (by-destination degrees), cached on the graph per scaler buffer.  Rows are
    padded (with zeros) to a whole number of 128-row GEMM tiles, as tg_gemm_nt_scaled_bf16 addresses them."""
    cache = graph.__dict__.setdefault("_scalers", {})
    key = (avg_log.data_ptr(), avg_log._version)
    sc = cache.get(key)
    if sc is None:
        sc = torch.zeros((graph.N + 127) // 128 * 128, 2, dtype=torch.float32, device=avg_log.device)
        L.call("tg_pna_degree_scalers", L.ptr(graph.by_dst[0]), L.ptr(avg_log), L.ptr(sc), graph.N, L.stream())
        cache[key] = sc
    return sc


_FUSED_POST = os.environ.get("TABGNN_NO_FUSED_POST") != "1"
_POST_FWD_KERNEL = os.environ.get("TABGNN_NO_POST_FWD_KERNEL") != "1"     # same-box A/B switch: the two-GEMM forward
_POST_DAGG_KERNEL = os.environ.get("TABGNN_NO_POST_DAGG_KERNEL") != "1"   # same-box A/B switch: d agg on the scaled NT GEMM


def post_scaled_ok(x, agg, agg_width=None):
    """bf16, F = 128 node width, 4F-wide aggregate: the shapes tg_gemm_nt_scaled_bf16 / tg_gemm_tn_scaled_bf16 take.
    ``agg`` may be None when only its width is known yet (``agg_width``)."""
    if agg is not None and (agg.dtype != torch.bfloat16 or not agg.is_cuda or agg.shape[0] != x.shape[0]):
        return False
    K = agg.shape[1] if agg is not None else agg_width
    # (F == 128 spelled out: nt_ok's old "N == 128 or K == 128" policy used to imply it; the kernels are built for F = 128)
    return (_FUSED_POST and x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 2 and x.shape[1] == 128
            and nt_ok(x, 128, 128) and K % 128 == 0)


class _PNAPostScaled(torch.autograd.Function):
    """out = x w_x^T + b + agg W_0^T + amp*(agg W_1^T) + att*(agg W_2^T) with w_st = [W_0; W_1; W_2] ([3F,4F], fp32,
    the folded post projection): the degree scalers live inside the GEMMs (tg_gemm_nt_scaled_bf16 /
    tg_gemm_tn_scaled_bf16), so neither G = agg w_st^T [N,3F] nor its gradient exists."""

    @staticmethod
    def forward(ctx, x, w_x, b_x, agg, w_st, graph, avg_log, sink_x=None):
        ctx.sink_x = sink_x
        x, agg = x.contiguous(), agg.contiguous()
        N, F = x.shape
        K = agg.shape[1]
        packs = getattr(w_st, "_packs", None)        # written by the fold kernel (tg_pna_fold_fwd) with the weights
        if packs is not None and getattr(w_x, "_lp", None) is not None:
            wx_lp, wx_t = w_x._lp, w_x._lp_t
            w_cat, wt_cat = packs
        else:
            wx_lp = w_x.detach().to(torch.bfloat16).contiguous()
            wx_t = wx_lp.t().contiguous()
            w_lp = w_st.detach().to(torch.bfloat16).view(3, F, K)
            # [F, 3K], 128-column block 3c+s = W_s[:, 128c:128c+128] (the kernel's virtual-chunk order)
            w_cat = w_lp.view(3, F, K // 128, 128).permute(1, 2, 0, 3).reshape(F, 3 * K).contiguous()
            wt_cat = w_lp.permute(2, 0, 1).reshape(K, 3 * F).contiguous()      # [K, 3F] = [W_0^T | W_1^T | W_2^T]
        scales = degree_scalers(graph, avg_log)
        bias = b_x.detach().float().contiguous()
        if _POST_FWD_KERNEL and F == 128:        # one MFMA-bound kernel: x term, three scaler sets, bias (post_scaled.hip)
            out = torch.empty(N, F, dtype=x.dtype, device=x.device)
            _launch("tg_pna_post_fwd_bf16", L.ptr(agg), L.ptr(x), L.ptr(w_cat), L.ptr(wx_lp.contiguous()), L.ptr(bias),
                    L.ptr(scales), L.ptr(out), N, K, agg.stride(0), x.stride(0), out.stride(0), L.stream(),
                    nbytes=2 * N * (K + 2 * F))
        else:
            out = gemm_nt(x, wx_lp, bias)
            L.call("tg_gemm_nt_scaled_bf16", L.ptr(agg), L.ptr(w_cat), L.ptr(scales), L.ptr(out), N, F, K, agg.stride(0),
                   out.stride(0), NT_ACCUM, L.stream())
        ctx.save_for_backward(x, wx_t, agg, wt_cat, scales)
        return out

    @staticmethod
    def backward(ctx, g):
        x, wx_t, agg, wt_cat, scales = ctx.saved_tensors
        g = g.contiguous()
        N, F = g.shape
        K = agg.shape[1]
        dx = None
        if ctx.needs_input_grad[0]:      # g Wx, written (first user) or added in the GEMM epilogue into x's shared buffer
            buf, acc, ret = _sink_target(ctx.sink_x, (N, wx_t.shape[0]), g.dtype, g.device)
            gemm_nt(g, wx_t, None, NT_ACCUM if acc else 0, out=buf)
            dx = buf if ret else None
        dwx, dbx = weight_grad(g, x, True)
        if dbx is None:
            dbx = g.sum(0, dtype=torch.float32)
        dagg = torch.empty_like(agg)
        if _POST_DAGG_KERNEL and F == 128 and wt_cat.is_contiguous():
            # the forward's three-accumulator kernel, one launch over the K / 128 column tiles of dagg (post_scaled.hip)
            _launch("tg_pna_post_dagg_bf16", L.ptr(g), L.ptr(wt_cat), L.ptr(scales), L.ptr(dagg), N, K, g.stride(0),
                    dagg.stride(0), L.stream(), nbytes=2 * N * (K + F))
        else:
            L.call("tg_gemm_nt_scaled_bf16", L.ptr(g), L.ptr(wt_cat), L.ptr(scales), L.ptr(dagg), N, K, F, g.stride(0),
                   dagg.stride(0), 0, L.stream())
        dw = torch.empty(3 * F, K, dtype=torch.float32, device=g.device)
        ws = _workspace(L.load().tg_gemm_tn_workspace_floats(N, 3 * F, K), g.device)
        L.call("tg_gemm_tn_scaled_bf16", L.ptr(g), L.ptr(agg), L.ptr(scales), L.ptr(dw), L.ptr(ws), N, F, K, g.stride(0),
               agg.stride(0), 0, L.stream())
        return dx, dwx, dbx, dagg, dw, None, None, None


def pna_post_scaled(x, w_x, b_x, agg, w_st, graph, avg_log, sink_x=None):
    return _PNAPostScaled.apply(x, w_x, b_x, agg, w_st, graph, avg_log, sink_x)


class _GINEAggregate(torch.autograd.Function):
    """out[n] = self_scale*x[n] + sum_{e: dst[e]=n} relu(x[src[e]] + le[e])  (GINEConv.propagate + the (1+eps)*x_r term,
    src/nn/gnn/gine.py:18-19 through torch_geometric GINEConv); the [E,F] message tensor never exists."""

    @staticmethod
    def forward(ctx, x, le, graph, self_scale):
        x, le = x.contiguous(), le.contiguous()
        N, F = x.shape
        out = torch.empty_like(x)
        rowptr, perm = graph.by_dst
        hub = torch.empty(L.load().tg_segment_hub_ints(max(graph.E, 1)), dtype=torch.int32, device=x.device)
        _launch("tg_gine_aggregate_fwd", L.ptr(x), L.ptr(le), L.ptr(graph.src), L.ptr(rowptr), L.ptr(perm),
                float(self_scale), L.ptr(out), N, F, L.ptr(hub), L.dt(x), L.stream())
        ctx.save_for_backward(x, le)
        ctx.graph, ctx.self_scale = graph, float(self_scale)
        return out

    @staticmethod
    def backward(ctx, g):
        x, le = ctx.saved_tensors
        graph = ctx.graph
        g = g.contiguous()
        N, F = x.shape
        dle = torch.empty_like(le)
        L.call("tg_gine_message_bwd", L.ptr(x), L.ptr(le), L.ptr(g), L.ptr(graph.src), L.ptr(graph.dst), L.ptr(dle),
               graph.E, F, L.dt(x), L.stream())
        dx = torch.empty_like(x)
        hub = torch.empty(L.load().tg_segment_hub_ints(max(graph.E, 1)), dtype=torch.int32, device=x.device)
        L.call("tg_segment_sum2", L.ptr(dle), F, 0, L.ptr(graph.by_src[0]), L.ptr(graph.by_src[1]), 0, None, None, 0,
               None, L.ptr(dx), N, F, L.ptr(hub), 0, L.dt(x), L.stream())
        if ctx.self_scale != 0.0:
            L.call("tg_axpby", L.ptr(dx), L.ptr(g), L.ptr(dx), dx.numel(), 1.0, ctx.self_scale, L.dt(dx), L.stream())
        return dx, dle, None, None


def gine_aggregate(x, le, graph, self_scale=1.0):
    """x [N,F], le [E,F] (edge embeddings through GINEConv.lin, edge order) -> [N,F]."""
    return _GINEAggregate.apply(x, le, graph, self_scale)


# --------------------------------------------------------------------------- fused-layer tail: CLS merge + pooling


def _row_head_scale(src, ld_src, w_src, dst, B, W, C, s_head, s_tail, offset=0):
    """dst[r, :W] = (c < C ? s_head : s_tail) * src[r, offset + c] for c < w_src, zero beyond (tg_row_head_scale)."""
    if B == 0:
        return
    L.call("tg_row_head_scale", src.data_ptr() + offset * src.element_size(), ld_src, w_src, L.ptr(dst), B, W, C, s_head,
           s_tail, L.dt(src), L.stream())


class _ClsMerge(torch.autograd.Function):
    """x_tab with its CLS token replaced by (cls + xf[:, :C]) / 2  (fused.py:259-260)."""

    @staticmethod
    def forward(ctx, xtab, xf):
        xtab, xf = xtab.contiguous(), xf.contiguous()
        B, S, C = xtab.shape
        out = torch.empty_like(xtab)
        L.call("tg_cls_merge_fwd", L.ptr(xtab), L.ptr(xf), L.ptr(out), B, S, C, xf.shape[1], L.dt(xtab), L.stream())
        ctx.cfg = (C, xf.shape[1])
        return out

    @staticmethod
    def backward(ctx, g):
        C, D = ctx.cfg
        g = g.contiguous()
        B, S, _ = g.shape
        dxtab = torch.empty_like(g)                                       # g with token 0 halved
        _row_head_scale(g, S * C, S * C, dxtab, B, S * C, C, 0.5, 1.0)
        dxf = torch.empty(B, D, dtype=g.dtype, device=g.device)           # [g[:, 0] / 2 | 0]
        _row_head_scale(g, S * C, C, dxf, B, D, C, 0.5, 0.0)
        return dxtab, dxf


def cls_merge(xtab, xf):
    return _ClsMerge.apply(xtab, xf)


class _SeedPool(torch.autograd.Function):
    """x_gnn with every seed endpoint averaged with the mean of its fused embeddings (fused.py:261-268)."""

    @staticmethod
    def forward(ctx, x, xf, seeds, C, inplace, sink_x=None, stash=False):
        xf = xf.contiguous()
        N, F = x.shape
        ctx.seeds, ctx.cfg = seeds, (N, F, C)
        ctx.sink_x = sink_x
        if inplace and x.is_contiguous():       # touch only the <= 2B seed rows of x (fused.py:268 is in place too)
            if stash:       # keep the rows about to change for a consumer that re-reads x in its backward (_MLPReluGather)
                idx = seeds.tei.long()
                x._pool_patch = (idx, x.index_select(0, idx), x._version, None)
            ctx.mark_dirty(x)
            L.call("tg_seed_pool_inplace", L.ptr(x), L.ptr(xf), L.ptr(seeds.tei), L.ptr(seeds.rowptr), L.ptr(seeds.perm),
                   N, F, seeds.B, C, L.dt(x), L.stream())
            return x
        x = x.contiguous()
        out = torch.empty_like(x)
        L.call("tg_seed_pool_fwd", L.ptr(x), L.ptr(xf), L.ptr(seeds.rowptr), L.ptr(seeds.perm), L.ptr(out), N, F,
               seeds.B, C, L.dt(x), L.stream())
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        seeds = ctx.seeds
        N, F, C = ctx.cfg
        dx, acc, ret = _sink_target(ctx.sink_x, g.shape, g.dtype, g.device)
        add_to = None
        if acc:                          # not the sink's first user (it is, in the fused layer: created last, run first)
            add_to, dx = dx, torch.empty_like(g)
        dxf = torch.empty(seeds.B, C + 2 * F, dtype=g.dtype, device=g.device)
        L.call("tg_seed_pool_bwd", L.ptr(g), L.ptr(seeds.tei), L.ptr(seeds.rowptr), L.ptr(dx), L.ptr(dxf), N, F,
               seeds.B, C, L.dt(g), L.stream())
        if add_to is not None:
            add_to.add_(dx)
        return (dx if ret else None), dxf, None, None, None, None, None


POOL_RESTORE = os.environ.get("TABGNN_NO_POOL_RESTORE") != "1"


def seed_pool(x, xf, seeds, C, inplace=False, sink_x=None, stash=False):
    """``inplace``: update x itself.  It must be an intermediate nobody needs unchanged in the backward: autograd's
    version check catches tensors saved with save_for_backward; ``edge_mlp_relu`` (which re-reads its node input by
    plain reference) checks the version by hand and raises unless the change was stashed, see ``stash``.
    ``stash``: (in place only) keep the pre-pool rows on the tensor as ``_pool_patch`` = (row ids, rows, version seen,
    version left) for _MLPReluGather's backward, which puts them back before it re-reads x."""
    out = _SeedPool.apply(x, xf, seeds, C, inplace, sink_x, bool(stash and inplace))
    patch = getattr(x, "_pool_patch", None)
    if stash and inplace and patch is not None:
        out._pool_patch = x._pool_patch = (patch[0], patch[1], patch[2], out._version)
    return out


# --------------------------------------------------------------------------- loss


class _WeightedCE(torch.autograd.Function):
    """torch.nn.CrossEntropyLoss(weight=w) (main.py:335): sum_i w[y_i] * nll_i / sum_i w[y_i]."""

    @staticmethod
    def forward(ctx, logits, y, w):
        logits, y = logits.contiguous(), y.contiguous()
        B, K = logits.shape
        lossden = torch.empty(2, dtype=torch.float32, device=logits.device)
        partials = _workspace(512, logits.device)
        L.call("tg_weighted_ce_fwd", L.ptr(logits), L.ptr(y), L.ptr(w), B, K, L.ptr(lossden), L.ptr(partials),
               L.dt(logits), L.stream())
        ctx.save_for_backward(logits, y, w, lossden)
        return lossden[0]

    @staticmethod
    def backward(ctx, g):
        logits, y, w, lossden = ctx.saved_tensors
        B, K = logits.shape
        dl = torch.empty_like(logits)
        g = g.reshape(1).float().contiguous()
        L.call("tg_weighted_ce_bwd", L.ptr(logits), L.ptr(y), L.ptr(w), L.ptr(lossden), L.ptr(g), B, K, L.ptr(dl),
               L.dt(logits), L.stream())
        return dl, None, None


def weighted_cross_entropy(logits, y, weight=None):
    if y.dtype != torch.int64:
        y = y.long()
    return _WeightedCE.apply(logits, y.view(-1), weight)
