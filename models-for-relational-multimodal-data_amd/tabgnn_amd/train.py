"""Train-step machinery around the model (reference ``main.py:33-76,335-336``), MI355X-first:

* ``FlatParams``  — all parameters (and gradients) are views into ONE contiguous fp32 buffer, so the optimiser is
  one kernel launch and the data-parallel gradient exchange is one (bucketed) RCCL all-reduce;
  a bf16 shadow buffer (same layout) feeds the GEMMs in bf16 mode and is refreshed by the Adam kernel.
* ``FusedAdam``   — ``torch.optim.Adam(lr)`` semantics (betas 0.9/0.999, eps 1e-8), one HIP kernel.
* ``DataParallel``— one process per GPU; rank-0 broadcast at start, gradient all-reduce (mean) per step over
  ``torch.distributed`` (backend "nccl" = RCCL over xGMI on MI355X; "gloo" in the CPU tests).
* ``train_step``  — zero_grad -> forward -> weighted CE on the first batch_size rows -> backward -> (all-reduce) -> Adam.
"""
from __future__ import annotations

import os
import weakref

import torch
import torch.distributed as dist

from . import _lib as L
from . import ops


class FlatParams:
    def __init__(self, module: torch.nn.Module, shadow_dtype=None):
        params = [p for p in module.parameters() if p.requires_grad]
        seen, uniq = set(), []
        for p in params:
            if id(p) not in seen:
                seen.add(id(p)); uniq.append(p)
        self.params = uniq
        dev = uniq[0].device
        sizes = [(p.numel() + 7) // 8 * 8 for p in uniq]           # keep every view 32-byte aligned
        self.offsets = [0]
        for s in sizes:
            self.offsets.append(self.offsets[-1] + s)
        n = self.offsets[-1]
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(n, dtype=torch.float32, device=dev)
        self.shadow = torch.zeros(n, dtype=shadow_dtype, device=dev) if shadow_dtype not in (None, torch.float32) else None
        # transposed bf16 shadows of the 2-D parameters (``p._lp_t`` [in, out]): the dX GEMMs' row-major weight
        self.shadow_t = torch.zeros_like(self.shadow) if self.shadow is not None and dev.type == "cuda" else None
        t_rows = []
        for p, off in zip(uniq, self.offsets):
            view = self.flat[off:off + p.numel()].view_as(p)
            view.copy_(p.data)
            p.data = view
            p.grad = self.grad[off:off + p.numel()].view_as(p)
            if self.shadow is not None:
                p._lp = self.shadow[off:off + p.numel()].view_as(p)
                if self.shadow_t is not None and p.dim() == 2:
                    p._lp_t = self.shadow_t[off:off + p.numel()].view(p.shape[1], p.shape[0])
                    t_rows.append((off, p.shape[0], p.shape[1]))
        self.t_table = torch.tensor(t_rows, dtype=torch.int64, device=dev).reshape(-1, 3) if t_rows else None
        ref = weakref.ref(self)
        for p in uniq:
            p._flat_ref = ref                 # ops.shadow()/ops.wt() refresh through it when a shadow went stale
        self.refresh_shadow()
        # a checkpoint restore (main.py:271-274) writes the fp32 views in place: refresh the bf16 shadows right after it
        if self.shadow is not None:
            def _after_load(mod, incompatible, ref=ref):
                f = ref()
                if f is not None:
                    f.refresh_shadow()
            self._load_hook = module.register_load_state_dict_post_hook(_after_load)

    @property
    def numel(self):
        return self.flat.numel()

    def refresh_shadow(self):
        """Recompute the bf16 shadows (and their transposes) from the fp32 masters and stamp every parameter with the
        version the shadow was taken at: an in-place write through torch afterwards (``load_state_dict`` on a child
        module, ``p.copy_()``, a torch optimiser) bumps ``p._version`` and ``ops.shadow`` / ``ops.wt`` then refresh
        before handing the shadow out.  The Adam kernel keeps masters and shadows in step itself (``stamp()`` only)."""
        if self.shadow is None:
            return
        if self.flat.is_cuda:
            L.call("tg_cast_f32_to_bf16", L.ptr(self.flat), L.ptr(self.shadow), self.flat.numel(), L.stream())
        else:
            self.shadow.copy_(self.flat)
        self.refresh_transposed()
        self.stamp()

    def stamp(self):
        for p in self.params:
            p._lp_ver = p._version

    def refresh_transposed(self):
        if getattr(self, "t_table", None) is not None:
            L.call("tg_transpose_batched_bf16", L.ptr(self.shadow), L.ptr(self.shadow_t), L.ptr(self.t_table),
                   self.t_table.shape[0], L.stream())

    def zero_grad(self):
        if self.grad.is_cuda:
            L.call("tg_zero", L.ptr(self.grad), 4 * self.grad.numel(), L.stream())     # a kernel, never a memset node
        else:
            self.grad.zero_()
        for p, off in zip(self.params, self.offsets):     # autograd may have replaced .grad; re-point the views
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * off:
                p.grad = self.grad[off:off + p.numel()].view_as(p)


class FusedAdam:
    def __init__(self, flat: FlatParams, lr, betas=(0.9, 0.999), eps=1e-8):
        self.flat, self.lr, self.betas, self.eps = flat, lr, betas, eps
        self.m = torch.zeros_like(flat.flat)
        self.v = torch.zeros_like(flat.flat)
        self.t = 0

    def step(self, grad_scale=1.0, zero_grad=False, state=None):
        """``state`` (``graph_step.StepState``): bias corrections of the step count kept on the device — the form a
        captured HIP graph replays; ``self.t`` then only mirrors it."""
        self.t += 1
        f = self.flat
        if state is not None:
            L.call("tg_adam_step_dev", L.ptr(f.flat), L.ptr(f.grad), L.ptr(self.m), L.ptr(self.v), L.ptr(f.shadow),
                   f.flat.numel(), self.betas[0], self.betas[1], self.eps, L.ptr(state.buf), grad_scale, int(zero_grad),
                   L.stream())
            f.refresh_transposed()
            return
        L.call("tg_adam_step", L.ptr(f.flat), L.ptr(f.grad), L.ptr(self.m), L.ptr(self.v), L.ptr(f.shadow),
               f.flat.numel(), self.lr, self.betas[0], self.betas[1], self.eps, self.t, grad_scale, int(zero_grad),
               L.stream())
        f.refresh_transposed()
        # (the kernel wrote masters and shadows together through raw pointers: versions did not move)


class DataParallel:
    """Gradient-averaging data parallelism over independently sampled mini-batches."""

    def __init__(self, module, flat: FlatParams, bucket_mb=32, sync_buffers=True, sync_batchnorm=False):
        """``sync_batchnorm``: BatchNorm statistics over the batches of ALL ranks (optional; default is the
        DDP convention of per-rank statistics): every ``layers.BatchNorm`` all-reduces its [2F+1] statistics vector
        in the forward and its [2F] gradient-statistics vector in the backward."""
        self.module, self.flat = module, flat
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.bucket = int(bucket_mb * (1 << 20) // 4)
        # TABGNN_FORCE_ALLREDUCE=1 runs the collective path even with one rank (smoke test of the RCCL plumbing)
        self.active = self.world > 1 or (dist.is_initialized() and os.environ.get("TABGNN_FORCE_ALLREDUCE") == "1")
        self.calls = self.bytes = 0        # collective bookkeeping (bench.py's `collective` object)
        if dist.is_initialized():       # per-rank dropout stream (fixed base + rank): ranks never share a mask,
            # and constructing a second DataParallel in the same process gives the same stream again
            ops.DropoutRNG.seed = (ops.DropoutRNG.BASE_SEED + 0x9E3779B97F4A7C15 * dist.get_rank()) & ((1 << 63) - 1)
        if self.active:
            dist.broadcast(flat.flat, src=0)
            if sync_buffers:
                for b in module.buffers():
                    dist.broadcast(b, src=0)
            flat.refresh_shadow()
        if sync_batchnorm and dist.is_initialized() and dist.get_world_size() > 1:
            from .layers import BatchNorm
            for m in module.modules():
                if isinstance(m, BatchNorm):
                    m.sync_group = dist.group.WORLD

    # ---- bucket-ready exchange ("overlapped with backward")
    def enable_overlap(self, stages):
        """``stages``: modules in FORWARD order (e.g. the backbone layers, then the head).  The gradient ranges of stage
        i+1 .. are exchanged as soon as the backward reaches the output of stage i, while the earlier stages' backward
        kernels still run; everything else (stage 0, parameters outside the stages) goes after the backward as before.

        Why "the backward reached stage i's output" means "the later stages' gradients are final": the autograd engine
        pops the ready node with the HIGHEST sequence number, and a node is only ready once every node that consumes
        its outputs has run — so when a node created in stage i is popped, every live node created after it (all of
        stages i+1 .., the head, the loss) has already run, including the ones that only write parameter gradients
        (weight folds, the in-place weight-gradient GEMMs of the composite nodes).  The all-reduce is issued from the
        node's pre-hook on the stream the node runs on, so the collective stream waits for exactly the kernels queued
        so far.  ``tests/test_ddp_gloo.py`` holds the result against the post-backward exchange bit for bit.  Every backward
        must be followed by ``all_reduce_grads()`` (``train_step`` does); ``GraphedTrainStep`` switches the overlap off (a
        replayed graph runs no autograd hooks)."""
        stages = list(stages)
        index = {id(p): i for i, p in enumerate(self.flat.params)}
        self._stage_ranges = []
        for m in stages:
            ids = sorted({index[id(p)] for p in m.parameters() if id(p) in index})
            ranges = []
            for i in ids:                               # contiguous runs of the flat buffer
                lo, hi = self.flat.offsets[i], self.flat.offsets[i + 1]
                if ranges and ranges[-1][1] == lo:
                    ranges[-1][1] = hi
                else:
                    ranges.append([lo, hi])
            self._stage_ranges.append([tuple(r) for r in ranges])
        seen = set()
        for rs in self._stage_ranges:                   # a parameter shared by two stages is final with the EARLIER one only
            for r in rs:
                if r in seen:
                    raise ValueError("stages share parameters: exchange them after the backward instead")
                seen.add(r)
        self._hooks = [m.register_forward_hook(self._make_forward_hook(i)) for i, m in enumerate(stages[:-1])]
        self._works, self._done_from = [], len(stages)
        self.overlapped_bytes = 0

    def disable_overlap(self):
        for h in getattr(self, "_hooks", []):
            h.remove()
        self._hooks, self._stage_ranges = [], None

    def _make_forward_hook(self, i):
        def hook(module, inputs, output):
            if not (self.active and torch.is_grad_enabled()) or (torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()):
                return
            outs = output if isinstance(output, (tuple, list)) else (output,)
            fired = []

            def ready(_grads):
                if not fired:                            # once per forward, whichever output's node is reached first
                    fired.append(True)
                    self._exchange_stages_from(i + 1)
            for t in outs:
                if isinstance(t, torch.Tensor) and t.grad_fn is not None:
                    t.grad_fn.register_prehook(ready)
        return hook

    def _issue(self, lo, hi):
        g = self.flat.grad
        for a in range(lo, hi, self.bucket):
            b = min(a + self.bucket, hi)
            self._works.append(dist.all_reduce(g[a:b], op=dist.ReduceOp.SUM, async_op=True))
            self.bytes += 4 * (b - a)

    def _exchange_stages_from(self, first):
        """Stages ``first`` .. (not yet exchanged in this step) are final: start their all-reduces."""
        before = self.bytes
        for s in range(first, self._done_from):
            for lo, hi in self._stage_ranges[s]:
                self._issue(lo, hi)
        self._done_from = min(self._done_from, first)
        self.overlapped_bytes += self.bytes - before

    def all_reduce_grads(self):
        """Sum over ranks, bucketed (async, in flight together); the 1/world factor is folded into the optimiser.  With
        ``enable_overlap`` the ranges already in flight are only waited for; the rest is exchanged here."""
        if not self.active:
            return 1.0
        g = self.flat.grad
        stage_ranges = getattr(self, "_stage_ranges", None)
        if stage_ranges is None:
            self._works = []
            self._issue(0, g.numel())
        else:
            done = sorted(r for s in range(self._done_from, len(stage_ranges)) for r in stage_ranges[s])
            pos = 0
            for lo, hi in done + [(g.numel(), g.numel())]:      # the complement of what is in flight
                if lo > pos:
                    self._issue(pos, lo)
                pos = max(pos, hi)
            self._done_from = len(stage_ranges)                 # (armed for the next step)
        for w in self._works:
            w.wait()
        self.calls += len(self._works)
        self._works = []
        return 1.0 / self.world

    def sync_buffers(self):
        """BatchNorm running statistics evolve on rank-local batches (torch DDP re-broadcasts buffers every forward;
        here they stay local during training, which costs nothing per step): call this before evaluating or saving a
        checkpoint so that every rank holds the MEAN of the ranks' running statistics (``num_batches_tracked`` and
        other integer buffers are taken from rank 0)."""
        if not self.active:
            return
        for b in self.module.buffers():
            if b.is_floating_point():
                dist.all_reduce(b, op=dist.ReduceOp.SUM)
                b.div_(self.world)
            else:
                dist.broadcast(b, src=0)


class IndexGuard:
    """Out-of-range node ids in ``edge_index`` / ``target_edge_index`` are clamped by the id-conversion kernel so that
    no kernel faults, and flagged in a device word (``SubgraphIndex.err`` / ``SeedIndex.err``).  The reference raises an
    IndexError at ``x_gnn[src]``; here the flags of a step are summed and copied to pinned host memory WITHOUT a
    synchronisation, and the next ``train_step`` (or ``check(wait=True)``) raises once the copy has landed."""
    _pending = []          # (event, pinned host int32) of earlier steps

    @classmethod
    def collect(cls):
        flags = ops.take_index_errors()
        if not flags:
            return
        total = flags[0] if len(flags) == 1 else torch.stack([f.reshape(()) for f in flags]).sum().reshape(1).int()
        host = torch.empty(1, dtype=torch.int32, pin_memory=True)
        host.copy_(total.reshape(1), non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        cls._pending.append((ev, host))

    @classmethod
    def check(cls, wait=False):
        keep = []
        for ev, host in cls._pending:
            if wait:
                ev.synchronize()
            if ev.query():
                if int(host[0]) != 0:
                    cls._pending = []
                    raise RuntimeError("edge_index / target_edge_index held node ids outside [0, num_nodes) "
                                       "(detected after the step that used them)")
            else:
                keep.append((ev, host))
        cls._pending = keep


def train_step(model, flat, opt, batch, loss_weight, ddp=None, step_seed=None):
    """One supervised step (main.py:41-75).  batch = (node_tf, edge_index, edge_tf, y)."""
    node_tf, edge_index, edge_tf, y = batch
    IndexGuard.check()
    ops.take_index_errors()               # flags of index structures built outside a step are not this step's business
    ops.DropoutRNG.new_step(step_seed)
    flat.zero_grad()
    logits = model(node_tf, edge_index, edge_tf)
    bs = y.shape[0]
    loss = ops.weighted_cross_entropy(logits[:bs], y.view(-1), loss_weight)
    loss.backward()
    scale = ddp.all_reduce_grads() if ddp is not None else 1.0
    opt.step(grad_scale=scale)
    IndexGuard.collect()
    return loss.detach(), logits.detach()
