"""HIP-graph replay of the supervised train step over shape buckets (the launch-bound regime).

At the reference's default ``--batch_size 200`` (``utils.py:40-44``) one step of ``main.py:41-75`` is ~330 kernel launches
here for well under 2 ms of GPU work: the step is bound by the host issuing launches.  A HIP graph removes that cost,
but a graph bakes in every pointer, size and host scalar of its nodes, and a sampled batch never repeats its (E, N).
What this module does about each:

* **shapes** — a batch is padded to a *bucket* ``(E_pad, N_pad)`` (``bucket_size``: eighth-of-an-octave steps, at most
  12.5 % extra rows).  Padding edges are self loops on padding nodes, so no real node ever aggregates a padding
  message; padding rows carry the raw values of table row 0 (any valid row does).  Every operator of the path is local
  to a row, an edge or a seed — their padding rows get a zero upstream gradient and add nothing to any weight
  gradient — except BatchNorm, whose statistics couple the rows: the BatchNorm kernels take the number of real rows
  from device memory (``tg_set_bn_row_limit``) and give the padding rows a zero input gradient.
* **pointers** — every input of the step lives in static buffers owned by the bucket (``_Bucket.static``): raw rows
  or row ids of the two frames, labels, ``edge_index``, the host-built index structures (one int32 array whose part
  offsets depend on the bucket alone) and the real node count.  A step copies the batch into them and replays.
  Intermediates come from the graph's private pool (one pool for all buckets: replays are sequential).
* **host scalars** — the dropout seed, Adam's step count and bias corrections live in a device record
  (``StepState`` -> ``tg_advance_step``), advanced by the first node of the graph.
* **library nodes** — the body records kernels only: gradient zeroing happens in the Adam kernel, ``tg_zero`` replaces
  memsets, and the input copies stay outside the graph.

``GraphedTrainStep.run_eager`` executes the very same body without capture (the parity twin of the replay test).
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib as L
from . import ops
from .frame import TensorFrame
from .sampler import fill_host_batch_index, host_batch_index, host_index_offsets, index_over


class StepState:
    """The device-resident scalars of a step: ``buf`` = int64[4] (seed word, step count t, two packed floats
    lr/(1-b1^t) and 1/sqrt(1-b2^t), reserved) — layout of ``include/tabgnn_hip.h:tg_advance_step``."""

    def __init__(self, device, seed=0x1234ABCD, t=0):
        self.buf = torch.tensor([int(seed) & ((1 << 63) - 1), int(t), 0, 0], dtype=torch.int64, device=device)

    def advance(self, lr, betas):
        L.call("tg_advance_step", L.ptr(self.buf), float(lr), float(betas[0]), float(betas[1]), L.stream())

    def seed_arg(self):
        """The ``seed`` argument that makes a kernel read THIS record's seed word when it runs
        (``include/tabgnn_hip.h:TG_SEED_DEVICE``): an explicit per-launch argument, no library-side state."""
        return (1 << 63) | self.buf.data_ptr()


def bucket_size(n, floor=64):
    """Smallest multiple of 2^(floor(log2 n) - 3) that is >= n: eight steps per octave, <= 12.5 % padding."""
    n = max(int(n), floor)
    q = 1 << max(n.bit_length() - 4, 0)
    return -(-n // q) * q


def pad_edges(edge_index, n_real, e_pad, n_pad):
    """``edge_index`` (int64 [2,E], host) with ``e_pad - E`` self loops appended, spread over the padding nodes
    ``n_real .. n_pad-1``."""
    ei = np.ascontiguousarray(edge_index.numpy() if isinstance(edge_index, torch.Tensor) else np.asarray(edge_index),
                              dtype=np.int64)
    extra = e_pad - ei.shape[1]
    if extra < 0 or n_pad < n_real:
        raise ValueError("bucket smaller than the batch")
    if extra == 0:
        return ei
    if n_pad == n_real:
        raise ValueError("padding edges need at least one padding node")
    loops = n_real + np.arange(extra, dtype=np.int64) % (n_pad - n_real)
    return np.concatenate([ei, np.stack([loops, loops])], axis=1)


def _pad_rows(t, n):
    """[R, ...] -> [n, ...]: the padding rows repeat row 0."""
    if t.shape[0] == n:
        return t
    return torch.cat([t, t[:1].expand(n - t.shape[0], *t.shape[1:])], dim=0)


def _pack(tensors):
    """All parts in ONE byte arena (256-byte aligned parts; on the GPU when any part already is): a step then moves a
    batch into its bucket with a single copy.  Returns (arena, views by name, layout)."""
    dev = next((v.device for v in tensors.values() if v.is_cuda), torch.device("cpu"))
    layout, off = [], 0
    for k, v in tensors.items():
        layout.append((k, v.dtype, tuple(v.shape), off))
        off += (v.numel() * v.element_size() + 255) // 256 * 256
    # host arenas are pinned when a GPU is there: the step's one upload is then a true asynchronous copy
    pin = dev.type == "cpu" and torch.cuda.is_available()
    arena = torch.empty(off, dtype=torch.uint8, device=dev, pin_memory=pin)
    views = _views(arena, layout)
    for k, v in tensors.items():
        views[k].copy_(v, non_blocking=True)
    return arena, views, layout


def _views(arena, layout):
    out = {}
    for k, dtype, shape, off in layout:
        n = int(np.prod(shape)) * torch.empty(0, dtype=dtype).element_size()
        out[k] = arena[off:off + n].view(dtype).view(shape)
    return out


class Prepared:
    """One batch in bucket form: ``key`` = (E_pad, N_pad), ``arena`` = every part in one byte buffer (host or device),
    ``tensors`` = the parts as views of it, ``off`` = part offsets of the index array."""
    __slots__ = ("key", "arena", "tensors", "layout", "off", "e_real", "n_real", "lazy")

    def __init__(self, key, arena, layout, off, e_real, n_real, lazy):
        self.key, self.arena, self.layout, self.off = key, arena, layout, off
        self.tensors = _views(arena, layout)
        self.e_real, self.n_real, self.lazy = e_real, n_real, lazy

    def to(self, device):
        """The same batch with its arena on ``device`` (uploaded ahead of the step, as a prefetching loader does)."""
        return Prepared(self.key, self.arena.to(device, non_blocking=True), self.layout, self.off, self.e_real,
                        self.n_real, self.lazy)


def prepare(batch, n_seed, key=None):
    """Pad ``batch`` = (node_tf, edge_index, edge_tf, y) to its bucket (or to ``key``) and build the index structures of
    the padded graph on the host.  ``edge_index`` may be a host or device int64 tensor; lazy frames (row ids over an
    HBM-resident table) stay lazy.  Pure host/torch work outside any capture: a sampler thread can run it."""
    node_tf, edge_index, edge_tf, y = batch
    if isinstance(edge_index, ops.BatchIndex):
        edge_index = edge_index.edge_index
    E, N = int(edge_index.shape[1]), int(node_tf.num_rows)
    e_pad, n_pad = key if key is not None else (bucket_size(E), bucket_size(N + 1))
    ei = pad_edges(edge_index.cpu() if isinstance(edge_index, torch.Tensor) else edge_index, N, e_pad, n_pad)
    flat, off, ei = host_batch_index(ei, n_pad, n_seed)
    lazy = node_tf.row_ids is not None
    if lazy != (edge_tf.row_ids is not None):
        raise ValueError("node and edge frame must both be lazy or both be materialised")
    t = {"flat": torch.from_numpy(flat), "ei": torch.from_numpy(ei),
         "n_real": torch.tensor([N], dtype=torch.int32), "y": y.reshape(-1)}
    if lazy:
        t["node.ids"], t["edge.ids"] = _pad_rows(node_tf.row_ids, n_pad), _pad_rows(edge_tf.row_ids, e_pad)
    else:
        for name, tf, n in (("node", node_tf, n_pad), ("edge", edge_tf, e_pad)):
            for st, v in tf.feat_dict.items():
                t[f"{name}.{st.value}"] = _pad_rows(v, n)
    arena, _, layout = _pack(t)
    return Prepared((e_pad, n_pad), arena, layout, off, E, N, lazy)


_SAMPLE_LAYOUTS = {}         # (e_pad, n_pad, n_seed) -> (layout, arena bytes, index part offsets): functions of the bucket alone


def _sample_layout(e_pad, n_pad, n_seed):
    key = (e_pad, n_pad, n_seed)
    hit = _SAMPLE_LAYOUTS.get(key)
    if hit is None:
        off = host_index_offsets(e_pad, n_pad, n_seed)
        parts = (("flat", torch.int32, (int(off[13]),)), ("ei", torch.int64, (2, e_pad)), ("n_real", torch.int32, (1,)),
                 ("y", torch.int64, (n_seed,)), ("node.ids", torch.int64, (n_pad,)), ("edge.ids", torch.int64, (e_pad,)))
        layout, pos = [], 0
        for name, dt, shape in parts:
            layout.append((name, dt, shape, pos))
            pos += (int(np.prod(shape)) * torch.empty(0, dtype=dt).element_size() + 255) // 256 * 256
        hit = _SAMPLE_LAYOUTS[key] = (layout, pos, off)
    return hit


def prepare_sample(eid, edge_index, nodes, y, n_seed, key=None):
    """The sampler's output (``NeighborSampler.sample``: edge ids, local ``edge_index``, node ids — host tensors, seed
    edges first) and the seed labels as a bucket-form batch of LAZY frames: the frames' rows are ids into the
    HBM-resident tables (``ColumnStore``), so a batch is ids + index parts, one pinned arena, one upload.  Every part is
    written STRAIGHT into the arena (numpy views of the pinned bytes; the index structures by ``tg_host_batch_index`` into
    its slot): no intermediate tensors, no torch dispatch — on a loaded host a ``torch.cat`` of 30 k ids cost milliseconds
    (round 3: 5.7-8 ms per batch and thread, more than the 3 ms GPU step; now the index pass itself, < 1.5 ms).  Pure host
    work that releases the GIL in its heavy part: run it in the sampler thread."""
    def as_np(t, dt):
        if isinstance(t, torch.Tensor):
            if t.is_cuda:        # (a device-resident sample: tabgnn_amd.device_sampler.prepare_sample_device builds the arena on the GPU)
                raise ValueError("prepare_sample takes host tensors; use prepare_sample_device for a DeviceNeighborSampler batch")
            t = t.detach().numpy()
        return np.ascontiguousarray(np.asarray(t), dtype=dt)
    eid, nodes, y = as_np(eid, np.int64).reshape(-1), as_np(nodes, np.int64).reshape(-1), as_np(y, np.int64).reshape(-1)
    ei_in = as_np(edge_index, np.int64)
    E, N = int(eid.shape[0]), int(nodes.shape[0])
    e_pad, n_pad = key if key is not None else (bucket_size(E), bucket_size(N + 1))
    if e_pad < E or n_pad < N or (e_pad > E and n_pad == N):
        raise ValueError("bucket smaller than the batch (padding edges need at least one padding node)")
    if y.shape[0] != n_seed:
        raise ValueError(f"{y.shape[0]} labels for n_seed={n_seed}")
    layout, nbytes, off = _sample_layout(e_pad, n_pad, int(n_seed))
    arena = torch.empty(nbytes, dtype=torch.uint8, pin_memory=torch.cuda.is_available())
    a = arena.numpy()
    view = {name: a[pos:pos + int(np.prod(shape)) * np.dtype(_NP[dt]).itemsize].view(_NP[dt]).reshape(shape)
            for name, dt, shape, pos in layout}
    ei = view["ei"]
    ei[:, :E] = ei_in
    if e_pad > E:                       # padding edges: self loops spread over the padding nodes (pad_edges)
        ei[:, E:] = N + np.arange(e_pad - E, dtype=np.int64) % (n_pad - N)
    fill_host_batch_index(ei, n_pad, int(n_seed), view["flat"])
    view["n_real"][0] = N
    view["y"][:] = y
    view["node.ids"][:N] = nodes
    view["node.ids"][N:] = nodes[0]     # padding rows carry the raw values of the batch's first row (any valid row does)
    view["edge.ids"][:E] = eid
    view["edge.ids"][E:] = eid[0]
    return Prepared((e_pad, n_pad), arena, layout, off, E, N, True)


_NP = {torch.int32: np.int32, torch.int64: np.int64, torch.float32: np.float32}


class _Bucket:
    def __init__(self, prep, frames, device):
        self.key, self.off, self.layout = prep.key, prep.off, prep.layout
        self.arena = torch.empty(prep.arena.shape, dtype=torch.uint8, device=device)
        self.static = _views(self.arena, prep.layout)
        node_tf, edge_tf = frames

        def frame(name, tf):
            if prep.lazy:
                return TensorFrame(tf.feat_dict, tf.col_names_dict, None, self.static[f"{name}.ids"])
            return TensorFrame({st: self.static[f"{name}.{st.value}"] for st in tf.feat_dict}, tf.col_names_dict)
        self.node_tf, self.edge_tf = frame("node", node_tf), frame("edge", edge_tf)
        self.graph = None
        self.loss = self.logits = None
        self.err = []            # out-of-range flags of index structures built INSIDE the body (index=False), static memory

    def load(self, prep):
        if prep.layout != self.layout:
            raise RuntimeError("batch parts do not fit this bucket (different shapes or columns)")
        self.arena.copy_(prep.arena, non_blocking=True)          # the step's only copy

    def index(self, n_seed):
        """Fresh index objects over the static buffers (no cached per-graph tensors from an earlier batch)."""
        return index_over(self.static["flat"], self.off, self.key[1], n_seed, self.static["ei"])


class GraphedTrainStep:
    """``train.train_step`` (reference ``main.py:41-75``) as one HIP graph per shape bucket.

    ``step(prepare(batch, n_seed))`` -> (loss, logits of the padded batch; rows [:n_seed] are the seed edges).  With a
    ``DataParallel`` the graph ends after the backward and the all-reduce + Adam run eagerly behind it."""

    def __init__(self, model, flat, opt, loss_weight, n_seed, ddp=None, state=None, warmup=2, index=True):
        self.model, self.flat, self.opt, self.loss_weight = model, flat, opt, loss_weight
        self.n_seed, self.ddp, self.warmup = int(n_seed), ddp, int(warmup)
        if ddp is not None and getattr(ddp, "_stage_ranges", None):
            # a replayed graph runs no autograd, so no pre-hook can announce a finished stage — and hooks fired by the
            # warm-up / capture passes would leave ranges marked "in flight" for a step that never exchanged them
            ddp.disable_overlap()
        # index=False: hand the wrapper the plain int64 edge_index (GNN / TABGNNS of utils.py take nothing else): the batch's
        # CSRs are then built by the index kernels INSIDE the graph, from the bucket's static edge_index
        self.index = bool(index)
        self.device = flat.flat.device
        self.host_seed = ops.DropoutRNG.seed         # (per rank under DataParallel)
        # the record's seed word starts from the host seed, so ranks (and differently seeded runs) draw different masks
        self.state = state if state is not None else StepState(self.device, seed=0x1234ABCD ^ self.host_seed, t=opt.t)
        self.buckets = {}
        self.pool = None
        self.captured = None                 # (lr, betas) frozen into the graphs' tg_advance_step nodes
        self.t_next = None                   # opt.t the device record expects at the next step (None: not stepped yet)
        flat.zero_grad()                     # afterwards every Adam launch leaves the gradients zeroed

    # ---- the step itself: every launch below is a kernel on the current stream
    def _body(self, b):
        # every dropout launch of the body takes seed = TG_SEED_DEVICE(this record): the stream ids are the same in every
        # step, the device seed word (advanced by the first node) makes the masks differ
        ops.DropoutRNG.new_step(self.state.seed_arg())
        self.state.advance(self.opt.lr, self.opt.betas)
        ops.StepContext.set_bn_row_limit(b.static["n_real"])    # argument of BatchNorm's forward AND backward launches
        pending = ops.take_index_errors()                       # flags of earlier eager work: handed back below, not dropped
        try:
            logits = self.model(b.node_tf, b.index(self.n_seed) if self.index else b.static["ei"], b.edge_tf)
            loss = ops.weighted_cross_entropy(logits[:self.n_seed], b.static["y"], self.loss_weight)
            loss.backward()
        finally:
            ops.StepContext.set_bn_row_limit(None)
            ops.DropoutRNG.new_step(self.host_seed)             # eager code after the step draws from host seeds again
            b.err = ops.take_index_errors()                     # (index=False: the in-body conversions' flags, static memory)
            ops.restore_index_errors(pending)
        if self.ddp is None:
            self.opt.step(zero_grad=True, state=self.state)
        return loss.detach(), logits.detach()

    def _tail(self):
        if self.ddp is not None:
            self.opt.step(grad_scale=self.ddp.all_reduce_grads(), zero_grad=True, state=self.state)

    def _snapshot(self):
        bufs = [b for b in self.model.buffers()]
        return (self.flat.flat.clone(), self.opt.m.clone(), self.opt.v.clone(), self.state.buf.clone(),
                [b.clone() for b in bufs], bufs, self.opt.t)

    def _restore(self, snap):
        w, m, v, st, saved, bufs, t = snap
        self.flat.flat.copy_(w); self.opt.m.copy_(m); self.opt.v.copy_(v); self.state.buf.copy_(st)
        for b, s in zip(bufs, saved):
            b.copy_(s)
        self.opt.t = t
        self.flat.zero_grad()
        self.flat.refresh_shadow()

    def _bucket(self, prep, frames):
        b = self.buckets.get(prep.key)
        if b is not None:
            return b
        b = _Bucket(prep, frames, self.device)
        b.load(prep)
        snap = self._snapshot()              # warm-up steps and the capture must not train the model
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(self.warmup):
                self._body(b)
                if self.ddp is not None:
                    self.opt.step(zero_grad=True, state=self.state)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        if self.pool is None:
            self.pool = torch.cuda.graph_pool_handle()
        # thread_local: sampler threads keep running ``prepare_sample`` while a new bucket is captured, and their pinned
        # allocations (hipHostMalloc / event queries of the caching host allocator) are refused under the default global
        # capture mode — a worker would die and the loop would wait on its queue for ever
        with torch.cuda.graph(g, pool=self.pool, capture_error_mode="thread_local"):
            b.loss, b.logits = self._body(b)
        b.graph = g
        torch.cuda.synchronize()
        self._restore(snap)
        self.captured = (float(self.opt.lr), tuple(float(x) for x in self.opt.betas))
        self.buckets[prep.key] = b
        return b

    def _check(self, prep):
        """What a graph froze or assumes and the caller may have changed since (raise instead of silently ignoring it)."""
        n_y = int(prep.tensors["y"].numel())
        if n_y != self.n_seed:
            raise ValueError(f"batch has {n_y} seed rows, this GraphedTrainStep was built for n_seed={self.n_seed} "
                             "(a short last batch of an epoch needs its own instance or train_step)")
        if self.captured is not None:
            now = (float(self.opt.lr), tuple(float(x) for x in self.opt.betas))
            if now != self.captured:
                raise RuntimeError(f"opt.lr / opt.betas changed from {self.captured} to {now} after capture: the graphs hold "
                                   "the old values (tg_advance_step node); build a new GraphedTrainStep")
        if self.t_next is not None and self.opt.t != self.t_next:
            raise RuntimeError(f"opt.t = {self.opt.t}, but the device step record is at {self.t_next}: an eager opt.step() ran "
                               "between replays (Adam's bias corrections would go out of step)")

    def __call__(self, prep, frames=None):
        """``frames`` = (node_tf, edge_tf) templates, needed the first time a bucket is seen (column names, and for lazy
        frames the HBM-resident tables)."""
        from .train import IndexGuard
        IndexGuard.check()                   # raises for bad ids of an EARLIER step once their flag copy has landed
        self._check(prep)
        b = self.buckets.get(prep.key)
        if b is None:
            if frames is None:
                raise RuntimeError("first batch of a bucket: pass frames=(node_tf, edge_tf)")
            b = self._bucket(prep, frames)
        b.load(prep)
        b.graph.replay()
        if self.ddp is None:
            self.opt.t += 1                  # the Adam node ran inside the graph: mirror its device-side step count
        self._tail()
        self.t_next = self.opt.t
        if b.err:                            # index=False: the replay rewrote the flags of its in-graph index build
            ops.restore_index_errors(b.err)
            IndexGuard.collect()             # non-blocking copy; checked at the next step
        return b.loss, b.logits

    def run_eager(self, prep, frames):
        """The same body without a graph (same static buffers, same device state): the replay test's twin."""
        b = self.buckets.get(prep.key)
        if b is None:
            b = _Bucket(prep, frames, self.device)
            self.buckets[prep.key] = b
        self._check(prep)
        b.load(prep)
        loss, logits = self._body(b)
        self._tail()
        self.t_next = self.opt.t
        return loss, logits
