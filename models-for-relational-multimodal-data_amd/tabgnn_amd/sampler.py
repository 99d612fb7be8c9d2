"""Host-side input pipeline pieces either side of the hot path:

* ``NeighborSampler`` — native (C++/OpenMP, ``libtabgnn_sampler.so``) k-hop edge-seeded sampler + relabel replacing
  ``sample_neighbors`` / the dict relabel of ``get_graph_inputs``
  (``src/datasets/ibm_transactions_for_aml.py:61-112,159-180``; PyG ``NeighborSampler`` built at
  ``src/datasets/util/graph.py:38,46,53``).
* ``ColumnStore`` — columnar raw edge/node tables (optionally resident in HBM) with batch row gather: the
  ``TensorFrame.__getitem__`` calls of ``get_graph_inputs`` (``ibm…py:163,168``), so a mini-batch is assembled as
  ``(node_tf, edge_index, edge_tf, y)`` exactly as ``main.py:48`` receives it.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch

from .frame import TensorFrame

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libtabgnn_sampler.so")
_lib = None


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise RuntimeError(f"{_LIB_PATH} not found: build it with `make -C models-for-relational-multimodal-data_amd`")
        lib = C.CDLL(_LIB_PATH)
        i64p, i32p = C.POINTER(C.c_int64), C.POINTER(C.c_int32)
        lib.tg_sampler_last_error.restype = C.c_char_p
        lib.tg_sampler_create.argtypes = [i64p, i64p, C.c_int64, C.c_int64]
        lib.tg_sampler_create.restype = C.c_void_p
        lib.tg_sampler_destroy.argtypes = [C.c_void_p]
        lib.tg_sampler_num_edges.argtypes = [C.c_void_p]
        lib.tg_sampler_num_edges.restype = C.c_int64
        lib.tg_sampler_max_edges.argtypes = [C.c_int64, i32p, C.c_int32]
        lib.tg_sampler_max_edges.restype = C.c_int64
        lib.tg_sampler_sample.argtypes = [C.c_void_p, i64p, i64p, i64p, C.c_int64, i32p, C.c_int32, C.c_uint64,
                                          C.c_int32, C.c_int64, i64p, i64p, i64p, i64p, i64p]
        lib.tg_sampler_sample.restype = C.c_int
        lib.tg_sampler_draw.argtypes = [C.c_void_p, i64p, i64p, i64p, C.c_int64, i32p, C.c_int32, C.c_uint64,
                                        C.c_int32, C.c_int64, i64p, i64p]
        lib.tg_sampler_draw.restype = C.c_int
        lib.tg_sampler_emit.argtypes = [C.c_void_p, C.c_int32, C.c_int64, i64p, i64p, i64p]
        lib.tg_sampler_emit.restype = C.c_int
        lib.tg_host_csr.argtypes = [i64p, C.c_int64, C.c_int64, i32p, i32p]
        lib.tg_host_csr.restype = C.c_int
        lib.tg_host_batch_index.argtypes = [i64p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, i32p, i64p]
        lib.tg_host_batch_index.restype = C.c_int
        _lib = lib
    return _lib


def _p64(a):
    return a.ctypes.data_as(C.POINTER(C.c_int64))


class NeighborSampler:
    """k-hop sampler over the in-edges of a directed graph; edge id = position in ``edge_index``."""

    def __init__(self, edge_index, num_nodes, num_neighbors=(100, 100), num_threads=0):
        lib = _load()
        ei = np.ascontiguousarray(np.asarray(edge_index, dtype=np.int64))
        if ei.ndim != 2 or ei.shape[0] != 2:
            raise ValueError("edge_index must be [2, E]")
        self._src, self._dst = np.ascontiguousarray(ei[0]), np.ascontiguousarray(ei[1])
        self.num_nodes, self.num_edges = int(num_nodes), ei.shape[1]
        self.fanout = np.asarray(list(num_neighbors), dtype=np.int32)
        self.num_threads = int(num_threads)
        self._h = lib.tg_sampler_create(_p64(self._src), _p64(self._dst), self.num_edges, self.num_nodes)
        if not self._h:
            raise ValueError(lib.tg_sampler_last_error().decode())

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.tg_sampler_destroy(self._h)
            self._h = None

    def sample(self, seed_eids, rng_seed=0):
        """-> (eid int64 [E_out] (seed edges first, in order), edge_index int64 [2, E_out] LOCAL ids,
        nodes int64 [N_out] sorted global ids)."""
        lib = _load()
        seeds = np.ascontiguousarray(np.asarray(seed_eids, dtype=np.int64))
        B = seeds.shape[0]
        if B == 0:
            raise ValueError("need at least one seed edge")
        if seeds.min() < 0 or seeds.max() >= self.num_edges:
            raise ValueError("seed edge id out of range")
        s_src, s_dst = np.ascontiguousarray(self._src[seeds]), np.ascontiguousarray(self._dst[seeds])
        fan = self.fanout
        bound = lib.tg_sampler_max_edges(B, fan.ctypes.data_as(C.POINTER(C.c_int32)), len(fan)) if (fan >= 0).all() \
            else self.num_edges + B
        cap = int(min(bound, self.num_edges + B))
        # two phases: the draw stays in the handle's staging and reports its sizes, the outputs are then allocated EXACTLY
        # and written once (edge_index compact, row stride = n_edges) — no worst-case buffers, no copies of the used part
        ne, nn = C.c_int64(0), C.c_int64(0)
        rc = lib.tg_sampler_draw(self._h, _p64(s_src), _p64(s_dst), _p64(seeds), B,
                                 fan.ctypes.data_as(C.POINTER(C.c_int32)), len(fan), int(rng_seed) & (2 ** 64 - 1),
                                 self.num_threads, cap, C.byref(ne), C.byref(nn))
        if rc != 0:
            raise RuntimeError(lib.tg_sampler_last_error().decode())
        ne, nn = ne.value, nn.value
        out_eid = np.empty(ne, dtype=np.int64)
        out_ei = np.empty((2, ne), dtype=np.int64)
        out_nodes = np.empty(nn, dtype=np.int64)
        if lib.tg_sampler_emit(self._h, self.num_threads, ne, _p64(out_eid), _p64(out_ei), _p64(out_nodes)) != 0:
            raise RuntimeError(lib.tg_sampler_last_error().decode())
        return torch.from_numpy(out_eid), torch.from_numpy(out_ei), torch.from_numpy(out_nodes)


def host_csr(keys, num_nodes):
    """(rowptr int32 [N+1], perm int32 [M]) of int64 ``keys`` in [0, N): stable counting sort on the host
    (``tg_host_csr``), the structure ``tg_csr_build`` makes on the device."""
    lib = _load()
    keys = np.ascontiguousarray(np.asarray(keys, dtype=np.int64))
    rowptr = np.empty(num_nodes + 1, dtype=np.int32)
    perm = np.empty(max(keys.shape[0], 1), dtype=np.int32)
    rc = lib.tg_host_csr(_p64(keys), keys.shape[0], int(num_nodes), rowptr.ctypes.data_as(C.POINTER(C.c_int32)),
                         perm.ctypes.data_as(C.POINTER(C.c_int32)))
    if rc != 0:
        raise ValueError(lib.tg_sampler_last_error().decode())
    return rowptr, perm


def host_batch_index(edge_index, num_nodes, n_seed):
    """The index structures of one sampled batch as ONE host int32 array + part offsets (``tg_host_batch_index``: stable
    counting sorts, the destination-sorted layout, the seed CSR).  Pure host work that releases the GIL: the sampler's
    worker thread runs it right after ``sample`` so the training thread only uploads."""
    lib = _load()
    ei = np.ascontiguousarray(edge_index.numpy() if isinstance(edge_index, torch.Tensor) else np.asarray(edge_index),
                              dtype=np.int64)
    E = ei.shape[1]
    off = np.zeros(14, dtype=np.int64)
    args = (_p64(ei), E, E, int(n_seed), int(num_nodes))
    if lib.tg_host_batch_index(*args, None, _p64(off)) != 0:
        raise ValueError(lib.tg_sampler_last_error().decode())
    flat = np.empty(int(off[13]), dtype=np.int32)
    if lib.tg_host_batch_index(*args, flat.ctypes.data_as(C.POINTER(C.c_int32)), _p64(off)) != 0:
        raise ValueError(lib.tg_sampler_last_error().decode())
    return flat, off, ei


def host_index_offsets(E, num_nodes, n_seed):
    """Part offsets (int64 [14], in int32 elements; [13] = total) of ``host_batch_index``'s array: a function of
    (E, N, n_seed) alone."""
    lib = _load()
    off = np.zeros(14, dtype=np.int64)
    if lib.tg_host_batch_index(None, int(E), int(E), int(n_seed), int(num_nodes), None, _p64(off)) != 0:
        raise ValueError(lib.tg_sampler_last_error().decode())
    return off


def fill_host_batch_index(ei, num_nodes, n_seed, flat_out):
    """``host_batch_index`` written into a caller-owned int32 buffer (``graph_step.prepare_sample``: a slot of the batch's
    pinned arena).  ``ei``: C-contiguous int64 [2, E]."""
    lib = _load()
    E = ei.shape[1]
    if not (ei.flags.c_contiguous and ei.dtype == np.int64 and flat_out.flags.c_contiguous and flat_out.dtype == np.int32):
        raise ValueError("fill_host_batch_index: contiguous int64 edge_index and int32 output expected")
    off = np.zeros(14, dtype=np.int64)
    if lib.tg_host_batch_index(_p64(ei), E, E, int(n_seed), int(num_nodes), flat_out.ctypes.data_as(C.POINTER(C.c_int32)),
                               _p64(off)) != 0:
        raise ValueError(lib.tg_sampler_last_error().decode())
    if int(off[13]) != flat_out.shape[0]:
        raise ValueError("fill_host_batch_index: output buffer of the wrong size")


def batch_index(edge_index, num_nodes, n_seed, device, prebuilt=None):
    """``ops.BatchIndex`` of one sampled batch: the CSR-by-destination / by-source of the neighbour edges (columns
    ``n_seed:``) and the CSR of the 2B seed endpoints, built on the host next to the sampler and uploaded in ONE transfer
    — what ``SubgraphIndex.build`` / ``SeedIndex`` otherwise rebuild on the device in every forward (SURVEY 8f rank 1:
    the sampler emits the CSR the aggregation kernels read).  ``prebuilt`` = the result of ``host_batch_index`` (made
    in the sampler thread)."""
    from . import ops
    flat_h, off, ei = prebuilt if prebuilt is not None else host_batch_index(edge_index, num_nodes, n_seed)
    flat = torch.from_numpy(flat_h).to(device, non_blocking=True)                                     # ONE upload
    return index_over(flat, off, num_nodes, n_seed, torch.from_numpy(ei).to(device, non_blocking=True))


def index_over(flat, off, num_nodes, n_seed, edge_index):
    """``ops.BatchIndex`` whose parts are views of ``flat`` (device int32, laid out by ``host_batch_index``); the part
    offsets depend on (E, N, n_seed) alone, so a static buffer serves every batch of one shape bucket."""
    from . import ops
    v = [flat[int(off[i]):int(off[i + 1])] for i in range(13)]
    src, dst, rp_d, pm_d, rp_s, pm_s, tei_d, rp_t, pm_t, dst_sorted, src_sorted, inv_d, s2s = v
    graph = ops.SubgraphIndex(src, dst, (rp_d, pm_d), (rp_s, pm_s), int(num_nodes))
    graph._sorted = dict(perm=pm_d, dst=dst_sorted, src=src_sorted, inv=inv_d, src_to_sorted=s2s)
    seeds = ops.SeedIndex.from_parts(tei_d, rp_t, pm_t, int(n_seed), int(num_nodes))
    return ops.BatchIndex(graph, seeds, edge_index)


class ColumnStore:
    """Raw edge table (dict stype -> tensor [E_total, ...]) + labels, node table (dict stype -> [V, ...]); tensors may
    live on the host or on the MI355X (then the per-batch gather runs on the device and nothing crosses PCIe but ids)."""

    def __init__(self, edge_feats, edge_cols, node_feats, node_cols, labels):
        self.edge_feats, self.edge_cols = edge_feats, edge_cols
        self.node_feats, self.node_cols = node_feats, node_cols
        self.labels = labels

    def to(self, device):
        return ColumnStore({k: v.to(device) for k, v in self.edge_feats.items()}, self.edge_cols,
                           {k: v.to(device) for k, v in self.node_feats.items()}, self.node_cols,
                           self.labels.to(device))

    @property
    def device(self):
        return self.labels.device

    def graph_inputs(self, sampler: NeighborSampler, seed_eids, rng_seed=0, lazy=None, index=False):
        """``get_graph_inputs`` (ibm…py:159-180): (node_tf, edge_index, edge_tf, y) with the seed edges first.
        ``lazy`` (default: when the store is on the GPU): the TensorFrames carry the sampled ids (``row_ids``) over the
        whole HBM-resident table and the stype encoders read the raw columns by id — ``tensor_frame[idx]``
        (ibm…py:163,168) without ever materialising the gathered rows; ``lazy=False`` gathers them (index_select).
        ``edge_index`` is the plain int64 [2, E] tensor every wrapper of ``utils.py`` takes (drop-in for main.py:48);
        ``index=True`` (opt-in, GPU store, ``TABGNNFusedS`` only) hands over an ``ops.BatchIndex`` instead: the same
        tensor plus the batch's CSRs built on the host next to the sampler."""
        eid, edge_index, nodes = sampler.sample(seed_eids, rng_seed)
        return self.batch(eid, edge_index, nodes, len(seed_eids), lazy, index)

    def batch(self, eid, edge_index, nodes, n_seed, lazy=None, index=False):
        """``index``: False (plain ``edge_index``), True (build the ``ops.BatchIndex`` here) or the host-side result of
        ``host_batch_index`` made by the sampler thread (only the upload happens here)."""
        dev = self.device
        lazy = (dev.type == "cuda") if lazy is None else lazy
        eid_d, nodes_d = eid.to(dev, non_blocking=True), nodes.to(dev, non_blocking=True)
        if lazy:
            edge_tf = TensorFrame(self.edge_feats, self.edge_cols, None, eid_d)
            node_tf = TensorFrame(self.node_feats, self.node_cols, None, nodes_d)
        else:
            edge_tf = TensorFrame({k: v.index_select(0, eid_d) for k, v in self.edge_feats.items()}, self.edge_cols)
            node_tf = TensorFrame({k: v.index_select(0, nodes_d) for k, v in self.node_feats.items()}, self.node_cols)
        y = self.labels.index_select(0, eid_d[:n_seed])
        if index is not False and index is not None:      # index structures built on the host too: the model skips its CSR kernels
            if dev.type != "cuda":
                raise ValueError("index=True needs a store on the GPU (ops.BatchIndex holds device CSRs)")
            pre = index if isinstance(index, tuple) else None
            return node_tf, batch_index(edge_index, nodes.numel(), n_seed, dev, pre), edge_tf, y
        return node_tf, edge_index.to(dev, non_blocking=True), edge_tf, y

    def lp_inputs(self, sampler: NeighborSampler, seed_eids, num_neg_samples=64, rng_seed=0):
        """``lp_inputs`` (``src/utils/batch_processing.py:104-147``) for link-prediction pre-training:
        (node_tf, edge_index, edge_tf, neigh_edge_index, neigh_edge_tf, target_edge_index, target_edge_tf) where the
        targets are the B positive (seed) edges followed by the sampled negatives, and every positive's raw row is
        repeated ``num_neg_samples`` times (contiguously) behind the positives' rows for its negatives."""
        node_tf, edge_index, edge_tf, _ = self.graph_inputs(sampler, seed_eids, rng_seed, lazy=False)
        B = len(seed_eids)
        pos = edge_index[:, :B]
        neg = generate_negative_samples(edge_index, pos, num_neg_samples, seed=rng_seed).to(edge_index.device)
        rep = torch.arange(B, device=edge_index.device).repeat_interleave(int(num_neg_samples))
        rows = torch.cat([torch.arange(B, device=edge_index.device), rep])
        target_tf = TensorFrame({k: v.index_select(0, rows) for k, v in edge_tf.feat_dict.items()}, self.edge_cols)
        keep = torch.arange(B, edge_index.shape[1], device=edge_index.device)
        neigh_tf = TensorFrame({k: v.index_select(0, keep) for k, v in edge_tf.feat_dict.items()}, self.edge_cols)
        return (node_tf, edge_index, edge_tf, edge_index[:, B:], neigh_tf, torch.cat([pos, neg], dim=1), target_tf)


class ShardedSeedLoader:
    """Seed-edge mini-batches for rank r of `world`: one shuffle of the train edge ids per epoch (same permutation on
    every rank: seeded by `seed + epoch`), rank r takes every world-th id, full batches only — the reference's
    ``DataLoader(train_ids, batch_size, shuffle=True)`` (``src/datasets/util/graph.py:38-53``) made rank-disjoint,
    and without the short last batch ``TABGNNFusedS.forward`` would mis-slice (``utils.py:355-356``)."""

    def __init__(self, seed_ids, batch_size, rank=0, world=1, seed=0):
        self.ids = np.ascontiguousarray(np.asarray(seed_ids, dtype=np.int64))
        self.batch_size, self.rank, self.world, self.seed = int(batch_size), int(rank), int(world), int(seed)
        if not 0 <= self.rank < self.world:
            raise ValueError("rank must be in [0, world)")
        self.epoch = 0

    def set_epoch(self, epoch):
        self.epoch = int(epoch)

    def __len__(self):
        return (len(self.ids) // self.world) // self.batch_size

    def __iter__(self):
        perm = np.random.default_rng(self.seed + self.epoch).permutation(len(self.ids))
        per_rank = len(self.ids) // self.world                      # equal share on every rank (tail ids dropped)
        mine = self.ids[perm[self.rank:per_rank * self.world:self.world]]
        for i in range(len(self)):
            yield mine[i * self.batch_size:(i + 1) * self.batch_size]


def generate_negative_samples(edge_index, pos_edge_index, num_neg_samples, seed=0, num_threads=0):
    """Drop-in for the reference's pybind11 ``negative_sampling.generate_negative_samples(edge_index, pos_edge_index,
    num_neg_samples)`` (``negative_sampling.cpp:10-12,78-81``; call site ``src/utils/batch_processing.py:145``):
    same argument meaning (2 x E and 2 x B nested lists, arrays or tensors of local node ids), same
    ``ValueError`` for ``num_neg_samples <= 0``, same output layout — returned as an int64 tensor ``[2, B*2*(k//2)]``
    instead of nested lists — plus a ``seed`` (the reference is unseeded)."""
    lib = _load()
    if not hasattr(lib.tg_negative_sample, "_bound"):
        i64p = C.POINTER(C.c_int64)
        lib.tg_negative_sample.argtypes = [i64p, i64p, C.c_int64, i64p, i64p, C.c_int64, C.c_int32, C.c_uint64,
                                           C.c_int32, i64p, i64p]
        lib.tg_negative_sample.restype = C.c_int
        lib.tg_negative_sample._bound = True
    as_np = lambda a: np.ascontiguousarray(a.cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a), dtype=np.int64)
    ei, pos = as_np(edge_index), as_np(pos_edge_index)
    if ei.ndim != 2 or ei.shape[0] != 2 or pos.ndim != 2 or pos.shape[0] != 2:
        raise ValueError("edge_index and pos_edge_index must be [2, E] and [2, B]")
    k = int(num_neg_samples)
    B, per = pos.shape[1], 2 * (max(k, 0) // 2)
    out = np.empty((2, B * per), dtype=np.int64)
    src, dst, ps, pd = (np.ascontiguousarray(a) for a in (ei[0], ei[1], pos[0], pos[1]))
    rc = lib.tg_negative_sample(_p64(src), _p64(dst), ei.shape[1], _p64(ps), _p64(pd), B, k,
                                int(seed) & (2 ** 64 - 1), int(num_threads), _p64(out[0]), _p64(out[1]))
    if rc == 1:
        raise ValueError(lib.tg_sampler_last_error().decode())
    if rc != 0:
        raise RuntimeError(lib.tg_sampler_last_error().decode())
    return torch.from_numpy(out)


def edge_ports(edge_index, timestamps=None, num_nodes=None, num_threads=0):
    """Port numbers of every edge: replaces ``to_adj_nodes_with_times`` + ``ports`` + the two calls of ``add_ports``
    (``src/datasets/util/graph.py:68-101``; 22 s of Python on the reference's dummy file).  Returns
    ``(in_ports, out_ports)`` as float32 ``[E, 1]`` tensors like the reference's ``ports`` (graph.py:82)."""
    lib = _load()
    if not hasattr(lib.tg_edge_ports, "_bound"):
        i64p, i32p = C.POINTER(C.c_int64), C.POINTER(C.c_int32)
        lib.tg_edge_ports.argtypes = [i64p, i64p, i64p, C.c_int64, C.c_int64, C.c_int32, i32p, i32p]
        lib.tg_edge_ports.restype = C.c_int
        lib.tg_edge_ports._bound = True
    as_np = lambda a: np.ascontiguousarray(a.cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a), dtype=np.int64)
    ei = as_np(edge_index)
    if ei.ndim != 2 or ei.shape[0] != 2:
        raise ValueError("edge_index must be [2, E]")
    E = ei.shape[1]
    N = int(num_nodes) if num_nodes is not None else (int(ei.max()) + 1 if E else 0)
    src, dst = np.ascontiguousarray(ei[0]), np.ascontiguousarray(ei[1])
    ts = None
    if timestamps is not None:
        ts = as_np(timestamps).reshape(-1)          # graph.py:72,75: int(t)
        if ts.shape[0] != E:
            raise ValueError("timestamps must hold one value per edge")
    inp, outp = np.empty(E, dtype=np.int32), np.empty(E, dtype=np.int32)
    p32 = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
    rc = lib.tg_edge_ports(_p64(src), _p64(dst), _p64(ts) if ts is not None else None, E, N, int(num_threads),
                           p32(inp), p32(outp))
    if rc != 0:
        raise ValueError(lib.tg_sampler_last_error().decode())
    f = lambda a: torch.from_numpy(a.astype(np.float32)).reshape(-1, 1)
    return f(inp), f(outp)


def add_ego_ids(x, seed_edge_index, column="EgoID"):
    """``add_EgoIDs`` (src/datasets/util/graph.py:121-145): the ``EgoID`` relation column of the node frame is 1 for
    the endpoints of the seed edges and 0 elsewhere; written in place on the frame's device, no ``unique``."""
    from .frame import stype
    rel = x.feat_dict[stype.relation]
    idx = x.col_names_dict[stype.relation].index(column)
    rel[:, idx] = 0
    rel[seed_edge_index.reshape(-1).to(rel.device), idx] = 1
    return x


def add_ego_ids_from_nodes(x, batch_size, column="EgoID"):
    """``add_EgoIDs_from_nodes`` (graph.py:110-119): the first ``batch_size`` node rows are the seeds."""
    from .frame import stype
    rel = x.feat_dict[stype.relation]
    idx = x.col_names_dict[stype.relation].index(column)
    rel[:, idx] = 0
    rel[:batch_size, idx] = 1
    return x
