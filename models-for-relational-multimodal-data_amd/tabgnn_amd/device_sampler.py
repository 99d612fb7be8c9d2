"""``DeviceNeighborSampler`` — the k-hop edge-seeded sampler + relabel of ``sampler.NeighborSampler`` running on the GPU
over the HBM-resident graph (``csrc/sampler_gpu.hip``): ``sample_neighbors`` +
``get_graph_inputs`` of ``src/datasets/ibm_transactions_for_aml.py:61-112,159-180`` without the host in the loop.
Same output contract as the host sampler — ``(eid, edge_index, nodes)``: seed edges first and in order, sampled non-seed
edges after them, ``nodes`` the sorted unique endpoints, ``edge_index`` their ranks — as device tensors.
``DeviceBatchLoader`` makes it the data path of the training loop: batches are drawn, indexed (or padded into their
HIP-graph bucket) one step ahead on a side stream, so the training stream runs no sampler or index kernel and the host
never waits for a size."""
from __future__ import annotations

import ctypes

import torch

from . import _lib as L


class DeviceNeighborSampler:
    def __init__(self, edge_index, num_nodes, num_neighbors=(100, 100), device="cuda:0"):
        ei = torch.as_tensor(edge_index, dtype=torch.int64)
        if ei.dim() != 2 or ei.shape[0] != 2:
            raise ValueError("edge_index must be [2, E]")
        dev = torch.device(device)
        self.device = dev
        self.num_nodes, self.num_edges = int(num_nodes), int(ei.shape[1])
        if self.num_edges == 0 or self.num_edges >= 2 ** 31 - 1 or self.num_nodes >= 2 ** 31 - 1:
            raise ValueError("graph sizes must fit int32")
        if ei.numel() and (int(ei.min()) < 0 or int(ei.max()) >= self.num_nodes):
            raise ValueError("edge endpoint out of range")
        self.fanout = [int(f) for f in num_neighbors]
        if any(f > 128 for f in self.fanout):
            raise ValueError("fan-out per hop above 128 is not supported on the device (use sampler.NeighborSampler)")
        self._fan = (ctypes.c_int32 * len(self.fanout))(*self.fanout)
        ei = ei.to(dev).contiguous()
        self.src, self.dst = ei[0].contiguous(), ei[1].contiguous()
        # CSC: stable counting sort of the destination column (in-edges of v in ascending edge id)
        key = self.dst.to(torch.int32)
        self.colptr = torch.empty(self.num_nodes + 1, dtype=torch.int32, device=dev)
        self.in_eid = torch.empty(self.num_edges, dtype=torch.int32, device=dev)
        work = torch.empty(L.load().tg_csr_workspace_ints(self.num_edges, self.num_nodes), dtype=torch.int32, device=dev)
        L.call("tg_csr_build", L.ptr(key), self.num_edges, self.num_nodes, L.ptr(self.colptr), L.ptr(self.in_eid), L.ptr(work),
               L.stream())
        self.in_src = self.src.to(torch.int32)[self.in_eid.long()].contiguous()
        self._seedbit = torch.zeros(L.load().tg_gsampler_seedbit_bytes(self.num_edges), dtype=torch.uint8, device=dev)
        self._ws, self._cap = None, 0

    def _bound(self, B):
        n, tot = 2 * B, B
        for f in self.fanout:
            if f < 0:
                return self.num_edges + B
            n *= f
            tot += n
            if tot > self.num_edges + B:
                return self.num_edges + B
        return min(tot, self.num_edges + B)

    def draw_async(self, seed_eids, rng_seed=0):
        """The k-hop draw on the CURRENT stream, its two output sizes on their way to pinned host memory: no
        synchronisation.  ``emit`` (same stream) finishes the batch once the sizes have landed — a loader calls it one
        step later, when they long have (``DeviceBatchLoader``)."""
        seeds = torch.as_tensor(seed_eids, dtype=torch.int64).to(self.device, non_blocking=True).contiguous()
        B = int(seeds.shape[0])
        if B == 0:
            raise ValueError("need at least one seed edge")
        cap = self._bound(B)
        if cap > 8 * 1024 * 1024:
            raise ValueError("sample bound above the device sampler's staging limit (8 Mi edges)")
        if self._ws is None or self._cap < cap:
            self._ws = torch.empty(L.load().tg_gsampler_workspace_bytes(self.num_nodes, cap), dtype=torch.uint8, device=self.device)
            self._cap = cap
        counts = torch.empty(4, dtype=torch.int64, device=self.device)
        L.call("tg_gsampler_draw", L.ptr(seeds), B, L.ptr(self.src), L.ptr(self.dst), self.num_edges, L.ptr(self.colptr),
               L.ptr(self.in_src), L.ptr(self.in_eid), self.num_nodes, ctypes.addressof(self._fan), len(self.fanout),
               int(rng_seed) & (2 ** 64 - 1), self._cap, L.ptr(self._seedbit), L.ptr(self._ws), L.ptr(counts), L.stream())
        host = torch.empty(4, dtype=torch.int64, pin_memory=True)
        host.copy_(counts, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        return seeds, B, counts, host, ev

    def emit(self, drawn):
        """-> (eid, edge_index, nodes) of a ``draw_async`` handle, on the current stream (the draw's stream, or one that
        waits for it).  The host waits for the SIZE copy only."""
        seeds, B, counts, host, ev = drawn
        ev.synchronize()
        ne, nn, _, err = (int(v) for v in host.tolist())
        if err != 0:
            self._seedbit.zero_()                      # the draw may have set seed bits before it found the bad id
            raise ValueError({1: "seed edge id out of range", 2: "edge endpoint out of range",
                              3: "staging capacity too small"}.get(err, f"device sampler error {err}"))
        out_eid = torch.empty(ne, dtype=torch.int64, device=self.device)          # ne >= B >= 1
        out_ei = torch.empty(2, ne, dtype=torch.int64, device=self.device)
        out_nodes = torch.empty(nn, dtype=torch.int64, device=self.device)
        L.call("tg_gsampler_emit", L.ptr(seeds), B, self.num_nodes, self.num_edges, self._cap, ne, L.ptr(self._seedbit),
               L.ptr(self._ws), L.ptr(out_eid), L.ptr(out_ei), L.ptr(out_nodes), L.stream())
        return out_eid, out_ei, out_nodes

    def sample(self, seed_eids, rng_seed=0):
        """-> (eid int64 [E_out], edge_index int64 [2, E_out] local ids, nodes int64 [N_out] sorted global ids), on the
        device.  One host read-back of the two output sizes per call (the caller allocates exact outputs)."""
        return self.emit(self.draw_async(seed_eids, rng_seed))


def device_batch_index(edge_index, num_nodes, n_seed, flat=None, off=None):
    """``ops.BatchIndex`` of a DEVICE-resident batch (``edge_index`` int64 [2, E] on the GPU, seed edges first): the int32
    endpoints, the by-destination / by-source CSRs of the neighbour edges, the seed CSR and the destination-sorted views
    — the 13 parts ``sampler.host_batch_index`` builds on the host, in the same flat int32 layout (``host_index_offsets``:
    a function of (E, N, n_seed) alone), built by the index kernels (``tg_ids_to_i32``, ``tg_csr_build``: stable counting
    sorts, the same permutations as the host's).  ``flat``: write into this buffer (a bucket's static arena slot)."""
    from .sampler import host_index_offsets, index_over
    dev = edge_index.device
    E, N, B = int(edge_index.shape[1]), int(num_nodes), int(n_seed)
    En = E - B
    if off is None:
        off = host_index_offsets(E, N, B)
    if flat is None:
        flat = torch.empty(int(off[13]), dtype=torch.int32, device=dev)
    v = [flat[int(off[i]):int(off[i + 1])] for i in range(13)]
    src, dst, rp_d, pm_d, rp_s, pm_s, tei, rp_t, pm_t, dst_s, src_s, inv, s2s = v
    err = torch.zeros(1, dtype=torch.int32, device=dev)
    lib = L.load()
    st = L.stream()
    row0, row1 = edge_index[0], edge_index[1]
    if En > 0:
        L.call("tg_ids_to_i32", row0[B:].data_ptr(), En, N, L.ptr(src), L.ptr(err), st)
        L.call("tg_ids_to_i32", row1[B:].data_ptr(), En, N, L.ptr(dst), L.ptr(err), st)
    if B > 0:
        L.call("tg_ids_to_i32", row0.data_ptr(), B, N, tei.data_ptr(), L.ptr(err), st)
        L.call("tg_ids_to_i32", row1.data_ptr(), B, N, tei.data_ptr() + 4 * B, L.ptr(err), st)
    work = torch.empty(lib.tg_csr_workspace_ints(max(En, 2 * B, 1), N), dtype=torch.int32, device=dev)
    for key, M, rp, pm in ((dst, En, rp_d, pm_d), (src, En, rp_s, pm_s), (tei, 2 * B, rp_t, pm_t)):
        L.call("tg_csr_build", key.data_ptr() if M else None, M, N, L.ptr(rp), L.ptr(pm), L.ptr(work), st)
    if En > 0:
        perm = pm_d[:En].long()
        torch.index_select(dst, 0, perm, out=dst_s)
        torch.index_select(src, 0, perm, out=src_s)
        inv.index_copy_(0, perm, torch.arange(En, dtype=torch.int32, device=dev))
        torch.index_select(inv, 0, pm_s[:En].long(), out=s2s)
    idx = index_over(flat, off, N, B, edge_index)
    idx.graph.err = err
    return idx


class DeviceBatchLoader:
    """Seed batches -> training batches that never leave the GPU, prepared one step AHEAD on a side stream:

        take k:   [host] sizes of draw k+1 have landed (launched at take k-1)  ->  [side stream] emit k+1, its index
                  structures (or its padded bucket arena), then draw k+2       ->  [main stream] waits for batch k's
                  ready event, which was recorded a whole step ago

    so the training stream sees no sampler kernel, no CSR build (``tg_csr_build`` runs on the side stream, under the
    previous step) and no size read-back, and the host never blocks on the GPU.  ``mode``: "index" -> (node_tf,
    ops.BatchIndex, edge_tf, y) for ``train_step``; "bucket" -> ``graph_step.Prepared`` for ``GraphedTrainStep`` (padding
    and index parts built on the device, ``prepare_sample_device``).  Reference loop: ``AMLData.sample_neighbors`` +
    ``get_graph_inputs``, ``src/datasets/ibm_transactions_for_aml.py:61-112,159-180``."""

    def __init__(self, sampler, store, seed_batches, mode="index", rng_seed=0):
        self.sampler, self.store, self.mode = sampler, store, mode
        self.it = iter(seed_batches)
        self.rng = int(rng_seed)
        self.side = torch.cuda.Stream(device=sampler.device)
        self.drawn = self.built = None
        self.n = 0
        self._advance()          # draw 0
        self._advance()          # emit + index 0, draw 1

    def _next_seeds(self):
        try:
            return next(self.it)
        except StopIteration:
            return None

    def _advance(self):
        """One pipeline step on the side stream: finish the drawn batch, start the next draw."""
        main = torch.cuda.current_stream(self.sampler.device)
        with torch.cuda.stream(self.side):
            built = None
            if self.drawn is not None:
                eid, ei, nodes = self.sampler.emit(self.drawn)
                B = self.drawn[1]
                built = self._finish(eid, ei, nodes, B)
                ev = torch.cuda.Event()
                ev.record()
                built = (built, ev, (eid, ei, nodes))
            seeds = self._next_seeds()
            self.drawn = None
            if seeds is not None:
                self.drawn = self.sampler.draw_async(seeds, self.rng + self.n)
                self.n += 1
        self.built, out = built, self.built
        return out

    def _finish(self, eid, ei, nodes, B):
        if self.mode == "bucket":
            y = self.store.labels.index_select(0, eid[:B])
            return prepare_sample_device(eid, ei, nodes, y, B)
        node_tf, _, edge_tf, y = self.store.batch(eid, ei, nodes, B, lazy=True, index=False)
        return node_tf, device_batch_index(ei, int(nodes.shape[0]), B), edge_tf, y

    def take(self):
        """The next batch (None when the seed iterator is exhausted), ready for the current stream."""
        out = self._advance()
        if out is None:
            return None
        batch, ev, raw = out
        main = torch.cuda.current_stream(self.sampler.device)
        main.wait_event(ev)
        for t in raw:                                    # allocated on the side stream, consumed on this one
            t.record_stream(main)
        if self.mode == "bucket":
            batch.arena.record_stream(main)
        else:
            batch[1].graph.src.record_stream(main)       # (the flat index buffer: every part is a view of it)
            batch[3].record_stream(main)
        return batch

    def __iter__(self):
        while True:
            b = self.take()
            if b is None:
                return
            yield b


def prepare_sample_device(eid, edge_index, nodes, y, n_seed, key=None):
    """``graph_step.prepare_sample`` for a batch that is already on the GPU (``DeviceNeighborSampler``): the same bucket
    arena — index parts, padded ``edge_index`` (padding edges = self loops on padding nodes), real node count, labels,
    padded row ids — built by device kernels into ONE device buffer; ``GraphedTrainStep`` then moves it into the bucket's
    static buffers with one device-to-device copy.  Only the two sizes (known on the host since ``emit``) pick the bucket."""
    from . import graph_step as G
    dev = edge_index.device
    E, N, B = int(eid.shape[0]), int(nodes.shape[0]), int(n_seed)
    e_pad, n_pad = key if key is not None else (G.bucket_size(E), G.bucket_size(N + 1))
    if e_pad < E or n_pad < N or (e_pad > E and n_pad == N):
        raise ValueError("bucket smaller than the batch (padding edges need at least one padding node)")
    if int(y.shape[0]) != B:
        raise ValueError(f"{int(y.shape[0])} labels for n_seed={B}")
    layout, nbytes, off = G._sample_layout(e_pad, n_pad, B)
    arena = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    view = G._views(arena, layout)
    ei = view["ei"]
    ei[:, :E] = edge_index
    if e_pad > E:
        loops = N + torch.arange(e_pad - E, dtype=torch.int64, device=dev) % (n_pad - N)
        ei[0, E:] = loops
        ei[1, E:] = loops
    device_batch_index(ei, n_pad, B, flat=view["flat"], off=off)
    view["n_real"].fill_(N)
    view["y"].copy_(y.reshape(-1))
    view["node.ids"][:N] = nodes
    view["node.ids"][N:] = nodes[0]
    view["edge.ids"][:E] = eid
    view["edge.ids"][E:] = eid[0]
    return G.Prepared((e_pad, n_pad), arena, layout, off, E, N, True)
