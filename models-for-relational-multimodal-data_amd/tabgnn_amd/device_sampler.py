"""``DeviceNeighborSampler`` — the k-hop edge-seeded sampler + relabel of ``sampler.NeighborSampler`` running on the GPU
over the HBM-resident graph (``csrc/sampler_gpu.hip``): ``sample_neighbors`` +
``get_graph_inputs`` of ``src/datasets/ibm_transactions_for_aml.py:61-112,159-180`` without the host in the loop.
Same output contract as the host sampler — ``(eid, edge_index, nodes)``: seed edges first and in order, sampled non-seed
edges after them, ``nodes`` the sorted unique endpoints, ``edge_index`` their ranks — as device tensors."""
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

    def sample(self, seed_eids, rng_seed=0):
        """-> (eid int64 [E_out], edge_index int64 [2, E_out] local ids, nodes int64 [N_out] sorted global ids), on the
        device.  One host read-back of the two output sizes per call (the caller allocates exact outputs)."""
        seeds = torch.as_tensor(seed_eids, dtype=torch.int64).to(self.device).contiguous()
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
        ne, nn, _, err = (int(v) for v in counts.tolist())
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
