// k-hop neighbour sampler + relabel over the HBM-resident graph ("the sampler over the resident CSR").
//
// Same semantics as the host sampler (csrc/sampler.cpp, which restates sample_neighbors + get_graph_inputs,
// src/datasets/ibm_transactions_for_aml.py:61-112,159-180): hop h expands every node of the current frontier once,
// drawing min(in-degree, fanout[h]) of its incoming edges uniformly without replacement (Floyd); new source nodes form
// the next frontier; output = the seed edges in their order, then the sampled edges that are not seed edges; nodes =
// sorted unique endpoints, edge_index relabelled to ranks among them.  The draw of node v at hop h depends only on
// (rng_seed, h, v), and every list is produced by walking bitmaps / prefix sums in index order, so a sample is a pure
// function of its inputs (no atomics decide an order).  The reference's own sampler is unseeded: parity is structural
// (tests/test_gpu_sampler.py), as for the host sampler.
//
// Data layout: the graph as CSC — colptr int32 [N+1], in_src / in_eid int32 [E] (in-edges of node v at
// colptr[v] .. colptr[v+1], ascending edge id: tg_csr_build of the destination column).  Per-call scratch lives in ONE
// caller-owned workspace (tg_gsampler_workspace_bytes): node bitmaps (visited, frontier, next), the seed-edge bitmap,
// per-word popcount prefixes, the frontier list with its draw counts / offsets, and the staging of the drawn edges
// (src, dst, eid, keep) with the compaction offsets.  Everything is a streaming pass over at most N / 32 words or the
// drawn edges; the draws themselves are one wave per frontier node.
#include "../../include/tabgnn_hip.h"
#include "common.hpp"

namespace tg {

constexpr int SG_SCAN_ELEMS = 8192;      // elements per block of the two-level scan (1024 threads x 8)
constexpr int SG_SCAN_PT = SG_SCAN_ELEMS / 1024;
constexpr int SG_MAX_FAN = 128;          // fan-out per hop the Floyd draw keeps in two registers per lane

struct SgWs {            // carved out of the caller's workspace (all 16-byte aligned)
  unsigned *visited, *frontier, *next, *seedbit;
  int *wcnt, *wpre;      // [nw + 1] popcounts of a node bitmap and their exclusive prefix
  int *flist, *fcnt, *foff;          // frontier nodes [N], draws per node [N + 1], exclusive prefix [N + 1]
  int *e_src, *e_dst, *e_eid, *keep, *koff;     // staging [cap] (+1 for koff)
  int *tops;             // block totals of the scans [<= 1024 * SG_TOPS_PT + 1]
  long long* counts;     // [4]: staged edges, kept edges, nodes, error flag
};

__device__ __forceinline__ unsigned long long sg_mix(unsigned long long z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}

// ------------------------------------------------------------------ two-level exclusive scan of int32 (n <= 67 M)
// n_dev (optional): the live length sits in device memory (the staged-edge count); entries at or past it read as zero
__device__ __forceinline__ int sg_len(int n, const long long* n_dev) {
  if (!n_dev) return n;
  const long long m = *n_dev;
  return m < n ? (int)m : n;
}
__global__ void __launch_bounds__(1024) k_sg_scan_block(const int* __restrict__ in, int* __restrict__ out, int n_,
                                                        const long long* __restrict__ n_dev, int* __restrict__ tops) {
  __shared__ int wsum[16];
  const int n = sg_len(n_, n_dev);
  const int base = blockIdx.x * SG_SCAN_ELEMS + threadIdx.x * SG_SCAN_PT;
  int v[SG_SCAN_PT], s = 0;
#pragma unroll
  for (int j = 0; j < SG_SCAN_PT; ++j) { v[j] = base + j < n ? in[base + j] : 0; s += v[j]; }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int inc = s;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  if (wave == 0) {
    int w = lane < 16 ? wsum[lane] : 0, winc = w;
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
      const int t = __shfl_up(winc, o, 64);
      if (lane >= o) winc += t;
    }
    if (lane < 16) wsum[lane] = winc - w;           // exclusive prefix of the wave sums
    if (lane == 15) tops[blockIdx.x] = winc;        // block total
  }
  __syncthreads();
  int ex = wsum[wave] + inc - s;
#pragma unroll
  for (int j = 0; j < SG_SCAN_PT; ++j) {
    if (base + j < n) out[base + j] = ex;
    ex += v[j];
  }
}
// exclusive scan of the block totals in place (nb <= 1024 * SG_TOPS_PT: eight consecutive totals per thread), grand total
// to tops[nb] and to *total_out (optional)
constexpr int SG_TOPS_PT = 8;            // -> scans of up to 8192 * 8192 = 67 M elements (the 10 M-node graph of configs[4])
__global__ void __launch_bounds__(1024) k_sg_scan_tops(int* __restrict__ tops, int nb, long long* __restrict__ total_out) {
  __shared__ int wsum[16];
  const int b0 = (int)threadIdx.x * SG_TOPS_PT;
  int tv[SG_TOPS_PT], v = 0;
#pragma unroll
  for (int j = 0; j < SG_TOPS_PT; ++j) { tv[j] = b0 + j < nb ? tops[b0 + j] : 0; v += tv[j]; }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  if (wave == 0) {
    int w = lane < 16 ? wsum[lane] : 0, winc = w;
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
      const int t = __shfl_up(winc, o, 64);
      if (lane >= o) winc += t;
    }
    if (lane < 16) wsum[lane] = winc - w;
    if (lane == 15) { tops[nb] = winc; if (total_out) *total_out = winc; }
  }
  __syncthreads();
  int ex = wsum[wave] + inc - v;
#pragma unroll
  for (int j = 0; j < SG_TOPS_PT; ++j) {
    if (b0 + j < nb) tops[b0 + j] = ex;
    ex += tv[j];
  }
}
__global__ void __launch_bounds__(1024) k_sg_scan_add(int* __restrict__ out, int n_, const long long* __restrict__ n_dev,
                                                      const int* __restrict__ tops, int nb) {
  const int n = sg_len(n_, n_dev);
  const int add = tops[blockIdx.x];
  const int base = blockIdx.x * SG_SCAN_ELEMS + threadIdx.x * SG_SCAN_PT;
#pragma unroll
  for (int j = 0; j < SG_SCAN_PT; ++j)
    if (base + j < n) out[base + j] += add;
  if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = tops[nb];       // out has n + 1 entries: the total closes it
}
// out[0..n] = exclusive prefix of in[0..n) (out[n] = total); tops: >= nb + 1 ints; n <= 1024 * SG_TOPS_PT * SG_SCAN_ELEMS
static int sg_scan(const int* in, int* out, int n, const long long* n_dev, int* tops, long long* total_out, hipStream_t st) {
  const int nb = n > 0 ? (n + SG_SCAN_ELEMS - 1) / SG_SCAN_ELEMS : 1;
  hipLaunchKernelGGL(k_sg_scan_block, dim3(nb), dim3(1024), 0, st, in, out, n, n_dev, tops);
  hipLaunchKernelGGL(k_sg_scan_tops, dim3(1), dim3(1024), 0, st, tops, nb, total_out);
  hipLaunchKernelGGL(k_sg_scan_add, dim3(nb), dim3(1024), 0, st, out, n, n_dev, (const int*)tops, nb);
  return 0;
}

// ------------------------------------------------------------------ bitmaps
__global__ void k_sg_popc(const unsigned* __restrict__ bits, int nw, int* __restrict__ cnt) {
  const int w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w < nw) cnt[w] = __popc(bits[w]);
}
// the set bits of `bits` in ascending order: list[pre[w] + rank] = 32 w + bit
__global__ void k_sg_expand(const unsigned* __restrict__ bits, int nw, const int* __restrict__ pre, int* __restrict__ list) {
  const int w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= nw) return;
  unsigned b = bits[w];
  int o = pre[w];
  while (b) {
    const int i = __ffs((int)b) - 1;
    list[o++] = 32 * w + i;
    b &= b - 1;
  }
}

// seed edges: staged first, in order; their endpoints are visited and form the first frontier; their ids are marked
__global__ void k_sg_seeds(const long long* __restrict__ seeds, int B, const long long* __restrict__ esrc,
                           const long long* __restrict__ edst, long long E, int N, SgWs w) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  const long long e = seeds[i];
  if (e < 0 || e >= E) { w.counts[3] = 1; return; }
  const long long s = esrc[e], d = edst[e];
  if (s < 0 || s >= N || d < 0 || d >= N) { w.counts[3] = 2; return; }
  w.e_src[i] = (int)s; w.e_dst[i] = (int)d; w.e_eid[i] = (int)e; w.keep[i] = 1;
  atomicOr(w.seedbit + (e >> 5), 1u << (e & 31));
  atomicOr(w.visited + (s >> 5), 1u << (s & 31));
  atomicOr(w.visited + (d >> 5), 1u << (d & 31));
  atomicOr(w.frontier + (s >> 5), 1u << (s & 31));
  atomicOr(w.frontier + (d >> 5), 1u << (d & 31));
}

__global__ void k_sg_count(const int* __restrict__ flist, const long long* __restrict__ nf_p, const int* __restrict__ colptr,
                           int fan, int* __restrict__ fcnt, int nmax) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nmax) return;
  const int nf = (int)*nf_p;
  int c = 0;
  if (i < nf) {
    const int v = flist[i];
    const int deg = colptr[v + 1] - colptr[v];
    c = fan < 0 || deg < fan ? deg : fan;
  }
  fcnt[i] = c;
}

// One wave per frontier node: its draws land at staging positions base + foff[i] .. in increasing draw order.
// deg <= fan: every in-edge.  Else Floyd's k distinct positions: for j = deg - k .. deg - 1: t = rand(0 .. j); take t, or j
// when t was taken before — the chosen set sits two entries per lane, membership is one ballot.
__global__ void __launch_bounds__(256) k_sg_draw(const int* __restrict__ flist, const long long* __restrict__ nf_p,
                                                 const int* __restrict__ colptr, const int* __restrict__ in_src,
                                                 const int* __restrict__ in_eid, int fan, int hop,
                                                 unsigned long long rng_seed, const int* __restrict__ foff,
                                                 const long long* __restrict__ base_p, long long cap, SgWs w) {
  const int lane = threadIdx.x & 63;
  const int nf = (int)*nf_p;
  const long long base = *base_p;
  const int waves = gridDim.x * (blockDim.x >> 6);
  for (int i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); i < nf; i += waves) {
    const int v = flist[i];
    const int c0 = colptr[v], deg = colptr[v + 1] - c0;
    const int k = fan < 0 || deg < fan ? deg : fan;
    const long long o = base + foff[i];
    if (o + k > cap) { if (lane == 0) w.counts[3] = 3; continue; }          // (host sized cap as an upper bound)
    if (k == deg) {
      for (int j = lane; j < deg; j += 64) {
        const int s = in_src[c0 + j], e = in_eid[c0 + j];
        w.e_src[o + j] = s; w.e_dst[o + j] = v; w.e_eid[o + j] = e;
        w.keep[o + j] = (w.seedbit[e >> 5] >> (e & 31)) & 1u ? 0 : 1;
        const unsigned bit = 1u << (s & 31);
        const unsigned old = atomicOr(w.visited + (s >> 5), bit);
        if (!(old & bit)) atomicOr(w.next + (s >> 5), bit);
      }
      continue;
    }
    int c_lo = -1, c_hi = -1;                 // chosen positions n = lane and n = lane + 64
    unsigned long long st = sg_mix(rng_seed ^ (0x9E3779B97F4A7C15ULL * (unsigned long long)(hop + 1)) ^ ((unsigned long long)v << 20));
    for (int n = 0; n < k; ++n) {
      const int j = deg - k + n;
      st = sg_mix(st + 0x9E3779B97F4A7C15ULL);
      const int t = (int)(((st >> 32) * (unsigned long long)(j + 1)) >> 32);        // uniform in [0, j]
      const bool hit = c_lo == t || c_hi == t;
      const int pick = __ballot(hit) ? j : t;
      if (n < 64) { if (lane == n) c_lo = pick; }
      else if (lane == n - 64) c_hi = pick;
    }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int n = lane + 64 * half, pos = half ? c_hi : c_lo;
      if (n < k) {
        const int s = in_src[c0 + pos], e = in_eid[c0 + pos];
        w.e_src[o + n] = s; w.e_dst[o + n] = v; w.e_eid[o + n] = e;
        w.keep[o + n] = (w.seedbit[e >> 5] >> (e & 31)) & 1u ? 0 : 1;
        const unsigned bit = 1u << (s & 31);
        const unsigned old = atomicOr(w.visited + (s >> 5), bit);
        if (!(old & bit)) atomicOr(w.next + (s >> 5), bit);
      }
    }
  }
}

// counts[0] += the draws of this hop (device-side running base of the staging area)
__global__ void k_sg_advance(long long* counts, const int* __restrict__ foff, const long long* __restrict__ nf_p) {
  counts[0] += foff[(int)*nf_p];
}
__global__ void k_sg_init_counts(long long* counts, int B) { counts[0] = B; counts[1] = 0; counts[2] = 0; counts[3] = 0; }

// kept edges to their final positions, endpoints as ranks among the visited nodes
__global__ void k_sg_emit(SgWs w, int nw, long long ld, long long* __restrict__ out_eid, long long* __restrict__ out_ei) {
  const long long n = w.counts[0];
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    if (!w.keep[i]) continue;
    const long long o = w.koff[i];
    const int s = w.e_src[i], d = w.e_dst[i];
    out_eid[o] = w.e_eid[i];
    out_ei[o] = w.wpre[s >> 5] + __popc(w.visited[s >> 5] & ((1u << (s & 31)) - 1u));
    out_ei[ld + o] = w.wpre[d >> 5] + __popc(w.visited[d >> 5] & ((1u << (d & 31)) - 1u));
  }
}
__global__ void k_sg_nodes(const unsigned* __restrict__ bits, int nw, const int* __restrict__ pre, long long* __restrict__ out) {
  const int w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= nw) return;
  unsigned b = bits[w];
  int o = pre[w];
  while (b) {
    const int i = __ffs((int)b) - 1;
    out[o++] = 32LL * w + i;
    b &= b - 1;
  }
}

// clears the bits of `seeds` in the seed-edge bitmap
__global__ void k_sg_clear_seeds(const long long* __restrict__ seeds, int B, long long E, unsigned* __restrict__ seedbit) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  const long long e = seeds[i];
  if (e >= 0 && e < E) atomicAnd(seedbit + (e >> 5), ~(1u << (e & 31)));
}

static size_t sg_al(size_t b) { return (b + 255) & ~(size_t)255; }
static size_t sg_carve(char* base, int32_t N, int64_t cap, SgWs* w) {
  const size_t nw = ((size_t)N + 31) / 32;
  size_t o = 0;
  auto take = [&](size_t bytes) { char* p = base ? base + o : nullptr; o += sg_al(bytes); return p; };
  unsigned* vis = (unsigned*)take(nw * 4);
  unsigned* fro = (unsigned*)take(nw * 4);
  unsigned* nxt = (unsigned*)take(nw * 4);
  int* wcnt = (int*)take((nw + 1) * 4);
  int* wpre = (int*)take((nw + 1) * 4);
  int* flist = (int*)take((size_t)N * 4);
  int* fcnt = (int*)take(((size_t)N + 1) * 4);
  int* foff = (int*)take(((size_t)N + 1) * 4);
  int* es = (int*)take((size_t)cap * 4);
  int* ed = (int*)take((size_t)cap * 4);
  int* ee = (int*)take((size_t)cap * 4);
  int* kp = (int*)take((size_t)cap * 4);
  int* ko = (int*)take(((size_t)cap + 1) * 4);
  int* tops = (int*)take((1024 * SG_TOPS_PT + 2) * 4);
  long long* counts = (long long*)take(4 * 8);
  if (w) {
    w->visited = vis; w->frontier = fro; w->next = nxt; w->wcnt = wcnt; w->wpre = wpre; w->flist = flist; w->fcnt = fcnt;
    w->foff = foff; w->e_src = es; w->e_dst = ed; w->e_eid = ee; w->keep = kp; w->koff = ko; w->tops = tops; w->counts = counts;
  }
  return o;
}

}  // namespace tg

using namespace tg;

extern "C" int64_t tg_gsampler_seedbit_bytes(int64_t E) { return (int64_t)sg_al((size_t)((E + 31) / 32) * 4); }
extern "C" int64_t tg_gsampler_workspace_bytes(int32_t N, int64_t cap) { return (int64_t)sg_carve(nullptr, N, cap, nullptr); }

// Draw.  seeds int64 [B] edge ids; esrc / edst int64 [E] = the graph's edge_index rows; colptr / in_src / in_eid = its CSC;
// fanout host int32 [hops] (< 0 = every in-edge, else <= 128); cap = staging capacity in edges (an upper bound on seed +
// drawn edges: min(tg_sampler_max_edges, E + B)); seedbit = tg_gsampler_seedbit_bytes(E) bytes that are ZERO on entry and
// zero again on return (the call clears the bits it set); workspace = tg_gsampler_workspace_bytes(N, cap) bytes.
// counts_out (device int64 [4]) = {kept edges, nodes, staged edges, error (0 = ok)} when the stream reaches the end.
extern "C" int tg_gsampler_draw(const int64_t* seeds, int64_t B, const int64_t* esrc, const int64_t* edst, int64_t E,
                                const int32_t* colptr, const int32_t* in_src, const int32_t* in_eid, int32_t N,
                                const int32_t* fanout, int32_t hops, uint64_t rng_seed, int64_t cap, void* seedbit,
                                void* workspace, int64_t* counts_out, void* stream) {
  TG_CHECK(seeds && esrc && edst && colptr && in_src && in_eid && fanout && seedbit && workspace && counts_out,
           "tg_gsampler_draw: null operand");
  TG_CHECK(B > 0 && B <= cap && N > 0 && E > 0 && E < 2147483647LL && hops >= 1 && hops <= 8 && cap <= 1024LL * SG_SCAN_ELEMS,
           "tg_gsampler_draw: bad sizes (B=%lld N=%d E=%lld cap=%lld hops=%d)", (long long)B, N, (long long)E, (long long)cap, hops);
  TG_CHECK((size_t)N <= 1024ull * SG_TOPS_PT * SG_SCAN_ELEMS, "tg_gsampler_draw: N above the scan limit (%d M nodes)", 8 * SG_TOPS_PT);
  for (int h = 0; h < hops; ++h)
    TG_CHECK(fanout[h] < 0 || fanout[h] <= SG_MAX_FAN, "tg_gsampler_draw: fan-out %d above %d", fanout[h], SG_MAX_FAN);
  hipStream_t st = (hipStream_t)stream;
  SgWs w;
  sg_carve((char*)workspace, N, cap, &w);
  w.seedbit = (unsigned*)seedbit;
  const int nw = (N + 31) / 32;
  zero_async(w.visited, (size_t)nw * 4, st);
  zero_async(w.frontier, (size_t)nw * 4, st);
  zero_async(w.next, (size_t)nw * 4, st);
  hipLaunchKernelGGL(k_sg_init_counts, dim3(1), dim3(1), 0, st, w.counts, (int)B);
  hipLaunchKernelGGL(k_sg_seeds, dim3(ceil_div(B, 256)), dim3(256), 0, st, (const long long*)seeds, (int)B,
                     (const long long*)esrc, (const long long*)edst, (long long)E, N, w);
  TG_LAUNCH_CHECK();
  for (int h = 0; h < hops; ++h) {
    // frontier bitmap -> ascending node list (counts[2] = its length)
    hipLaunchKernelGGL(k_sg_popc, dim3(ceil_div(nw, 256)), dim3(256), 0, st, (const unsigned*)w.frontier, nw, w.wcnt);
    sg_scan(w.wcnt, w.wpre, nw, nullptr, w.tops, w.counts + 2, st);
    hipLaunchKernelGGL(k_sg_expand, dim3(ceil_div(nw, 256)), dim3(256), 0, st, (const unsigned*)w.frontier, nw,
                       (const int*)w.wpre, w.flist);
    // draws per node and their offsets; the frontier can hold up to N nodes, so the count runs over N entries
    hipLaunchKernelGGL(k_sg_count, dim3(ceil_div(N, 256)), dim3(256), 0, st, (const int*)w.flist,
                       (const long long*)(w.counts + 2), colptr, fanout[h], w.fcnt, N);
    sg_scan(w.fcnt, w.foff, N, nullptr, w.tops, nullptr, st);
    zero_async(w.next, (size_t)nw * 4, st);
    hipLaunchKernelGGL(k_sg_draw, dim3(2048), dim3(256), 0, st, (const int*)w.flist, (const long long*)(w.counts + 2),
                       colptr, in_src, in_eid, fanout[h], h, (unsigned long long)rng_seed, (const int*)w.foff,
                       (const long long*)w.counts, (long long)cap, w);
    hipLaunchKernelGGL(k_sg_advance, dim3(1), dim3(1), 0, st, w.counts, (const int*)w.foff, (const long long*)(w.counts + 2));
    TG_LAUNCH_CHECK();
    unsigned* t = w.frontier; w.frontier = w.next; w.next = t;           // (host-side swap of the two carved buffers)
  }
  // compaction offsets of the kept edges over the whole staging area, node ranks of the visited bitmap
  sg_scan(w.keep, w.koff, (int)cap, w.counts, w.tops, w.counts + 1, st);
  hipLaunchKernelGGL(k_sg_popc, dim3(ceil_div(nw, 256)), dim3(256), 0, st, (const unsigned*)w.visited, nw, w.wcnt);
  sg_scan(w.wcnt, w.wpre, nw, nullptr, w.tops, w.counts + 2, st);
  TG_LAUNCH_CHECK();
  (void)hipMemcpyAsync(counts_out, w.counts + 1, 8, hipMemcpyDeviceToDevice, st);          // kept edges
  (void)hipMemcpyAsync(counts_out + 1, w.counts + 2, 8, hipMemcpyDeviceToDevice, st);      // nodes
  (void)hipMemcpyAsync(counts_out + 2, w.counts, 8, hipMemcpyDeviceToDevice, st);          // staged edges
  (void)hipMemcpyAsync(counts_out + 3, w.counts + 3, 8, hipMemcpyDeviceToDevice, st);      // error flag
  return 0;
}

// Emit the pending draw of this workspace: out_eid int64 [n_edges], out_edge_index int64 [2, ld] (LOCAL ids, ld >= n_edges),
// out_nodes int64 [n_nodes] (sorted global ids); clears the seed bits the draw set.  n_edges / n_nodes are the counts the
// draw reported (the caller read them to allocate).
extern "C" int tg_gsampler_emit(const int64_t* seeds, int64_t B, int32_t N, int64_t E, int64_t cap, int64_t ld, void* seedbit,
                                void* workspace, int64_t* out_eid, int64_t* out_edge_index, int64_t* out_nodes,
                                void* stream) {
  TG_CHECK(seeds && seedbit && workspace && out_eid && out_edge_index && out_nodes, "tg_gsampler_emit: null operand");
  hipStream_t st = (hipStream_t)stream;
  SgWs w;
  sg_carve((char*)workspace, N, cap, &w);
  w.seedbit = (unsigned*)seedbit;
  const int nw = (N + 31) / 32;
  hipLaunchKernelGGL(k_sg_emit, dim3(1024), dim3(256), 0, st, w, nw, (long long)ld, (long long*)out_eid,
                     (long long*)out_edge_index);
  hipLaunchKernelGGL(k_sg_nodes, dim3(ceil_div(nw, 256)), dim3(256), 0, st, (const unsigned*)w.visited, nw,
                     (const int*)w.wpre, (long long*)out_nodes);
  hipLaunchKernelGGL(k_sg_clear_seeds, dim3(ceil_div(B, 256)), dim3(256), 0, st, (const long long*)seeds, (int)B, (long long)E,
                     (unsigned*)seedbit);
  TG_LAUNCH_CHECK();
  return 0;
}

