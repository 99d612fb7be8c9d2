// PNA message passing on the sampled subgraph (torch_geometric 2.5.3 PNAConv as configured at
// src/nn/models/fused.py:200-207; restated in oracle/pna.py):
//   * gather-concat of node/edge rows (PNAConv.message input [x_i, x_j, e]; edge update fused.py:254;
//     fuse input fused.py:257; ClassifierHead input decoder.py:18-19),
//   * its backward as a deterministic segmented sum over the CSR (no atomics),
//   * the multi-aggregation mean/max/min/std per destination ("the SpMM"; HBM-bound:
//     E*(F*b+4) bytes read + N*4F*b bytes written per call) and its backward,
//   * the degree-scaler combine (identity / amplification / attenuation) folded AFTER the post GEMM so the
//     [N,12F] / [N,13F] tensors of the reference never exist.
#include "common.hpp"
#include "../../include/tabgnn_hip.h"

namespace tg {

struct GatherPart {
  const void* src;
  const int* idx;       // nullptr = identity
  long long stride;     // elements between source rows
  int width;            // elements copied
  int relu;
};

template <typename T, int VEC>
__global__ void k_gather_concat3(GatherPart p0, GatherPart p1, GatherPart p2, T* __restrict__ out, long long rows) {
  const int W = p0.width + p1.width + p2.width;
  const int vpr = W / VEC;
  long long total = rows * vpr;
  long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  long long stride = (long long)gridDim.x * blockDim.x;
  for (; i < total; i += stride) {
    long long r = i / vpr;
    int c = (int)(i % vpr) * VEC;
    const GatherPart& p = c < p0.width ? p0 : (c < p0.width + p1.width ? p1 : p2);
    int cc = c < p0.width ? c : (c < p0.width + p1.width ? c - p0.width : c - p0.width - p1.width);
    long long sr = p.idx ? (long long)p.idx[r] : r;
    float v[VEC];
    loadv<T, VEC>((const T*)p.src + sr * p.stride + cc, v);
    if (p.relu) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) v[j] = fmaxf(v[j], 0.f);
    }
    storev<T, VEC>(out + r * W + c, v);
  }
}

// dx[i, :] = sum_{q in segA(i)} g[rowA(q), offA:offA+F] + sum_{q in segB(i)} g[rowB(q), offB:offB+F]
// seedB > 0: ONE CSR over 2*seedB slots; slot s < seedB reads (row s, offA), else (row s-seedB, offB).
// Nodes whose segments together exceed HUB_THRESH rows are deferred to k_segment_sum2_hub (one block per hub),
// so a heavy-tailed degree distribution does not serialise on one lane group.
constexpr int HUB_THRESH = 256;
// hub_work layout of the segmented sum: [0] = number of hubs, [1 .. HUB_SPLIT_HUBS] = arrival tickets of the big hubs,
// [HUB_BIG_CNT] = number of big hubs, [HUB_BIG_IDS ..) their ids, [HUB_PART_OFF ..) = HUB_SPLIT_HUBS * HUB_SPLIT *
// HUB_FMAX fp32 partial rows, [HUB_IDS ..] = ids of the other hubs.
// Hubs of more than HUB_BIG rows (the first HUB_SPLIT_HUBS of them) are reduced by HUB_SPLIT workgroups each (a
// 12.8 k-row hub account was one 1024-thread block walking 25 dependent rounds: 73 us, three times per step); the
// block that arrives last sums the partial rows in part order, so the result does not depend on the arrival order.
constexpr int HUB_SPLIT = 8, HUB_SPLIT_HUBS = 32, HUB_FMAX = 512, HUB_BIG = 2048, HUB_BIG_CNT = 1 + HUB_SPLIT_HUBS,
              HUB_BIG_IDS = HUB_BIG_CNT + 1, HUB_PART_OFF = 128,
              HUB_IDS = HUB_PART_OFF + HUB_SPLIT_HUBS * HUB_SPLIT * HUB_FMAX;

template <typename T, int VEC>
__global__ void k_segment_sum2(const T* __restrict__ g, long long gstride, int offA, const int* __restrict__ rpA,
                               const int* __restrict__ pmA, int offB, const int* __restrict__ rpB,
                               const int* __restrict__ pmB, int seedB, const T* __restrict__ relu_src,
                               T* __restrict__ dx, int N, int F, int* __restrict__ hub /*[0]=count, [1..]=ids*/,
                               int accumulate) {
  const int lpn = F / VEC;  // lanes per node
  long long gid = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  long long stride = (long long)gridDim.x * blockDim.x;
  long long total = (long long)N * lpn;
  for (; gid < total; gid += stride) {
    int n = (int)(gid / lpn), c = (int)(gid % lpn) * VEC;
    int sA = rpA[n], eA = rpA[n + 1];
    int sB = rpB ? rpB[n] : 0, eB = rpB ? rpB[n + 1] : 0;
    if ((eA - sA) + (eB - sB) > HUB_THRESH) {
      if (c == 0) {
        bool big = (eA - sA) + (eB - sB) > HUB_BIG;
        if (big) {
          const int k = atomicAdd(hub + HUB_BIG_CNT, 1);
          if (k < HUB_SPLIT_HUBS) hub[HUB_BIG_IDS + k] = n;
          else big = false;                                  // more big hubs than split slots: an ordinary hub
        }
        if (!big) hub[HUB_IDS + atomicAdd(hub, 1)] = n;
      }
      continue;
    }
    if (accumulate && eA == sA && eB == sB) continue;      // dx += 0: the row is not touched at all
    float acc[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
    int q = sA;
    for (; q + 1 < eA; q += 2) {   // two rows in flight
      int r0 = pmA ? pmA[q] : q, r1 = pmA ? pmA[q + 1] : q + 1, o0 = offA, o1 = offA;
      if (seedB > 0) {
        if (r0 >= seedB) { r0 -= seedB; o0 = offB; }
        if (r1 >= seedB) { r1 -= seedB; o1 = offB; }
      }
      float t0[VEC], t1[VEC];
      loadv<T, VEC>(g + (long long)r0 * gstride + o0 + c, t0);
      loadv<T, VEC>(g + (long long)r1 * gstride + o1 + c, t1);
#pragma unroll
      for (int j = 0; j < VEC; ++j) acc[j] = (acc[j] + t0[j]) + t1[j];
    }
    if (q < eA) {
      int row = pmA ? pmA[q] : q, off = offA;
      if (seedB > 0 && row >= seedB) { row -= seedB; off = offB; }
      float t[VEC];
      loadv<T, VEC>(g + (long long)row * gstride + off + c, t);
#pragma unroll
      for (int j = 0; j < VEC; ++j) acc[j] += t[j];
    }
    q = sB;
    for (; q + 1 < eB; q += 2) {
      float t0[VEC], t1[VEC];
      loadv<T, VEC>(g + (long long)pmB[q] * gstride + offB + c, t0);
      loadv<T, VEC>(g + (long long)pmB[q + 1] * gstride + offB + c, t1);
#pragma unroll
      for (int j = 0; j < VEC; ++j) acc[j] = (acc[j] + t0[j]) + t1[j];
    }
    if (q < eB) {
      float t[VEC];
      loadv<T, VEC>(g + (long long)pmB[q] * gstride + offB + c, t);
#pragma unroll
      for (int j = 0; j < VEC; ++j) acc[j] += t[j];
    }
    if (relu_src) {
      float x[VEC];
      loadv<T, VEC>(relu_src + (long long)n * F + c, x);
#pragma unroll
      for (int j = 0; j < VEC; ++j) acc[j] = x[j] > 0.f ? acc[j] : 0.f;
    }
    if (accumulate) {       // dx already holds another consumer's gradient of the same tensor (ops.GradSink)
      float old[VEC];
      loadv<T, VEC>(dx + (long long)n * F + c, old);
#pragma unroll
      for (int j = 0; j < VEC; ++j) acc[j] += old[j];
    }
    storev<T, VEC>(dx + (long long)n * F + c, acc);
  }
}

// hub nodes: 1024-thread blocks, lane groups take strided rows (eight in flight each), partial sums meet in LDS in group
// order.  The big hubs are cut into HUB_SPLIT contiguous parts of their CSR ranges (one work item each); the ordinary
// hubs get a whole block each.
template <typename T, int VEC>
__global__ void __launch_bounds__(1024) k_segment_sum2_hub(const T* __restrict__ g, long long gstride, int offA,
                                                            const int* __restrict__ rpA, const int* __restrict__ pmA,
                                                            int offB, const int* __restrict__ rpB,
                                                            const int* __restrict__ pmB, int seedB,
                                                            const T* __restrict__ relu_src, T* __restrict__ dx, int F,
                                                            int* __restrict__ hub, float* __restrict__ partials,
                                                            int accumulate) {
  extern __shared__ float part[];  // [groups][F]
  __shared__ int ticket_s;
  const int lpn = F / VEC, groups = 1024 / lpn;
  const int gi = threadIdx.x / lpn, c = (threadIdx.x % lpn) * VEC;
  const int nh = hub[0];
  const int nsplit = hub[HUB_BIG_CNT] < HUB_SPLIT_HUBS ? hub[HUB_BIG_CNT] : HUB_SPLIT_HUBS;
  // work items: (big hub, part) pairs first, then one item per ordinary hub
  const int nitems = nsplit * HUB_SPLIT + nh;
  for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
    const bool split = item < nsplit * HUB_SPLIT;
    const int hIdx = split ? item / HUB_SPLIT : item - nsplit * HUB_SPLIT;
    const int pi = split ? item % HUB_SPLIT : 0, np = split ? HUB_SPLIT : 1;
    const int n = split ? hub[HUB_BIG_IDS + hIdx] : hub[HUB_IDS + hIdx];
    float acc[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
    if (gi < groups) {
      // this block's share of segment A and of segment B (contiguous slices of the CSR ranges)
      const int sA0 = rpA[n], eA0 = rpA[n + 1];
      const int lenA = eA0 - sA0, sA = sA0 + (int)((long long)lenA * pi / np), eA = sA0 + (int)((long long)lenA * (pi + 1) / np);
      for (int q = sA + gi; q < eA; q += groups * 8) {
        float t[8][VEC];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int qq = q + u * groups;
          int row = pmA ? pmA[qq < eA ? qq : q] : (qq < eA ? qq : q), off = offA;
          if (seedB > 0 && row >= seedB) { row -= seedB; off = offB; }
          loadv<T, VEC>(g + (long long)row * gstride + off + c, t[u]);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const bool ok = q + u * groups < eA;
#pragma unroll
          for (int j = 0; j < VEC; ++j) acc[j] += ok ? t[u][j] : 0.f;
        }
      }
      if (rpB) {
        const int sB0 = rpB[n], eB0 = rpB[n + 1];
        const int lenB = eB0 - sB0, sB = sB0 + (int)((long long)lenB * pi / np), eB = sB0 + (int)((long long)lenB * (pi + 1) / np);
        for (int q = sB + gi; q < eB; q += groups * 8) {
          float t[8][VEC];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int qq = q + u * groups;
            loadv<T, VEC>(g + (long long)pmB[qq < eB ? qq : q] * gstride + offB + c, t[u]);
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const bool ok = q + u * groups < eB;
#pragma unroll
            for (int j = 0; j < VEC; ++j) acc[j] += ok ? t[u][j] : 0.f;
          }
        }
      }
#pragma unroll
      for (int j = 0; j < VEC; ++j) part[gi * F + c + j] = acc[j];
    }
    __syncthreads();
    bool finish = true;
    float* prow = partials + ((long long)hIdx * HUB_SPLIT) * HUB_FMAX;      // this hub's HUB_SPLIT partial rows
    if (split) {
      for (int f = threadIdx.x; f < F; f += 1024) {
        float t = 0.f;
        for (int g2 = 0; g2 < groups; ++g2) t += part[g2 * F + f];
        prow[pi * HUB_FMAX + f] = t;
      }
      __threadfence();                                   // the partial row is visible before the ticket is taken
      __syncthreads();
      if (threadIdx.x == 0) ticket_s = atomicAdd(hub + 1 + hIdx, 1);
      __syncthreads();
      finish = ticket_s == HUB_SPLIT - 1;                // the last part to arrive combines them, in part order
      if (finish) __threadfence();
    }
    if (finish) {
      for (int f = threadIdx.x; f < F; f += 1024) {
        float t = 0.f;
        if (split) {
          for (int k = 0; k < HUB_SPLIT; ++k) t += __builtin_nontemporal_load(prow + k * HUB_FMAX + f);
        } else {
          for (int g2 = 0; g2 < groups; ++g2) t += part[g2 * F + f];
        }
        if (relu_src && !(to_f<T>(relu_src[(long long)n * F + f]) > 0.f)) t = 0.f;
        if (accumulate) t += to_f<T>(dx[(long long)n * F + f]);
        dx[(long long)n * F + f] = from_f<T>(t);
      }
    }
    __syncthreads();
  }
}

// dst[r, 0:W] (+)= src[idx ? idx[r] : r, 0:W]  (src row pitch `sstride` elements): one column block of a wider gradient
// (the edge-attribute third of d[x_i | x_j | e]), optionally row-gathered back to edge order, delivered into the
// shared gradient buffer of a tensor with several consumers (ops.GradSink) without an intermediate copy.
template <typename T, int VEC>
__global__ void k_rows_add(T* __restrict__ dst, const T* __restrict__ src, const int* __restrict__ idx, long long rows,
                           int W, long long sstride, int accumulate) {
  const int lpr = W / VEC;
  long long gid = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  const long long stride = (long long)gridDim.x * blockDim.x, total = rows * lpr;
  for (; gid < total; gid += stride) {
    const long long r = gid / lpr;
    const int c = (int)(gid % lpr) * VEC;
    const long long sr = idx ? idx[r] : r;
    float v[VEC];
    loadv<T, VEC>(src + sr * sstride + c, v);
    if (accumulate) {
      float o[VEC];
      loadv<T, VEC>(dst + r * W + c, o);
#pragma unroll
      for (int j = 0; j < VEC; ++j) v[j] += o[j];
    }
    storev<T, VEC>(dst + r * W + c, v);
  }
}

// ------------------------------------------------------------------ multi-aggregation forward
// One group of F/VEC lanes per destination node (16 lanes for bf16 F=128 -> 4 nodes per wave),
// 16-byte loads of whole message rows in CSR order, single pass for sum, sum of squares, max, min.
constexpr float STD_EPS = 1e-5f;
// Destinations with more rows than this (reverse message passing turns the heavy-tailed SOURCES into destinations: one
// with 12.7 k in-edges at B = 8192) are reduced by a whole workgroup instead of one lane group: 2.0 -> ~0.2 ms forward,
// 8.4 -> ~0.4 ms backward on the flipped bench graph.
constexpr int AGG_HUB = 512;

// SORTED: the messages are already in CSR (destination-sorted) order — row q of h IS position q of the CSR, so the
// perm indirection disappears and consecutive destinations read consecutive rows (a pure stream).
template <typename T, int VEC, bool SORTED>
__global__ void __launch_bounds__(256) k_pna_aggregate_fwd(const T* __restrict__ h, const int* __restrict__ rowptr,
                                                            const int* __restrict__ perm, T* __restrict__ agg, int N,
                                                            int F, int* __restrict__ hub) {
  const int lpn = F / VEC;
  long long gid = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  long long stride = (long long)gridDim.x * blockDim.x;
  long long total = (long long)N * lpn;
  if (gid >= total) return;
  int pn = (int)(gid / lpn);
  int ps = rowptr[pn], pe = rowptr[pn + 1];
  for (; gid < total; gid += stride) {
    int n = (int)(gid / lpn), c = (int)(gid % lpn) * VEC;
    int s = ps, e = pe;
    {  // software pipeline: the next item's segment bounds are in flight while this item's rows stream
      long long ng = gid + stride;
      int nn = (int)((ng < total ? ng : gid) / lpn);
      ps = rowptr[nn]; pe = rowptr[nn + 1];
    }
    if (hub && e - s > AGG_HUB) {          // deferred to k_pna_aggregate_fwd_hub (one workgroup per hub destination)
      if (c == 0) hub[1 + atomicAdd(hub, 1)] = n;
      continue;
    }
    float s1[VEC], s2[VEC], mx[VEC], mn[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) { s1[j] = 0.f; s2[j] = 0.f; mx[j] = -INFINITY; mn[j] = INFINITY; }
    int q = s;
    // long segments (the few high in-degree destinations set the kernel's tail): eight rows in flight, summed in
    // CSR order as always
    for (; q + 7 < e; q += 8) {
      int r[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) r[u] = SORTED ? q + u : perm[q + u];
      float v[8][VEC];
#pragma unroll
      for (int u = 0; u < 8; ++u) loadv<T, VEC>(h + (long long)r[u] * F + c, v[u]);
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          s1[j] += v[u][j]; s2[j] += v[u][j] * v[u][j]; mx[j] = fmaxf(mx[j], v[u][j]); mn[j] = fminf(mn[j], v[u][j]);
        }
    }
    for (; q + 1 < e; q += 2) {  // two rows in flight
      float a[VEC], b[VEC];
      int r0 = SORTED ? q : perm[q], r1 = SORTED ? q + 1 : perm[q + 1];
      loadv<T, VEC>(h + (long long)r0 * F + c, a);
      loadv<T, VEC>(h + (long long)r1 * F + c, b);
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        s1[j] += a[j]; s2[j] += a[j] * a[j]; mx[j] = fmaxf(mx[j], a[j]); mn[j] = fminf(mn[j], a[j]);
        s1[j] += b[j]; s2[j] += b[j] * b[j]; mx[j] = fmaxf(mx[j], b[j]); mn[j] = fminf(mn[j], b[j]);
      }
    }
    if (q < e) {
      float a[VEC];
      loadv<T, VEC>(h + (long long)(SORTED ? q : perm[q]) * F + c, a);
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        s1[j] += a[j]; s2[j] += a[j] * a[j]; mx[j] = fmaxf(mx[j], a[j]); mn[j] = fminf(mn[j], a[j]);
      }
    }
    float mean[VEC], sd[VEC];
    if (e > s) {
      float cnt = (float)(e - s);   // true division (sum / count, as the reference's scatter-mean): exact on constants
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        mean[j] = s1[j] / cnt;
        float var = s2[j] / cnt - mean[j] * mean[j];
        float t = sqrtf(fmaxf(var, STD_EPS));
        sd[j] = t <= sqrtf(STD_EPS) ? 0.f : t;
      }
    } else {
#pragma unroll
      for (int j = 0; j < VEC; ++j) { mean[j] = 0.f; mx[j] = 0.f; mn[j] = 0.f; sd[j] = 0.f; }
    }
    T* o = agg + (long long)n * 4 * F + c;
    storev<T, VEC>(o, mean);
    storev<T, VEC>(o + F, mx);
    storev<T, VEC>(o + 2 * F, mn);
    storev<T, VEC>(o + 3 * F, sd);
  }
}

// One 1024-thread workgroup per hub destination (4 channels per lane; groups of F/4 lanes take strided rows, four in
// flight; the partial (sum, sum of squares, max, min) meet in LDS in group order: deterministic).
template <typename T, bool SORTED>
__global__ void __launch_bounds__(1024) k_pna_aggregate_fwd_hub(const T* __restrict__ h, const int* __restrict__ rowptr,
                                                                 const int* __restrict__ perm, T* __restrict__ agg, int F,
                                                                 const int* __restrict__ hub) {
  constexpr int VEC = 4;
  extern __shared__ float part[];                     // [groups][4][F]
  const int lpn = F / VEC, groups = 1024 / lpn;
  const int gi = threadIdx.x / lpn, c = (threadIdx.x % lpn) * VEC;
  const int nh = hub[0];
  for (int hIdx = blockIdx.x; hIdx < nh; hIdx += gridDim.x) {
    const int n = hub[1 + hIdx];
    const int s = rowptr[n], e = rowptr[n + 1];
    if (gi < groups) {
      float s1[VEC], s2[VEC], mx[VEC], mn[VEC];
#pragma unroll
      for (int j = 0; j < VEC; ++j) { s1[j] = 0.f; s2[j] = 0.f; mx[j] = -INFINITY; mn[j] = INFINITY; }
      for (int q = s + gi; q < e; q += groups * 4) {
        float v[4][VEC];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int qq = q + u * groups < e ? q + u * groups : q;          // clamped: no load behind a branch
          loadv<T, VEC>(h + (long long)(SORTED ? qq : perm[qq]) * F + c, v[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const bool ok = q + u * groups < e;
#pragma unroll
          for (int j = 0; j < VEC; ++j) {
            const float x = v[u][j];
            s1[j] += ok ? x : 0.f; s2[j] += ok ? x * x : 0.f;
            mx[j] = ok ? fmaxf(mx[j], x) : mx[j]; mn[j] = ok ? fminf(mn[j], x) : mn[j];
          }
        }
      }
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        part[(gi * 4 + 0) * F + c + j] = s1[j]; part[(gi * 4 + 1) * F + c + j] = s2[j];
        part[(gi * 4 + 2) * F + c + j] = mx[j]; part[(gi * 4 + 3) * F + c + j] = mn[j];
      }
    }
    __syncthreads();
    for (int f = threadIdx.x; f < F; f += 1024) {
      float s1 = 0.f, s2 = 0.f, mx = -INFINITY, mn = INFINITY;
      for (int g2 = 0; g2 < groups; ++g2) {
        s1 += part[(g2 * 4 + 0) * F + f]; s2 += part[(g2 * 4 + 1) * F + f];
        mx = fmaxf(mx, part[(g2 * 4 + 2) * F + f]); mn = fminf(mn, part[(g2 * 4 + 3) * F + f]);
      }
      const float cnt = (float)(e - s);
      const float mean = s1 / cnt;
      const float t = sqrtf(fmaxf(s2 / cnt - mean * mean, STD_EPS));
      T* o = agg + (long long)n * 4 * F + f;
      o[0] = from_f<T>(mean); o[F] = from_f<T>(mx); o[2 * F] = from_f<T>(mn);
      o[3 * F] = from_f<T>(t <= sqrtf(STD_EPS) ? 0.f : t);
    }
    __syncthreads();
  }
}

template <typename T, bool SORTED>
__global__ void __launch_bounds__(1024) k_pna_aggregate_bwd_hub(const T* __restrict__ h, const T* __restrict__ agg,
                                                                 const T* __restrict__ dagg, const int* __restrict__ rowptr,
                                                                 const int* __restrict__ perm, T* __restrict__ dh, int F,
                                                                 const int* __restrict__ hub) {
  constexpr int VEC = 4;
  extern __shared__ float part[];                     // [groups][2][F] tie counts, then totals in part[0..2F)
  const int lpn = F / VEC, groups = 1024 / lpn;
  const int gi = threadIdx.x / lpn, c = (threadIdx.x % lpn) * VEC;
  const int nh = hub[0];
  for (int hIdx = blockIdx.x; hIdx < nh; hIdx += gridDim.x) {
    const int n = hub[1 + hIdx];
    const int s = rowptr[n], e = rowptr[n + 1];
    const T* a = agg + (long long)n * 4 * F + c;
    const T* d = dagg + (long long)n * 4 * F + c;
    float mean[VEC], mx[VEC], mn[VEC], sd[VEC], gm[VEC], gx[VEC], gn[VEC], gs[VEC];
    if (gi < groups) {
      loadv<T, VEC>(a, mean); loadv<T, VEC>(a + F, mx); loadv<T, VEC>(a + 2 * F, mn); loadv<T, VEC>(a + 3 * F, sd);
      loadv<T, VEC>(d, gm); loadv<T, VEC>(d + F, gx); loadv<T, VEC>(d + 2 * F, gn); loadv<T, VEC>(d + 3 * F, gs);
      float tx[VEC], tn[VEC];
#pragma unroll
      for (int j = 0; j < VEC; ++j) { tx[j] = 0.f; tn[j] = 0.f; }
      for (int q = s + gi; q < e; q += groups * 4) {
        float v[4][VEC];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int qq = q + u * groups < e ? q + u * groups : q;
          loadv<T, VEC>(h + (long long)(SORTED ? qq : perm[qq]) * F + c, v[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const bool ok = q + u * groups < e;
#pragma unroll
          for (int j = 0; j < VEC; ++j) { tx[j] += (ok && v[u][j] == mx[j]); tn[j] += (ok && v[u][j] == mn[j]); }
        }
      }
#pragma unroll
      for (int j = 0; j < VEC; ++j) { part[(gi * 2 + 0) * F + c + j] = tx[j]; part[(gi * 2 + 1) * F + c + j] = tn[j]; }
    }
    __syncthreads();
    float totx = 0.f, totn = 0.f;                     // thread f < F sums the group counts of channel f
    if ((int)threadIdx.x < F) {
      for (int g2 = 0; g2 < groups; ++g2) { totx += part[(g2 * 2 + 0) * F + threadIdx.x]; totn += part[(g2 * 2 + 1) * F + threadIdx.x]; }
    }
    __syncthreads();
    if ((int)threadIdx.x < F) { part[threadIdx.x] = totx; part[F + threadIdx.x] = totn; }
    __syncthreads();
    if (gi < groups) {
      const float inv = 1.f / (float)(e - s);
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        gm[j] *= inv;
        gx[j] = gx[j] / fmaxf(part[c + j], 1.f);
        gn[j] = gn[j] / fmaxf(part[F + c + j], 1.f);
        gs[j] = sd[j] > 0.f ? gs[j] * inv / sd[j] : 0.f;
      }
      for (int q = s + gi; q < e; q += groups) {
        const long long row = SORTED ? q : perm[q];
        float v[VEC], o[VEC];
        loadv<T, VEC>(h + row * F + c, v);
#pragma unroll
        for (int j = 0; j < VEC; ++j)
          o[j] = gm[j] + (v[j] == mx[j] ? gx[j] : 0.f) + (v[j] == mn[j] ? gn[j] : 0.f) + gs[j] * (v[j] - mean[j]);
        storev<T, VEC>(dh + row * F + c, o);
      }
    }
    __syncthreads();
  }
}

// backward: dh[e] = g_mean/cnt + [h==max] g_max/ties + [h==min] g_min/ties + g_std (h-mean)/(cnt*std)
// (ties share the gradient evenly, as torch.scatter_reduce amax/amin backward does)
// Destination-sorted messages, staged: a block owns NPB consecutive destinations, whose message rows are ONE contiguous
// range of h.  The block streams that range into LDS with fully coalesced, segment-independent 16-byte loads (the
// per-destination kernel above reads the same bytes behind a rowptr -> row dependency, at ~1/3 of the stream rate),
// then lane groups reduce their destination's rows out of LDS in CSR order (same summation order, same results).
// Blocks whose range does not fit the tile (hub destinations) read their rows straight from HBM.
// Measured at N=524k, F=128 inside the bench step (us): NPB/tile KB 64/32: 144, 64/16: 138, 32/16: 133, 32/20: 131,
// 16/8: 139, 8/8: 162; wave-private tiles (no block barrier) 8 per wave: 135; unstaged per-destination kernel: 158.
// A persistent variant (2048 workgroups walking tiles, next tile's rows prefetched into registers under the current
// tile's reduction) measured 180 us: on gfx9 stores and loads share vmcnt, so waiting for the prefetched rows also
// drains the stores issued after them and every tile pays a store round trip; one tile per workgroup keeps all
// loads ahead of all stores and lets the hardware overlap workgroups instead.  Assembling the block's [32 x 4F] output
// in LDS and writing it as one contiguous run of 16-byte-lane rows (48 KB of LDS, one more barrier) measured 200 us
// against 136 us: the 256-byte segments of the direct stores are not what holds this kernel back, occupancy is.
constexpr int AGG_NPB = 32;
template <typename T, int VEC, int NPB>
__global__ void __launch_bounds__(256) k_pna_aggregate_fwd_staged(const T* __restrict__ h, const int* __restrict__ rowptr,
                                                                   T* __restrict__ agg, int N, int F, int cap_rows,
                                                                   int* __restrict__ hub) {
  static_assert(NPB + 1 <= 256, "one thread per staged rowptr entry");
  extern __shared__ __align__(16) unsigned char agg_smem[];
  __shared__ int rp[NPB + 1];
  T* tile = reinterpret_cast<T*>(agg_smem);
  const int n0 = blockIdx.x * NPB;
  const int nn = min(NPB, N - n0);
  if ((int)threadIdx.x <= nn) rp[threadIdx.x] = rowptr[n0 + threadIdx.x];
  __syncthreads();
  const int r0 = rp[0], rows = rp[nn] - r0;
  const bool staged = rows <= cap_rows;
  if (staged && rows > 0) {
    const int pieces = rows * (F * (int)sizeof(T) / 16);
    const uint4* src = reinterpret_cast<const uint4*>(h + (long long)r0 * F);
    uint4* dst = reinterpret_cast<uint4*>(tile);
    for (int p = threadIdx.x; p < pieces; p += 4 * 256) {   // four independent 16-byte loads in flight per lane
      uint4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = src[min(p + u * 256, pieces - 1)];
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (p + u * 256 < pieces) dst[p + u * 256] = v[u];
    }
  }
  __syncthreads();
  const int lpn = F / VEC, groups = 256 / lpn;
  const int g = threadIdx.x / lpn, c = (threadIdx.x % lpn) * VEC;
  for (int i = g; i < nn; i += groups) {
    const int s = rp[i] - r0, e = rp[i + 1] - r0;
    if (hub && e - s > AGG_HUB) {
      if (c == 0) hub[1 + atomicAdd(hub, 1)] = n0 + i;
      continue;
    }
    float s1[VEC], s2[VEC], mx[VEC], mn[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) { s1[j] = 0.f; s2[j] = 0.f; mx[j] = -INFINITY; mn[j] = INFINITY; }
    if (staged) {
      for (int q = s; q < e; ++q) {
        float a[VEC];
        loadv<T, VEC>(tile + q * F + c, a);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          s1[j] += a[j]; s2[j] += a[j] * a[j]; mx[j] = fmaxf(mx[j], a[j]); mn[j] = fminf(mn[j], a[j]);
        }
      }
    } else {
      const T* base = h + (long long)r0 * F + c;
      int q = s;
      for (; q + 3 < e; q += 4) {
        float v[4][VEC];
#pragma unroll
        for (int u = 0; u < 4; ++u) loadv<T, VEC>(base + (long long)(q + u) * F, v[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int j = 0; j < VEC; ++j) {
            s1[j] += v[u][j]; s2[j] += v[u][j] * v[u][j]; mx[j] = fmaxf(mx[j], v[u][j]); mn[j] = fminf(mn[j], v[u][j]);
          }
      }
      for (; q < e; ++q) {
        float a[VEC];
        loadv<T, VEC>(base + (long long)q * F, a);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          s1[j] += a[j]; s2[j] += a[j] * a[j]; mx[j] = fmaxf(mx[j], a[j]); mn[j] = fminf(mn[j], a[j]);
        }
      }
    }
    float mean[VEC], sd[VEC];
    if (e > s) {
      float cnt = (float)(e - s);
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        mean[j] = s1[j] / cnt;
        float var = s2[j] / cnt - mean[j] * mean[j];
        float t = sqrtf(fmaxf(var, STD_EPS));
        sd[j] = t <= sqrtf(STD_EPS) ? 0.f : t;
      }
    } else {
#pragma unroll
      for (int j = 0; j < VEC; ++j) { mean[j] = 0.f; mx[j] = 0.f; mn[j] = 0.f; sd[j] = 0.f; }
    }
    T* o = agg + (long long)(n0 + i) * 4 * F + c;
    storev<T, VEC>(o, mean);
    storev<T, VEC>(o + F, mx);
    storev<T, VEC>(o + 2 * F, mn);
    storev<T, VEC>(o + 3 * F, sd);
  }
}

template <typename T, int VEC, bool SORTED>
__global__ void __launch_bounds__(256) k_pna_aggregate_bwd(const T* __restrict__ h, const T* __restrict__ agg,
                                                            const T* __restrict__ dagg, const int* __restrict__ rowptr,
                                                            const int* __restrict__ perm, T* __restrict__ dh, int N,
                                                            int F, int* __restrict__ hub) {
  const int lpn = F / VEC;
  long long gid = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  long long stride = (long long)gridDim.x * blockDim.x;
  long long total = (long long)N * lpn;
  for (; gid < total; gid += stride) {
    int n = (int)(gid / lpn), c = (int)(gid % lpn) * VEC;
    int s = rowptr[n], e = rowptr[n + 1];
    if (e == s) continue;
    if (hub && e - s > AGG_HUB) {
      if (c == 0) hub[1 + atomicAdd(hub, 1)] = n;
      continue;
    }
    const T* a = agg + (long long)n * 4 * F + c;
    const T* d = dagg + (long long)n * 4 * F + c;
    float mean[VEC], mx[VEC], mn[VEC], sd[VEC], gm[VEC], gx[VEC], gn[VEC], gs[VEC];
    loadv<T, VEC>(a, mean); loadv<T, VEC>(a + F, mx); loadv<T, VEC>(a + 2 * F, mn); loadv<T, VEC>(a + 3 * F, sd);
    loadv<T, VEC>(d, gm); loadv<T, VEC>(d + F, gx); loadv<T, VEC>(d + 2 * F, gn); loadv<T, VEC>(d + 3 * F, gs);
    float inv = 1.f / (float)(e - s);
    float tx[VEC], tn[VEC];
    if (e - s > 1) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) { tx[j] = 0.f; tn[j] = 0.f; }
      int q = s;
      for (; q + 3 < e; q += 4) {   // four rows in flight (high in-degree destinations set the tail)
        float v[4][VEC];
#pragma unroll
        for (int u = 0; u < 4; ++u) loadv<T, VEC>(h + (long long)(SORTED ? q + u : perm[q + u]) * F + c, v[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int j = 0; j < VEC; ++j) { tx[j] += (v[u][j] == mx[j]); tn[j] += (v[u][j] == mn[j]); }
      }
      for (; q < e; ++q) {
        float v[VEC];
        loadv<T, VEC>(h + (long long)(SORTED ? q : perm[q]) * F + c, v);
#pragma unroll
        for (int j = 0; j < VEC; ++j) { tx[j] += (v[j] == mx[j]); tn[j] += (v[j] == mn[j]); }
      }
    } else {
#pragma unroll
      for (int j = 0; j < VEC; ++j) { tx[j] = 1.f; tn[j] = 1.f; }
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      gm[j] *= inv;
      gx[j] = gx[j] / fmaxf(tx[j], 1.f);
      gn[j] = gn[j] / fmaxf(tn[j], 1.f);
      gs[j] = sd[j] > 0.f ? gs[j] * inv / sd[j] : 0.f;
    }
    int q = s;
    for (; q + 3 < e; q += 4) {
      int row[4];
      float v[4][VEC];
#pragma unroll
      for (int u = 0; u < 4; ++u) { row[u] = SORTED ? q + u : perm[q + u]; loadv<T, VEC>(h + (long long)row[u] * F + c, v[u]); }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float o[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j)
          o[j] = gm[j] + (v[u][j] == mx[j] ? gx[j] : 0.f) + (v[u][j] == mn[j] ? gn[j] : 0.f) + gs[j] * (v[u][j] - mean[j]);
        storev<T, VEC>(dh + (long long)row[u] * F + c, o);
      }
    }
    for (; q < e; ++q) {
      int row = SORTED ? q : perm[q];
      float v[VEC], o[VEC];
      loadv<T, VEC>(h + (long long)row * F + c, v);
#pragma unroll
      for (int j = 0; j < VEC; ++j)
        o[j] = gm[j] + (v[j] == mx[j] ? gx[j] : 0.f) + (v[j] == mn[j] ? gn[j] : 0.f) + gs[j] * (v[j] - mean[j]);
      storev<T, VEC>(dh + (long long)row * F + c, o);
    }
  }
}

// ------------------------------------------------------------------ degree-scaler combine
// out = xw + G[:,0:F] + amp*G[:,F:2F] + att*G[:,2F:3F],  amp = log(deg+1)/avg, att = avg/log(max(deg,1)+1)
__device__ __forceinline__ void scalers(const int* rowptr, int n, float avg_log, float& amp, float& att) {
  float deg = (float)(rowptr[n + 1] - rowptr[n]);
  amp = logf(deg + 1.f) / avg_log;
  att = avg_log / logf(fmaxf(deg, 1.f) + 1.f);
}

// (amp, att) of every node as fp32 pairs: the row scales of the fused post projection (tg_gemm_*_scaled_bf16)
__global__ void k_degree_scalers(const int* __restrict__ rowptr, const float* __restrict__ avg_log,
                                 float2* __restrict__ out, int N) {
  const float al = avg_log[0];
  for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < N; n += gridDim.x * blockDim.x) {
    float amp, att;
    scalers(rowptr, n, al, amp, att);
    out[n] = make_float2(amp, att);
  }
}

template <typename T, int VEC>
__global__ void k_scale_combine_fwd(const T* __restrict__ xw, const T* __restrict__ G, const int* __restrict__ rowptr,
                                    const float* __restrict__ avg_log, T* __restrict__ out, int N, int F) {
  const int vpr = F / VEC;
  long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  long long stride = (long long)gridDim.x * blockDim.x;
  long long total = (long long)N * vpr;
  float al = avg_log[0];
  for (; i < total; i += stride) {
    int n = (int)(i / vpr), c = (int)(i % vpr) * VEC;
    float amp, att;
    scalers(rowptr, n, al, amp, att);
    float a[VEC], g0[VEC], g1[VEC], g2[VEC];
    loadv<T, VEC>(xw + (long long)n * F + c, a);
    const T* g = G + (long long)n * 3 * F + c;
    loadv<T, VEC>(g, g0); loadv<T, VEC>(g + F, g1); loadv<T, VEC>(g + 2 * F, g2);
#pragma unroll
    for (int j = 0; j < VEC; ++j) a[j] = a[j] + g0[j] + amp * g1[j] + att * g2[j];
    storev<T, VEC>(out + (long long)n * F + c, a);
  }
}

template <typename T, int VEC>
__global__ void k_scale_combine_bwd(const T* __restrict__ gout, const int* __restrict__ rowptr,
                                    const float* __restrict__ avg_log, T* __restrict__ dG, int N, int F) {
  const int vpr = F / VEC;
  long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  long long stride = (long long)gridDim.x * blockDim.x;
  long long total = (long long)N * vpr;
  float al = avg_log[0];
  for (; i < total; i += stride) {
    int n = (int)(i / vpr), c = (int)(i % vpr) * VEC;
    float amp, att;
    scalers(rowptr, n, al, amp, att);
    float g[VEC], g1[VEC], g2[VEC];
    loadv<T, VEC>(gout + (long long)n * F + c, g);
#pragma unroll
    for (int j = 0; j < VEC; ++j) { g1[j] = amp * g[j]; g2[j] = att * g[j]; }
    T* o = dG + (long long)n * 3 * F + c;
    storev<T, VEC>(o, g); storev<T, VEC>(o + F, g1); storev<T, VEC>(o + 2 * F, g2);
  }
}

}  // namespace tg

using namespace tg;

#define DISPATCH_T(dt, ...)                       \
  if ((dt) == F32) {                              \
    using T = float;                              \
    constexpr int VEC = 4;                        \
    __VA_ARGS__                                   \
  } else {                                        \
    using T = bf16_t;                             \
    constexpr int VEC = 8;                        \
    __VA_ARGS__                                   \
  }

extern "C" int tg_gather_concat3(const void* a, const int32_t* ia, int64_t sa, int32_t wa, int32_t relu_a,
                                 const void* b, const int32_t* ib, int64_t sb, int32_t wb, int32_t relu_b,
                                 const void* c, const int32_t* ic, int64_t sc, int32_t wc, void* out, int64_t rows,
                                 int32_t dt, void* stream) {
  TG_CHECK(wa % 8 == 0 && wb % 8 == 0 && wc % 8 == 0 && sa % 8 == 0 && sb % 8 == 0 && sc % 8 == 0,
           "tg_gather_concat3: widths/strides must be multiples of 8 (%d,%d,%d)", wa, wb, wc);
  if (rows == 0) return 0;
  GatherPart p0{a, ia, sa, wa, relu_a}, p1{b, ib, sb, wb, relu_b}, p2{c, ic, sc, wc, 0};
  DISPATCH_T(dt, {
    long long total = rows * ((wa + wb + wc) / VEC);
    hipLaunchKernelGGL((k_gather_concat3<T, VEC>), dim3(grid_cap(ceil_div(total, 256))), dim3(256), 0,
                       (hipStream_t)stream, p0, p1, p2, (T*)out, (long long)rows);
  })
  TG_LAUNCH_CHECK();
  return 0;
}

// ints of hub_work for total_rows CSR rows (also sizes the work lists of the aggregation's hub pass, which uses its
// own [count | ids] prefix of the same buffer)
extern "C" int64_t tg_segment_hub_ints(int64_t total_rows) {
  return (int64_t)HUB_IDS + total_rows / HUB_THRESH + 4;
}

extern "C" int tg_segment_sum2(const void* g, int64_t gstride, int32_t offA, const int32_t* rpA, const int32_t* pmA,
                               int32_t offB, const int32_t* rpB, const int32_t* pmB, int32_t seedB,
                               const void* relu_src, void* dx, int32_t N, int32_t F, int32_t* hub_work,
                               int32_t accumulate, int32_t dt, void* stream) {
  TG_CHECK(F % 8 == 0 && offA % 8 == 0 && offB % 8 == 0 && gstride % 8 == 0, "tg_segment_sum2: misaligned F=%d", F);
  TG_CHECK(rpA && hub_work, "tg_segment_sum2: CSR A and hub workspace required");
  hipStream_t st = (hipStream_t)stream;
  TG_CHECK(F <= HUB_FMAX, "tg_segment_sum2: F = %d exceeds %d", F, HUB_FMAX);
  zero_async(hub_work, sizeof(int) * (HUB_BIG_CNT + 1), st);     // hub counts + the big hubs' tickets
  // CSR rows = the larger row pointer's total: the id list is sized for it by tg_segment_hub_ints(total rows)
  DISPATCH_T(dt, {
    TG_CHECK(1024 % (F / VEC) == 0, "tg_segment_sum2: F/VEC must divide 1024 (F=%d)", F);
    long long total = (long long)N * (F / VEC);
    hipLaunchKernelGGL((k_segment_sum2<T, VEC>), dim3(grid_cap(ceil_div(total, 256), 256 * 16)), dim3(256), 0, st,
                       (const T*)g, (long long)gstride, offA, rpA, pmA, offB, rpB, pmB, seedB, (const T*)relu_src,
                       (T*)dx, N, F, hub_work, accumulate);
    size_t shm = (size_t)(1024 / (F / VEC)) * F * sizeof(float);
    hipLaunchKernelGGL((k_segment_sum2_hub<T, VEC>), dim3(256), dim3(1024), shm, st, (const T*)g, (long long)gstride,
                       offA, rpA, pmA, offB, rpB, pmB, seedB, (const T*)relu_src, (T*)dx, F, hub_work,
                       reinterpret_cast<float*>(hub_work + HUB_PART_OFF), accumulate);
  })
  TG_LAUNCH_CHECK();
  return 0;
}

extern "C" int tg_rows_add(void* dst, const void* src, const int32_t* idx, int64_t rows, int32_t W, int64_t sstride,
                           int32_t accumulate, int32_t dt, void* stream) {
  if (rows == 0) return 0;
  TG_CHECK(dst && src && rows > 0 && W > 0 && W % 8 == 0 && sstride % 8 == 0, "tg_rows_add: bad shape (W=%d)", W);
  TG_CHECK((reinterpret_cast<uintptr_t>(dst) & 15) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0,
           "tg_rows_add: operands must be 16-byte aligned");
  DISPATCH_T(dt, {
    const long long total = rows * (W / VEC);
    hipLaunchKernelGGL((k_rows_add<T, VEC>), dim3(grid_cap(ceil_div(total, 256), 256 * 32)), dim3(256), 0,
                       (hipStream_t)stream, (T*)dst, (const T*)src, idx, (long long)rows, W, (long long)sstride, accumulate);
  })
  TG_LAUNCH_CHECK();
  return 0;
}

// hub_work: tg_segment_hub_ints(E) ints with hub_work[0] == 0 on entry, or NULL (then every destination is reduced by
// one lane group).  The main launches only LIST the hub destinations; tg_pna_aggregate_hubs reduces them.
static bool agg_hub_ok(int F) { return 1024 % (F / 4) == 0 && (size_t)(1024 / (F / 4)) * 4 * F * sizeof(float) <= 150 * 1024; }

template <typename T> static void launch_agg_hub(bool fwd, const void* h, const void* agg, const void* dagg,
                                                 const int32_t* rowptr, const int32_t* perm, void* out, int F,
                                                 const int32_t* hub, hipStream_t st) {
  const int groups = 1024 / (F / 4);
  const size_t shm = (size_t)groups * (fwd ? 4 : 2) * F * sizeof(float);
  if (fwd) {
    if (perm) {
      (void)hipFuncSetAttribute((const void*)k_pna_aggregate_fwd_hub<T, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
      hipLaunchKernelGGL((k_pna_aggregate_fwd_hub<T, false>), dim3(64), dim3(1024), shm, st, (const T*)h, rowptr, perm, (T*)out, F, hub);
    } else {
      (void)hipFuncSetAttribute((const void*)k_pna_aggregate_fwd_hub<T, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
      hipLaunchKernelGGL((k_pna_aggregate_fwd_hub<T, true>), dim3(64), dim3(1024), shm, st, (const T*)h, rowptr, perm, (T*)out, F, hub);
    }
  } else {
    if (perm) {
      (void)hipFuncSetAttribute((const void*)k_pna_aggregate_bwd_hub<T, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
      hipLaunchKernelGGL((k_pna_aggregate_bwd_hub<T, false>), dim3(64), dim3(1024), shm, st, (const T*)h, (const T*)agg, (const T*)dagg, rowptr, perm, (T*)out, F, hub);
    } else {
      (void)hipFuncSetAttribute((const void*)k_pna_aggregate_bwd_hub<T, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
      hipLaunchKernelGGL((k_pna_aggregate_bwd_hub<T, true>), dim3(64), dim3(1024), shm, st, (const T*)h, (const T*)agg, (const T*)dagg, rowptr, perm, (T*)out, F, hub);
    }
  }
}

extern "C" int tg_pna_aggregate_fwd(const void* h, const int32_t* rowptr, const int32_t* perm, void* agg, int32_t N,
                                    int32_t F, int64_t E, int32_t* hub_work, int32_t dt, void* stream) {
  TG_CHECK(F % 8 == 0 && N > 0, "tg_pna_aggregate_fwd: F must be a multiple of 8 (F=%d)", F);
  hipStream_t st = (hipStream_t)stream;
  if (hub_work && !agg_hub_ok(F)) hub_work = nullptr;
  if (dt == BF16 && !perm && 256 % (F / 4) == 0) {
    // destination-sorted bf16 messages: rows staged through LDS by blocks of 32 destinations, 8-byte lanes for the
    // reduction and the stores (2x the lanes per destination of 16-byte lanes: measured fastest on MI355X)
    const int tile_bytes = 16 * 1024;            // 8 blocks (all 32 waves) resident per CU; 64 rows at F=128
    const int cap_rows = tile_bytes / (F * 2);
    hipLaunchKernelGGL((k_pna_aggregate_fwd_staged<bf16_t, 4, AGG_NPB>), dim3(ceil_div((long long)N, AGG_NPB)),
                       dim3(256), tile_bytes, st, (const bf16_t*)h, rowptr, (bf16_t*)agg, N, F, cap_rows, hub_work);
    TG_LAUNCH_CHECK();
    return 0;
  }
  DISPATCH_T(dt, {
    long long total = (long long)N * (F / VEC);
    if (perm)
      hipLaunchKernelGGL((k_pna_aggregate_fwd<T, VEC, false>), dim3(grid_cap(ceil_div(total, 256), 256 * 32)),
                         dim3(256), 0, st, (const T*)h, rowptr, perm, (T*)agg, N, F, hub_work);
    else
      hipLaunchKernelGGL((k_pna_aggregate_fwd<T, VEC, true>), dim3(grid_cap(ceil_div(total, 256), 256 * 32)),
                         dim3(256), 0, st, (const T*)h, rowptr, perm, (T*)agg, N, F, hub_work);
  })
  TG_LAUNCH_CHECK();
  return 0;
}

extern "C" int tg_pna_aggregate_bwd(const void* h, const void* agg, const void* dagg, const int32_t* rowptr,
                                    const int32_t* perm, void* dh, int32_t N, int32_t F, int32_t* hub_work, int32_t dt,
                                    void* stream) {
  TG_CHECK(F % 8 == 0 && N > 0, "tg_pna_aggregate_bwd: F must be a multiple of 8 (F=%d)", F);
  hipStream_t st = (hipStream_t)stream;
  if (hub_work && !agg_hub_ok(F)) hub_work = nullptr;
  DISPATCH_T(dt, {
    long long total = (long long)N * (F / VEC);
    if (perm)
      hipLaunchKernelGGL((k_pna_aggregate_bwd<T, VEC, false>), dim3(grid_cap(ceil_div(total, 256), 256 * 32)),
                         dim3(256), 0, st, (const T*)h, (const T*)agg, (const T*)dagg, rowptr, perm, (T*)dh, N, F,
                         hub_work);
    else
      hipLaunchKernelGGL((k_pna_aggregate_bwd<T, VEC, true>), dim3(grid_cap(ceil_div(total, 256), 256 * 32)),
                         dim3(256), 0, st, (const T*)h, (const T*)agg, (const T*)dagg, rowptr, perm, (T*)dh, N, F,
                         hub_work);
  })
  TG_LAUNCH_CHECK();
  return 0;
}

// second pass over the destinations the main launch listed in hub_work (forward: dagg == dh == NULL, writes agg;
// backward: writes dh)
extern "C" int tg_pna_aggregate_hubs(const void* h, const void* agg, const void* dagg, const int32_t* rowptr,
                                     const int32_t* perm, void* dh, int32_t F, const int32_t* hub_work, int32_t dt,
                                     void* stream) {
  TG_CHECK(F % 8 == 0 && hub_work, "tg_pna_aggregate_hubs: bad arguments (F=%d)", F);
  if (!agg_hub_ok(F)) return 0;        // the main launch then handled every destination itself
  const bool fwd = dh == nullptr;
  if (dt == F32) launch_agg_hub<float>(fwd, h, agg, dagg, rowptr, perm, fwd ? const_cast<void*>(agg) : dh, F, hub_work, (hipStream_t)stream);
  else launch_agg_hub<bf16_t>(fwd, h, agg, dagg, rowptr, perm, fwd ? const_cast<void*>(agg) : dh, F, hub_work, (hipStream_t)stream);
  TG_LAUNCH_CHECK();
  return 0;
}

extern "C" int tg_pna_scale_combine_fwd(const void* xw, const void* G, const int32_t* rowptr, const float* avg_log,
                                        void* out, int32_t N, int32_t F, int32_t dt, void* stream) {
  TG_CHECK(F % 8 == 0, "tg_pna_scale_combine_fwd: F must be a multiple of 8 (F=%d)", F);
  DISPATCH_T(dt, {
    long long total = (long long)N * (F / VEC);
    hipLaunchKernelGGL((k_scale_combine_fwd<T, VEC>), dim3(grid_cap(ceil_div(total, 256))), dim3(256), 0,
                       (hipStream_t)stream, (const T*)xw, (const T*)G, rowptr, avg_log, (T*)out, N, F);
  })
  TG_LAUNCH_CHECK();
  return 0;
}

extern "C" int tg_pna_scale_combine_bwd(const void* gout, const int32_t* rowptr, const float* avg_log, void* dG,
                                        int32_t N, int32_t F, int32_t dt, void* stream) {
  TG_CHECK(F % 8 == 0, "tg_pna_scale_combine_bwd: F must be a multiple of 8 (F=%d)", F);
  DISPATCH_T(dt, {
    long long total = (long long)N * (F / VEC);
    hipLaunchKernelGGL((k_scale_combine_bwd<T, VEC>), dim3(grid_cap(ceil_div(total, 256))), dim3(256), 0,
                       (hipStream_t)stream, (const T*)gout, rowptr, avg_log, (T*)dG, N, F);
  })
  TG_LAUNCH_CHECK();
  return 0;
}

extern "C" int tg_pna_degree_scalers(const int32_t* rowptr, const float* avg_log, float* out, int32_t N, void* stream) {
  TG_CHECK(rowptr && avg_log && out && (reinterpret_cast<uintptr_t>(out) & 7) == 0, "tg_pna_degree_scalers: bad argument");
  if (N == 0) return 0;
  hipLaunchKernelGGL(k_degree_scalers, dim3(grid_cap(ceil_div(N, 256))), dim3(256), 0, (hipStream_t)stream, rowptr,
                     avg_log, (float2*)out, N);
  TG_LAUNCH_CHECK();
  return 0;
}
