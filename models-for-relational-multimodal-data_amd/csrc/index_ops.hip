// Index-structure kernels: int64 -> int32 ids, stable CSR (counting sort) of a sampled
// subgraph's edge list by destination / source, and the seed-node pooling of the fused layer.
//
// The reference never builds a CSR: PyG's PNAConv scatters with atomics
// (torch.scatter_reduce).  Here one stable counting sort per forward call
// (reference call site: src/nn/models/fused.py:252-254 uses edge_index in both directions)
// turns every later scatter into a deterministic segmented reduction.
#include "common.hpp"
#include "../../include/tabgnn_hip.h"

namespace tg {

// ------------------------------------------------------------------ ids
__global__ void k_ids_to_i32(const long long* __restrict__ in, int* __restrict__ out, long long M, int N,
                             int* __restrict__ err) {
  long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  long long stride = (long long)gridDim.x * blockDim.x;
  for (; i < M; i += stride) {
    long long v = in[i];
    if (v < 0 || v >= N) {  // never index out of bounds on the device: clamp and flag
      atomicOr(err, 1);
      v = v < 0 ? 0 : N - 1;
    }
    out[i] = (int)v;
  }
}

// ------------------------------------------------------------------ histogram / scan / fill / rank
// Hot keys (a hub account with 12.8 k of the 438 k edges) serialise global atomics on one address: 147 us here and
// again in k_fill.  Each block therefore counts its CSR_CH consecutive keys in a small LDS table first (slot =
// hash(key); the first key to arrive owns the slot, other keys hashing there go straight to global memory) and
// flushes one global atomic per occupied slot: the hub costs one global atomic per block instead of ~60.
constexpr int CSR_KPT = 8, CSR_CH = 256 * CSR_KPT, CSR_HS = 512;
__device__ __forceinline__ int csr_slot(int k) { return (int)(((unsigned)k * 2654435761u) >> 23) & (CSR_HS - 1); }
__device__ __forceinline__ bool csr_own(int* tag, int slot, int k) {
  int t = tag[slot];
  if (t == -1) {
    t = atomicCAS(&tag[slot], -1, k);
    if (t == -1) t = k;
  }
  return t == k;
}

__global__ void __launch_bounds__(256) k_hist(const int* __restrict__ key, long long M, int* __restrict__ rowptr) {
  __shared__ int tag[CSR_HS], cnt[CSR_HS];
  for (int i = threadIdx.x; i < CSR_HS; i += 256) { tag[i] = -1; cnt[i] = 0; }
  __syncthreads();
  const long long base = (long long)blockIdx.x * CSR_CH;
#pragma unroll
  for (int j = 0; j < CSR_KPT; ++j) {
    const long long i = base + j * 256 + threadIdx.x;
    if (i < M) {
      const int k = key[i], slot = csr_slot(k);
      if (csr_own(tag, slot, k)) atomicAdd(&cnt[slot], 1);
      else atomicAdd(&rowptr[k + 1], 1);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < CSR_HS; i += 256)
    if (cnt[i]) atomicAdd(&rowptr[tag[i] + 1], cnt[i]);
}

constexpr int SCAN_T = 256, SCAN_E = 8, SCAN_CHUNK = SCAN_T * SCAN_E;

// inclusive scan of one 2048-element chunk per block; writes chunk total
__global__ void __launch_bounds__(SCAN_T) k_scan_chunks(int* __restrict__ a, long long n, int* __restrict__ totals) {
  __shared__ int wsum[SCAN_T / 64];
  long long base = (long long)blockIdx.x * SCAN_CHUNK + (long long)threadIdx.x * SCAN_E;
  int v[SCAN_E];
  int run = 0;
#pragma unroll
  for (int j = 0; j < SCAN_E; ++j) {
    long long idx = base + j;
    run += (idx < n) ? a[idx] : 0;
    v[j] = run;
  }
  int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  int incl = run;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int t = __shfl_up(incl, o, 64);
    if (lane >= o) incl += t;
  }
  if (lane == 63) wsum[wid] = incl;
  __syncthreads();
  int woff = 0;
  for (int w = 0; w < wid; ++w) woff += wsum[w];
  int excl = incl - run + woff;
#pragma unroll
  for (int j = 0; j < SCAN_E; ++j) {
    long long idx = base + j;
    if (idx < n) a[idx] = v[j] + excl;
  }
  if (threadIdx.x == SCAN_T - 1) totals[blockIdx.x] = excl + run;
}

// single block: exclusive scan of the chunk totals (in place)
__global__ void __launch_bounds__(1024) k_scan_totals(int* __restrict__ totals, int nchunks) {
  __shared__ int wsum[16];
  __shared__ int carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  for (int base = 0; base < nchunks; base += 1024) {
    int i = base + threadIdx.x;
    int x = i < nchunks ? totals[i] : 0;
    int incl = x;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      int t = __shfl_up(incl, o, 64);
      if (lane >= o) incl += t;
    }
    if (lane == 63) wsum[wid] = incl;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wid; ++w) woff += wsum[w];
    int c = carry;
    if (i < nchunks) totals[i] = c + woff + incl - x;
    __syncthreads();
    if (threadIdx.x == 1023) carry = c + woff + incl;
    __syncthreads();
  }
}

__global__ void __launch_bounds__(SCAN_T) k_scan_add(int* __restrict__ a, long long n, const int* __restrict__ totals,
                                                     int* __restrict__ cursor) {
  long long base = (long long)blockIdx.x * SCAN_CHUNK + (long long)threadIdx.x * SCAN_E;
  int off = totals[blockIdx.x];
#pragma unroll
  for (int j = 0; j < SCAN_E; ++j) {
    long long idx = base + j;
    if (idx < n) {
      int v = a[idx] + off;
      a[idx] = v;
      if (idx < n - 1) cursor[idx] = v;  // cursor[k] = rowptr[k]
    }
  }
}

// slot of every element inside its key's segment (any order: k_rank makes it stable afterwards); hot keys take their
// positions from the block's LDS table and ONE global cursor bump per (block, key), see k_hist
__global__ void __launch_bounds__(256) k_fill(const int* __restrict__ key, long long M, int* __restrict__ cursor,
                                              int* __restrict__ tmp) {
  __shared__ int tag[CSR_HS], cnt[CSR_HS];
  for (int i = threadIdx.x; i < CSR_HS; i += 256) { tag[i] = -1; cnt[i] = 0; }
  __syncthreads();
  const long long base = (long long)blockIdx.x * CSR_CH;
  int local[CSR_KPT];                                 // >= 0: rank inside the block's share of the key; -1: done / none
#pragma unroll
  for (int j = 0; j < CSR_KPT; ++j) {
    const long long i = base + j * 256 + threadIdx.x;
    local[j] = -1;
    if (i < M) {
      const int k = key[i], slot = csr_slot(k);
      if (csr_own(tag, slot, k)) local[j] = atomicAdd(&cnt[slot], 1);
      else tmp[atomicAdd(&cursor[k], 1)] = (int)i;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < CSR_HS; i += 256)
    if (cnt[i]) cnt[i] = atomicAdd(&cursor[tag[i]], cnt[i]);      // cnt := the block's base position for that key
  __syncthreads();
#pragma unroll
  for (int j = 0; j < CSR_KPT; ++j) {
    const long long i = base + j * 256 + threadIdx.x;
    if (local[j] >= 0) tmp[cnt[csr_slot(key[i])] + local[j]] = (int)i;
  }
}

// restore input order inside every segment (stable sort): rank by counting smaller ids.
// Segments longer than RANK_HUB are queued for k_rank_hub (O(len^2) here would serialise on a hub node).
constexpr int RANK_HUB = 512;
constexpr int RANK_WIN_WORDS = 8192;                  // 64-bit words per window = 512Ki ids (64 KiB bitmap + 32 KiB prefix)
constexpr int RANK_WIN_IDS = RANK_WIN_WORDS * 64;

__global__ void k_rank(const int* __restrict__ key, const int* __restrict__ rowptr, const int* __restrict__ tmp,
                       int* __restrict__ perm, long long M, int* __restrict__ hub /*[0]=count, [1..]=keys*/) {
  long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  long long stride = (long long)gridDim.x * blockDim.x;
  for (; p < M; p += stride) {
    int me = tmp[p];
    int k = key[me];
    int s = rowptr[k], e = rowptr[k + 1];
    if (e - s > RANK_HUB) {
      if (p == s) hub[1 + atomicAdd(hub, 1)] = k;
      continue;
    }
    int rank = 0;
    if (e - s > 1)
      for (int q = s; q < e; ++q) rank += tmp[q] < me;
    perm[s + rank] = me;
  }
}

// one 1024-thread block per hub segment: a bitmap of the member ids (ids are unique) lives in LDS, an exclusive
// prefix of its popcounts gives every member its rank = number of smaller members.  O(len + M/64) per hub.
__global__ void __launch_bounds__(1024) k_rank_hub(const int* __restrict__ rowptr, const int* __restrict__ tmp,
                                                    int* __restrict__ perm, long long M, const int* __restrict__ hub) {
  __shared__ unsigned long long bits[RANK_WIN_WORDS];
  __shared__ unsigned pre[RANK_WIN_WORDS];
  __shared__ unsigned wsum[16];
  __shared__ unsigned carry;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int nh = hub[0];
  for (int hIdx = blockIdx.x; hIdx < nh; hIdx += gridDim.x) {
    const int k = hub[1 + hIdx];
    const int s = rowptr[k], e = rowptr[k + 1];
    unsigned base = 0;
    for (long long w0 = 0; w0 < M; w0 += RANK_WIN_IDS) {
      for (int i = tid; i < RANK_WIN_WORDS; i += 1024) bits[i] = 0ull;
      if (tid == 0) carry = 0;
      __syncthreads();
      for (int q = s + tid; q < e; q += 1024) {
        long long id = (long long)tmp[q] - w0;
        if (id >= 0 && id < RANK_WIN_IDS) atomicOr(&bits[id >> 6], 1ull << (id & 63));
      }
      __syncthreads();
      // exclusive prefix of popcounts: 8 consecutive words per thread, wave scan, wave totals through LDS
      unsigned loc[RANK_WIN_WORDS / 1024], run = 0;
#pragma unroll
      for (int j = 0; j < RANK_WIN_WORDS / 1024; ++j) {
        loc[j] = run;
        run += __popcll(bits[tid * (RANK_WIN_WORDS / 1024) + j]);
      }
      unsigned incl = run;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        unsigned t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
      }
      if (lane == 63) wsum[wid] = incl;
      __syncthreads();
      unsigned woff = 0;
      for (int w = 0; w < wid; ++w) woff += wsum[w];
      unsigned excl = incl - run + woff;
#pragma unroll
      for (int j = 0; j < RANK_WIN_WORDS / 1024; ++j) pre[tid * (RANK_WIN_WORDS / 1024) + j] = excl + loc[j];
      if (tid == 1023) carry = excl + run;            // members inside this window
      __syncthreads();
      for (int q = s + tid; q < e; q += 1024) {
        int me = tmp[q];
        long long id = (long long)me - w0;
        if (id >= 0 && id < RANK_WIN_IDS) {
          unsigned r = base + pre[id >> 6] + __popcll(bits[id >> 6] & ((1ull << (id & 63)) - 1ull));
          perm[s + r] = me;
        }
      }
      base += carry;
      __syncthreads();
    }
  }
}

// ------------------------------------------------------------------ seed-node pooling (fused.py:261-268)
// x_out[i] = x[i]                       when node i is no seed endpoint
//          = (x[i] + mean_slots emb)/2  otherwise; slot s<B -> xf[s, C:C+F], slot s>=B -> xf[s-B, C+F:C+2F]
template <typename T, int VEC>
__global__ void k_seed_pool_fwd(const T* __restrict__ x, const T* __restrict__ xf, const int* __restrict__ rowptr,
                                const int* __restrict__ perm, T* __restrict__ out, int N, int F, int B, int C, int D) {
  int vpr = F / VEC;  // vectors per row
  long long total = (long long)N * vpr;
  long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  long long stride = (long long)gridDim.x * blockDim.x;
  for (; i < total; i += stride) {
    int n = (int)(i / vpr), c = (int)(i % vpr) * VEC;
    float v[VEC];
    loadv<T, VEC>(x + (long long)n * F + c, v);
    int s = rowptr[n], e = rowptr[n + 1];
    if (e > s) {
      float acc[VEC];
#pragma unroll
      for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
      for (int q = s; q < e; ++q) {
        int slot = perm[q];
        int b = slot < B ? slot : slot - B;
        int off = slot < B ? C : C + F;
        float t[VEC];
        loadv<T, VEC>(xf + (long long)b * D + off + c, t);
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[j] += t[j];
      }
      float inv = 1.f / (float)(e - s);
#pragma unroll
      for (int j = 0; j < VEC; ++j) v[j] = (v[j] + acc[j] * inv) * 0.5f;
    }
    storev<T, VEC>(out + (long long)n * F + c, v);
  }
}

// In-place form (the reference updates x_gnn in place too, fused.py:268): only the seed endpoints' rows are touched.
// One lane group per CSR position q of the 2B endpoint slots; the group at the FIRST position of a node's segment
// owns that node (same summation order as k_seed_pool_fwd, so the two agree bit for bit).
template <typename T, int VEC>
__global__ void k_seed_pool_inplace(T* __restrict__ x, const T* __restrict__ xf, const int* __restrict__ tei,
                                    const int* __restrict__ rowptr, const int* __restrict__ perm, int F, int B, int C,
                                    int D) {
  int vpr = F / VEC;
  long long total = (long long)2 * B * vpr;
  long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  long long stride = (long long)gridDim.x * blockDim.x;
  for (; i < total; i += stride) {
    int q = (int)(i / vpr), c = (int)(i % vpr) * VEC;
    int n = tei[perm[q]];
    int s = rowptr[n], e = rowptr[n + 1];
    if (q != s) continue;
    float acc[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
    int qq = s;
    for (; qq + 3 < e; qq += 4) {     // a hub endpoint is shared by hundreds of seed edges: four rows in flight
      float t[4][VEC];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        int slot = perm[qq + u];
        int b = slot < B ? slot : slot - B;
        int off = slot < B ? C : C + F;
        loadv<T, VEC>(xf + (long long)b * D + off + c, t[u]);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[j] += t[u][j];
    }
    for (; qq < e; ++qq) {
      int slot = perm[qq];
      int b = slot < B ? slot : slot - B;
      int off = slot < B ? C : C + F;
      float t[VEC];
      loadv<T, VEC>(xf + (long long)b * D + off + c, t);
#pragma unroll
      for (int j = 0; j < VEC; ++j) acc[j] += t[j];
    }
    float v[VEC];
    loadv<T, VEC>(x + (long long)n * F + c, v);
    float inv = 1.f / (float)(e - s);
#pragma unroll
    for (int j = 0; j < VEC; ++j) v[j] = (v[j] + acc[j] * inv) * 0.5f;
    storev<T, VEC>(x + (long long)n * F + c, v);
  }
}

// dx[i] = g[i] * (seed ? .5 : 1)
template <typename T, int VEC>
__global__ void k_seed_pool_bwd_x(const T* __restrict__ g, const int* __restrict__ rowptr, T* __restrict__ dx, int N,
                                  int F) {
  int vpr = F / VEC;
  long long total = (long long)N * vpr;
  long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  long long stride = (long long)gridDim.x * blockDim.x;
  for (; i < total; i += stride) {
    int n = (int)(i / vpr), c = (int)(i % vpr) * VEC;
    float v[VEC];
    loadv<T, VEC>(g + (long long)n * F + c, v);
    float sc = rowptr[n + 1] > rowptr[n] ? 0.5f : 1.f;
#pragma unroll
    for (int j = 0; j < VEC; ++j) v[j] *= sc;
    storev<T, VEC>(dx + (long long)n * F + c, v);
  }
}

// dxf[b, 0:C] = 0; dxf[b, C + part*F + f] = g[node(part,b), f] * 0.5 / cnt(node)
template <typename T, int VEC>
__global__ void k_seed_pool_bwd_f(const T* __restrict__ g, const int* __restrict__ tei /*[2B]*/,
                                  const int* __restrict__ rowptr, T* __restrict__ dxf, int B, int F, int C, int D) {
  int vpr = D / VEC;
  long long total = (long long)B * vpr;
  long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  long long stride = (long long)gridDim.x * blockDim.x;
  for (; i < total; i += stride) {
    int b = (int)(i / vpr), c = (int)(i % vpr) * VEC;
    float v[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) v[j] = 0.f;
    if (c >= C) {
      int part = (c - C) / F, f = (c - C) % F;
      int node = tei[part * B + b];
      float sc = 0.5f / (float)(rowptr[node + 1] - rowptr[node]);
      loadv<T, VEC>(g + (long long)node * F + f, v);
#pragma unroll
      for (int j = 0; j < VEC; ++j) v[j] *= sc;
    }
    storev<T, VEC>(dxf + (long long)b * D + c, v);
  }
}

}  // namespace tg

using namespace tg;

extern "C" int tg_ids_to_i32(const int64_t* ids, int64_t M, int32_t N, int32_t* out, int32_t* err_flag, void* stream) {
  TG_CHECK(M >= 0 && N > 0, "tg_ids_to_i32: bad sizes M=%lld N=%d", (long long)M, N);
  if (M == 0) return 0;
  hipLaunchKernelGGL(k_ids_to_i32, dim3(grid_cap(ceil_div(M, 256))), dim3(256), 0, (hipStream_t)stream,
                     (const long long*)ids, out, (long long)M, N, err_flag);
  TG_LAUNCH_CHECK();
  return 0;
}

extern "C" int64_t tg_csr_workspace_ints(int64_t M, int32_t N) {
  return (int64_t)N + 1 + M + ceil_div((long long)N + 1, SCAN_CHUNK) + 16 + (2 + M / RANK_HUB);
}

extern "C" int tg_csr_build(const int32_t* key, int64_t M, int32_t N, int32_t* rowptr, int32_t* perm, int32_t* work,
                            void* stream_) {
  TG_CHECK(M >= 0 && N > 0 && M < 2147483647LL, "tg_csr_build: bad sizes M=%lld N=%d", (long long)M, N);
  hipStream_t st = (hipStream_t)stream_;
  long long n1 = (long long)N + 1;
  int nchunks = ceil_div(n1, SCAN_CHUNK);
  int* cursor = work;              // [N+1]
  int* tmp = work + n1;            // [M]
  int* totals = work + n1 + M;     // [nchunks]
  zero_async(rowptr, (size_t)n1 * sizeof(int), st);
  if (M > 0) {
    hipLaunchKernelGGL(k_hist, dim3(ceil_div(M, CSR_CH)), dim3(256), 0, st, key, (long long)M, rowptr);
  }
  hipLaunchKernelGGL(k_scan_chunks, dim3(nchunks), dim3(SCAN_T), 0, st, rowptr, n1, totals);
  hipLaunchKernelGGL(k_scan_totals, dim3(1), dim3(1024), 0, st, totals, nchunks);
  hipLaunchKernelGGL(k_scan_add, dim3(nchunks), dim3(SCAN_T), 0, st, rowptr, n1, totals, cursor);
  if (M > 0) {
    hipLaunchKernelGGL(k_fill, dim3(ceil_div(M, CSR_CH)), dim3(256), 0, st, key, (long long)M, cursor, tmp);
    int* hub = totals + nchunks + 8;   // [2 + M / RANK_HUB]
    zero_async(hub, sizeof(int), st);
    hipLaunchKernelGGL(k_rank, dim3(grid_cap(ceil_div(M, 256), 256 * 16)), dim3(256), 0, st, key, rowptr, tmp, perm,
                       (long long)M, hub);
    hipLaunchKernelGGL(k_rank_hub, dim3(128), dim3(1024), 0, st, rowptr, tmp, perm, (long long)M, hub);
  }
  TG_LAUNCH_CHECK();
  return 0;
}

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

extern "C" int tg_seed_pool_fwd(const void* x, const void* xf, const int32_t* rowptr, const int32_t* perm, void* out,
                                int32_t N, int32_t F, int32_t B, int32_t C, int32_t dt, void* stream) {
  TG_CHECK(F % 8 == 0 && C % 8 == 0, "tg_seed_pool_fwd: F and C must be multiples of 8 (F=%d C=%d)", F, C);
  int D = C + 2 * F;
  DISPATCH_T(dt, {
    long long total = (long long)N * (F / VEC);
    hipLaunchKernelGGL((k_seed_pool_fwd<T, VEC>), dim3(grid_cap(ceil_div(total, 256))), dim3(256), 0,
                       (hipStream_t)stream, (const T*)x, (const T*)xf, rowptr, perm, (T*)out, N, F, B, C, D);
  })
  TG_LAUNCH_CHECK();
  return 0;
}

extern "C" int tg_seed_pool_inplace(void* x, const void* xf, const int32_t* tei, const int32_t* rowptr,
                                    const int32_t* perm, int32_t N, int32_t F, int32_t B, int32_t C, int32_t dt,
                                    void* stream) {
  TG_CHECK(F % 8 == 0 && C % 8 == 0, "tg_seed_pool_inplace: F and C must be multiples of 8 (F=%d C=%d)", F, C);
  if (B == 0) return 0;
  int D = C + 2 * F;
  DISPATCH_T(dt, {
    long long total = (long long)2 * B * (F / VEC);
    hipLaunchKernelGGL((k_seed_pool_inplace<T, VEC>), dim3(grid_cap(ceil_div(total, 256))), dim3(256), 0,
                       (hipStream_t)stream, (T*)x, (const T*)xf, tei, rowptr, perm, F, B, C, D);
  })
  TG_LAUNCH_CHECK();
  return 0;
}

extern "C" int tg_seed_pool_bwd(const void* g, const int32_t* tei, const int32_t* rowptr, void* dx, void* dxf,
                                int32_t N, int32_t F, int32_t B, int32_t C, int32_t dt, void* stream) {
  TG_CHECK(F % 8 == 0 && C % 8 == 0, "tg_seed_pool_bwd: F and C must be multiples of 8 (F=%d C=%d)", F, C);
  int D = C + 2 * F;
  DISPATCH_T(dt, {
    long long total = (long long)N * (F / VEC);
    hipLaunchKernelGGL((k_seed_pool_bwd_x<T, VEC>), dim3(grid_cap(ceil_div(total, 256))), dim3(256), 0,
                       (hipStream_t)stream, (const T*)g, rowptr, (T*)dx, N, F);
    long long tf = (long long)B * (D / VEC);
    hipLaunchKernelGGL((k_seed_pool_bwd_f<T, VEC>), dim3(grid_cap(ceil_div(tf, 256))), dim3(256), 0,
                       (hipStream_t)stream, (const T*)g, tei, rowptr, (T*)dxf, B, F, C, D);
  })
  TG_LAUNCH_CHECK();
  return 0;
}
