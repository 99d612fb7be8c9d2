// GINEConv message passing on the sampled subgraph (torch_geometric 2.5.3 GINEConv as configured at
// src/nn/gnn/gine.py:18-19,62-72; restated in oracle/gine.py):
//   out[n] = self_scale * x[n] + sum_{e : dst[e] = n} relu(x[src[e]] + le[e]),   le = lin(edge_attr)
// (self_scale = 1 + eps for GINEConv(x), 0 for the ((x, None)) form GINEConvHetero uses, gine.py:31-32).
// The [E,F] message tensor never exists: the forward adds, rectifies and sums in one pass over the by-destination CSR
// (HBM: E*(2F*b + 8) bytes read + N*F*b written), in CSR order (deterministic, no atomics); the backward rebuilds the
// rectifier mask from x[src] + le, writes d_le edge-parallel, and the by-source segmented sum of d_le (tg_segment_sum2,
// which already has a hub pass) gives dx.
#include "common.hpp"
#include "../../include/tabgnn_hip.h"

namespace tg {

constexpr int GINE_HUB = 256;   // destinations with more rows go to the block-per-hub pass

template <typename T, int VEC>
__global__ void __launch_bounds__(256) k_gine_aggregate_fwd(const T* __restrict__ x, const T* __restrict__ le,
                                                             const int* __restrict__ src,
                                                             const int* __restrict__ rowptr,
                                                             const int* __restrict__ perm, float self_scale,
                                                             T* __restrict__ out, int N, int F,
                                                             int* __restrict__ hub /*[0]=count, [1..]=ids*/) {
  const int lpn = F / VEC;
  long long gid = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  const long long stride = (long long)gridDim.x * blockDim.x;
  const long long total = (long long)N * lpn;
  for (; gid < total; gid += stride) {
    const int n = (int)(gid / lpn), c = (int)(gid % lpn) * VEC;
    const int s = rowptr[n], e = rowptr[n + 1];
    if (e - s > GINE_HUB) {
      if (c == 0) hub[1 + atomicAdd(hub, 1)] = n;
      continue;
    }
    float acc[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
    int q = s;
    for (; q + 1 < e; q += 2) {   // two edges (four rows) in flight
      const int e0 = perm[q], e1 = perm[q + 1];
      const int s0 = src[e0], s1 = src[e1];
      float a0[VEC], b0[VEC], a1[VEC], b1[VEC];
      loadv<T, VEC>(x + (long long)s0 * F + c, a0);
      loadv<T, VEC>(le + (long long)e0 * F + c, b0);
      loadv<T, VEC>(x + (long long)s1 * F + c, a1);
      loadv<T, VEC>(le + (long long)e1 * F + c, b1);
#pragma unroll
      for (int j = 0; j < VEC; ++j) acc[j] = (acc[j] + fmaxf(a0[j] + b0[j], 0.f)) + fmaxf(a1[j] + b1[j], 0.f);
    }
    if (q < e) {
      const int e0 = perm[q];
      float a0[VEC], b0[VEC];
      loadv<T, VEC>(x + (long long)src[e0] * F + c, a0);
      loadv<T, VEC>(le + (long long)e0 * F + c, b0);
#pragma unroll
      for (int j = 0; j < VEC; ++j) acc[j] += fmaxf(a0[j] + b0[j], 0.f);
    }
    if (self_scale != 0.f) {
      float xs[VEC];
      loadv<T, VEC>(x + (long long)n * F + c, xs);
#pragma unroll
      for (int j = 0; j < VEC; ++j) acc[j] += self_scale * xs[j];
    }
    storev<T, VEC>(out + (long long)n * F + c, acc);
  }
}

// one 1024-thread block per hub destination: lane groups take strided edges (four in flight, clamped index + select so
// no load sits behind a branch), partial sums meet in LDS and are combined in group order (deterministic)
template <typename T, int VEC>
__global__ void __launch_bounds__(1024) k_gine_aggregate_fwd_hub(const T* __restrict__ x, const T* __restrict__ le,
                                                                  const int* __restrict__ src,
                                                                  const int* __restrict__ rowptr,
                                                                  const int* __restrict__ perm, float self_scale,
                                                                  T* __restrict__ out, int F,
                                                                  const int* __restrict__ hub) {
  extern __shared__ float part[];  // [groups][F]
  const int lpn = F / VEC, groups = 1024 / lpn;
  const int gi = threadIdx.x / lpn, c = (threadIdx.x % lpn) * VEC;
  const int nh = hub[0];
  for (int hIdx = blockIdx.x; hIdx < nh; hIdx += gridDim.x) {
    const int n = hub[1 + hIdx];
    const int s = rowptr[n], e = rowptr[n + 1];
    float acc[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
    for (int q = s + gi; q < e; q += groups * 4) {
      float a[4][VEC], b[4][VEC];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int qq = q + u * groups;
        const int ee = perm[qq < e ? qq : q];
        loadv<T, VEC>(x + (long long)src[ee] * F + c, a[u]);
        loadv<T, VEC>(le + (long long)ee * F + c, b[u]);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const bool ok = q + u * groups < e;
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[j] += ok ? fmaxf(a[u][j] + b[u][j], 0.f) : 0.f;
      }
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) part[gi * F + c + j] = acc[j];
    __syncthreads();
    for (int f = threadIdx.x; f < F; f += 1024) {
      float t = 0.f;
      for (int g2 = 0; g2 < groups; ++g2) t += part[g2 * F + f];
      if (self_scale != 0.f) t += self_scale * to_f<T>(x[(long long)n * F + f]);
      out[(long long)n * F + f] = from_f<T>(t);
    }
    __syncthreads();
  }
}

// d_le[e] = (x[src[e]] + le[e] > 0) ? dout[dst[e]] : 0     (edge-parallel; the mask is rebuilt, nothing was saved)
template <typename T, int VEC>
__global__ void __launch_bounds__(256) k_gine_message_bwd(const T* __restrict__ x, const T* __restrict__ le,
                                                           const T* __restrict__ dout, const int* __restrict__ src,
                                                           const int* __restrict__ dst, T* __restrict__ dle,
                                                           long long E, int F) {
  const int lpn = F / VEC;
  long long gid = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  const long long stride = (long long)gridDim.x * blockDim.x;
  const long long total = E * lpn;
  for (; gid < total; gid += stride) {
    const long long e = gid / lpn;
    const int c = (int)(gid % lpn) * VEC;
    float a[VEC], b[VEC], g[VEC];
    loadv<T, VEC>(x + (long long)src[e] * F + c, a);
    loadv<T, VEC>(le + e * F + c, b);
    loadv<T, VEC>(dout + (long long)dst[e] * F + c, g);
#pragma unroll
    for (int j = 0; j < VEC; ++j) g[j] = (a[j] + b[j] > 0.f) ? g[j] : 0.f;
    storev<T, VEC>(dle + e * F + c, g);
  }
}

}  // namespace tg

using namespace tg;

#define GINE_DISPATCH_T(dt, ...)                  \
  if ((dt) == F32) {                              \
    using T = float;                              \
    constexpr int VEC = 4;                        \
    __VA_ARGS__                                   \
  } else {                                        \
    using T = bf16_t;                             \
    constexpr int VEC = 8;                        \
    __VA_ARGS__                                   \
  }

extern "C" int tg_gine_aggregate_fwd(const void* x, const void* le, const int32_t* src, const int32_t* rowptr,
                                     const int32_t* perm, float self_scale, void* out, int32_t N, int32_t F,
                                     int32_t* hub_work, int32_t dt, void* stream) {
  TG_CHECK(F % 8 == 0 && F >= 8, "tg_gine_aggregate_fwd: F=%d must be a multiple of 8", F);
  TG_CHECK(x && le && src && rowptr && perm && out && hub_work, "tg_gine_aggregate_fwd: null argument");
  if (N == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  zero_async(hub_work, sizeof(int), st);
  GINE_DISPATCH_T(dt, {
    TG_CHECK(1024 % (F / VEC) == 0 && (size_t)(1024 / (F / VEC)) * F * sizeof(float) <= 150 * 1024,
             "tg_gine_aggregate_fwd: unsupported F=%d", F);
    long long total = (long long)N * (F / VEC);
    hipLaunchKernelGGL((k_gine_aggregate_fwd<T, VEC>), dim3(grid_cap(ceil_div(total, 256), 256 * 16)), dim3(256), 0, st,
                       (const T*)x, (const T*)le, src, rowptr, perm, self_scale, (T*)out, N, F, hub_work);
    size_t shm = (size_t)(1024 / (F / VEC)) * F * sizeof(float);
    hipLaunchKernelGGL((k_gine_aggregate_fwd_hub<T, VEC>), dim3(256), dim3(1024), shm, st, (const T*)x, (const T*)le,
                       src, rowptr, perm, self_scale, (T*)out, F, hub_work);
  })
  TG_LAUNCH_CHECK();
  return 0;
}

extern "C" int tg_gine_message_bwd(const void* x, const void* le, const void* dout, const int32_t* src,
                                   const int32_t* dst, void* dle, int64_t E, int32_t F, int32_t dt, void* stream) {
  TG_CHECK(F % 8 == 0 && F >= 8, "tg_gine_message_bwd: F=%d must be a multiple of 8", F);
  if (E == 0) return 0;
  TG_CHECK(x && le && dout && src && dst && dle, "tg_gine_message_bwd: null argument");
  GINE_DISPATCH_T(dt, {
    long long total = (long long)E * (F / VEC);
    hipLaunchKernelGGL((k_gine_message_bwd<T, VEC>), dim3(grid_cap(ceil_div(total, 256), 256 * 16)), dim3(256), 0,
                       (hipStream_t)stream, (const T*)x, (const T*)le, (const T*)dout, src, dst, (T*)dle, (long long)E, F);
  })
  TG_LAUNCH_CHECK();
  return 0;
}
