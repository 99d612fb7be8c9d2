// Train-step tail of the path (reference main.py:70-75, 335-336): class-weighted cross-entropy on the seed-edge
// logits and one fused Adam update over the flat parameter buffer (fp32 master weights; optional bf16 shadow
// copy for the next forward and gradient zeroing in the same pass: 16 B read + 12(+2) B written per parameter).
#include "common.hpp"
#include "../../include/tabgnn_hip.h"

namespace tg {

template <typename T>
__global__ void __launch_bounds__(256) k_ce_partials(const T* __restrict__ logits, const long long* __restrict__ y,
                                                      const float* __restrict__ w, long long B, int K,
                                                      float* __restrict__ partials) {
  __shared__ float sn[4], sd[4];
  float num = 0.f, den = 0.f;
  for (long long r = blockIdx.x * 256LL + threadIdx.x; r < B; r += (long long)gridDim.x * 256) {
    const T* row = logits + r * K;
    float m = -INFINITY;
    for (int k = 0; k < K; ++k) m = fmaxf(m, to_f<T>(row[k]));
    float s = 0.f;
    for (int k = 0; k < K; ++k) s += expf(to_f<T>(row[k]) - m);
    long long yy = y[r];
    yy = yy < 0 ? 0 : (yy >= K ? K - 1 : yy);
    float wy = w ? w[yy] : 1.f;
    num += wy * (m + logf(s) - to_f<T>(row[yy]));
    den += wy;
  }
  num = group_sum<64>(num);
  den = group_sum<64>(den);
  int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) { sn[wid] = num; sd[wid] = den; }
  __syncthreads();
  if (threadIdx.x == 0) {
    partials[2 * blockIdx.x] = sn[0] + sn[1] + sn[2] + sn[3];
    partials[2 * blockIdx.x + 1] = sd[0] + sd[1] + sd[2] + sd[3];
  }
}

__global__ void k_ce_finalize(const float* __restrict__ partials, int nblk, float* __restrict__ out /*[2]: loss, den*/) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float n = 0.f, d = 0.f;
  for (int i = 0; i < nblk; ++i) { n += partials[2 * i]; d += partials[2 * i + 1]; }
  out[0] = n / d;
  out[1] = d;
}

template <typename T>
__global__ void k_ce_bwd(const T* __restrict__ logits, const long long* __restrict__ y, const float* __restrict__ w,
                         const float* __restrict__ lossden, const float* __restrict__ gloss, long long B, int K,
                         T* __restrict__ dlogits) {
  float scale = gloss[0] / lossden[1];
  for (long long r = blockIdx.x * 256LL + threadIdx.x; r < B; r += (long long)gridDim.x * 256) {
    const T* row = logits + r * K;
    float m = -INFINITY;
    for (int k = 0; k < K; ++k) m = fmaxf(m, to_f<T>(row[k]));
    float s = 0.f;
    for (int k = 0; k < K; ++k) s += expf(to_f<T>(row[k]) - m);
    long long yy = y[r];
    yy = yy < 0 ? 0 : (yy >= K ? K - 1 : yy);
    float wy = (w ? w[yy] : 1.f) * scale;
    float inv = 1.f / s;
    for (int k = 0; k < K; ++k) {
      float pk = expf(to_f<T>(row[k]) - m) * inv;
      dlogits[r * K + k] = from_f<T>(wy * (pk - (k == yy ? 1.f : 0.f)));
    }
  }
}

__global__ void k_adam(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                       unsigned short* __restrict__ p_bf16, long long n, float step_size, float beta1, float beta2,
                       float inv_sqrt_bc2, float eps, float grad_scale, int zero_grad,
                       const float* __restrict__ coef) {
  if (coef) { step_size = coef[0]; inv_sqrt_bc2 = coef[1]; }      // device-resident step state (tg_advance_step)
  long long i = (blockIdx.x * (long long)blockDim.x + threadIdx.x) * 4;
  long long stride = (long long)gridDim.x * blockDim.x * 4;
  for (; i < n; i += stride) {
    if (i + 4 <= n) {
      float4 pp = *reinterpret_cast<float4*>(p + i), gg = *reinterpret_cast<float4*>(g + i);
      float4 mm = *reinterpret_cast<float4*>(m + i), vv = *reinterpret_cast<float4*>(v + i);
      float P[4] = {pp.x, pp.y, pp.z, pp.w}, G[4] = {gg.x, gg.y, gg.z, gg.w};
      float M[4] = {mm.x, mm.y, mm.z, mm.w}, V[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float gj = G[j] * grad_scale;
        M[j] = beta1 * M[j] + (1.f - beta1) * gj;
        V[j] = beta2 * V[j] + (1.f - beta2) * gj * gj;
        P[j] -= step_size * M[j] / (sqrtf(V[j]) * inv_sqrt_bc2 + eps);
      }
      *reinterpret_cast<float4*>(p + i) = make_float4(P[0], P[1], P[2], P[3]);
      *reinterpret_cast<float4*>(m + i) = make_float4(M[0], M[1], M[2], M[3]);
      *reinterpret_cast<float4*>(v + i) = make_float4(V[0], V[1], V[2], V[3]);
      if (zero_grad) *reinterpret_cast<float4*>(g + i) = make_float4(0.f, 0.f, 0.f, 0.f);
      if (p_bf16) {
        uint2 o;
        o.x = (unsigned)f2bf(P[0]) | ((unsigned)f2bf(P[1]) << 16);
        o.y = (unsigned)f2bf(P[2]) | ((unsigned)f2bf(P[3]) << 16);
        *reinterpret_cast<uint2*>(p_bf16 + i) = o;
      }
    } else {
      for (long long k = i; k < n; ++k) {
        float gj = g[k] * grad_scale;
        float mk = beta1 * m[k] + (1.f - beta1) * gj;
        float vk = beta2 * v[k] + (1.f - beta2) * gj * gj;
        float pk = p[k] - step_size * mk / (sqrtf(vk) * inv_sqrt_bc2 + eps);
        p[k] = pk; m[k] = mk; v[k] = vk;
        if (zero_grad) g[k] = 0.f;
        if (p_bf16) p_bf16[k] = f2bf(pk);
      }
    }
  }
}

__global__ void k_cast_f32_bf16(const float* __restrict__ x, unsigned short* __restrict__ y, long long n) {
  long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  long long stride = (long long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) y[i] = f2bf(x[i]);
}

}  // namespace tg

using namespace tg;

extern "C" int tg_weighted_ce_fwd(const void* logits, const int64_t* y, const float* w, int64_t B, int32_t K,
                                  float* lossden, float* partials /*[2*256]*/, int32_t dt, void* stream) {
  TG_CHECK(B > 0 && K > 0, "tg_weighted_ce_fwd: bad sizes B=%lld K=%d", (long long)B, K);
  hipStream_t st = (hipStream_t)stream;
  int grid = grid_cap(ceil_div(B, 256), 256);
  if (dt == F32)
    hipLaunchKernelGGL((k_ce_partials<float>), dim3(grid), dim3(256), 0, st, (const float*)logits, (const long long*)y,
                       w, (long long)B, K, partials);
  else
    hipLaunchKernelGGL((k_ce_partials<bf16_t>), dim3(grid), dim3(256), 0, st, (const bf16_t*)logits,
                       (const long long*)y, w, (long long)B, K, partials);
  hipLaunchKernelGGL(k_ce_finalize, dim3(1), dim3(64), 0, st, partials, grid, lossden);
  TG_LAUNCH_CHECK();
  return 0;
}

extern "C" int tg_weighted_ce_bwd(const void* logits, const int64_t* y, const float* w, const float* lossden,
                                  const float* gloss, int64_t B, int32_t K, void* dlogits, int32_t dt, void* stream) {
  TG_CHECK(B > 0 && K > 0, "tg_weighted_ce_bwd: bad sizes B=%lld K=%d", (long long)B, K);
  hipStream_t st = (hipStream_t)stream;
  int grid = grid_cap(ceil_div(B, 256), 2048);
  if (dt == F32)
    hipLaunchKernelGGL((k_ce_bwd<float>), dim3(grid), dim3(256), 0, st, (const float*)logits, (const long long*)y, w,
                       lossden, gloss, (long long)B, K, (float*)dlogits);
  else
    hipLaunchKernelGGL((k_ce_bwd<bf16_t>), dim3(grid), dim3(256), 0, st, (const bf16_t*)logits, (const long long*)y, w,
                       lossden, gloss, (long long)B, K, (bf16_t*)dlogits);
  TG_LAUNCH_CHECK();
  return 0;
}

extern "C" int tg_adam_step(float* p, float* g, float* m, float* v, void* p_bf16, int64_t n, float lr, float beta1,
                            float beta2, float eps, int32_t t, float grad_scale, int32_t zero_grad, void* stream) {
  TG_CHECK(n >= 0 && t >= 1, "tg_adam_step: bad n=%lld t=%d", (long long)n, t);
  if (n == 0) return 0;
  double bc1 = 1.0 - pow((double)beta1, (double)t), bc2 = 1.0 - pow((double)beta2, (double)t);
  hipLaunchKernelGGL(k_adam, dim3(grid_cap(ceil_div(n, 1024))), dim3(256), 0, (hipStream_t)stream, p, g, m, v,
                     (unsigned short*)p_bf16, (long long)n, (float)(lr / bc1), beta1, beta2, (float)(1.0 / sqrt(bc2)),
                     eps, grad_scale, zero_grad, (const float*)nullptr);
  TG_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------- device-resident step state (HIP-graph replay)
// A captured graph bakes every host-side scalar into its kernel nodes.  The three scalars of the train step that
// change from step to step therefore live in a 32-byte device record `state`:
//   word 0 (u64)  dropout seed word (kernels read it through a TG_SEED_DEVICE seed argument: common.hpp:live_seed)
//   word 1 (i64)  optimiser step count t
//   word 2        two floats: lr / (1 - beta1^t), 1 / sqrt(1 - beta2^t)   (Adam's bias corrections for this t)
//   word 3        reserved
// tg_advance_step (one single-thread kernel, first node of the graph) moves the record to the next step;
// tg_adam_step_dev reads the corrections from it.
namespace tg {
__global__ void k_step_advance(unsigned long long* __restrict__ state, float lr, float beta1, float beta2) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  // (bit 63 stays clear: a seed word with it set would read as a device address, common.hpp:live_seed)
  state[0] = (state[0] * 6364136223846793005ULL + 1442695040888963407ULL) & 0x7fffffffffffffffULL;
  const long long t = (long long)state[1] + 1;
  state[1] = (unsigned long long)t;
  const double bc1 = 1.0 - pow((double)beta1, (double)t), bc2 = 1.0 - pow((double)beta2, (double)t);
  float* c = reinterpret_cast<float*>(state + 2);
  c[0] = (float)((double)lr / bc1);
  c[1] = (float)(1.0 / sqrt(bc2));
}
}  // namespace tg

extern "C" int tg_advance_step(uint64_t* state, float lr, float beta1, float beta2, void* stream) {
  TG_CHECK(state != nullptr, "tg_advance_step: %s", "state is NULL");
  hipLaunchKernelGGL(tg::k_step_advance, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned long long*)state, lr, beta1, beta2);
  TG_LAUNCH_CHECK();
  return 0;
}

extern "C" int tg_adam_step_dev(float* p, float* g, float* m, float* v, void* p_bf16, int64_t n, float beta1,
                                float beta2, float eps, const uint64_t* state, float grad_scale, int32_t zero_grad,
                                void* stream) {
  TG_CHECK(n >= 0 && state != nullptr, "tg_adam_step_dev: bad n=%lld or NULL state", (long long)n);
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_adam, dim3(grid_cap(ceil_div(n, 1024))), dim3(256), 0, (hipStream_t)stream, p, g, m, v,
                     (unsigned short*)p_bf16, (long long)n, 0.f, beta1, beta2, 1.f, eps, grad_scale, zero_grad,
                     reinterpret_cast<const float*>(state + 2));
  TG_LAUNCH_CHECK();
  return 0;
}

// Stream-ordered zero fill by a kernel (never a memset node: see common.hpp:zero_async)
extern "C" int tg_zero(void* p, int64_t bytes, void* stream) {
  TG_CHECK(bytes >= 0 && bytes % 4 == 0, "tg_zero: bytes=%lld must be a multiple of 4", (long long)bytes);
  if (bytes) tg::zero_async(p, (size_t)bytes, (hipStream_t)stream);
  TG_LAUNCH_CHECK();
  return 0;
}


extern "C" int tg_cast_f32_to_bf16(const float* x, void* y, int64_t n, void* stream) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_cast_f32_bf16, dim3(grid_cap(ceil_div(n, 256))), dim3(256), 0, (hipStream_t)stream, x,
                     (unsigned short*)y, (long long)n);
  TG_LAUNCH_CHECK();
  return 0;
}

// Batched transpose of the 2-D parameters' bf16 shadows (one launch for the whole model, after the optimiser step):
// dst[off .. off+rows*cols) viewed [cols, rows] = transpose of src[off ..) viewed [rows, cols].  The input-gradient
// GEMMs (dX = G W) read W^T as their row-major weight, so no per-call transpose kernels run.
namespace tg {
__global__ void __launch_bounds__(256) k_transpose_batched(const unsigned short* __restrict__ src,
                                                           unsigned short* __restrict__ dst,
                                                           const long long* __restrict__ table) {
  __shared__ unsigned short tile[32][33];
  const long long off = table[3 * blockIdx.z];
  const int rows = (int)table[3 * blockIdx.z + 1], cols = (int)table[3 * blockIdx.z + 2];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8 threads, 32 x 32 tiles
  for (int r0 = blockIdx.y * 32; r0 < rows; r0 += gridDim.y * 32)
    for (int c0 = blockIdx.x * 32; c0 < cols; c0 += gridDim.x * 32) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int r = r0 + ty + 8 * k, c = c0 + tx;
        if (r < rows && c < cols) tile[ty + 8 * k][tx] = src[off + (long long)r * cols + c];
      }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int c = c0 + ty + 8 * k, r = r0 + tx;
        if (r < rows && c < cols) dst[off + (long long)c * rows + r] = tile[tx][ty + 8 * k];
      }
      __syncthreads();
    }
}
}  // namespace tg

// table: int64 [n][3] on the device = (element offset, rows, cols) of each matrix inside src / dst
extern "C" int tg_transpose_batched_bf16(const void* src, void* dst, const int64_t* table, int32_t n, void* stream) {
  if (n <= 0) return 0;
  TG_CHECK(src && dst && table, "tg_transpose_batched_bf16: null argument");
  // 16 x 16 workgroups per matrix: the four 1536-wide fuse-MLP matrices (86 % of the elements) get 256 workgroups each
  // (with 8 x 8 the launch ran on 256 workgroups in all: 56 us for 28 MB)
  hipLaunchKernelGGL(tg::k_transpose_batched, dim3(16, 16, n), dim3(256), 0, (hipStream_t)stream,
                     (const unsigned short*)src, (unsigned short*)dst, (const long long*)table);
  TG_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------- library plumbing
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <math.h>
namespace tg {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace tg

extern "C" const char* tg_last_error(void) { return tg::g_err; }
extern "C" int tg_abi_version(void) { return TABGNN_HIP_ABI_VERSION; }
extern "C" int tg_device_check(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n == 0) {
    tg::set_error("no HIP device visible");
    return 1;
  }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, 0) != hipSuccess) {
    tg::set_error("hipGetDeviceProperties failed");
    return 2;
  }
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    tg::set_error("device 0 is %s, this library is built for gfx950 only", prop.gcnArchName);
    return 3;
  }
  return 0;
}
