// Normalisation / elementwise epilogues of the hot path, fused the way the reference composes them:
//   * LayerNorm with the residual + dropout(bias + GEMM-out) pre-add of torch's post-norm encoder layer
//     (norm1/norm2, src/nn/models/fused.py:83-92) and with the residual combine that follows tab_norm /
//     fuse_norm (fused.py:160,164,249,258; tabgnn.py:219):  out = alpha*res + beta*LN(a + drop(b + bias_b))
//   * BatchNorm1d + ReLU + residual average of the PNA branch (fused.py:252):  out = alpha*res + beta*relu(BN(c))
//   * activation + dropout after a Linear (FFN, fuse MLP, heads).
// All HBM-bound: one read of each input, one write of the output, 16-byte accesses, statistics in fp32.
#include "common.hpp"
#include "../../include/tabgnn_hip.h"

namespace tg {

constexpr int LN_BLOCK = 256;

// ------------------------------------------------------------------ LayerNorm forward
template <typename T, int VEC, int VPL>
__global__ void __launch_bounds__(LN_BLOCK) k_ln_fwd(const T* __restrict__ a, const T* __restrict__ b,
                                                      const float* __restrict__ bias_b, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, const T* __restrict__ res,
                                                      T* __restrict__ out, float* __restrict__ stats, long long M, int C,
                                                      int lpr, float eps, float alpha, float beta_c, unsigned thresh,
                                                      float inv_keep, unsigned long long seed, unsigned rstream) {
  seed = live_seed(seed);
  const int groups = LN_BLOCK / lpr;
  const int gl = threadIdx.x % lpr;
  long long row = (long long)blockIdx.x * groups + threadIdx.x / lpr;
  const long long rstride = (long long)gridDim.x * groups;
  const int nvec = C / VEC;
  // a thread keeps its channel vectors for every row it visits: per-channel parameters live in registers
  float gmr[VPL][VEC], btr[VPL][VEC], bsr[VPL][VEC];
#pragma unroll
  for (int k = 0; k < VPL; ++k) {
    int vi = gl + k * lpr;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      bool ok = vi < nvec;
      gmr[k][j] = ok ? gamma[vi * VEC + j] : 0.f;
      btr[k][j] = ok ? beta[vi * VEC + j] : 0.f;
      bsr[k][j] = (ok && bias_b) ? bias_b[vi * VEC + j] : 0.f;
    }
  }
  for (; row < M; row += rstride) {  // lpr divides 64, M rows: whole groups iterate together
    float x[VPL][VEC];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < VPL; ++k) {
      int vi = gl + k * lpr;
      if (vi < nvec) {
        int c = vi * VEC;
        loadv<T, VEC>(a + row * C + c, x[k]);
        if (b) {
          float t[VEC];
          loadv<T, VEC>(b + row * C + c, t);
          const unsigned long long e0 = (unsigned long long)(row * C + c);
          const unsigned key = rng_key(seed, rstream, (unsigned)(e0 >> 32));   // c % VEC == 0: no carry within the vector
#pragma unroll
          for (int j = 0; j < VEC; ++j) {   // branch-free: the mask is a select, so nothing splits the loads
            float m = drop_scale_key(key, (unsigned)e0 + j, thresh, inv_keep);
            x[k][j] += (t[j] + bsr[k][j]) * (thresh ? m : 1.f);
          }
        }
#pragma unroll
        for (int j = 0; j < VEC; ++j) s += x[k][j];
      } else {
#pragma unroll
        for (int j = 0; j < VEC; ++j) x[k][j] = 0.f;
      }
    }
    for (int o = lpr >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    float mu = s / (float)C;
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < VPL; ++k) {
      int vi = gl + k * lpr;
      if (vi < nvec) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) { float d = x[k][j] - mu; v += d * d; }
      }
    }
    for (int o = lpr >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    float rstd = rsqrtf(v / (float)C + eps);
    if (gl == 0 && stats) { stats[2 * row] = mu; stats[2 * row + 1] = rstd; }
#pragma unroll
    for (int k = 0; k < VPL; ++k) {
      int vi = gl + k * lpr;
      if (vi < nvec) {
        int c = vi * VEC;
        float y[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) y[j] = beta_c * ((x[k][j] - mu) * rstd * gmr[k][j] + btr[k][j]);
        if (res) {
          float r[VEC];
          loadv<T, VEC>(res + row * C + c, r);
#pragma unroll
          for (int j = 0; j < VEC; ++j) y[j] += alpha * r[j];
        }
        storev<T, VEC>(out + row * C + c, y);
      }
    }
  }
}

// ------------------------------------------------------------------ LayerNorm backward
// da = dpre; db = dpre * dropmask; dres = alpha*dout; partials[blk][3][C] = (dgamma, dbeta, dbias_b)
template <typename T, int VEC, int VPL>
__global__ void __launch_bounds__(LN_BLOCK) k_ln_bwd(const T* __restrict__ a, const T* __restrict__ b,
                                                      const float* __restrict__ bias_b, const float* __restrict__ gamma,
                                                      const float* __restrict__ stats, const T* __restrict__ dout,
                                                      T* __restrict__ da, T* __restrict__ db, T* __restrict__ dres,
                                                      float* __restrict__ partials, long long M, int C, int lpr,
                                                      float alpha, float beta_c, unsigned thresh, float inv_keep,
                                                      unsigned long long seed, unsigned rstream, int accum_da) {
  seed = live_seed(seed);
  extern __shared__ float red[];  // [groups][3][C]
  const int groups = LN_BLOCK / lpr;
  const int gl = threadIdx.x % lpr, gi = threadIdx.x / lpr;
  long long row = (long long)blockIdx.x * groups + gi;
  const long long rstride = (long long)gridDim.x * groups;
  const int nvec = C / VEC;
  float dg[VPL][VEC], dbt[VPL][VEC], dbs[VPL][VEC];
#pragma unroll
  for (int k = 0; k < VPL; ++k)
#pragma unroll
    for (int j = 0; j < VEC; ++j) { dg[k][j] = 0.f; dbt[k][j] = 0.f; dbs[k][j] = 0.f; }
  float gmr[VPL][VEC], bsr[VPL][VEC];   // per-channel parameters in registers (fixed channel vectors per thread)
#pragma unroll
  for (int k = 0; k < VPL; ++k) {
    int vi = gl + k * lpr;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      bool ok = vi < nvec;
      gmr[k][j] = ok ? gamma[vi * VEC + j] : 0.f;
      bsr[k][j] = (ok && bias_b) ? bias_b[vi * VEC + j] : 0.f;
    }
  }
  for (; row < M; row += rstride) {
    float mu = stats[2 * row], rstd = stats[2 * row + 1];
    float xh[VPL][VEC], gx[VPL][VEC], msk[VPL][VEC];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < VPL; ++k) {
      int vi = gl + k * lpr;
      if (vi < nvec) {
        int c = vi * VEC;
        float x[VEC], g[VEC];
        loadv<T, VEC>(a + row * C + c, x);
        // b == NULL with db given ("z mode"): a already IS the pre-norm sum z = a0 + drop(b + bias) written by the
        // fused GEMM+LayerNorm forward; only the mask is needed, for db = da * mask
        if (b || db) {
          float t[VEC];
#pragma unroll
          for (int j = 0; j < VEC; ++j) t[j] = 0.f;
          if (b) loadv<T, VEC>(b + row * C + c, t);
          const unsigned long long e0 = (unsigned long long)(row * C + c);
          const unsigned key = rng_key(seed, rstream, (unsigned)(e0 >> 32));
#pragma unroll
          for (int j = 0; j < VEC; ++j) {   // branch-free (select), see k_ln_fwd
            float m = drop_scale_key(key, (unsigned)e0 + j, thresh, inv_keep);
            m = thresh ? m : 1.f;
            msk[k][j] = m;
            if (b) x[j] += (t[j] + bsr[k][j]) * m;
          }
        }
        loadv<T, VEC>(dout + row * C + c, g);
        if (dres) {
          float r[VEC];
#pragma unroll
          for (int j = 0; j < VEC; ++j) r[j] = alpha * g[j];
          storev<T, VEC>(dres + row * C + c, r);
        }
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          float gj = g[j] * beta_c;
          xh[k][j] = (x[j] - mu) * rstd;
          dg[k][j] += gj * xh[k][j];
          dbt[k][j] += gj;
          gx[k][j] = gj * gmr[k][j];
          s1 += gx[k][j];
          s2 += gx[k][j] * xh[k][j];
        }
      }
    }
    for (int o = lpr >> 1; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
    float m1 = s1 / (float)C, m2 = s2 / (float)C;
#pragma unroll
    for (int k = 0; k < VPL; ++k) {
      int vi = gl + k * lpr;
      if (vi < nvec) {
        int c = vi * VEC;
        float d[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) d[j] = rstd * (gx[k][j] - m1 - xh[k][j] * m2);
        if (accum_da) {   // da already holds another branch's gradient of the same tensor: add instead of overwrite
          float old[VEC], sum[VEC];
          loadv<T, VEC>(da + row * C + c, old);
#pragma unroll
          for (int j = 0; j < VEC; ++j) sum[j] = old[j] + d[j];
          storev<T, VEC>(da + row * C + c, sum);
        } else {
          storev<T, VEC>(da + row * C + c, d);
        }
        if (b || db) {
#pragma unroll
          for (int j = 0; j < VEC; ++j) { d[j] *= msk[k][j]; dbs[k][j] += d[j]; }
          storev<T, VEC>(db + row * C + c, d);
        }
      }
    }
  }
  // block reduction of the three column sums (fixed order -> deterministic)
#pragma unroll
  for (int k = 0; k < VPL; ++k) {
    int vi = gl + k * lpr;
    if (vi < nvec) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        int c = vi * VEC + j;
        red[(gi * 3 + 0) * C + c] = dg[k][j];
        red[(gi * 3 + 1) * C + c] = dbt[k][j];
        red[(gi * 3 + 2) * C + c] = dbs[k][j];
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 3 * C; i += LN_BLOCK) {
    float t = 0.f;
    for (int g2 = 0; g2 < groups; ++g2) t += red[g2 * 3 * C + i];
    partials[(long long)blockIdx.x * 3 * C + i] = t;
  }
}

// out[i] = sum_blk partials[blk][i]   (i < width)
// The launch is a handful of blocks (width/64), so it is pure latency: 1024 threads = 64 columns x 16 strips of
// blocks, eight independent loads in flight per thread; strips meet in LDS in strip order (deterministic).
constexpr int RP_STRIPS = 16;
__global__ void __launch_bounds__(1024) k_reduce_partials(const float* __restrict__ partials, int nblk, int width,
                                                           float* __restrict__ out) {
  __shared__ float red[RP_STRIPS][64];
  const int lane = threadIdx.x & 63, strip = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + lane;
  float t = 0.f;
  if (col < width) {
    const int per = (nblk + RP_STRIPS - 1) / RP_STRIPS, b0 = strip * per, b1 = b0 + per < nblk ? b0 + per : nblk;
    float acc[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] = 0.f;
    int b = b0;
    for (; b + 7 < b1; b += 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) acc[u] += partials[(long long)(b + u) * width + col];
    }
    for (; b < b1; ++b) acc[0] += partials[(long long)b * width + col];
    t = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  }
  red[strip][lane] = t;
  __syncthreads();
  if (strip == 0 && col < width) {
    float r = 0.f;
#pragma unroll
    for (int u = 0; u < RP_STRIPS; ++u) r += red[u][lane];
    out[col] = r;
  }
}

// LayerNorm parameter gradients straight into the parameters' gradient buffers: the [3][C] reduction of
// k_reduce_partials with one destination per C-wide segment (dgamma, dbeta, dbias; a NULL destination is skipped) and
// "+=" semantics, so autograd never runs an AccumulateGrad add for them.
__global__ void __launch_bounds__(1024) k_reduce_partials_acc3(const float* __restrict__ partials, int nblk, int C,
                                                                float* __restrict__ o0, float* __restrict__ o1,
                                                                float* __restrict__ o2) {
  __shared__ float red[RP_STRIPS][64];
  const int width = 3 * C;
  const int lane = threadIdx.x & 63, strip = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + lane;
  float t = 0.f;
  if (col < width) {
    const int per = (nblk + RP_STRIPS - 1) / RP_STRIPS, b0 = strip * per, b1 = b0 + per < nblk ? b0 + per : nblk;
    float acc[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] = 0.f;
    int b = b0;
    for (; b + 7 < b1; b += 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) acc[u] += partials[(long long)(b + u) * width + col];
    }
    for (; b < b1; ++b) acc[0] += partials[(long long)b * width + col];
    t = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  }
  red[strip][lane] = t;
  __syncthreads();
  if (strip == 0 && col < width) {
    float r = 0.f;
#pragma unroll
    for (int u = 0; u < RP_STRIPS; ++u) r += red[u][lane];
    const int seg = col / C, c = col - seg * C;
    float* o = seg == 0 ? o0 : (seg == 1 ? o1 : o2);
    if (o) o[c] += r;
  }
}

// ------------------------------------------------------------------ BatchNorm (+ReLU, + residual average)
// column statistics: each lane owns VEC channels, rows strided over lane groups and blocks
template <typename T, int VEC>
__global__ void __launch_bounds__(256) k_col_stats(const T* __restrict__ x, const T* __restrict__ y2,
                                                    const float* __restrict__ mean, const float* __restrict__ rstd,
                                                    const T* __restrict__ gate, float* __restrict__ partials,
                                                    long long N, int F, int mode, const int* __restrict__ nlim) {
  if (nlim) N = min(N, (long long)nlim[0]);       // padded batch: rows past the device-side limit are not part of the batch
  // mode 0: partial (sum x, sum x^2).   mode 1 (backward): with dz = x (upstream, already scaled) gated by
  // gate>0 (if gate), xhat = (y2-mean)*rstd: partial (sum dz, sum dz*xhat)
  extern __shared__ float red[];  // [groups][2][F]
  const int lpr = F / VEC, groups = 256 / lpr;
  const int gl = threadIdx.x % lpr, gi = threadIdx.x / lpr;
  float s1[VEC], s2[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
  const int c = gl * VEC;
  if (gi < groups) {
    for (long long r = (long long)blockIdx.x * groups + gi; r < N; r += (long long)gridDim.x * groups) {
      float v[VEC];
      loadv<T, VEC>(x + r * F + c, v);
      if (mode == 0) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) { s1[j] += v[j]; s2[j] += v[j] * v[j]; }
      } else {
        float u[VEC];
        loadv<T, VEC>(y2 + r * F + c, u);
        if (gate) {
          float gt[VEC];
          loadv<T, VEC>(gate + r * F + c, gt);
#pragma unroll
          for (int j = 0; j < VEC; ++j) v[j] = gt[j] > 0.f ? v[j] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < VEC; ++j) { s1[j] += v[j]; s2[j] += v[j] * (u[j] - mean[c + j]) * rstd[c + j]; }
      }
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) { red[(gi * 2 + 0) * F + c + j] = s1[j]; red[(gi * 2 + 1) * F + c + j] = s2[j]; }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * F; i += 256) {
    float t = 0.f;
    for (int g2 = 0; g2 < groups; ++g2) t += red[g2 * 2 * F + i];
    partials[(long long)blockIdx.x * 2 * F + i] = t;
  }
}

__global__ void k_bn_finalize(const float* __restrict__ partials, int nblk, long long N, int F, float eps,
                              float momentum, float* __restrict__ mean, float* __restrict__ rstd,
                              float* __restrict__ running_mean, float* __restrict__ running_var,
                              const int* __restrict__ nlim) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= F) return;
  if (nlim) N = min(N, (long long)nlim[0]);
  float s1 = partials[c], s2 = partials[F + c];   // already reduced over blocks (k_reduce_partials)
  float mu = s1 / (float)N;
  float var = fmaxf(s2 / (float)N - mu * mu, 0.f);
  mean[c] = mu;
  rstd[c] = rsqrtf(var + eps);
  if (running_mean) {
    float unb = N > 1 ? var * ((float)N / (float)(N - 1)) : var;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mu;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * unb;
  }
}

__global__ void k_bn_eval_stats(const float* __restrict__ running_mean, const float* __restrict__ running_var, int F,
                                float eps, float* __restrict__ mean, float* __restrict__ rstd) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= F) return;
  mean[c] = running_mean[c];
  rstd[c] = rsqrtf(running_var[c] + eps);
}

// out = alpha*res + beta_c*act(BN(x));  bn_out (optional) keeps BN(x) for the ReLU gate of the backward
template <typename T, int VEC>
__global__ void k_bn_apply(const T* __restrict__ x, const T* __restrict__ res, const float* __restrict__ mean,
                           const float* __restrict__ rstd, const float* __restrict__ gamma,
                           const float* __restrict__ beta, T* __restrict__ out, long long N, int F, int relu,
                           float alpha, float beta_c) {
  const int vpr = F / VEC;
  long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  long long stride = (long long)gridDim.x * blockDim.x;
  long long total = N * vpr;
  // 256 % vpr == 0 (checked by the host): a thread keeps its channel vector for the whole grid-stride loop,
  // so the per-channel parameters are loaded once into registers
  const int c = (int)(i % vpr) * VEC;
  float mu[VEC], sc[VEC], bt[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    mu[j] = mean[c + j];
    sc[j] = rstd[c + j] * gamma[c + j];
    bt[j] = beta[c + j];
  }
  for (; i < total; i += stride) {
    long long r = i / vpr;
    float v[VEC];
    loadv<T, VEC>(x + r * F + c, v);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      float y = (v[j] - mu[j]) * sc[j] + bt[j];
      if (relu) y = fmaxf(y, 0.f);
      v[j] = beta_c * y;
    }
    if (res) {
      float t[VEC];
      loadv<T, VEC>(res + r * F + c, t);
#pragma unroll
      for (int j = 0; j < VEC; ++j) v[j] += alpha * t[j];
    }
    storev<T, VEC>(out + r * F + c, v);
  }
}

// backward apply.  dz = beta_c*dout gated by BN(x)>0;  training: dx = gamma*rstd*(dz - sum_dz/N - xhat*sum_dzxhat/N)
// eval: dx = gamma*rstd*dz.  dres = alpha*dout.
template <typename T, int VEC>
__global__ void k_bn_bwd_apply(const T* __restrict__ x, const T* __restrict__ dout, const float* __restrict__ mean,
                               const float* __restrict__ rstd, const float* __restrict__ gamma,
                               const float* __restrict__ beta, const float* __restrict__ sums /*[2][F]*/,
                               T* __restrict__ dx, T* __restrict__ dres, long long N, long long n_stat, int F, int relu,
                               int training, float alpha, float beta_c, const int* __restrict__ nlim) {
  long long n_real = N;                           // rows past it (padding) get dx = 0: they never were in the statistics
  if (nlim) { n_real = min(N, (long long)nlim[0]); n_stat = min(n_stat, n_real); }
  const int vpr = F / VEC;
  long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  long long stride = (long long)gridDim.x * blockDim.x;
  long long total = N * vpr;
  float invN = 1.f / (float)n_stat;
  // per-channel parameters in registers (a thread keeps its channel vector across the loop: 256 % vpr == 0)
  const int c = (int)(i % vpr) * VEC;
  float mu[VEC], rs[VEC], gm[VEC], bt[VEC], m1[VEC], m2[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    mu[j] = mean[c + j]; rs[j] = rstd[c + j]; gm[j] = gamma[c + j]; bt[j] = beta[c + j];
    m1[j] = training ? sums[c + j] * invN : 0.f;
    m2[j] = training ? sums[F + c + j] * invN : 0.f;
  }
  for (; i < total; i += stride) {
    long long r = i / vpr;
    float v[VEC], g[VEC], o[VEC];
    loadv<T, VEC>(x + r * F + c, v);
    loadv<T, VEC>(dout + r * F + c, g);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      float xh = (v[j] - mu[j]) * rs[j];
      float y = xh * gm[j] + bt[j];
      float dz = (relu && !(y > 0.f)) ? 0.f : beta_c * g[j];
      float t = dz - m1[j] - xh * m2[j];
      o[j] = r < n_real ? gm[j] * rs[j] * t : 0.f;
    }
    storev<T, VEC>(dx + r * F + c, o);
    if (dres) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) o[j] = alpha * g[j];
      storev<T, VEC>(dres + r * F + c, o);
    }
  }
}

// column sums for the BN backward need the gated dz: computed on the fly from (x, dout)
template <typename T, int VEC>
__global__ void __launch_bounds__(256) k_bn_bwd_stats(const T* __restrict__ x, const T* __restrict__ dout,
                                                       const float* __restrict__ mean, const float* __restrict__ rstd,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       float* __restrict__ partials, long long N, int F, int relu,
                                                       float beta_c, const int* __restrict__ nlim) {
  if (nlim) N = min(N, (long long)nlim[0]);
  extern __shared__ float red[];
  const int lpr = F / VEC, groups = 256 / lpr;
  const int gl = threadIdx.x % lpr, gi = threadIdx.x / lpr;
  const int c = gl * VEC;
  float s1[VEC], s2[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
  if (gi < groups) {
    for (long long r = (long long)blockIdx.x * groups + gi; r < N; r += (long long)gridDim.x * groups) {
      float v[VEC], g[VEC];
      loadv<T, VEC>(x + r * F + c, v);
      loadv<T, VEC>(dout + r * F + c, g);
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        float xh = (v[j] - mean[c + j]) * rstd[c + j];
        float y = xh * gamma[c + j] + beta[c + j];
        float dz = (relu && !(y > 0.f)) ? 0.f : beta_c * g[j];
        s1[j] += dz;
        s2[j] += dz * xh;
      }
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) { red[(gi * 2 + 0) * F + c + j] = s1[j]; red[(gi * 2 + 1) * F + c + j] = s2[j]; }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * F; i += 256) {
    float t = 0.f;
    for (int g2 = 0; g2 < groups; ++g2) t += red[g2 * 2 * F + i];
    partials[(long long)blockIdx.x * 2 * F + i] = t;
  }
}

// ------------------------------------------------------------------ activation + dropout (after a Linear)
// act: 0 none, 1 relu, 2 leaky_relu(0.01)
template <typename T, int VEC>
__global__ void k_act_dropout_fwd(const T* __restrict__ x, T* __restrict__ y, long long n, int act, unsigned thresh,
                                  float inv_keep, unsigned long long seed, unsigned rstream) {
  seed = live_seed(seed);
  long long i = (blockIdx.x * (long long)blockDim.x + threadIdx.x) * VEC;
  long long stride = (long long)gridDim.x * blockDim.x * VEC;
  for (; i < n; i += stride) {
    const bool full = i + VEC <= n;
    float v[VEC];
    if (full) loadv<T, VEC>(x + i, v);
    else
#pragma unroll
      for (int j = 0; j < VEC; ++j) v[j] = i + j < n ? to_f<T>(x[i + j]) : 0.f;
    const unsigned key = rng_key(seed, rstream, (unsigned)((unsigned long long)i >> 32));   // i % VEC == 0
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      float u = v[j];
      if (act == 1) u = fmaxf(u, 0.f);
      else if (act == 2) u = u > 0.f ? u : 0.01f * u;
      if (thresh) u *= drop_scale_key(key, (unsigned)i + j, thresh, inv_keep);
      v[j] = u;
    }
    if (full) storev<T, VEC>(y + i, v);
    else
#pragma unroll
      for (int j = 0; j < VEC; ++j) if (i + j < n) y[i + j] = from_f<T>(v[j]);
  }
}

// dx = dy * mask * act'(x)   (x = pre-activation, kept by the caller as the Linear's output)
template <typename T, int VEC>
__global__ void k_act_dropout_bwd(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx, long long n,
                                  int act, unsigned thresh, float inv_keep, unsigned long long seed, unsigned rstream) {
  seed = live_seed(seed);
  long long i = (blockIdx.x * (long long)blockDim.x + threadIdx.x) * VEC;
  long long stride = (long long)gridDim.x * blockDim.x * VEC;
  for (; i < n; i += stride) {
    const bool full = i + VEC <= n;
    float v[VEC], g[VEC];
    if (full) { loadv<T, VEC>(x + i, v); loadv<T, VEC>(dy + i, g); }
    else
#pragma unroll
      for (int j = 0; j < VEC; ++j) { v[j] = i + j < n ? to_f<T>(x[i + j]) : 0.f; g[j] = i + j < n ? to_f<T>(dy[i + j]) : 0.f; }
    const unsigned key = rng_key(seed, rstream, (unsigned)((unsigned long long)i >> 32));   // i % VEC == 0
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      float d = g[j];
      if (act == 1) d = v[j] > 0.f ? d : 0.f;
      else if (act == 2) d = v[j] > 0.f ? d : 0.01f * d;
      if (thresh) d *= drop_scale_key(key, (unsigned)i + j, thresh, inv_keep);
      g[j] = d;
    }
    if (full) storev<T, VEC>(dx + i, g);
    else
#pragma unroll
      for (int j = 0; j < VEC; ++j) if (i + j < n) dx[i + j] = from_f<T>(g[j]);
  }
}

// y = alpha*a + beta*b  (edge update residual: fused.py:254 (.5,.5); tabgnn.py:190 (1,.5); x_tab merge fused.py:172)
template <typename T, int VEC>
__global__ void k_axpby(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ y, long long n, float alpha,
                        float beta) {
  long long i = (blockIdx.x * (long long)blockDim.x + threadIdx.x) * VEC;
  long long stride = (long long)gridDim.x * blockDim.x * VEC;
  for (; i < n; i += stride) {
    float u[VEC], v[VEC];
    loadv<T, VEC>(a + i, u);
    loadv<T, VEC>(b + i, v);
#pragma unroll
    for (int j = 0; j < VEC; ++j) u[j] = alpha * u[j] + beta * v[j];
    storev<T, VEC>(y + i, u);
  }
}

// y1 = alpha*a + beta*b and y2 = gamma*b in ONE pass over b (a == nullptr: y1 = beta*b): the backward of an axpby whose
// first input's gradient goes to a shared gradient buffer (ops.GradSink) while the second input takes its own scaled copy
template <typename T, int VEC>
__global__ void k_axpby2(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ y1, T* __restrict__ y2,
                         long long n, float alpha, float beta, float gamma) {
  long long i = (blockIdx.x * (long long)blockDim.x + threadIdx.x) * VEC;
  long long stride = (long long)gridDim.x * blockDim.x * VEC;
  for (; i < n; i += stride) {
    float u[VEC], v[VEC], w[VEC];
    loadv<T, VEC>(b + i, v);
    if (a) {
      loadv<T, VEC>(a + i, u);
#pragma unroll
      for (int j = 0; j < VEC; ++j) u[j] = alpha * u[j] + beta * v[j];
    } else {
#pragma unroll
      for (int j = 0; j < VEC; ++j) u[j] = beta * v[j];
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) w[j] = gamma * v[j];
    storev<T, VEC>(y1 + i, u);
    storev<T, VEC>(y2 + i, w);
  }
}

// Column sums of a strided [R, C] matrix (the CLS-vector gradient: column 0 of a [R, S, C] gradient, fused.py:158-159):
// block b sums rows [b chunk, (b+1) chunk) — C / VEC lanes across a row, 256 / (C / VEC) rows per pass — and writes
// part[b][C]; k_col_sum_reduce adds the partials in block order (deterministic, no atomics).
template <typename T, int VEC>
__global__ void __launch_bounds__(256) k_col_sum_part(const T* __restrict__ x, long long R, int C, long long ld,
                                                      long long chunk, float* __restrict__ part) {
  __shared__ float red[256 * 8];
  const int lpr = C / VEC, rpp = 256 / lpr;                  // lanes per row, rows per pass
  const int lane = threadIdx.x % lpr, rg = threadIdx.x / lpr;
  const long long r0 = blockIdx.x * chunk, r1 = r0 + chunk < R ? r0 + chunk : R;
  float acc[VEC], acc2[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) acc[j] = acc2[j] = 0.f;
  if (rg < rpp) {
    long long r = r0 + rg;
    for (; r + rpp < r1; r += 2 * rpp) {
      float u[VEC], v[VEC];
      loadv<T, VEC>(x + r * ld + lane * VEC, u);
      loadv<T, VEC>(x + (r + rpp) * ld + lane * VEC, v);
#pragma unroll
      for (int j = 0; j < VEC; ++j) { acc[j] += u[j]; acc2[j] += v[j]; }
    }
    if (r < r1) {
      float u[VEC];
      loadv<T, VEC>(x + r * ld + lane * VEC, u);
#pragma unroll
      for (int j = 0; j < VEC; ++j) acc[j] += u[j];
    }
  }
#pragma unroll
  for (int j = 0; j < VEC; ++j) red[threadIdx.x * VEC + j] = acc[j] + acc2[j];
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float t = 0.f;
    for (int g = 0; g < rpp; ++g) t += red[(g * lpr + c / VEC) * VEC + c % VEC];
    part[(long long)blockIdx.x * C + c] = t;
  }
}

__global__ void __launch_bounds__(256) k_col_sum_reduce(const float* __restrict__ part, int nblk, int C,
                                                        float* __restrict__ out, int accumulate) {
  __shared__ float red[256];
  constexpr int cpb = 8, strips = 256 / cpb;                 // 8 columns per block, 32 strips of partials each
  const int c = blockIdx.x * cpb + threadIdx.x % cpb, strip = threadIdx.x / cpb;
  float t0 = 0.f, t1 = 0.f;
  if (c < C) {
    const int per = (nblk + strips - 1) / strips, b0 = strip * per, b1 = b0 + per < nblk ? b0 + per : nblk;
    int b = b0;
    for (; b + 1 < b1; b += 2) {
      t0 += part[(long long)b * C + c];
      t1 += part[(long long)(b + 1) * C + c];
    }
    if (b < b1) t0 += part[(long long)b * C + c];
  }
  red[threadIdx.x] = t0 + t1;
  __syncthreads();
  if (strip == 0 && c < C) {
    float u = 0.f;
    for (int g = 0; g < strips; ++g) u += red[g * cpb + threadIdx.x];
    out[c] = accumulate ? out[c] + u : u;
  }
}

// dst[r, 0:W] = scale(c) * src[r, c], scale = s_head for c < C else s_tail, columns at or past w_src read as zero
// (src row pitch ld_src): the row-wise pieces of the CLS merge / seed gather backward — "token 0 of the row times 1/2",
// "the CLS slice of a [B, D] gradient padded to a [B, S*C] row", "a column block made contiguous" — in one pass each
template <typename T, int VEC>
__global__ void k_row_head_scale(const T* __restrict__ src, long long ld_src, int w_src, T* __restrict__ dst, long long B,
                                 int W, int C, float s_head, float s_tail) {
  const int vpr = W / VEC;
  long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  const long long total = B * vpr, stride = (long long)gridDim.x * blockDim.x;
  for (; i < total; i += stride) {
    const long long r = i / vpr;
    const int c = (int)(i - r * vpr) * VEC;
    float v[VEC];
    if (c < w_src) {
      loadv<T, VEC>(src + r * ld_src + c, v);
      const float sc = c < C ? s_head : s_tail;
#pragma unroll
      for (int j = 0; j < VEC; ++j) v[j] *= sc;
    } else {
#pragma unroll
      for (int j = 0; j < VEC; ++j) v[j] = 0.f;
    }
    storev<T, VEC>(dst + r * (long long)W + c, v);
  }
}

// CLS-token merge of the fused layer (fused.py:259-260): out = x_tab with token 0 <- (x_tab[:,0] + xf[:, :C]) / 2
template <typename T, int VEC>
__global__ void k_cls_merge_fwd(const T* __restrict__ xtab, const T* __restrict__ xf, T* __restrict__ out, long long B,
                                int S, int C, int D) {
  const int vpr = S * C / VEC;
  long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  long long stride = (long long)gridDim.x * blockDim.x;
  long long total = B * vpr;
  for (; i < total; i += stride) {
    long long r = i / vpr;
    int c = (int)(i % vpr) * VEC;
    float v[VEC];
    loadv<T, VEC>(xtab + r * S * C + c, v);
    if (c < C) {
      float t[VEC];
      loadv<T, VEC>(xf + r * D + c, t);
#pragma unroll
      for (int j = 0; j < VEC; ++j) v[j] = 0.5f * (v[j] + t[j]);
    }
    storev<T, VEC>(out + r * S * C + c, v);
  }
}


// ------------------------------------------------------------------ tail LayerNorm + norm2 backward in one pass
// The encoder layer ends  x2 = LN2(z2),  out = alpha*x + beta_c*LN_t(x2)  (encoder_layer.py).  Run as two k_ln_bwd
// launches the gradient d_x2 [rows, C] is written by the first and read back by the second; here it stays in
// registers:  dout -> (d_x2) -> d_x1 = dLN2/dz2,  d_y2 = d_x1 * dropout mask,  plus dres = alpha*dout and the five
// parameter-gradient column sums (dgamma_t, dbeta_t, dgamma_2, dbeta_2, dbias_2) through per-block partials.
// Reads x2, z2, dout (+ two stats pairs); writes dres, d_x1, d_y2: 6 row streams instead of 8.
template <typename T, int VEC, int VPL>
__global__ void __launch_bounds__(LN_BLOCK) k_ln_tail_ln_bwd(const T* __restrict__ x2, const T* __restrict__ z2,
                                                              const float* __restrict__ gamma_t,
                                                              const float* __restrict__ stats3,
                                                              const float* __restrict__ gamma_2,
                                                              const float* __restrict__ stats2,
                                                              const T* __restrict__ dout, T* __restrict__ dres,
                                                              T* __restrict__ d_x1, T* __restrict__ d_y2,
                                                              float* __restrict__ partials, long long M, int C, int lpr,
                                                              float alpha, float beta_c, unsigned thresh, float inv_keep,
                                                              unsigned long long seed, unsigned rstream) {
  seed = live_seed(seed);
  extern __shared__ float red[];  // [groups][5][C]
  const int groups = LN_BLOCK / lpr;
  const int gl = threadIdx.x % lpr, gi = threadIdx.x / lpr;
  long long row = (long long)blockIdx.x * groups + gi;
  const long long rstride = (long long)gridDim.x * groups;
  const int nvec = C / VEC;
  float agt[VPL][VEC], abt[VPL][VEC], ag2[VPL][VEC], ab2[VPL][VEC], abs_[VPL][VEC];   // the five column sums
  float gtr[VPL][VEC], g2r[VPL][VEC];                                                  // gamma_t, gamma_2 in registers
#pragma unroll
  for (int k = 0; k < VPL; ++k) {
    const int vi = gl + k * lpr;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      agt[k][j] = abt[k][j] = ag2[k][j] = ab2[k][j] = abs_[k][j] = 0.f;
      const bool ok = vi < nvec;
      gtr[k][j] = ok ? gamma_t[vi * VEC + j] : 0.f;
      g2r[k][j] = ok ? gamma_2[vi * VEC + j] : 0.f;
    }
  }
  for (; row < M; row += rstride) {
    const float mu3 = stats3[2 * row], rstd3 = stats3[2 * row + 1];
    const float mu2 = stats2[2 * row], rstd2 = stats2[2 * row + 1];
    float xh[VPL][VEC], gx[VPL][VEC], zh[VPL][VEC];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < VPL; ++k) {
      const int vi = gl + k * lpr;
      if (vi < nvec) {
        const int c = vi * VEC;
        float x[VEC], g[VEC], z[VEC];
        loadv<T, VEC>(x2 + row * C + c, x);
        loadv<T, VEC>(dout + row * C + c, g);
        loadv<T, VEC>(z2 + row * C + c, z);          // needed after the first reduction: in flight with the others
        if (dres) {
          float r[VEC];
#pragma unroll
          for (int j = 0; j < VEC; ++j) r[j] = alpha * g[j];
          storev<T, VEC>(dres + row * C + c, r);
        }
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          const float gj = g[j] * beta_c;
          xh[k][j] = (x[j] - mu3) * rstd3;
          zh[k][j] = (z[j] - mu2) * rstd2;
          agt[k][j] += gj * xh[k][j];
          abt[k][j] += gj;
          gx[k][j] = gj * gtr[k][j];
          s1 += gx[k][j];
          s2 += gx[k][j] * xh[k][j];
        }
      }
    }
    for (int o = lpr >> 1; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
    const float m1 = s1 / (float)C, m2 = s2 / (float)C;
    float t1 = 0.f, t2 = 0.f;
#pragma unroll
    for (int k = 0; k < VPL; ++k) {
      const int vi = gl + k * lpr;
      if (vi < nvec) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          const float dx2 = rstd3 * (gx[k][j] - m1 - xh[k][j] * m2);     // gradient of x2, never leaves the registers
          ag2[k][j] += dx2 * zh[k][j];
          ab2[k][j] += dx2;
          gx[k][j] = dx2 * g2r[k][j];
          t1 += gx[k][j];
          t2 += gx[k][j] * zh[k][j];
        }
      }
    }
    for (int o = lpr >> 1; o > 0; o >>= 1) { t1 += __shfl_xor(t1, o, 64); t2 += __shfl_xor(t2, o, 64); }
    const float n1 = t1 / (float)C, n2 = t2 / (float)C;
#pragma unroll
    for (int k = 0; k < VPL; ++k) {
      const int vi = gl + k * lpr;
      if (vi < nvec) {
        const int c = vi * VEC;
        float d[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) d[j] = rstd2 * (gx[k][j] - n1 - zh[k][j] * n2);
        storev<T, VEC>(d_x1 + row * C + c, d);
        const unsigned long long e0 = (unsigned long long)(row * C + c);
        const unsigned key = rng_key(seed, rstream, (unsigned)(e0 >> 32));
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          float m = drop_scale_key(key, (unsigned)e0 + j, thresh, inv_keep);
          m = thresh ? m : 1.f;
          d[j] *= m;
          abs_[k][j] += d[j];
        }
        storev<T, VEC>(d_y2 + row * C + c, d);
      }
    }
  }
  // block reduction of the five column sums (fixed order -> deterministic)
#pragma unroll
  for (int k = 0; k < VPL; ++k) {
    const int vi = gl + k * lpr;
    if (vi < nvec) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const int c = vi * VEC + j;
        red[(gi * 5 + 0) * C + c] = agt[k][j];
        red[(gi * 5 + 1) * C + c] = abt[k][j];
        red[(gi * 5 + 2) * C + c] = ag2[k][j];
        red[(gi * 5 + 3) * C + c] = ab2[k][j];
        red[(gi * 5 + 4) * C + c] = abs_[k][j];
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 5 * C; i += LN_BLOCK) {
    float t = 0.f;
    for (int g2 = 0; g2 < groups; ++g2) t += red[g2 * 5 * C + i];
    partials[(long long)blockIdx.x * 5 * C + i] = t;
  }
}

struct AccPtrs5 { float* o[5]; };
// out_s[c] += sum over blocks of partials[blk][s][c] for the five column sums (NULL targets are skipped)
__global__ void __launch_bounds__(1024) k_reduce_partials_acc5(const float* __restrict__ partials, int nblk, int C,
                                                                AccPtrs5 acc) {
  __shared__ float red[RP_STRIPS][64];
  const int width = 5 * C;
  const int lane = threadIdx.x & 63, strip = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + lane;
  float t = 0.f;
  if (col < width) {
    const int per = (nblk + RP_STRIPS - 1) / RP_STRIPS, b0 = strip * per, b1 = b0 + per < nblk ? b0 + per : nblk;
    float a[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) a[u] = 0.f;
    int b = b0;
    for (; b + 7 < b1; b += 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] += partials[(long long)(b + u) * width + col];
    }
    for (; b < b1; ++b) a[0] += partials[(long long)b * width + col];
    t = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  }
  red[strip][lane] = t;
  __syncthreads();
  if (strip == 0 && col < width) {
    float r = 0.f;
#pragma unroll
    for (int u = 0; u < RP_STRIPS; ++u) r += red[u][lane];
    const int seg = col / C, c = col - seg * C;
    float* o = acc.o[seg];
    if (o) o[c] += r;
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

static int ln_geometry(int C, int vec, int& lpr, int& vpl) {
  int nvec = C / vec;
  lpr = 64;
  while (lpr > 1 && (nvec % lpr != 0)) lpr >>= 1;
  // prefer at most 64 lanes; odd factors (e.g. 48 = 16*3) go to VPL
  vpl = nvec / lpr;
  while (vpl > 4 && lpr < 64) { lpr <<= 1; vpl = (nvec + lpr - 1) / lpr; }
  return vpl <= 4 ? 0 : 1;
}

extern "C" int tg_ln_partials_floats(int64_t M, int32_t C) { return 2048 * 3 * C; }

#define LN_LAUNCH(KERN, VPLV, ...)                                                      \
  switch (VPLV) {                                                                       \
    case 1: hipLaunchKernelGGL((KERN<T, VEC, 1>), __VA_ARGS__); break;                  \
    case 2: hipLaunchKernelGGL((KERN<T, VEC, 2>), __VA_ARGS__); break;                  \
    case 3: hipLaunchKernelGGL((KERN<T, VEC, 3>), __VA_ARGS__); break;                  \
    default: hipLaunchKernelGGL((KERN<T, VEC, 4>), __VA_ARGS__); break;                 \
  }

extern "C" int tg_ln_fwd(const void* a, const void* b, const float* bias_b, const float* gamma, const float* beta,
                         const void* res, void* out, float* stats, int64_t M, int32_t C, float eps, float alpha,
                         float beta_c, float p_drop, uint64_t seed, uint32_t rstream, int32_t dt, void* stream) {
  TG_CHECK(C % 8 == 0 && C <= 2048, "tg_ln_fwd: C must be a multiple of 8 and <= 2048 (C=%d)", C);
  if (M == 0) return 0;
  unsigned thresh = (b && p_drop > 0.f) ? drop_threshold(p_drop) : 0u;
  float inv_keep = p_drop < 1.f ? 1.f / (1.f - p_drop) : 0.f;
  DISPATCH_T(dt, {
    int lpr, vpl;
    TG_CHECK(ln_geometry(C, VEC, lpr, vpl) == 0, "tg_ln_fwd: unsupported width C=%d", C);
    int groups = LN_BLOCK / lpr;
    int grid = grid_cap(ceil_div(M, groups), 256 * 16);
    LN_LAUNCH(k_ln_fwd, vpl, dim3(grid), dim3(LN_BLOCK), 0, (hipStream_t)stream, (const T*)a, (const T*)b, bias_b,
              gamma, beta, (const T*)res, (T*)out, stats, (long long)M, C, lpr, eps, alpha, beta_c, thresh, inv_keep,
              (unsigned long long)seed, rstream);
  })
  TG_LAUNCH_CHECK();
  return 0;
}

// dparams: float[3*C] = (dgamma, dbeta, dbias_b) — written (not accumulated)
extern "C" int tg_ln_bwd(const void* a, const void* b, const float* bias_b, const float* gamma, const float* stats,
                         const void* dout, void* da, void* db, void* dres, float* dparams, float* partials, int64_t M,
                         int32_t C, float alpha, float beta_c, float p_drop, uint64_t seed, uint32_t rstream,
                         int32_t accum_da, float* acc_gamma, float* acc_beta, float* acc_bias, int32_t dt,
                         void* stream) {
  TG_CHECK(C % 8 == 0 && C <= 2048, "tg_ln_bwd: C must be a multiple of 8 and <= 2048 (C=%d)", C);
  hipStream_t st = (hipStream_t)stream;
  const bool acc_params = acc_gamma != nullptr || acc_beta != nullptr || acc_bias != nullptr;
  if (M == 0) {
    if (!acc_params) zero_async(dparams, 3 * (size_t)C * sizeof(float), st);
    return 0;
  }
  unsigned thresh = ((b || db) && p_drop > 0.f) ? drop_threshold(p_drop) : 0u;
  float inv_keep = p_drop < 1.f ? 1.f / (1.f - p_drop) : 0.f;
  int grid = 1;
  DISPATCH_T(dt, {
    int lpr, vpl;
    TG_CHECK(ln_geometry(C, VEC, lpr, vpl) == 0, "tg_ln_bwd: unsupported width C=%d", C);
    int groups = LN_BLOCK / lpr;
    grid = grid_cap(ceil_div(M, groups), 2048);
    size_t shm = (size_t)groups * 3 * C * sizeof(float);
    TG_CHECK(shm <= 160 * 1024, "tg_ln_bwd: LDS %zu too large", shm);
    LN_LAUNCH(k_ln_bwd, vpl, dim3(grid), dim3(LN_BLOCK), shm, st, (const T*)a, (const T*)b, bias_b, gamma, stats,
              (const T*)dout, (T*)da, (T*)db, (T*)dres, partials, (long long)M, C, lpr, alpha, beta_c, thresh, inv_keep,
              (unsigned long long)seed, rstream, (int)accum_da);
  })
  if (acc_params)
    hipLaunchKernelGGL(k_reduce_partials_acc3, dim3(ceil_div(3 * C, 64)), dim3(1024), 0, st, partials, grid, C, acc_gamma,
                       acc_beta, acc_bias);
  else
    hipLaunchKernelGGL(k_reduce_partials, dim3(ceil_div(3 * C, 64)), dim3(1024), 0, st, partials, grid, 3 * C, dparams);
  TG_LAUNCH_CHECK();
  return 0;
}

// Tail LayerNorm backward + norm2 backward ("z mode") of the encoder layer in one pass (k_ln_tail_ln_bwd).
// dparams: float[5*C] = (dgamma_t, dbeta_t, dgamma_2, dbeta_2, dbias_2), written — or, when acc != NULL (five [C]
// pointers, NULL entries skipped), added into the parameters' gradient buffers instead.
// partials: 2048 * 5 * C floats.  dres may be NULL (alpha == 0).
extern "C" int tg_ln_tail_ln_bwd(const void* x2, const void* z2, const float* gamma_t, const float* stats3,
                                 const float* gamma_2, const float* stats2, const void* dout, void* dres, void* d_x1,
                                 void* d_y2, float* dparams, float* partials, int64_t M, int32_t C, float alpha,
                                 float beta_c, float p_drop, uint64_t seed, uint32_t rstream, float* const* acc,
                                 int32_t dt, void* stream) {
  TG_CHECK(C % 8 == 0 && C <= 2048, "tg_ln_tail_ln_bwd: C must be a multiple of 8 and <= 2048 (C=%d)", C);
  TG_CHECK(x2 && z2 && gamma_t && stats3 && gamma_2 && stats2 && dout && d_x1 && d_y2 && partials && (dparams || acc),
           "tg_ln_tail_ln_bwd: null argument");
  hipStream_t st = (hipStream_t)stream;
  if (M == 0) {
    if (!acc) zero_async(dparams, 5 * (size_t)C * sizeof(float), st);
    return 0;
  }
  unsigned thresh = p_drop > 0.f ? drop_threshold(p_drop) : 0u;
  float inv_keep = p_drop < 1.f ? 1.f / (1.f - p_drop) : 0.f;
  int grid = 1;
  DISPATCH_T(dt, {
    int lpr, vpl;
    TG_CHECK(ln_geometry(C, VEC, lpr, vpl) == 0, "tg_ln_tail_ln_bwd: unsupported width C=%d", C);
    int groups = LN_BLOCK / lpr;
    grid = grid_cap(ceil_div(M, groups), 2048);
    size_t shm = (size_t)groups * 5 * C * sizeof(float);
    TG_CHECK(shm <= 160 * 1024, "tg_ln_tail_ln_bwd: LDS %zu too large", shm);
    LN_LAUNCH(k_ln_tail_ln_bwd, vpl, dim3(grid), dim3(LN_BLOCK), shm, st, (const T*)x2, (const T*)z2, gamma_t, stats3,
              gamma_2, stats2, (const T*)dout, (T*)dres, (T*)d_x1, (T*)d_y2, partials, (long long)M, C, lpr, alpha,
              beta_c, thresh, inv_keep, (unsigned long long)seed, rstream);
  })
  if (acc) {
    AccPtrs5 a;
    for (int i = 0; i < 5; ++i) a.o[i] = acc[i];
    hipLaunchKernelGGL(k_reduce_partials_acc5, dim3(ceil_div(5 * C, 64)), dim3(1024), 0, st, partials, grid, C, a);
  } else {
    hipLaunchKernelGGL(k_reduce_partials, dim3(ceil_div(5 * C, 64)), dim3(1024), 0, st, partials, grid, 5 * C, dparams);
  }
  TG_LAUNCH_CHECK();
  return 0;
}

extern "C" int tg_bn_partials_floats(int64_t N, int32_t F) { return 513 * 2 * F; }

// training=1: batch statistics (and running-stat update when running_* given); training=0: running statistics.
// mean/rstd [F] are outputs kept for the backward.
extern "C" int tg_bn_act_res_fwd(const void* x, const void* res, const float* gamma, const float* beta,
                                 float* running_mean, float* running_var, float* mean, float* rstd, void* out,
                                 float* partials, int64_t N, int32_t F, int32_t training, float momentum, float eps,
                                 int32_t relu, float alpha, float beta_c, int64_t n_stat, int32_t phase,
                                 const int32_t* row_limit, int32_t dt, void* stream) {
  TG_CHECK(F % 8 == 0 && F <= 1024 && N > 0, "tg_bn_act_res_fwd: bad F=%d N=%lld", F, (long long)N);
  TG_CHECK(phase >= 0 && phase <= 2 && (phase == 0 || training), "tg_bn_act_res_fwd: bad phase %d", phase);
  hipStream_t st = (hipStream_t)stream;
  if (n_stat <= 0 || phase == 0) n_stat = N;
  TG_CHECK(row_limit == nullptr || (phase == 0 && training), "tg_bn_act_res_fwd: %s", "row_limit needs training, phase 0");
  const int* nlim = (const int*)row_limit;      // device-side row count of a padded batch (not combined with the synchronised phases)
  DISPATCH_T(dt, {
    TG_CHECK(256 % (F / VEC) == 0, "tg_bn_act_res_fwd: F/VEC must divide 256 (F=%d)", F);
    float* sums = partials + (size_t)512 * 2 * F;      // (sum x, sum x^2): the vector a synchronised BN all-reduces
    if (training) {
      if (phase != 2) {
        int groups = 256 / (F / VEC);
        int grid = grid_cap(ceil_div(N, (long long)groups * 4), 512);
        size_t shm = (size_t)groups * 2 * F * sizeof(float);
        hipLaunchKernelGGL((k_col_stats<T, VEC>), dim3(grid), dim3(256), shm, st, (const T*)x, (const T*)nullptr,
                           (const float*)nullptr, (const float*)nullptr, (const T*)nullptr, partials, (long long)N, F, 0, nlim);
        hipLaunchKernelGGL(k_reduce_partials, dim3(ceil_div(2 * F, 64)), dim3(1024), 0, st, partials, grid, 2 * F, sums);
      }
      if (phase == 1) {
        TG_LAUNCH_CHECK();
        return 0;
      }
      hipLaunchKernelGGL(k_bn_finalize, dim3(ceil_div(F, 256)), dim3(256), 0, st, sums, 1, (long long)n_stat, F, eps,
                         momentum, mean, rstd, running_mean, running_var, nlim);
    } else {
      hipLaunchKernelGGL(k_bn_eval_stats, dim3(ceil_div(F, 256)), dim3(256), 0, st, running_mean, running_var, F, eps,
                         mean, rstd);
    }
    long long total = (long long)N * (F / VEC);
    hipLaunchKernelGGL((k_bn_apply<T, VEC>), dim3(grid_cap(ceil_div(total, 256))), dim3(256), 0, st, (const T*)x,
                       (const T*)res, mean, rstd, gamma, beta, (T*)out, (long long)N, F, relu, alpha, beta_c);
  })
  TG_LAUNCH_CHECK();
  return 0;
}

// dparams: float[2*F] = (dbeta_sum = sum dz, dgamma = sum dz*xhat)
extern "C" int tg_bn_act_res_bwd(const void* x, const void* dout, const float* gamma, const float* beta,
                                 const float* mean, const float* rstd, void* dx, void* dres, float* dparams,
                                 float* partials, int64_t N, int32_t F, int32_t training, int32_t relu, float alpha,
                                 float beta_c, int64_t n_stat, int32_t phase, const int32_t* row_limit, int32_t dt,
                                 void* stream) {
  TG_CHECK(F % 8 == 0 && F <= 1024 && N > 0, "tg_bn_act_res_bwd: bad F=%d N=%lld", F, (long long)N);
  TG_CHECK(phase >= 0 && phase <= 2, "tg_bn_act_res_bwd: bad phase %d", phase);
  hipStream_t st = (hipStream_t)stream;
  if (n_stat <= 0 || phase == 0) n_stat = N;
  TG_CHECK(row_limit == nullptr || (phase == 0 && training), "tg_bn_act_res_bwd: %s", "row_limit needs training, phase 0");
  const int* nlim = (const int*)row_limit;
  DISPATCH_T(dt, {
    TG_CHECK(256 % (F / VEC) == 0, "tg_bn_act_res_bwd: F/VEC must divide 256 (F=%d)", F);
    if (phase != 2) {
      int groups = 256 / (F / VEC);
      int grid = grid_cap(ceil_div(N, (long long)groups * 4), 512);
      size_t shm = (size_t)groups * 2 * F * sizeof(float);
      hipLaunchKernelGGL((k_bn_bwd_stats<T, VEC>), dim3(grid), dim3(256), shm, st, (const T*)x, (const T*)dout, mean,
                         rstd, gamma, beta, partials, (long long)N, F, relu, beta_c, nlim);
      hipLaunchKernelGGL(k_reduce_partials, dim3(ceil_div(2 * F, 64)), dim3(1024), 0, st, partials, grid, 2 * F, dparams);
    }
    if (phase != 1) {    // the statistics terms divide by the number of rows behind dparams (all ranks when synchronised)
      long long total = (long long)N * (F / VEC);
      hipLaunchKernelGGL((k_bn_bwd_apply<T, VEC>), dim3(grid_cap(ceil_div(total, 256))), dim3(256), 0, st, (const T*)x,
                         (const T*)dout, mean, rstd, gamma, beta, dparams, (T*)dx, (T*)dres, (long long)N,
                         (long long)n_stat, F, relu, training, alpha, beta_c, nlim);
    }
  })
  TG_LAUNCH_CHECK();
  return 0;
}

extern "C" int tg_act_dropout_fwd(const void* x, void* y, int64_t n, int32_t act, float p_drop, uint64_t seed,
                                  uint32_t rstream, int32_t dt, void* stream) {
  if (n == 0) return 0;
  unsigned thresh = p_drop > 0.f ? drop_threshold(p_drop) : 0u;
  float inv_keep = p_drop < 1.f ? 1.f / (1.f - p_drop) : 0.f;
  DISPATCH_T(dt, {
    hipLaunchKernelGGL((k_act_dropout_fwd<T, VEC>), dim3(grid_full(ceil_div(ceil_div(n, VEC), 256))), dim3(256), 0,
                       (hipStream_t)stream, (const T*)x, (T*)y, (long long)n, act, thresh, inv_keep,
                       (unsigned long long)seed, rstream);
  })
  TG_LAUNCH_CHECK();
  return 0;
}

extern "C" int tg_act_dropout_bwd(const void* x, const void* dy, void* dx, int64_t n, int32_t act, float p_drop,
                                  uint64_t seed, uint32_t rstream, int32_t dt, void* stream) {
  if (n == 0) return 0;
  unsigned thresh = p_drop > 0.f ? drop_threshold(p_drop) : 0u;
  float inv_keep = p_drop < 1.f ? 1.f / (1.f - p_drop) : 0.f;
  DISPATCH_T(dt, {
    hipLaunchKernelGGL((k_act_dropout_bwd<T, VEC>), dim3(grid_full(ceil_div(ceil_div(n, VEC), 256))), dim3(256), 0,
                       (hipStream_t)stream, (const T*)x, (const T*)dy, (T*)dx, (long long)n, act, thresh, inv_keep,
                       (unsigned long long)seed, rstream);
  })
  TG_LAUNCH_CHECK();
  return 0;
}

extern "C" int tg_axpby(const void* a, const void* b, void* y, int64_t n, float alpha, float beta, int32_t dt,
                        void* stream) {
  TG_CHECK(n % 8 == 0, "tg_axpby: n must be a multiple of 8 (n=%lld)", (long long)n);
  if (n == 0) return 0;
  DISPATCH_T(dt, {
    hipLaunchKernelGGL((k_axpby<T, VEC>), dim3(grid_full(ceil_div(n / VEC, 256))), dim3(256), 0, (hipStream_t)stream,
                       (const T*)a, (const T*)b, (T*)y, (long long)n, alpha, beta);
  })
  TG_LAUNCH_CHECK();
  return 0;
}

extern "C" int tg_axpby2(const void* a, const void* b, void* y1, void* y2, int64_t n, float alpha, float beta, float gamma,
                         int32_t dt, void* stream) {
  TG_CHECK(n % 8 == 0, "tg_axpby2: n must be a multiple of 8 (n=%lld)", (long long)n);
  TG_CHECK(b && y1 && y2, "tg_axpby2: null operand");
  if (n == 0) return 0;
  DISPATCH_T(dt, {
    hipLaunchKernelGGL((k_axpby2<T, VEC>), dim3(grid_full(ceil_div(n / VEC, 256))), dim3(256), 0, (hipStream_t)stream,
                       (const T*)a, (const T*)b, (T*)y1, (T*)y2, (long long)n, alpha, beta, gamma);
  })
  TG_LAUNCH_CHECK();
  return 0;
}

static int col_sum_blocks(int64_t R, int rows_per_pass) {
  const int64_t want = ceil_div(R, (int64_t)rows_per_pass * 8);      // >= 8 passes per block
  return (int)(want < 1 ? 1 : want > 512 ? 512 : want);
}
extern "C" int64_t tg_col_sum_workspace_floats(int64_t R, int32_t C) { return (int64_t)1024 * C; }

// out[C] (fp32) (+)= sum over r of x[r*ld + 0..C)  — column sums of a strided matrix, in a fixed order
extern "C" int tg_col_sum(const void* x, int64_t R, int32_t C, int64_t ld, float* out, float* workspace,
                          int32_t accumulate, int32_t dt, void* stream) {
  TG_CHECK(x && out && workspace, "tg_col_sum: null operand");
  TG_CHECK(C > 0 && C % 8 == 0 && C <= 2048 && ld >= C && ld % 8 == 0 && R >= 0, "tg_col_sum: bad shape (R=%lld C=%d ld=%lld)",
           (long long)R, C, (long long)ld);
  TG_CHECK((reinterpret_cast<uintptr_t>(x) & 15) == 0, "tg_col_sum: x must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  int nblk = 0;
  if (R > 0) {
    DISPATCH_T(dt, {
      TG_CHECK(C / VEC <= 256, "tg_col_sum: C too wide");
      const int rpp = 256 / (C / VEC);
      nblk = col_sum_blocks(R, rpp);
      const long long chunk = ceil_div(R, (int64_t)nblk);
      hipLaunchKernelGGL((k_col_sum_part<T, VEC>), dim3(nblk), dim3(256), 0, st, (const T*)x, (long long)R, C, (long long)ld,
                         chunk, workspace);
      TG_LAUNCH_CHECK();
    })
  }
  hipLaunchKernelGGL(k_col_sum_reduce, dim3(ceil_div(C, 8)), dim3(256), 0, st, workspace, nblk, C, out, accumulate);
  TG_LAUNCH_CHECK();
  return 0;
}

extern "C" int tg_row_head_scale(const void* src, int64_t ld_src, int32_t w_src, void* dst, int64_t B, int32_t W, int32_t C,
                                 float s_head, float s_tail, int32_t dt, void* stream) {
  TG_CHECK(src && dst, "tg_row_head_scale: null operand");
  TG_CHECK(W > 0 && W % 8 == 0 && C % 8 == 0 && w_src % 8 == 0 && ld_src % 8 == 0 && w_src >= 0 && w_src <= ld_src && C >= 0,
           "tg_row_head_scale: widths must be multiples of 8 (W=%d C=%d w_src=%d ld=%lld)", W, C, w_src, (long long)ld_src);
  TG_CHECK(((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0,
           "tg_row_head_scale: operands must be 16-byte aligned");
  if (B == 0) return 0;
  DISPATCH_T(dt, {
    const long long total = (long long)B * (W / VEC);
    hipLaunchKernelGGL((k_row_head_scale<T, VEC>), dim3(grid_cap(ceil_div(total, 256))), dim3(256), 0, (hipStream_t)stream,
                       (const T*)src, (long long)ld_src, w_src, (T*)dst, (long long)B, W, C, s_head, s_tail);
  })
  TG_LAUNCH_CHECK();
  return 0;
}

extern "C" int tg_cls_merge_fwd(const void* xtab, const void* xf, void* out, int64_t B, int32_t S, int32_t C,
                                int32_t D, int32_t dt, void* stream) {
  TG_CHECK(C % 8 == 0 && D % 8 == 0, "tg_cls_merge_fwd: C, D must be multiples of 8");
  if (B == 0) return 0;
  DISPATCH_T(dt, {
    long long total = (long long)B * (S * C / VEC);
    hipLaunchKernelGGL((k_cls_merge_fwd<T, VEC>), dim3(grid_cap(ceil_div(total, 256))), dim3(256), 0,
                       (hipStream_t)stream, (const T*)xtab, (const T*)xf, (T*)out, (long long)B, S, C, D);
  })
  TG_LAUNCH_CHECK();
  return 0;
}

