// The two weight folds of PNAConv (torch_geometric 2.5.3 PNAConv with towers=1, pre_layers=post_layers=1, as built at
// src/nn/models/fused.py:200-207): Linear maps with nothing between them are multiplied on the (tiny, fp32) weights
// once per step so that no edge- or node-scale intermediate exists for them:
//     w_msg = [P[:, :2F] | P[:, 2F:] We],   b_msg = pb + P[:, 2F:] be      (edge_encoder into pre_nn)
//     w_eff = Lw Qw,  b_eff = Lw qb + lb,   w_x = w_eff[:, :F]             (lin into post_nn)
//     w_st[s*F + f, kk*F + i] = w_eff[f, F + (s*4 + order[kk])*F + i]      ([3F,4F]: scaler s, the aggregation kernel's
//                                                                          mean|max|min|std block order)
// One launch writes every layout the step needs (fp32 for autograd, bf16 row-major and transposed for the GEMMs, the
// scaled-projection packs of tg_gemm_nt_scaled_bf16), one more pair of launches turns the five weight gradients back
// into the eight parameter gradients, accumulated in place.  These replace ~45 five-microsecond library launches per
// convolution and step (small GEMMs, cat, permute, dtype copies, addmm_/addmv_).  All of it is fp32 dot products of
// length F or 13F over matrices that live in L2: not a roofline matter, a launch-count one.
#include "common.hpp"
#include "../../include/tabgnn_hip.h"

namespace tg {

struct FoldDims {
  int F, Fe;          // node width, raw edge width (edge_encoder: Fe -> F)
  int inv[4];         // inv[a] = kernel block kk with order[kk] == a
};

// All products here are [F x F] x [F x (F..13F)] or [F x 13F] x [13F x F] in fp32: each 32 x 32 output tile is one
// 256-thread workgroup (2 x 2 outputs per thread), the inner dimension walks in 32-wide slices through LDS with the next
// slice's global loads in flight under the FMAs.  Operands are described by element accessors (so a transposed or a
// column-permuted operand costs nothing extra), `AK` / `BK` say which index of A / B is contiguous in memory so that the
// slice loads stay coalesced.
constexpr int FT = 32;
constexpr int LW_CHUNK = 128;     // inner-index chunk of the dLw product per workgroup

template <bool AK /*A contiguous along k*/, bool BK /*B contiguous along k*/, typename FA, typename FBm, typename FE>
__device__ __forceinline__ void fold_tile(int m0, int n0, int kbeg, int K, FA a, FBm b, FE epi, float (*sa)[FT + 1],
                                          float (*sb)[FT + 1]) {      // inner index range [kbeg, K)
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  float ra[4], rb[4];
  auto load = [&](int k0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = tid + 256 * q, lo = e & 31, hi = e >> 5;
      const int am = AK ? hi : lo, ak = AK ? lo : hi;
      const int bn = BK ? hi : lo, bk = BK ? lo : hi;
      ra[q] = k0 + ak < K ? a(m0 + am, k0 + ak) : 0.f;
      rb[q] = k0 + bk < K ? b(k0 + bk, n0 + bn) : 0.f;
    }
  };
  float c00 = 0.f, c01 = 0.f, c10 = 0.f, c11 = 0.f;
  load(kbeg);
  for (int k0 = kbeg; k0 < K; k0 += FT) {
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = tid + 256 * q, lo = e & 31, hi = e >> 5;
      sa[AK ? lo : hi][AK ? hi : lo] = ra[q];           // sa[k][m]
      sb[BK ? lo : hi][BK ? hi : lo] = rb[q];           // sb[k][n]
    }
    __syncthreads();
    if (k0 + FT < K) load(k0 + FT);
#pragma unroll
    for (int kk = 0; kk < FT; ++kk) {
      const float a0 = sa[kk][ty], a1 = sa[kk][ty + 16], b0 = sb[kk][tx], b1 = sb[kk][tx + 16];
      c00 += a0 * b0; c01 += a0 * b1; c10 += a1 * b0; c11 += a1 * b1;
    }
  }
  epi(m0 + ty, n0 + tx, c00); epi(m0 + ty, n0 + tx + 16, c01);
  epi(m0 + ty + 16, n0 + tx, c10); epi(m0 + ty + 16, n0 + tx + 16, c11);
}

__global__ void __launch_bounds__(256) k_pna_fold_fwd(tg_fold_params p, tg_fold_out o, FoldDims d) {
  __shared__ float sa[FT][FT + 1], sb[FT][FT + 1];
  const int F = d.F, Fe = d.Fe, WM = 2 * F + Fe, K = 4 * F;
  const int tf = F / FT, nb1 = tf * (Fe / FT), nb2 = tf * (13 * F / FT);
  int blk = blockIdx.x;
  unsigned short* w_msg_lp = reinterpret_cast<unsigned short*>(o.w_msg_lp);
  unsigned short* w_msg_lp_t = reinterpret_cast<unsigned short*>(o.w_msg_lp_t);
  auto put_msg = [&](int f, int j, float v) {
    o.w_msg[(long long)f * WM + j] = v;
    if (w_msg_lp) {
      w_msg_lp[(long long)f * WM + j] = f2bf(v);
      w_msg_lp_t[(long long)j * F + f] = f2bf(v);
    }
  };
  if (blk < nb1) {                                 // ---- w_msg[:, 2F:] = P3 We
    const int m0 = (blk % tf) * FT, n0 = (blk / tf) * FT;
    fold_tile<true, false>(m0, n0, 0, F, [&](int f, int k) { return p.P[(long long)f * 3 * F + 2 * F + k]; },
                           [&](int k, int c) { return p.We[(long long)k * Fe + c]; },
                           [&](int f, int c, float v) { put_msg(f, 2 * F + c, v); }, sa, sb);
    return;
  }
  blk -= nb1;
  if (blk < nb2) {                                 // ---- w_eff = Lw Qw, scattered into w_x / w_st and their packs
    const int m0 = (blk % tf) * FT, n0 = (blk / tf) * FT;
    unsigned short* w_x_lp = reinterpret_cast<unsigned short*>(o.w_x_lp);
    unsigned short* w_x_lp_t = reinterpret_cast<unsigned short*>(o.w_x_lp_t);
    unsigned short* w_cat = reinterpret_cast<unsigned short*>(o.w_cat);
    unsigned short* wt_cat = reinterpret_cast<unsigned short*>(o.wt_cat);
    fold_tile<true, false>(m0, n0, 0, F, [&](int f, int k) { return p.Lw[(long long)f * F + k]; },
                           [&](int k, int j) { return p.Qw[(long long)k * 13 * F + j]; },
                           [&](int f, int j, float v) {
                             if (j < F) {
                               o.w_x[(long long)f * F + j] = v;
                               if (w_x_lp) {
                                 w_x_lp[(long long)f * F + j] = f2bf(v);
                                 w_x_lp_t[(long long)j * F + f] = f2bf(v);
                               }
                             } else {
                               const int jj = j - F, s = jj / K, a = (jj / F) & 3, ii = jj % F;
                               const int col = d.inv[a] * F + ii;
                               o.w_st[((long long)s * F + f) * K + col] = v;
                               if (w_cat) {
                                 w_cat[(long long)f * 3 * K + ((col >> 7) * 3 + s) * 128 + (col & 127)] = f2bf(v);
                                 wt_cat[(long long)col * 3 * F + s * F + f] = f2bf(v);
                               }
                             }
                           }, sa, sb);
    return;
  }
  blk -= nb2;
  // ---- the copied part of w_msg and the two folded biases: plain element threads
  const long long nA = (long long)F * 2 * F;
  long long t = blk * 256LL + threadIdx.x;
  if (t < nA) {
    const int f = (int)(t / (2 * F)), j = (int)(t % (2 * F));
    put_msg(f, j, p.P[(long long)f * 3 * F + j]);
    return;
  }
  t -= nA;
  if (t < 2 * F) {
    const int f = (int)(t % F);
    if (t < F) {
      const float* pr = p.P + (long long)f * 3 * F + 2 * F;
      float v = p.pb[f];
      for (int k = 0; k < F; ++k) v += pr[k] * p.be[k];
      o.b_msg[f] = v;
    } else {
      const float* lr = p.Lw + (long long)f * F;
      float v = p.lb[f];
      for (int k = 0; k < F; ++k) v += lr[k] * p.qb[k];
      o.b_eff[f] = v;
    }
  }
}

__device__ __forceinline__ void fold_put(float* out, long long i, float v, bool acc) {
  if (out) out[i] = acc ? out[i] + v : v;
}

// gradient of w_eff at (f, j) from its two carriers (either may be absent = zero)
__device__ __forceinline__ float fold_dweff(const tg_fold_grads& g, const FoldDims& d, int f, int j) {
  const int F = d.F;
  if (j < F) return g.dw_x ? g.dw_x[(long long)f * F + j] : 0.f;
  if (!g.dw_st) return 0.f;
  const int jj = j - F, s = jj / (4 * F), a = (jj / F) & 3, i = jj % F;
  return g.dw_st[((long long)s * F + f) * 4 * F + d.inv[a] * F + i];
}

__global__ void __launch_bounds__(256) k_pna_fold_bwd(tg_fold_params p, tg_fold_grads g, tg_fold_dparams o, FoldDims d) {
  __shared__ float sa[FT][FT + 1], sb[FT][FT + 1];
  const int F = d.F, Fe = d.Fe, WM = 2 * F + Fe;
  const int tf = F / FT, nchunk = (13 * F + LW_CHUNK - 1) / LW_CHUNK;
  const int nb_lw = tf * tf * nchunk, nb_q = tf * (13 * F / FT), nb_p = tf * tf, nb_e = tf * (Fe / FT);
  int blk = blockIdx.x;
  if (blk < nb_lw) {          // ---- partials of dLw = dw_eff Qw^T: the 13F-long inner index is split into 128-wide chunks
    if (!o.dLw) return;       //      (one workgroup each; summed in chunk order by k_pna_fold_lw_sum: deterministic)
    const int chunk = blk / (tf * tf), tile = blk % (tf * tf);
    const int m0 = (tile % tf) * FT, n0 = (tile / tf) * FT;
    const int kend = (chunk + 1) * LW_CHUNK < 13 * F ? (chunk + 1) * LW_CHUNK : 13 * F;
    float* part = o.ws + (long long)chunk * F * F;
    fold_tile<true, true>(m0, n0, chunk * LW_CHUNK, kend, [&](int f, int j) { return fold_dweff(g, d, f, j); },
                          [&](int j, int k) { return p.Qw[(long long)k * 13 * F + j]; },
                          [&](int f, int k, float v) { part[(long long)f * F + k] = v; }, sa, sb);
    return;
  }
  blk -= nb_lw;
  if (blk < nb_q) {                                // ---- dQw = Lw^T dw_eff
    if (!o.dQw) return;
    const int m0 = (blk % tf) * FT, n0 = (blk / tf) * FT;
    fold_tile<false, false>(m0, n0, 0, F, [&](int k, int f) { return p.Lw[(long long)f * F + k]; },
                            [&](int f, int j) { return fold_dweff(g, d, f, j); },
                            [&](int k, int j, float v) { fold_put(o.dQw, (long long)k * 13 * F + j, v, o.accumulate & 16); },
                            sa, sb);
    return;
  }
  blk -= nb_q;
  if (blk < nb_p) {                                // ---- dP[:, 2F:] = d3 We^T + db_msg (x) be
    if (!o.dP) return;
    const int m0 = (blk % tf) * FT, n0 = (blk / tf) * FT;
    fold_tile<true, true>(m0, n0, 0, Fe, [&](int f, int k) { return g.dw_msg ? g.dw_msg[(long long)f * WM + 2 * F + k] : 0.f; },
                          [&](int k, int c) { return p.We[(long long)c * Fe + k]; },
                          [&](int f, int c, float v) {
                            if (g.db_msg) v += g.db_msg[f] * p.be[c];
                            fold_put(o.dP, (long long)f * 3 * F + 2 * F + c, v, o.accumulate & 1);
                          }, sa, sb);
    return;
  }
  blk -= nb_p;
  if (blk < nb_e) {                                // ---- dWe = P3^T d3
    if (!o.dWe) return;
    const int m0 = (blk % tf) * FT, n0 = (blk / tf) * FT;
    fold_tile<false, false>(m0, n0, 0, F, [&](int c, int f) { return p.P[(long long)f * 3 * F + 2 * F + c]; },
                            [&](int f, int k) { return g.dw_msg ? g.dw_msg[(long long)f * WM + 2 * F + k] : 0.f; },
                            [&](int c, int k, float v) { fold_put(o.dWe, (long long)c * Fe + k, v, o.accumulate & 4); },
                            sa, sb);
    return;
  }
  blk -= nb_e;
  // ---- the copied part of dP and the four vectors: plain element threads
  const long long nA = (long long)F * 2 * F;
  long long t = blk * 256LL + threadIdx.x;
  if (t < nA) {
    const int f = (int)(t / (2 * F)), j = (int)(t % (2 * F));
    fold_put(o.dP, (long long)f * 3 * F + j, g.dw_msg ? g.dw_msg[(long long)f * WM + j] : 0.f, o.accumulate & 1);
    return;
  }
  t -= nA;
  if (t < 4 * F) {
    const int c = (int)(t % F), which = (int)(t / F);
    float v = 0.f;
    if (which == 0) {                              // dpb = db_msg
      v = g.db_msg ? g.db_msg[c] : 0.f;
      fold_put(o.dpb, c, v, o.accumulate & 2);
    } else if (which == 1) {                       // dbe[c] = sum_f P3[f,c] db_msg[f]
      if (g.db_msg)
        for (int f = 0; f < F; ++f) v += p.P[(long long)f * 3 * F + 2 * F + c] * g.db_msg[f];
      fold_put(o.dbe, c, v, o.accumulate & 8);
    } else if (which == 2) {                       // dqb[k] = sum_f Lw[f,k] db_eff[f]
      if (g.db_eff)
        for (int f = 0; f < F; ++f) v += p.Lw[(long long)f * F + c] * g.db_eff[f];
      fold_put(o.dqb, c, v, o.accumulate & 32);
    } else {                                       // dlb = db_eff
      v = g.db_eff ? g.db_eff[c] : 0.f;
      fold_put(o.dlb, c, v, o.accumulate & 128);
    }
  }
}

// dLw[f,k] (+)= sum_chunks partial[chunk][f,k] + db_eff[f] qb[k]
__global__ void __launch_bounds__(256) k_pna_fold_lw_sum(tg_fold_params p, tg_fold_grads g, tg_fold_dparams o, int F, int nchunk) {
  const long long e = blockIdx.x * 256LL + threadIdx.x;
  if (e >= (long long)F * F) return;
  float v = 0.f;
  for (int c = 0; c < nchunk; ++c) v += o.ws[(long long)c * F * F + e];
  if (g.db_eff) v += g.db_eff[e / F] * p.qb[e % F];
  fold_put(o.dLw, e, v, o.accumulate & 64);
}

}  // namespace tg

using namespace tg;

static bool fold_dims(int32_t F, int32_t Fe, const int32_t* order, FoldDims& d) {
  d.F = F;
  d.Fe = Fe;
  bool seen[4] = {false, false, false, false};
  for (int kk = 0; kk < 4; ++kk) {
    if (order[kk] < 0 || order[kk] > 3 || seen[order[kk]]) return false;
    seen[order[kk]] = true;
    d.inv[order[kk]] = kk;
  }
  return true;
}

extern "C" int tg_pna_fold_fwd(const tg_fold_params* p, const tg_fold_out* o, int32_t F, int32_t Fe,
                               const int32_t* order, void* stream) {
  TG_CHECK(p && o && order && F > 0 && Fe > 0, "tg_pna_fold_fwd: null argument or empty shape");
  TG_CHECK(p->P && p->pb && p->We && p->be && p->Qw && p->qb && p->Lw && p->lb, "tg_pna_fold_fwd: null parameter");
  TG_CHECK(o->w_msg && o->b_msg && o->w_x && o->b_eff && o->w_st, "tg_pna_fold_fwd: null fp32 output");
  const bool lp = o->w_msg_lp || o->w_msg_lp_t || o->w_x_lp || o->w_x_lp_t || o->w_cat || o->wt_cat;
  TG_CHECK(!lp || (o->w_msg_lp && o->w_msg_lp_t && o->w_x_lp && o->w_x_lp_t && o->w_cat && o->wt_cat),
           "tg_pna_fold_fwd: the bf16 outputs come all or none");
  TG_CHECK(F % FT == 0 && Fe % FT == 0, "tg_pna_fold_fwd: F and Fe must be multiples of 32 (F=%d Fe=%d)", F, Fe);
  FoldDims d;
  TG_CHECK(fold_dims(F, Fe, order, d), "tg_pna_fold_fwd: order must be a permutation of 0..3");
  const int tf = F / FT;
  const long long blocks = (long long)tf * (Fe / FT) + (long long)tf * (13 * F / FT) + ceil_div((long long)F * 2 * F + 2 * F, 256);
  hipLaunchKernelGGL(k_pna_fold_fwd, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, *p, *o, d);
  TG_LAUNCH_CHECK();
  return 0;
}

extern "C" int64_t tg_pna_fold_ws_floats(int32_t F) { return (int64_t)((13 * F + LW_CHUNK - 1) / LW_CHUNK) * F * F; }

extern "C" int tg_pna_fold_bwd(const tg_fold_params* p, const tg_fold_grads* g, const tg_fold_dparams* o, int32_t F,
                               int32_t Fe, const int32_t* order, void* stream) {
  TG_CHECK(p && g && o && order && F > 0 && Fe > 0, "tg_pna_fold_bwd: null argument or empty shape");
  TG_CHECK(p->P && p->We && p->be && p->Qw && p->qb && p->Lw, "tg_pna_fold_bwd: null parameter");
  TG_CHECK(F % FT == 0 && Fe % FT == 0, "tg_pna_fold_bwd: F and Fe must be multiples of 32 (F=%d Fe=%d)", F, Fe);
  FoldDims d;
  TG_CHECK(fold_dims(F, Fe, order, d), "tg_pna_fold_bwd: order must be a permutation of 0..3");
  const int tf = F / FT, nchunk = (13 * F + LW_CHUNK - 1) / LW_CHUNK;
  TG_CHECK(!o->dLw || o->ws, "tg_pna_fold_bwd: dLw needs the workspace (tg_pna_fold_ws_floats(F) floats)");
  const long long blocks = (long long)tf * tf * (nchunk + 1) + (long long)tf * (13 * F / FT) + (long long)tf * (Fe / FT) +
                           ceil_div((long long)F * 2 * F + 4 * F, 256);
  hipLaunchKernelGGL(k_pna_fold_bwd, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, *p, *g, *o, d);
  if (o->dLw)
    hipLaunchKernelGGL(k_pna_fold_lw_sum, dim3((unsigned)ceil_div((long long)F * F, 256)), dim3(256), 0,
                       (hipStream_t)stream, *p, *g, *o, F, nchunk);
  TG_LAUNCH_CHECK();
  return 0;
}
