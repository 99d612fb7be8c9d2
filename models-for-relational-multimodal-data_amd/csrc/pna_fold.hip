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

__global__ void __launch_bounds__(256) k_pna_fold_fwd(tg_fold_params p, tg_fold_out o, FoldDims d) {
  const int F = d.F, Fe = d.Fe, WM = 2 * F + Fe, K = 4 * F;
  const long long nA = (long long)F * WM, nB = (long long)F * 13 * F, nC = 2 * F;
  long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (t < nA) {                                    // ---- w_msg
    const int f = (int)(t / WM), j = (int)(t % WM);
    float v;
    if (j < 2 * F) v = p.P[(long long)f * 3 * F + j];
    else {
      const int c = j - 2 * F;
      const float* pr = p.P + (long long)f * 3 * F + 2 * F;
      v = 0.f;
      for (int k = 0; k < F; ++k) v += pr[k] * p.We[(long long)k * Fe + c];
    }
    o.w_msg[t] = v;
    if (o.w_msg_lp) {
      reinterpret_cast<unsigned short*>(o.w_msg_lp)[t] = f2bf(v);
      reinterpret_cast<unsigned short*>(o.w_msg_lp_t)[(long long)j * F + f] = f2bf(v);
    }
    return;
  }
  t -= nA;
  if (t < nB) {                                    // ---- w_eff = Lw Qw, scattered into w_x / w_st and their packs
    const int f = (int)(t / (13 * F)), j = (int)(t % (13 * F));
    const float* lr = p.Lw + (long long)f * F;
    float v = 0.f;
    for (int k = 0; k < F; ++k) v += lr[k] * p.Qw[(long long)k * 13 * F + j];
    if (j < F) {
      o.w_x[(long long)f * F + j] = v;
      if (o.w_x_lp) {
        reinterpret_cast<unsigned short*>(o.w_x_lp)[(long long)f * F + j] = f2bf(v);
        reinterpret_cast<unsigned short*>(o.w_x_lp_t)[(long long)j * F + f] = f2bf(v);
      }
    } else {
      const int jj = j - F, s = jj / K, a = (jj / F) & 3, i = jj % F;
      const int col = d.inv[a] * F + i;
      o.w_st[((long long)s * F + f) * K + col] = v;
      if (o.w_cat) {
        reinterpret_cast<unsigned short*>(o.w_cat)[(long long)f * 3 * K + ((col >> 7) * 3 + s) * 128 + (col & 127)] = f2bf(v);
        reinterpret_cast<unsigned short*>(o.wt_cat)[(long long)col * 3 * F + s * F + f] = f2bf(v);
      }
    }
    return;
  }
  t -= nB;
  if (t < nC) {                                    // ---- folded biases
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

// gradient of w_eff at (f, j) from its two carriers (either may be absent = zero)
__device__ __forceinline__ float fold_dweff(const tg_fold_grads& g, const FoldDims& d, int f, int j) {
  const int F = d.F;
  if (j < F) return g.dw_x ? g.dw_x[(long long)f * F + j] : 0.f;
  if (!g.dw_st) return 0.f;
  const int jj = j - F, s = jj / (4 * F), a = (jj / F) & 3, i = jj % F;
  return g.dw_st[((long long)s * F + f) * 4 * F + d.inv[a] * F + i];
}

__device__ __forceinline__ void fold_put(float* out, long long i, float v, bool acc) {
  if (out) out[i] = acc ? out[i] + v : v;
}

// every parameter gradient but dLw: one thread per output element, sums over F rows
__global__ void __launch_bounds__(256) k_pna_fold_bwd(tg_fold_params p, tg_fold_grads g, tg_fold_dparams o, FoldDims d) {
  const int F = d.F, Fe = d.Fe, WM = 2 * F + Fe;
  const long long n1 = (long long)F * 3 * F, n2 = (long long)F * Fe, n3 = (long long)F * 13 * F, n4 = 4 * F;
  long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (t < n1) {                                    // ---- dP
    const int f = (int)(t / (3 * F)), j = (int)(t % (3 * F));
    float v = 0.f;
    if (j < 2 * F) v = g.dw_msg ? g.dw_msg[(long long)f * WM + j] : 0.f;
    else {
      const int c = j - 2 * F;
      if (g.dw_msg) {
        const float* dr = g.dw_msg + (long long)f * WM + 2 * F;
        const float* wr = p.We + (long long)c * Fe;
        for (int k = 0; k < Fe; ++k) v += dr[k] * wr[k];
      }
      if (g.db_msg) v += g.db_msg[f] * p.be[c];
    }
    fold_put(o.dP, t, v, o.accumulate & 1);
    return;
  }
  t -= n1;
  if (t < n2) {                                    // ---- dWe[c,k] = sum_f P3[f,c] d3[f,k]
    const int c = (int)(t / Fe), k = (int)(t % Fe);
    float v = 0.f;
    if (g.dw_msg)
      for (int f = 0; f < F; ++f) v += p.P[(long long)f * 3 * F + 2 * F + c] * g.dw_msg[(long long)f * WM + 2 * F + k];
    fold_put(o.dWe, t, v, o.accumulate & 4);
    return;
  }
  t -= n2;
  if (t < n3) {                                    // ---- dQw[k,j] = sum_f Lw[f,k] dw_eff[f,j]
    const int k = (int)(t / (13 * F)), j = (int)(t % (13 * F));
    float v = 0.f;
    for (int f = 0; f < F; ++f) v += p.Lw[(long long)f * F + k] * fold_dweff(g, d, f, j);
    fold_put(o.dQw, t, v, o.accumulate & 16);
    return;
  }
  t -= n3;
  if (t < n4) {                                    // ---- the four vectors
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

// dLw[f,k] = sum_j dw_eff[f,j] Qw[k,j] + db_eff[f] qb[k]: a 13F-long dot product per element, one wave each
__global__ void __launch_bounds__(256) k_pna_fold_bwd_lw(tg_fold_params p, tg_fold_grads g, tg_fold_dparams o, FoldDims d) {
  const int F = d.F, lane = threadIdx.x & 63;
  const long long e = blockIdx.x * 4LL + (threadIdx.x >> 6);
  if (e >= (long long)F * F) return;
  const int f = (int)(e / F), k = (int)(e % F);
  const float* qr = p.Qw + (long long)k * 13 * F;
  float v = 0.f;
  for (int j = lane; j < 13 * F; j += 64) v += fold_dweff(g, d, f, j) * qr[j];
  v = group_sum<64>(v);
  if (lane == 0) {
    if (g.db_eff) v += g.db_eff[f] * p.qb[k];
    fold_put(o.dLw, e, v, o.accumulate & 64);
  }
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
  TG_CHECK(!lp || F % 32 == 0, "tg_pna_fold_fwd: the scaled-projection pack needs 4F %% 128 == 0 (F=%d)", F);
  FoldDims d;
  TG_CHECK(fold_dims(F, Fe, order, d), "tg_pna_fold_fwd: order must be a permutation of 0..3");
  const long long n = (long long)F * (2 * F + Fe) + (long long)F * 13 * F + 2 * F;
  hipLaunchKernelGGL(k_pna_fold_fwd, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, *p, *o, d);
  TG_LAUNCH_CHECK();
  return 0;
}

extern "C" int tg_pna_fold_bwd(const tg_fold_params* p, const tg_fold_grads* g, const tg_fold_dparams* o, int32_t F,
                               int32_t Fe, const int32_t* order, void* stream) {
  TG_CHECK(p && g && o && order && F > 0 && Fe > 0, "tg_pna_fold_bwd: null argument or empty shape");
  TG_CHECK(p->P && p->We && p->be && p->Qw && p->qb && p->Lw, "tg_pna_fold_bwd: null parameter");
  FoldDims d;
  TG_CHECK(fold_dims(F, Fe, order, d), "tg_pna_fold_bwd: order must be a permutation of 0..3");
  const long long n = (long long)F * 3 * F + (long long)F * Fe + (long long)F * 13 * F + 4 * F;
  hipLaunchKernelGGL(k_pna_fold_bwd, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, *p, *g, *o, d);
  if (o->dLw)
    hipLaunchKernelGGL(k_pna_fold_bwd_lw, dim3((unsigned)ceil_div((long long)F * F, 4)), dim3(256), 0,
                       (hipStream_t)stream, *p, *g, *o, d);
  TG_LAUNCH_CHECK();
  return 0;
}
