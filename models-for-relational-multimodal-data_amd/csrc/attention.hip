// Column self-attention core of the FT-Transformer layer (torch nn.MultiheadAttention inside
// nn.TransformerEncoderLayer, configured at src/nn/models/fused.py:83-92,187-196; restated in
// oracle/transformer.py).  The "sequence" is the S = ncols+1 column tokens of one table row
// (6 for AML, 130 for ogbn-arxiv nodes), so the score matrix is tiny and lives in registers:
// one thread per (row, head, query), online softmax over the S keys, dropout on P from the
// counter RNG (recomputed in backward: no mask, no [R,H,S,S] tensor in HBM).
#include "common.hpp"
#include "../../include/tabgnn_hip.h"

namespace tg {

template <typename T, int D> struct HeadVec {
  // D contiguous elements of one head -> float registers
  __device__ static __forceinline__ void load(const T* p, float (&o)[D]) {
    constexpr int V = (V16<T>::N <= D) ? V16<T>::N : D;
#pragma unroll
    for (int i = 0; i < D; i += V) {
      float t[V];
      loadv<T, V>(p + i, t);
#pragma unroll
      for (int j = 0; j < V; ++j) o[i + j] = t[j];
    }
  }
  __device__ static __forceinline__ void store(T* p, const float (&o)[D]) {
    constexpr int V = (V16<T>::N <= D) ? V16<T>::N : D;
#pragma unroll
    for (int i = 0; i < D; i += V) {
      float t[V];
#pragma unroll
      for (int j = 0; j < V; ++j) t[j] = o[i + j];
      storev<T, V>(p + i, t);
    }
  }
};

// qkv [R,S,3C]; out [R,S,C]; lse [R,H,S] (log-sum-exp of the scaled scores, for the backward)
template <typename T, int D>
__global__ void __launch_bounds__(256) k_attn_fwd(const T* __restrict__ qkv, T* __restrict__ out,
                                                   float* __restrict__ lse, long long R, int S, int H, float scale,
                                                   unsigned thresh, float inv_keep, unsigned long long seed,
                                                   unsigned rstream) {
  seed = live_seed(seed);
  const int C = H * D;
  long long total = R * H * S;
  long long gid = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  long long stride = (long long)gridDim.x * blockDim.x;
  for (; gid < total; gid += stride) {
    int sq = (int)(gid % S);
    long long rh = gid / S;
    // the S*S mask indices of one (row, head) share the dropout key unless they straddle a 2^32 boundary
    const unsigned long long blk0 = (unsigned long long)rh * S * S;
    const bool one_key = (blk0 >> 32) == ((blk0 + (unsigned long long)(S * S - 1)) >> 32);
    const unsigned key = rng_key(seed, rstream, (unsigned)(blk0 >> 32));
    int h = (int)(rh % H);
    long long r = rh / H;
    const T* base = qkv + r * S * 3 * C + h * D;
    float q[D], acc[D];
    HeadVec<T, D>::load(base + (long long)sq * 3 * C, q);
#pragma unroll
    for (int i = 0; i < D; ++i) { q[i] *= scale; acc[i] = 0.f; }
    float m = -INFINITY, l = 0.f;
    for (int k = 0; k < S; ++k) {
      float kk[D], vv[D];
      HeadVec<T, D>::load(base + (long long)k * 3 * C + C, kk);
      HeadVec<T, D>::load(base + (long long)k * 3 * C + 2 * C, vv);
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < D; ++i) s += q[i] * kk[i];
      float mn = fmaxf(m, s);
      float corr = __expf(m - mn);
      float p = __expf(s - mn);
      l = l * corr + p;
      float pm = p;
      if (thresh) pm *= one_key ? drop_scale_key(key, (unsigned)(blk0 + (unsigned long long)(sq * S + k)), thresh, inv_keep)
                                : drop_scale(seed, rstream, blk0 + (unsigned long long)(sq * S + k), thresh, inv_keep);
#pragma unroll
      for (int i = 0; i < D; ++i) acc[i] = acc[i] * corr + pm * vv[i];
      m = mn;
    }
    float inv = 1.f / l;
#pragma unroll
    for (int i = 0; i < D; ++i) acc[i] *= inv;
    HeadVec<T, D>::store(out + (r * S + sq) * C + h * D, acc);
    if (lse) lse[gid] = m + __logf(l);
  }
}

// dqkv [R,S,3C] from dout [R,S,C]; recomputes P from (q,k,lse); delta_q = dO_q . O_q
template <typename T, int D>
__global__ void __launch_bounds__(256) k_attn_bwd(const T* __restrict__ qkv, const T* __restrict__ o,
                                                   const T* __restrict__ dout, const float* __restrict__ lse,
                                                   T* __restrict__ dqkv, long long R, int S, int H, float scale,
                                                   unsigned thresh, float inv_keep, unsigned long long seed,
                                                   unsigned rstream) {
  seed = live_seed(seed);
  const int C = H * D;
  long long total = R * H * S;
  long long gid = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  long long stride = (long long)gridDim.x * blockDim.x;
  for (; gid < total; gid += stride) {
    int me = (int)(gid % S);
    long long rh = gid / S;
    // the S*S mask indices of one (row, head) share the dropout key unless they straddle a 2^32 boundary
    const unsigned long long blk0 = (unsigned long long)rh * S * S;
    const bool one_key = (blk0 >> 32) == ((blk0 + (unsigned long long)(S * S - 1)) >> 32);
    const unsigned key = rng_key(seed, rstream, (unsigned)(blk0 >> 32));
    int h = (int)(rh % H);
    long long r = rh / H;
    const T* base = qkv + r * S * 3 * C + h * D;
    const T* obase = o + r * S * C + h * D;
    const T* gbase = dout + r * S * C + h * D;
    const float* lbase = lse + rh * S;
    T* dbase = dqkv + r * S * 3 * C + h * D + (long long)me * 3 * C;
    // ---- pass A: me as QUERY.  dq = sum_j dS[me,j] * K[j]
    {
      float qm[D], gm[D], dq[D];
      HeadVec<T, D>::load(base + (long long)me * 3 * C, qm);
      HeadVec<T, D>::load(gbase + (long long)me * C, gm);
      float delta_me = 0.f;
      {
        float om[D];
        HeadVec<T, D>::load(obase + (long long)me * C, om);
#pragma unroll
        for (int i = 0; i < D; ++i) delta_me += gm[i] * om[i];
      }
      float lse_me = lbase[me];
#pragma unroll
      for (int i = 0; i < D; ++i) dq[i] = 0.f;
      for (int j = 0; j < S; ++j) {
        float kj[D], vj[D];
        HeadVec<T, D>::load(base + (long long)j * 3 * C + C, kj);
        HeadVec<T, D>::load(base + (long long)j * 3 * C + 2 * C, vj);
        float s = 0.f, dp = 0.f;
#pragma unroll
        for (int i = 0; i < D; ++i) { s += qm[i] * kj[i]; dp += gm[i] * vj[i]; }
        float p = __expf(s * scale - lse_me);
        float mk = !thresh ? 1.f
                   : one_key ? drop_scale_key(key, (unsigned)(blk0 + (unsigned long long)(me * S + j)), thresh, inv_keep)
                             : drop_scale(seed, rstream, blk0 + (unsigned long long)(me * S + j), thresh, inv_keep);
        float ds = p * (dp * mk - delta_me) * scale;
#pragma unroll
        for (int i = 0; i < D; ++i) dq[i] += ds * kj[i];
      }
      HeadVec<T, D>::store(dbase, dq);
    }
    // ---- pass B: me as KEY.  dk = sum_j dS[j,me] * Q[j],  dv = sum_j P[j,me]*mask * dO[j]
    {
      float km[D], vm[D], dk[D], dv[D];
      HeadVec<T, D>::load(base + (long long)me * 3 * C + C, km);
      HeadVec<T, D>::load(base + (long long)me * 3 * C + 2 * C, vm);
#pragma unroll
      for (int i = 0; i < D; ++i) { dk[i] = 0.f; dv[i] = 0.f; }
      for (int j = 0; j < S; ++j) {
        float qj[D], gj[D];
        float s = 0.f, dp = 0.f, delta_j = 0.f;
        HeadVec<T, D>::load(gbase + (long long)j * C, gj);
        {
          float oj[D];
          HeadVec<T, D>::load(obase + (long long)j * C, oj);
#pragma unroll
          for (int i = 0; i < D; ++i) delta_j += gj[i] * oj[i];
        }
        HeadVec<T, D>::load(base + (long long)j * 3 * C, qj);
#pragma unroll
        for (int i = 0; i < D; ++i) { s += qj[i] * km[i]; dp += gj[i] * vm[i]; }
        float p = __expf(s * scale - lbase[j]);
        float mk = !thresh ? 1.f
                   : one_key ? drop_scale_key(key, (unsigned)(blk0 + (unsigned long long)(j * S + me)), thresh, inv_keep)
                             : drop_scale(seed, rstream, blk0 + (unsigned long long)(j * S + me), thresh, inv_keep);
        float ds = p * (dp * mk - delta_j) * scale;
        float pm = p * mk;
#pragma unroll
        for (int i = 0; i < D; ++i) { dk[i] += ds * qj[i]; dv[i] += pm * gj[i]; }
      }
      HeadVec<T, D>::store(dbase + C, dk);
      HeadVec<T, D>::store(dbase + 2 * C, dv);
    }
  }
}

// Lane-split backward for head dims >= 16: the D channels of a (row, head, token) are spread over LPD = D/8 adjacent
// lanes (8 channels = one 16-byte load each), dot products finish with xor-shuffles inside that lane group.  The
// thread-per-token kernel above keeps 5 x D values live (250 VGPRs at D = 32 -> 2 waves per SIMD, latency-bound:
// 1.84 ms at R = 430 k); here a lane holds 5 x 8.  Same arithmetic order per channel, the dot products only differ by
// the (pairwise) order of the final cross-lane adds.  D = 32 at R = 430 k: 1.84 -> 1.58 ms with 8 channels per lane;
// 4 channels per lane (more lanes, more replicated exp / mask / shuffle work) measured 2.2 ms.
template <typename T, int D, int DL>
__global__ void __launch_bounds__(256) k_attn_bwd_split(const T* __restrict__ qkv, const T* __restrict__ o,
                                                         const T* __restrict__ dout, const float* __restrict__ lse,
                                                         T* __restrict__ dqkv, long long R, int S, int H, float scale,
                                                         unsigned thresh, float inv_keep, unsigned long long seed,
                                                         unsigned rstream) {
  seed = live_seed(seed);
  constexpr int LPD = D / DL;
  static_assert(D % DL == 0 && (LPD & (LPD - 1)) == 0 && LPD <= 16, "head dim must be DL * power of two");
  const int C = H * D;
  const long long total = R * H * S * LPD;
  long long gid = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  const long long stride = (long long)gridDim.x * blockDim.x;      // multiple of LPD: lane groups iterate together
#define TG_RED(v) for (int m_ = LPD >> 1; m_ > 0; m_ >>= 1) v += __shfl_xor(v, m_, 64);
  for (; gid < total; gid += stride) {
    const int dq = (int)(gid % LPD) * DL;
    const long long tok = gid / LPD;
    const int me = (int)(tok % S);
    const long long rh = tok / S;
    const int h = (int)(rh % H);
    const long long r = rh / H;
    const unsigned long long blk0 = (unsigned long long)rh * S * S;
    const bool one_key = (blk0 >> 32) == ((blk0 + (unsigned long long)(S * S - 1)) >> 32);
    const unsigned key = rng_key(seed, rstream, (unsigned)(blk0 >> 32));
    const T* base = qkv + r * S * 3 * C + h * D + dq;
    const T* obase = o + r * S * C + h * D + dq;
    const T* gbase = dout + r * S * C + h * D + dq;
    const float* lbase = lse + rh * S;
    T* dbase = dqkv + r * S * 3 * C + h * D + dq + (long long)me * 3 * C;
    float qm[DL], km[DL], vm[DL], gm[DL];
    loadv<T, DL>(base + (long long)me * 3 * C, qm);
    loadv<T, DL>(base + (long long)me * 3 * C + C, km);
    loadv<T, DL>(base + (long long)me * 3 * C + 2 * C, vm);
    loadv<T, DL>(gbase + (long long)me * C, gm);
    float delta_me = 0.f;
    {
      float om[DL];
      loadv<T, DL>(obase + (long long)me * C, om);
#pragma unroll
      for (int i = 0; i < DL; ++i) delta_me += gm[i] * om[i];
    }
    TG_RED(delta_me)
    const float lse_me = lbase[me];
    float dqv[DL], dk[DL], dv[DL];
#pragma unroll
    for (int i = 0; i < DL; ++i) { dqv[i] = 0.f; dk[i] = 0.f; dv[i] = 0.f; }
    for (int j = 0; j < S; ++j) {
      float qj[DL], kj[DL], vj[DL], gj[DL], oj[DL];
      loadv<T, DL>(base + (long long)j * 3 * C, qj);
      loadv<T, DL>(base + (long long)j * 3 * C + C, kj);
      loadv<T, DL>(base + (long long)j * 3 * C + 2 * C, vj);
      loadv<T, DL>(gbase + (long long)j * C, gj);
      loadv<T, DL>(obase + (long long)j * C, oj);
      // me as QUERY against key j; me as KEY against query j
      float sA = 0.f, dpA = 0.f, sB = 0.f, dpB = 0.f, delta_j = 0.f;
#pragma unroll
      for (int i = 0; i < DL; ++i) {
        sA += qm[i] * kj[i]; dpA += gm[i] * vj[i];
        sB += qj[i] * km[i]; dpB += gj[i] * vm[i]; delta_j += gj[i] * oj[i];
      }
      TG_RED(sA) TG_RED(dpA) TG_RED(sB) TG_RED(dpB) TG_RED(delta_j)
      const float pA = __expf(sA * scale - lse_me);
      const float pB = __expf(sB * scale - lbase[j]);
      float mkA = 1.f, mkB = 1.f;
      if (thresh) {
        const unsigned long long iA = blk0 + (unsigned long long)(me * S + j), iB = blk0 + (unsigned long long)(j * S + me);
        mkA = one_key ? drop_scale_key(key, (unsigned)iA, thresh, inv_keep) : drop_scale(seed, rstream, iA, thresh, inv_keep);
        mkB = one_key ? drop_scale_key(key, (unsigned)iB, thresh, inv_keep) : drop_scale(seed, rstream, iB, thresh, inv_keep);
      }
      const float dsA = pA * (dpA * mkA - delta_me) * scale;
      const float dsB = pB * (dpB * mkB - delta_j) * scale;
      const float pmB = pB * mkB;
#pragma unroll
      for (int i = 0; i < DL; ++i) { dqv[i] += dsA * kj[i]; dk[i] += dsB * qj[i]; dv[i] += pmB * gj[i]; }
    }
    storev<T, DL>(dbase, dqv);
    storev<T, DL>(dbase + C, dk);
    storev<T, DL>(dbase + 2 * C, dv);
  }
#undef TG_RED
}

}  // namespace tg

using namespace tg;

struct AttnArgs {
  const void *qkv, *o, *dout;
  void *out, *dqkv;
  float* lse;
  long long R;
  int S, H;
  float scale;
  unsigned thresh;
  float inv_keep;
  unsigned long long seed;
  unsigned rstream;
  int grid;
  hipStream_t st;
};

template <typename T, int D> static void launch_fwd(const AttnArgs& a) {
  // (a lane-split forward like k_attn_bwd_split measured slower: 674 vs 608 us at R = 430 k, D = 32 — the forward
  // keeps only q and the output accumulator live, so splitting buys no occupancy and pays the shuffles)
  hipLaunchKernelGGL((k_attn_fwd<T, D>), dim3(a.grid), dim3(256), 0, a.st, (const T*)a.qkv, (T*)a.out, a.lse, a.R, a.S,
                     a.H, a.scale, a.thresh, a.inv_keep, a.seed, a.rstream);
}
template <typename T, int D> static void launch_bwd(const AttnArgs& a) {
  if constexpr (D >= 16 && sizeof(T) == 2) {          // bf16, 16-byte lane slices: lane-split kernel
    int grid = grid_cap(ceil_div(a.R * a.H * a.S * (D / 8), 256), 256 * 32);
    hipLaunchKernelGGL((k_attn_bwd_split<T, D, 8>), dim3(grid), dim3(256), 0, a.st, (const T*)a.qkv, (const T*)a.o,
                       (const T*)a.dout, (const float*)a.lse, (T*)a.dqkv, a.R, a.S, a.H, a.scale, a.thresh, a.inv_keep,
                       a.seed, a.rstream);
  } else {
    hipLaunchKernelGGL((k_attn_bwd<T, D>), dim3(a.grid), dim3(256), 0, a.st, (const T*)a.qkv, (const T*)a.o,
                       (const T*)a.dout, (const float*)a.lse, (T*)a.dqkv, a.R, a.S, a.H, a.scale, a.thresh, a.inv_keep,
                       a.seed, a.rstream);
  }
}
template <typename T> static int dispatch(const AttnArgs& a, int D, bool fwd) {
  switch (D) {
    case 4: fwd ? launch_fwd<T, 4>(a) : launch_bwd<T, 4>(a); break;
    case 8: fwd ? launch_fwd<T, 8>(a) : launch_bwd<T, 8>(a); break;
    case 16: fwd ? launch_fwd<T, 16>(a) : launch_bwd<T, 16>(a); break;
    case 32: fwd ? launch_fwd<T, 32>(a) : launch_bwd<T, 32>(a); break;
    case 64: fwd ? launch_fwd<T, 64>(a) : launch_bwd<T, 64>(a); break;
    default: set_error("attention: unsupported head dim %d (need 4/8/16/32/64)", D); return 1;
  }
  return 0;
}

static int attn_common(AttnArgs& a, int64_t R, int32_t S, int32_t C, int32_t H, float p_drop, uint64_t seed,
                       uint32_t rstream, void* stream) {
  a.R = R; a.S = S; a.H = H;
  a.scale = 1.f / sqrtf((float)(C / H));
  a.thresh = p_drop > 0.f ? drop_threshold(p_drop) : 0u;
  a.inv_keep = p_drop < 1.f ? 1.f / (1.f - p_drop) : 0.f;
  a.seed = seed; a.rstream = rstream;
  a.grid = grid_cap(ceil_div((long long)R * H * S, 256), 256 * 32);
  a.st = (hipStream_t)stream;
  return 0;
}

namespace tg {      // attention_mfma.hip: long rows in bf16 on MFMA
bool attn_mfma_ok(int32_t S, int32_t C, int32_t H, int32_t dt);
void attn_mfma_fwd(const void* qkv, void* out, float* lse, int64_t R, int32_t S, int32_t C, int32_t H, float scale,
                   unsigned thresh, float inv_keep, uint64_t seed, uint32_t rstream, hipStream_t st);
void attn_mfma_bwd(const void* qkv, const void* o, const void* dout, const float* lse, void* dqkv, int64_t R, int32_t S,
                   int32_t C, int32_t H, float scale, unsigned thresh, float inv_keep, uint64_t seed, uint32_t rstream,
                   hipStream_t st);
}

extern "C" int tg_attn_fwd(const void* qkv, void* out, float* lse, int64_t R, int32_t S, int32_t C, int32_t H,
                           float p_drop, uint64_t seed, uint32_t rstream, int32_t dt, void* stream) {
  TG_CHECK(H > 0 && C % H == 0 && S > 0, "tg_attn_fwd: bad geometry C=%d H=%d S=%d", C, H, S);
  if (R == 0) return 0;
  AttnArgs a{};
  a.qkv = qkv; a.out = out; a.lse = lse;
  attn_common(a, R, S, C, H, p_drop, seed, rstream, stream);
  if (attn_mfma_ok(S, C, H, dt)) {
    attn_mfma_fwd(qkv, out, lse, R, S, C, H, a.scale, a.thresh, a.inv_keep, seed, rstream, a.st);
    TG_LAUNCH_CHECK();
    return 0;
  }
  int rc = dt == F32 ? dispatch<float>(a, C / H, true) : dispatch<bf16_t>(a, C / H, true);
  if (rc) return rc;
  TG_LAUNCH_CHECK();
  return 0;
}

extern "C" int tg_attn_bwd(const void* qkv, const void* o, const void* dout, const float* lse, void* dqkv, int64_t R,
                           int32_t S, int32_t C, int32_t H, float p_drop, uint64_t seed, uint32_t rstream, int32_t dt,
                           void* stream) {
  TG_CHECK(H > 0 && C % H == 0 && S > 0, "tg_attn_bwd: bad geometry C=%d H=%d S=%d", C, H, S);
  if (R == 0) return 0;
  AttnArgs a{};
  a.qkv = qkv; a.o = o; a.dout = dout; a.lse = const_cast<float*>(lse); a.dqkv = dqkv;
  attn_common(a, R, S, C, H, p_drop, seed, rstream, stream);
  if (attn_mfma_ok(S, C, H, dt)) {
    TG_CHECK(lse != nullptr, "tg_attn_bwd: the MFMA path needs the forward's log-sum-exp");
    attn_mfma_bwd(qkv, o, dout, lse, dqkv, R, S, C, H, a.scale, a.thresh, a.inv_keep, seed, rstream, a.st);
    TG_LAUNCH_CHECK();
    return 0;
  }
  int rc = dt == F32 ? dispatch<float>(a, C / H, false) : dispatch<bf16_t>(a, C / H, false);
  if (rc) return rc;
  TG_LAUNCH_CHECK();
  return 0;
}

