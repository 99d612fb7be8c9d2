// Column self-attention core on MFMA for LONG token rows (S up to a few hundred: 130 for the ogbn-arxiv node table of
// BASELINE configs[3], 65 for the 64-column table of configs[4]) — bf16, head dim 16 or 32.
//   torch nn.MultiheadAttention inside nn.TransformerEncoderLayer as configured at src/nn/models/fused.py:83-92 and
//   src/nn/models/tabgnn.py:100-129; restated in oracle/transformer.py.  Same entry points (tg_attn_fwd / tg_attn_bwd), same
//   dropout element indexing ((row * H + head) * S + q) * S + k and the same log-sum-exp tensor as the thread-per-query
//   kernels of attention.hip, which stay for short rows and fp32.
// Round 2 ran these rows on the thread-per-query kernels: 96.6 of the 126 ms of a tabgnn-arxiv step
// (profiles/r03_arxiv_kernel_stats_before.csv).
//
// Layout of one 32 x 32 tile (MFMA 32x32x16 bf16): KEYS on the accumulator rows, QUERIES on the lanes,
//     S^T[key, q] = sum_d K[key, d] Q[q, d]      A = K rows, B = Q^T: both operands are 16-byte pieces of qkv rows, read
//                                                straight from global memory into fragment registers (no LDS),
// so the softmax of a query runs down one lane's registers (+ one xor-32 exchange) and its statistics are lane scalars.
//   forward (one wave per (row, head, query tile), online softmax over the key tiles):
//     O^T[d, q] += sum_key V[key, d] Pd^T[key, q]   A = V^T: the V tile goes through a wave-private LDS tile and comes back
//                                                   transposed (ds_read_b64_tr_b16), B = the packed probabilities.
//   backward, two kernels (every wave recomputes S^T, P^T from the saved log-sum-exp; dP^T = V dO^T; dS^T):
//     dQ  (one wave per (row, head, query tile)):  dQ^T[d, q] += sum_key K[key, d] dS^T[key, q]   (K^T by transposed LDS read)
//     dKV (one wave per (row, head, key tile)):    Pd, dS with queries on the k axis are the tile TRANSPOSES (LDS), then
//         dV^T[d, key] += sum_q dO[q, d] Pd[q, key] ;  dK^T[d, key] += sum_q Q[q, d] dS[q, key]   (dO^T, Q^T by transposed reads)
// Rows of a head are 32 or 64 bytes inside 3C-wide token rows, so global traffic is in 16-byte pieces of whole 32/64-byte
// segments; the tensors are small against the layer's projections (the op is latency / issue bound, not HBM bound).
#include "common.hpp"
#include "../../include/tabgnn_hip.h"

namespace tg {

typedef __bf16 am_v8bf __attribute__((ext_vector_type(8)));
typedef float am_f32x16 __attribute__((ext_vector_type(16)));
typedef float am_f32x8 __attribute__((ext_vector_type(8)));
typedef short am_v4s __attribute__((ext_vector_type(4)));
typedef am_v4s __attribute__((address_space(3))) * am_lds_v4s_ptr;

constexpr int AM_ROWB = 80;                    // LDS tile row: 32 bf16 (64 B) padded to 80 B
constexpr int AM_TILE = 32 * AM_ROWB;          // 2560 B
#define AM_MFMA(A, B, C) __builtin_amdgcn_mfma_f32_32x32x16_bf16((A), (B), (C), 0, 0, 0)

__device__ __forceinline__ am_f32x16 am_zero() {
  am_f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.f;
  return z;
}
template <int S8> __device__ __forceinline__ am_v8bf am_pack(const am_f32x16& a) {
  am_f32x8 t;
#pragma unroll
  for (int j = 0; j < 8; ++j) t[j] = a[8 * S8 + j];
  return __builtin_convertvector(t, am_v8bf);
}
__device__ __forceinline__ float am_xor32(float v) { return __shfl_xor(v, 32, 64); }
__device__ __forceinline__ unsigned am_bitmask(int hs, int bit) {      // all ones when bit `bit` of hs is set (opaque)
  unsigned m = (unsigned)__builtin_amdgcn_sbfe(hs, bit, 1);
  asm("" : "+v"(m));
  return m;
}

struct AmArgs {
  const unsigned short *qkv, *o, *dout;
  unsigned short *out, *dqkv;
  float* lse;                     // [R, H, S], natural log domain (as attention.hip)
  long long R;
  int S, H, C;
  float scale;
  unsigned thresh;
  float inv_keep;
  unsigned long long seed;
  unsigned rstream;
};

// 16 bytes of a token row as an MFMA fragment (lane (tl, h): 8 consecutive elements starting at column col0 + 8h of token
// `tok`, clamped to the last token of the row group; invalid lanes are zeroed by the caller where it matters)
__device__ __forceinline__ am_v8bf am_load_frag(const unsigned short* base, long long tok, int ld, int col) {
  return __builtin_bit_cast(am_v8bf, *reinterpret_cast<const uint4*>(base + tok * ld + col));
}

// A [32 rows][HD columns] bf16 tile of token rows (row r = token tok0 + r, clamped to tok_last) -> wave-private LDS tile
// (80-byte rows), in two halves so that the NEXT tile's global loads are in flight while the current one is computed:
// am_stage_load (registers) ... am_stage_write (LDS).  HD = 16: lane = (row = lane >> 1, 16-byte piece lane & 1);
// HD = 32: two rounds of (lane >> 2, lane & 3).
template <int HD> struct AmStage { uint4 v[HD / 16]; };
template <int HD>
__device__ __forceinline__ void am_stage_load(AmStage<HD>& st, const unsigned short* base, long long tok0, long long tok_last,
                                              int ld, int col, int lane) {
  constexpr int PPR = HD / 8;                  // 16-byte pieces per row
#pragma unroll
  for (int rnd = 0; rnd < HD / 16; ++rnd) {
    const int idx = rnd * 64 + lane, row = idx / PPR, pc = idx % PPR;
    long long tok = tok0 + row;
    tok = tok > tok_last ? tok_last : tok;
    st.v[rnd] = *reinterpret_cast<const uint4*>(base + tok * ld + col + 8 * pc);
  }
}
template <int HD>
__device__ __forceinline__ void am_stage_write(const AmStage<HD>& st, char* tile, int lane) {
  constexpr int PPR = HD / 8;
#pragma unroll
  for (int rnd = 0; rnd < HD / 16; ++rnd) {
    const int idx = rnd * 64 + lane, row = idx / PPR, pc = idx % PPR;
    *reinterpret_cast<uint4*>(tile + AM_ROWB * row + 16 * pc) = st.v[rnd];
  }
}
// transposed fragments of a staged tile Z[row][col] as the A operand that goes with a PACKED ACCUMULATOR as B: the k order
// of such a B fragment is the accumulator's row order (element j of lane half h <-> row 16ks + 8(j>>2) + 4h + (j&3)), so
// lane (c = lane & 31, h) gets Z[16ks + 8(j>>2) + 4h + (j&3)][c], j = 0..7: two transposed reads of 4 rows each.
//   trb = tile + 80 * (4 * h + ((lane & 15) >> 2)) + 32 * ((lane >> 4) & 1) + 8 * (lane & 3)
__device__ __forceinline__ am_v8bf am_tr_frag(const char* trb, int ks) {
  const uint2 lo = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((am_lds_v4s_ptr)(trb + AM_ROWB * (16 * ks))));
  const uint2 hi = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((am_lds_v4s_ptr)(trb + AM_ROWB * (16 * ks + 8))));
  return __builtin_bit_cast(am_v8bf, make_uint4(lo.x, lo.y, hi.x, hi.y));
}
// transpose of a 32 x 32 tile held as two packed accumulator fragments (see encoder_fused.hip: ef_transpose32)
__device__ __forceinline__ void am_transpose32(am_v8bf f0, am_v8bf f1, char* tw, const char* ttr, am_v8bf& t0, am_v8bf& t1) {
  const uint4 a = __builtin_bit_cast(uint4, f0), b = __builtin_bit_cast(uint4, f1);
  *reinterpret_cast<uint2*>(tw) = make_uint2(a.x, a.y);
  *reinterpret_cast<uint2*>(tw + 16) = make_uint2(a.z, a.w);
  *reinterpret_cast<uint2*>(tw + 32) = make_uint2(b.x, b.y);
  *reinterpret_cast<uint2*>(tw + 48) = make_uint2(b.z, b.w);
  uint2 r[4];
#pragma unroll
  for (int k = 0; k < 4; ++k)
    r[k] = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((am_lds_v4s_ptr)(ttr + AM_ROWB * (16 * (k >> 1) + 8 * (k & 1)))));
  t0 = __builtin_bit_cast(am_v8bf, make_uint4(r[0].x, r[0].y, r[1].x, r[1].y));
  t1 = __builtin_bit_cast(am_v8bf, make_uint4(r[2].x, r[2].y, r[3].x, r[3].y));
}

// Dropout of one probability tile: register i <-> key kbase + (i & 3) + 8 (i >> 2) + 4h of query q (this lane).  Element
// index e(i) = e0 + (i & 3) + 8 (i >> 2), e0 = ((row * H + head) * S + q) * S + kbase + 4h (64-bit).  Returns AND masks.
// The RNG key depends on the high word of the index only: ka / kb = keys of hi0 and hi0 + 1, formed once per work item.
struct AmKeys { unsigned hi0, ka, kb; };
__device__ __forceinline__ AmKeys am_keys(unsigned long long e_first, const AmArgs& a) {
  AmKeys k;
  k.hi0 = (unsigned)(e_first >> 32);
  k.ka = rng_key(a.seed, a.rstream, k.hi0);
  k.kb = rng_key(a.seed, a.rstream, k.hi0 + 1u);
  return k;
}
template <int DROP>
__device__ __forceinline__ void am_drop_masks(unsigned long long e0, const AmKeys& kk, const AmArgs& a, unsigned (&m)[16]) {
  if constexpr (DROP == 1) {
    const unsigned lo = (unsigned)e0;
    // (an item spans S * S <= 2^17 consecutive indices: at most one carry into the high word)
    const unsigned k0 = (unsigned)(e0 >> 32) == kk.hi0 ? kk.ka : kk.kb, k1 = (unsigned)((e0 + 32) >> 32) == kk.hi0 ? kk.ka : kk.kb;
    const unsigned g0 = lo >> 5;
    const unsigned w0 = mix32(g0 ^ k0), w1 = mix32(((g0 + 1u) & 0x07ffffffu) ^ k1);
    const int hs = (int)__builtin_amdgcn_alignbit(w1, w0, lo & 31u);
#pragma unroll
    for (int i = 0; i < 16; ++i) m[i] = am_bitmask(hs, (i & 3) + 8 * (i >> 2));
  } else if constexpr (DROP != 0) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const unsigned long long e = e0 + (unsigned long long)((i & 3) + 8 * (i >> 2));
      m[i] = drop_scale(a.seed, a.rstream, e, a.thresh, 1.f) != 0.f ? 0xffffffffu : 0u;
    }
  }
}

// ---------------------------------------------------------------------------------------------- forward
template <int HD, int DROP>
__global__ void __launch_bounds__(256) k_attn_mfma_fwd(const AmArgs a_) {
  AmArgs a = a_;
  a.seed = live_seed(a_.seed);
  __shared__ __attribute__((aligned(16))) char smem[4 * AM_TILE];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, tl = lane & 31, h = lane >> 5;
  char* tile = smem + wave * AM_TILE;
  const char* trb = tile + AM_ROWB * (4 * h + ((lane & 15) >> 2)) + 32 * ((lane >> 4) & 1) + 8 * (lane & 3);
  constexpr int NS = HD / 16;
  const int S = a.S, H = a.H, C = a.C, ld = 3 * C;
  const int nt = (S + 31) / 32;
  const long long n_items = a.R * H * nt;
  const float c2 = a.scale * 1.4426950408889634f;
  const float keep = DROP ? a.inv_keep : 1.f;
  for (long long item = (long long)blockIdx.x * 4 + wave; item < n_items; item += (long long)gridDim.x * 4) {
    const int qt = (int)(item % nt);
    const long long rh = item / nt;
    const int hd = (int)(rh % H);
    const long long r = rh / H;
    const long long t0 = r * S, t_last = t0 + S - 1;
    const int q = 32 * qt + tl;
    const bool q_ok = q < S;
    const long long tq = t0 + (q_ok ? q : S - 1);
    am_v8bf qf[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) qf[s] = am_load_frag(a.qkv, tq, ld, hd * HD + 16 * s + 8 * h);
    float m = -INFINITY, l = 0.f;
    am_f32x16 ot = am_zero();
    const unsigned long long e_item = ((unsigned long long)rh * S + (unsigned long long)(q_ok ? q : 0)) * (unsigned long long)S;
    AmKeys kk{};
    if constexpr (DROP == 1) kk = am_keys(e_item, a);
    // software pipeline: the K fragments and the V tile of key tile kt + 1 are loaded while tile kt is computed
    am_v8bf kf_n[NS];
    AmStage<HD> vs_n;
#define AM_FWD_LOAD(KT)                                                                               \
    {                                                                                                 \
      const int key_ = 32 * (KT) + tl;                                                                \
      const long long tk_ = t0 + (key_ < S ? key_ : S - 1);                                           \
      _Pragma("unroll") for (int s = 0; s < NS; ++s) kf_n[s] = am_load_frag(a.qkv, tk_, ld, C + hd * HD + 16 * s + 8 * h); \
      am_stage_load<HD>(vs_n, a.qkv, t0 + 32 * (KT), t_last, ld, 2 * C + hd * HD, lane);             \
    }
    AM_FWD_LOAD(0)
    for (int kt = 0; kt < nt; ++kt) {
      am_v8bf kf[NS];
#pragma unroll
      for (int s = 0; s < NS; ++s) kf[s] = kf_n[s];
      am_stage_write<HD>(vs_n, tile, lane);
      if (kt + 1 < nt) AM_FWD_LOAD(kt + 1)
      am_f32x16 st = am_zero();
#pragma unroll
      for (int s = 0; s < NS; ++s) st = AM_MFMA(kf[s], qf[s], st);
      if (32 * kt + 32 > S) {               // last, ragged key tile: keys past the row end never win the softmax
#pragma unroll
        for (int i = 0; i < 16; ++i) st[i] = (32 * kt + (i & 3) + 8 * (i >> 2) + 4 * h) < S ? st[i] : -INFINITY;
      }
      float mx = fmaxf(fmaxf(st[0], st[1]), st[2]);
#pragma unroll
      for (int i = 3; i < 15; i += 2) mx = fmaxf(fmaxf(mx, st[i]), st[i + 1]);
      mx = fmaxf(mx, st[15]);
      mx = fmaxf(mx, am_xor32(mx)) * c2;
      const float mn = fmaxf(m, mx);
      const float corr = __builtin_amdgcn_exp2f(m - mn);
      float ps = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        st[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(st[i], c2, -mn));
        ps += st[i];
      }
      ps += am_xor32(ps);
      l = l * corr + ps;
      m = mn;
      if constexpr (DROP != 0) {
        unsigned dm[16];
        am_drop_masks<DROP>(e_item + (unsigned long long)(32 * kt + 4 * h), kk, a, dm);
#pragma unroll
        for (int i = 0; i < 16; ++i) st[i] = __uint_as_float(__float_as_uint(st[i]) & dm[i]);
      }
      const am_v8bf pf0 = am_pack<0>(st), pf1 = am_pack<1>(st);
#pragma unroll
      for (int i = 0; i < 16; ++i) ot[i] *= corr;
      ot = AM_MFMA(am_tr_frag(trb, 0), pf0, ot);
      ot = AM_MFMA(am_tr_frag(trb, 1), pf1, ot);
    }
    const float sc = keep / l;
    if (q_ok) {
      // O^T[d (rows), q]: register group g holds d = 8g + 4h .. +3 -> one 8-byte store per group with d < HD
      unsigned short* orow = a.out + (t0 + q) * C + hd * HD + 4 * h;
#pragma unroll
      for (int g = 0; g < HD / 8; ++g) {
        typedef __bf16 v4bf __attribute__((ext_vector_type(4)));
        typedef float v4f __attribute__((ext_vector_type(4)));
        v4f v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = ot[4 * g + j] * sc;
        *reinterpret_cast<uint2*>(orow + 8 * g) = __builtin_bit_cast(uint2, __builtin_convertvector(v, v4bf));
      }
      if (h == 0 && a.lse) a.lse[rh * S + q] = (m + __builtin_amdgcn_logf(l)) * 0.6931471805599453f;
    }
  }
}

// ---------------------------------------------------------------------------------------------- backward, common front
// For the tile pair (key tile on the accumulator rows, query tile on the lanes): the normalised probabilities P^T, the
// dropped ones Pd^T and dS^T = P^T (mask . dPd^T / keep - delta) * scale, from fragments the caller loaded.
template <int HD, int DROP>
__device__ __forceinline__ void am_bwd_front(const am_v8bf (&kf)[HD / 16], const am_v8bf (&qf)[HD / 16],
                                             const am_v8bf (&vf)[HD / 16], const am_v8bf (&dof)[HD / 16], float lse2, float delta,
                                             float c2, float scale, float keep, int kbase, int h, int S, unsigned long long e0,
                                             const AmKeys& kk, const AmArgs& a, am_f32x16& pd, am_f32x16& ds) {
  constexpr int NS = HD / 16;
  am_f32x16 st = am_zero(), dp = am_zero();
#pragma unroll
  for (int s = 0; s < NS; ++s) st = AM_MFMA(kf[s], qf[s], st);
#pragma unroll
  for (int s = 0; s < NS; ++s) dp = AM_MFMA(vf[s], dof[s], dp);
#pragma unroll
  for (int i = 0; i < 16; ++i) st[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(st[i], c2, -lse2));     // P^T (lse2 = +inf: invalid query)
  if (kbase + 32 > S) {
#pragma unroll
    for (int i = 0; i < 16; ++i) st[i] = (kbase + (i & 3) + 8 * (i >> 2) + 4 * h) < S ? st[i] : 0.f;
  }
  if constexpr (DROP != 0) {
    unsigned dm[16];
    am_drop_masks<DROP>(e0, kk, a, dm);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      pd[i] = __uint_as_float(__float_as_uint(st[i]) & dm[i]) * keep;
      dp[i] = __uint_as_float(__float_as_uint(dp[i]) & dm[i]) * keep;
    }
  } else {
    pd = st;
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) ds[i] = (st[i] * scale) * (dp[i] - delta);
}

// delta[q] = dO[q, :] . O[q, :] over this head's dims (lane (q, h) holds 8 NS of them; both halves end with the sum)
template <int HD>
__device__ __forceinline__ float am_delta(const am_v8bf (&dof)[HD / 16], const unsigned short* o, long long tq, int C, int col) {
  float d = 0.f;
#pragma unroll
  for (int s = 0; s < HD / 16; ++s) {
    const am_v8bf of = am_load_frag(o, tq, C, col + 16 * s);
#pragma unroll
    for (int j = 0; j < 8; ++j) d += (float)dof[s][j] * (float)of[j];
  }
  return d + am_xor32(d);
}

// ---------------------------------------------------------------------------------------------- backward: dQ
template <int HD, int DROP>
__global__ void __launch_bounds__(256) k_attn_mfma_bwd_dq(const AmArgs a_) {
  AmArgs a = a_;
  a.seed = live_seed(a_.seed);
  __shared__ __attribute__((aligned(16))) char smem[4 * AM_TILE];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, tl = lane & 31, h = lane >> 5;
  char* tile = smem + wave * AM_TILE;
  const char* trb = tile + AM_ROWB * (4 * h + ((lane & 15) >> 2)) + 32 * ((lane >> 4) & 1) + 8 * (lane & 3);
  constexpr int NS = HD / 16;
  const int S = a.S, H = a.H, C = a.C, ld = 3 * C;
  const int nt = (S + 31) / 32;
  const long long n_items = a.R * H * nt;
  const float c2 = a.scale * 1.4426950408889634f;
  const float keep = DROP ? a.inv_keep : 1.f;
  for (long long item = (long long)blockIdx.x * 4 + wave; item < n_items; item += (long long)gridDim.x * 4) {
    const int qt = (int)(item % nt);
    const long long rh = item / nt;
    const int hd = (int)(rh % H);
    const long long r = rh / H;
    const long long t0 = r * S, t_last = t0 + S - 1;
    const int q = 32 * qt + tl;
    const bool q_ok = q < S;
    const long long tq = t0 + (q_ok ? q : S - 1);
    am_v8bf qf[NS], dof[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      qf[s] = am_load_frag(a.qkv, tq, ld, hd * HD + 16 * s + 8 * h);
      dof[s] = am_load_frag(a.dout, tq, C, hd * HD + 16 * s + 8 * h);
    }
    const float delta = am_delta<HD>(dof, a.o, tq, C, hd * HD + 8 * h);
    const float lse2 = q_ok ? a.lse[rh * S + q] * 1.4426950408889634f : INFINITY;
    am_f32x16 dq = am_zero();
    const unsigned long long e_item = ((unsigned long long)rh * S + (unsigned long long)(q_ok ? q : 0)) * (unsigned long long)S;
    AmKeys kk{};
    if constexpr (DROP == 1) kk = am_keys(e_item, a);
    am_v8bf kf_n[NS], vf_n[NS];
    AmStage<HD> ks_n;
#define AM_DQ_LOAD(KT)                                                                                \
    {                                                                                                 \
      const int key_ = 32 * (KT) + tl;                                                                \
      const long long tk_ = t0 + (key_ < S ? key_ : S - 1);                                           \
      _Pragma("unroll") for (int s = 0; s < NS; ++s) {                                                \
        kf_n[s] = am_load_frag(a.qkv, tk_, ld, C + hd * HD + 16 * s + 8 * h);                         \
        vf_n[s] = am_load_frag(a.qkv, tk_, ld, 2 * C + hd * HD + 16 * s + 8 * h);                     \
      }                                                                                               \
      am_stage_load<HD>(ks_n, a.qkv, t0 + 32 * (KT), t_last, ld, C + hd * HD, lane);      /* K tile: K^T fragments */ \
    }
    AM_DQ_LOAD(0)
    for (int kt = 0; kt < nt; ++kt) {
      am_v8bf kf[NS], vf[NS];
#pragma unroll
      for (int s = 0; s < NS; ++s) { kf[s] = kf_n[s]; vf[s] = vf_n[s]; }
      am_stage_write<HD>(ks_n, tile, lane);
      if (kt + 1 < nt) AM_DQ_LOAD(kt + 1)
      am_f32x16 pd, ds;
      am_bwd_front<HD, DROP>(kf, qf, vf, dof, lse2, delta, c2, a.scale, keep, 32 * kt, h, S,
                             e_item + (unsigned long long)(32 * kt + 4 * h), kk, a, pd, ds);
      dq = AM_MFMA(am_tr_frag(trb, 0), am_pack<0>(ds), dq);         // dQ^T[d, q] += K^T[d, key] dS^T[key, q]
      dq = AM_MFMA(am_tr_frag(trb, 1), am_pack<1>(ds), dq);
    }
    if (q_ok) {
      unsigned short* drow = a.dqkv + (t0 + q) * ld + hd * HD + 4 * h;
#pragma unroll
      for (int g = 0; g < HD / 8; ++g) {
        typedef __bf16 v4bf __attribute__((ext_vector_type(4)));
        typedef float v4f __attribute__((ext_vector_type(4)));
        v4f v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = dq[4 * g + j];
        *reinterpret_cast<uint2*>(drow + 8 * g) = __builtin_bit_cast(uint2, __builtin_convertvector(v, v4bf));
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------- backward: dK, dV
template <int HD, int DROP>
__global__ void __launch_bounds__(256) k_attn_mfma_bwd_dkv(const AmArgs a_) {
  AmArgs a = a_;
  a.seed = live_seed(a_.seed);
  __shared__ __attribute__((aligned(16))) char smem[4 * 3 * AM_TILE];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, tl = lane & 31, h = lane >> 5;
  char* tq_tile = smem + wave * 3 * AM_TILE;           // Q tile (row-major), dO tile, transpose tile
  char* tdo_tile = tq_tile + AM_TILE;
  char* tt = tq_tile + 2 * AM_TILE;
  const int troff = AM_ROWB * (4 * h + ((lane & 15) >> 2)) + 32 * ((lane >> 4) & 1) + 8 * (lane & 3);
  char* tw = tt + AM_ROWB * tl + 8 * h;
  const char* ttr = tt + AM_ROWB * (4 * h + ((lane & 15) >> 2)) + 32 * ((lane >> 4) & 1) + 8 * (lane & 3);
  constexpr int NS = HD / 16;
  const int S = a.S, H = a.H, C = a.C, ld = 3 * C;
  const int nt = (S + 31) / 32;
  const long long n_items = a.R * H * nt;
  const float c2 = a.scale * 1.4426950408889634f;
  const float keep = DROP ? a.inv_keep : 1.f;
  for (long long item = (long long)blockIdx.x * 4 + wave; item < n_items; item += (long long)gridDim.x * 4) {
    const int kt = (int)(item % nt);
    const long long rh = item / nt;
    const int hd = (int)(rh % H);
    const long long r = rh / H;
    const long long t0 = r * S, t_last = t0 + S - 1;
    const int key = 32 * kt + tl;
    const long long tk = t0 + (key < S ? key : S - 1);
    am_v8bf kf[NS], vf[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      kf[s] = am_load_frag(a.qkv, tk, ld, C + hd * HD + 16 * s + 8 * h);
      vf[s] = am_load_frag(a.qkv, tk, ld, 2 * C + hd * HD + 16 * s + 8 * h);
    }
    am_f32x16 dk = am_zero(), dv = am_zero();
    AmKeys kk{};
    if constexpr (DROP == 1) kk = am_keys((unsigned long long)rh * S * (unsigned long long)S, a);
    // software pipeline over the query tiles: fragments, tiles, O (for delta) and the log-sum-exp of tile qt + 1 in flight
    am_v8bf qf_n[NS], dof_n[NS], of_n[NS];
    AmStage<HD> qs_n, dos_n;
    float lse_n;
#define AM_DKV_LOAD(QT)                                                                               \
    {                                                                                                 \
      const int q_ = 32 * (QT) + tl;                                                                  \
      const long long tq_ = t0 + (q_ < S ? q_ : S - 1);                                               \
      _Pragma("unroll") for (int s = 0; s < NS; ++s) {                                                \
        qf_n[s] = am_load_frag(a.qkv, tq_, ld, hd * HD + 16 * s + 8 * h);                             \
        dof_n[s] = am_load_frag(a.dout, tq_, C, hd * HD + 16 * s + 8 * h);                            \
        of_n[s] = am_load_frag(a.o, tq_, C, hd * HD + 16 * s + 8 * h);                                \
      }                                                                                               \
      am_stage_load<HD>(qs_n, a.qkv, t0 + 32 * (QT), t_last, ld, hd * HD, lane);          /* Q tile -> Q^T fragments */ \
      am_stage_load<HD>(dos_n, a.dout, t0 + 32 * (QT), t_last, C, hd * HD, lane);         /* dO tile -> dO^T fragments */ \
      lse_n = q_ < S ? a.lse[rh * S + q_] * 1.4426950408889634f : INFINITY;                           \
    }
    AM_DKV_LOAD(0)
    for (int qt = 0; qt < nt; ++qt) {
      const int q = 32 * qt + tl;
      const bool q_ok = q < S;
      am_v8bf qf[NS], dof[NS];
      float delta = 0.f;
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        qf[s] = qf_n[s]; dof[s] = dof_n[s];
#pragma unroll
        for (int j = 0; j < 8; ++j) delta += (float)dof_n[s][j] * (float)of_n[s][j];
      }
      delta += am_xor32(delta);
      const float lse2 = lse_n;
      am_stage_write<HD>(qs_n, tq_tile, lane);
      am_stage_write<HD>(dos_n, tdo_tile, lane);
      if (qt + 1 < nt) AM_DKV_LOAD(qt + 1)
      am_f32x16 pd, ds;
      const unsigned long long e0 = ((unsigned long long)rh * S + (unsigned long long)(q_ok ? q : 0)) * (unsigned long long)S
                                    + (unsigned long long)(32 * kt + 4 * h);
      am_bwd_front<HD, DROP>(kf, qf, vf, dof, lse2, delta, c2, a.scale, keep, 32 * kt, h, S, e0, kk, a, pd, ds);
      am_v8bf p0, p1, s0, s1;
      am_transpose32(am_pack<0>(pd), am_pack<1>(pd), tw, ttr, p0, p1);          // Pd [k = q][col = key]
      am_transpose32(am_pack<0>(ds), am_pack<1>(ds), tw, ttr, s0, s1);          // dS [k = q][col = key]
      dv = AM_MFMA(am_tr_frag(tdo_tile + troff, 0), p0, dv);                    // dV^T[d, key] += dO^T[d, q] Pd[q, key]
      dv = AM_MFMA(am_tr_frag(tdo_tile + troff, 1), p1, dv);
      dk = AM_MFMA(am_tr_frag(tq_tile + troff, 0), s0, dk);                     // dK^T[d, key] += Q^T[d, q] dS[q, key]
      dk = AM_MFMA(am_tr_frag(tq_tile + troff, 1), s1, dk);
    }
    if (key < S) {
      unsigned short* drow = a.dqkv + (t0 + key) * ld + hd * HD + 4 * h;
#pragma unroll
      for (int g = 0; g < HD / 8; ++g) {
        typedef __bf16 v4bf __attribute__((ext_vector_type(4)));
        typedef float v4f __attribute__((ext_vector_type(4)));
        v4f v, w;
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] = dk[4 * g + j]; w[j] = dv[4 * g + j]; }
        *reinterpret_cast<uint2*>(drow + C + 8 * g) = __builtin_bit_cast(uint2, __builtin_convertvector(v, v4bf));
        *reinterpret_cast<uint2*>(drow + 2 * C + 8 * g) = __builtin_bit_cast(uint2, __builtin_convertvector(w, v4bf));
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------- backward, one pass
// dQ, dK and dV from ONE recompute of S^T / P^T / dP^T per (key tile, query tile) pair: a wave owns a whole (row, head),
// walks the key tiles outside and the query tiles inside (as the dKV kernel does for its one key tile), and keeps the
// dQ^T accumulator tiles of ALL query tiles of the row in wave-private LDS (fp32, the HD/2 accumulator registers a lane
// really uses, chunk-major so that lanes are 16 bytes apart): a tile is loaded as the C operand of its two MFMAs and
// written back.  Saves the second recompute (the dQ kernel: 8.9 of 21.5 ms of attention backward on tabgnn-arxiv).
// LDS sets the shapes: HD = 16 with up to 5 query tiles (S <= 160): 4 waves x (4 tiles + 10 KiB of accumulators) = 80 KiB,
// two workgroups per CU — the occupancy the dKV kernel has; HD = 32 with up to 3 query tiles (S <= 96): 3 waves x
// (4 tiles + 12 KiB) = 66 KiB, two workgroups per CU (6 waves instead of 8).  Longer rows keep the two kernels.
template <int HD> constexpr int am_one_ntmax() { return HD == 16 ? 5 : 3; }
template <int HD> constexpr int am_one_waves() { return HD == 16 ? 4 : 3; }
template <int HD> constexpr int am_one_wave_lds() { return 4 * AM_TILE + am_one_ntmax<HD>() * 64 * (HD / 2) * 4; }

template <int HD, int DROP>
__global__ void __launch_bounds__(64 * am_one_waves<HD>()) k_attn_mfma_bwd_one(const AmArgs a_) {
  constexpr int WAVES = am_one_waves<HD>();
  AmArgs a = a_;
  a.seed = live_seed(a_.seed);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int DQC = HD / 8;                          // 16-byte chunks of accumulator registers per lane and tile
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, tl = lane & 31, h = lane >> 5;
  char* tq_tile = smem + wave * am_one_wave_lds<HD>();   // Q tile (row-major), dO tile, transpose tile, K tile, dQ accumulators
  char* tdo_tile = tq_tile + AM_TILE;
  char* tt = tq_tile + 2 * AM_TILE;
  char* tk_tile = tq_tile + 3 * AM_TILE;
  char* dq_acc = tq_tile + 4 * AM_TILE + 16 * lane;      // chunk c of query tile qt: + ((qt * DQC + c) * 64) * 16
  const int troff = AM_ROWB * (4 * h + ((lane & 15) >> 2)) + 32 * ((lane >> 4) & 1) + 8 * (lane & 3);
  char* tw = tt + AM_ROWB * tl + 8 * h;
  const char* ttr = tt + troff;
  constexpr int NS = HD / 16;
  const int S = a.S, H = a.H, C = a.C, ld = 3 * C;
  const int nt = (S + 31) / 32;
  const long long n_items = a.R * H;
  const float c2 = a.scale * 1.4426950408889634f;
  const float keep = DROP ? a.inv_keep : 1.f;
  for (long long rh = (long long)blockIdx.x * WAVES + wave; rh < n_items; rh += (long long)gridDim.x * WAVES) {
    const int hd = (int)(rh % H);
    const long long r = rh / H;
    const long long t0 = r * S, t_last = t0 + S - 1;
    AmKeys kk{};
    if constexpr (DROP == 1) kk = am_keys((unsigned long long)rh * S * (unsigned long long)S, a);
    for (int kt = 0; kt < nt; ++kt) {
      const int key = 32 * kt + tl;
      const long long tk = t0 + (key < S ? key : S - 1);
      am_v8bf kf[NS], vf[NS];
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        kf[s] = am_load_frag(a.qkv, tk, ld, C + hd * HD + 16 * s + 8 * h);
        vf[s] = am_load_frag(a.qkv, tk, ld, 2 * C + hd * HD + 16 * s + 8 * h);
      }
      {
        AmStage<HD> ks;                                                          // K tile -> K^T fragments of the dQ product
        am_stage_load<HD>(ks, a.qkv, t0 + 32 * kt, t_last, ld, C + hd * HD, lane);
        am_stage_write<HD>(ks, tk_tile, lane);
      }
      am_f32x16 dk = am_zero(), dv = am_zero();
      am_v8bf qf_n[NS], dof_n[NS], of_n[NS];
      AmStage<HD> qs_n, dos_n;
      float lse_n;
#define AM_ONE_LOAD(QT)                                                                               \
      {                                                                                               \
        const int q_ = 32 * (QT) + tl;                                                                \
        const long long tq_ = t0 + (q_ < S ? q_ : S - 1);                                             \
        _Pragma("unroll") for (int s = 0; s < NS; ++s) {                                              \
          qf_n[s] = am_load_frag(a.qkv, tq_, ld, hd * HD + 16 * s + 8 * h);                           \
          dof_n[s] = am_load_frag(a.dout, tq_, C, hd * HD + 16 * s + 8 * h);                          \
          of_n[s] = am_load_frag(a.o, tq_, C, hd * HD + 16 * s + 8 * h);                              \
        }                                                                                             \
        am_stage_load<HD>(qs_n, a.qkv, t0 + 32 * (QT), t_last, ld, hd * HD, lane);                    \
        am_stage_load<HD>(dos_n, a.dout, t0 + 32 * (QT), t_last, C, hd * HD, lane);                   \
        lse_n = q_ < S ? a.lse[rh * S + q_] * 1.4426950408889634f : INFINITY;                         \
      }
      AM_ONE_LOAD(0)
      for (int qt = 0; qt < nt; ++qt) {
        const int q = 32 * qt + tl;
        const bool q_ok = q < S;
        am_v8bf qf[NS], dof[NS];
        float delta = 0.f;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          qf[s] = qf_n[s]; dof[s] = dof_n[s];
#pragma unroll
          for (int j = 0; j < 8; ++j) delta += (float)dof_n[s][j] * (float)of_n[s][j];
        }
        delta += am_xor32(delta);
        const float lse2 = lse_n;
        am_stage_write<HD>(qs_n, tq_tile, lane);
        am_stage_write<HD>(dos_n, tdo_tile, lane);
        if (qt + 1 < nt) AM_ONE_LOAD(qt + 1)
        am_f32x16 pd, ds;
        const unsigned long long e0 = ((unsigned long long)rh * S + (unsigned long long)(q_ok ? q : 0)) * (unsigned long long)S
                                      + (unsigned long long)(32 * kt + 4 * h);
        am_bwd_front<HD, DROP>(kf, qf, vf, dof, lse2, delta, c2, a.scale, keep, 32 * kt, h, S, e0, kk, a, pd, ds);
        // dQ^T[d, q] += K^T[d, key] dS^T[key, q]: the tile's accumulator lives in LDS between key tiles
        am_f32x16 dq = am_zero();
        char* dqp = dq_acc + qt * (DQC * 64 * 16);
        if (kt > 0) {
#pragma unroll
          for (int c = 0; c < DQC; ++c) {
            const float4 v = *reinterpret_cast<const float4*>(dqp + c * 64 * 16);
            dq[4 * c] = v.x; dq[4 * c + 1] = v.y; dq[4 * c + 2] = v.z; dq[4 * c + 3] = v.w;
          }
        }
        dq = AM_MFMA(am_tr_frag(tk_tile + troff, 0), am_pack<0>(ds), dq);
        dq = AM_MFMA(am_tr_frag(tk_tile + troff, 1), am_pack<1>(ds), dq);
#pragma unroll
        for (int c = 0; c < DQC; ++c)
          *reinterpret_cast<float4*>(dqp + c * 64 * 16) = make_float4(dq[4 * c], dq[4 * c + 1], dq[4 * c + 2], dq[4 * c + 3]);
        am_v8bf p0, p1, s0, s1;
        am_transpose32(am_pack<0>(pd), am_pack<1>(pd), tw, ttr, p0, p1);          // Pd [k = q][col = key]
        am_transpose32(am_pack<0>(ds), am_pack<1>(ds), tw, ttr, s0, s1);          // dS [k = q][col = key]
        dv = AM_MFMA(am_tr_frag(tdo_tile + troff, 0), p0, dv);                    // dV^T[d, key] += dO^T[d, q] Pd[q, key]
        dv = AM_MFMA(am_tr_frag(tdo_tile + troff, 1), p1, dv);
        dk = AM_MFMA(am_tr_frag(tq_tile + troff, 0), s0, dk);                     // dK^T[d, key] += Q^T[d, q] dS[q, key]
        dk = AM_MFMA(am_tr_frag(tq_tile + troff, 1), s1, dk);
      }
#undef AM_ONE_LOAD
      if (key < S) {
        unsigned short* drow = a.dqkv + (t0 + key) * ld + hd * HD + 4 * h;
#pragma unroll
        for (int g = 0; g < HD / 8; ++g) {
          typedef __bf16 v4bf __attribute__((ext_vector_type(4)));
          typedef float v4f __attribute__((ext_vector_type(4)));
          v4f v, w;
#pragma unroll
          for (int j = 0; j < 4; ++j) { v[j] = dk[4 * g + j]; w[j] = dv[4 * g + j]; }
          *reinterpret_cast<uint2*>(drow + C + 8 * g) = __builtin_bit_cast(uint2, __builtin_convertvector(v, v4bf));
          *reinterpret_cast<uint2*>(drow + 2 * C + 8 * g) = __builtin_bit_cast(uint2, __builtin_convertvector(w, v4bf));
        }
      }
    }
    for (int qt = 0; qt < nt; ++qt) {                    // the row's dQ tiles: fp32 accumulators -> bf16 rows
      const int q = 32 * qt + tl;
      if (q < S) {
        unsigned short* drow = a.dqkv + (t0 + q) * ld + hd * HD + 4 * h;
        const char* dqp = dq_acc + qt * (DQC * 64 * 16);
#pragma unroll
        for (int g = 0; g < DQC; ++g) {
          typedef __bf16 v4bf __attribute__((ext_vector_type(4)));
          typedef float v4f __attribute__((ext_vector_type(4)));
          const float4 t = *reinterpret_cast<const float4*>(dqp + g * 64 * 16);
          v4f v;
          v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
          *reinterpret_cast<uint2*>(drow + 8 * g) = __builtin_bit_cast(uint2, __builtin_convertvector(v, v4bf));
        }
      }
    }
  }
}

template <int HD, int DROP> static void am_launch(const AmArgs& a, int which, hipStream_t st) {
  const int nt = (a.S + 31) / 32;
  const long long items = a.R * a.H * nt;
  long long blocks = (items + 3) / 4;
  if (blocks > 256 * 16) blocks = 256 * 16;
  if (which == 0) hipLaunchKernelGGL((k_attn_mfma_fwd<HD, DROP>), dim3((unsigned)blocks), dim3(256), 0, st, a);
  else if (which == 1) hipLaunchKernelGGL((k_attn_mfma_bwd_dq<HD, DROP>), dim3((unsigned)blocks), dim3(256), 0, st, a);
  else hipLaunchKernelGGL((k_attn_mfma_bwd_dkv<HD, DROP>), dim3((unsigned)blocks), dim3(256), 0, st, a);
}
template <int HD, int DROP> static void am_launch_one(const AmArgs& a, hipStream_t st) {
  constexpr int WAVES = am_one_waves<HD>();
  const int lds = WAVES * am_one_wave_lds<HD>();
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_attn_mfma_bwd_one<HD, DROP>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr = true;
  }
  const long long items = a.R * a.H;
  long long blocks = (items + WAVES - 1) / WAVES;
  if (blocks > 256 * 2) blocks = 256 * 2;          // two resident workgroups per CU, every wave walks its (row, head) items
  hipLaunchKernelGGL((k_attn_mfma_bwd_one<HD, DROP>), dim3((unsigned)blocks), dim3(64 * WAVES), lds, st, a);
}
template <int HD> static void am_launch_one_d(const AmArgs& a, hipStream_t st) {
  switch (drop_mode(a.thresh)) {
    case 0: am_launch_one<HD, 0>(a, st); break;
    case 1: am_launch_one<HD, 1>(a, st); break;
    case 8: am_launch_one<HD, 8>(a, st); break;
    default: am_launch_one<HD, 16>(a, st); break;
  }
}
template <int HD> static void am_launch_d(const AmArgs& a, int which, hipStream_t st) {
  switch (drop_mode(a.thresh)) {
    case 0: am_launch<HD, 0>(a, which, st); break;
    case 1: am_launch<HD, 1>(a, which, st); break;
    case 8: am_launch<HD, 8>(a, which, st); break;
    default: am_launch<HD, 16>(a, which, st); break;
  }
}

// bf16, head dim 16 / 32, 16-byte aligned rows: the MFMA kernels take the call (attention.hip asks)
bool attn_mfma_ok(int32_t S, int32_t C, int32_t H, int32_t dt) {
  const bool off = getenv("TABGNN_NO_MFMA_ATTN") != nullptr;          // same-box A/B switch (read per call: tests flip it)
  const int hd = H > 0 ? C / H : 0;
  return !off && dt == BF16 && (hd == 16 || hd == 32) && C % 8 == 0 && S > 8;
}
void attn_mfma_fwd(const void* qkv, void* out, float* lse, int64_t R, int32_t S, int32_t C, int32_t H, float scale,
                   unsigned thresh, float inv_keep, uint64_t seed, uint32_t rstream, hipStream_t st) {
  AmArgs a{};
  a.qkv = (const unsigned short*)qkv; a.out = (unsigned short*)out; a.lse = lse; a.R = R; a.S = S; a.H = H; a.C = C;
  a.scale = scale; a.thresh = thresh; a.inv_keep = inv_keep; a.seed = seed; a.rstream = rstream;
  if (C / H == 16) am_launch_d<16>(a, 0, st); else am_launch_d<32>(a, 0, st);
}
void attn_mfma_bwd(const void* qkv, const void* o, const void* dout, const float* lse, void* dqkv, int64_t R, int32_t S,
                   int32_t C, int32_t H, float scale, unsigned thresh, float inv_keep, uint64_t seed, uint32_t rstream,
                   hipStream_t st) {
  AmArgs a{};
  a.qkv = (const unsigned short*)qkv; a.o = (const unsigned short*)o; a.dout = (const unsigned short*)dout;
  a.lse = const_cast<float*>(lse); a.dqkv = (unsigned short*)dqkv; a.R = R; a.S = S; a.H = H; a.C = C;
  a.scale = scale; a.thresh = thresh; a.inv_keep = inv_keep; a.seed = seed; a.rstream = rstream;
  static const bool two_pass = getenv("TABGNN_ATTN_TWO_PASS") != nullptr;     // same-box A/B switch
  if (C / H == 16 && S <= 32 * am_one_ntmax<16>() && !two_pass) am_launch_one_d<16>(a, st);      // dQ, dK, dV from one recompute
  // (HD = 32 with three-wave workgroups was measured on wide64-c256: 34.7 -> 34.5 ms/step — with dropout the kernel needs
  //  > 128 registers and LDS for 6 waves per CU at most, which eats what the saved recompute gives; those rows keep two kernels)
  else if (C / H == 16) { am_launch_d<16>(a, 1, st); am_launch_d<16>(a, 2, st); }
  else { am_launch_d<32>(a, 1, st); am_launch_d<32>(a, 2, st); }
}

}  // namespace tg

