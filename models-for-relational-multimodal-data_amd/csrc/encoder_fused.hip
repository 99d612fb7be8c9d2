// The column-transformer layer of the fused path as ONE kernel per direction (bf16, d_model = feed-forward = 128):
//     torch nn.TransformerEncoderLayer(d_model=C, nhead, dim_feedforward=C, dropout, relu, batch_first), post-norm,
//     as configured at src/nn/models/fused.py:83-92,187-196 and called at fused.py:160,164,249 / tabgnn.py:127-129,219,
//     followed by the tab_norm LayerNorm + residual combine every call site applies:
//         out = alpha * x + beta_c * LN_t(enc(x))                 (or out = enc(x))
//
// Design (MI355X, wave64, MFMA 32x32x16 bf16):
//   * a WAVE owns 32 consecutive token slots = floor(32/S) whole table rows (S = columns + CLS: 6 for AML); nothing a
//     wave computes depends on another wave's tokens, so the whole layer — QKV projection, S x S attention, output
//     projection, LayerNorm, feed-forward, LayerNorm, tail LayerNorm — runs on that wave's REGISTERS:
//       - every GEMM is out^T[n, t] = W[n, :] . act[t, :] with A := W tile (from LDS), B := activations of the wave's 32
//         tokens; the result has the token on the LANE (column) and 64 channels of that token in the lane's registers;
//       - an accumulator tile converted to bf16 IS the B operand of the next MFMA (k = accumulator row), so q/k/v, the
//         softmax probabilities, the attention output, x1 and the feed-forward hidden state never leave registers; the
//         k order inside a 16-deep MFMA step is the accumulator's row order (16s + 8(j>>2) + 4h + (j&3)), so the weight
//         images are stored in LDS with their k axis permuted the same way (done once per call by k_encoder_pack) and
//         x itself is loaded from HBM straight into that fragment order;
//       - attention per head: S^T = K Q^T (keys on accumulator rows), softmax down the rows in registers (one xor-32
//         exchange), P^T back in as the B operand of O^T = V^T P^T — the block-diagonal structure (a query only sees
//         the keys of its own table row) is a mask on the 32 x 32 tile;
//       - LayerNorm is a reduction over the lane's 64 registers plus one xor-32 exchange.
//   * the only LDS traffic is weights: six 32 KiB stages per group of eight wave tiles (per 32-channel head block:
//     Wq | Wk | Wv rows + the matching k slice of Wo; then W1; then W2), double-buffered, filled by LDS-DMA
//     (global_load_lds 16 B) from a pre-swizzled image, one barrier per stage;
//   * HBM: x is read once (fragment order, 8-byte pieces of whole rows), out / z1 / z2 are written once through a small
//     wave-private LDS restage as whole 128-byte row segments.  qkv, scores, o, x1, h never exist in memory.
// Dropout masks are the package's counter RNG on the same element indices the unfused kernels use (attention
// probabilities: ((row*H + head)*S + q)*S + k; linear outputs: token*128 + channel), so either path can run the backward.
#include "common.hpp"
#include "../../include/tabgnn_hip.h"

namespace tg {

typedef __bf16 ef_v8bf __attribute__((ext_vector_type(8)));
typedef __bf16 ef_v2bf __attribute__((ext_vector_type(2)));
typedef float ef_f32x16 __attribute__((ext_vector_type(16)));
typedef float ef_f32x8 __attribute__((ext_vector_type(8)));
typedef float ef_f2 __attribute__((ext_vector_type(2)));
typedef unsigned ef_u4 __attribute__((ext_vector_type(4)));

constexpr int EF_C = 128;
// LDS weight images: rows of 128 bf16 (256 B) PADDED to 272 B.  A lane's fragment address is then
//     272 * (lane & 31) + 16 * (lane >> 5)  +  a compile-time constant (32 * k-step + 272 * row block + part / buffer offset)
// i.e. ONE lane-constant base register and the instruction's immediate offset (round 2 used 256-B rows with an XOR
// swizzle: ~3 integer instructions per ds_read_b128, 1 300 per forward tile), and the 16-lane groups of a ds_read_b128
// still cover all 64 banks exactly once (68 dwords per row = 4 mod 64).  The images are built in this layout in global
// memory by the pack kernels, so the LDS-DMA (a linear byte copy) needs no address arithmetic either.
constexpr int EF_ROWB = 272;
constexpr int EF_PART_BYTES = 32 * EF_ROWB;           // 32 rows: one head block of Wq / Wk / Wv / Wo^T
constexpr int EF_UNIT_BYTES = 2 * EF_PART_BYTES;      // 64 rows: half a 128 x 128 tile (17 KiB)
constexpr int EF_STAGE_BYTES = 4 * EF_PART_BYTES;     // 128 rows (34 KiB)
#ifndef EF_WAVES_N
#define EF_WAVES_N 4
#endif
#ifndef EF_WG_PER_CU
#define EF_WG_PER_CU 2       // resident workgroups per CU the kernels are built for (waves per SIMD with EF_WAVES_N = 4)
#endif
// Geometry: EF_WAVES_N waves per workgroup.  4 waves + ONE stage buffer (72 KiB of LDS) lets two workgroups share
// a CU: their barriers, stage DMAs and input loads overlap each other (measured against 8 waves + two buffers, one
// workgroup per CU, whose waves all stall together).  EF_DBUF = 1: double-buffered stages (needs EF_WAVES_N = 8 to pay).
constexpr int EF_WAVES = EF_WAVES_N;         // waves per workgroup
constexpr int EF_THREADS = 64 * EF_WAVES;
constexpr int EF_NSTAGE = 7;              // 4 head blocks (Wq|Wk|Wv rows), Wo, W1, W2
constexpr int EF_STG_ROWB = 144;          // output restage: 128-byte half rows padded to 144 B (one base + immediates)
// fp32 parameter block (floats): b_in[384] | b_o | g1 | be1 | b1 | b2 | g2 | be2 | gt | bt  (128 each)
enum { EF_P_BIN = 0, EF_P_BO = 384, EF_P_G1 = 512, EF_P_BE1 = 640, EF_P_B1 = 768, EF_P_B2 = 896, EF_P_G2 = 1024,
       EF_P_BE2 = 1152, EF_P_GT = 1280, EF_P_BT = 1408, EF_P_FLOATS = 1536 };

__device__ __forceinline__ int ef_off(int row, int ch) {      // padded rows, 16-byte chunk ch (0..15)
  return EF_ROWB * row + 16 * ch;
}

// ---------------------------------------------------------------------------------------------- weight pack
// wpack: [7 stages][34 KiB] bf16 LDS images, prm: EF_P_FLOATS floats.  Data chunk c = 2*ks + h of a row holds channels
// {16ks + 4h + 0..3, 16ks + 8 + 4h + 0..3}: the accumulator row order of MFMA k-step ks, lane half h.
// TRANSPOSED variant (backward): stage s holds W^T tiles laid out the same way (see k_encoder_pack_tiles below).
__device__ __forceinline__ uint4 ef_perm_chunk(const unsigned short* wrow, int ks, int h) {
  const uint2 a = *reinterpret_cast<const uint2*>(wrow + 16 * ks + 4 * h);
  const uint2 b = *reinterpret_cast<const uint2*>(wrow + 16 * ks + 8 + 4 * h);
  return make_uint4(a.x, a.y, b.x, b.y);
}
// one padded image row: 16 data chunks + the 16-byte pad (zero)
__device__ __forceinline__ void ef_pack_row(char* dst_row, const unsigned short* wrow, int c /*0..16*/) {
  *reinterpret_cast<uint4*>(dst_row + 16 * c) = c < 16 ? ef_perm_chunk(wrow, c >> 1, c & 1) : make_uint4(0u, 0u, 0u, 0u);
}

__global__ void __launch_bounds__(256) k_encoder_pack(const unsigned short* __restrict__ w_in,   // [384,128]
                                                       const unsigned short* __restrict__ w_o,    // [128,128]
                                                       const unsigned short* __restrict__ w1,
                                                       const unsigned short* __restrict__ w2,
                                                       const float* __restrict__ b_in, const float* __restrict__ b_o,
                                                       const float* __restrict__ g1, const float* __restrict__ be1,
                                                       const float* __restrict__ b1, const float* __restrict__ b2,
                                                       const float* __restrict__ g2, const float* __restrict__ be2,
                                                       const float* __restrict__ gt, const float* __restrict__ bt,
                                                       char* __restrict__ wpack, float* __restrict__ prm) {
  const int stage = blockIdx.x;                // 0..6: four head blocks, Wo, W1, W2 -> two 17 KiB units each
  char* dst = wpack + (size_t)stage * EF_STAGE_BYTES;
  if (stage < 4) {
    // unit 2*stage: Wq rows | Wk rows (one part each); unit 2*stage + 1: Wv rows (+ one part of zeros)
    for (int p = threadIdx.x; p < 128 * 17; p += blockDim.x) {
      const int row = p / 17, c = p - 17 * row, part = row >> 5, r = row & 31;
      if (part < 3) ef_pack_row(dst + EF_PART_BYTES * part + EF_ROWB * r, w_in + (size_t)(128 * part + 32 * stage + r) * EF_C, c);
      else *reinterpret_cast<uint4*>(dst + EF_PART_BYTES * part + EF_ROWB * r + 16 * c) = make_uint4(0u, 0u, 0u, 0u);
    }
  } else {
    // rows 0..63 -> first unit, 64..127 -> second
    const unsigned short* w = stage == 4 ? w_o : stage == 5 ? w1 : w2;
    for (int p = threadIdx.x; p < 128 * 17; p += blockDim.x) {
      const int row = p / 17, c = p - 17 * row;
      ef_pack_row(dst + EF_ROWB * row, w + (size_t)row * EF_C, c);
    }
  }
  if (stage == 0) {
    for (int i = threadIdx.x; i < EF_P_FLOATS; i += blockDim.x) {
      float v;
      if (i < 384) v = b_in[i];
      else {
        const int k = (i - 384) >> 7, j = i & 127;
        const float* src = k == 0 ? b_o : k == 1 ? g1 : k == 2 ? be1 : k == 3 ? b1 : k == 4 ? b2 : k == 5 ? g2 : k == 6 ? be2
                         : k == 7 ? gt : bt;
        v = src ? src[j] : (k == 7 ? 1.f : 0.f);
      }
      prm[i] = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------- device helpers
__device__ __forceinline__ ef_v8bf ef_frag(const char* tile, int off) {
  return __builtin_bit_cast(ef_v8bf, *reinterpret_cast<const uint4*>(tile + off));
}
template <int S8> __device__ __forceinline__ ef_v8bf ef_pack(const ef_f32x16& a) {   // regs 8*S8 .. 8*S8+7 -> 8 bf16
  ef_f32x8 t;
#pragma unroll
  for (int j = 0; j < 8; ++j) t[j] = a[8 * S8 + j];
  return __builtin_convertvector(t, ef_v8bf);
}
__device__ __forceinline__ float ef_bf(ef_v8bf v, int j) { return (float)v[j]; }
__device__ __forceinline__ float ef_xor32(float v) { return __shfl_xor(v, 32, 64); }
__device__ __forceinline__ ef_f32x16 ef_zero16() {
  ef_f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.f;
  return z;
}
// two bf16 of one dword <-> two floats (v_lshlrev / v_and; v_cvt_pk_bf16_f32)
__device__ __forceinline__ ef_f2 ef_unpk(unsigned w) {
  ef_f2 r;
  r.x = __uint_as_float(w << 16);
  r.y = __uint_as_float(w & 0xffff0000u);
  return r;
}
__device__ __forceinline__ unsigned ef_pk(ef_f2 v) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, ef_v2bf));
}
__device__ __forceinline__ ef_f2 ef_splat(float a) { ef_f2 r; r.x = a; r.y = a; return r; }
__device__ __forceinline__ ef_f2 ef_fma2(ef_f2 a, ef_f2 b, ef_f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ unsigned ef_dw(const ef_v8bf& v, int d) { return __builtin_bit_cast(uint4, v)[d]; }
// all-ones when bit `bit` of hs is set, else 0 (v_bfe_i32); opaque, or the AND it feeds becomes a compare + select
__device__ __forceinline__ unsigned ef_bitmask(int hs, int bit) {
  unsigned m = (unsigned)__builtin_amdgcn_sbfe(hs, bit, 1);
  asm("" : "+v"(m));
  return m;
}
#ifndef EF_ABL
#define EF_ABL 0
#endif
// (A/B switches for tools/encoder_ablate.sh; 0 = the shipped kernel)
#define EF_FENCE() do { if (!(EF_ABL & 64)) __builtin_amdgcn_sched_barrier(0); } while (0)
#define EF_MID_FENCE() do { if (!(EF_ABL & 1)) EF_FENCE(); } while (0)
#if EF_ABL & 256      // timing-only build: no matrix instructions (operands kept alive)
__device__ __forceinline__ ef_f32x16 ef_no_mfma(ef_v8bf a, ef_v8bf b, ef_f32x16 c) {
  asm volatile("" :: "v"(a), "v"(b));
  return c;
}
#define EF_MFMA(A, B, C) ef_no_mfma((A), (B), (C))
#else
#define EF_MFMA(A, B, C) __builtin_amdgcn_mfma_f32_32x32x16_bf16((A), (B), (C), 0, 0, 0)
#endif

// LDS-DMA of BYTES (rounded up to 1 KiB pieces: the images are padded accordingly) by the workgroup: one 1 KiB piece per
// wave-instruction; a wave takes a CONTIGUOUS run of pieces (17 pieces over 4 waves: 5 | 4 | 4 | 4).
// INLINE ASM on purpose: for the builtin (__builtin_amdgcn_global_load_lds) hipcc assumes that every later LDS access
// may alias the DMA's destination and puts `s_waitcnt vmcnt(0)` in front of the next ds_read / ds_write — i.e. the wave
// stalled for the full DMA latency right after issuing it, at every one of the 14 unit boundaries of a tile (the
// "prefetch" never ran ahead).  Through asm the compiler sees no LDS write; the landing is awaited by the counted
// `s_waitcnt vmcnt(K)` + barrier of the unit boundary (EF_UNIT_NEXT_K), and the "memory" clobber keeps the compiler from
// moving or merging LDS reads across the statement.
// M0 (the LDS base of a DMA) is written ONCE per call, to the middle of the wave's run; the pieces are addressed by the
// instruction's immediate offset (-2048 .. +2048), which the hardware adds to the LDS address AND to the global address
// (tools/hwtests/lds_dma_offset.hip) — so the scalar base points at the same middle.  M0 is read when the instruction
// issues: tools/hwtests/lds_dma_race.hip part B rewrites M0 0 / 8 / 32 wait states behind a DMA under full load, 5.2e8
// pieces, none lands at the new value — no wait states are needed behind a DMA, and nothing has to be restored (hipcc
// does not use M0 in this file: tests/test_cabi_and_host.py checks the compiled assembly).
// `s_nop 4` in front of the first DMA: hipcc may reload a spilled address pair with v_readlane right in front of the
// statement, and a VALU-written SGPR needs 5 wait states before a vector-memory instruction reads it — the hazard
// recogniser does not look inside inline asm (round 4: the dropout forward with its spilled SGPRs faulted without it).
// Round 4 shipped this helper with `s_waitcnt vmcnt(0)` between a wave's pieces and a late start of half the grid
// against "one wave tile wrong at percent rates"; the cause was not the DMA at all but LDS reads in flight across the
// unit barrier (EF_WAIT_VM below) — with that closed the pieces issue back to back again, in every grid.
#ifndef EF_DMA_FORM
#define EF_DMA_FORM 1        // 0: the round-3/4 form (M0 per piece, pieces strided over the waves) for the A/B probe only
#endif
#define EF_GLDS(OFF) "global_load_lds_dwordx4 %0, %1 offset:" #OFF "\n\t"
template <int BYTES>
__device__ __forceinline__ void ef_dma(const char* __restrict__ src, char* lds_dst, int wave, int lane16) {
  constexpr int NP = (BYTES + 1023) / 1024;
  unsigned l16 = (unsigned)lane16;
  asm volatile("" : "+v"(l16));      // opaque: not hoisted out of the persistent loop as 64-bit address pairs
  const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)lds_dst;
#if EF_DMA_FORM == 1
  constexpr int BASE = NP / EF_WAVES, REM = NP % EF_WAVES;     // waves below REM issue BASE + 1 pieces
  static_assert(BASE == 1 || BASE == 2 || BASE == 4, "piece runs of 1, 2 or 4 (+1)");
  const int first = wave * BASE + (wave < REM ? wave : REM);   // wave-uniform
  const char* g = src + first * 1024 + 2048;
  const unsigned m0v = lds0 + (unsigned)(first * 1024 + 2048);
  if constexpr (BASE == 4)
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 4\n\t" EF_GLDS(-2048) EF_GLDS(-1024) EF_GLDS(0) EF_GLDS(1024) :: "v"(l16), "s"(g), "s"(m0v) : "memory");
  else if constexpr (BASE == 2)
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 4\n\t" EF_GLDS(-2048) EF_GLDS(-1024) :: "v"(l16), "s"(g), "s"(m0v) : "memory");
  else
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 4\n\t" EF_GLDS(-2048) :: "v"(l16), "s"(g), "s"(m0v) : "memory");
  if (REM != 0 && wave < REM) {      // (M0 written again: this is a statement of its own)
    if constexpr (BASE == 4) asm volatile("s_mov_b32 m0, %2\n\ts_nop 4\n\t" EF_GLDS(2048) :: "v"(l16), "s"(g), "s"(m0v) : "memory");
    else if constexpr (BASE == 2) asm volatile("s_mov_b32 m0, %2\n\ts_nop 4\n\t" EF_GLDS(0) :: "v"(l16), "s"(g), "s"(m0v) : "memory");
    else asm volatile("s_mov_b32 m0, %2\n\ts_nop 4\n\t" EF_GLDS(-1024) :: "v"(l16), "s"(g), "s"(m0v) : "memory");
  }
#else
#pragma unroll
  for (int p = 0; p < (NP + EF_WAVES - 1) / EF_WAVES; ++p) {
    const int q = p * EF_WAVES + wave;                         // wave-uniform piece
    if ((p + 1) * EF_WAVES <= NP || q < NP) {
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep)
                   : "v"(l16), "s"(src + q * 1024), "s"(lds0 + (unsigned)(q * 1024))
                   : "memory");
    }
  }
#endif
}
// (the DMA is inline asm: the compiler does not wait for it at a __syncthreads(), EF_DMA_LANDED() does)
#define EF_DMA_LANDED() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")

// mean / rstd of a token's 128 channels held as 8 packed fragments (this lane's 64 + the xor-32 partner's 64): ONE pass
// over the bf16-rounded values (sum and sum of squares, packed fp32 math: one v_pk_add + one v_pk_fma per dword)
__device__ __forceinline__ void ef_row_stats(const ef_v8bf (&zp)[8], float eps, float& mu, float& rstd) {
  if (EF_ABL & 512) { mu = (float)zp[0][0]; rstd = 1.f + eps; return; }
  ef_f2 s1 = ef_splat(0.f), s2 = ef_splat(0.f);
#pragma unroll
  for (int f = 0; f < 8; ++f)
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const ef_f2 v = ef_unpk(ef_dw(zp[f], d));
      s1 += v;
      s2 = ef_fma2(v, v, s2);
    }
  float sum = s1.x + s1.y, sq = s2.x + s2.y;
  sum += ef_xor32(sum);
  sq += ef_xor32(sq);
  mu = sum * (1.f / 128.f);
  const float var = fmaxf(sq * (1.f / 128.f) - mu * mu, 0.f);
  rstd = rsqrtf(var + eps);
}
// (z - mu) * rstd * gamma + beta for one fragment; gp / bp: LDS addresses of the fragment's first float4 (channels
// 16f + 4h ..), the second float4 sits 8 floats further (channels 16f + 8 + 4h ..).  nmr = -mu * rstd.
__device__ __forceinline__ ef_v8bf ef_ln_apply(ef_v8bf z, float nmr, float rstd, const float* gp, const float* bp) {
  const float4 g0 = *reinterpret_cast<const float4*>(gp), g1 = *reinterpret_cast<const float4*>(gp + 8);
  const float4 b0 = *reinterpret_cast<const float4*>(bp), b1 = *reinterpret_cast<const float4*>(bp + 8);
  const ef_f2 gg[4] = {{g0.x, g0.y}, {g0.z, g0.w}, {g1.x, g1.y}, {g1.z, g1.w}};
  const ef_f2 bb[4] = {{b0.x, b0.y}, {b0.z, b0.w}, {b1.x, b1.y}, {b1.z, b1.w}};
  uint4 o;
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    const ef_f2 t = ef_fma2(ef_unpk(ef_dw(z, d)), ef_splat(rstd), ef_splat(nmr));
    o[d] = ef_pk(ef_fma2(t, gg[d], bb[d]));
  }
  return __builtin_bit_cast(ef_v8bf, o);
}
__device__ __forceinline__ ef_v8bf ef_ln_combine(ef_v8bf z, ef_v8bf x, float nmr, float rstd, const float* gp,
                                                 const float* bp, float alpha, float beta_c) {
  const float4 g0 = *reinterpret_cast<const float4*>(gp), g1 = *reinterpret_cast<const float4*>(gp + 8);
  const float4 b0 = *reinterpret_cast<const float4*>(bp), b1 = *reinterpret_cast<const float4*>(bp + 8);
  const ef_f2 gg[4] = {{g0.x, g0.y}, {g0.z, g0.w}, {g1.x, g1.y}, {g1.z, g1.w}};
  const ef_f2 bb[4] = {{b0.x, b0.y}, {b0.z, b0.w}, {b1.x, b1.y}, {b1.z, b1.w}};
  uint4 o;
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    const ef_f2 t = ef_fma2(ef_unpk(ef_dw(z, d)), ef_splat(rstd), ef_splat(nmr));
    const ef_f2 v = ef_fma2(t, gg[d], bb[d]) * ef_splat(beta_c);
    o[d] = ef_pk(ef_fma2(ef_unpk(ef_dw(x, d)), ef_splat(alpha), v));
  }
  return __builtin_bit_cast(ef_v8bf, o);
}
// Rows of 128 channels held as 8 packed fragments -> memory, through the wave-private restage (64 channels at a time,
// 144-byte rows): every lane then stores 16 bytes of a 128-byte row segment.  dst = first token of the wave tile.
//   sw = stg + 144 * tl + 8 * h (this lane's write base), sr = stg + 144 * (lane >> 3) + 16 * (lane & 7) (read base),
//   go = 256 * (lane >> 3) + 16 * (lane & 7) (byte offset of the lane's first 16-byte piece from dst), t0 = lane >> 3.
struct EfLaneAddr { char* sw; const char* sr; unsigned go; };
// buffer descriptor of the nvalid * 256 bytes of a [T,128] bf16 tensor that belong to this wave tile (base and size are
// wave-uniform): loads of rows past the end return zeros, stores to them are dropped — no exec masking, no branches
__device__ __forceinline__ __amdgpu_buffer_rsrc_t ef_tile_rsrc(const unsigned short* base, long long tok0, int nvalid) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(base + tok0 * EF_C), 0, nvalid * (EF_C * 2), 0x00020000);
}
__device__ __forceinline__ void ef_store_rows(const ef_v8bf (&zp)[8], const EfLaneAddr& la, __amdgpu_buffer_rsrc_t dst) {
#pragma unroll
  for (int half = 0; half < 2; ++half) {
#pragma unroll
    for (int ff = 0; ff < 4; ++ff) {
      const uint4 v = __builtin_bit_cast(uint4, zp[4 * half + ff]);
      // fragment f: channels 16f + 4h + 0..3 (.xy) and 16f + 8 + 4h + 0..3 (.zw) -> bytes 32ff + 8h and 32ff + 16 + 8h of the half row
      *reinterpret_cast<uint2*>(la.sw + 32 * ff) = make_uint2(v.x, v.y);
      *reinterpret_cast<uint2*>(la.sw + 32 * ff + 16) = make_uint2(v.z, v.w);
    }
    ef_u4 rv[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) rv[p] = __builtin_bit_cast(ef_u4, *reinterpret_cast<const uint4*>(la.sr + 8 * p * EF_STG_ROWB));
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      if (!(EF_ABL & 2))
        // soffset stays the IMMEDIATE 0 (the half-row offset rides in the instruction's offset field; the range check adds
        // it to voffset, and row * 256 + 128 + 16 c + 16 <= (row + 1) * 256 keeps the verdict per row): with a REGISTER
        // soffset hipcc does not pad the store-data hazard of a 128-bit store, and the VALU instruction behind the last
        // store of the second half overwrote its third data register before the store had read it (z2 rows 25 / 27 / 29
        // of a tile, channels 100-125; tools/hwtests/store_data_hazard.hip, tools/store_hazard_audit.py)
        __builtin_amdgcn_raw_buffer_store_b128(rv[p], dst, la.go + (unsigned)(2048 * p + 128 * half), 0, 0);
    }
    // (belt: four more wait states before anything may redefine the stored registers — the statement "writes" them)
    asm volatile("s_nop 3" : "+v"(rv[0]), "+v"(rv[1]), "+v"(rv[2]), "+v"(rv[3]));
  }
}

// Dropout of one 32-channel accumulator tile (registers i <-> channel 8(i>>2) + 4h + (i&3) of the tile) of a linear
// site: dropped elements are zeroed, kept ones left as they are (the caller folds the 1/(1-p) scale into its next fma).
// e32 = low word of the element index of the tile's channel 0 for this lane's token (a multiple of 32).
template <int DROP>
__device__ __forceinline__ void ef_drop_tile(ef_f32x16& acc, unsigned key, unsigned e32, int h4 /* 4 * h */, unsigned thresh) {
  if constexpr (DROP == 1) {
    asm volatile("" : "+v"(e32));      // the hash is formed HERE (its inputs are known from the top of the tile: hoisted, it only holds registers)
    const int hs = (int)(mix32((e32 >> 5) ^ key) >> h4);         // bit 8(i>>2) + (i&3) decides register i
#pragma unroll
    for (int i = 0; i < 16; ++i)
      acc[i] = __uint_as_float(__float_as_uint(acc[i]) & ef_bitmask(hs, 8 * (i >> 2) + (i & 3)));
  } else if constexpr (DROP != 0) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float dm[4];
      drop_scale4_t<DROP>(key, e32 + (unsigned)(8 * g) + (unsigned)h4, thresh, 1.f, dm);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[4 * g + j] = dm[j] != 0.f ? acc[4 * g + j] : 0.f;
    }
  }
}

// Diagnostic build (EF_ABL & 1024, tools/enc_timeline.py): s_memtime stamps at section edges of the forward, summed per
// section and wave in scalar registers, written once at kernel end to a buffer of their own (never an output).
#if EF_ABL & 1024
constexpr int EF_NSEC = 16;
__device__ unsigned long long ef_dbg[EF_NSEC * 4096];
#define EF_STAMP(K)                                                                                   \
  {                                                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                                \
    unsigned long long t_;                                                                            \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                      \
    __builtin_amdgcn_sched_barrier(0);                                                                \
    sec[K] += t_ - t_prev;                                                                            \
    t_prev = t_;                                                                                      \
  }
#else
#define EF_STAMP(K)
#endif

struct EfArgs {
  const unsigned short* x;        // [T,128]
  unsigned short* out;            // [T,128]
  unsigned short* z1;             // [T,128] or null
  unsigned short* z2;             // [T,128] or null
  const char* wpack;
  const float* prm;
  long long R;                    // table rows
  int S;                          // tokens per row
  int tail;                       // 1: out = alpha*x + beta_c*LN_t(x2)
  float alpha, beta_c, eps;
  unsigned thresh;                // dropout threshold (0 = off)
  float inv_keep;
  unsigned long long seed;
  unsigned rs0, rs1, rs2, rs3;    // attention, norm1, ffn, norm2 dropout streams
  int small_idx;                  // 1: every dropout element index of the call is below 2^32 (one RNG key per site)
  const unsigned short* o_in;     // ATT = false: the attention output o [T,128] computed elsewhere (rows of more than 32 tokens)
  float* st1;                     // ATT = false, optional: LayerNorm-1 statistics [T][2] = (mean, rstd) for an op-by-op backward
};

// Forward.  HD = head dim (16 or 32).  One workgroup = EF_WAVES wave tiles of 32 token slots per iteration.
// Unit pipeline: the weights stream through TWO 17 KiB buffers in units of half a stage (64 rows of a 128 x 128 tile;
// Wq|Wk rows or Wv rows of a head block); 14 units per iteration, unit u in buffer u & 1.  Invariant: while unit u is
// read, the DMA of unit u+1 is in flight into the other buffer.  EF_UNIT_NEXT at the end of unit u: barrier (every wave
// done with u, every wave's pieces of u+1 landed), then the DMA of u+2 goes into u's buffer at once, so it has the
// whole of unit u+1 to land.  One barrier per unit, no exposed DMA wait.
#define EF_SEC 0
#define EF_UBUF(U) (smem + ((U) & 1) * EF_UNIT_BYTES)
// Boundary at the end of unit U: wait until this wave's pieces of unit U+1 have landed, barrier, then issue the DMA of
// unit U+2 into U's buffer.  The wait is COUNTED: K = the vector-memory operations this wave issued AFTER the DMA of
// unit U+1 (output stores, the next tile's x loads) — they complete in issue order, so "at most K outstanding" means
// the DMA is done while the stores stay in flight.  (__syncthreads() would wait vmcnt(0): every output store of the tile
// then stalled the next barrier — stripping the stores saved 0.26 ms of a 1.34 ms launch.)  K must never exceed the real
// count: each call site states what was issued.
// ... and lgkmcnt(0): the wave's own LDS reads of the buffer that the barrier hands over to the next DMA must have
// RETURNED, not just been issued.  hipcc sinks the MFMAs of a unit's last k-steps below the boundary (an asm statement
// orders memory operations, not register-only instructions), so their `ds_read_b128` were still in the LDS queue when
// the wave crossed the raw s_barrier, and the DMA of unit U+2 — issued by ANY wave right behind the barrier — could land
// in U's buffer before those reads executed (round 4's "one wave tile wrong at percent rates with two workgroups per
// CU": DESIGN.md, LDS-DMA hazard; tools/hwtests/lds_dma_race.hip part C reproduces it without the arithmetic).
#ifndef EF_LGKM_FENCE
#define EF_LGKM_FENCE 1      // 0: the round-3/4 boundary (for the A/B of tools/enc_det_probe4.py only)
#endif
#if EF_LGKM_FENCE
#define EF_WAIT_VM(K) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(K) : "memory")
#else
#define EF_WAIT_VM(K) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(K) : "memory")
#endif
#define EF_UNIT_NEXT_K(U, VALID, U2, BYTES, WAIT)                                                     \
  {                                                                                                   \
    EF_STAMP(EF_SEC)                                                                                  \
    if (!(EF_ABL & 128)) { WAIT; EF_STAMP(12) __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }       \
    EF_STAMP(13)                                                                                      \
    if (!(EF_ABL & 32) && (VALID)) ef_dma<BYTES>(a.wpack + (size_t)(U2) * EF_UNIT_BYTES, EF_UBUF(U), wave, lane16); \
    EF_STAMP(14)                                                                                      \
  }
#define EF_UNIT_NEXT(U, VALID, U2, BYTES) EF_UNIT_NEXT_K(U, VALID, U2, BYTES, EF_WAIT_VM(0))
// ATT = false (round 5): everything BEHIND the attention — out-proj + dropout + residual + LN1, FFN, LN2, tail — for token
// rows whose attention ran elsewhere (rows of more than 32 tokens: tabgnn.py:127-129,219 S = 130; the 64-column table):
// the attention output arrives in `o_in`, units 0..7 (the QKV weights) are never fetched, the tile loop starts at unit 8.
#ifndef EF_FFN_X_LATE
#define EF_FFN_X_LATE 1      // ATT = false: 1 = only o is prefetched a tile ahead, x is loaded at the tile's top (it is first needed
#endif                       // behind the first out-proj chain); 0 = both prefetched (64 staging registers: 49 spilled)
template <int HD, int DROP /* 0 = off, else hash bits per element: 1 | 8 | 16 (common.hpp) */, bool ATT = true>
__global__ void __launch_bounds__(EF_THREADS, EF_WG_PER_CU) k_encoder_fwd(const EfArgs a_) {
  constexpr int U0 = ATT ? 0 : 8;             // first weight unit of a tile
  constexpr bool X_LATE = !ATT && EF_FFN_X_LATE;
  EfArgs a = a_;
  a.seed = live_seed(a_.seed);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* prm = reinterpret_cast<float*>(smem + 2 * EF_UNIT_BYTES);                         // 6 KiB
  char* stg_all = smem + 2 * EF_UNIT_BYTES + EF_P_FLOATS * 4;                              // EF_WAVES x 8 KiB
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tl = lane & 31, h = lane >> 5, h4 = 4 * h, lane16 = 16 * lane;
  char* stg = stg_all + wave * 8192;          // wave-private: output restage (4.5 KiB) / parked x1 fragments (8 KiB)
  constexpr int NH = EF_C / HD;                 // heads
  constexpr int HB = 32 / HD;                   // heads per 32-channel block
  // softmax in the exp2 domain: q is scaled by log2(e) / sqrt(head dim) when it is packed
  const float qscale = (HD == 32 ? 0.17677669529663687f : 0.25f) * 1.4426950408889634f;
  const int S = a.S;
  const int RW = 32 / S;                        // table rows per wave tile
  const long long n_wt = (a.R + RW - 1) / RW;   // wave tiles
  const long long n_it = (n_wt + EF_WAVES - 1) / EF_WAVES;

  for (int i = tid; i < EF_P_FLOATS; i += EF_THREADS) prm[i] = a.prm[i];
  if (blockIdx.x < n_it) ef_dma<EF_UNIT_BYTES>(a.wpack + (size_t)U0 * EF_UNIT_BYTES, smem, wave, lane16);
  EF_DMA_LANDED();
  __syncthreads();
  if (blockIdx.x < n_it) {
    if constexpr (ATT) ef_dma<EF_PART_BYTES>(a.wpack + EF_UNIT_BYTES, smem + EF_UNIT_BYTES, wave, lane16);
    else ef_dma<EF_UNIT_BYTES>(a.wpack + (size_t)(U0 + 1) * EF_UNIT_BYTES, smem + EF_UNIT_BYTES, wave, lane16);
  }

  // lane-constant addresses: every LDS access below is one of these bases + an immediate
  const int fb = EF_ROWB * tl + 16 * h;                          // weight fragment of row tl (+ 32 ks + 272 * row block)
  const char* pb = reinterpret_cast<const char*>(prm) + 16 * h;  // parameter float4 of this lane half (+ 4 * index)
  char* park = stg + lane16;                                     // x1 fragments parked for the second residual
  EfLaneAddr la;
  la.sw = stg + EF_STG_ROWB * tl + 8 * h;
  la.sr = stg + EF_STG_ROWB * (lane >> 3) + 16 * (lane & 7);
  la.go = (unsigned)(256 * (lane >> 3) + 16 * (lane & 7));
  const unsigned xoff = (unsigned)(256 * tl + 8 * h);             // this lane's first 8 bytes of x in its token row
  // geometry of this lane's query slot: table row q_row of the tile, tokens [row_lo, row_lo + S)
  const int q_row = tl / S;
  const int row_lo = q_row * S;
  // block-diagonal attention mask as one more MFMA k-step: U[token][r] = 16 * [token's row == r]; U U^T adds 256 to the
  // exp2-domain scores of (key, query) pairs of the same table row, so every other pair underflows in the softmax
  ef_v8bf uf;
#pragma unroll
  for (int j = 0; j < 8; ++j) uf[j] = (__bf16)((8 * h + j) == q_row ? 16.f : 0.f);
  // one RNG key per dropout site when the indices fit 32 bits (scalar registers), else per lane and tile
  const unsigned skey0 = rng_key(a.seed, a.rs0, 0u), skey1 = rng_key(a.seed, a.rs1, 0u), skey2 = rng_key(a.seed, a.rs2, 0u),
                 skey3 = rng_key(a.seed, a.rs3, 0u);

  // x of a tile in fragment order: xf[ks] element j = x[t][16ks + 8(j>>2) + 4h + (j&3)]   (rows past the end read as zeros)
#define EF_LOAD_X(DST, RS)                                                                            \
    _Pragma("unroll") for (int ks = 0; ks < 8; ++ks) {                                                \
      /* constant part in soffset: the range check (voffset against the tile's valid bytes) decides per token row */  \
      const uint2 lo = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(RS, xoff, 32 * ks, 0));          \
      const uint2 hi = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(RS, xoff, 32 * ks + 16, 0));     \
      DST[ks] = __builtin_bit_cast(ef_v8bf, make_uint4(lo.x, lo.y, hi.x, hi.y));                     \
    }
  // geometry of the wave tile of iteration IT: first table row (opaque scalar, see below), valid token slots, first token
#define EF_TILE_GEOM(IT, ROW0, NVALID, TOK0)                                                          \
    long long ROW0 = ((IT) * EF_WAVES + wave) * RW;                                                   \
    asm volatile("" : "+s"(ROW0));                                                                    \
    long long rh_##ROW0 = a.R - ROW0;                                                                 \
    rh_##ROW0 = rh_##ROW0 < 0 ? 0 : (rh_##ROW0 > RW ? RW : rh_##ROW0);                                \
    const int NVALID = (int)rh_##ROW0 * S;                                                            \
    const long long TOK0 = ROW0 * S;
  ef_v8bf xn[8];                // x of the NEXT tile, loaded while this tile's feed-forward runs
  ef_v8bf on[ATT ? 1 : 8];      // ATT = false: its attention output too
  if (blockIdx.x < n_it) {
    EF_TILE_GEOM((long long)blockIdx.x, row0p, nvalidp, tok0p)
    const __amdgpu_buffer_rsrc_t xrs0 = ef_tile_rsrc(a.x, tok0p, nvalidp);
    if constexpr (!X_LATE) { EF_LOAD_X(xn, xrs0) }
    if constexpr (!ATT) {
      const __amdgpu_buffer_rsrc_t ors0 = ef_tile_rsrc(a.o_in, tok0p, nvalidp);
      EF_LOAD_X(on, ors0)
    }
  }

#if EF_ABL & 1024
  unsigned long long sec[EF_NSEC] = {0}, t_prev;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_prev)::"memory");
#endif
  for (long long it = blockIdx.x; it < n_it; it += gridDim.x) {
    // (row0 opaque, in scalar registers: nothing below is an affine function of the loop counter for the compiler, so it
    // cannot strength-reduce the per-lane 64-bit addresses into loop-carried VGPR pairs — 11 of them, spilled, before)
    EF_TILE_GEOM(it, row0, nvalid, tok0)
    const bool has_next = it + gridDim.x < n_it;
    ef_v8bf xf[8];
    if constexpr (X_LATE) {
      const __amdgpu_buffer_rsrc_t xrs_top = ef_tile_rsrc(a.x, tok0, nvalid);
      EF_LOAD_X(xf, xrs_top)
    } else {
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) xf[ks] = xn[ks];
    }

    // accumulator initialised with a per-row bias (float4 per register group, straight from LDS: no VALU)
#define EF_ACC_BIAS(ACC, POFF)                                                                        \
    _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                   \
      const float4 b = *reinterpret_cast<const float4*>(pb + 4 * ((POFF) + 8 * g));                   \
      ACC[4 * g] = b.x; ACC[4 * g + 1] = b.y; ACC[4 * g + 2] = b.z; ACC[4 * g + 3] = b.w;             \
    }
    // 8 k-steps of out^T[rows, tokens] += W[rows, :] . B : weight fragments at WT + fb + IMM + 32 ks
#define EF_CHAIN(ACC, WT, IMM, BOP)                                                                   \
    _Pragma("unroll") for (int ks = 0; ks < 8; ++ks) {                                                \
      ACC = EF_MFMA(ef_frag((WT) + (IMM) + 32 * ks, fb), BOP[ks], ACC);                               \
      if (ks == 3) EF_MID_FENCE();                                                                    \
    }

    EF_STAMP(10)
#undef EF_SEC
#define EF_SEC 0
    // ================================================================ attention, per 32-channel head block
    ef_v8bf of[8];               // attention output o^T, packed: the B operand of the output projection
    // attention-dropout geometry: element index of (table row, head, query q, key k) = ((row * NH + head) * S + q) * S + k
    // wave-uniform 64-bit part (scalar) + lane part (32 bits)
    const unsigned long long att_u = (unsigned long long)row0 * (unsigned long long)(NH * S * S);
    const unsigned att_l = (unsigned)(q_row * NH * S * S + (tl - row_lo) * S + h4 - row_lo);   // head 0, this query, register 0's key
    if constexpr (!ATT) {
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) of[ks] = on[ks];
    }
#pragma unroll
    for (int blk = 0; blk < (ATT ? 4 : 0); ++blk) {
      const char* wu = EF_UBUF(2 * blk);            // unit 2 blk: Wq rows | Wk rows
      ef_f32x16 acc;
      // K^T and Q^T blocks [32 d, 32 tokens]
      EF_ACC_BIAS(acc, EF_P_BIN + 128 + 32 * blk)
      EF_CHAIN(acc, wu, EF_PART_BYTES, xf)
      const ef_v8bf kf0 = ef_pack<0>(acc), kf1 = ef_pack<1>(acc);
      EF_FENCE();
      EF_ACC_BIAS(acc, EF_P_BIN + 32 * blk)
      EF_CHAIN(acc, wu, 0, xf)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] *= qscale;
      const ef_v8bf qf0 = ef_pack<0>(acc), qf1 = ef_pack<1>(acc);
      if (blk == 0) {
        // issued since the DMA of unit 1 (at the end of the previous tile): its stores of out (8) and of z2 (8, when written)
        if (a.z2) { EF_UNIT_NEXT_K(0, true, 2, EF_UNIT_BYTES, EF_WAIT_VM(16)) }
        else { EF_UNIT_NEXT_K(0, true, 2, EF_UNIT_BYTES, EF_WAIT_VM(8)) }
      } else {
        EF_UNIT_NEXT(2 * blk, true, 2 * blk + 2, EF_UNIT_BYTES)
      }
      wu = EF_UBUF(2 * blk + 1);                    // unit 2 blk + 1: Wv rows
      // V block in the transposed orientation [32 tokens (rows), 32 d (columns)]
      {
        const float bv = prm[EF_P_BIN + 256 + 32 * blk + tl];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = bv;
      }
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        acc = EF_MFMA(xf[ks], ef_frag(wu + 32 * ks, fb), acc);
        if (ks == 3) EF_MID_FENCE();
      }
      const ef_v8bf vf0 = ef_pack<0>(acc), vf1 = ef_pack<1>(acc);
      if (blk < 3) { EF_UNIT_NEXT(2 * blk + 1, true, 2 * blk + 3, EF_PART_BYTES) }     // next head block's Wv rows
      else { EF_UNIT_NEXT(2 * blk + 1, true, 2 * blk + 3, EF_UNIT_BYTES) }            // (units 8, 9: Wo)

      EF_STAMP(0)
#pragma unroll
      for (int hh = 0; hh < HB; ++hh) {
        // scores S^T[key (rows), query (columns)] + 256 * [same table row], exp2 domain
        ef_f32x16 st = EF_MFMA(uf, uf, ef_zero16());
        if constexpr (HB == 1) {
          st = EF_MFMA(kf0, qf0, st);
          st = EF_MFMA(kf1, qf1, st);
        } else {
          st = hh == 0 ? EF_MFMA(kf0, qf0, st) : EF_MFMA(kf1, qf1, st);
        }
        float sc = 1.f;           // 1 / sum (x 1 / keep), applied to the O^T columns below
        if (!(EF_ABL & 8)) {
          float mx = fmaxf(fmaxf(st[0], st[1]), st[2]);
#pragma unroll
          for (int i = 3; i < 15; i += 2) mx = fmaxf(fmaxf(mx, st[i]), st[i + 1]);
          mx = fmaxf(mx, st[15]);
          mx = fmaxf(mx, ef_xor32(mx));
          float l = 0.f;
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            st[i] = __builtin_amdgcn_exp2f(st[i] - mx);
            l += st[i];
          }
          l += ef_xor32(l);
          sc = __builtin_amdgcn_rcpf(l);
          if constexpr (DROP == 1) {
            // 32 consecutive elements per hash: this query's keys sit at att + 4h + (i&3) + 8(i>>2) - row_lo
            // (element of register i) = ab + (i&3) + 8(i>>2)
            unsigned al = att_l;
            asm volatile("" : "+v"(al));        // (index and hashes formed here: hoisted per head, they were spilled)
            const unsigned ab32 = (unsigned)att_u + al + (unsigned)((blk * HB + hh) * S * S);
            unsigned k0 = skey0, k1 = skey0;
            if (!a.small_idx) {
              const unsigned long long ab = att_u + (unsigned long long)(al + (unsigned)((blk * HB + hh) * S * S));
              k0 = rng_key(a.seed, a.rs0, (unsigned)(ab >> 32));
              k1 = rng_key(a.seed, a.rs0, (unsigned)((ab + 32) >> 32));
            }
            const unsigned g0 = ab32 >> 5;
            const unsigned w0 = mix32(g0 ^ k0), w1 = mix32(((g0 + 1u) & 0x07ffffffu) ^ k1);
            const int hs = (int)__builtin_amdgcn_alignbit(w1, w0, ab32 & 31u);
#pragma unroll
            for (int i = 0; i < 16; ++i)
              st[i] = __uint_as_float(__float_as_uint(st[i]) & ef_bitmask(hs, (i & 3) + 8 * (i >> 2)));
            sc *= a.inv_keep;
          } else if constexpr (DROP != 0) {
            const int head = blk * HB + hh;
            const unsigned long long blk0 = att_u + (unsigned long long)((q_row * NH + head) * S * S);
            const unsigned lo0 = (unsigned)blk0 + (unsigned)((tl - row_lo) * S);
            const unsigned key0 = rng_key(a.seed, a.rs0, (unsigned)(blk0 >> 32));
            const unsigned key1 = rng_key(a.seed, a.rs0, (unsigned)(blk0 >> 32) + 1u);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
              const int key = (i & 3) + 8 * (i >> 2) + h4;
              const unsigned lo = lo0 + (unsigned)(key - row_lo);
              const unsigned kk = lo < (unsigned)blk0 ? key1 : key0;       // carry into the high word
              st[i] = drop_scale_key_t<DROP>(kk, lo, a.thresh, 1.f) != 0.f ? st[i] : 0.f;
            }
            sc *= a.inv_keep;
          }
        }
        const ef_v8bf pf0 = ef_pack<0>(st), pf1 = ef_pack<1>(st);
        // O^T[d (rows), query (columns)] = V^T P^T, columns scaled by 1 / sum
        ef_f32x16 ot = EF_MFMA(vf0, pf0, ef_zero16());
        ot = EF_MFMA(vf1, pf1, ot);
        if constexpr (HB == 1) {
#pragma unroll
          for (int i = 0; i < 16; ++i) ot[i] *= sc;
          of[2 * blk] = ef_pack<0>(ot); of[2 * blk + 1] = ef_pack<1>(ot);
        } else {
#pragma unroll
          for (int i = 0; i < 8; ++i) ot[8 * hh + i] *= sc;
          if (hh == 0) of[2 * blk] = ef_pack<0>(ot); else of[2 * blk + 1] = ef_pack<1>(ot);
        }
      }
      EF_STAMP(1)
    }

#undef EF_SEC
#define EF_SEC 2
    // ================================================================ output projection + LayerNorm 1 (registers)
    // z1 = x + drop(o Wo^T + b_o), rounded to bf16 (what is stored and what the backward recomputes from);
    // x1 = LN(z1) * g1 + be1, rounded to bf16: the feed-forward input and residual
    const unsigned long long e_base = (unsigned long long)tok0 * EF_C + (unsigned long long)(tl * EF_C);   // (scalar + lane part)
    const unsigned e_lo = (unsigned)((unsigned long long)tok0 * EF_C) + (unsigned)(tl * EF_C);
    const float keep_scale = DROP ? a.inv_keep : 1.f;
    ef_v8bf x1f[8];
    {
      ef_v8bf zp[8];
      unsigned dkey = skey1;
      if (DROP && !a.small_idx) dkey = rng_key(a.seed, a.rs1, (unsigned)(e_base >> 32));
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const char* wu = EF_UBUF(8 + (m >> 1));
        ef_f32x16 acc;
        EF_ACC_BIAS(acc, EF_P_BO + 32 * m)
        EF_CHAIN(acc, wu, EF_PART_BYTES * (m & 1), of)
        if (m & 1) {                                                                       // Wo unit done -> W1 units
          if (!ATT && m == 1) {
            // (this tile's first boundary: issued since the DMA of unit 9 at the end of the previous tile: its out / z2 stores)
            if (a.z2) { EF_UNIT_NEXT_K(8, true, 10, EF_UNIT_BYTES, EF_WAIT_VM(16)) }
            else { EF_UNIT_NEXT_K(8, true, 10, EF_UNIT_BYTES, EF_WAIT_VM(8)) }
          } else {
            EF_UNIT_NEXT(8 + (m >> 1), true, 10 + (m >> 1), EF_UNIT_BYTES)
          }
        }
        ef_drop_tile<DROP>(acc, dkey, e_lo + 32u * m, h4, a.thresh);
        uint4 o0, o1;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          ef_f2 u0, u1;
          u0.x = acc[2 * d]; u0.y = acc[2 * d + 1]; u1.x = acc[8 + 2 * d]; u1.y = acc[8 + 2 * d + 1];
          o0[d] = ef_pk(ef_fma2(u0, ef_splat(keep_scale), ef_unpk(ef_dw(xf[2 * m], d))));
          o1[d] = ef_pk(ef_fma2(u1, ef_splat(keep_scale), ef_unpk(ef_dw(xf[2 * m + 1], d))));
        }
        zp[2 * m] = __builtin_bit_cast(ef_v8bf, o0);
        zp[2 * m + 1] = __builtin_bit_cast(ef_v8bf, o1);
        EF_FENCE();
      }
      EF_STAMP(2)
      if (a.z1) ef_store_rows(zp, la, ef_tile_rsrc(a.z1, tok0, nvalid));
      EF_STAMP(3)
      float mu, rstd;
      ef_row_stats(zp, a.eps, mu, rstd);
      if constexpr (!ATT) {
        if (a.st1 && h == 0 && tl < nvalid) *reinterpret_cast<float2*>(a.st1 + 2 * (tok0 + tl)) = make_float2(mu, rstd);
      }
      const float nmr = -mu * rstd;
#pragma unroll
      for (int f = 0; f < 8; ++f) {
        EF_FENCE();
        x1f[f] = ef_ln_apply(zp[f], nmr, rstd, reinterpret_cast<const float*>(pb) + EF_P_G1 + 16 * f,
                             reinterpret_cast<const float*>(pb) + EF_P_BE1 + 16 * f);
        // parked in LDS (lane-linear, read back by the same lane) for the residual of the second sub-layer: 32
        // registers less across the feed-forward
        *reinterpret_cast<uint4*>(park + 1024 * f) = __builtin_bit_cast(uint4, x1f[f]);
      }
    }

    EF_STAMP(4)
#undef EF_SEC
#define EF_SEC 5
    // ================================================================ feed-forward 1: h = drop(relu(W1 x1 + b1))
    ef_v8bf hf[8];
    {
      unsigned dkey = skey2;
      if (DROP && !a.small_idx) dkey = rng_key(a.seed, a.rs2, (unsigned)(e_base >> 32));
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const char* wu = EF_UBUF(10 + (m >> 1));
        ef_f32x16 acc;
        EF_ACC_BIAS(acc, EF_P_B1 + 32 * m)
        EF_CHAIN(acc, wu, EF_PART_BYTES * (m & 1), x1f)
        if (m == 1) {                                // W1 units done -> W2 units
          // issued since the DMA of unit 11: the 8 stores of z1 (when it is written)
          if (a.z1) { EF_UNIT_NEXT_K(10, true, 12, EF_UNIT_BYTES, EF_WAIT_VM(8)) }
          else { EF_UNIT_NEXT_K(10, true, 12, EF_UNIT_BYTES, EF_WAIT_VM(0)) }
          // the next tile's x (its registers are free from here on): 16 loads that the next boundary must not wait for
          if (has_next) {
            EF_TILE_GEOM(it + gridDim.x, row0n, nvalidn, tok0n)
            const __amdgpu_buffer_rsrc_t xrsn = ef_tile_rsrc(a.x, tok0n, nvalidn);
            if constexpr (!X_LATE) { EF_LOAD_X(xn, xrsn) }
            if constexpr (!ATT) {
              const __amdgpu_buffer_rsrc_t orsn = ef_tile_rsrc(a.o_in, tok0n, nvalidn);
              EF_LOAD_X(on, orsn)
            }
          }
        } else if (m == 3) {
          // issued since the DMA of unit 12: the 16 x loads above (when there is a next tile; 32 with the o loads)
          if (has_next) { EF_UNIT_NEXT_K(11, true, 13, EF_UNIT_BYTES, EF_WAIT_VM(ATT || X_LATE ? 16 : 32)) }
          else { EF_UNIT_NEXT_K(11, true, 13, EF_UNIT_BYTES, EF_WAIT_VM(0)) }
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = fmaxf(acc[i], 0.f);
        ef_drop_tile<DROP>(acc, dkey, e_lo + 32u * m, h4, a.thresh);
        if constexpr (DROP != 0) {
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[i] *= keep_scale;
        }
        hf[2 * m] = ef_pack<0>(acc);
        hf[2 * m + 1] = ef_pack<1>(acc);
        EF_FENCE();
      }
    }

    EF_STAMP(5)
#undef EF_SEC
#define EF_SEC 6
    // ================================================================ feed-forward 2 + LayerNorm 2 (+ tail LayerNorm)
    {
      ef_v8bf zp[8];
      const bool reload_x = a.tail && a.alpha != 0.f;
      const __amdgpu_buffer_rsrc_t xrs = ef_tile_rsrc(a.x, tok0, nvalid);
      unsigned dkey = skey3;
      if (DROP && !a.small_idx) dkey = rng_key(a.seed, a.rs3, (unsigned)(e_base >> 32));
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const char* wu = EF_UBUF(12 + (m >> 1));
        ef_f32x16 acc;
        EF_ACC_BIAS(acc, EF_P_B2 + 32 * m)
        EF_CHAIN(acc, wu, EF_PART_BYTES * (m & 1), hf)
        if (m == 1) {                                // W2 units done -> the next iteration's first units
          EF_UNIT_NEXT(12, has_next, U0, EF_UNIT_BYTES)
          // this tile's x again for the tail combine (L2-hot): 16 loads the next boundary must not wait for
          if (reload_x) { EF_LOAD_X(xf, xrs) }
        } else if (m == 3) {
          if constexpr (ATT) {
            if (reload_x) { EF_UNIT_NEXT_K(13, has_next, 1, EF_PART_BYTES, EF_WAIT_VM(16)) }
            else { EF_UNIT_NEXT_K(13, has_next, 1, EF_PART_BYTES, EF_WAIT_VM(0)) }
          } else {
            if (reload_x) { EF_UNIT_NEXT_K(13, has_next, U0 + 1, EF_UNIT_BYTES, EF_WAIT_VM(16)) }
            else { EF_UNIT_NEXT_K(13, has_next, U0 + 1, EF_UNIT_BYTES, EF_WAIT_VM(0)) }
          }
        }
        const uint4 r0 = *reinterpret_cast<const uint4*>(park + 1024 * (2 * m));
        const uint4 r1 = *reinterpret_cast<const uint4*>(park + 1024 * (2 * m + 1));
        ef_drop_tile<DROP>(acc, dkey, e_lo + 32u * m, h4, a.thresh);
        uint4 o0, o1;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          ef_f2 u0, u1;
          u0.x = acc[2 * d]; u0.y = acc[2 * d + 1]; u1.x = acc[8 + 2 * d]; u1.y = acc[8 + 2 * d + 1];
          o0[d] = ef_pk(ef_fma2(u0, ef_splat(keep_scale), ef_unpk(r0[d])));
          o1[d] = ef_pk(ef_fma2(u1, ef_splat(keep_scale), ef_unpk(r1[d])));
        }
        zp[2 * m] = __builtin_bit_cast(ef_v8bf, o0);
        zp[2 * m + 1] = __builtin_bit_cast(ef_v8bf, o1);
        EF_FENCE();
      }
      EF_STAMP(6)
      if (a.z2) ef_store_rows(zp, la, ef_tile_rsrc(a.z2, tok0, nvalid));
      EF_STAMP(7)
      float mu, rstd;
      ef_row_stats(zp, a.eps, mu, rstd);
      float nmr = -mu * rstd;
      // x2 = LN2(z2), rounded to bf16 as the unfused path stores it
#pragma unroll
      for (int f = 0; f < 8; ++f) {
        EF_FENCE();
        zp[f] = ef_ln_apply(zp[f], nmr, rstd, reinterpret_cast<const float*>(pb) + EF_P_G2 + 16 * f,
                            reinterpret_cast<const float*>(pb) + EF_P_BE2 + 16 * f);
      }
      if (a.tail) {
        ef_row_stats(zp, a.eps, mu, rstd);
        nmr = -mu * rstd;
#pragma unroll
        for (int f = 0; f < 8; ++f) {
          EF_FENCE();
          zp[f] = ef_ln_combine(zp[f], xf[f], nmr, rstd, reinterpret_cast<const float*>(pb) + EF_P_GT + 16 * f,
                                reinterpret_cast<const float*>(pb) + EF_P_BT + 16 * f, a.alpha, a.beta_c);
        }
      }
      EF_STAMP(8)
      ef_store_rows(zp, la, ef_tile_rsrc(a.out, tok0, nvalid));
      EF_STAMP(9)
    }
  }
#if EF_ABL & 1024
  if (lane == 0) {
    const int slot = (blockIdx.x * EF_WAVES + wave) & 4095;
    for (int k = 0; k < EF_NSEC; ++k) ef_dbg[slot * EF_NSEC + k] = sec[k];
  }
#endif
}


#undef EF_STAMP
#define EF_STAMP(K)
// ---------------------------------------------------------------------------------------------- backward, feed-forward half
// Everything between the layer output and x1, recomputed from (z1, z2) in the same register-chained form:
//     g = d out -> (tail LayerNorm backward) -> LayerNorm-2 backward -> d_z2 ; d_y2 = mask3 . d_z2
//     x1 = LN1(z1) ; h = drop(relu(W1 x1 + b1))                           (recomputed, one GEMM)
//     d_h = d_y2 W2 ; d_hpre = d_h . [h > 0] / keep ; d_x1 = d_z2 + d_hpre W1
// Written: d_x1 (the gradient the attention half continues from) and the four operands of the two weight-gradient
// GEMMs (d_y2, h) and (d_hpre, x1) — tg_gemm_tn_bf16 also sums the bias gradients from d_y2 / d_hpre — and, per
// workgroup, the partial LayerNorm parameter gradients of norm2 and the tail norm (see ef_ln_bwd below; round 2 read
// g, z2 again in a separate streaming kernel for them).
// Units (LDS weight images, k-permuted like the forward's, two 64-row units per [128,128] tile): W1 | W2^T | W1^T,
// streamed through the forward's double-buffered unit pipeline (round 2: one 32 KiB buffer, DMA latency exposed).
constexpr int EF_WAVE_LDS = 9728;          // per-wave scratch: staging regions A / B (32 rows x 144 B each) + a 512-byte table
struct EbArgs {
  const unsigned short *g, *z1, *z2;
  unsigned short *dx1, *dy2, *hout, *dhpre, *x1out;
  const char* wpack;
  const float* prm;
  long long R;
  int S, tail;
  float beta_c, eps;
  unsigned thresh;
  float inv_keep;
  unsigned long long seed;
  unsigned rs2, rs3;
  int small_idx;
  float* lnp;                    // [grid][4][128] partial sums: d gamma2, d beta2, d gamma_t, d beta_t
  float *dwp, *dbp;              // DW variant: [grid][2][128][128] / [grid][2][128] partial weight / bias gradients (W1, W2)
};
// every LDS write of this wave has landed (before a barrier that publishes staged tiles to the other waves)
#define EF_LDS_DONE() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

__device__ __forceinline__ void ef_dot2c(float& acc, unsigned a, unsigned b) {      // acc += a.lo*b.lo + a.hi*b.hi (bf16 pairs)
  asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(acc) : "v"(a), "v"(b));
}
// gfx950 data hazard: the result of a DOT instruction may be read by a different VALU opcode only 3 wait states later
// (back-to-back accumulation by the same DOT opcode is fine).  hipcc inserts those for its own DOT instructions, not for
// inline asm: every accumulator of an ef_dot2c chain passes through this before anything else reads it.  (Round 4: with
// another schedule of the same source the last v_dot2c of the d-beta sums was followed directly by its v_add — the
// gradient of tail.bias lost the tokens of the last read, 15 % off, while every other sum was right.)
#define EF_DOT_SETTLE4(A, B, C, D) asm volatile("s_nop 3" : "+v"(A), "+v"(B), "+v"(C), "+v"(D))
typedef short ef_v4s __attribute__((ext_vector_type(4)));
typedef ef_v4s __attribute__((address_space(3))) * ef_lds_v4s_ptr;

// LayerNorm backward of a wave tile (token on the lane, 64 channels in its registers) + the parameter gradients.
//   dy, z: packed fragments; dy_scale: factor on dy (the tail norm's beta_c); gamma: LDS floats (+ 4h applied);
//   out[f] = bf16( rstd * (dg - mean(dg) - xhat * mean(dg * xhat)) ),  dg = dy * dy_scale * gamma,  xhat = (z - mu) * rstd
//   OUT2: out2[f] = bf16(post(f, d, r)) — a second result formed from every fp32 result pair r before it is rounded (the
//   caller's dropout-masked copy); out2 may be the registers of dy (a fragment of dy is dead once it has been staged).
// Parameter gradients (sums over TOKENS = over lanes): dy and bf16(xhat) of 64 channels at a time are staged as
// [token][channel] rows in the wave's LDS regions A / B and read back transposed (ds_read_b64_tr_b16: lane = channel, 4
// tokens per read); v_dot2c_f32_bf16 then adds dy . xhat (d gamma) and dy . 1 (d beta) over token pairs into this lane's
// accumulators acc[2 * half + 0 / 1] — 64 matrix-free VALU instructions per LayerNorm, no second pass over HBM.
template <bool OUT2, typename Post>
__device__ __forceinline__ void ef_ln_bwd(const ef_v8bf (&dy)[8], const ef_v8bf (&z)[8], float dy_scale, float mu, float rstd,
                                          const float* gamma, ef_v8bf (&out)[8], ef_v8bf (&out2)[8], const EfLaneAddr& la,
                                          const char* trb, float (&acc)[4], Post post) {
  constexpr bool want_params = true;      // (always: the kernels are only launched by a training backward)
  const float nmr = -mu * rstd;
  ef_f2 s1 = ef_splat(0.f), s2 = ef_splat(0.f);
#pragma unroll
  for (int f = 0; f < 8; ++f) {
    const float4 g0 = *reinterpret_cast<const float4*>(gamma + 16 * f), g1 = *reinterpret_cast<const float4*>(gamma + 16 * f + 8);
    const ef_f2 gg[4] = {{g0.x, g0.y}, {g0.z, g0.w}, {g1.x, g1.y}, {g1.z, g1.w}};
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const ef_f2 dg = ef_unpk(ef_dw(dy[f], d)) * gg[d];
      const ef_f2 xh = ef_fma2(ef_unpk(ef_dw(z[f], d)), ef_splat(rstd), ef_splat(nmr));
      s1 += dg;
      s2 = ef_fma2(dg, xh, s2);
    }
  }
  float m1 = s1.x + s1.y, m2 = s2.x + s2.y;
  m1 += ef_xor32(m1);
  m2 += ef_xor32(m2);
  const float c0 = rstd * dy_scale, c1 = -m1 * (dy_scale * (1.f / 128.f)) * rstd, c2 = -m2 * (dy_scale * (1.f / 128.f)) * rstd;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
#pragma unroll
    for (int ff = 0; ff < 4; ++ff) {
      const int f = 4 * half + ff;
      EF_FENCE();
      // second pass over dy, z, gamma: opaque copies, or the compiler keeps the first pass's 64 + 64 unpacked values and
      // the 64 gamma values alive across the reductions (spills)
      const float* gam2 = gamma;
      ef_v8bf dyo = dy[f], zo = z[f];
      asm volatile("" : "+v"(gam2), "+v"(dyo), "+v"(zo));
      const float4 g0 = *reinterpret_cast<const float4*>(gam2 + 16 * f), g1 = *reinterpret_cast<const float4*>(gam2 + 16 * f + 8);
      const ef_f2 gg[4] = {{g0.x, g0.y}, {g0.z, g0.w}, {g1.x, g1.y}, {g1.z, g1.w}};
      const uint4 dyw = __builtin_bit_cast(uint4, dyo);
      uint4 o, o2, xw;
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const ef_f2 dg = ef_unpk(dyw[d]) * gg[d];
        const ef_f2 xh = ef_fma2(ef_unpk(ef_dw(zo, d)), ef_splat(rstd), ef_splat(nmr));
        const ef_f2 r = ef_fma2(xh, ef_splat(c2), ef_fma2(dg, ef_splat(c0), ef_splat(c1)));
        if constexpr (OUT2) o2[d] = ef_pk(post(f, d, r));
        o[d] = ef_pk(r);
        xw[d] = ef_pk(xh);
      }
      if (want_params) {        // (dy is staged BEFORE out[f] is written: out may be the same registers)
        *reinterpret_cast<uint2*>(la.sw + 32 * ff) = make_uint2(dyw.x, dyw.y);
        *reinterpret_cast<uint2*>(la.sw + 32 * ff + 16) = make_uint2(dyw.z, dyw.w);
        *reinterpret_cast<uint2*>(la.sw + 4608 + 32 * ff) = make_uint2(xw.x, xw.y);
        *reinterpret_cast<uint2*>(la.sw + 4608 + 32 * ff + 16) = make_uint2(xw.z, xw.w);
      }
      out[f] = __builtin_bit_cast(ef_v8bf, o);
      if constexpr (OUT2) out2[f] = __builtin_bit_cast(ef_v8bf, o2);
    }
    if (want_params) {
      float ga = 0.f, gb = 0.f, ba = 0.f, bb = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const uint2 dv = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((ef_lds_v4s_ptr)(trb + 576 * j)));
        const uint2 xv = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((ef_lds_v4s_ptr)(trb + 4608 + 576 * j)));
        ef_dot2c(ga, dv.x, xv.x); ef_dot2c(gb, dv.y, xv.y);
        ef_dot2c(ba, dv.x, 0x3f803f80u); ef_dot2c(bb, dv.y, 0x3f803f80u);
      }
      EF_DOT_SETTLE4(ga, gb, ba, bb);
      acc[2 * half] += (ga + gb) * dy_scale;
      acc[2 * half + 1] += (ba + bb) * dy_scale;
    }
  }
}

// workgroup partial of the per-lane LayerNorm parameter sums: lane = channel (64 half + lane) -> lnp[block][4][128]
__device__ __forceinline__ void ef_ln_partials(const float (&q)[8], float* red /* LDS [EF_WAVES][4][128] */, float* lnp, int wave,
                                               int lane, int tid) {
  __syncthreads();               // every wave is done with its scratch (red aliases the staging regions)
#pragma unroll
  for (int v = 0; v < 4; ++v)
#pragma unroll
    for (int half = 0; half < 2; ++half) red[(wave * 4 + v) * 128 + 64 * half + lane] = q[2 * v + half];
  __syncthreads();
  for (int i = tid; i < 512; i += EF_THREADS) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < EF_WAVES; ++w) t += red[w * 512 + i];
    lnp[(size_t)blockIdx.x * 512 + i] = t;
  }
}

// ---- weight gradients INSIDE the chained backward kernels (DW variants, round 4)
// dW[n][k] = sum over tokens G[t][n] X[t][k] contracts over TOKENS, which sit on the lanes of a wave tile: the MFMA
// operands are the transposes of what a wave holds, and a wave alone would need all 16 accumulator tiles of a
// [128,128] weight.  So the four waves of a workgroup stage their G and X tiles as [token][channel] bf16 rows in LDS
// (ef_stage_tile: the same writes ef_store_rows makes on the way to HBM), and after a unit-boundary barrier wave
// (wr, wc) = (wave >> 1, wave & 1) accumulates rows 64 wr .., columns 64 wc .. of dW over all 128 token slots of the
// iteration: fragments by ds_read_b64_tr_b16 (lane = channel, 4 tokens per read; the token order of a k-step is the
// same for both operands, which is all a contraction needs), 4 accumulator tiles per weight that live across the
// persistent loop (AGPRs: the DW kernels are built for ONE workgroup per CU, 512 registers per wave), written once per
// workgroup as fp32 partials [grid][weights][128][128] and summed in block order by k_ef_dw_reduce.  Bias gradients
// (column sums of G) come from the A fragments by v_dot2c with ones in the wc == 0 waves.
// The operand tensors (d_y2, h, d_hpre, x1 | d_y, o, d_qkv) are never written to HBM and the four / two weight-gradient
// GEMMs per layer that read them back are gone.
constexpr int EF_HALF_B = 32 * EF_STG_ROWB;           // 4608: one staged [32 token][64 channel] block
constexpr int EF_TILE_B = 2 * EF_HALF_B;              // 9216: a staged [32 token][128 channel] tile
constexpr int EF_WAVE_LDS_FFN_DW = 3 * EF_TILE_B;     // G | X | X1 regions of a wave (feed-forward half)
constexpr int EF_DW_TILE_FLOATS = 128 * 128;

__device__ __forceinline__ void ef_stage_tile(const ef_v8bf (&zp)[8], char* sw /* region + 144 * tl + 8 * h */) {
#pragma unroll
  for (int half = 0; half < 2; ++half)
#pragma unroll
    for (int ff = 0; ff < 4; ++ff) {
      const uint4 v = __builtin_bit_cast(uint4, zp[4 * half + ff]);
      *reinterpret_cast<uint2*>(sw + EF_HALF_B * half + 32 * ff) = make_uint2(v.x, v.y);
      *reinterpret_cast<uint2*>(sw + EF_HALF_B * half + 32 * ff + 16) = make_uint2(v.z, v.w);
    }
}
// fragment (lane = channel c of a 32-channel block, 8 tokens {4h'+0..3, 8+4h'+0..3} of a 16-token k-step) of a staged tile
__device__ __forceinline__ ef_v8bf ef_tr_frag(const char* p) {
  const uint2 r0 = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((ef_lds_v4s_ptr)(p)));
  const uint2 r1 = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((ef_lds_v4s_ptr)(p + 8 * EF_STG_ROWB)));
  return __builtin_bit_cast(ef_v8bf, make_uint4(r0.x, r0.y, r1.x, r1.y));
}
// ga / xa: stg_all + region offset + EF_HALF_B * (wr | wc) + the lane's transposed-read base (ef_tr_lane); WL = wave stride
__device__ __forceinline__ int ef_tr_lane(int lane) {
  return EF_STG_ROWB * (4 * (lane >> 5) + ((lane & 15) >> 2)) + 32 * ((lane >> 4) & 1) + 8 * (lane & 3);
}
template <int WL, bool BIAS>
__device__ __forceinline__ void ef_dw_phase(ef_f32x16 (&acc)[4], float (&bsum)[2], const char* ga, const char* xa) {
#pragma unroll
  for (int ws = 0; ws < EF_WAVES; ++ws) {          // source wave
#pragma unroll
    for (int t = 0; t < 2; ++t) {                  // tokens 16 t .. 16 t + 15 of its tile
      const int o = ws * WL + 16 * t * EF_STG_ROWB;
      const ef_v8bf a0 = ef_tr_frag(ga + o), a1 = ef_tr_frag(ga + o + 64);
      const ef_v8bf b0 = ef_tr_frag(xa + o), b1 = ef_tr_frag(xa + o + 64);
      acc[0] = EF_MFMA(a0, b0, acc[0]);
      acc[1] = EF_MFMA(a0, b1, acc[1]);
      acc[2] = EF_MFMA(a1, b0, acc[2]);
      acc[3] = EF_MFMA(a1, b1, acc[3]);
      if constexpr (BIAS) {
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          ef_dot2c(bsum[0], ef_dw(a0, d), 0x3f803f80u);
          ef_dot2c(bsum[1], ef_dw(a1, d), 0x3f803f80u);
        }
      }
    }
  }
}
// the workgroup's partial of one weight: tile (i, j) of wave (wr, wc) = rows 64 wr + 32 i .., columns 64 wc + 32 j ..
__device__ __forceinline__ void ef_dw_write(const ef_f32x16 (&acc)[4], float* dst /* [128][128] of this block and weight */,
                                            int wr, int wc, int lane) {
  const int tl = lane & 31, h = lane >> 5;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        dst[(64 * wr + 32 * i + 8 * (r >> 2) + 4 * h + (r & 3)) * 128 + 64 * wc + 32 * j + tl] = acc[2 * i + j][r];
}
__device__ __forceinline__ void ef_db_write(const float (&bsum)[2], float* dst /* [128] */, int wr, int lane) {
  float b0 = bsum[0], b1 = bsum[1];
  asm volatile("s_nop 3" : "+v"(b0), "+v"(b1));          // (DOT result -> other VALU hazard, see EF_DOT_SETTLE4)
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const float v = (i ? b1 : b0) + ef_xor32(i ? b1 : b0);
    if (lane < 32) dst[64 * wr + 32 * i + lane] = v;
  }
}

template <int DROP /* 0 = off, else hash bits per element: 1 | 8 | 16 (common.hpp) */, bool DW /* weight gradients inside */>
__global__ void __launch_bounds__(EF_THREADS, DW ? 1 : EF_WG_PER_CU) k_encoder_bwd_ffn(const EbArgs a_) {
  EbArgs a = a_;
  a.seed = live_seed(a_.seed);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* prm = reinterpret_cast<float*>(smem + 2 * EF_UNIT_BYTES);
  char* stg_all = smem + 2 * EF_UNIT_BYTES + EF_P_FLOATS * 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tl = lane & 31, h = lane >> 5, h4 = 4 * h, lane16 = 16 * lane;
  constexpr int WL = DW ? EF_WAVE_LDS_FFN_DW : EF_WAVE_LDS;
  char* stg = stg_all + wave * WL;
  // DW: accumulator tiles of dW2 (G = d_y2, X = h) and dW1 (G = d_hpre, X = x1) + bias column sums
  const int wr = wave >> 1, wc = wave & 1;
  ef_f32x16 dw2[4], dw1[4];
  float db2[2] = {0.f, 0.f}, db1[2] = {0.f, 0.f};
  if constexpr (DW) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { dw2[i] = ef_zero16(); dw1[i] = ef_zero16(); }
  }
  const char* tr_g = stg_all + EF_HALF_B * wr + ef_tr_lane(lane);                  // G region, this wave's row blocks
  const char* tr_x = stg_all + EF_TILE_B + EF_HALF_B * wc + ef_tr_lane(lane);     // X region, this wave's column blocks
  const char* tr_x1 = tr_x + EF_TILE_B;                                          // X1 region
  const int S = a.S;
  const int RW = 32 / S;
  const long long n_wt = (a.R + RW - 1) / RW;
  const long long n_it = (n_wt + EF_WAVES - 1) / EF_WAVES;
  for (int i = tid; i < EF_P_FLOATS; i += EF_THREADS) prm[i] = a.prm[i];
  if (blockIdx.x < n_it) ef_dma<EF_UNIT_BYTES>(a.wpack, smem, wave, lane16);
  EF_DMA_LANDED();
  __syncthreads();
  if (blockIdx.x < n_it) ef_dma<EF_UNIT_BYTES>(a.wpack + EF_UNIT_BYTES, smem + EF_UNIT_BYTES, wave, lane16);

  const int fb = EF_ROWB * tl + 16 * h;
  const char* pb = reinterpret_cast<const char*>(prm) + 16 * h;
  const float* pf = reinterpret_cast<const float*>(pb);           // parameter floats of this lane half (+ 4h applied)
  EfLaneAddr la;
  la.sw = stg + EF_STG_ROWB * tl + 8 * h;
  la.sr = stg + EF_STG_ROWB * (lane >> 3) + 16 * (lane & 7);
  la.go = (unsigned)(256 * (lane >> 3) + 16 * (lane & 7));
  const unsigned xoff = (unsigned)(256 * tl + 8 * h);
  // transposed reads of the staged [token][channel] rows: 16-lane group g reads channels 16g..16g+15, lane 4q+p of the
  // group addresses row q (+ 4j), columns 4p..4p+3
  const char* trb = stg + EF_STG_ROWB * ((lane & 15) >> 2) + 32 * (lane >> 4) + 8 * (lane & 3);
  const unsigned skey2 = rng_key(a.seed, a.rs2, 0u), skey3 = rng_key(a.seed, a.rs3, 0u);
  const float keep_scale = DROP ? a.inv_keep : 1.f;
  float q2[4] = {0.f, 0.f, 0.f, 0.f}, qt[4] = {0.f, 0.f, 0.f, 0.f};      // norm2 / tail: (d gamma, d beta) x channel halves

  for (long long it = blockIdx.x; it < n_it; it += gridDim.x) {
    EF_TILE_GEOM(it, row0, nvalid, tok0)
    const bool has_next = it + gridDim.x < n_it;
    const unsigned long long e_base = (unsigned long long)tok0 * EF_C + (unsigned long long)(tl * EF_C);
    const unsigned e_lo = (unsigned)((unsigned long long)tok0 * EF_C) + (unsigned)(tl * EF_C);
    ef_v8bf gf[8], zf[8];
    {
      const __amdgpu_buffer_rsrc_t rg = ef_tile_rsrc(a.g, tok0, nvalid), rz = ef_tile_rsrc(a.z2, tok0, nvalid);
      EF_LOAD_X(gf, rg)
      EF_LOAD_X(zf, rz)
    }

    // ---- tail LayerNorm backward and LayerNorm-2 backward (registers); gf becomes d_x2, then d_y2; dzf = d_z2
    float mu2, rstd2;
    ef_row_stats(zf, a.eps, mu2, rstd2);
    if (a.tail) {
      ef_v8bf x2f[8];
      const float nmr2 = -mu2 * rstd2;
#pragma unroll
      for (int f = 0; f < 8; ++f) {
        EF_FENCE();
        x2f[f] = ef_ln_apply(zf[f], nmr2, rstd2, pf + EF_P_G2 + 16 * f, pf + EF_P_BE2 + 16 * f);
      }
      float mut, rstdt;
      ef_row_stats(x2f, a.eps, mut, rstdt);
      ef_ln_bwd<false>(gf, x2f, a.beta_c, mut, rstdt, pf + EF_P_GT, gf, gf, la, trb, qt,
                       [](int, int, ef_f2 r) { return r; });
    }
    ef_v8bf dzf[8];
    {
      unsigned dkey = skey3;
      if (DROP && !a.small_idx) dkey = rng_key(a.seed, a.rs3, (unsigned)(e_base >> 32));
      int hs[4] = {0, 0, 0, 0};
      if constexpr (DROP == 1) {
#pragma unroll
        for (int m = 0; m < 4; ++m) hs[m] = (int)(mix32(((e_lo >> 5) + (unsigned)m) ^ dkey) >> h4);
      }
      // out = d_z2 (dzf); out2 = d_y2 = dropout mask of the norm2 site on d_z2 (fp32, before rounding), over gf
      ef_ln_bwd<true>(gf, zf, 1.f, mu2, rstd2, pf + EF_P_G2, dzf, gf, la, trb, q2, [&](int f, int d, ef_f2 r) {
        if constexpr (DROP == 1) {
          const int bit = 16 * (f & 1) + 8 * (d >> 1) + 2 * (d & 1);
          r.x = __uint_as_float(__float_as_uint(r.x) & ef_bitmask(hs[f >> 1], bit));
          r.y = __uint_as_float(__float_as_uint(r.y) & ef_bitmask(hs[f >> 1], bit + 1));
          r *= ef_splat(keep_scale);
        } else if constexpr (DROP != 0) {
          // (two elements of one group of four: the hash is recomputed per pair; not the reference's default p)
          float dm[4];
          drop_scale4_t<DROP>(dkey, e_lo + (unsigned)(16 * f + 8 * (d >> 1)) + (unsigned)h4, a.thresh, a.inv_keep, dm);
          r.x *= dm[2 * (d & 1)];
          r.y *= dm[2 * (d & 1) + 1];
        }
        return r;
      });
    }
    if constexpr (DW) ef_stage_tile(gf, la.sw);                          // d_y2 -> G region: operand of dW2 (and db2)
    else ef_store_rows(gf, la, ef_tile_rsrc(a.dy2, tok0, nvalid));

    // ---- x1 = LN1(z1) (recomputed), written as the X operand of dW1
    ef_v8bf x1f[8];
    {
      const __amdgpu_buffer_rsrc_t rz1 = ef_tile_rsrc(a.z1, tok0, nvalid);
      EF_LOAD_X(zf, rz1)
      float mu1, rstd1;
      ef_row_stats(zf, a.eps, mu1, rstd1);
      const float nmr1 = -mu1 * rstd1;
#pragma unroll
      for (int f = 0; f < 8; ++f) {
        EF_FENCE();
        x1f[f] = ef_ln_apply(zf[f], nmr1, rstd1, pf + EF_P_G1 + 16 * f, pf + EF_P_BE1 + 16 * f);
      }
      if constexpr (DW) ef_stage_tile(x1f, la.sw + 2 * EF_TILE_B);       // -> X1 region (read by the dW1 phase below)
      else ef_store_rows(x1f, la, ef_tile_rsrc(a.x1out, tok0, nvalid));
    }

    // ---- units 0, 1 (W1): h = drop(relu(W1 x1 + b1)) (recomputed), written as the X operand of dW2
    ef_v8bf hf[8];
    {
      unsigned dkey = skey2;
      if (DROP && !a.small_idx) dkey = rng_key(a.seed, a.rs2, (unsigned)(e_base >> 32));
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const char* wu = EF_UBUF(m >> 1);
        ef_f32x16 acc;
        EF_ACC_BIAS(acc, EF_P_B1 + 32 * m)
        EF_CHAIN(acc, wu, EF_PART_BYTES * (m & 1), x1f)
        if (m == 1) {
          // issued since the DMA of unit 1 (end of the previous tile): its 8 d_x1 stores, this tile's 48 loads (all
          // consumed by now) and the 16 stores of d_y2 and x1
          // (DW: nothing was stored since; the loads are consumed, so the count is zero)
          if constexpr (DW) { EF_UNIT_NEXT_K(0, true, 2, EF_UNIT_BYTES, EF_WAIT_VM(0)) }
          else { EF_UNIT_NEXT_K(0, true, 2, EF_UNIT_BYTES, EF_WAIT_VM(16)) }
        } else if (m == 3) {
          EF_UNIT_NEXT(1, true, 3, EF_UNIT_BYTES)
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = fmaxf(acc[i], 0.f);
        ef_drop_tile<DROP>(acc, dkey, e_lo + 32u * m, h4, a.thresh);
        if constexpr (DROP != 0) {
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[i] *= keep_scale;
        }
        hf[2 * m] = ef_pack<0>(acc);
        hf[2 * m + 1] = ef_pack<1>(acc);
        EF_FENCE();
      }
      if constexpr (DW) ef_stage_tile(hf, la.sw + EF_TILE_B);            // h -> X region
      else ef_store_rows(hf, la, ef_tile_rsrc(a.hout, tok0, nvalid));
    }

    // ---- units 2, 3 (W2^T): d_h = d_y2 W2, gated by the recomputed h: d_hpre (G operand of dW1, and db1)
    {
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const char* wu = EF_UBUF(2 + (m >> 1));
        ef_f32x16 acc = ef_zero16();
        EF_CHAIN(acc, wu, EF_PART_BYTES * (m & 1), gf)
        if (m == 1) {
          if constexpr (DW) {
            // this barrier publishes every wave's d_y2 and h tiles; the dW2 phase runs between it and the next one, after
            // which the G / X regions are free again
            EF_LDS_DONE();
            EF_UNIT_NEXT_K(2, true, 4, EF_UNIT_BYTES, EF_WAIT_VM(0))
            if (wc == 0) ef_dw_phase<WL, true>(dw2, db2, tr_g, tr_x);
            else ef_dw_phase<WL, false>(dw2, db2, tr_g, tr_x);
          } else {
            EF_UNIT_NEXT_K(2, true, 4, EF_UNIT_BYTES, EF_WAIT_VM(8))     // since the DMA of unit 3: the 8 stores of h
          }
        } else if (m == 3) {
          EF_UNIT_NEXT(3, true, 5, EF_UNIT_BYTES)
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = ef_bf(hf[2 * m + (i >> 3)], i & 7) > 0.f ? acc[i] * keep_scale : 0.f;
        hf[2 * m] = ef_pack<0>(acc);            // hf now holds d_hpre for the blocks done
        hf[2 * m + 1] = ef_pack<1>(acc);
        EF_FENCE();
      }
      if constexpr (DW) ef_stage_tile(hf, la.sw);                        // d_hpre -> G region (after the unit-3 barrier)
      else ef_store_rows(hf, la, ef_tile_rsrc(a.dhpre, tok0, nvalid));
    }

    // ---- units 4, 5 (W1^T): d_x1 = d_z2 + d_hpre W1
    {
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const char* wu = EF_UBUF(4 + (m >> 1));
        ef_f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = ef_bf(dzf[2 * m + (i >> 3)], i & 7);
        EF_CHAIN(acc, wu, EF_PART_BYTES * (m & 1), hf)
        if (m == 1) {
          if constexpr (DW) {
            EF_LDS_DONE();                                               // publishes d_hpre (x1 has been there since the top)
            EF_UNIT_NEXT_K(4, has_next, 0, EF_UNIT_BYTES, EF_WAIT_VM(0))
            if (wc == 0) ef_dw_phase<WL, true>(dw1, db1, tr_g, tr_x1);
            else ef_dw_phase<WL, false>(dw1, db1, tr_g, tr_x1);
          } else {
            EF_UNIT_NEXT_K(4, has_next, 0, EF_UNIT_BYTES, EF_WAIT_VM(8)) // since the DMA of unit 5: the 8 stores of d_hpre
          }
        } else if (m == 3) {
          EF_UNIT_NEXT(5, has_next, 1, EF_UNIT_BYTES)                    // (DW: every wave is past its dW1 reads here)
        }
        dzf[2 * m] = ef_pack<0>(acc);
        dzf[2 * m + 1] = ef_pack<1>(acc);
        EF_FENCE();
      }
      ef_store_rows(dzf, la, ef_tile_rsrc(a.dx1, tok0, nvalid));
    }
  }
  {
    const float q[8] = {q2[0], q2[2], q2[1], q2[3], qt[0], qt[2], qt[1], qt[3]};      // [d gamma2 | d beta2 | d gamma_t | d beta_t] x halves
    ef_ln_partials(q, reinterpret_cast<float*>(stg_all), a.lnp, wave, lane, tid);
  }
  if constexpr (DW) {       // this workgroup's partial weight gradients: [block][dW1 | dW2][128][128], [block][db1 | db2][128]
    float* wp = a.dwp + (size_t)blockIdx.x * 2 * EF_DW_TILE_FLOATS;
    ef_dw_write(dw1, wp, wr, wc, lane);
    ef_dw_write(dw2, wp + EF_DW_TILE_FLOATS, wr, wc, lane);
    if (wc == 0) {
      float* bp = a.dbp + (size_t)blockIdx.x * 2 * 128;
      ef_db_write(db1, bp, wr, lane);
      ef_db_write(db2, bp + 128, wr, lane);
    }
  }
}

// ---------------------------------------------------------------------------------------------- backward, attention half
// From d_x1 (gradient wrt x1, written by the feed-forward half) back to the layer input, everything recomputed from
// (x, z1) in registers, per wave tile of 32 tokens:
//     LayerNorm-1 backward -> d_z1 ;  d_y = mask1 . d_z1 ;  d_x(partial) = d_z1 (+ alpha * g) ; LayerNorm-1 parameter sums
//     per 32-channel block (one head of 32 dims or two of 16): K, Q, V, dO recomputed in BOTH MFMA orientations (a
//     product that contracts over tokens needs its operand with tokens on the accumulator rows);  per head:
//         S^T (+ the block-diagonal mask as one more k-step) -> P^T (exp2 softmax) -> Pd^T = dropout(P^T)
//         O^T = V^T Pd^T                       (written: operand of dWo)
//         dPd^T = V dO^T ; dS^T = P^T (mask . dPd^T - delta), delta = sum_key P^T mask . dPd^T
//         dQ^T = K-contracted with dS^T ;  Pd and dS with queries on the accumulator rows are the TRANSPOSES of the
//         packed Pd^T / dS^T tiles (through a 2.5 KiB wave-private LDS tile: 4 ds_write_b64 + 4 ds_read_b64_tr_b16 each;
//         round 2 recomputed the scores, the softmax and the dropout hash in the second orientation instead):
//         dV^T = dO-contracted with Pd ;  dK^T = Q-contracted with dS.
// Written: d_x (partial: the QKV projection's input gradient is accumulated into it by the next GEMM), d_y, o, d_qkv
// (operands of the weight-gradient GEMMs for Wo and W_in) and the workgroup's partial norm1 parameter sums.
// Units: per 32-channel block  (Wq rows | Wk rows)  and  (Wv rows | Wo^T rows), 17 KiB each, double-buffered.
struct EaArgs {
  const unsigned short *dx1, *z1, *x, *g;
  unsigned short *dx, *dy, *o, *dqkv;
  const char* wpack;
  const float* prm;
  long long R;
  int S;
  float alpha, eps;
  unsigned thresh;
  float inv_keep;
  unsigned long long seed;
  unsigned rs0, rs1;
  int small_idx;
  float* lnp;                    // [grid][4][128] partial sums: d gamma1, d beta1, 0, 0
  int dx_fold;                   // 1: d_x += d_qkv W_in inside the kernel (wpack stages 4..6 = the three [128,128] parts of W_in^T)
};

#ifndef EA_DX_LAST3
#define EA_DX_LAST3 1            // d_x epilogue: 1 = the last head block's d_qkv slices parked in LDS (no wait for its stores, no re-read)
#endif
#ifndef EA_TR_RECOMPUTE
#define EA_TR_RECOMPUTE 0        // 1: round 3's second-orientation recompute of K / Q / V / dO (A/B builds)
#endif
constexpr int EF_T_ROWB = 80;    // transpose tile: 32 rows x 32 bf16 (64 B) padded to 80 B
// Transpose of a 32 x 32 bf16 tile held as two packed fragments (lane (c, h), fragment s, element j <-> X[16s + 8(j>>2)
// + 4h + (j&3)][c]) through LDS: returns the fragments of X^T in the same form.  tw = tile + 80 * tl + 8 * h (write
// base), tr = tile + 80 * (4 * h + ((lane & 15) >> 2)) + 32 * ((lane >> 4) & 1) + 8 * (lane & 3) (transposed-read base).
__device__ __forceinline__ void ef_transpose32(ef_v8bf f0, ef_v8bf f1, char* tw, const char* tr, ef_v8bf& t0, ef_v8bf& t1) {
  const uint4 a = __builtin_bit_cast(uint4, f0), b = __builtin_bit_cast(uint4, f1);
  // row tl of Z = X^T rows: elements r = 16s + 8g + 4h .. +3 at byte 2r
  *reinterpret_cast<uint2*>(tw) = make_uint2(a.x, a.y);
  *reinterpret_cast<uint2*>(tw + 16) = make_uint2(a.z, a.w);
  *reinterpret_cast<uint2*>(tw + 32) = make_uint2(b.x, b.y);
  *reinterpret_cast<uint2*>(tw + 48) = make_uint2(b.z, b.w);
  uint2 r[4];
#pragma unroll
  for (int k = 0; k < 4; ++k)        // k = 2 s' + g': rows 16 s' + 8 g' + 4 h' .. +3 of Z, this group's 16 columns
    r[k] = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((ef_lds_v4s_ptr)(tr + EF_T_ROWB * (16 * (k >> 1) + 8 * (k & 1)))));
  t0 = __builtin_bit_cast(ef_v8bf, make_uint4(r[0].x, r[0].y, r[1].x, r[1].y));
  t1 = __builtin_bit_cast(ef_v8bf, make_uint4(r[2].x, r[2].y, r[3].x, r[3].y));
}

// a [32 token][32 channel] block held as two packed fragments -> columns col .. col+31 of a [T, ld] bf16 tensor, via a
// wave-private restage (64-byte rows padded to 80 B); rs = descriptor of the tile's valid rows (ef_tile_rsrc_ld)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t ef_tile_rsrc_ld(const unsigned short* base, long long tok0, int nvalid, int ld) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(base + tok0 * ld), 0, nvalid * (ld * 2), 0x00020000);
}
__device__ __forceinline__ void ef_store_block32(ef_v8bf f0, ef_v8bf f1, char* bw, const char* br, __amdgpu_buffer_rsrc_t rs,
                                                 unsigned go, int ld2 /* row bytes */, int col2 /* column byte offset */) {
  const uint4 v0 = __builtin_bit_cast(uint4, f0), v1 = __builtin_bit_cast(uint4, f1);
  // fragment s: channels 16s + 4h + 0..3 (.xy), 16s + 8 + 4h + 0..3 (.zw); bw = tile + 80 * tl + 8 * h
  *reinterpret_cast<uint2*>(bw) = make_uint2(v0.x, v0.y);
  *reinterpret_cast<uint2*>(bw + 16) = make_uint2(v0.z, v0.w);
  *reinterpret_cast<uint2*>(bw + 32) = make_uint2(v1.x, v1.y);
  *reinterpret_cast<uint2*>(bw + 48) = make_uint2(v1.z, v1.w);
#pragma unroll
  for (int p = 0; p < 2; ++p) {      // br = tile + 80 * (lane >> 2) + 16 * (lane & 3): rows (lane >> 2) + 16 p
    const uint4 v = *reinterpret_cast<const uint4*>(br + 16 * p * EF_T_ROWB);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(ef_u4, v), rs, go + (unsigned)(16 * p * ld2 + col2), 0, 0);   // (immediate soffset: see ef_store_rows)
  }
}

template <int HD, int DROP /* 0 = off, else hash bits per element: 1 | 8 | 16 (common.hpp) */>
__global__ void __launch_bounds__(EF_THREADS, EF_WG_PER_CU) k_encoder_bwd_attn(const EaArgs a_) {
  EaArgs a = a_;
  a.seed = live_seed(a_.seed);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* prm = reinterpret_cast<float*>(smem + 2 * EF_UNIT_BYTES);
  char* stg_all = smem + 2 * EF_UNIT_BYTES + EF_P_FLOATS * 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tl = lane & 31, h = lane >> 5, h4 = 4 * h, lane16 = 16 * lane;
  char* stg = stg_all + wave * EF_WAVE_LDS;
  constexpr int NH = EF_C / HD, HB = 32 / HD;
  const float scale = HD == 32 ? 0.17677669529663687f : 0.25f;
  const float qscale = scale * 1.4426950408889634f;
  const int S = a.S;
  const int RW = 32 / S;
  const long long n_wt = (a.R + RW - 1) / RW;
  const long long n_it = (n_wt + EF_WAVES - 1) / EF_WAVES;
  for (int i = tid; i < EF_P_FLOATS; i += EF_THREADS) prm[i] = a.prm[i];
  if (blockIdx.x < n_it) ef_dma<EF_UNIT_BYTES>(a.wpack, smem, wave, lane16);
  EF_DMA_LANDED();
  __syncthreads();
  if (blockIdx.x < n_it) ef_dma<EF_UNIT_BYTES>(a.wpack + EF_UNIT_BYTES, smem + EF_UNIT_BYTES, wave, lane16);

  const int fb = EF_ROWB * tl + 16 * h;
  const char* pb = reinterpret_cast<const char*>(prm) + 16 * h;
  const float* pf = reinterpret_cast<const float*>(pb);
  EfLaneAddr la;
  la.sw = stg + EF_STG_ROWB * tl + 8 * h;
  la.sr = stg + EF_STG_ROWB * (lane >> 3) + 16 * (lane & 7);
  la.go = (unsigned)(256 * (lane >> 3) + 16 * (lane & 7));
  const unsigned xoff = (unsigned)(256 * tl + 8 * h);
  const char* trb = stg + EF_STG_ROWB * ((lane & 15) >> 2) + 32 * (lane >> 4) + 8 * (lane & 3);
  // 32 x 32 tiles (transposes, block stores) live in staging region B
  char* tile = stg + 4608;
  char* tw = tile + EF_T_ROWB * tl + 8 * h;
  const char* ttr = tile + EF_T_ROWB * (4 * h + ((lane & 15) >> 2)) + 32 * ((lane >> 4) & 1) + 8 * (lane & 3);
  const char* tbr = tile + EF_T_ROWB * (lane >> 2) + 16 * (lane & 3);
  const int q_row = tl / S;
  const int row_lo = q_row * S;
  ef_v8bf uf;
#pragma unroll
  for (int j = 0; j < 8; ++j) uf[j] = (__bf16)((8 * h + j) == q_row ? 16.f : 0.f);
  const unsigned skey0 = rng_key(a.seed, a.rs0, 0u), skey1 = rng_key(a.seed, a.rs1, 0u);
  const float keep_scale = DROP ? a.inv_keep : 1.f;
  float q1[4] = {0.f, 0.f, 0.f, 0.f};

  for (long long it = blockIdx.x; it < n_it; it += gridDim.x) {
    EF_TILE_GEOM(it, row0, nvalid, tok0)
    const bool has_next = it + gridDim.x < n_it;
    const unsigned long long e_base = (unsigned long long)tok0 * EF_C + (unsigned long long)(tl * EF_C);
    const unsigned e_lo = (unsigned)((unsigned long long)tok0 * EF_C) + (unsigned)(tl * EF_C);

    // ---- LayerNorm-1 backward: d_z1 -> d_x (partial) and d_y
    ef_v8bf dyf[8], xf[8];
    {
      ef_v8bf zf[8], df[8];
      const __amdgpu_buffer_rsrc_t rd = ef_tile_rsrc(a.dx1, tok0, nvalid), rz = ef_tile_rsrc(a.z1, tok0, nvalid);
      EF_LOAD_X(df, rd)
      EF_LOAD_X(zf, rz)
      const bool with_g = a.alpha != 0.f;
      if (with_g) {
        const __amdgpu_buffer_rsrc_t rg = ef_tile_rsrc(a.g, tok0, nvalid);
        EF_LOAD_X(xf, rg)
      } else {
#pragma unroll
        for (int f = 0; f < 8; ++f) xf[f] = __builtin_bit_cast(ef_v8bf, make_uint4(0u, 0u, 0u, 0u));
      }
      float mu, rstd;
      ef_row_stats(zf, a.eps, mu, rstd);
      unsigned dkey = skey1;
      if (DROP && !a.small_idx) dkey = rng_key(a.seed, a.rs1, (unsigned)(e_base >> 32));
      int hs[4] = {0, 0, 0, 0};
      if constexpr (DROP == 1) {
#pragma unroll
        for (int m = 0; m < 4; ++m) hs[m] = (int)(mix32(((e_lo >> 5) + (unsigned)m) ^ dkey) >> h4);
      }
      // out2 = d_y = mask1 . d_z1 (over the registers of d_x1); the plain d_z1 (out) is only needed as d_x (partial) =
      // d_z1 + alpha * g, formed from the rounded out afterwards
      ef_ln_bwd<true>(df, zf, 1.f, mu, rstd, pf + EF_P_G1, zf, dyf, la, trb, q1, [&](int f, int d, ef_f2 r) {
        if constexpr (DROP == 1) {
          const int bit = 16 * (f & 1) + 8 * (d >> 1) + 2 * (d & 1);
          r.x = __uint_as_float(__float_as_uint(r.x) & ef_bitmask(hs[f >> 1], bit));
          r.y = __uint_as_float(__float_as_uint(r.y) & ef_bitmask(hs[f >> 1], bit + 1));
          r *= ef_splat(keep_scale);
        } else if constexpr (DROP != 0) {
          float dm[4];
          drop_scale4_t<DROP>(dkey, e_lo + (unsigned)(16 * f + 8 * (d >> 1)) + (unsigned)h4, a.thresh, a.inv_keep, dm);
          r.x *= dm[2 * (d & 1)];
          r.y *= dm[2 * (d & 1) + 1];
        }
        return r;
      });
      if (with_g) {
#pragma unroll
        for (int f = 0; f < 8; ++f) {
          uint4 w;
#pragma unroll
          for (int d = 0; d < 4; ++d) w[d] = ef_pk(ef_fma2(ef_unpk(ef_dw(xf[f], d)), ef_splat(a.alpha), ef_unpk(ef_dw(zf[f], d))));
          zf[f] = __builtin_bit_cast(ef_v8bf, w);
        }
      }
      ef_store_rows(zf, la, ef_tile_rsrc(a.dx, tok0, nvalid));
      ef_store_rows(dyf, la, ef_tile_rsrc(a.dy, tok0, nvalid));
    }
    {
      const __amdgpu_buffer_rsrc_t rx = ef_tile_rsrc(a.x, tok0, nvalid);
      EF_LOAD_X(xf, rx)
    }
    const __amdgpu_buffer_rsrc_t ro = ef_tile_rsrc_ld(a.o, tok0, nvalid, EF_C), rq = ef_tile_rsrc_ld(a.dqkv, tok0, nvalid, 3 * EF_C);
    const unsigned go_o = (unsigned)((lane >> 2) * (EF_C * 2) + 16 * (lane & 3)), go_q = (unsigned)((lane >> 2) * (3 * EF_C * 2) + 16 * (lane & 3));
    const unsigned long long att_u = (unsigned long long)row0 * (unsigned long long)(NH * S * S);
    const unsigned att_l = (unsigned)(q_row * NH * S * S + (tl - row_lo) * S + h4 - row_lo);

#pragma unroll 1
    for (int blk = 0; blk < 4; ++blk) {
      // transposed-orientation chain: D[token rows, 32 columns] += A . W^T, weight fragments at WT + fb + IMM + 32 ks
#define EA_CHAIN_TR(ACC, WT, IMM, AOP)                                                                \
      _Pragma("unroll") for (int ks = 0; ks < 8; ++ks) {                                              \
        ACC = EF_MFMA(AOP[ks], ef_frag((WT) + (IMM) + 32 * ks, fb), ACC);                             \
        if (ks == 3) EF_MID_FENCE();                                                                  \
      }
#define EA_SPLAT(ACC, V)                                                                              \
      _Pragma("unroll") for (int i = 0; i < 16; ++i) ACC[i] = (V);
      const float* pbias = pf + 32 * blk;                 // (+ 4h applied) row biases of this block: + EF_P_BIN + 128 part
      const float* plane = prm + 32 * blk + tl;           // per-lane (column) biases
      ef_f32x16 acc;
      // ---- unit 2 blk: Wq rows | Wk rows
      const char* wa = EF_UBUF(0);
      // K^T, Q^T [32 d (rows), 32 tokens]
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 b = *reinterpret_cast<const float4*>(pbias + EF_P_BIN + 128 + 8 * g);
        acc[4 * g] = b.x; acc[4 * g + 1] = b.y; acc[4 * g + 2] = b.z; acc[4 * g + 3] = b.w;
      }
      EF_CHAIN(acc, wa, EF_PART_BYTES, xf)
      const ef_v8bf kf0 = ef_pack<0>(acc), kf1 = ef_pack<1>(acc);
      EF_FENCE();
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 b = *reinterpret_cast<const float4*>(pbias + EF_P_BIN + 8 * g);
        acc[4 * g] = b.x; acc[4 * g + 1] = b.y; acc[4 * g + 2] = b.z; acc[4 * g + 3] = b.w;
      }
      EF_CHAIN(acc, wa, 0, xf)
      // K [key (rows), d], Q [q (rows), d]: operands of the products that contract over tokens.  EA_TR_RECOMPUTE = 0 (round
      // 4): the 32 x 32 TRANSPOSES of the packed K^T / Q^T / V^T / dO^T tiles through the wave's LDS tile (4 ds_write_b64 +
      // 4 ds_read_b64_tr_b16 each, no VALU) instead of a second projection chain per operand in the other MFMA orientation
      // (8 MFMAs + 8 weight fragments + bias splat + pack each: 128 MFMAs and 128 ds_read_b128 per tile less)
#if EA_TR_RECOMPUTE
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] *= qscale;
      const ef_v8bf qf0 = ef_pack<0>(acc), qf1 = ef_pack<1>(acc);
      EF_FENCE();
      EA_SPLAT(acc, plane[EF_P_BIN + 128])
      EA_CHAIN_TR(acc, wa, EF_PART_BYTES, xf)
      const ef_v8bf kt0 = ef_pack<0>(acc), kt1 = ef_pack<1>(acc);
      EF_FENCE();
      EA_SPLAT(acc, plane[EF_P_BIN])
      EA_CHAIN_TR(acc, wa, 0, xf)
      const ef_v8bf qt0 = ef_pack<0>(acc), qt1 = ef_pack<1>(acc);
#else
      ef_v8bf kt0, kt1, qt0, qt1;
      ef_transpose32(ef_pack<0>(acc), ef_pack<1>(acc), tw, ttr, qt0, qt1);        // the UNSCALED Q (dK = Q-contracted dS)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] *= qscale;
      const ef_v8bf qf0 = ef_pack<0>(acc), qf1 = ef_pack<1>(acc);
      ef_transpose32(kf0, kf1, tw, ttr, kt0, kt1);
#endif
      // boundary: wait for unit 2 blk + 1; since its DMA was issued: the 8 block stores of the previous head block
      if (blk == 0) { EF_UNIT_NEXT_K(0, true, 2 * blk + 2, EF_UNIT_BYTES, EF_WAIT_VM(0)) }
      else if (blk < 3) { EF_UNIT_NEXT_K(0, true, 2 * blk + 2, EF_UNIT_BYTES, EF_WAIT_VM(8)) }
      else if (a.dx_fold) { EF_UNIT_NEXT_K(0, true, 8, EF_UNIT_BYTES, EF_WAIT_VM(8)) }       // -> the d_x units
      else { EF_UNIT_NEXT_K(0, has_next, 0, EF_UNIT_BYTES, EF_WAIT_VM(8)) }

      // ---- unit 2 blk + 1: Wv rows | Wo^T rows
      const char* wb = EF_UBUF(1);
#if EA_TR_RECOMPUTE
      EA_SPLAT(acc, plane[EF_P_BIN + 256])
      EA_CHAIN_TR(acc, wb, 0, xf)                                        // V [key (rows), d]
      const ef_v8bf vt0 = ef_pack<0>(acc), vt1 = ef_pack<1>(acc);
      EF_FENCE();
#endif
      acc = ef_zero16();
      EF_CHAIN(acc, wb, EF_PART_BYTES, dyf)                              // dO^T [d, q] = Wo^T rows . d_y
      const ef_v8bf dof0 = ef_pack<0>(acc), dof1 = ef_pack<1>(acc);
      EF_FENCE();
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 b = *reinterpret_cast<const float4*>(pbias + EF_P_BIN + 256 + 8 * g);
        acc[4 * g] = b.x; acc[4 * g + 1] = b.y; acc[4 * g + 2] = b.z; acc[4 * g + 3] = b.w;
      }
      EF_CHAIN(acc, wb, 0, xf)                                           // V^T [d, key]
      const ef_v8bf vf0 = ef_pack<0>(acc), vf1 = ef_pack<1>(acc);
#if EA_TR_RECOMPUTE
      EF_FENCE();
      acc = ef_zero16();
      EA_CHAIN_TR(acc, wb, EF_PART_BYTES, dyf)                           // dO [q (rows), d]
      const ef_v8bf dotf0 = ef_pack<0>(acc), dotf1 = ef_pack<1>(acc);
#else
      ef_v8bf vt0, vt1, dotf0, dotf1;
      ef_transpose32(vf0, vf1, tw, ttr, vt0, vt1);                       // V [key (rows), d]
      ef_transpose32(dof0, dof1, tw, ttr, dotf0, dotf1);                 // dO [q (rows), d]
#endif
      if (blk < 3) { EF_UNIT_NEXT(1, true, 2 * blk + 3, EF_UNIT_BYTES) }
      else if (a.dx_fold) { EF_UNIT_NEXT(1, true, 9, EF_UNIT_BYTES) }
      else { EF_UNIT_NEXT(1, has_next, 1, EF_UNIT_BYTES) }

      // ---- per head
      ef_f32x16 o_all, dq_all, dk_all, dv_all;
#pragma unroll
      for (int hh = 0; hh < HB; ++hh) {
        ef_f32x16 pt = EF_MFMA(uf, uf, ef_zero16());
        if constexpr (HB == 1) {
          pt = EF_MFMA(kf0, qf0, pt);
          pt = EF_MFMA(kf1, qf1, pt);
        } else {
          pt = hh == 0 ? EF_MFMA(kf0, qf0, pt) : EF_MFMA(kf1, qf1, pt);
        }
        // P^T: exp2 softmax down the keys of the query's table row (other rows sit 256 lower: they underflow)
        float mx = fmaxf(fmaxf(pt[0], pt[1]), pt[2]);
#pragma unroll
        for (int i = 3; i < 15; i += 2) mx = fmaxf(fmaxf(mx, pt[i]), pt[i + 1]);
        mx = fmaxf(mx, pt[15]);
        mx = fmaxf(mx, ef_xor32(mx));
        float l = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          pt[i] = __builtin_amdgcn_exp2f(pt[i] - mx);
          l += pt[i];
        }
        l += ef_xor32(l);
        const float inv = __builtin_amdgcn_rcpf(l);
#pragma unroll
        for (int i = 0; i < 16; ++i) pt[i] *= inv;
        // dropout mask of this (row, head): AND masks for the 16 (key, query) pairs of this lane
        unsigned dmask[16];      // (DROP 8 / 16: kept; DROP 1: a bit of hsw per register, re-extracted at each use)
        int hsw = 0;
#define EA_DMASK(I) (DROP == 1 ? ef_bitmask(hsw, ((I) & 3) + 8 * ((I) >> 2)) : dmask[I])
        if constexpr (DROP == 1) {
          unsigned al = att_l;
          asm volatile("" : "+v"(al));
          const unsigned hoff = (unsigned)((blk * HB + hh) * S * S);
          const unsigned ab32 = (unsigned)att_u + al + hoff;
          unsigned k0 = skey0, k1 = skey0;
          if (!a.small_idx) {
            const unsigned long long ab = att_u + (unsigned long long)(al + hoff);
            k0 = rng_key(a.seed, a.rs0, (unsigned)(ab >> 32));
            k1 = rng_key(a.seed, a.rs0, (unsigned)((ab + 32) >> 32));
          }
          const unsigned g0 = ab32 >> 5;
          const unsigned w0 = mix32(g0 ^ k0), w1 = mix32(((g0 + 1u) & 0x07ffffffu) ^ k1);
          hsw = (int)__builtin_amdgcn_alignbit(w1, w0, ab32 & 31u);
        } else if constexpr (DROP != 0) {
          const int head = blk * HB + hh;
          const unsigned long long blk0 = att_u + (unsigned long long)((q_row * NH + head) * S * S);
          const unsigned lo0 = (unsigned)blk0 + (unsigned)((tl - row_lo) * S);
          const unsigned key0 = rng_key(a.seed, a.rs0, (unsigned)(blk0 >> 32));
          const unsigned key1 = rng_key(a.seed, a.rs0, (unsigned)(blk0 >> 32) + 1u);
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int key = (i & 3) + 8 * (i >> 2) + h4;
            const unsigned lo = lo0 + (unsigned)(key - row_lo);
            dmask[i] = drop_scale_key_t<DROP>(lo < (unsigned)blk0 ? key1 : key0, lo, a.thresh, 1.f) != 0.f ? 0xffffffffu : 0u;
          }
        }
        // Pd^T (packed: k = key) and O^T = V^T Pd^T
        ef_v8bf pd0, pd1;
        {
          ef_f32x16 pd;
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            if constexpr (DROP != 0) pd[i] = __uint_as_float(__float_as_uint(pt[i]) & EA_DMASK(i)) * keep_scale;
            else pd[i] = pt[i];
          }
          pd0 = ef_pack<0>(pd); pd1 = ef_pack<1>(pd);
        }
        {
          ef_f32x16 ot = EF_MFMA(vt0, pd0, ef_zero16());
          ot = EF_MFMA(vt1, pd1, ot);
          if constexpr (HB == 1) o_all = ot;
          else {
#pragma unroll
            for (int i = 0; i < 8; ++i) o_all[8 * hh + i] = ot[8 * hh + i];
          }
        }
        // dPd^T = V dO^T (contract over this head's dims) ; dS^T
        ef_v8bf ds0, ds1;
        {
          ef_f32x16 dp = ef_zero16();
          if constexpr (HB == 1) {
            dp = EF_MFMA(vf0, dof0, dp);
            dp = EF_MFMA(vf1, dof1, dp);
          } else {
            dp = hh == 0 ? EF_MFMA(vf0, dof0, dp) : EF_MFMA(vf1, dof1, dp);
          }
          if constexpr (DROP != 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) dp[i] = __uint_as_float(__float_as_uint(dp[i]) & EA_DMASK(i)) * keep_scale;
          }
          float delta = 0.f;
#pragma unroll
          for (int i = 0; i < 16; ++i) delta += pt[i] * dp[i];
          delta += ef_xor32(delta);
#pragma unroll
          for (int i = 0; i < 16; ++i) dp[i] = (pt[i] * scale) * (dp[i] - delta);     // (1/sqrt(d) of dQ, dK folded in)
          ds0 = ef_pack<0>(dp); ds1 = ef_pack<1>(dp);
        }
        // dQ^T [d, q] = K-contracted (over keys) with dS^T
        {
          ef_f32x16 dq = EF_MFMA(kt0, ds0, ef_zero16());
          dq = EF_MFMA(kt1, ds1, dq);
          if constexpr (HB == 1) dq_all = dq;
          else {
#pragma unroll
            for (int i = 0; i < 8; ++i) dq_all[8 * hh + i] = dq[8 * hh + i];
          }
        }
        // Pd, dS with queries on the accumulator rows: transposes of the packed tiles
        ef_v8bf pq0, pq1, sq0, sq1;
        ef_transpose32(pd0, pd1, tw, ttr, pq0, pq1);
        ef_transpose32(ds0, ds1, tw, ttr, sq0, sq1);
        {
          ef_f32x16 dv = EF_MFMA(dotf0, pq0, ef_zero16());                 // dV^T [d, key] = sum_q dO[q, d] Pd[q, key]
          dv = EF_MFMA(dotf1, pq1, dv);
          ef_f32x16 dk = EF_MFMA(qt0, sq0, ef_zero16());                   // dK^T [d, key] = sum_q Q[q, d] dS[q, key]
          dk = EF_MFMA(qt1, sq1, dk);
          if constexpr (HB == 1) { dv_all = dv; dk_all = dk; }
          else {
#pragma unroll
            for (int i = 0; i < 8; ++i) { dv_all[8 * hh + i] = dv[8 * hh + i]; dk_all[8 * hh + i] = dk[8 * hh + i]; }
          }
        }
      }
      ef_store_block32(ef_pack<0>(o_all), ef_pack<1>(o_all), tw, tbr, ro, go_o, EF_C * 2, 64 * blk);
      const ef_v8bf dqf0 = ef_pack<0>(dq_all), dqf1 = ef_pack<1>(dq_all), dkf0 = ef_pack<0>(dk_all), dkf1 = ef_pack<1>(dk_all),
                    dvf0 = ef_pack<0>(dv_all), dvf1 = ef_pack<1>(dv_all);
      ef_store_block32(dqf0, dqf1, tw, tbr, rq, go_q, 3 * EF_C * 2, 64 * blk);
      ef_store_block32(dkf0, dkf1, tw, tbr, rq, go_q, 3 * EF_C * 2, 256 + 64 * blk);
      ef_store_block32(dvf0, dvf1, tw, tbr, rq, go_q, 3 * EF_C * 2, 512 + 64 * blk);
#if EA_DX_LAST3
      if (blk == 3 && a.dx_fold) {      // parked lane-linear in the wave's idle staging bytes (region A and the tail of region B)
        *reinterpret_cast<uint4*>(stg + lane16) = __builtin_bit_cast(uint4, dqf0);
        *reinterpret_cast<uint4*>(stg + lane16 + 1024) = __builtin_bit_cast(uint4, dqf1);
        *reinterpret_cast<uint4*>(stg + lane16 + 2048) = __builtin_bit_cast(uint4, dkf0);
        *reinterpret_cast<uint4*>(stg + lane16 + 3072) = __builtin_bit_cast(uint4, dkf1);
        *reinterpret_cast<uint4*>(stg + lane16 + 7168) = __builtin_bit_cast(uint4, dvf0);
        *reinterpret_cast<uint4*>(stg + lane16 + 8192) = __builtin_bit_cast(uint4, dvf1);
      }
#endif
    }
    // ---- d_x += d_qkv W_in (round 4; was a separate 0.8 ms GEMM per big launch that re-read d_qkv and d_x from HBM):
    //      d_x^T[c, token] = partial + sum over the 384 gradient channels of W_in^T[c, k] d_qkv[token, k], as three
    //      128-deep parts (q | k | v) against units 8..13 = the [128,128] tiles of W_in^T, two 64-row units each.  The
    //      wave reads back what it has just stored (the partial d_x and its d_qkv rows, L2-resident) in B-fragment order;
    //      96 MFMAs per tile, no VALU beyond the unpack of the partial and the final pack.
    if (a.dx_fold) {
      // the stores of head blocks 0..2 and of the partial d_x (and the DMA of unit 9) have landed once at most the 8 block
      // stores of head block 3 are outstanding; that block's d_qkv slices are still in registers (last3)
      EF_WAIT_VM(EA_DX_LAST3 ? 8 : 0);
      ef_v8bf bq[8], bn[8];
      ef_f32x16 dxa[4];
      {
        const __amdgpu_buffer_rsrc_t rdx = ef_tile_rsrc(a.dx, tok0, nvalid);
        EF_LOAD_X(bn, rdx)
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int i = 0; i < 16; ++i) dxa[m][i] = ef_bf(bn[2 * m + (i >> 3)], i & 7);
      }
      const unsigned qoff = (unsigned)(3 * EF_C * 2 * tl + 8 * h);
#define EA_LOAD_PART(DST, P)                                                                          \
      if (EA_DX_LAST3) {      /* k-steps 6, 7 = head block 3: parked in LDS by the loop above, never re-read from memory */ \
        DST[6] = ef_frag(stg + ((P) == 2 ? 7168 : 2048 * (P)), lane16);                               \
        DST[7] = ef_frag(stg + ((P) == 2 ? 8192 : 2048 * (P) + 1024), lane16);                        \
      }                                                                                               \
      _Pragma("unroll") for (int ks = 0; ks < (EA_DX_LAST3 ? 6 : 8); ++ks) {                          \
        const uint2 lo = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(rq, qoff, 256 * (P) + 32 * ks, 0));      \
        const uint2 hi = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(rq, qoff, 256 * (P) + 32 * ks + 16, 0)); \
        DST[ks] = __builtin_bit_cast(ef_v8bf, make_uint4(lo.x, lo.y, hi.x, hi.y));                   \
      }
      EA_LOAD_PART(bq, 0)
#pragma unroll
      for (int e = 0; e < 6; ++e) {            // unit 8 + e in buffer e & 1: rows 64 (e & 1) .. of part e >> 1
        const char* wu = EF_UBUF(e);
        EF_CHAIN(dxa[2 * (e & 1)], wu, 0, bq)
        EF_FENCE();
        EF_CHAIN(dxa[2 * (e & 1) + 1], wu, EF_PART_BYTES, bq)
        if (e == 0 || e == 2) {
          // nothing newer than the DMA of unit 9 + e is in flight
          EF_UNIT_NEXT_K(e, true, 10 + e, EF_UNIT_BYTES, EF_WAIT_VM(0))
          if (e == 0) { EA_LOAD_PART(bn, 1) } else { EA_LOAD_PART(bn, 2) }       // the next part: 12 loads the next boundary lets fly
        } else if (e == 1 || e == 3) {
          EF_UNIT_NEXT_K(e, true, 10 + e, EF_UNIT_BYTES, EF_WAIT_VM(EA_DX_LAST3 ? 12 : 16))
#pragma unroll
          for (int ks = 0; ks < 8; ++ks) bq[ks] = bn[ks];
        } else if (e == 4) {
          EF_UNIT_NEXT_K(4, has_next, 0, EF_UNIT_BYTES, EF_WAIT_VM(0))
        } else {
          EF_UNIT_NEXT_K(5, has_next, 1, EF_UNIT_BYTES, EF_WAIT_VM(0))
        }
      }
#undef EA_LOAD_PART
      ef_v8bf zo[8];
#pragma unroll
      for (int m = 0; m < 4; ++m) { zo[2 * m] = ef_pack<0>(dxa[m]); zo[2 * m + 1] = ef_pack<1>(dxa[m]); }
      ef_store_rows(zo, la, ef_tile_rsrc(a.dx, tok0, nvalid));
    }
  }
  {
    const float q[8] = {q1[0], q1[2], q1[1], q1[3], 0.f, 0.f, 0.f, 0.f};
    ef_ln_partials(q, reinterpret_cast<float*>(stg_all), a.lnp, wave, lane, tid);
  }
}

// wpack stage blk (0..3) = Wq rows | Wk rows | Wv rows | Wo^T rows (32 rows each, rows 32 blk ..), k-permuted images
__global__ void __launch_bounds__(256) k_encoder_pack_attn_bwd(const unsigned short* __restrict__ w_in,
                                                                const unsigned short* __restrict__ w_o_t, int ld_ot,
                                                                const unsigned short* __restrict__ w_in_t, int ld_it,
                                                                char* __restrict__ wpack) {
  const int stage = blockIdx.x;
  char* dst = wpack + (size_t)stage * EF_STAGE_BYTES;
  if (stage >= 4) {        // stages 4..6: W_in^T[:, 128 (stage - 4) ..]: rows = input channel c, k = gradient channel of q | k | v
    for (int p = threadIdx.x; p < 128 * 17; p += blockDim.x) {
      const int row = p / 17, c = p - 17 * row;
      ef_pack_row(dst + EF_ROWB * row, w_in_t + (size_t)row * ld_it + 128 * (stage - 4), c);
    }
    return;
  }
  for (int p = threadIdx.x; p < 128 * 17; p += blockDim.x) {
    const int row = p / 17, c = p - 17 * row, part = row >> 5, r = row & 31;
    const unsigned short* wrow = part < 3 ? w_in + (size_t)(128 * part + 32 * stage + r) * EF_C
                                          : w_o_t + (size_t)(32 * stage + r) * ld_ot;
    ef_pack_row(dst + EF_PART_BYTES * part + EF_ROWB * r, wrow, c);
  }
}

// ---------------------------------------------------------------------------------------------- LayerNorm parameter gradients
// The chained backward kernels leave one partial row [4][128] per workgroup (ef_ln_partials); k_ef_reduce sums the rows
// in block order (deterministic, no float atomics).  (Round 2 had a separate streaming pass over g, z2 / d_x1, z1 for
// these sums: 0.83 ms per step.)
struct EgOut { float* dst[4]; };
__global__ void __launch_bounds__(1024) k_ef_reduce(const float* __restrict__ partials, int nblk, const EgOut o, int accumulate) {
  __shared__ float red[8][128];
  const int v = blockIdx.x, c = threadIdx.x & 127, grp = threadIdx.x >> 7;
  float t = 0.f;
  if (o.dst[v]) {       // fixed order: deterministic.  Eight loads in flight per thread (a dependent chain of 64 took 20 us)
    const float* p = partials + v * 128 + c;
    float u[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int b = grp;
    for (; b + 56 < nblk; b += 64) {
#pragma unroll
      for (int k = 0; k < 8; ++k) u[k] += p[(size_t)(b + 8 * k) * 512];
    }
    for (; b < nblk; b += 8) u[0] += p[(size_t)b * 512];
    t = ((u[0] + u[1]) + (u[2] + u[3])) + ((u[4] + u[5]) + (u[6] + u[7]));
  }
  red[grp][c] = t;
  __syncthreads();
  if (grp == 0 && o.dst[v]) {
    float r = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) r += red[k][c];
    o.dst[v][c] = accumulate ? o.dst[v][c] + r : r;
  }
}

// Sum of the DW kernels' per-workgroup partial weight gradients in block order (deterministic): part [nblk][nw][128][128],
// partb [nblk][nw][128].  Blocks [0, 16 nw): 1024 floats of weight blockIdx / 16 each (thread = one float4, four groups of
// 256 threads split the block range); blocks [16 nw, 17 nw): the bias vector of weight blockIdx - 16 nw.
struct EwOut { float* w[4]; float* b[4]; };
__global__ void __launch_bounds__(1024) k_ef_dw_reduce(const float* __restrict__ part, const float* __restrict__ partb, int nblk,
                                                       int nw, const EwOut o, int accumulate) {
  __shared__ float4 red[4][256];
  const int grp = threadIdx.x >> 8, t = threadIdx.x & 255;
  if ((int)blockIdx.x < 16 * nw) {
    const int w = blockIdx.x >> 4;
    float* dst = o.w[w];
    if (!dst) return;
    const size_t e = (size_t)(blockIdx.x & 15) * 1024 + 4 * t;
    const float* p = part + (size_t)w * EF_DW_TILE_FLOATS + e;
    const size_t stride = (size_t)nw * EF_DW_TILE_FLOATS;
    float4 u[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) u[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    int b = grp;
    for (; b + 12 < nblk; b += 16) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float4 v = *reinterpret_cast<const float4*>(p + (size_t)(b + 4 * k) * stride);
        u[k].x += v.x; u[k].y += v.y; u[k].z += v.z; u[k].w += v.w;
      }
    }
    for (; b < nblk; b += 4) {
      const float4 v = *reinterpret_cast<const float4*>(p + (size_t)b * stride);
      u[0].x += v.x; u[0].y += v.y; u[0].z += v.z; u[0].w += v.w;
    }
    red[grp][t] = make_float4((u[0].x + u[1].x) + (u[2].x + u[3].x), (u[0].y + u[1].y) + (u[2].y + u[3].y),
                              (u[0].z + u[1].z) + (u[2].z + u[3].z), (u[0].w + u[1].w) + (u[2].w + u[3].w));
    __syncthreads();
    if (grp == 0) {
      float4 r = red[0][t];
#pragma unroll
      for (int k = 1; k < 4; ++k) { r.x += red[k][t].x; r.y += red[k][t].y; r.z += red[k][t].z; r.w += red[k][t].w; }
      float4* d = reinterpret_cast<float4*>(dst + e);
      if (accumulate) { const float4 c = *d; r.x += c.x; r.y += c.y; r.z += c.z; r.w += c.w; }
      *d = r;
    }
  } else {
    const int w = blockIdx.x - 16 * nw;
    float* dst = o.b[w];
    if (!dst) return;
    float* redf = reinterpret_cast<float*>(red);        // [8][128]
    const int c = threadIdx.x & 127, g8 = threadIdx.x >> 7;
    float v = 0.f;
    for (int b = g8; b < nblk; b += 8) v += partb[((size_t)b * nw + w) * 128 + c];
    redf[g8 * 128 + c] = v;
    __syncthreads();
    if (g8 == 0) {
      float r = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) r += redf[k * 128 + c];
      dst[c] = accumulate ? dst[c] + r : r;
    }
  }
}

// ---------------------------------------------------------------------------------------------- generic tile pack
struct EfTileList {
  const unsigned short* src[8];    // [128 rows][128] bf16 row-major tiles (row stride ld[i])
  int ld[8];
  int n;
};
// wpack stage i = k-permuted, swizzled LDS image of tile i (see k_encoder_pack)
__global__ void __launch_bounds__(256) k_encoder_pack_tiles(const EfTileList t, char* __restrict__ wpack) {
  const int stage = blockIdx.x;
  if (stage >= t.n) return;
  char* dst = wpack + (size_t)stage * EF_STAGE_BYTES;
  const unsigned short* w = t.src[stage];
  const int ld = t.ld[stage];
  for (int p = threadIdx.x; p < 128 * 17; p += blockDim.x) {
    const int row = p / 17, c = p - 17 * row;
    ef_pack_row(dst + EF_ROWB * row, w + (size_t)row * ld, c);
  }
}

}  // namespace tg

using namespace tg;

extern "C" int64_t tg_encoder_pack_bytes(void) { return (int64_t)EF_NSTAGE * EF_STAGE_BYTES; }
extern "C" int64_t tg_encoder_stage_bytes(void) { return EF_STAGE_BYTES; }
#if EF_ABL & 1024
extern "C" int tg_encoder_dbg_read(unsigned long long* host, int n_slots) {      // diagnostic build only
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(tg::ef_dbg), sizeof(unsigned long long) * tg::EF_NSEC * n_slots);
}
#endif
extern "C" int64_t tg_encoder_prm_floats(void) { return EF_P_FLOATS; }

// Builds the LDS weight images + the fp32 parameter block of one ColumnTransformerLayer call (bf16 weights [out,in]
// row-major; biases / LayerNorm parameters fp32; gt/bt may be NULL when there is no tail norm).
extern "C" int tg_encoder_pack(const void* w_in, const void* w_o, const void* w1, const void* w2, const float* b_in,
                               const float* b_o, const float* g1, const float* be1, const float* b1, const float* b2,
                               const float* g2, const float* be2, const float* gt, const float* bt, void* wpack,
                               float* prm, void* stream) {
  TG_CHECK(w_in && w_o && w1 && w2 && b_in && b_o && g1 && be1 && b1 && b2 && g2 && be2 && wpack && prm,
           "tg_encoder_pack: null operand");
  TG_CHECK(((reinterpret_cast<uintptr_t>(w_in) | reinterpret_cast<uintptr_t>(w_o) | reinterpret_cast<uintptr_t>(w1) |
             reinterpret_cast<uintptr_t>(w2) | reinterpret_cast<uintptr_t>(wpack)) & 15) == 0,
           "tg_encoder_pack: operands must be 16-byte aligned");
  hipLaunchKernelGGL(k_encoder_pack, dim3(EF_NSTAGE), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)w_in,
                     (const unsigned short*)w_o, (const unsigned short*)w1, (const unsigned short*)w2, b_in, b_o, g1, be1,
                     b1, b2, g2, be2, gt, bt, (char*)wpack, prm);
  TG_LAUNCH_CHECK();
  return 0;
}

extern "C" int32_t tg_encoder_fused_supported(int32_t S, int32_t C, int32_t H, int32_t FF) {
  return C == 128 && FF == 128 && (H == 4 || H == 8) && S >= 2 && S <= 32;     // (S = 1: the mask k-step holds 16 table rows per tile)
}

// out [R,S,128] = encoder layer (+ tail) of x [R,S,128], bf16; z1 / z2 (pre-LayerNorm sums, what a recomputing backward
// needs) optional.  rs[4] = dropout streams (attention, norm1, ffn, norm2).
extern "C" int tg_encoder_fwd_bf16(const void* x, void* out, void* z1, void* z2, const void* wpack, const float* prm,
                                   int64_t R, int32_t S, int32_t H, int32_t tail, float alpha, float beta_c, float eps,
                                   float p_drop, uint64_t seed, const uint32_t* rs, void* stream) {
  TG_CHECK(tg_encoder_fused_supported(S, 128, H, 128), "tg_encoder_fwd_bf16: unsupported geometry S=%d H=%d", S, H);
  TG_CHECK(x && out && wpack && prm && rs, "tg_encoder_fwd_bf16: null operand");
  TG_CHECK(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(z1) |
             reinterpret_cast<uintptr_t>(z2) | reinterpret_cast<uintptr_t>(wpack)) & 15) == 0,
           "tg_encoder_fwd_bf16: operands must be 16-byte aligned");
  if (R <= 0) return 0;
  EfArgs a;
  a.x = (const unsigned short*)x; a.out = (unsigned short*)out; a.z1 = (unsigned short*)z1; a.z2 = (unsigned short*)z2;
  a.o_in = nullptr; a.st1 = nullptr;
  a.wpack = (const char*)wpack; a.prm = prm; a.R = R; a.S = S; a.tail = tail; a.alpha = alpha; a.beta_c = beta_c; a.eps = eps;
  a.thresh = p_drop > 0.f ? drop_threshold(p_drop) : 0u;
  a.inv_keep = p_drop < 1.f ? 1.f / (1.f - p_drop) : 0.f;
  a.seed = seed; a.rs0 = rs[0]; a.rs1 = rs[1]; a.rs2 = rs[2]; a.rs3 = rs[3];
  // every dropout element index (R*S*128 of the linear sites, R*H*S*S of the attention probabilities, + one tile) < 2^32
  a.small_idx = ((double)(R + 32) * S * 128.0 < 4294967296.0 && (double)(R + 32) * H * S * S < 4294967296.0) ? 1 : 0;
  const int RW = 32 / S;
  const long long n_wt = (R + RW - 1) / RW, n_it = (n_wt + EF_WAVES - 1) / EF_WAVES;
  static int n_cu = 0;
  if (!n_cu) {
    hipDeviceProp_t prop;
    int dev = 0;
    (void)hipGetDevice(&dev);
    n_cu = hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 256;
  }
  const long long slots = (long long)n_cu * (EF_WAVES >= 8 ? 1 : EF_WG_PER_CU);
  const unsigned grid = (unsigned)(n_it < slots ? n_it : slots);
  const size_t lds = 2 * EF_UNIT_BYTES + EF_P_FLOATS * 4 + EF_WAVES * 8192;
  // DROP template value: 0 = no dropout, else the hash bits per element the threshold allows (common.hpp)
  const int drop = drop_mode(a.thresh);
#define EF_LAUNCH_FWD(HD_, DR_)                                                                                \
  {                                                                                                            \
    static bool attr_done = false;                                                                             \
    if (!attr_done) {                                                                                          \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_encoder_fwd<HD_, DR_>),                       \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                         \
      attr_done = true;                                                                                        \
    }                                                                                                          \
    hipLaunchKernelGGL((k_encoder_fwd<HD_, DR_>), dim3(grid), dim3(EF_THREADS), lds, (hipStream_t)stream, a);  \
  }
  if (H == 4) {
    if (drop == 0) EF_LAUNCH_FWD(32, 0) else if (drop == 1) EF_LAUNCH_FWD(32, 1) else if (drop == 8) EF_LAUNCH_FWD(32, 8) else EF_LAUNCH_FWD(32, 16)
  } else {
    if (drop == 0) EF_LAUNCH_FWD(16, 0) else if (drop == 1) EF_LAUNCH_FWD(16, 1) else if (drop == 8) EF_LAUNCH_FWD(16, 8) else EF_LAUNCH_FWD(16, 16)
  }
#undef EF_LAUNCH_FWD
  TG_LAUNCH_CHECK();
  return 0;
}

// Everything behind the attention for token rows of ANY length (k_encoder_fwd<., ., ATT = false>): x, o [T,128] ->
//     z1 = x + drop(o Wo^T + b_o), x1 = LN1(z1), z2 = x1 + drop(W2 drop(relu(W1 x1 + b1)) + b2), out = (tail) LN2(z2)
// on the flat token stream cut into R pseudo rows of S tokens (2 <= S <= 32, R * S = T): out, optional z1 / z2 and the
// LayerNorm-1 statistics st1 [T][2] (mean, rstd) that the op-by-op attention-half backward reads.  rs[1..3] = the norm1 /
// ffn / norm2 dropout streams (rs[0], the attention's, is not used).  wpack / prm: tg_encoder_pack.
extern "C" int tg_encoder_ffn_fwd_bf16(const void* x, const void* o, void* out, void* z1, void* z2, float* st1, const void* wpack,
                                       const float* prm, int64_t R, int32_t S, int32_t tail, float alpha, float beta_c,
                                       float eps, float p_drop, uint64_t seed, const uint32_t* rs, void* stream) {
  TG_CHECK(S >= 2 && S <= 32, "tg_encoder_ffn_fwd_bf16: unsupported S=%d", S);
  TG_CHECK(x && o && out && wpack && prm && rs, "tg_encoder_ffn_fwd_bf16: null operand");
  TG_CHECK(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(o) | reinterpret_cast<uintptr_t>(out) |
             reinterpret_cast<uintptr_t>(z1) | reinterpret_cast<uintptr_t>(z2) | reinterpret_cast<uintptr_t>(wpack)) & 15) == 0 &&
           (reinterpret_cast<uintptr_t>(st1) & 7) == 0, "tg_encoder_ffn_fwd_bf16: operands must be 16-byte aligned");
  if (R <= 0) return 0;
  EfArgs a;
  a.x = (const unsigned short*)x; a.out = (unsigned short*)out; a.z1 = (unsigned short*)z1; a.z2 = (unsigned short*)z2;
  a.o_in = (const unsigned short*)o; a.st1 = st1;
  a.wpack = (const char*)wpack; a.prm = prm; a.R = R; a.S = S; a.tail = tail; a.alpha = alpha; a.beta_c = beta_c; a.eps = eps;
  a.thresh = p_drop > 0.f ? drop_threshold(p_drop) : 0u;
  a.inv_keep = p_drop < 1.f ? 1.f / (1.f - p_drop) : 0.f;
  a.seed = seed; a.rs0 = rs[0]; a.rs1 = rs[1]; a.rs2 = rs[2]; a.rs3 = rs[3];
  a.small_idx = ((double)(R + 32) * S * 128.0 < 4294967296.0) ? 1 : 0;
  const int RW = 32 / S;
  const long long n_wt = (R + RW - 1) / RW, n_it = (n_wt + EF_WAVES - 1) / EF_WAVES;
  static int n_cu = 0;
  if (!n_cu) {
    hipDeviceProp_t prop;
    int dev = 0;
    (void)hipGetDevice(&dev);
    n_cu = hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 256;
  }
  const long long slots = (long long)n_cu * (EF_WAVES >= 8 ? 1 : EF_WG_PER_CU);
  const unsigned grid = (unsigned)(n_it < slots ? n_it : slots);
  const size_t lds = 2 * EF_UNIT_BYTES + EF_P_FLOATS * 4 + EF_WAVES * 8192;
  const int drop = drop_mode(a.thresh);
#define EF_LAUNCH_FFN(DR_)                                                                                     \
  {                                                                                                            \
    static bool attr_done = false;                                                                             \
    if (!attr_done) {                                                                                          \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_encoder_fwd<32, DR_, false>),                 \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                         \
      attr_done = true;                                                                                        \
    }                                                                                                          \
    hipLaunchKernelGGL((k_encoder_fwd<32, DR_, false>), dim3(grid), dim3(EF_THREADS), lds, (hipStream_t)stream, a); \
  }
  if (drop == 0) EF_LAUNCH_FFN(0) else if (drop == 1) EF_LAUNCH_FFN(1) else if (drop == 8) EF_LAUNCH_FFN(8) else EF_LAUNCH_FFN(16)
#undef EF_LAUNCH_FFN
  TG_LAUNCH_CHECK();
  return 0;
}

// Backward weight images: stage i = tile i ([128,128] bf16 row-major, row stride ld[i]); the feed-forward half takes
// (W1, W2^T, W1^T) in this order.  wpack: n * 32 KiB.
extern "C" int tg_encoder_pack_tiles(const void* const* tiles, const int32_t* ld, int32_t n, void* wpack, void* stream) {
  TG_CHECK(tiles && ld && wpack && n >= 1 && n <= 8, "tg_encoder_pack_tiles: bad arguments (n=%d)", n);
  EfTileList t;
  t.n = n;
  for (int i = 0; i < 8; ++i) { t.src[i] = i < n ? (const unsigned short*)tiles[i] : nullptr; t.ld[i] = i < n ? ld[i] : 0; }
  for (int i = 0; i < n; ++i)
    TG_CHECK(t.src[i] && (reinterpret_cast<uintptr_t>(t.src[i]) & 7) == 0 && t.ld[i] >= 128 && t.ld[i] % 4 == 0,
             "tg_encoder_pack_tiles: tile %d must be 8-byte aligned with a row stride >= 128", i);
  hipLaunchKernelGGL(k_encoder_pack_tiles, dim3(n), dim3(256), 0, (hipStream_t)stream, t, (char*)wpack);
  TG_LAUNCH_CHECK();
  return 0;
}

static size_t ef_lds_bytes() { return 2 * EF_UNIT_BYTES + EF_P_FLOATS * 4 + EF_WAVES * EF_WAVE_LDS; }
static size_t ef_lds_bytes_ffn_dw() { return 2 * EF_UNIT_BYTES + EF_P_FLOATS * 4 + EF_WAVES * EF_WAVE_LDS_FFN_DW; }
static unsigned ef_grid(long long R, int S, int wg_per_cu = EF_WG_PER_CU) {
  const int RW = 32 / S;
  const long long n_wt = (R + RW - 1) / RW, n_it = (n_wt + EF_WAVES - 1) / EF_WAVES;
  static int n_cu = 0;
  if (!n_cu) {
    hipDeviceProp_t prop;
    int dev = 0;
    (void)hipGetDevice(&dev);
    n_cu = hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 256;
  }
  const long long slots = (long long)n_cu * (EF_WAVES >= 8 ? 1 : wg_per_cu);
  return (unsigned)(n_it < slots ? n_it : slots);
}

// Feed-forward half of the layer backward (see k_encoder_bwd_ffn).  g = d out [R,S,128]; z1, z2 from the forward;
// wpack = tg_encoder_pack_tiles(W1, W2^T, W1^T); prm = the forward's parameter block.  Writes d_x1 and the weight-gradient
// operands d_y2, h, d_hpre, x1 (all [R,S,128] bf16).  rs: the forward's dropout streams.
static int ef_launch_bwd_ffn(bool dw, const void* g, const void* z1, const void* z2, void* dx1, void* dy2, void* hout,
                             void* dhpre, void* x1out, const void* wpack, const float* prm, int64_t R, int32_t S,
                             int32_t tail, float beta_c, float eps, float p_drop, uint64_t seed, const uint32_t* rs,
                             float* lnp, float* dwp, float* dbp, void* stream) {
  EbArgs a;
  a.g = (const unsigned short*)g; a.z1 = (const unsigned short*)z1; a.z2 = (const unsigned short*)z2;
  a.dx1 = (unsigned short*)dx1; a.dy2 = (unsigned short*)dy2; a.hout = (unsigned short*)hout; a.dhpre = (unsigned short*)dhpre;
  a.x1out = (unsigned short*)x1out; a.wpack = (const char*)wpack; a.prm = prm; a.R = R; a.S = S; a.tail = tail;
  a.beta_c = beta_c; a.eps = eps;
  a.thresh = p_drop > 0.f ? drop_threshold(p_drop) : 0u;
  a.inv_keep = p_drop < 1.f ? 1.f / (1.f - p_drop) : 0.f;
  a.seed = seed; a.rs2 = rs[2]; a.rs3 = rs[3];
  a.small_idx = (double)(R + 32) * S * 128.0 < 4294967296.0 ? 1 : 0;
  a.lnp = lnp; a.dwp = dwp; a.dbp = dbp;
  const size_t lds = dw ? ef_lds_bytes_ffn_dw() : ef_lds_bytes();
  const unsigned grid = ef_grid(R, S, dw ? 1 : EF_WG_PER_CU);
  const int drop = drop_mode(a.thresh);      // DROP template value (common.hpp)
#define EF_LAUNCH_B(DR_, DW_)                                                                                  \
  {                                                                                                            \
    static bool attr_done = false;                                                                             \
    if (!attr_done) {                                                                                          \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_encoder_bwd_ffn<DR_, DW_>),                   \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                         \
      attr_done = true;                                                                                        \
    }                                                                                                          \
    hipLaunchKernelGGL((k_encoder_bwd_ffn<DR_, DW_>), dim3(grid), dim3(EF_THREADS), lds, (hipStream_t)stream, a);  \
  }
  if (dw) {
    if (drop == 0) EF_LAUNCH_B(0, true) else if (drop == 1) EF_LAUNCH_B(1, true) else if (drop == 8) EF_LAUNCH_B(8, true) else EF_LAUNCH_B(16, true)
  } else {
    if (drop == 0) EF_LAUNCH_B(0, false) else if (drop == 1) EF_LAUNCH_B(1, false) else if (drop == 8) EF_LAUNCH_B(8, false) else EF_LAUNCH_B(16, false)
  }
#undef EF_LAUNCH_B
  TG_LAUNCH_CHECK();
  return 0;
}

extern "C" int tg_encoder_bwd_ffn_bf16(const void* g, const void* z1, const void* z2, void* dx1, void* dy2, void* hout,
                                       void* dhpre, void* x1out, const void* wpack, const float* prm, int64_t R, int32_t S,
                                       int32_t tail, float beta_c, float eps, float p_drop, uint64_t seed,
                                       const uint32_t* rs, float* lnp, void* stream) {
  TG_CHECK(S >= 2 && S <= 32, "tg_encoder_bwd_ffn_bf16: unsupported S=%d", S);
  TG_CHECK(g && z1 && z2 && dx1 && dy2 && hout && dhpre && x1out && wpack && prm && rs && lnp, "tg_encoder_bwd_ffn_bf16: null operand");
  if (R <= 0) return 0;
  return ef_launch_bwd_ffn(false, g, z1, z2, dx1, dy2, hout, dhpre, x1out, wpack, prm, R, S, tail, beta_c, eps, p_drop, seed,
                           rs, lnp, nullptr, nullptr, stream);
}

// The same half with the FFN weight gradients INSIDE (k_encoder_bwd_ffn<., true>): writes d_x1 only; the operand tensors
// d_y2, h, d_hpre, x1 stay in LDS.  nblk = tg_encoder_dw_blocks(R, S) workgroups leave lnp [nblk][4][128] (as above),
// dwp [nblk][2][128][128] (partial dW1 | dW2, [out][in]) and dbp [nblk][2][128] (partial db1 | db2); tg_encoder_dw_reduce
// sums them in block order.
extern "C" int64_t tg_encoder_dw_blocks(int64_t R, int32_t S) { return S >= 1 && S <= 32 && R > 0 ? (int64_t)ef_grid(R, S, 1) : 0; }
extern "C" int tg_encoder_bwd_ffn_dw_bf16(const void* g, const void* z1, const void* z2, void* dx1, const void* wpack,
                                          const float* prm, int64_t R, int32_t S, int32_t tail, float beta_c, float eps,
                                          float p_drop, uint64_t seed, const uint32_t* rs, float* lnp, float* dwp,
                                          float* dbp, void* stream) {
  TG_CHECK(S >= 2 && S <= 32, "tg_encoder_bwd_ffn_dw_bf16: unsupported S=%d", S);
  TG_CHECK(g && z1 && z2 && dx1 && wpack && prm && rs && lnp && dwp && dbp, "tg_encoder_bwd_ffn_dw_bf16: null operand");
  if (R <= 0) return 0;
  return ef_launch_bwd_ffn(true, g, z1, z2, dx1, nullptr, nullptr, nullptr, nullptr, wpack, prm, R, S, tail, beta_c, eps,
                           p_drop, seed, rs, lnp, dwp, dbp, stream);
}
// out_w[i] (fp32 [128][128]) / out_b[i] (fp32 [128]) (+)= sum over blocks of weight i's partials; NULL = not wanted
extern "C" int tg_encoder_dw_reduce(const float* dwp, const float* dbp, int64_t nblk, int32_t nw, float* const* out_w,
                                    float* const* out_b, int32_t accumulate, void* stream) {
  TG_CHECK(dwp && dbp && out_w && out_b && nblk >= 0 && nw >= 1 && nw <= 4, "tg_encoder_dw_reduce: bad arguments");
  if (nblk == 0) return 0;
  EwOut o;
  for (int i = 0; i < 4; ++i) { o.w[i] = i < nw ? out_w[i] : nullptr; o.b[i] = i < nw ? out_b[i] : nullptr; }
  for (int i = 0; i < nw; ++i)
    TG_CHECK((reinterpret_cast<uintptr_t>(o.w[i]) & 15) == 0, "tg_encoder_dw_reduce: weight gradient %d must be 16-byte aligned", i);
  hipLaunchKernelGGL(k_ef_dw_reduce, dim3(17 * nw), dim3(1024), 0, (hipStream_t)stream, dwp, dbp, (int)nblk, (int)nw, o, accumulate);
  TG_LAUNCH_CHECK();
  return 0;
}

// The chained backward kernels leave per-workgroup partial LayerNorm parameter sums (lnp [nblk][4][128], nblk =
// tg_encoder_ln_partial_blocks(R, S)); this sums them in block order (deterministic) into out[i] (fp32 [128] or NULL;
// accumulate = 1 adds: .grad semantics).  Feed-forward half: d gamma2, d beta2, d gamma_t, d beta_t; attention half:
// d gamma1, d beta1.
extern "C" int64_t tg_encoder_ln_partial_blocks(int64_t R, int32_t S) { return S >= 1 && S <= 32 && R > 0 ? (int64_t)ef_grid(R, S) : 0; }
extern "C" int tg_encoder_ln_reduce(const float* lnp, int64_t nblk, float* const* out, int32_t accumulate, void* stream) {
  TG_CHECK(lnp && out && nblk >= 0, "tg_encoder_ln_reduce: bad arguments");
  if (nblk == 0) return 0;
  EgOut o;
  for (int i = 0; i < 4; ++i) o.dst[i] = out[i];
  hipLaunchKernelGGL(k_ef_reduce, dim3(4), dim3(1024), 0, (hipStream_t)stream, lnp, (int)nblk, o, accumulate);
  TG_LAUNCH_CHECK();
  return 0;
}

// Attention half of the layer backward (see k_encoder_bwd_attn; 4 heads).  wpack (4 x 32 KiB) is built here from W_in
// [384,128] and Wo^T [128,128] (row stride ld_ot).  d_x1 from tg_encoder_bwd_ffn_bf16; g = d out (read when alpha != 0).
// Writes dx (partial: add d_qkv W_in to it), dy, o ([R,S,128]) and dqkv ([R,S,384]).
extern "C" int tg_encoder_bwd_attn_bf16(const void* dx1, const void* z1, const void* x, const void* g, void* dx, void* dy,
                                        void* o, void* dqkv, const void* w_in, const void* w_o_t, int32_t ld_ot,
                                        const void* w_in_t, int32_t ld_it, void* wpack, const float* prm, int64_t R,
                                        int32_t S, int32_t H, float alpha, float eps, float p_drop, uint64_t seed,
                                        const uint32_t* rs, float* lnp, void* stream) {
  TG_CHECK(S >= 2 && S <= 32 && (H == 4 || H == 8), "tg_encoder_bwd_attn_bf16: unsupported geometry S=%d H=%d", S, H);
  TG_CHECK(dx1 && z1 && x && dx && dy && o && dqkv && w_in && w_o_t && wpack && prm && rs && lnp && (g || alpha == 0.f),
           "tg_encoder_bwd_attn_bf16: null operand");
  if (R <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  TG_CHECK(!w_in_t || (ld_it >= 384 && ld_it % 4 == 0 && (reinterpret_cast<uintptr_t>(w_in_t) & 7) == 0),
           "tg_encoder_bwd_attn_bf16: W_in^T must be an 8-byte aligned [128, >=384] matrix (ld_it=%d)", ld_it);
  hipLaunchKernelGGL(k_encoder_pack_attn_bwd, dim3(w_in_t ? 7 : 4), dim3(256), 0, st, (const unsigned short*)w_in,
                     (const unsigned short*)w_o_t, ld_ot, (const unsigned short*)w_in_t, ld_it, (char*)wpack);
  EaArgs a;
  a.dx1 = (const unsigned short*)dx1; a.z1 = (const unsigned short*)z1; a.x = (const unsigned short*)x;
  a.g = (const unsigned short*)g; a.dx = (unsigned short*)dx; a.dy = (unsigned short*)dy; a.o = (unsigned short*)o;
  a.dqkv = (unsigned short*)dqkv; a.wpack = (const char*)wpack; a.prm = prm; a.R = R; a.S = S; a.alpha = alpha; a.eps = eps;
  a.thresh = p_drop > 0.f ? drop_threshold(p_drop) : 0u;
  a.inv_keep = p_drop < 1.f ? 1.f / (1.f - p_drop) : 0.f;
  a.seed = seed; a.rs0 = rs[0]; a.rs1 = rs[1];
  a.small_idx = ((double)(R + 32) * S * 128.0 < 4294967296.0 && (double)(R + 32) * H * S * S < 4294967296.0) ? 1 : 0;
  a.lnp = lnp;
  a.dx_fold = w_in_t ? 1 : 0;
  const size_t lds = ef_lds_bytes();
  const unsigned grid = ef_grid(R, S);
  const int drop = drop_mode(a.thresh);      // DROP template value (common.hpp)
#define EF_LAUNCH_B(HD_, DR_)                                                                                  \
  {                                                                                                            \
    static bool attr_done = false;                                                                             \
    if (!attr_done) {                                                                                          \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_encoder_bwd_attn<HD_, DR_>),                  \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                         \
      attr_done = true;                                                                                        \
    }                                                                                                          \
    hipLaunchKernelGGL((k_encoder_bwd_attn<HD_, DR_>), dim3(grid), dim3(EF_THREADS), lds, st, a);              \
  }
  if (H == 4) {
    if (drop == 0) EF_LAUNCH_B(32, 0) else if (drop == 1) EF_LAUNCH_B(32, 1) else if (drop == 8) EF_LAUNCH_B(32, 8) else EF_LAUNCH_B(32, 16)
  } else {
    if (drop == 0) EF_LAUNCH_B(16, 0) else if (drop == 1) EF_LAUNCH_B(16, 1) else if (drop == 8) EF_LAUNCH_B(16, 8) else EF_LAUNCH_B(16, 16)
  }
#undef EF_LAUNCH_B
  TG_LAUNCH_CHECK();
  return 0;
}

