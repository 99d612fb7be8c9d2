// PNA post projection with the degree scalers folded in (reference: torch_geometric PNAConv.forward, reached from
// src/nn/models/fused.py:204-214 — post_nns over [x | scalers x aggregators], then lin):
//
//     out[r, :] = b + x[r] Wx^T + agg[r] W_0^T + amp(r) * (agg[r] W_1^T) + att(r) * (agg[r] W_2^T)
//
// agg [R, K] (K = 4F = 512: mean | max | min | std), x [R, F], F = 128, (amp, att) fp32 per row.
//
// 206 GFLOP per launch at the bench subgraph (R = 515 k) against 0.8 GB of operands: the one GEMM of the path that is
// MFMA-bound, not HBM-bound.  tg_gemm_nt_scaled_bf16 (gemm_nt.hip) multiplies the X tile by the row scale on its way
// into LDS — three register-staged, VALU-rescaled copies of every X chunk, two workgroup barriers per 128-deep chunk —
// and ran at 0.24 MFMA utilisation (370-400 us).  This kernel keeps THREE accumulator sets instead (one per scaler) and
// combines them once in the epilogue, so the operands go HBM/L2 -> LDS by LDS-DMA untouched:
//   * workgroup = 8 waves = 256 rows x 128 output columns; wave tile 64 rows x 64 columns x 3 sets = 192 accumulator
//     registers (two waves per SIMD, 256 registers each),
//   * a stage = 64 k of the row tile: X piece 256 x 64 (32 KiB) + the three W_s pieces 128 x 64 (3 x 16 KiB), fetched by
//     `global_load_lds_dwordx4` (10 instructions per wave and stage, each 8 rows x one whole 128-byte line), two stages
//     in LDS (160 KiB, all of it): the DMA of stage t+1 is issued at the top of stage t — 3 072 MFMA cycles per SIMD of
//     cover.  (First version: 32-k stages, 64-byte rows, three-deep ring — 640 half-used lines per 32 k through the TA
//     at ~4.5 cycles each made the data path alone take 226 us; whole lines halve that.)
//   * one workgroup barrier per stage,
//   * 128-byte LDS rows, chunk position = chunk ^ ((row >> 1) & 7): the 16 rows of a ds_read_b128 lane group fall on 16
//     different 16-byte slots; the DMA applies the swizzle on the SOURCE side (lane -> global chunk),
//   * x Wx^T rides in accumulator set 0 as four more stages (X from x, W from Wx), bias in the epilogue: the separate
//     GEMM launch and the read-modify-write of `out` go.
// Measured at R = 524 k (tools/nt_scaled_probe.py): 299 us = 746 TFLOP/s (the two launches it replaces: 380 + 75 us;
// the first version with 32-k stages and half-used lines: 313 us).  PS_ABL builds of that first version: without the
// MFMAs 226 us, without the DMA 216 us, without DMA and fragment reads 190 us — and 190 us of pure MFMA + barriers is
// already 74 % of what the MFMA pipe delivers at the clock these kernels run at (~1.5 GHz under MFMA load).
// Also measured on the 64-k version and not kept: a persistent workgroup per CU with stage 0 of the next tile in flight
// under the epilogue (296 us: the per-tile dispatch / prologue is not what is missing) and the next stage's ten DMA
// instructions spread between the MFMA groups instead of issued at the stage top (318 us: each asm statement with its
// M0 save / restore and `s_nop` breaks the rolling fragment prefetch).  What is left is producer waves (the consumers
// would have to fit 232 registers) and a deeper ring than 160 KB allows at this tile.
#include "common.hpp"
#include "../../include/tabgnn_hip.h"

#ifndef PS_ABL
#define PS_ABL 0      // diagnostic builds: 1 = no MFMAs, 2 = no DMA (stale LDS), 4 = no stage barrier
#endif

namespace tg {

typedef __bf16 ps_v8bf __attribute__((ext_vector_type(8)));
typedef float ps_f32x16 __attribute__((ext_vector_type(16)));

constexpr int PS_ROWS = 256, PS_F = 128, PS_THREADS = 512, PS_NSTAGE = 2, PS_BK = 64;
constexpr int PS_ROWB = PS_BK * 2;              // 128-byte LDS rows = whole cache lines per DMA lane group
constexpr int PS_XB = PS_ROWS * PS_ROWB;        // X piece: 256 rows x 64 bf16 (32 KiB)
constexpr int PS_WB = PS_F * PS_ROWB;           // one W_s piece: 128 rows x 64 bf16 (16 KiB)
constexpr int PS_STAGE = PS_XB + 3 * PS_WB;     // 80 KiB
constexpr int PS_LDS = PS_NSTAGE * PS_STAGE;    // 160 KiB: the whole LDS of a CU (the output tile is restaged in it afterwards)
constexpr int PS_NDMA = PS_STAGE / 1024 / (PS_THREADS / 64);      // DMA instructions per wave and stage (10)

struct PsArgs {
  const char* agg; const char* x; const char* wcat; const char* wx;
  const float* bias; const float* scales;
  unsigned short* out;
  long long R, ld_agg, ld_x, ld_out;            // row strides in elements
  int K;
  int nct;                                      // column tiles of a wider output walked by ONE workgroup (1 = the forward)
  long long w_ct_bytes, out_ct_cols;            // column tile ct: wcat += ct * w_ct_bytes, out += ct * out_ct_cols
};

// LDS-DMA pieces of a stage.  M0 (the LDS base) is written once per GROUP of pieces whose LDS destinations are 1 KiB
// apart; piece i of a group goes to M0 + 1024 (i - MID) through the instruction's immediate offset, which the hardware
// adds to the global address too — the piece's scalar base is moved back by the same amount.  M0 is read when a
// `global_load_lds` ISSUES (tools/hwtests/lds_dma_race.hip part B: M0 rewritten 0 wait states behind a DMA under full
// load, 5.2e8 pieces, none lands at the new value), so the next group's M0 write needs no distance from the previous
// group's last DMA.  (Rounds 2-4 suspected a late M0 read behind the column-transformer kernels' wrong tiles and kept 32
// wait states here; that defect was LDS reads in flight across a raw s_barrier — encoder_fused.hip:EF_WAIT_VM —, which
// this kernel's stage boundary excludes the same way: PS_STAGE_BODY waits lgkmcnt(0) in front of its barrier.)
// hipcc does not use M0 in this file (checked by tests/test_cabi_and_host.py).
#define PS_M0_SET(ADDR) asm volatile("s_mov_b32 m0, %0\n\ts_nop 4" ::"s"(ADDR) : "memory")
template <int OFF>
__device__ __forceinline__ void ps_dma(unsigned voff, const char* sbase) {
  // (s_nop 4: a v_readlane reload of the address pair right in front needs 5 wait states before a VMEM read; see ef_dma)
  asm volatile("s_nop 4\n\tglobal_load_lds_dwordx4 %0, %1 offset:%2" ::"v"(voff), "s"(sbase - OFF), "n"(OFF) : "memory");
}

__device__ __forceinline__ ps_v8bf ps_frag(const char* p) {
  if (PS_ABL & 8) { ps_v8bf z; for (int i = 0; i < 8; ++i) z[i] = (__bf16)(float)(int)((unsigned long long)p & 15); return z; }   // no LDS reads
  return __builtin_bit_cast(ps_v8bf, *reinterpret_cast<const uint4*>(p));
}

__device__ __forceinline__ int ps_out_off(int row, int ch) {       // restaged output tile: 256-byte rows, 16 chunks
  return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3)));
}

template <bool MULTI /* several column tiles per workgroup, no x / bias terms: the input-gradient use */>
__global__ void __launch_bounds__(PS_THREADS, 1) k_pna_post_fwd(const PsArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave & 3, wn = wave >> 2;                       // wave tile: rows 64*wr.., columns 64*wn..
  const long long r0 = (long long)blockIdx.x * PS_ROWS;
  const int rows_here = (int)(a.R - r0 < PS_ROWS ? a.R - r0 : PS_ROWS);
  const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;

  // ---- DMA lane geometry: one instruction = 8 rows x 128 bytes (whole lines); lane -> (row l >> 3, chunk position l & 7).
  // LDS rows are 128 bytes, chunk position = chunk ^ ((row >> 1) & 7): the 16 rows of a ds_read_b128 lane group fall on
  // 16 different 16-byte slots of the 256-byte bank row.  row = 8*j + (l >> 3): (row >> 1) & 7 = (l >> 4) ^ 4*(j & 1).
  const int drow = lane >> 3;
  const unsigned dpos = (unsigned)((lane & 7) ^ (lane >> 4));
  // T stages per column tile (x == NULL: no x Wx^T stages — the input-gradient use); with nct > 1 the stages of all the
  // column tiles form ONE pipeline (global stage index tt, buffer tt & 1): the row tile's operand pieces are re-fetched
  // per column tile (L2), the weights change, and a column tile's epilogue restages through the buffer its last stage
  // just freed while the next tile's first stage is already landing in the other one
  const int nct = MULTI ? a.nct : 1;        // (a full unroll over 4 column tiles: 43 spilled registers instead of 20)
  const int KT = a.K / PS_BK, T = KT + (MULTI ? 0 : PS_F / PS_BK), TT = T * nct;

  auto issue = [&](int tt) {
    const unsigned sb = lds0 + (unsigned)((tt % PS_NSTAGE) * PS_STAGE);
    const int ct = MULTI ? (T == 2 ? tt >> 1 : tt / T) : 0, t = tt - ct * T;
    const char* wcat = a.wcat + (MULTI ? (long long)ct * a.w_ct_bytes : 0ll);
    const bool tail = t >= KT;                                   // x Wx^T stages
    const int u = tail ? t - KT : t;
    const char* xb = tail ? a.x + (r0 * a.ld_x + (long long)PS_BK * u) * 2 : a.agg + (r0 * a.ld_agg + (long long)PS_BK * u) * 2;
    const unsigned ldx = (unsigned)((tail ? a.ld_x : a.ld_agg) * 2);
    int drow_o = drow;
    unsigned dpos_o = dpos;
    asm volatile("" : "+v"(drow_o), "+v"(dpos_o));     // opaque: per-lane byte offsets are recomputed, not hoisted as live registers
    PS_M0_SET(sb + 1024u * (unsigned)(4 * wave + 2));
#define PS_X_PIECE(I)                                                                                 \
    {                                                  /* X: 32 instructions, wave w takes j = 4w .. 4w+3 */ \
      const int j = 4 * wave + (I);                                                                   \
      int rr = 8 * j + drow_o;                                                                        \
      rr = rr < rows_here ? rr : rows_here - 1;        /* clamped: rows past R are never stored */     \
      ps_dma<1024 * ((I) - 2)>((unsigned)rr * ldx + 16u * (dpos_o ^ (unsigned)(4 * (j & 1))), xb);   \
    }
    PS_X_PIECE(0) PS_X_PIECE(1) PS_X_PIECE(2) PS_X_PIECE(3)
#undef PS_X_PIECE
    const unsigned ldw = tail ? 2u * PS_F : 6u * (unsigned)a.K;
    PS_M0_SET(sb + (unsigned)PS_XB + 1024u * (unsigned)(6 * wave + 3));
#define PS_W_PIECE(Q)                                                                                 \
    {                                                  /* W: 3 sets x 16 instructions, wave w takes i = 6w .. 6w+5 */ \
      const int i = 6 * wave + (Q), s = i >> 4, j = i & 15;                                           \
      /* W_s[n, k] of the agg stages sits at wcat[n, ((k >> 7) * 3 + s) * 128 + (k & 127)] (virtual-chunk order of    \
         tg_pna_fold_fwd); the tail stages read Wx for every set (only set 0 is multiplied) */                          \
      const char* wb = tail ? a.wx + 2ll * PS_BK * u                                                  \
                            : wcat + (((long long)(u >> 1) * 3 + s) * 128 + (long long)PS_BK * (u & 1)) * 2;           \
      ps_dma<1024 * ((Q) - 3)>((unsigned)(8 * j + drow_o) * ldw + 16u * (dpos_o ^ (unsigned)(4 * (j & 1))), wb);      \
    }
    PS_W_PIECE(0) PS_W_PIECE(1) PS_W_PIECE(2) PS_W_PIECE(3) PS_W_PIECE(4) PS_W_PIECE(5)
#undef PS_W_PIECE
  };

  ps_f32x16 acc[3][2][2];

  // fragment lane offsets inside a piece: row r = lane & 31, 16-byte chunk 2*ks + (lane >> 5), swizzled
  const int fr = lane & 31, fh = lane >> 5;
  int fo[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) fo[ks] = PS_ROWB * fr + 16 * ((2 * ks + fh) ^ ((fr >> 1) & 7));

  // one stage: its DMA was issued a stage ago — wait for it, one barrier (every wave is also done with the other
  // buffer), issue the next stage into that buffer, then the MFMAs.  NSET = 3 on the agg stages, 1 on the x Wx^T stages.
#define PS_STAGE_BODY(NSET)                                                                           \
  {                                                                                                   \
    /* this wave's pieces of the stage have landed (vmcnt) AND its fragment reads of the other buffer have returned   \
       (lgkmcnt): the barrier hands that buffer to the DMA issued right behind it */                                    \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                       \
    if (!(PS_ABL & 4)) __builtin_amdgcn_s_barrier();                                                  \
    asm volatile("" ::: "memory");                                                                    \
    if (!(PS_ABL & 2) && tt + 1 < TT) issue(tt + 1);                                                  \
    const char* xs = smem + (tt % PS_NSTAGE) * PS_STAGE + (wr * 64) * PS_ROWB;                        \
    const char* ws = smem + (tt % PS_NSTAGE) * PS_STAGE + PS_XB + (wn * 64) * PS_ROWB;                \
    /* rolling fragment prefetch, one (k-step, set) group ahead: the reads of group g+1 are issued before the four   \
       MFMAs of group g (32 fragment registers: two A pairs, two B pairs) */                                          \
    ps_v8bf bA[2][2], aA[2][2];                                                                       \
    bA[0][0] = ps_frag(xs + fo[0]); bA[0][1] = ps_frag(xs + 32 * PS_ROWB + fo[0]);                    \
    aA[0][0] = ps_frag(ws + fo[0]); aA[0][1] = ps_frag(ws + 32 * PS_ROWB + fo[0]);                    \
    _Pragma("unroll") for (int g = 0; g < 4 * (NSET); ++g) {                                          \
      const int ks = g / (NSET), sg = g % (NSET);                                                     \
      if (g + 1 < 4 * (NSET)) {                                                                       \
        const int ks2 = (g + 1) / (NSET), s2 = (g + 1) % (NSET);                                      \
        aA[(g + 1) & 1][0] = ps_frag(ws + s2 * PS_WB + fo[ks2]);                                      \
        aA[(g + 1) & 1][1] = ps_frag(ws + s2 * PS_WB + 32 * PS_ROWB + fo[ks2]);                       \
        if (s2 == 0) { bA[ks2 & 1][0] = ps_frag(xs + fo[ks2]); bA[ks2 & 1][1] = ps_frag(xs + 32 * PS_ROWB + fo[ks2]); } \
      }                                                                                               \
      __builtin_amdgcn_sched_barrier(0);                                                              \
      if (PS_ABL & 1) { acc[sg][0][0][0] += (float)aA[g & 1][0][0] + (float)aA[g & 1][1][0] + (float)bA[ks & 1][0][0] + (float)bA[ks & 1][1][0]; continue; } \
      acc[sg][0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aA[g & 1][0], bA[ks & 1][0], acc[sg][0][0], 0, 0, 0); \
      acc[sg][0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aA[g & 1][0], bA[ks & 1][1], acc[sg][0][1], 0, 0, 0); \
      acc[sg][1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aA[g & 1][1], bA[ks & 1][0], acc[sg][1][0], 0, 0, 0); \
      acc[sg][1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aA[g & 1][1], bA[ks & 1][1], acc[sg][1][1], 0, 0, 0); \
      __builtin_amdgcn_sched_barrier(0);                                                              \
    }                                                                                                 \
  }
  issue(0);
  int tt = 0;
#pragma unroll 1
  for (int ct = 0; ct < nct; ++ct) {
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[s][p][b][i] = 0.f;
    for (int t = 0; t < KT; ++t, ++tt) PS_STAGE_BODY(3)
    if constexpr (!MULTI) { for (int t = KT; t < T; ++t, ++tt) PS_STAGE_BODY(1) }
    __syncthreads();                                             // every wave is done reading the buffer of stage tt - 1
    char* og = smem + ((tt - 1) % PS_NSTAGE) * PS_STAGE;         // ... which takes the restaged output tile (64 of its 80 KiB)
    unsigned short* outp = a.out + (MULTI ? (long long)ct * a.out_ct_cols : 0ll);

    // ---- epilogue.  C/D map of a 32x32 tile: column (-> row r) = lane & 31, row (-> feature n) = (reg & 3) +
    // 8*(reg >> 2) + 4*(lane >> 5): 4 consecutive n per register group.  amp / att are per-lane scalars.
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int rl = wr * 64 + b * 32 + fr;
      const long long rg = r0 + (rl < rows_here ? rl : rows_here - 1);
      const float2 sc = *reinterpret_cast<const float2*>(a.scales + 2 * rg);
#pragma unroll
      for (int p = 0; p < 2; ++p) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int nl = wn * 64 + p * 32 + 8 * g + 4 * fh;
          const float4 bv = MULTI ? make_float4(0.f, 0.f, 0.f, 0.f) : *reinterpret_cast<const float4*>(a.bias + nl);
          const float bb[4] = {bv.x, bv.y, bv.z, bv.w};
          float v[4];
#pragma unroll
          for (int j = 0; j < 4; ++j)
            v[j] = acc[0][p][b][4 * g + j] + bb[j] + sc.x * acc[1][p][b][4 * g + j] + sc.y * acc[2][p][b][4 * g + j];
          uint2 pk;
          pk.x = (unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16);
          pk.y = (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16);
          *reinterpret_cast<uint2*>(og + ps_out_off(rl, nl >> 3) + 2 * (nl & 7)) = pk;
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 8; ++p) {                                // whole-row stores: piece -> (row, 16-byte chunk)
      const int piece = tid + PS_THREADS * p, row = piece >> 4, ch = piece & 15;
      if (row < rows_here)
        *reinterpret_cast<uint4*>(outp + (r0 + row) * a.ld_out + ch * 8) =
            *reinterpret_cast<const uint4*>(og + ps_out_off(row, ch));
    }
    // (the next stage's barrier comes before the DMA that refills this buffer: every wave has read its pieces by then)
  }
#undef PS_STAGE_BODY
}

}  // namespace tg

using namespace tg;

// out [R,128] (bf16) = bias + x Wx^T + agg W_0^T + amp * (agg W_1^T) + att * (agg W_2^T)
// wcat [128, 3K] bf16 in the virtual-chunk order tg_pna_fold_fwd writes (128-column block 3c+s = W_s[:, 128c:128c+128]),
// wx [128,128] bf16 row-major, scales fp32 [>= R][2] = (amp, att).
extern "C" int tg_pna_post_fwd_bf16(const void* agg, const void* x, const void* wcat, const void* wx, const float* bias,
                                    const float* scales, void* out, int64_t R, int32_t K, int64_t ld_agg, int64_t ld_x,
                                    int64_t ld_out, void* stream) {
  TG_CHECK(R > 0 && K > 0 && K % 128 == 0, "tg_pna_post_fwd_bf16: need K %% 128 == 0 (R=%lld K=%d)", (long long)R, K);
  TG_CHECK(agg && x && wcat && wx && bias && scales && out, "tg_pna_post_fwd_bf16: %s", "null operand");
  TG_CHECK(ld_agg % 8 == 0 && ld_x % 8 == 0 && ld_out % 8 == 0 && ld_agg >= K && ld_x >= PS_F && ld_out >= PS_F,
           "tg_pna_post_fwd_bf16: %s", "bad row strides");
  TG_CHECK(((reinterpret_cast<uintptr_t>(agg) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(wcat) |
             reinterpret_cast<uintptr_t>(wx) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(bias)) & 15) == 0 &&
               (reinterpret_cast<uintptr_t>(scales) & 7) == 0,
           "tg_pna_post_fwd_bf16: %s", "operands must be 16-byte aligned");
  // the DMA's per-lane byte offsets are 32-bit: one row tile must span < 4 GiB of either operand
  TG_CHECK((long long)PS_ROWS * ld_agg * 2 < (1ll << 31) && 128ll * 3 * K * 2 < (1ll << 31),
           "tg_pna_post_fwd_bf16: %s", "row stride too large");
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pna_post_fwd<false>), hipFuncAttributeMaxDynamicSharedMemorySize, PS_LDS);
    attr = true;
  }
  const long long tiles = (R + PS_ROWS - 1) / PS_ROWS;
  TG_CHECK(tiles <= 2147483647LL, "tg_pna_post_fwd_bf16: %s", "too many tiles");
  PsArgs a;
  a.agg = (const char*)agg; a.x = (const char*)x; a.wcat = (const char*)wcat; a.wx = (const char*)wx;
  a.bias = bias; a.scales = scales; a.out = (unsigned short*)out;
  a.R = R; a.ld_agg = ld_agg; a.ld_x = ld_x; a.ld_out = ld_out; a.K = K;
  a.nct = 1; a.w_ct_bytes = 0; a.out_ct_cols = 0;
  hipLaunchKernelGGL(k_pna_post_fwd<false>, dim3((unsigned)tiles), dim3(PS_THREADS), PS_LDS, (hipStream_t)stream, a);
  TG_LAUNCH_CHECK();
  return 0;
}

// The input gradient of the same projection w.r.t. the aggregates, on the same kernel:
//     dagg[r, :] = g[r] W_0 + amp(r) * (g[r] W_1) + att(r) * (g[r] W_2)          (dagg [R, K], g [R, 128])
// Column tile j of dagg is a post projection with agg := g (K' = 128), W_s' := W_s[:, 128 j ..]^T and no x / bias terms;
// wt_cat [K, 384] = [W_0^T | W_1^T | W_2^T] (what tg_pna_fold_fwd already writes for tg_gemm_nt_scaled_bf16) read as
// K / 128 blocks of [128, 384] IS the kernel's virtual-chunk weight layout for K' = 128.  One workgroup walks the K / 128
// column tiles of its row tile as ONE stage pipeline (separate launches per column tile have two stages each and ran
// prologue-bound: no faster than the GEMM they replace).  Replaces tg_gemm_nt_scaled_bf16's VALU-rescaled operand copies (MFMA busy 0.24).
extern "C" int tg_pna_post_dagg_bf16(const void* g, const void* wt_cat, const float* scales, void* dagg, int64_t R,
                                     int32_t K, int64_t ld_g, int64_t ld_dagg, void* stream) {
  TG_CHECK(R > 0 && K > 0 && K % 128 == 0, "tg_pna_post_dagg_bf16: need K %% 128 == 0 (R=%lld K=%d)", (long long)R, K);
  TG_CHECK(g && wt_cat && scales && dagg, "tg_pna_post_dagg_bf16: %s", "null operand");
  TG_CHECK(ld_g % 8 == 0 && ld_dagg % 8 == 0 && ld_g >= PS_F && ld_dagg >= K, "tg_pna_post_dagg_bf16: %s", "bad row strides");
  TG_CHECK(((reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(wt_cat) | reinterpret_cast<uintptr_t>(dagg)) & 15) == 0 &&
               (reinterpret_cast<uintptr_t>(scales) & 7) == 0,
           "tg_pna_post_dagg_bf16: %s", "operands must be 16-byte aligned");
  TG_CHECK((long long)PS_ROWS * ld_g * 2 < (1ll << 31), "tg_pna_post_dagg_bf16: %s", "row stride too large");
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pna_post_fwd<true>), hipFuncAttributeMaxDynamicSharedMemorySize, PS_LDS);
    attr = true;
  }
  const long long tiles = (R + PS_ROWS - 1) / PS_ROWS;
  TG_CHECK(tiles <= 2147483647LL, "tg_pna_post_dagg_bf16: %s", "too many tiles");
  PsArgs a;
  a.agg = (const char*)g; a.x = nullptr; a.wcat = (const char*)wt_cat; a.wx = nullptr;
  a.bias = nullptr; a.scales = scales; a.out = (unsigned short*)dagg;
  a.R = R; a.ld_agg = ld_g; a.ld_x = 0; a.ld_out = ld_dagg; a.K = PS_F;
  a.nct = K / 128; a.w_ct_bytes = 128ll * 384 * 2; a.out_ct_cols = 128;
  hipLaunchKernelGGL(k_pna_post_fwd<true>, dim3((unsigned)tiles), dim3(PS_THREADS), PS_LDS, (hipStream_t)stream, a);
  TG_LAUNCH_CHECK();
  return 0;
}
