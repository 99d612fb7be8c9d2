// Weight-gradient GEMM of every Linear on the path:  dW[M,N] = G[R,M]^T X[R,N]  with R = table rows / edges /
// nodes (10^5..10^7) and M,N = 128..1536.  "Tall-skinny TN": both operands are contracted over their ROW index,
// the output is tiny, so the op is HBM-bound (R*(M+N)*2 bytes read once) and needs split-K over the rows —
// which a stock GEMM does not do for this shape (hipBLASLt ran it on a handful of workgroups).
//
// Structure (bf16 in, fp32 out):
//   * grid = (M/128) x (N/128) output tiles x row slabs (split-K, ~3 workgroups per CU),
//   * each 256-thread workgroup streams its slab in 64-row steps: 16-byte global loads -> registers ->
//     XOR-swizzled row-major LDS image (double buffered), so HBM reads stay whole 256-byte rows,
//   * both MFMA operands need 8 consecutive k (= rows) of ONE column per lane: read with
//     ds_read_b64_tr_b16 (hardware transpose) from the row-major image — no transposed copy of G or X ever exists,
//   * v_mfma_f32_32x32x16_bf16, 2x2 accumulator tiles per wave (64x64 per wave, 128x128 per workgroup),
//   * fp32 slab partials are summed by a second kernel in slab order: deterministic, no float atomics.
#include <stdlib.h>
#include "common.hpp"
#include "../../include/tabgnn_hip.h"

namespace tg {

typedef short v4s __attribute__((ext_vector_type(4)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int GM = 128, GN = 128, GK = 64;       // workgroup tile and rows per step
constexpr int TILE_BYTES = GK * 256;              // one operand tile: 64 rows x 128 bf16

// byte offset of 16-byte chunk ch (0..15) of row `row` (conflict-free for row writes and transposed reads)
__device__ __forceinline__ int lds_off(int row, int ch) {
  return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3)));
}

// 8 consecutive k (rows r0..r0+7 of the tile) of column (cb*32 + lane&31): MFMA 32x32x16 A/B fragment
__device__ __forceinline__ uint4 tn_scale8(uint4 v, float s) {   // 8 packed bf16 times s, round to nearest even
  typedef float f32x8 __attribute__((ext_vector_type(8)));
  // vector conversions keep (element 2j, element 2j+1) together: one v_cvt_pk_bf16_f32 per output word, no re-pairing
  f32x8 f = __builtin_convertvector(__builtin_bit_cast(v8bf, v), f32x8);
  f *= s;
  return __builtin_bit_cast(uint4, __builtin_convertvector(f, v8bf));
}

__device__ __forceinline__ v8bf frag_tr(const char* tile, int ks, int cb, int lane) {
  const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
  const int r0 = ks * 16 + (g >> 1) * 8;
  const int c0 = cb * 4 + (g & 1) * 2;
  typedef v4s __attribute__((address_space(3))) * lds_v4s_ptr;
  const int a0 = lds_off(r0 + q, c0 + (p >> 1)) + 8 * (p & 1);
  const int a1 = lds_off(r0 + 4 + q, c0 + (p >> 1)) + 8 * (p & 1);
  v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_ptr)(tile + a0));
  v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_ptr)(tile + a1));
  typedef short v8s __attribute__((ext_vector_type(8)));
  v8s r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(v8bf, r);
}

// SCALED (the PNA post projection's weight gradient with the degree scalers folded in): G is [R, mreal] and M = 3*mreal
// is virtual — output row block s = m0 / mreal holds (f_s(r) * G)^T X with f = (1, amp(r), att(r)), scales = fp32
// [R,2] (amp, att); the G pieces of blocks 1 and 2 are multiplied in registers on their way into LDS (re-rounded to
// bf16, exactly what the materialised [g | amp*g | att*g] operand of the unfused path held).
template <bool SCALED>
__global__ void __launch_bounds__(256, 2) k_gemm_tn_bf16(const unsigned short* __restrict__ G,
                                                       const unsigned short* __restrict__ X,
                                                       float* __restrict__ partial, float* __restrict__ colsum_part,
                                                       long long R, int M, int N, long long ldg, long long ldx,
                                                       long long rows_per_slab, int tm, int tn, int nslab,
                                                       const float* __restrict__ scales, int mreal) {
  __shared__ __attribute__((aligned(16))) char lds[2 * 2 * TILE_BYTES];   // [buf][G|X][64 x 256 B] = 64 KiB
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // XCD-aware order: workgroup b runs on XCD b % 8 (round-robin dispatch) and every XCD has its own L2.  The tm*tn
  // output tiles of one row slab take consecutive slots of the SAME XCD, so they stream the slab together and the
  // G / X column blocks they share (each G block feeds tn tiles, each X block tm tiles) are fetched from HBM once
  // and hit that L2 afterwards.  With one output tile (M = N = 128) the map is the identity.
  const int T_ = tm * tn;
  const int slot = blockIdx.x >> 3;
  const int slab = (slot / T_) * 8 + (blockIdx.x & 7), tile = slot % T_;
  if (slab >= nslab) return;
  const int bx = tile % tm, by = tile / tm;
  const int m0 = bx * GM, n0 = by * GN;
  const int gset = SCALED ? m0 / mreal : 0;           // which scaler this output block carries (uniform)
  const int gcol0 = SCALED ? m0 % mreal : m0;         // first real G column of the block
  const long long r_begin = (long long)slab * rows_per_slab;
  long long r_end = r_begin + rows_per_slab;
  if (r_end > R) r_end = R;
  const int wm = wave >> 1, wn = wave & 1;

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

  // staging map: piece = tid + 256*p -> row = piece >> 4 (0..63), chunk = piece & 15
  // TWO staging register sets (A: rg/rx/sg, B: rh/ry/sh): the tile two steps ahead is already in flight while the tile
  // one step ahead waits for its LDS slot, so every HBM read has two steps of MFMA work to land (one set left each
  // read ~0.5 us of cover against a > 1 us loaded latency: the slab loop ran at 4.0 TB/s)
  uint4 rg0, rg1, rg2, rg3, rx0, rx1, rx2, rx3;
  uint4 rh0, rh1, rh2, rh3, ry0, ry1, ry2, ry3;
  float sg0 = 1.f, sg1 = 1.f, sg2 = 1.f, sg3 = 1.f;
  float sh0 = 1.f, sh1 = 1.f, sh2 = 1.f, sh3 = 1.f;
#define TG_LOAD_PIECE(P, RG, RX, SG)                                                                          \
  {                                                                                                           \
    int piece = tid + 256 * P, row = piece >> 4, ch = piece & 15;                                             \
    long long r = r0_ + row;                                                                                  \
    /* unconditional loads from clamped (always valid) addresses, zeroed by select: no branch, no early wait */ \
    bool okr = r < r_end;                                                                                     \
    long long rc_ = okr ? r : r_end - 1;                                                                      \
    int mc_ = SCALED ? gcol0 + ch * 8 : (m0 + ch * 8 < M ? m0 + ch * 8 : M - 8);                              \
    int nc_ = n0 + ch * 8 < N ? n0 + ch * 8 : N - 8;                                                          \
    uint4 vg_ = *reinterpret_cast<const uint4*>(G + rc_ * ldg + mc_);                                         \
    uint4 vx_ = *reinterpret_cast<const uint4*>(X + rc_ * ldx + nc_);                                         \
    if constexpr (SCALED) {                                                                                   \
      if (gset > 0) SG = scales[2 * rc_ + gset - 1];      /* applied at store time: no wait on the loads here */ \
    }                                                                                                         \
    bool okg_ = okr && m0 + ch * 8 < M, okx_ = okr && n0 + ch * 8 < N;                                        \
    RG = make_uint4(okg_ ? vg_.x : 0u, okg_ ? vg_.y : 0u, okg_ ? vg_.z : 0u, okg_ ? vg_.w : 0u);              \
    RX = make_uint4(okx_ ? vx_.x : 0u, okx_ ? vx_.y : 0u, okx_ ? vx_.z : 0u, okx_ ? vx_.w : 0u);              \
  }
#define load_tile(R0)                                                                                         \
  {                                                                                                           \
    long long r0_ = (R0);                                                                                     \
    TG_LOAD_PIECE(0, rg0, rx0, sg0) TG_LOAD_PIECE(1, rg1, rx1, sg1) TG_LOAD_PIECE(2, rg2, rx2, sg2) TG_LOAD_PIECE(3, rg3, rx3, sg3) \
  }
#define load_tile_b(R0)                                                                                       \
  {                                                                                                           \
    long long r0_ = (R0);                                                                                     \
    TG_LOAD_PIECE(0, rh0, ry0, sh0) TG_LOAD_PIECE(1, rh1, ry1, sh1) TG_LOAD_PIECE(2, rh2, ry2, sh2) TG_LOAD_PIECE(3, rh3, ry3, sh3) \
  }
#define TG_STORE_PIECE(P, RG, RX, SG)                                                                         \
  {                                                                                                           \
    int piece = tid + 256 * P, off = lds_off(piece >> 4, piece & 15);                                         \
    if constexpr (SCALED) {                                                                                   \
      if (gset > 0) RG = tn_scale8(RG, SG);                                                                   \
    }                                                                                                         \
    *reinterpret_cast<uint4*>(tg_w + off) = RG;                                                               \
    *reinterpret_cast<uint4*>(tg_w + TILE_BYTES + off) = RX;                                                  \
  }
#define store_tile(BUF)                                                                                       \
  {                                                                                                           \
    char* tg_w = lds + (BUF) * 2 * TILE_BYTES;                                                                \
    TG_STORE_PIECE(0, rg0, rx0, sg0) TG_STORE_PIECE(1, rg1, rx1, sg1) TG_STORE_PIECE(2, rg2, rx2, sg2) TG_STORE_PIECE(3, rg3, rx3, sg3) \
  }
#define store_tile_b(BUF)                                                                                     \
  {                                                                                                           \
    char* tg_w = lds + (BUF) * 2 * TILE_BYTES;                                                                \
    TG_STORE_PIECE(0, rh0, ry0, sh0) TG_STORE_PIECE(1, rh1, ry1, sh1) TG_STORE_PIECE(2, rh2, ry2, sh2) TG_STORE_PIECE(3, rh3, ry3, sh3) \
  }

  // bias gradient: thread t sums column (t & 127) over rows (t >> 7)*32 .. +31 of every G tile
  const bool do_cs = colsum_part != nullptr && by == 0;
  float cs = 0.f;
  const int cs_col = tid & 127, cs_r0 = (tid >> 7) * 32;

  // software pipeline, two tiles deep: at the top of step s the LDS buffer `buf` holds tile s, set A holds tile s+1
  // (loads issued one step ago) and set B's loads of tile s+2 have just been issued; after the MFMAs tile s+1 goes to
  // the other LDS buffer and the sets swap roles (the loop is unrolled by two so that each set keeps its names).
#define TG_COMPUTE_TILE()                                                                                     \
  {                                                                                                           \
    const char* tg_ = lds + buf * 2 * TILE_BYTES;                                                             \
    const char* tx_ = tg_ + TILE_BYTES;                                                                       \
    _Pragma("unroll") for (int ks = 0; ks < GK / 16; ++ks) {                                                  \
      v8bf a0 = frag_tr(tg_, ks, wm * 2 + 0, lane), a1 = frag_tr(tg_, ks, wm * 2 + 1, lane);                 \
      v8bf b0 = frag_tr(tx_, ks, wn * 2 + 0, lane), b1 = frag_tr(tx_, ks, wn * 2 + 1, lane);                 \
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);                        \
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);                        \
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);                        \
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);                        \
    }                                                                                                         \
    if (do_cs) {                                                                                              \
      _Pragma("unroll 8") for (int rr = 0; rr < 32; ++rr) {                                                   \
        int row = cs_r0 + rr;                                                                                 \
        unsigned short hv = *reinterpret_cast<const unsigned short*>(tg_ + lds_off(row, cs_col >> 3) + 2 * (cs_col & 7)); \
        cs += bf2f(hv);                                                                                       \
      }                                                                                                       \
    }                                                                                                         \
  }
  if (r_begin < r_end) {
    load_tile(r_begin)
    store_tile(0)
    if (r_begin + GK < r_end) load_tile(r_begin + GK)              // tile 1 -> set A
  }
  __syncthreads();
  int buf = 0;
  for (long long r0 = r_begin; r0 < r_end; r0 += 2 * GK) {
    // ---- step s (even): set A holds tile s+1
    if (r0 + 2 * GK < r_end) load_tile_b(r0 + 2 * GK)              // tile s+2 -> set B
    TG_COMPUTE_TILE()
    if (r0 + GK < r_end) store_tile(buf ^ 1)
    __syncthreads();
    buf ^= 1;
    if (r0 + GK >= r_end) break;
    // ---- step s+1 (odd): set B holds tile s+2
    if (r0 + 3 * GK < r_end) load_tile(r0 + 3 * GK)                // tile s+3 -> set A
    TG_COMPUTE_TILE()
    if (r0 + 2 * GK < r_end) store_tile_b(buf ^ 1)
    __syncthreads();
    buf ^= 1;
  }

  if (do_cs) {   // the two row halves meet in LDS (all tiles consumed: last loop iteration ended with a barrier)
    float* red = reinterpret_cast<float*>(lds);
    if (tid >= 128) red[cs_col] = cs;
    __syncthreads();
    if (tid < 128 && m0 + cs_col < M) colsum_part[(long long)slab * M + m0 + cs_col] = cs + red[cs_col];
  }

  // C/D map of the 32x32 tile: col = lane & 31, row = (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5)
  float* out = partial + (long long)slab * M * N;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      int n = n0 + wn * 64 + b * 32 + (lane & 31);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        int m = m0 + wm * 64 + a * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
        if (m < M && n < N) out[(long long)m * N + n] = acc[a][b][i];
      }
    }
}

#undef TG_LOAD_PIECE
#undef TG_STORE_PIECE
#undef TG_COMPUTE_TILE
#undef load_tile
#undef load_tile_b
#undef store_tile
#undef store_tile_b

// ---------------------------------------------------------------------------------------------------------------
// WIDE variant: one 512-thread workgroup (8 waves, the only one resident on its CU: 128 KiB of LDS) owns a 384 x 128
// output block = THREE 128-column tiles of the "wide" operand against ONE tile of the "narrow" operand, so the narrow
// tile is staged through LDS once per three output tiles instead of once per tile, and every wave feeds 6 MFMAs from 5
// fragment reads (2 x 2 per 4 reads in the kernel above): the LDS traffic per MFMA drops by 40 %, which is what bounds
// the multi-tile problems (W_in: 384 x 128, the PNA message / edge-update first layers: 128 x 384, the fuse MLP, and
// the scaled post projection, where the three wide tiles are the SAME 128 columns of G times (1, amp, att) — G is
// then read from HBM once and written to LDS three times).
//   WIDE_G = true : wide operand = G (M side), narrow = X;  false: wide = X (N side), narrow = G  (operand order of
//   the MFMA is swapped so that the output's N index stays on the lanes: coalesced partial writes either way).
// Software pipeline: one staging register set; the loads of tile s+2 are issued right after tile s+1 went to LDS and
// have the barrier plus the whole MFMA phase of step s+1 to land.
// The bias gradient (column sums of G) is accumulated from the staging registers: thread t always stages chunk t & 15
// of its rows, so it keeps 8 running sums per G tile it touches and the 32 threads of a chunk meet in LDS at the end.
// GATHER (WIDE_G = false, N = 384): the wide operand X is never materialised — tile t (128 columns) of row r is row
// idx[t][r] (or r when idx[t] is NULL) of src[t]: the weight gradient of tg_gemm_nt_gather3_bf16.  The slab's indices
// (at most TN_GATHER_ROWS rows per slab) are staged in LDS behind the operand tiles once.
struct TnGather {
  const unsigned short* src[3];
  const int* idx[3];
  long long stride[3];      // row pitch of src[t] in elements
};
constexpr int TN_GATHER_ROWS = 2048;

template <bool SCALED, bool WIDE_G, bool GATHER = false>
__global__ void __launch_bounds__(512) k_gemm_tn_wide(const unsigned short* __restrict__ G,
                                                      const unsigned short* __restrict__ X,
                                                      float* __restrict__ partial, float* __restrict__ colsum_part,
                                                      long long R, int M, int N, long long ldg, long long ldx,
                                                      long long rows_per_slab, int tw, int tnar, int nslab,
                                                      const float* __restrict__ scales, int mreal, TnGather gth) {
  extern __shared__ __attribute__((aligned(16))) char lds[];     // [buf][W0|W1|W2|Nn][64 x 256 B] = 128 KiB (+ indices)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int T_ = tw * tnar;
  const int slot = blockIdx.x >> 3;
  const int slab = (slot / T_) * 8 + (blockIdx.x & 7), tile = slot % T_;
  if (slab >= nslab) return;
  const int bw = tile % tw, bn = tile / tw;            // wide block (384 columns, or the scaled 128) and narrow block
  // column origins of the two operand streams
  const unsigned short* __restrict__ Wp = WIDE_G ? G : X;
  const unsigned short* __restrict__ Np = WIDE_G ? X : G;
  const long long ldw = WIDE_G ? ldg : ldx, ldn = WIDE_G ? ldx : ldg;
  const int w0 = SCALED ? bw * 128 : bw * 384, n0 = bn * 128;
  const long long r_begin = (long long)slab * rows_per_slab;
  long long r_end = r_begin + rows_per_slab;
  if (r_end > R) r_end = R;
  const int wm = wave >> 1, wn = wave & 1;
  int* gidx = reinterpret_cast<int*>(lds + 2 * 4 * TILE_BYTES);      // [3][TN_GATHER_ROWS]
  if constexpr (GATHER) {
    const int nrows = (int)(r_end - r_begin);
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const int* ip = gth.idx[t];
      for (int i = tid; i < nrows; i += 512) gidx[t * TN_GATHER_ROWS + i] = ip ? ip[r_begin + i] : (int)(r_begin + i);
    }
    __syncthreads();
  }

  f32x16 acc[3][2];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

  constexpr int NWP = SCALED ? 2 : 6;                   // wide pieces per thread per step
  constexpr int NSET = SCALED ? 3 : 2;                  // staging register sets = steps a global load has to land
  uint4 rw[NSET][NWP], rn[NSET][2];
  float sc[NSET][2][2];
  const int prow = tid >> 4, pch = tid & 15;            // piece q: row prow + 32 q, 16-byte chunk pch (fixed per thread)
  // Addressing: a UNIFORM row base per step (scalar registers) plus a per-lane 32-bit byte offset (row-in-tile,
  // clamped to the slab's last row, times the row pitch + column): two vector instructions per piece row and step.
  // The load sequence is the SAME straight-line code for every tile — full, partial or past the end — so that the
  // s_waitcnt before each LDS store counts exactly the loads of the other sets as outstanding (any `if` around or
  // between the loads makes the compiler wait for all of them: the prefetch distance collapses to one step).  Rows
  // past the end are zeroed at STORE time, under a uniform branch that holds no memory instruction and is taken only
  // for a slab's last, partial tile.  [64-bit per-lane addresses, clamps and zeroing selects on every piece made the
  // load/store phase ~300 VALU instructions per step and wave: more SIMD time than the 24 MFMAs it feeds.]
  const unsigned coln = (unsigned)(n0 + pch * 8) * 2u, colw = (unsigned)(w0 + pch * 8) * 2u;
  const unsigned pitchn = (unsigned)ldn * 2u, pitchw = (unsigned)ldw * 2u;
  const int soff0 = lds_off(prow, pch), soff1 = lds_off(prow + 32, pch);
#define TG_WIDE_LOAD(K_, R0)                                                                                  \
  {                                                                                                           \
    const long long r0_ = (R0);                                                                               \
    long long left_ = r_end - r0_;                              /* uniform; <= 0 past the end */              \
    const long long rb_ = left_ >= 1 ? r0_ : r_end - 1;         /* first row read: always inside the slab */  \
    const int last_ = left_ >= GK ? GK - 1 : (left_ >= 1 ? (int)left_ - 1 : 0);                               \
    const char* nb_ = reinterpret_cast<const char*>(Np) + rb_ * ldn * 2;                                      \
    const char* wb_ = reinterpret_cast<const char*>(Wp) + rb_ * ldw * 2;                                      \
    const char* sb_ = reinterpret_cast<const char*>(scales) + rb_ * 8;                                        \
    _Pragma("unroll") for (int q = 0; q < 2; ++q) {                                                           \
      const unsigned rr_ = (unsigned)min(prow + 32 * q, last_);                                               \
      rn[K_][q] = *reinterpret_cast<const uint4*>(nb_ + (rr_ * pitchn + coln));                               \
      if constexpr (SCALED) {                                                                                 \
        rw[K_][q] = *reinterpret_cast<const uint4*>(wb_ + (rr_ * pitchw + colw));                             \
        const float2 s2 = *reinterpret_cast<const float2*>(sb_ + rr_ * 8u);                                   \
        sc[K_][q][0] = s2.x; sc[K_][q][1] = s2.y;                                                             \
      } else if constexpr (GATHER) {                                                                          \
        const int li_ = (int)(rb_ - r_begin) + (int)rr_;                                                      \
        _Pragma("unroll") for (int t = 0; t < 3; ++t)                                                         \
          rw[K_][t * 2 + q] = *reinterpret_cast<const uint4*>(gth.src[t] + (long long)gidx[t * TN_GATHER_ROWS + li_] * gth.stride[t] + pch * 8); \
      } else {                                                                                                \
        _Pragma("unroll") for (int t = 0; t < 3; ++t)                                                         \
          rw[K_][t * 2 + q] = *reinterpret_cast<const uint4*>(wb_ + (rr_ * pitchw + colw) + t * 256);         \
      }                                                                                                       \
    }                                                                                                         \
  }
  // LEFT = rows of the tile in the set that exist (uniform); only a partial tile pays for the zeroing
#define TG_WIDE_MASK(K_, LEFT)                                                                                \
  if ((LEFT) < GK) {                                                                                          \
    _Pragma("unroll") for (int q = 0; q < 2; ++q) {                                                           \
      if (prow + 32 * q >= (LEFT)) {                                                                          \
        rn[K_][q] = make_uint4(0u, 0u, 0u, 0u);                                                               \
        _Pragma("unroll") for (int t = 0; t < (SCALED ? 1 : 3); ++t) rw[K_][SCALED ? q : t * 2 + q] = make_uint4(0u, 0u, 0u, 0u); \
      }                                                                                                       \
    }                                                                                                         \
  }
#define TG_WIDE_STORE(K_, BUF)                                                                                \
  {                                                                                                           \
    char* base = lds + (BUF) * 4 * TILE_BYTES;                                                                \
    _Pragma("unroll") for (int q = 0; q < 2; ++q) {                                                           \
      const int off = q ? soff1 : soff0;                                                                      \
      *reinterpret_cast<uint4*>(base + 3 * TILE_BYTES + off) = rn[K_][q];                                     \
      if constexpr (SCALED) {                                                                                 \
        *reinterpret_cast<uint4*>(base + off) = rw[K_][q];                                                    \
        *reinterpret_cast<uint4*>(base + TILE_BYTES + off) = tn_scale8(rw[K_][q], sc[K_][q][0]);             \
        *reinterpret_cast<uint4*>(base + 2 * TILE_BYTES + off) = tn_scale8(rw[K_][q], sc[K_][q][1]);         \
      } else {                                                                                                \
        _Pragma("unroll") for (int t = 0; t < 3; ++t)                                                         \
          *reinterpret_cast<uint4*>(base + t * TILE_BYTES + off) = rw[K_][t * 2 + q];                         \
      }                                                                                                       \
    }                                                                                                         \
  }

  // bias gradient = column sums of G (never with SCALED): from the staging registers, 8 columns per G tile per thread
  const bool do_cs = !SCALED && colsum_part != nullptr && (WIDE_G ? bn == 0 : bw == 0);
  constexpr int NCS = WIDE_G ? 3 : 1;
  float cs[NCS][8];
#pragma unroll
  for (int t = 0; t < NCS; ++t)
#pragma unroll
    for (int j = 0; j < 8; ++j) cs[t][j] = 0.f;
#define TG_WIDE_CS(K_)                                                                                        \
  if constexpr (!SCALED) {                                                                                    \
    if (do_cs) {                                                                                              \
      _Pragma("unroll") for (int t = 0; t < NCS; ++t)                                                         \
        _Pragma("unroll") for (int q = 0; q < 2; ++q) {                                                       \
          const uint4 v = WIDE_G ? rw[K_][t * 2 + q] : rn[K_][q];                                             \
          const unsigned w[4] = {v.x, v.y, v.z, v.w};                                                         \
          _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                     \
            cs[t][2 * j] += __uint_as_float(w[j] << 16);                                                      \
            cs[t][2 * j + 1] += __uint_as_float(w[j] & 0xffff0000u);                                          \
          }                                                                                                   \
        }                                                                                                     \
    }                                                                                                         \
  }

  // fragment addresses: everything but the LDS buffer and the k-step (an immediate: 16 rows = 4096 bytes, and the
  // swizzle term does not depend on it) is a per-lane constant — ten registers computed once instead of ~15 vector
  // adds per k-step
  int fwo[3][2], fno[2][2];
  {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3;
#pragma unroll
    for (int hi = 0; hi < 2; ++hi) {
      const int row = (g >> 1) * 8 + 4 * hi + q;
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const int blk = wm * 3 + a;
        fwo[a][hi] = (blk >> 2) * TILE_BYTES + lds_off(row, (blk & 3) * 4 + (g & 1) * 2 + (pp >> 1)) + 8 * (pp & 1);
      }
#pragma unroll
      for (int b2 = 0; b2 < 2; ++b2)
        fno[b2][hi] = 3 * TILE_BYTES + lds_off(row, (wn * 2 + b2) * 4 + (g & 1) * 2 + (pp >> 1)) + 8 * (pp & 1);
    }
  }
  typedef v4s __attribute__((address_space(3))) * lds_v4s_ptr;
  typedef short v8s_t __attribute__((ext_vector_type(8)));
  auto frag_at = [&](const char* base, int off_lo, int off_hi, int ks) -> v8bf {
    v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_ptr)(base + off_lo + ks * 4096));
    v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_ptr)(base + off_hi + ks * 4096));
    v8s_t r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(v8bf, r);
  };
  auto compute_tile = [&](int buf) {
    const char* base = lds + buf * 4 * TILE_BYTES;
#pragma unroll
    for (int ks = 0; ks < GK / 16; ++ks) {
      v8bf fw[3], fn[2];
#pragma unroll
      for (int a = 0; a < 3; ++a) fw[a] = frag_at(base, fwo[a][0], fwo[a][1], ks);
#pragma unroll
      for (int b2 = 0; b2 < 2; ++b2) fn[b2] = frag_at(base, fno[b2][0], fno[b2][1], ks);
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
          acc[a][b] = WIDE_G ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[a], fn[b], acc[a][b], 0, 0, 0)
                             : __builtin_amdgcn_mfma_f32_32x32x16_bf16(fn[b], fw[a], acc[a][b], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);   // fragments of one k-step live at a time (registers: no spills)
    }
  };

  // pipeline: at the top of step s LDS buffer s & 1 holds tile s and the register sets hold tiles s+1 .. s+NSET (tile
  // j in set j % NSET); after the MFMAs tile s+1 goes to the other LDS buffer and its set takes tile s+1+NSET
  // Every load below is UNCONDITIONAL (rows past the slab read the clamped last row and are zeroed by `ok`): with the
  // loads of later tiles under an `if`, the compiler's s_waitcnt placement no longer knows how many are outstanding
  // and waits for (nearly) all of them before each LDS store — the prefetch distance collapses to one step.
  TG_WIDE_LOAD(0, r_begin)
  TG_WIDE_MASK(0, r_end - r_begin)
  TG_WIDE_CS(0)
  TG_WIDE_STORE(0, 0)
#pragma unroll
  for (int j = 1; j <= NSET; ++j) TG_WIDE_LOAD(j % NSET, r_begin + (long long)j * GK)
  __syncthreads();
  int buf = 0;
  for (long long r0 = r_begin; r0 < r_end; r0 += (long long)NSET * GK) {
#pragma unroll
    for (int u = 0; u < NSET; ++u) {
      const long long rs = r0 + (long long)u * GK;                // first row of tile s
      if (rs >= r_end) break;
      compute_tile(buf);
      TG_WIDE_MASK((u + 1) % NSET, r_end - (rs + GK))             // tile s+1: partial or (all zeros) past the end
      TG_WIDE_CS((u + 1) % NSET)
      TG_WIDE_STORE((u + 1) % NSET, buf ^ 1)
      TG_WIDE_LOAD((u + 1) % NSET, rs + (long long)(1 + NSET) * GK)
      __syncthreads();
      buf ^= 1;
    }
  }
#undef TG_WIDE_LOAD
#undef TG_WIDE_STORE
#undef TG_WIDE_MASK
#undef TG_WIDE_CS

  if (do_cs) {   // all tiles consumed (the loop ended with a barrier): the 32 row groups of a chunk meet in LDS
    float* red = reinterpret_cast<float*>(lds);                 // [32][NCS*128]
#pragma unroll
    for (int t = 0; t < NCS; ++t)
#pragma unroll
      for (int j = 0; j < 8; ++j) red[prow * (NCS * 128) + t * 128 + pch * 8 + j] = cs[t][j];
    __syncthreads();
    if (tid < NCS * 128) {
      float s = 0.f;
#pragma unroll 8
      for (int k = 0; k < 32; ++k) s += red[k * (NCS * 128) + tid];
      colsum_part[(long long)slab * M + (WIDE_G ? w0 : n0) + tid] = s;
    }
  }

  // C/D map of the 32x32 tile: col = lane & 31 (the B operand's column), row = (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5)
  float* out = partial + (long long)slab * M * N;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const int blk = wm * 3 + a;
    const int wcol = SCALED ? (blk >> 2) * mreal + w0 + (blk & 3) * 32 : w0 + blk * 32;   // first wide index of the tile
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int ncol = n0 + wn * 64 + b * 32;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int rr = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5), cc = lane & 31;
        if constexpr (WIDE_G) out[(long long)(wcol + rr) * N + ncol + cc] = acc[a][b][i];      // rows = wide = m
        else out[(long long)(ncol + rr) * N + wcol + cc] = acc[a][b][i];                         // rows = narrow = m
      }
    }
  }
}

// out[i] = sum_s partial[s][i]: 256 threads = 16 float4 columns x 16 slab strips, four loads in flight per
// thread, strips combined through LDS in strip order (deterministic)
// Two jobs in one launch (the weight slabs and the bias column sums of the same GEMM): blocks [0, nb1) reduce job 1,
// the rest job 2.
__global__ void __launch_bounds__(256) k_sum_slabs(const float* __restrict__ partial1, int nslab, long long mn1,
                                                    float* __restrict__ out1, int nb1, const float* __restrict__ partial2,
                                                    long long mn2, float* __restrict__ out2, int accumulate) {
  __shared__ float4 red[16][16];
  const int col = threadIdx.x & 15, strip = threadIdx.x >> 4;
  const bool second = (int)blockIdx.x >= nb1;
  const float* __restrict__ partial = second ? partial2 : partial1;
  float* __restrict__ out = second ? out2 : out1;
  const long long mn = second ? mn2 : mn1;
  const long long i = ((long long)(second ? blockIdx.x - nb1 : blockIdx.x) * 16 + col) * 4;
  float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < mn) {
    const int per = (nslab + 15) / 16, s0 = strip * per, s1 = s0 + per < nslab ? s0 + per : nslab;
    if (i + 4 <= mn) {
      float4 u0 = t, u1 = t, u2 = t, u3 = t;
      int sl = s0;
      for (; sl + 3 < s1; sl += 4) {
        float4 v0 = *reinterpret_cast<const float4*>(partial + (long long)sl * mn + i);
        float4 v1 = *reinterpret_cast<const float4*>(partial + (long long)(sl + 1) * mn + i);
        float4 v2 = *reinterpret_cast<const float4*>(partial + (long long)(sl + 2) * mn + i);
        float4 v3 = *reinterpret_cast<const float4*>(partial + (long long)(sl + 3) * mn + i);
        u0.x += v0.x; u0.y += v0.y; u0.z += v0.z; u0.w += v0.w;
        u1.x += v1.x; u1.y += v1.y; u1.z += v1.z; u1.w += v1.w;
        u2.x += v2.x; u2.y += v2.y; u2.z += v2.z; u2.w += v2.w;
        u3.x += v3.x; u3.y += v3.y; u3.z += v3.z; u3.w += v3.w;
      }
      for (; sl < s1; ++sl) {
        float4 v0 = *reinterpret_cast<const float4*>(partial + (long long)sl * mn + i);
        u0.x += v0.x; u0.y += v0.y; u0.z += v0.z; u0.w += v0.w;
      }
      t.x = (u0.x + u1.x) + (u2.x + u3.x); t.y = (u0.y + u1.y) + (u2.y + u3.y);
      t.z = (u0.z + u1.z) + (u2.z + u3.z); t.w = (u0.w + u1.w) + (u2.w + u3.w);
    } else {
      float e[4] = {0.f, 0.f, 0.f, 0.f};
      for (int sl = s0; sl < s1; ++sl)
        for (int k = 0; k < 4; ++k)
          if (i + k < mn) e[k] += partial[(long long)sl * mn + i + k];
      t = make_float4(e[0], e[1], e[2], e[3]);
    }
  }
  red[strip][col] = t;
  __syncthreads();
  if (strip == 0 && i < mn) {
    float4 r = red[0][col];
    for (int k = 1; k < 16; ++k) { float4 v = red[k][col]; r.x += v.x; r.y += v.y; r.z += v.z; r.w += v.w; }
    if (i + 4 <= mn) {
      if (accumulate) {   // gradient accumulation straight into the caller's (flat) gradient buffer
        float4 o = *reinterpret_cast<const float4*>(out + i);
        r.x += o.x; r.y += o.y; r.z += o.z; r.w += o.w;
      }
      *reinterpret_cast<float4*>(out + i) = r;
    } else {
      float e[4] = {r.x, r.y, r.z, r.w};
      for (int k = 0; k < 4; ++k) if (i + k < mn) out[i + k] = accumulate ? out[i + k] + e[k] : e[k];
    }
  }
}

}  // namespace tg

using namespace tg;

// Which kernel and split a weight-gradient problem gets.  kind: 0 = 128x128 tiles (k_gemm_tn_bf16), 1 = wide on the G
// (M) side, 2 = wide on the X (N) side (k_gemm_tn_wide: 384x128 blocks, one workgroup per CU).
struct TnPlan {
  int kind, tm, tn, nslab;
  long long rows_per_slab;
};
static TnPlan tn_plan(long long R, int M, int N, bool scaled = false, int mreal = 0, bool gather = false) {
  TnPlan p;
  const long long steps = ceil_div(R, GK);
  p.kind = 0;
  if (scaled) {
    if (N % 128 == 0 && mreal % 128 == 0) p.kind = 1;
  } else if (M % 384 == 0 && N % 128 == 0) {
    p.kind = 1;
  } else if (N % 384 == 0 && M % 128 == 0) {
    p.kind = 2;
  }
  if (p.kind) {
    // tm = wide blocks, tn = narrow blocks; ONE workgroup per CU is resident, so ~one round of 256 equal slabs
    p.tm = scaled ? mreal / 128 : (p.kind == 1 ? M / 384 : N / 384);
    p.tn = p.kind == 1 ? N / 128 : M / 128;
    const long long tiles = (long long)p.tm * p.tn;
    long long want = tiles >= 256 ? 1 : 256 / tiles;
    if (gather) {                                          // the slab's row indices must fit their LDS table
      const long long need = ceil_div(steps, (long long)(TN_GATHER_ROWS / GK));
      if (want < need) want = need;
      if (want > steps) want = steps;
    }
    if (steps < 2 * want && !gather) p.kind = 0;           // too few rows to pipeline: the small-tile kernel
    else {
      p.nslab = (int)want;
      p.rows_per_slab = ceil_div(steps, p.nslab) * (long long)GK;
      p.nslab = (int)ceil_div(R, p.rows_per_slab);
      return p;
    }
  }
  p.tm = ceil_div(M, GM);
  p.tn = ceil_div(N, GN);
  // 64 KiB of LDS per workgroup: two are resident per CU, so for a single output tile 512 workgroups fill the chip in
  // ONE round (768 ran as a full round plus a half-empty one: 337 -> 291 us on the 2.58 M-row 128 x 128 problem);
  // several output tiles per slab measured faster with the finer split.  A slab is never shorter than 4 steps (256
  // rows): below that the fp32 partials (64 KiB per tile and slab, written and read back) outweigh the rows it read.
  long long want = ceil_div(p.tm * p.tn == 1 ? 512 : 768, (long long)p.tm * p.tn);
  static const int min_steps = getenv("TABGNN_TN_MINSTEPS") ? atoi(getenv("TABGNN_TN_MINSTEPS")) : 4;   // tuning aid
  const long long cap = steps / min_steps > 0 ? steps / min_steps : 1;
  if (want > cap) want = cap;
  p.nslab = (int)(want < 1 ? 1 : want);
  p.rows_per_slab = ceil_div(steps, p.nslab) * (long long)GK;
  p.nslab = (int)ceil_div(R, p.rows_per_slab);
  return p;
}

static void tn_wide_attr() {
  static bool done = false;
  if (done) return;
  const int lds = 2 * 4 * TILE_BYTES;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_tn_wide<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_tn_wide<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_tn_wide<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_tn_wide<false, false, true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, lds + 3 * TN_GATHER_ROWS * (int)sizeof(int));
  done = true;
}

extern "C" int64_t tg_gemm_tn_workspace_floats(int64_t R, int32_t M, int32_t N) {
  // the scaled entry point asks with M = 3*mreal: its plan may have fewer slabs than this one, never more
  const TnPlan a = tn_plan(R, M, N), b = tn_plan(R, M, N, M % 384 == 0, M / 3);
  const int nslab = a.nslab > b.nslab ? a.nslab : b.nslab;
  return (int64_t)nslab * M * N + (int64_t)nslab * M;
}

// out[M,N] (fp32) = G[R,M]^T X[R,N];  G, X bf16 with row strides ldg, ldx (elements, multiples of 8)
// colsum (optional, fp32 [M]) = column sums of G = the bias gradient of the same Linear
extern "C" int tg_gemm_tn_bf16(const void* G, const void* X, float* out, float* colsum, float* workspace, int64_t R,
                               int32_t M, int32_t N, int64_t ldg, int64_t ldx, int32_t accumulate, void* stream) {
  TG_CHECK(R > 0 && M > 0 && N > 0, "tg_gemm_tn_bf16: empty problem");
  TG_CHECK(M % 8 == 0 && N % 8 == 0 && ldg % 8 == 0 && ldx % 8 == 0,
           "tg_gemm_tn_bf16: M, N and row strides must be multiples of 8 (M=%d N=%d)", M, N);
  TG_CHECK((reinterpret_cast<uintptr_t>(G) & 15) == 0 && (reinterpret_cast<uintptr_t>(X) & 15) == 0 &&
               (reinterpret_cast<uintptr_t>(out) & 15) == 0,
           "tg_gemm_tn_bf16: operands and output must be 16-byte aligned");
  const TnPlan pl = tn_plan(R, M, N);
  const int tm = pl.tm, tn = pl.tn, nslab = pl.nslab;
  const long long rps = pl.rows_per_slab;
  hipStream_t st = (hipStream_t)stream;
  const long long blocks = (long long)((nslab + 7) / 8) * 8 * tm * tn;
  TG_CHECK(blocks <= 2147483647LL, "tg_gemm_tn_bf16: too many tiles");
  float* cs_part = colsum ? workspace + (long long)nslab * M * N : nullptr;
  if (pl.kind == 1) {
    tn_wide_attr();
    hipLaunchKernelGGL((k_gemm_tn_wide<false, true>), dim3((unsigned)blocks), dim3(512), 2 * 4 * TILE_BYTES, st,
                       (const unsigned short*)G, (const unsigned short*)X, workspace, cs_part, (long long)R, M, N,
                       (long long)ldg, (long long)ldx, rps, tm, tn, nslab, (const float*)nullptr, 0, TnGather{});
  } else if (pl.kind == 2) {
    tn_wide_attr();
    hipLaunchKernelGGL((k_gemm_tn_wide<false, false>), dim3((unsigned)blocks), dim3(512), 2 * 4 * TILE_BYTES, st,
                       (const unsigned short*)G, (const unsigned short*)X, workspace, cs_part, (long long)R, M, N,
                       (long long)ldg, (long long)ldx, rps, tm, tn, nslab, (const float*)nullptr, 0, TnGather{});
  } else {
    hipLaunchKernelGGL(k_gemm_tn_bf16<false>, dim3((unsigned)blocks), dim3(256), 0, st, (const unsigned short*)G,
                       (const unsigned short*)X, workspace, cs_part, (long long)R, M, N, (long long)ldg, (long long)ldx,
                       rps, tm, tn, nslab, (const float*)nullptr, 0);
  }
  long long mn = (long long)M * N;
  const int nb1 = ceil_div(ceil_div(mn, 4), 16), nb2 = colsum ? ceil_div(ceil_div(M, 4), 16) : 0;
  hipLaunchKernelGGL(k_sum_slabs, dim3(nb1 + nb2), dim3(256), 0, st, workspace, nslab, mn, out, nb1,
                     workspace + (long long)nslab * mn, (long long)M, colsum, accumulate);
  TG_LAUNCH_CHECK();
  return 0;
}

// out[M,384] (fp32) (+)= G[R,M]^T [S0[i0[r]] | S1[i1[r]] | S2[i2[r]]]: tg_gemm_tn_bf16 with the X operand gathered on the
// fly (the weight gradient of tg_gemm_nt_gather3_bf16).  workspace: tg_gemm_tn_gather3_workspace_floats(R, M) floats.
extern "C" int64_t tg_gemm_tn_gather3_workspace_floats(int64_t R, int32_t M) {
  const TnPlan p = tn_plan(R, M, 384, false, 0, true);
  return (int64_t)p.nslab * M * 384 + (int64_t)p.nslab * M;
}

extern "C" int tg_gemm_tn_gather3_bf16(const void* G, const tg_gather3* gs, float* out, float* colsum, float* workspace,
                                       int64_t R, int32_t M, int64_t ldg, int32_t accumulate, void* stream) {
  const int N = 384;
  TG_CHECK(G && gs && out && workspace && R > 0 && M > 0 && M % 128 == 0 && ldg % 8 == 0 && ldg >= M,
           "tg_gemm_tn_gather3_bf16: bad shape (R=%lld M=%d)", (long long)R, M);
  TG_CHECK(R <= 2147483647LL, "tg_gemm_tn_gather3_bf16: row indices are 32-bit");
  TG_CHECK(((reinterpret_cast<uintptr_t>(G) | reinterpret_cast<uintptr_t>(out)) & 15) == 0,
           "tg_gemm_tn_gather3_bf16: operands must be 16-byte aligned");
  TnGather g;
  for (int c = 0; c < 3; ++c) {
    TG_CHECK(gs->src[c] && gs->stride[c] >= 128 && gs->stride[c] % 8 == 0 &&
                 (reinterpret_cast<uintptr_t>(gs->src[c]) & 15) == 0,
             "tg_gemm_tn_gather3_bf16: source %d must be a 16-byte aligned [*, >=128] bf16 matrix", c);
    g.src[c] = (const unsigned short*)gs->src[c];
    g.idx[c] = gs->idx[c];
    g.stride[c] = gs->stride[c];
  }
  const TnPlan pl = tn_plan(R, M, N, false, 0, true);
  TG_CHECK(pl.kind == 2 && pl.rows_per_slab <= TN_GATHER_ROWS, "tg_gemm_tn_gather3_bf16: no plan for R=%lld M=%d",
           (long long)R, M);
  const int tm = pl.tm, tn = pl.tn, nslab = pl.nslab;
  hipStream_t st = (hipStream_t)stream;
  const long long blocks = (long long)((nslab + 7) / 8) * 8 * tm * tn;
  TG_CHECK(blocks <= 2147483647LL, "tg_gemm_tn_gather3_bf16: too many tiles");
  float* cs_part = colsum ? workspace + (long long)nslab * M * N : nullptr;
  tn_wide_attr();
  hipLaunchKernelGGL((k_gemm_tn_wide<false, false, true>), dim3((unsigned)blocks), dim3(512),
                     2 * 4 * TILE_BYTES + 3 * TN_GATHER_ROWS * sizeof(int), st, (const unsigned short*)G,
                     (const unsigned short*)nullptr, workspace, cs_part, (long long)R, M, N, (long long)ldg, 0LL,
                     pl.rows_per_slab, tm, tn, nslab, (const float*)nullptr, 0, g);
  long long mn = (long long)M * N;
  const int nb1 = ceil_div(ceil_div(mn, 4), 16), nb2 = colsum ? ceil_div(ceil_div(M, 4), 16) : 0;
  hipLaunchKernelGGL(k_sum_slabs, dim3(nb1 + nb2), dim3(256), 0, st, workspace, nslab, mn, out, nb1,
                     workspace + (long long)nslab * mn, (long long)M, colsum, accumulate);
  TG_LAUNCH_CHECK();
  return 0;
}

// out[3*mreal, N] (fp32) = [G | amp*G | att*G]^T X with G [R,mreal], X [R,N] bf16 and scales fp32 [R,2] = (amp, att):
// the weight gradient of the PNA post projection with the degree scalers folded in ([dG0 | dG1 | dG2] never exists).
// workspace: tg_gemm_tn_workspace_floats(R, 3*mreal, N) floats.
extern "C" int tg_gemm_tn_scaled_bf16(const void* G, const void* X, const float* scales, float* out, float* workspace,
                                      int64_t R, int32_t mreal, int32_t N, int64_t ldg, int64_t ldx, int32_t accumulate,
                                      void* stream) {
  TG_CHECK(R > 0 && mreal > 0 && N > 0 && mreal % GM == 0 && N % 8 == 0 && ldg % 8 == 0 && ldx % 8 == 0 && ldg >= mreal,
           "tg_gemm_tn_scaled_bf16: need mreal %% 128 == 0, N %% 8 == 0 (mreal=%d N=%d)", mreal, N);
  TG_CHECK(G && X && scales && out && workspace, "tg_gemm_tn_scaled_bf16: null operand");
  TG_CHECK((reinterpret_cast<uintptr_t>(G) & 15) == 0 && (reinterpret_cast<uintptr_t>(X) & 15) == 0 &&
               (reinterpret_cast<uintptr_t>(out) & 15) == 0,
           "tg_gemm_tn_scaled_bf16: operands and output must be 16-byte aligned");
  const int M = 3 * mreal;
  const TnPlan pl = tn_plan(R, M, N, true, mreal);
  const int tm = pl.tm, tn = pl.tn, nslab = pl.nslab;
  const long long rps = pl.rows_per_slab;
  hipStream_t st = (hipStream_t)stream;
  const long long blocks = (long long)((nslab + 7) / 8) * 8 * tm * tn;
  TG_CHECK(blocks <= 2147483647LL, "tg_gemm_tn_scaled_bf16: too many tiles");
  if (pl.kind == 1) {
    tn_wide_attr();
    hipLaunchKernelGGL((k_gemm_tn_wide<true, true>), dim3((unsigned)blocks), dim3(512), 2 * 4 * TILE_BYTES, st,
                       (const unsigned short*)G, (const unsigned short*)X, workspace, (float*)nullptr, (long long)R, M, N,
                       (long long)ldg, (long long)ldx, rps, tm, tn, nslab, scales, mreal, TnGather{});
  } else {
    hipLaunchKernelGGL(k_gemm_tn_bf16<true>, dim3((unsigned)blocks), dim3(256), 0, st, (const unsigned short*)G,
                       (const unsigned short*)X, workspace, (float*)nullptr, (long long)R, M, N, (long long)ldg,
                       (long long)ldx, rps, tm, tn, nslab, scales, mreal);
  }
  long long mn = (long long)M * N;
  const int nb1 = ceil_div(ceil_div(mn, 4), 16);
  hipLaunchKernelGGL(k_sum_slabs, dim3(nb1), dim3(256), 0, st, workspace, nslab, mn, out, nb1,
                     workspace + (long long)nslab * mn, (long long)M, (float*)nullptr, accumulate);
  TG_LAUNCH_CHECK();
  return 0;
}
