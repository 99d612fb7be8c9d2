// Forward / input-gradient GEMM of the Linears on the path, with the elementwise tail fused:
//     Y[R,N] (bf16) = epilogue( X[R,K] W[N,K]^T )      R = table rows x columns / edges / nodes (10^5..10^7),
//                                                       K, N in {128, 384, 512, 768}: "tall-skinny NT".
// The op is HBM-bound (R*(K+N)*2 bytes; ~1 FLOP/byte-pair at K = 128), so the design goal is a clean row stream:
//   * one 256-thread workgroup per 128-row x 128-column output tile; 64 KiB LDS -> two workgroups per CU overlap
//     one's loads with the other's MFMAs and stores (no in-wave prefetch across the stores: loads and stores share
//     vmcnt on gfx9, so a prefetched load would be waited for together with every store issued after it),
//   * X rows and W rows are both contiguous along k, which is exactly what the MFMA 32x32x16 fragments want
//     (8 consecutive k of one row per lane): 16-byte global loads -> XOR-swizzled row-major LDS image ->
//     plain 16-byte ds_read fragments, no transposes anywhere,
//   * A := W tile (M dim = output feature n), B := X tile (N dim = row r), so every lane ends up with 4 consecutive
//     n of ONE row r per accumulator group: bias / ReLU / dropout / residual are applied in registers, packed to
//     4 x bf16 and staged through LDS (the dead X image) so that HBM sees whole 256-byte row stores,
//   * K > 128 loops over 128-wide k chunks with the next chunk's global loads in flight under the MFMAs
//     (no stores in between), W chunks come from L2.
// Epilogue (flags): + bias[n] (fp32) | ReLU | dropout(p; counter RNG on the element index r*N+n, the same stream the
// stand-alone act_dropout kernels use, so their backward applies unchanged) | gate: x inv_keep where gate[r,n] > 0 else
// 0 (the backward of drop(relu(.)) from the saved forward OUTPUT: > 0 <=> active and kept) | += existing Y.
#include "common.hpp"
#include "../../include/tabgnn_hip.h"

namespace tg {

typedef __bf16 nt_v8bf __attribute__((ext_vector_type(8)));
typedef float nt_f32x16 __attribute__((ext_vector_type(16)));

constexpr int NT_BM = 128, NT_BN = 128, NT_BK = 128;
constexpr int NT_TILE_BYTES = 128 * 256;          // 128 rows x 128 bf16

__device__ __forceinline__ int nt_off(int row, int ch) {      // byte offset of 16-byte chunk ch of tile row `row`
  return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3)));
}

__device__ __forceinline__ nt_v8bf nt_frag(const char* tile, int row, int ch) {
  uint4 v = *reinterpret_cast<const uint4*>(tile + nt_off(row, ch));
  return __builtin_bit_cast(nt_v8bf, v);
}

enum { NT_RELU = 1, NT_DROPOUT = 2, NT_ACCUM = 4, NT_GATE = 8, NT_LEAKY = 16 };

__device__ __forceinline__ uint4 nt_scale8(uint4 v, float s) {   // 8 packed bf16 times s, round to nearest even
  typedef float nt_f32x8 __attribute__((ext_vector_type(8)));
  // vector conversions keep (element 2j, element 2j+1) together: one v_cvt_pk_bf16_f32 per output word, no re-pairing
  nt_f32x8 f = __builtin_convertvector(__builtin_bit_cast(nt_v8bf, v), nt_f32x8);
  f *= s;
  return __builtin_bit_cast(uint4, __builtin_convertvector(f, nt_v8bf));
}

// WIDE (K == 128, N > 128: the QKV projection, edge_emb's input gradient): ONE workgroup per row tile walks the
// N/128 column tiles with the X image staged once; each W tile is prefetched into registers under the previous
// tile's MFMAs and the output tile is restaged through the (dead) W image.  The narrow form gives every column tile
// its own workgroup (XCD-aware order) and loops over k chunks instead.
//
// SCALED (narrow form only; the PNA post projection with its degree scalers folded in, layers.py PNAConv):
//     Y[r,:] (+)= sum_s f_s(r) * X[r, 0:kreal] W[:, s*kreal:(s+1)*kreal]^T,   s = 0..2,
//     f = (1, amp(r), att(r));  K is the VIRTUAL depth 3*kreal and W is laid out by virtual chunk: column block 3c+s
// holds W_s[:, 128c:128c+128].  The X pieces of the amp / att sets are
// multiplied by the row's scale in registers on their way into LDS (re-rounded to bf16, as the materialised
// [amp*agg] / [amp*g] operands of the unfused path were).  Rescaling the accumulators instead (one scale per MFMA B
// column) was tried first: 64 accumulators through the VALU cost 133 spilled VGPRs.
// scales = fp32 (amp, att) per row, padded to a whole number of 128-row tiles (rows >= R are never stored).
// GATHER (narrow form, K = 384): the X operand is never materialised — k chunk c (128 columns) of row r is row
// idx[c][r] (or r when idx[c] is NULL) of src[c]: [x[ia] | x[ib] | e[ic]] of the PNA message / edge-update projections
// read straight from the node and edge embeddings.  The row tile's 3 x 128 indices are staged in LDS once.
struct NtGather {
  const unsigned short* src[3];
  const int* idx[3];
  long long stride[3];      // row pitch of src[c] in elements
};

template <bool WIDE, bool SCALED, bool GATHER = false>
__global__ void __launch_bounds__(256, 2) k_gemm_nt_bf16(const unsigned short* __restrict__ X,
                                                           const unsigned short* __restrict__ W,
                                                           const float* __restrict__ bias,
                                                           const unsigned short* __restrict__ gate,
                                                           unsigned short* __restrict__ Y, long long R, int N, int K,
                                                           long long ldx, long long ldy,
                                                           int flags, unsigned thresh, float inv_keep,
                                                           unsigned long long seed, unsigned rstream,
                                                           const float* __restrict__ scales, int kreal, NtGather gth) {
  seed = live_seed(seed);
  __shared__ __attribute__((aligned(16))) char lds[2 * NT_TILE_BYTES];   // [X image | W image] = 64 KiB
  __shared__ int gidx[GATHER ? 3 * NT_BM : 1];
  char* xs = lds;
  char* ws = lds + NT_TILE_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ncol = N / NT_BN;
  long long row_tile;
  int n0;
  if constexpr (WIDE) {
    row_tile = blockIdx.x;
    n0 = 0;
  } else {
    // XCD-aware tile mapping: workgroup b runs on XCD b % 8 (round-robin dispatch), each XCD has its own L2.  The
    // N/128 column tiles of one row tile get consecutive slots of the SAME XCD, so they run together there and the
    // row tile's X rows are read from HBM once and from that L2 afterwards.
    const long long slot = blockIdx.x >> 3;
    row_tile = (slot / ncol) * 8 + (blockIdx.x & 7);
    n0 = (int)(slot % ncol) * NT_BN;
  }
  if (row_tile * NT_BM >= R) return;
  const long long r0 = row_tile * NT_BM;
  const int wn = wave >> 1, wr = wave & 1;           // wave tile: 64 n x 64 r
  if constexpr (GATHER) {
    for (int i = tid; i < 3 * NT_BM; i += 256) {
      const int c = i / NT_BM, rr = i % NT_BM;
      const long long r = r0 + rr < R - 1 ? r0 + rr : R - 1;      // clamped: rows past R are never stored
      const int* ip = c == 0 ? gth.idx[0] : (c == 1 ? gth.idx[1] : gth.idx[2]);
      gidx[i] = ip ? ip[r] : (int)r;
    }
    __syncthreads();
  }

  nt_f32x16 acc[2][2];
#define NT_ZERO_ACC()                                                                                 \
  _Pragma("unroll") for (int a = 0; a < 2; ++a) _Pragma("unroll") for (int b = 0; b < 2; ++b)         \
      _Pragma("unroll") for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
  NT_ZERO_ACC()

  // staging map: piece = tid + 256*p (p = 0..7) -> tile row = piece >> 4 (0..127), 16-byte chunk = piece & 15.
  // Named registers, not arrays: an array written in one unrolled loop and read in another lands in scratch.
  uint4 rx0, rx1, rx2, rx3, rx4, rx5, rx6, rx7, rw0, rw1, rw2, rw3, rw4, rw5, rw6, rw7;
  const long long rlast = R - 1;
  const int st_row = tid >> 4, st_ch = tid & 15;              // piece p: row st_row + 16*p, chunk st_ch
#define NT_LOADX1(P, RX, K0)                                                                          \
  {                                                                                                   \
    const int row = st_row + 16 * (P);                                                                \
    if constexpr (GATHER) {                                                                           \
      const int c_ = (K0) >> 7;                                   /* uniform */                       \
      const unsigned short* sp_ = c_ == 0 ? gth.src[0] : (c_ == 1 ? gth.src[1] : gth.src[2]);         \
      const long long ss_ = c_ == 0 ? gth.stride[0] : (c_ == 1 ? gth.stride[1] : gth.stride[2]);      \
      RX = *reinterpret_cast<const uint4*>(sp_ + (long long)gidx[c_ * NT_BM + row] * ss_ + st_ch * 8); \
    } else {                                                                                          \
      const long long r = r0 + row < rlast ? r0 + row : rlast; /* clamped: rows past R are never stored */ \
      RX = *reinterpret_cast<const uint4*>(X + r * ldx + (K0) + st_ch * 8);                           \
    }                                                                                                 \
  }
#define NT_LOADW1(P, RW, N0, K0)                                                                      \
  RW = *reinterpret_cast<const uint4*>(W + (long long)((N0) + st_row + 16 * (P)) * K + (K0) + st_ch * 8);
#define NT_LOADX(K0)                                                                                  \
  NT_LOADX1(0, rx0, K0) NT_LOADX1(1, rx1, K0) NT_LOADX1(2, rx2, K0) NT_LOADX1(3, rx3, K0)             \
  NT_LOADX1(4, rx4, K0) NT_LOADX1(5, rx5, K0) NT_LOADX1(6, rx6, K0) NT_LOADX1(7, rx7, K0)
#define NT_LOADW(N0, K0)                                                                              \
  NT_LOADW1(0, rw0, N0, K0) NT_LOADW1(1, rw1, N0, K0) NT_LOADW1(2, rw2, N0, K0) NT_LOADW1(3, rw3, N0, K0) \
  NT_LOADW1(4, rw4, N0, K0) NT_LOADW1(5, rw5, N0, K0) NT_LOADW1(6, rw6, N0, K0) NT_LOADW1(7, rw7, N0, K0)
#define NT_ST1(BUF, P, RV) *reinterpret_cast<uint4*>((BUF) + nt_off(st_row + 16 * (P), st_ch)) = (RV);
#define NT_STOREX()                                                                                   \
  NT_ST1(xs, 0, rx0) NT_ST1(xs, 1, rx1) NT_ST1(xs, 2, rx2) NT_ST1(xs, 3, rx3)                         \
  NT_ST1(xs, 4, rx4) NT_ST1(xs, 5, rx5) NT_ST1(xs, 6, rx6) NT_ST1(xs, 7, rx7)
#define NT_STOREW()                                                                                   \
  NT_ST1(ws, 0, rw0) NT_ST1(ws, 1, rw1) NT_ST1(ws, 2, rw2) NT_ST1(ws, 3, rw3)                         \
  NT_ST1(ws, 4, rw4) NT_ST1(ws, 5, rw5) NT_ST1(ws, 6, rw6) NT_ST1(ws, 7, rw7)
#define NT_MFMA_CHUNK()                                                                               \
  _Pragma("unroll") for (int ks = 0; ks < NT_BK / 16; ++ks) {                                         \
    const int ch = ks * 2 + (lane >> 5), rr = lane & 31;                                              \
    nt_v8bf a0 = nt_frag(ws, wn * 64 + rr, ch), a1 = nt_frag(ws, wn * 64 + 32 + rr, ch);             \
    nt_v8bf b0 = nt_frag(xs, wr * 64 + rr, ch), b1 = nt_frag(xs, wr * 64 + 32 + rr, ch);             \
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);                  \
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);                  \
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);                  \
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);                  \
  }

  // this lane's 32 bias values (features wn*64 + a*32 + 8g + 4*(lane>>5) + j of column tile N0)
  float bv[2][4][4];
#define NT_LOAD_BIAS(N0)                                                                              \
  _Pragma("unroll") for (int a = 0; a < 2; ++a) _Pragma("unroll") for (int g = 0; g < 4; ++g) {       \
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);                                                       \
    if (!SCALED && bias) t = *reinterpret_cast<const float4*>(bias + (N0) + wn * 64 + a * 32 + 8 * g + 4 * (lane >> 5)); \
    bv[a][g][0] = t.x; bv[a][g][1] = t.y; bv[a][g][2] = t.z; bv[a][g][3] = t.w;                       \
  }

  // epilogue in registers.  C/D map of a 32x32 tile: column (-> row r) = lane & 31,
  // row (-> feature n) = (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5): 4 consecutive n per register group.
  // OB = the dead LDS image that restages the output tile for whole-row stores.
#define NT_EPILOGUE(OB, N0)                                                                           \
  _Pragma("unroll") for (int a = 0; a < 2; ++a) _Pragma("unroll") for (int b = 0; b < 2; ++b) {       \
    const int rl = wr * 64 + b * 32 + (lane & 31);                                                    \
    const long long r = r0 + rl;                                                                      \
    _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                   \
      const int nl = wn * 64 + a * 32 + 8 * g + 4 * (lane >> 5);                                      \
      float v[4];                                                                                     \
      const unsigned long long e0 = (unsigned long long)(r * (long long)N + (N0) + nl);   /* multiple of 4 */ \
      const unsigned key = rng_key(seed, rstream, (unsigned)(e0 >> 32));                              \
      _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                 \
        float u = acc[a][b][4 * g + j] + bv[a][g][j];                                                 \
        if (flags & NT_RELU) u = fmaxf(u, 0.f);                                                       \
        if ((flags & (NT_LEAKY | NT_GATE)) == NT_LEAKY) u = u > 0.f ? u : 0.01f * u;   /* LeakyReLU(0.01) */  \
        if (flags & NT_DROPOUT) u *= drop_scale_key(key, (unsigned)e0 + j, thresh, inv_keep);         \
        v[j] = u;                                                                                     \
      }                                                                                               \
      uint2 pk;                                                                                       \
      pk.x = (unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16);                                     \
      pk.y = (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16);                                     \
      *reinterpret_cast<uint2*>((OB) + nt_off(rl, nl >> 3) + 2 * (nl & 7)) = pk;                      \
    }                                                                                                 \
  }
  // whole-row stores: piece -> (row, 16-byte chunk)
#define NT_WRITE_OUT(OB, N0)                                                                          \
  _Pragma("unroll") for (int p = 0; p < 8; ++p) {                                                     \
    const int piece = tid + 256 * p, row = piece >> 4, ch = piece & 15;                               \
    const long long r = r0 + row;                                                                     \
    if (r < R) {                                                                                      \
      uint4 o = *reinterpret_cast<const uint4*>((OB) + nt_off(row, ch));                              \
      unsigned short* dst = Y + r * ldy + (N0) + ch * 8;                                              \
      if (flags & NT_GATE) { /* backward of drop(relu(.)): x inv_keep where the saved output is > 0 */ \
        const uint4 gt = *reinterpret_cast<const uint4*>(gate + r * ldy + (N0) + ch * 8);             \
        const unsigned gw[4] = {gt.x, gt.y, gt.z, gt.w};                                              \
        unsigned nw[4] = {o.x, o.y, o.z, o.w};                                                        \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                               \
          const float glo = __uint_as_float(gw[j] << 16), ghi = __uint_as_float(gw[j] & 0xffff0000u);       \
          /* NT_LEAKY: a negative saved output = kept on the 0.01 slope; zero = dropped either way */        \
          float lo = glo > 0.f ? inv_keep : ((flags & NT_LEAKY) && glo < 0.f ? 0.01f * inv_keep : 0.f);       \
          float hi = ghi > 0.f ? inv_keep : ((flags & NT_LEAKY) && ghi < 0.f ? 0.01f * inv_keep : 0.f);       \
          lo *= __uint_as_float(nw[j] << 16);                                                             \
          hi *= __uint_as_float(nw[j] & 0xffff0000u);                                                     \
          nw[j] = (unsigned)f2bf(lo) | ((unsigned)f2bf(hi) << 16);                                    \
        }                                                                                             \
        o = make_uint4(nw[0], nw[1], nw[2], nw[3]);                                                   \
      }                                                                                               \
      if (flags & NT_ACCUM) {                                                                         \
        const uint4 old = *reinterpret_cast<const uint4*>(dst);                                       \
        const unsigned ow[4] = {old.x, old.y, old.z, old.w};                                          \
        unsigned nw[4] = {o.x, o.y, o.z, o.w};                                                        \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                               \
          float lo = __uint_as_float(ow[j] << 16) + __uint_as_float(nw[j] << 16);                     \
          float hi = __uint_as_float(ow[j] & 0xffff0000u) + __uint_as_float(nw[j] & 0xffff0000u);     \
          nw[j] = (unsigned)f2bf(lo) | ((unsigned)f2bf(hi) << 16);                                    \
        }                                                                                             \
        o = make_uint4(nw[0], nw[1], nw[2], nw[3]);                                                   \
      }                                                                                               \
      *reinterpret_cast<uint4*>(dst) = o;                                                             \
    }                                                                                                 \
  }

  if constexpr (WIDE) {
    NT_LOADX(0)
    NT_LOADW(0, 0)
    NT_STOREX()
    NT_STOREW()
    __syncthreads();
    for (int ct = 0; ct < ncol; ++ct) {
      const int nc0 = ct * NT_BN;
      // next W tile (the last one once more on the final pass: unconditional, always in bounds) and this tile's
      // bias fly under the MFMAs
      const int nnext = (ct + 1 < ncol ? ct + 1 : ct) * NT_BN;
      NT_LOADW(nnext, 0)
      NT_LOAD_BIAS(nc0)
      NT_ZERO_ACC()
      NT_MFMA_CHUNK()
      __syncthreads();                                // W image consumed -> it restages the output tile
      NT_EPILOGUE(ws, nc0)
      __syncthreads();
      NT_WRITE_OUT(ws, nc0)
      __syncthreads();
      NT_STOREW()
      __syncthreads();
    }
  } else {
    NT_LOAD_BIAS(n0)
    if constexpr (SCALED) {
      // Virtual chunk v = 3c+s pairs X chunk c with W block (c, s); the X pieces of sets 1 / 2 are multiplied by
      // amp(row) / att(row) in registers (fp32, re-rounded to bf16) on their way into LDS.  This thread stages tile
      // rows st_row + 16*p; their scales for the next chunk are fetched with its X pieces.  (Keeping one fetched X
      // chunk for all three sets was tried: the longer live ranges cost 57-102 spilled VGPRs at two workgroups per
      // CU; re-reading it from L2 does not.)
      float sc0 = 1.f, sc1 = 1.f, sc2 = 1.f, sc3 = 1.f, sc4 = 1.f, sc5 = 1.f, sc6 = 1.f, sc7 = 1.f;
      // scales is padded to whole 128-row tiles, so one base pointer + immediates address every row of the tile
      const float* scp = scales + 2 * (r0 + st_row) - 1;
#define NT_LOADS1(P, SC, SET) SC = scp[32 * (P) + (SET)];
#define NT_LOADSCALES(SET)                                                                            \
  if ((SET) > 0) {                                    /* uniform */                                   \
    NT_LOADS1(0, sc0, SET) NT_LOADS1(1, sc1, SET) NT_LOADS1(2, sc2, SET) NT_LOADS1(3, sc3, SET)       \
    NT_LOADS1(4, sc4, SET) NT_LOADS1(5, sc5, SET) NT_LOADS1(6, sc6, SET) NT_LOADS1(7, sc7, SET)       \
  }
#define NT_SCALEX(SET)                                                                                \
  if ((SET) > 0) {                                                                                    \
    rx0 = nt_scale8(rx0, sc0); rx1 = nt_scale8(rx1, sc1); rx2 = nt_scale8(rx2, sc2); rx3 = nt_scale8(rx3, sc3); \
    rx4 = nt_scale8(rx4, sc4); rx5 = nt_scale8(rx5, sc5); rx6 = nt_scale8(rx6, sc6); rx7 = nt_scale8(rx7, sc7); \
  }
      const int nv = K / NT_BK;                       // 3 * kreal / 128 virtual chunks
      NT_LOADX(0)
      NT_LOADW(n0, 0)
      int v = 0;
      for (; v + 1 < nv; ++v) {                       // last chunk peeled: every load unconditional and in bounds
        NT_SCALEX(v % 3)
        NT_STOREX()
        NT_STOREW()
        const int vn = v + 1;
        NT_LOADX((vn / 3) * NT_BK)                    // issued before the barrier: the wait for the other waves'
        NT_LOADW(n0, vn * NT_BK)                      // LDS stores overlaps the next chunk's HBM/L2 latency
        NT_LOADSCALES(vn % 3)
        __syncthreads();
        NT_MFMA_CHUNK()
        __syncthreads();
      }
      NT_SCALEX(v % 3)
      NT_STOREX()
      NT_STOREW()
      __syncthreads();
      NT_MFMA_CHUNK()
      __syncthreads();
    } else {
    // The last k chunk is peeled off the loop so that EVERY load in the loop body is unconditional and in bounds:
    // with `if (more) load(next)` hipcc hoisted 14 of the 16 next-chunk loads out of the guard (speculative reads
    // 256 bytes past each row; a fault when the operand ends on its mapping's last page).
    NT_LOADX(0)
    NT_LOADW(n0, 0)
    int k0 = 0;
    for (; k0 + NT_BK < K; k0 += NT_BK) {
      NT_STOREX()                                     // waits for the chunk's loads; registers free again
      NT_STOREW()
      NT_LOADX(k0 + NT_BK)                            // next k chunk's loads fly under the barrier and this chunk's
      NT_LOADW(n0, k0 + NT_BK)                        // MFMAs
      __syncthreads();
      NT_MFMA_CHUNK()
      __syncthreads();                                // both images consumed
    }
    NT_STOREX()
    NT_STOREW()
    __syncthreads();
    NT_MFMA_CHUNK()
    __syncthreads();
    }
    NT_EPILOGUE(xs, n0)                               // the X image is dead: output tile
    __syncthreads();
    NT_WRITE_OUT(xs, n0)
  }
}

// GEMM + bias + dropout + residual + LayerNorm in one pass (out_proj -> norm1 and linear2 -> norm2 of the column
// transformer, fused.py:83-92):   z = res + drop(X W^T + bias),   out = LN(z) * gamma + beta.
// N = 128 = d_model, so one workgroup's 128 x 128 tile holds whole rows: the accumulators (+bias, x mask) are restaged
// as fp32 through the two dead LDS images (64 KiB, 16-byte chunks XOR-swizzled by row), then 16 lanes per row add the
// residual, reduce mean / variance with xor-shuffles and write z (kept for the backward, which no longer needs the
// GEMM output or the residual), out and the (mean, rstd) pair.  One pass less than GEMM -> y, LN(x, y) -> out in the
// forward, one read less in the LayerNorm backward.
__global__ void __launch_bounds__(256, 2) k_gemm_nt_ln_bf16(const unsigned short* __restrict__ X,
                                                              const unsigned short* __restrict__ W,
                                                              const float* __restrict__ bias,
                                                              const unsigned short* __restrict__ res,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ beta,
                                                              unsigned short* __restrict__ Z, unsigned short* __restrict__ OUT,
                                                              float* __restrict__ stats, long long R, int K,
                                                              long long ldx, float eps, unsigned thresh, float inv_keep,
                                                              unsigned long long seed, unsigned rstream) {
  seed = live_seed(seed);
  __shared__ __attribute__((aligned(16))) char lds[2 * NT_TILE_BYTES];
  char* xs = lds;
  char* ws = lds + NT_TILE_BYTES;
  const int N = NT_BN;
  constexpr bool SCALED = false, GATHER = false;      // (the shared NT_LOAD_BIAS / NT_LOADX1 macros test them)
  const NtGather gth{};
  const int* gidx = nullptr;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long long row_tile = blockIdx.x;
  if (row_tile * NT_BM >= R) return;
  const int n0 = 0;
  const long long r0 = row_tile * NT_BM;
  const int wn = wave >> 1, wr = wave & 1;
  nt_f32x16 acc[2][2];
  NT_ZERO_ACC()
  uint4 rx0, rx1, rx2, rx3, rx4, rx5, rx6, rx7, rw0, rw1, rw2, rw3, rw4, rw5, rw6, rw7;
  const long long rlast = R - 1;
  const int st_row = tid >> 4, st_ch = tid & 15;
  float bv[2][4][4];
  NT_LOAD_BIAS(n0)
  NT_LOADX(0)
  NT_LOADW(n0, 0)
  int k0 = 0;
  for (; k0 + NT_BK < K; k0 += NT_BK) {
    NT_STOREX()
    NT_STOREW()
    NT_LOADX(k0 + NT_BK)
    NT_LOADW(n0, k0 + NT_BK)
    __syncthreads();
    NT_MFMA_CHUNK()
    __syncthreads();
  }
  NT_STOREX()
  NT_STOREW()
  __syncthreads();
  // the residual rows of the epilogue are fetched now (the staging registers are free again): their latency hides
  // under the last chunk's MFMAs instead of sitting in front of every row's reduction
  {
    const unsigned short* rb = res + st_ch * 8;
#define NT_RES1(P, RX) { const long long r = r0 + st_row + 16 * (P); RX = *reinterpret_cast<const uint4*>(rb + (r < rlast ? r : rlast) * N); }
    NT_RES1(0, rx0) NT_RES1(1, rx1) NT_RES1(2, rx2) NT_RES1(3, rx3) NT_RES1(4, rx4) NT_RES1(5, rx5) NT_RES1(6, rx6) NT_RES1(7, rx7)
#undef NT_RES1
  }
  NT_MFMA_CHUNK()
  __syncthreads();
  // fp32 restage: row rl, 16-byte chunk c (4 floats) at 512*rl + 16*(c ^ (rl & 31))
  float* ob = reinterpret_cast<float*>(lds);
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int rl = wr * 64 + b * 32 + (lane & 31);
      const long long r = r0 + rl;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int nl = wn * 64 + a * 32 + 8 * g + 4 * (lane >> 5);
        const unsigned long long e0 = (unsigned long long)(r * (long long)N + nl);
        const unsigned key = rng_key(seed, rstream, (unsigned)(e0 >> 32));
        float4 v;
        float* pv = &v.x;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float u = acc[a][b][4 * g + j] + bv[a][g][j];
          if (thresh) u *= drop_scale_key(key, (unsigned)e0 + j, thresh, inv_keep);
          pv[j] = u;
        }
        *reinterpret_cast<float4*>(ob + 128 * rl + 4 * ((nl >> 2) ^ (rl & 31))) = v;
      }
    }
  __syncthreads();
  // 16 lanes per row, 8 channels per lane (channels 8*ch .. 8*ch+7)
  float gm[8], bt[8];
  {
    const int ch = tid & 15;
    const float4 g0 = *reinterpret_cast<const float4*>(gamma + ch * 8), g1 = *reinterpret_cast<const float4*>(gamma + ch * 8 + 4);
    const float4 b0 = *reinterpret_cast<const float4*>(beta + ch * 8), b1 = *reinterpret_cast<const float4*>(beta + ch * 8 + 4);
    gm[0] = g0.x; gm[1] = g0.y; gm[2] = g0.z; gm[3] = g0.w; gm[4] = g1.x; gm[5] = g1.y; gm[6] = g1.z; gm[7] = g1.w;
    bt[0] = b0.x; bt[1] = b0.y; bt[2] = b0.z; bt[3] = b0.w; bt[4] = b1.x; bt[5] = b1.y; bt[6] = b1.z; bt[7] = b1.w;
  }
#define NT_LN_ROW(P, RV)                                                                              \
  {                                                                                                   \
    const int row = st_row + 16 * (P), ch = st_ch;                                                    \
    const long long r = r0 + row;                                                                     \
    const float4 u0 = *reinterpret_cast<const float4*>(ob + 128 * row + 4 * ((2 * ch) ^ (row & 31))); \
    const float4 u1 = *reinterpret_cast<const float4*>(ob + 128 * row + 4 * ((2 * ch + 1) ^ (row & 31))); \
    const unsigned rw[4] = {RV.x, RV.y, RV.z, RV.w};                                                  \
    float z[8] = {u0.x, u0.y, u0.z, u0.w, u1.x, u1.y, u1.z, u1.w};                                    \
    float s = 0.f;                                                                                    \
    _Pragma("unroll")          \
    for (int j = 0; j < 4; ++j) {                                                                     \
      z[2 * j] += __uint_as_float(rw[j] << 16);                                                       \
      z[2 * j + 1] += __uint_as_float(rw[j] & 0xffff0000u);                                           \
      s += z[2 * j] + z[2 * j + 1];                                                                   \
    }                                                                                                 \
    _Pragma("unroll")          \
    for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);                                        \
    const float mu = s * (1.f / 128.f);                                                               \
    float v = 0.f;                                                                                    \
    _Pragma("unroll")          \
    for (int j = 0; j < 8; ++j) { const float d = z[j] - mu; v += d * d; }                            \
    _Pragma("unroll")          \
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);                                        \
    const float rstd = rsqrtf(v * (1.f / 128.f) + eps);                                               \
    if (r < R) {                                                                                      \
      unsigned zw[4], ow[4];                                                                          \
    _Pragma("unroll")          \
      for (int j = 0; j < 4; ++j) {                                                                   \
        const float y0 = (z[2 * j] - mu) * rstd * gm[2 * j] + bt[2 * j];                              \
        const float y1 = (z[2 * j + 1] - mu) * rstd * gm[2 * j + 1] + bt[2 * j + 1];                  \
        zw[j] = (unsigned)f2bf(z[2 * j]) | ((unsigned)f2bf(z[2 * j + 1]) << 16);                      \
        ow[j] = (unsigned)f2bf(y0) | ((unsigned)f2bf(y1) << 16);                                      \
      }                                                                                               \
      *reinterpret_cast<uint4*>(Z + r * N + ch * 8) = make_uint4(zw[0], zw[1], zw[2], zw[3]);         \
      *reinterpret_cast<uint4*>(OUT + r * N + ch * 8) = make_uint4(ow[0], ow[1], ow[2], ow[3]);       \
      if (ch == 0) { stats[2 * r] = mu; stats[2 * r + 1] = rstd; }                                    \
    }                                                                                                 \
  }
  NT_LN_ROW(0, rx0) NT_LN_ROW(1, rx1) NT_LN_ROW(2, rx2) NT_LN_ROW(3, rx3)
  NT_LN_ROW(4, rx4) NT_LN_ROW(5, rx5) NT_LN_ROW(6, rx6) NT_LN_ROW(7, rx7)
#undef NT_LN_ROW
}

}  // namespace tg

using namespace tg;

// Y[R,N] = epilogue(X[R,K] W[N,K]^T);  X, W, Y bf16, bias fp32 [N] or NULL.  Requires N % 128 == 0, K % 128 == 0,
// ldx/ldy multiples of 8, 16-byte aligned operands (tg_gemm_nt_supported tells).  flags: 1 ReLU, 2 dropout, 4 Y += ,
// 8 gate (gate: bf16 [R, ldy] like Y; p_drop gives the 1/(1-p) factor).
extern "C" int32_t tg_gemm_nt_supported(int64_t R, int32_t N, int32_t K) {
  return R > 0 && N > 0 && K > 0 && N % NT_BN == 0 && K % NT_BK == 0;
}

extern "C" int tg_gemm_nt_bf16(const void* X, const void* W, const float* bias, const void* gate, void* Y, int64_t R,
                               int32_t N, int32_t K, int64_t ldx, int64_t ldy, int32_t flags, float p_drop, uint64_t seed,
                               uint32_t rstream, void* stream) {
  TG_CHECK(tg_gemm_nt_supported(R, N, K), "tg_gemm_nt_bf16: need N %% 128 == 0 and K %% 128 == 0 (R=%lld N=%d K=%d)",
           (long long)R, N, K);
  TG_CHECK(ldx % 8 == 0 && ldy % 8 == 0 && ldx >= K && ldy >= N, "tg_gemm_nt_bf16: bad row strides");
  TG_CHECK((reinterpret_cast<uintptr_t>(X) & 15) == 0 && (reinterpret_cast<uintptr_t>(W) & 15) == 0 &&
               (reinterpret_cast<uintptr_t>(Y) & 15) == 0,
           "tg_gemm_nt_bf16: operands must be 16-byte aligned");
  TG_CHECK(!(flags & NT_DROPOUT) || ldy == N, "tg_gemm_nt_bf16: dropout needs a contiguous output (ldy == N)");
  TG_CHECK(!(flags & NT_GATE) || (gate && (reinterpret_cast<uintptr_t>(gate) & 15) == 0),
           "tg_gemm_nt_bf16: gate flag needs a 16-byte aligned gate tensor (same layout as Y)");
  unsigned thresh = (flags & NT_DROPOUT) && p_drop > 0.f ? drop_threshold(p_drop) : 0u;
  float inv_keep = p_drop < 1.f ? 1.f / (1.f - p_drop) : 0.f;
  if (!thresh) flags &= ~NT_DROPOUT;
  if (!(flags & (NT_DROPOUT | NT_GATE)) || p_drop <= 0.f) inv_keep = 1.f;
  const long long row_tiles = (R + NT_BM - 1) / NT_BM;
  if (K == NT_BK && N > NT_BN) {       // wide outputs of a 128-deep product: one workgroup per row tile
    TG_CHECK(row_tiles <= 2147483647LL, "tg_gemm_nt_bf16: too many tiles (R=%lld)", (long long)R);
    hipLaunchKernelGGL((k_gemm_nt_bf16<true, false>), dim3((unsigned)row_tiles), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)X, (const unsigned short*)W, bias, (const unsigned short*)gate,
                       (unsigned short*)Y, (long long)R, N, K, (long long)ldx, (long long)ldy, flags, thresh, inv_keep,
                       (unsigned long long)seed, rstream, (const float*)nullptr, 0, NtGather{});
  } else {
    const long long blocks = ((row_tiles + 7) / 8) * 8 * (N / NT_BN);
    TG_CHECK(blocks <= 2147483647LL, "tg_gemm_nt_bf16: too many tiles (R=%lld)", (long long)R);
    hipLaunchKernelGGL((k_gemm_nt_bf16<false, false>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)X, (const unsigned short*)W, bias, (const unsigned short*)gate,
                       (unsigned short*)Y, (long long)R, N, K, (long long)ldx, (long long)ldy, flags, thresh, inv_keep,
                       (unsigned long long)seed, rstream, (const float*)nullptr, 0, NtGather{});
  }
  TG_LAUNCH_CHECK();
  return 0;
}

// Y[R,N] = epilogue([S0[i0[r]] | S1[i1[r]] | S2[i2[r]]] W[N,384]^T): tg_gemm_nt_bf16 with the X operand gathered on the
// fly from three 128-column sources (tg_gather3: row pitch in elements, idx NULL = identity).  flags: 1 ReLU, 4 Y +=.
extern "C" int tg_gemm_nt_gather3_bf16(const tg_gather3* gs, const void* W, const float* bias, void* Y, int64_t R, int32_t N,
                                       int64_t ldy, int32_t flags, void* stream) {
  TG_CHECK(gs && W && Y && R > 0 && N > 0 && N % NT_BN == 0, "tg_gemm_nt_gather3_bf16: bad shape (R=%lld N=%d)", (long long)R, N);
  TG_CHECK(ldy % 8 == 0 && ldy >= N && (flags & ~(NT_RELU | NT_ACCUM)) == 0, "tg_gemm_nt_gather3_bf16: bad ldy / flags");
  TG_CHECK(R <= 2147483647LL, "tg_gemm_nt_gather3_bf16: row indices are 32-bit");
  NtGather g;
  for (int c = 0; c < 3; ++c) {
    TG_CHECK(gs->src[c] && gs->stride[c] >= 128 && gs->stride[c] % 8 == 0 &&
                 (reinterpret_cast<uintptr_t>(gs->src[c]) & 15) == 0,
             "tg_gemm_nt_gather3_bf16: source %d must be a 16-byte aligned [*, >=128] bf16 matrix", c);
    g.src[c] = (const unsigned short*)gs->src[c];
    g.idx[c] = gs->idx[c];
    g.stride[c] = gs->stride[c];
  }
  TG_CHECK(((reinterpret_cast<uintptr_t>(W) | reinterpret_cast<uintptr_t>(Y)) & 15) == 0,
           "tg_gemm_nt_gather3_bf16: operands must be 16-byte aligned");
  const long long row_tiles = (R + NT_BM - 1) / NT_BM;
  const long long blocks = ((row_tiles + 7) / 8) * 8 * (N / NT_BN);
  TG_CHECK(blocks <= 2147483647LL, "tg_gemm_nt_gather3_bf16: too many tiles (R=%lld)", (long long)R);
  hipLaunchKernelGGL((k_gemm_nt_bf16<false, false, true>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     (const unsigned short*)nullptr, (const unsigned short*)W, bias, (const unsigned short*)nullptr,
                     (unsigned short*)Y, (long long)R, N, 3 * NT_BK, 0LL, (long long)ldy, flags, 0u, 1.f, 0ull, 0u,
                     (const float*)nullptr, 0, g);
  TG_LAUNCH_CHECK();
  return 0;
}

// Y[R,N] (+)= X W_0^T + amp(r) * X W_1^T + att(r) * X W_2^T; W is [N, 3*kreal] with 128-column block 3c+s =
// W_s[:, 128c:128c+128] (s: 0 identity, 1 amp, 2 att) and scales fp32 [R,2] = (amp, att): the PNA post projection with the degree scalers folded in
// (forward: X = agg [R,4F], N = F; input gradient: X = dOut [R,F], N = 4F).  flags: 0 or 4 (Y +=).
extern "C" int tg_gemm_nt_scaled_bf16(const void* X, const void* W, const float* scales, void* Y, int64_t R, int32_t N,
                                      int32_t kreal, int64_t ldx, int64_t ldy, int32_t flags, void* stream) {
  TG_CHECK(R > 0 && N > 0 && kreal > 0 && N % NT_BN == 0 && kreal % NT_BK == 0,
           "tg_gemm_nt_scaled_bf16: need N %% 128 == 0 and kreal %% 128 == 0 (R=%lld N=%d kreal=%d)", (long long)R, N, kreal);
  TG_CHECK(X && W && scales && Y, "tg_gemm_nt_scaled_bf16: null operand");
  TG_CHECK(ldx % 8 == 0 && ldy % 8 == 0 && ldx >= kreal && ldy >= N, "tg_gemm_nt_scaled_bf16: bad row strides");
  TG_CHECK(((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(W) | reinterpret_cast<uintptr_t>(Y)) & 15) == 0 &&
               (reinterpret_cast<uintptr_t>(scales) & 7) == 0,
           "tg_gemm_nt_scaled_bf16: operands must be 16-byte aligned");
  TG_CHECK((flags & ~NT_ACCUM) == 0, "tg_gemm_nt_scaled_bf16: only the accumulate flag is supported");
  const long long row_tiles = (R + NT_BM - 1) / NT_BM;
  const long long blocks = ((row_tiles + 7) / 8) * 8 * (N / NT_BN);
  TG_CHECK(blocks <= 2147483647LL, "tg_gemm_nt_scaled_bf16: too many tiles (R=%lld)", (long long)R);
  hipLaunchKernelGGL((k_gemm_nt_bf16<false, true>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     (const unsigned short*)X, (const unsigned short*)W, (const float*)nullptr,
                     (const unsigned short*)nullptr, (unsigned short*)Y, (long long)R, N, 3 * kreal, (long long)ldx,
                     (long long)ldy, flags, 0u, 1.f, 0ull, 0u, scales, kreal, NtGather{});
  TG_LAUNCH_CHECK();
  return 0;
}

// z = res + drop(X W^T + bias) (bf16, kept for the backward), out = LayerNorm(z) * gamma + beta, stats = (mean, rstd)
// per row.  N = d_model = 128; K % 128 == 0; res, Z, OUT contiguous [R,128].
extern "C" int tg_gemm_nt_ln_bf16(const void* X, const void* W, const float* bias, const void* res, const float* gamma,
                                  const float* beta, void* Z, void* OUT, float* stats, int64_t R, int32_t K, int64_t ldx,
                                  float eps, float p_drop, uint64_t seed, uint32_t rstream, void* stream) {
  TG_CHECK(R > 0 && K > 0 && K % NT_BK == 0 && ldx % 8 == 0 && ldx >= K, "tg_gemm_nt_ln_bf16: bad shape R=%lld K=%d",
           (long long)R, K);
  TG_CHECK(X && W && res && gamma && beta && Z && OUT && stats, "tg_gemm_nt_ln_bf16: null operand");
  TG_CHECK(((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(W) | reinterpret_cast<uintptr_t>(res) |
             reinterpret_cast<uintptr_t>(Z) | reinterpret_cast<uintptr_t>(OUT) | reinterpret_cast<uintptr_t>(gamma) |
             reinterpret_cast<uintptr_t>(beta)) & 15) == 0 && (!bias || (reinterpret_cast<uintptr_t>(bias) & 15) == 0),
           "tg_gemm_nt_ln_bf16: operands must be 16-byte aligned");
  unsigned thresh = p_drop > 0.f ? drop_threshold(p_drop) : 0u;
  float inv_keep = p_drop < 1.f ? 1.f / (1.f - p_drop) : 0.f;
  const long long row_tiles = (R + NT_BM - 1) / NT_BM;
  TG_CHECK(row_tiles <= 2147483647LL, "tg_gemm_nt_ln_bf16: too many tiles");
  hipLaunchKernelGGL(k_gemm_nt_ln_bf16, dim3((unsigned)row_tiles), dim3(256), 0, (hipStream_t)stream,
                     (const unsigned short*)X, (const unsigned short*)W, bias, (const unsigned short*)res, gamma, beta,
                     (unsigned short*)Z, (unsigned short*)OUT, stats, (long long)R, K, (long long)ldx, eps, thresh,
                     inv_keep, (unsigned long long)seed, rstream);
  TG_LAUNCH_CHECK();
  return 0;
}

