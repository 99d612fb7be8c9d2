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

__global__ void __launch_bounds__(256) k_gemm_tn_bf16(const unsigned short* __restrict__ G,
                                                       const unsigned short* __restrict__ X,
                                                       float* __restrict__ partial, long long R, int M, int N,
                                                       long long ldg, long long ldx, long long rows_per_slab) {
  __shared__ __attribute__((aligned(16))) char lds[2 * 2 * TILE_BYTES];   // [buf][G|X][64 x 256 B] = 64 KiB
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.x * GM, n0 = blockIdx.y * GN;
  const long long r_begin = (long long)blockIdx.z * rows_per_slab;
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
  uint4 rg0, rg1, rg2, rg3, rx0, rx1, rx2, rx3;
#define TG_LOAD_PIECE(P, RG, RX)                                                                              \
  {                                                                                                           \
    int piece = tid + 256 * P, row = piece >> 4, ch = piece & 15;                                             \
    long long r = r0_ + row;                                                                                  \
    bool okr = r < r_end;                                                                                     \
    RG = make_uint4(0, 0, 0, 0);                                                                              \
    RX = make_uint4(0, 0, 0, 0);                                                                              \
    if (okr && m0 + ch * 8 < M) RG = *reinterpret_cast<const uint4*>(G + r * ldg + m0 + ch * 8);             \
    if (okr && n0 + ch * 8 < N) RX = *reinterpret_cast<const uint4*>(X + r * ldx + n0 + ch * 8);             \
  }
#define load_tile(R0)                                                                                         \
  {                                                                                                           \
    long long r0_ = (R0);                                                                                     \
    TG_LOAD_PIECE(0, rg0, rx0) TG_LOAD_PIECE(1, rg1, rx1) TG_LOAD_PIECE(2, rg2, rx2) TG_LOAD_PIECE(3, rg3, rx3) \
  }
#define TG_STORE_PIECE(P, RG, RX)                                                                             \
  {                                                                                                           \
    int piece = tid + 256 * P, off = lds_off(piece >> 4, piece & 15);                                         \
    *reinterpret_cast<uint4*>(tg_w + off) = RG;                                                               \
    *reinterpret_cast<uint4*>(tg_w + TILE_BYTES + off) = RX;                                                  \
  }
#define store_tile(BUF)                                                                                       \
  {                                                                                                           \
    char* tg_w = lds + (BUF) * 2 * TILE_BYTES;                                                                \
    TG_STORE_PIECE(0, rg0, rx0) TG_STORE_PIECE(1, rg1, rx1) TG_STORE_PIECE(2, rg2, rx2) TG_STORE_PIECE(3, rg3, rx3) \
  }

  if (r_begin < r_end) {
    load_tile(r_begin)
    store_tile(0)
  }
  __syncthreads();
  int buf = 0;
  for (long long r0 = r_begin; r0 < r_end; r0 += GK) {
    const bool more = r0 + GK < r_end;
    if (more) load_tile(r0 + GK)                   // next tile's HBM reads fly under this tile's MFMAs
    const char* tg_ = lds + buf * 2 * TILE_BYTES;
    const char* tx_ = tg_ + TILE_BYTES;
#pragma unroll
    for (int ks = 0; ks < GK / 16; ++ks) {
      v8bf a0 = frag_tr(tg_, ks, wm * 2 + 0, lane), a1 = frag_tr(tg_, ks, wm * 2 + 1, lane);
      v8bf b0 = frag_tr(tx_, ks, wn * 2 + 0, lane), b1 = frag_tr(tx_, ks, wn * 2 + 1, lane);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
    }
    if (more) store_tile(buf ^ 1)
    __syncthreads();
    buf ^= 1;
  }

  // C/D map of the 32x32 tile: col = lane & 31, row = (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5)
  float* out = partial + (long long)blockIdx.z * M * N;
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

__global__ void k_sum_slabs(const float* __restrict__ partial, int nslab, long long mn, float* __restrict__ out) {
  long long i = (blockIdx.x * (long long)blockDim.x + threadIdx.x) * 4;
  if (i >= mn) return;
  if (i + 4 <= mn) {
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s = 0; s < nslab; ++s) {
      float4 v = *reinterpret_cast<const float4*>(partial + (long long)s * mn + i);
      t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
    }
    *reinterpret_cast<float4*>(out + i) = t;
  } else {
    for (long long k = i; k < mn; ++k) {
      float t = 0.f;
      for (int s = 0; s < nslab; ++s) t += partial[(long long)s * mn + k];
      out[k] = t;
    }
  }
}

}  // namespace tg

using namespace tg;

static void tn_geometry(long long R, int M, int N, int& tm, int& tn, int& nslab, long long& rows_per_slab) {
  tm = ceil_div(M, GM);
  tn = ceil_div(N, GN);
  long long steps = ceil_div(R, GK);
  long long want = ceil_div(768, (long long)tm * tn);
  nslab = (int)(want < 1 ? 1 : (want > steps ? steps : want));
  rows_per_slab = ceil_div(steps, nslab) * (long long)GK;
  nslab = (int)ceil_div(R, rows_per_slab);
}

extern "C" int64_t tg_gemm_tn_workspace_floats(int64_t R, int32_t M, int32_t N) {
  int tm, tn, nslab;
  long long rps;
  tn_geometry(R, M, N, tm, tn, nslab, rps);
  return (int64_t)nslab * M * N;
}

// out[M,N] (fp32) = G[R,M]^T X[R,N];  G, X bf16 with row strides ldg, ldx (elements, multiples of 8)
extern "C" int tg_gemm_tn_bf16(const void* G, const void* X, float* out, float* workspace, int64_t R, int32_t M,
                               int32_t N, int64_t ldg, int64_t ldx, void* stream) {
  TG_CHECK(R > 0 && M > 0 && N > 0, "tg_gemm_tn_bf16: empty problem");
  TG_CHECK(M % 8 == 0 && N % 8 == 0 && ldg % 8 == 0 && ldx % 8 == 0,
           "tg_gemm_tn_bf16: M, N and row strides must be multiples of 8 (M=%d N=%d)", M, N);
  TG_CHECK((reinterpret_cast<uintptr_t>(G) & 15) == 0 && (reinterpret_cast<uintptr_t>(X) & 15) == 0,
           "tg_gemm_tn_bf16: operands must be 16-byte aligned");
  int tm, tn, nslab;
  long long rps;
  tn_geometry(R, M, N, tm, tn, nslab, rps);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_gemm_tn_bf16, dim3(tm, tn, nslab), dim3(256), 0, st, (const unsigned short*)G,
                     (const unsigned short*)X, workspace, (long long)R, M, N, (long long)ldg, (long long)ldx, rps);
  long long mn = (long long)M * N;
  hipLaunchKernelGGL(k_sum_slabs, dim3(ceil_div(ceil_div(mn, 4), 256)), dim3(256), 0, st, workspace, nslab, mn, out);
  TG_LAUNCH_CHECK();
  return 0;
}
