// The readout MLP of ClassifierHead / NodeClassificationHead (src/nn/gnn/decoder.py:5-32):
//     Linear(D0, 50) -> ReLU -> Dropout -> Linear(50, 25) -> ReLU -> Dropout -> Linear(25, n_classes)
// as ONE forward kernel and ONE backward kernel (+ a block-ordered reduce of the parameter-gradient partials).
// The products are 50 / 25 / n_classes wide — far below an MFMA tile, 20 k multiply-adds per row — and what they cost
// op by op is launches: 3 GEMMs + 2 activation kernels + a cast forward, 6 GEMMs + 3 column sums + 2 activation
// kernels + casts + up to 6 gradient accumulations backward (0.25 ms of a 17 ms step at B = 8192, 0.2 of 3.0 ms at the
// reference's default B = 200).  Here a wave owns a row: lane j holds hidden unit j, the row and the weights sit in LDS
// (W1 rows padded by 16 bytes so that the 16-lane groups of a ds_read_b128 cover all banks), layers 2 and 3 exchange
// their activations through 256 bytes of wave-private LDS.  Arithmetic as the op-by-op path: fp32 accumulation,
// pre-activations and activations rounded to the storage type T, the last layer and the logits in fp32; dropout masks
// are the pure function of (seed, stream, element index) every kernel of the library shares (common.hpp), so forward,
// backward and the op-by-op composition agree on them.
#include "../../include/tabgnn_hip.h"
#include "common.hpp"

namespace tg {

struct HeadArgs {
  const void *h0, *w1, *w2, *z1c, *z2c;
  const float *b1, *b2, *w3, *b3, *g;
  void *z1, *z2, *dh0;
  float *logits, *part;
  long long B;
  int D0, NC;
  unsigned thresh;
  float inv_keep;
  unsigned long long seed;
  unsigned rs1, rs2;
};

constexpr int HD_WAVES = 4;
constexpr int HD_RPW = 4;                       // rows a wave works on at once (weight reads amortised, 4 independent chains)
constexpr int HD_TILE = HD_WAVES * HD_RPW;      // rows per workgroup pass

template <typename T> __device__ __forceinline__ float round_t(float v) { return to_f<T>(from_f<T>(v)); }

// LDS layout (bytes), shared by both kernels: W1 [H1][D0 + pad] T | W2 [H2][H1 + 1] f32 | W3 [NC][H2 + 1] f32 |
// b1 | b2 | b3 | per row of the tile: x [D0] T, then float rows a1[64] a2[64] dz1[64] dz2[64] g[16]
template <typename T, int H1, int H2> struct HeadLds {
  int D0, NC, ldw1;
  __device__ __host__ HeadLds(int d0, int nc) : D0(d0), NC(nc), ldw1(d0 + 16 / (int)sizeof(T)) {}
  __device__ __host__ size_t w1() const { return 0; }
  __device__ __host__ size_t w2() const { return ((size_t)H1 * ldw1 * sizeof(T) + 15) & ~(size_t)15; }
  __device__ __host__ size_t w3() const { return w2() + (size_t)H2 * (H1 + 1) * 4; }
  __device__ __host__ size_t bias() const { return w3() + (size_t)NC * (H2 + 1) * 4; }
  __device__ __host__ size_t row0() const { return (bias() + (size_t)(H1 + H2 + NC) * 4 + 15) & ~(size_t)15; }
  __device__ __host__ size_t xbytes() const { return ((size_t)D0 * sizeof(T) + 15) & ~(size_t)15; }
  __device__ __host__ size_t row_bytes() const { return xbytes() + (4 * 64 + 16) * 4; }
  __device__ __host__ size_t total() const { return row0() + HD_TILE * row_bytes(); }
};

template <typename T, int H1, int H2>
__device__ __forceinline__ void head_stage_weights(const HeadArgs& a, char* smem, const HeadLds<T, H1, H2>& L) {
  constexpr int VEC = V16<T>::N;
  T* w1s = reinterpret_cast<T*>(smem + L.w1());
  const T* w1 = (const T*)a.w1;
  const int vpr = a.D0 / VEC;
  for (int i = threadIdx.x; i < H1 * vpr; i += blockDim.x) {
    const int j = i / vpr, k = (i - j * vpr) * VEC;
    *reinterpret_cast<uint4*>(w1s + (size_t)j * L.ldw1 + k) = *reinterpret_cast<const uint4*>(w1 + (size_t)j * a.D0 + k);
  }
  float* w2s = reinterpret_cast<float*>(smem + L.w2());
  const T* w2 = (const T*)a.w2;
  for (int i = threadIdx.x; i < H2 * H1; i += blockDim.x) w2s[(i / H1) * (H1 + 1) + i % H1] = to_f<T>(w2[i]);
  float* w3s = reinterpret_cast<float*>(smem + L.w3());
  for (int i = threadIdx.x; i < a.NC * H2; i += blockDim.x) w3s[(i / H2) * (H2 + 1) + i % H2] = a.w3[i];
  if (a.b1) {                                     // (the backward does not use the biases)
    float* bs = reinterpret_cast<float*>(smem + L.bias());
    for (int i = threadIdx.x; i < H1 + H2 + a.NC; i += blockDim.x)
      bs[i] = i < H1 ? a.b1[i] : i < H1 + H2 ? a.b2[i - H1] : a.b3[i - H1 - H2];
  }
}

__device__ __forceinline__ float head_mask(unsigned long long seed, unsigned rs, unsigned long long idx, unsigned thresh,
                                           float inv_keep) {
  return thresh ? drop_scale(seed, rs, idx, thresh, inv_keep) : 1.f;
}

// the wave's HD_RPW rows of h0 into LDS (zeros for rows past B)
template <typename T>
__device__ __forceinline__ void head_load_rows(const HeadArgs& a, char* rows, size_t row_bytes, long long row0, int lane) {
  constexpr int VEC = V16<T>::N;
#pragma unroll
  for (int r = 0; r < HD_RPW; ++r) {
    T* xrow = reinterpret_cast<T*>(rows + r * row_bytes);
    const long long row = row0 + r;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int k = (lane + 64 * h) * VEC;
      if (k < a.D0) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (row < a.B) v = *reinterpret_cast<const uint4*>((const T*)a.h0 + row * a.D0 + k);
        *reinterpret_cast<uint4*>(xrow + k) = v;
      }
    }
  }
}

template <typename T, int H1, int H2>
__global__ void __launch_bounds__(64 * HD_WAVES) k_head_fwd(const HeadArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int VEC = V16<T>::N;
  const HeadLds<T, H1, H2> L(a.D0, a.NC);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const unsigned long long seed = live_seed(a.seed);
  head_stage_weights<T, H1, H2>(a, smem, L);
  const T* w1s = reinterpret_cast<const T*>(smem + L.w1());
  const float* w2s = reinterpret_cast<const float*>(smem + L.w2());
  const float* w3s = reinterpret_cast<const float*>(smem + L.w3());
  const float* bs = reinterpret_cast<const float*>(smem + L.bias());
  const size_t rb = L.row_bytes(), xb = L.xbytes();
  char* rows = smem + L.row0() + (size_t)wave * HD_RPW * rb;      // this wave's rows: x | a1 | a2 | ...
  __syncthreads();
  for (long long base = (long long)blockIdx.x * HD_TILE; base < a.B; base += (long long)gridDim.x * HD_TILE) {
    const long long row0 = base + wave * HD_RPW;
    head_load_rows<T>(a, rows, rb, row0, lane);
    __syncthreads();
    if (lane < H1) {
      float acc[HD_RPW];
#pragma unroll
      for (int r = 0; r < HD_RPW; ++r) acc[r] = 0.f;
      const T* wrow = w1s + (size_t)lane * L.ldw1;
      for (int k = 0; k < a.D0; k += VEC) {
        float w[VEC];
        loadv<T, VEC>(wrow + k, w);
#pragma unroll
        for (int r = 0; r < HD_RPW; ++r) {
          float x[VEC];
          loadv<T, VEC>(reinterpret_cast<const T*>(rows + r * rb) + k, x);
#pragma unroll
          for (int e = 0; e < VEC; ++e) acc[r] = fmaf(w[e], x[e], acc[r]);
        }
      }
#pragma unroll
      for (int r = 0; r < HD_RPW; ++r) {
        const long long row = row0 + r;
        const float z = round_t<T>(acc[r] + bs[lane]);
        if (row < a.B) ((T*)a.z1)[row * H1 + lane] = from_f<T>(z);
        reinterpret_cast<float*>(rows + r * rb + xb)[lane] =
            round_t<T>(fmaxf(z, 0.f) * head_mask(seed, a.rs1, (unsigned long long)row * H1 + lane, a.thresh, a.inv_keep));
      }
    }
    __syncthreads();
    if (lane < H2) {
      float acc[HD_RPW];
#pragma unroll
      for (int r = 0; r < HD_RPW; ++r) acc[r] = bs[H1 + lane];
#pragma unroll 10
      for (int j = 0; j < H1; ++j) {
        const float w = w2s[lane * (H1 + 1) + j];
#pragma unroll
        for (int r = 0; r < HD_RPW; ++r) acc[r] = fmaf(reinterpret_cast<const float*>(rows + r * rb + xb)[j], w, acc[r]);
      }
#pragma unroll
      for (int r = 0; r < HD_RPW; ++r) {
        const long long row = row0 + r;
        const float z = round_t<T>(acc[r]);
        if (row < a.B) ((T*)a.z2)[row * H2 + lane] = from_f<T>(z);
        reinterpret_cast<float*>(rows + r * rb + xb)[64 + lane] =
            round_t<T>(fmaxf(z, 0.f) * head_mask(seed, a.rs2, (unsigned long long)row * H2 + lane, a.thresh, a.inv_keep));
      }
    }
    __syncthreads();
    if (lane < a.NC) {
#pragma unroll
      for (int r = 0; r < HD_RPW; ++r) {
        const long long row = row0 + r;
        float acc = bs[H1 + H2 + lane];
#pragma unroll 5
        for (int j = 0; j < H2; ++j) acc = fmaf(reinterpret_cast<const float*>(rows + r * rb + xb)[64 + j], w3s[lane * (H2 + 1) + j], acc);
        if (row < a.B) a.logits[row * a.NC + lane] = acc;
      }
    }
  }
}

// Backward.  Phase A (a wave owns HD_RPW rows): g -> d z2 -> d z1 down the chain, activations recomputed from the saved
// pre-activations and the masks; everything phase B needs is parked in LDS per row.  Phase B (all 256 threads, the
// tile's 16 rows in groups of four): thread t owns input columns t and t + 256 — d h0 of those columns and the running
// W1 gradient [H1][2] in registers across ALL tiles of the persistent block — plus a strided share of the small
// W2 / W3 / bias gradients.  One fp32 partial [P] per block, P = H1 D0 + H1 + H2 H1 + H2 + NC H2 + NC; k_head_reduce
// adds them in block order.
template <typename T, int H1, int H2>
__global__ void __launch_bounds__(64 * HD_WAVES) k_head_bwd(const HeadArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NT = 64 * HD_WAVES;
  constexpr int W2E = (H1 * H2 + NT - 1) / NT;                 // W2 gradient elements per thread
  constexpr int H1P = H1;
  const HeadLds<T, H1, H2> L(a.D0, a.NC);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, t = threadIdx.x;
  const unsigned long long seed = live_seed(a.seed);
  head_stage_weights<T, H1, H2>(a, smem, L);
  const T* w1s = reinterpret_cast<const T*>(smem + L.w1());
  const float* w2s = reinterpret_cast<const float*>(smem + L.w2());
  const float* w3s = reinterpret_cast<const float*>(smem + L.w3());
  const size_t rb = L.row_bytes(), xb = L.xbytes();
  char* tile = smem + L.row0();
  char* rows = tile + (size_t)wave * HD_RPW * rb;
  const int c0 = t, c1 = t + NT;
  const bool has0 = c0 < a.D0, has1 = c1 < a.D0;
  const int cc0 = has0 ? c0 : 0, cc1 = has1 ? c1 : 0;          // clamped: reads need no branch, stores are masked
  float accw1[H1P][2];
#pragma unroll
  for (int j = 0; j < H1P; ++j) accw1[j][0] = accw1[j][1] = 0.f;
  float accw2[W2E];
#pragma unroll
  for (int i = 0; i < W2E; ++i) accw2[i] = 0.f;
  float accw3 = 0.f, accb = 0.f;                                // thread t < NC*H2: W3 element t; t < H1+H2+NC: a bias
  int w2a[W2E], w2b[W2E];                                        // LDS float offsets of this thread's W2 elements
#pragma unroll
  for (int i = 0; i < W2E; ++i) {
    const int e = t + i * NT;
    w2a[i] = e < H1 * H2 ? 192 + e / H1 : 255;                  // dz2[j2]   (slot 255 of the row is always zero)
    w2b[i] = e < H1 * H2 ? e % H1 : 63;                         // a1[j]
  }
  const int w3a = t < a.NC * H2 ? 256 + t / H2 : 255, w3b = t < a.NC * H2 ? 64 + t % H2 : 63;
  const int ba = t < H1 ? 128 + t : t < H1 + H2 ? 192 + t - H1 : t < H1 + H2 + a.NC ? 256 + t - H1 - H2 : 255;
  __syncthreads();
  for (long long base = (long long)blockIdx.x * HD_TILE; base < a.B; base += (long long)gridDim.x * HD_TILE) {
    const long long row0 = base + wave * HD_RPW;
    // ---- phase A
    head_load_rows<T>(a, rows, rb, row0, lane);
#pragma unroll
    for (int r = 0; r < HD_RPW; ++r) {
      const long long row = row0 + r;
      if (lane < 16) reinterpret_cast<float*>(rows + r * rb + xb)[256 + lane] = (row < a.B && lane < a.NC) ? a.g[row * a.NC + lane] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < HD_RPW; ++r) {
      const long long row = row0 + r;
      float* fs = reinterpret_cast<float*>(rows + r * rb + xb);
      float a2 = 0.f, dz2 = 0.f;
      if (row < a.B && lane < H2) {
        const float z = to_f<T>(((const T*)a.z2c)[row * H2 + lane]);
        const float m = head_mask(seed, a.rs2, (unsigned long long)row * H2 + lane, a.thresh, a.inv_keep);
        a2 = round_t<T>(fmaxf(z, 0.f) * m);
        float da = 0.f;
        for (int c = 0; c < a.NC; ++c) da = fmaf(fs[256 + c], w3s[c * (H2 + 1) + lane], da);
        dz2 = z > 0.f ? round_t<T>(round_t<T>(da) * m) : 0.f;
      }
      fs[64 + lane] = a2;
      fs[192 + lane] = dz2;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < HD_RPW; ++r) {
      const long long row = row0 + r;
      float* fs = reinterpret_cast<float*>(rows + r * rb + xb);
      float a1 = 0.f, dz1 = 0.f;
      if (row < a.B && lane < H1) {
        const float z = to_f<T>(((const T*)a.z1c)[row * H1 + lane]);
        const float m = head_mask(seed, a.rs1, (unsigned long long)row * H1 + lane, a.thresh, a.inv_keep);
        a1 = round_t<T>(fmaxf(z, 0.f) * m);
        float da = 0.f;
#pragma unroll 5
        for (int j = 0; j < H2; ++j) da = fmaf(fs[192 + j], w2s[j * (H1 + 1) + lane], da);
        dz1 = z > 0.f ? round_t<T>(round_t<T>(da) * m) : 0.f;
      }
      fs[lane] = a1;
      fs[128 + lane] = dz1;
    }
    __syncthreads();
    // ---- phase B: row by row, no branches inside (clamped column reads; stores masked)
#pragma unroll 1
    for (int q = 0; q < HD_TILE; ++q) {
      const char* r0 = tile + (size_t)q * rb;
      const T* xr = reinterpret_cast<const T*>(r0);
      const float* f = reinterpret_cast<const float*>(r0 + xb);
      const float x0 = has0 ? to_f<T>(xr[cc0]) : 0.f, x1 = has1 ? to_f<T>(xr[cc1]) : 0.f;
      float d0 = 0.f, d1 = 0.f;
      const T* wp0 = w1s + cc0;
      const T* wp1 = w1s + cc1;
#pragma unroll
      for (int j = 0; j < H1; ++j) {
        const float d = f[128 + j];
        const float wa = to_f<T>(wp0[(size_t)j * L.ldw1]), wb = to_f<T>(wp1[(size_t)j * L.ldw1]);
        accw1[j][0] = fmaf(d, x0, accw1[j][0]);
        accw1[j][1] = fmaf(d, x1, accw1[j][1]);
        d0 = fmaf(d, wa, d0);
        d1 = fmaf(d, wb, d1);
      }
      const long long rr = base + q;
      if (rr < a.B) {
        if (has0) ((T*)a.dh0)[rr * a.D0 + c0] = from_f<T>(d0);
        if (has1) ((T*)a.dh0)[rr * a.D0 + c1] = from_f<T>(d1);
      }
#pragma unroll
      for (int i = 0; i < W2E; ++i) accw2[i] = fmaf(f[w2a[i]], f[w2b[i]], accw2[i]);
      accw3 = fmaf(f[w3a], f[w3b], accw3);
      accb += f[ba];
    }
    __syncthreads();
  }
  // ---- this block's partial: [dW1 | db1 | dW2 | db2 | dW3 | db3]
  const int P = H1 * a.D0 + H1 + H2 * H1 + H2 + a.NC * H2 + a.NC;
  float* part = a.part + (size_t)blockIdx.x * P;
#pragma unroll
  for (int j = 0; j < H1; ++j) {
    if (has0) part[j * a.D0 + c0] = accw1[j][0];
    if (has1) part[j * a.D0 + c1] = accw1[j][1];
  }
  const int o_b1 = H1 * a.D0, o_w2 = o_b1 + H1, o_b2 = o_w2 + H2 * H1, o_w3 = o_b2 + H2, o_b3 = o_w3 + a.NC * H2;
#pragma unroll
  for (int i = 0; i < W2E; ++i) {
    const int e = t + i * NT;
    if (e < H1 * H2) part[o_w2 + e] = accw2[i];
  }
  if (t < a.NC * H2) part[o_w3 + t] = accw3;
  if (t < H1) part[o_b1 + t] = accb;
  else if (t < H1 + H2) part[o_b2 + t - H1] = accb;
  else if (t < H1 + H2 + a.NC) part[o_b3 + t - H1 - H2] = accb;
}

struct HeadOut {
  float* out[6];
  int off[7];
};

// out (+)= sum over blocks of part[b][i], in block order: 64 elements per workgroup, four strips of partials combined in LDS
__global__ void __launch_bounds__(256) k_head_reduce(const float* __restrict__ part, int nblk, int P, HeadOut o, int accumulate) {
  __shared__ float red[256];
  const int i = blockIdx.x * 64 + (threadIdx.x & 63), strip = threadIdx.x >> 6;
  float s0 = 0.f, s1 = 0.f;
  if (i < P) {
    const int per = (nblk + 3) / 4, b0 = strip * per, b1 = b0 + per < nblk ? b0 + per : nblk;
    int b = b0;
    for (; b + 1 < b1; b += 2) {
      s0 += part[(size_t)b * P + i];
      s1 += part[(size_t)(b + 1) * P + i];
    }
    if (b < b1) s0 += part[(size_t)b * P + i];
  }
  red[threadIdx.x] = s0 + s1;
  __syncthreads();
  if (strip == 0 && i < P) {
    const float s = (red[threadIdx.x] + red[threadIdx.x + 64]) + (red[threadIdx.x + 128] + red[threadIdx.x + 192]);
    int seg = 0;
#pragma unroll
    for (int k = 1; k < 6; ++k) seg += i >= o.off[k];
    float* dst = o.out[seg] + (i - o.off[seg]);
    *dst = accumulate ? *dst + s : s;
  }
}

static int head_blocks(int64_t B) {
  const int64_t tiles = (B + HD_TILE - 1) / HD_TILE;
  return (int)(tiles < 1 ? 1 : tiles > 256 ? 256 : tiles);
}

template <typename T> static int head_set_lds(const void* fn, size_t bytes) {
  return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess ? 0 : 1;
}

}  // namespace tg

using namespace tg;

static bool head_shape_ok(int64_t B, int D0, int H1, int H2, int NC) {
  return B >= 0 && D0 >= 8 && D0 % 8 == 0 && D0 <= 512 && H1 == 50 && H2 == 25 && NC >= 1 && NC <= 10;   // (NC * H2 <= 256: one W3 gradient element per thread)
}

extern "C" int32_t tg_head_mlp_supported(int32_t D0, int32_t H1, int32_t H2, int32_t NC) {
  return head_shape_ok(0, D0, H1, H2, NC) ? 1 : 0;
}

extern "C" int64_t tg_head_mlp_partial_floats(int32_t D0, int32_t H1, int32_t H2, int32_t NC) {
  return (int64_t)256 * ((int64_t)H1 * D0 + H1 + (int64_t)H2 * H1 + H2 + (int64_t)NC * H2 + NC);
}

extern "C" int tg_head_mlp_fwd(const void* h0, const void* w1, const float* b1, const void* w2, const float* b2,
                               const float* w3, const float* b3, void* z1, void* z2, float* logits, int64_t B, int32_t D0,
                               int32_t H1, int32_t H2, int32_t NC, float p_drop, uint64_t seed, uint32_t rs1, uint32_t rs2,
                               int32_t dt, void* stream) {
  TG_CHECK(head_shape_ok(B, D0, H1, H2, NC), "tg_head_mlp_fwd: unsupported shape (D0=%d H1=%d H2=%d NC=%d)", D0, H1, H2, NC);
  TG_CHECK(h0 && w1 && b1 && w2 && b2 && w3 && b3 && z1 && z2 && logits, "tg_head_mlp_fwd: null operand");
  TG_CHECK(((reinterpret_cast<uintptr_t>(h0) | reinterpret_cast<uintptr_t>(w1)) & 15) == 0,
           "tg_head_mlp_fwd: h0 and w1 must be 16-byte aligned");
  if (B == 0) return 0;
  HeadArgs a{};
  a.h0 = h0; a.w1 = w1; a.w2 = w2; a.b1 = b1; a.b2 = b2; a.w3 = w3; a.b3 = b3; a.z1 = z1; a.z2 = z2; a.logits = logits;
  a.B = B; a.D0 = D0; a.NC = NC;
  a.thresh = p_drop > 0.f ? drop_threshold(p_drop) : 0u;
  a.inv_keep = p_drop < 1.f ? 1.f / (1.f - p_drop) : 0.f;
  a.seed = seed; a.rs1 = rs1; a.rs2 = rs2;
  const int64_t tiles = (B + HD_TILE - 1) / HD_TILE;
  const int grid = (int)(tiles > 512 ? 512 : tiles);
  if (dt == F32) {
    const size_t lds = HeadLds<float, 50, 25>(D0, NC).total();
    static bool done = false;
    if (!done) { TG_CHECK(head_set_lds<float>((const void*)k_head_fwd<float, 50, 25>, 160 * 1024) == 0, "tg_head_mlp_fwd: LDS attribute"); done = true; }
    hipLaunchKernelGGL((k_head_fwd<float, 50, 25>), dim3(grid), dim3(64 * HD_WAVES), lds, (hipStream_t)stream, a);
  } else {
    const size_t lds = HeadLds<bf16_t, 50, 25>(D0, NC).total();
    static bool done = false;
    if (!done) { TG_CHECK(head_set_lds<bf16_t>((const void*)k_head_fwd<bf16_t, 50, 25>, 160 * 1024) == 0, "tg_head_mlp_fwd: LDS attribute"); done = true; }
    hipLaunchKernelGGL((k_head_fwd<bf16_t, 50, 25>), dim3(grid), dim3(64 * HD_WAVES), lds, (hipStream_t)stream, a);
  }
  TG_LAUNCH_CHECK();
  return 0;
}

extern "C" int tg_head_mlp_bwd(const float* g, const void* h0, const void* z1, const void* z2, const void* w1, const void* w2,
                               const float* w3, void* dh0, float* workspace, float* dw1, float* db1, float* dw2, float* db2,
                               float* dw3, float* db3, int32_t accumulate, int64_t B, int32_t D0, int32_t H1, int32_t H2,
                               int32_t NC, float p_drop, uint64_t seed, uint32_t rs1, uint32_t rs2, int32_t dt, void* stream) {
  TG_CHECK(head_shape_ok(B, D0, H1, H2, NC), "tg_head_mlp_bwd: unsupported shape (D0=%d H1=%d H2=%d NC=%d)", D0, H1, H2, NC);
  TG_CHECK(g && h0 && z1 && z2 && w1 && w2 && w3 && dh0 && workspace && dw1 && db1 && dw2 && db2 && dw3 && db3,
           "tg_head_mlp_bwd: null operand");
  TG_CHECK(((reinterpret_cast<uintptr_t>(h0) | reinterpret_cast<uintptr_t>(w1)) & 15) == 0,
           "tg_head_mlp_bwd: h0 and w1 must be 16-byte aligned");
  HeadArgs a{};
  a.g = g; a.h0 = h0; a.z1c = z1; a.z2c = z2; a.w1 = w1; a.w2 = w2; a.w3 = w3; a.dh0 = dh0; a.part = workspace;
  a.b1 = a.b2 = a.b3 = nullptr;
  a.B = B; a.D0 = D0; a.NC = NC;
  a.thresh = p_drop > 0.f ? drop_threshold(p_drop) : 0u;
  a.inv_keep = p_drop < 1.f ? 1.f / (1.f - p_drop) : 0.f;
  a.seed = seed; a.rs1 = rs1; a.rs2 = rs2;
  const int nblk = B > 0 ? head_blocks(B) : 0;
  const int P = H1 * D0 + H1 + H2 * H1 + H2 + NC * H2 + NC;
  if (B > 0) {
    if (dt == F32) {
      const size_t lds = HeadLds<float, 50, 25>(D0, NC).total();
      static bool done = false;
      if (!done) { TG_CHECK(head_set_lds<float>((const void*)k_head_bwd<float, 50, 25>, 160 * 1024) == 0, "tg_head_mlp_bwd: LDS attribute"); done = true; }
      hipLaunchKernelGGL((k_head_bwd<float, 50, 25>), dim3(nblk), dim3(64 * HD_WAVES), lds, (hipStream_t)stream, a);
    } else {
      const size_t lds = HeadLds<bf16_t, 50, 25>(D0, NC).total();
      static bool done = false;
      if (!done) { TG_CHECK(head_set_lds<bf16_t>((const void*)k_head_bwd<bf16_t, 50, 25>, 160 * 1024) == 0, "tg_head_mlp_bwd: LDS attribute"); done = true; }
      hipLaunchKernelGGL((k_head_bwd<bf16_t, 50, 25>), dim3(nblk), dim3(64 * HD_WAVES), lds, (hipStream_t)stream, a);
    }
    TG_LAUNCH_CHECK();
  }
  HeadOut o;
  o.out[0] = dw1; o.out[1] = db1; o.out[2] = dw2; o.out[3] = db2; o.out[4] = dw3; o.out[5] = db3;
  o.off[0] = 0; o.off[1] = H1 * D0; o.off[2] = o.off[1] + H1; o.off[3] = o.off[2] + H2 * H1; o.off[4] = o.off[3] + H2;
  o.off[5] = o.off[4] + NC * H2; o.off[6] = P;
  hipLaunchKernelGGL(k_head_reduce, dim3((P + 63) / 64), dim3(256), 0, (hipStream_t)stream, workspace, nblk, P, o, accumulate);
  TG_LAUNCH_CHECK();
  return 0;
}
