// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of the fused
// tabular-transformer + PNA hot path.  wave = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

namespace tg {

// ---------------------------------------------------------------- errors
void set_error(const char* fmt, ...);
#define TG_CHECK(cond, ...)                         \
  do {                                              \
    if (!(cond)) {                                  \
      tg::set_error(__VA_ARGS__);                   \
      return 1;                                     \
    }                                               \
  } while (0)
#define TG_LAUNCH_CHECK()                                                   \
  do {                                                                      \
    hipError_t e__ = hipGetLastError();                                     \
    if (e__ != hipSuccess) {                                                \
      tg::set_error("%s:%d launch failed: %s", __FILE__, __LINE__,          \
                    hipGetErrorString(e__));                                \
      return 2;                                                             \
    }                                                                       \
  } while (0)

enum DType : int { F32 = 0, BF16 = 1 };

// ---------------------------------------------------------------- bf16 storage type
struct bf16_t {
  unsigned short v;
};

__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }
__device__ __forceinline__ unsigned short f2bf(float f) {
  // round-to-nearest-even; NaN stays NaN (plain cast lowers to v_cvt_pk_bf16_f32 on gfx950)
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(unsigned short, b);
}

template <typename T> __device__ __forceinline__ float to_f(T x);
template <> __device__ __forceinline__ float to_f<float>(float x) { return x; }
template <> __device__ __forceinline__ float to_f<bf16_t>(bf16_t x) { return bf2f(x.v); }
template <typename T> __device__ __forceinline__ T from_f(float x);
template <> __device__ __forceinline__ float from_f<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16_t from_f<bf16_t>(float x) { return bf16_t{f2bf(x)}; }

// 16-byte vector access: 4 floats or 8 bf16 per lane (guide: 16 B/lane is the coalescing sweet spot)
template <typename T> struct V16;
template <> struct V16<float> {
  static constexpr int N = 4;
  __device__ static __forceinline__ void load(const float* p, float (&o)[4]) {
    float4 r = *reinterpret_cast<const float4*>(p);
    o[0] = r.x; o[1] = r.y; o[2] = r.z; o[3] = r.w;
  }
  __device__ static __forceinline__ void store(float* p, const float (&o)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]);
  }
};
template <> struct V16<bf16_t> {
  static constexpr int N = 8;
  __device__ static __forceinline__ void load(const bf16_t* p, float (&o)[8]) {
    uint4 r = *reinterpret_cast<const uint4*>(p);
    unsigned w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      o[2 * i] = __uint_as_float(w[i] << 16);
      o[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
  }
  __device__ static __forceinline__ void store(bf16_t* p, const float (&o)[8]) {
    unsigned w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = (unsigned)f2bf(o[2 * i]) | ((unsigned)f2bf(o[2 * i + 1]) << 16);
    *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
  }
};

// generic small-vector access (VEC elements, VEC*sizeof(T) in {4,8,16} bytes)
template <typename T, int VEC> __device__ __forceinline__ void loadv(const T* p, float (&o)[VEC]) {
  if constexpr (VEC == V16<T>::N) {
    V16<T>::load(p, o);
  } else if constexpr (sizeof(T) == 2 && VEC == 2) {   // 4-byte access: 2 bf16
    unsigned r = *reinterpret_cast<const unsigned*>(p);
    o[0] = __uint_as_float(r << 16); o[1] = __uint_as_float(r & 0xffff0000u);
  } else if constexpr (sizeof(T) == 2 && VEC == 4) {   // 8-byte access: 4 bf16
    uint2 r = *reinterpret_cast<const uint2*>(p);
    o[0] = __uint_as_float(r.x << 16); o[1] = __uint_as_float(r.x & 0xffff0000u);
    o[2] = __uint_as_float(r.y << 16); o[3] = __uint_as_float(r.y & 0xffff0000u);
  } else {
#pragma unroll
    for (int i = 0; i < VEC; ++i) o[i] = to_f<T>(p[i]);
  }
}
template <typename T, int VEC> __device__ __forceinline__ void storev(T* p, const float (&o)[VEC]) {
  if constexpr (VEC == V16<T>::N) {
    V16<T>::store(p, o);
  } else if constexpr (sizeof(T) == 2 && VEC == 2) {
    *reinterpret_cast<unsigned*>(p) = (unsigned)f2bf(o[0]) | ((unsigned)f2bf(o[1]) << 16);
  } else if constexpr (sizeof(T) == 2 && VEC == 4) {
    uint2 w;
    w.x = (unsigned)f2bf(o[0]) | ((unsigned)f2bf(o[1]) << 16);
    w.y = (unsigned)f2bf(o[2]) | ((unsigned)f2bf(o[3]) << 16);
    *reinterpret_cast<uint2*>(p) = w;
  } else {
#pragma unroll
    for (int i = 0; i < VEC; ++i) p[i] = from_f<T>(o[i]);
  }
}

// ---------------------------------------------------------------- counter-based dropout RNG
// keep(seed, stream, idx): one 32-bit hash per element; recomputed in backward from the same
// (seed, stream) so no mask is ever stored.
__device__ __forceinline__ unsigned mix32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
// The key depends only on (seed, stream, high word of the index): kernels compute it ONCE per vector of consecutive
// elements (a vector starting at a multiple of its length never crosses a 2^32 boundary) and pay one mix32 per
// element instead of three — the mask hash was ~25 % of the LayerNorm kernels' time.
__device__ __forceinline__ unsigned rng_key(unsigned long long seed, unsigned stream, unsigned hi) {
  return mix32((unsigned)seed ^ (stream * 0x9E3779B9U) ^ mix32(hi + (unsigned)(seed >> 32) + 0x85ebca6bU));
}
// The seed ARGUMENT of every mask-drawing entry point is either the seed itself (bit 63 clear) or — TG_SEED_DEVICE of
// include/tabgnn_hip.h, bit 63 set — the device address of a 64-bit seed word, read ONCE at the top of the kernel
// (live_seed).  A captured HIP graph bakes its kernels' arguments in; the word behind the pointer is advanced by the
// graph's first node (tg_advance_step), so every replay draws the masks of its own step.  There is no library-side
// state: two graphs with two step records do not see each other (ABI v6; v5 kept one hidden copy of the word per
// translation unit, written by tg_seed_source_sync).
__device__ __forceinline__ unsigned long long live_seed(unsigned long long seed) {
  if (seed >> 63) seed = *reinterpret_cast<const unsigned long long*>(seed & 0x7fffffffffffffffull);     // (wave-uniform)
  return seed;
}
__device__ __forceinline__ unsigned rng_u32(unsigned long long seed, unsigned stream, unsigned long long idx) {
  return mix32((unsigned)idx ^ rng_key(seed, stream, (unsigned)(idx >> 32)));
}
// Keep / drop decisions: ONE 32-bit hash decides several consecutive elements (the two integer multiplies of mix32 are
// quarter-rate instructions; at one hash per element the hash was ~20 % of the fused encoder kernels' time):
//   * BITS = 16: its low / high half against a 16-bit threshold decide elements 2q, 2q+1 (p honoured to 2^-16),
//   * BITS = 8 : its four bytes against an 8-bit threshold decide elements 4q .. 4q+3.  Used when the threshold loses
//     nothing in 8 bits, i.e. p is a multiple of 1/256 (drop_bits8()),
//   * BITS = 1 : p = 1/2 exactly (the reference's backbone dropout, fused.py:62; drop_bits1()): ONE hash decides the 32
//     consecutive elements 32q .. 32q+31, element e is kept when bit (e & 31) of mix32((e >> 5) ^ key) is set.  The
//     fused encoder kernels then pay 4 hashes per token row of 128 channels and site, and turn a bit into an AND mask
//     with one v_bfe_i32 (no compare / select).
// The choice is a function of the threshold alone and every kernel of the package derives its masks through these
// functions, so forward, backward, fused and op-by-op paths agree on every element.
__device__ __host__ __forceinline__ bool drop_bits1(unsigned thresh) { return thresh == 0x80000000u; }
__device__ __host__ __forceinline__ bool drop_bits8(unsigned thresh) { return (thresh & 0x00ffffffu) == 0u; }
// template value of a threshold: 0 = no dropout, else the hash bits per element (1 | 8 | 16)
__host__ __forceinline__ int drop_mode(unsigned thresh) {
  return thresh == 0u ? 0 : drop_bits1(thresh) ? 1 : drop_bits8(thresh) ? 8 : 16;
}
// returns the multiplicative factor: 0 or 1/(1-p)
template <int BITS>
__device__ __forceinline__ float drop_scale_key_t(unsigned key, unsigned lo, unsigned thresh, float inv_keep) {
  if constexpr (BITS == 1) {
    const unsigned hsh = mix32((lo >> 5) ^ key);
    return ((hsh >> (lo & 31u)) & 1u) ? inv_keep : 0.f;
  } else if constexpr (BITS == 8) {
    const unsigned hsh = mix32((lo >> 2) ^ key);
    return ((hsh >> (8u * (lo & 3u))) & 0xffu) >= (thresh >> 24) ? inv_keep : 0.f;
  } else {
    const unsigned hsh = mix32((lo >> 1) ^ key);
    const unsigned bits = (lo & 1u) ? (hsh >> 16) : (hsh & 0xffffu);
    return bits >= (thresh >> 16) ? inv_keep : 0.f;
  }
}
// same value as drop_scale(seed, stream, idx) with key = rng_key(seed, stream, idx >> 32), lo = (unsigned)idx
__device__ __forceinline__ float drop_scale_key(unsigned key, unsigned lo, unsigned thresh, float inv_keep) {
  return drop_bits1(thresh) ? drop_scale_key_t<1>(key, lo, thresh, inv_keep)        // uniform branches
       : drop_bits8(thresh) ? drop_scale_key_t<8>(key, lo, thresh, inv_keep)
                            : drop_scale_key_t<16>(key, lo, thresh, inv_keep);
}
__device__ __forceinline__ float drop_scale(unsigned long long seed, unsigned stream, unsigned long long idx,
                                            unsigned thresh, float inv_keep) {
  return drop_scale_key(rng_key(seed, stream, (unsigned)(idx >> 32)), (unsigned)idx, thresh, inv_keep);
}
// factors of the four consecutive elements lo0 .. lo0+3 (lo0 a multiple of 4, same key): two hashes, or one
template <int BITS>
__device__ __forceinline__ void drop_scale4_t(unsigned key, unsigned lo0, unsigned thresh, float inv_keep, float (&m)[4]) {
  if constexpr (BITS == 1) {
    const unsigned h = mix32((lo0 >> 5) ^ key) >> (lo0 & 31u);
    m[0] = (h & 1u) ? inv_keep : 0.f;
    m[1] = (h & 2u) ? inv_keep : 0.f;
    m[2] = (h & 4u) ? inv_keep : 0.f;
    m[3] = (h & 8u) ? inv_keep : 0.f;
  } else if constexpr (BITS == 8) {
    const unsigned t8 = thresh >> 24;
    const unsigned h = mix32((lo0 >> 2) ^ key);
    m[0] = (h & 0xffu) >= t8 ? inv_keep : 0.f;
    m[1] = ((h >> 8) & 0xffu) >= t8 ? inv_keep : 0.f;
    m[2] = ((h >> 16) & 0xffu) >= t8 ? inv_keep : 0.f;
    m[3] = (h >> 24) >= t8 ? inv_keep : 0.f;
  } else {
    const unsigned t16 = thresh >> 16;
    const unsigned h0 = mix32((lo0 >> 1) ^ key), h1 = mix32(((lo0 >> 1) + 1u) ^ key);
    m[0] = (h0 & 0xffffu) >= t16 ? inv_keep : 0.f;
    m[1] = (h0 >> 16) >= t16 ? inv_keep : 0.f;
    m[2] = (h1 & 0xffffu) >= t16 ? inv_keep : 0.f;
    m[3] = (h1 >> 16) >= t16 ? inv_keep : 0.f;
  }
}
__device__ __forceinline__ void drop_scale4(unsigned key, unsigned lo0, unsigned thresh, float inv_keep, float (&m)[4]) {
  if (drop_bits1(thresh)) drop_scale4_t<1>(key, lo0, thresh, inv_keep, m);
  else if (drop_bits8(thresh)) drop_scale4_t<8>(key, lo0, thresh, inv_keep, m);
  else drop_scale4_t<16>(key, lo0, thresh, inv_keep, m);
}
// Stream-ordered zero fill by a kernel.  hipMemsetAsync is NOT used on the path: on virtual-memory-managed
// allocations (hipMemCreate/hipMemMap: the debug fence allocator of tools/guard_alloc.cpp, torch's expandable segments)
// it was observed to run out of order with the kernels of the same stream.
template <int UNUSED>
__global__ void k_zero_words(unsigned* __restrict__ p, long long n) {
  long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = 0u;
}
inline void zero_async(void* p, size_t bytes, hipStream_t st) {      // bytes % 4 == 0
  const long long n = (long long)(bytes / 4);
  if (n == 0) return;
  long long blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL((k_zero_words<0>), dim3((unsigned)blocks), dim3(256), 0, st, (unsigned*)p, n);
}

inline unsigned drop_threshold(float p) {
  double t = (double)p * 4294967296.0;
  if (t < 0) t = 0;
  if (t > 4294967295.0) t = 4294967295.0;
  return (unsigned)t;
}

// ---------------------------------------------------------------- wave reductions (64 lanes)
template <int WIDTH> __device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = WIDTH / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
template <int WIDTH> __device__ __forceinline__ float group_max(float v) {
#pragma unroll
  for (int o = WIDTH / 2; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

inline int ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }
inline int grid_cap(long long blocks, int cap = 256 * 8) { return (int)(blocks < 1 ? 1 : (blocks > cap ? cap : blocks)); }
// One pass per workgroup (no grid-stride iterations) for pure streaming kernels without a per-block prologue: measured
// on 1 GiB bf16 operands, k_axpby 4.8 -> 6.0 TB/s and k_act_dropout 5.3 -> 6.3 TB/s against the 2048-block cap
// (tools/stream_probe.py; torch's elementwise add, launched the same way, reaches 6.1-6.4 TB/s).  Only those two
// use it: kernels with a per-block prologue lose badly without the cap (LayerNorm 4.0 -> 1.2 TB/s, k_bn_bwd_apply
// 0.24 -> 1.37 ms/step), the gathers lose a little (k_gather_concat3 0.69 -> 0.81 ms/step), attention, segmented sums
// and the aggregation backward do not care.
inline int grid_full(long long blocks, int old_cap = 256 * 8) {
  static const bool off_ = getenv("TG_NO_FULL_GRID") != nullptr;      // same-box A/B switch: the capped launch it replaced
  if (off_) return grid_cap(blocks, old_cap);
  return (int)(blocks < 1 ? 1 : (blocks > 2147483647LL ? 2147483647LL : blocks));
}

}  // namespace tg
