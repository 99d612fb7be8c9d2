// Where do the bytes of an LDS-DMA (global_load_lds_dwordx4) go, and when are they visible, when several pieces of one wave
// are in flight and two workgroups share a CU?  The weight-unit pipeline of csrc/encoder_fused.hip without the arithmetic:
// 14 units of 17 KiB per tile through two LDS buffers, one barrier per unit, unit u+2 issued into u's buffer right behind
// the barrier that ends u.  Every 16-byte chunk of the source image names its unit and its chunk, so a wrong chunk says
// where it came from:
//     STALE      the chunk of unit u-2 (the buffer's previous occupant): the DMA had not landed when it was read
//     MISPLACED  a chunk of the right unit, wrong position / a chunk of another unit
//     LATE       a STALE chunk that was right when read again 4 us later (landed after the barrier)
//     CANARY     bytes outside the two unit buffers changed
// Part C (CROSS): two ds_read_b128 of unit u's buffer are ISSUED in front of the boundary and awaited behind it (what hipcc
// does with the last k-steps of a unit when only vmcnt is waited for in front of the raw s_barrier): do they return unit u
// or the unit u+2 that the DMA behind the barrier brings?  CROSS = 2: s_waitcnt lgkmcnt(0) in front of the barrier.
// Part B: M0 is changed N wait states behind ONE LDS-DMA instruction (nothing else in flight from the wave): does the piece
// land at the old or at the new M0?
//   build: hipcc -O2 --offload-arch=gfx950 lds_dma_race.hip -o lds_dma_race      run: ./lds_dma_race [launches]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int UNIT = 17408;          // 17 pieces of 1 KiB
constexpr int NU = 14;
constexpr int CHUNKS = UNIT / 16;    // 1088
constexpr int LDS_BYTES = 72 * 1024; // two workgroups per CU (160 KiB)
constexpr int WAVES = 4;

struct Rec { unsigned block, it, unit, chunk, f0, f1, f2, f3, again0, again1; };
struct Stats { unsigned stale, late, misplaced, other, canary, nrec, checks, overtaken; };
__device__ Stats g_st;
__device__ Rec g_rec[512];

#define NOP32 "\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7"

// MODE 0: the product's back-to-back form (M0 saved / set / restored per piece, no wait states behind the DMA)
// MODE 1: back to back, M0 set per piece, never restored
// MODE 2: the product's shipped form: vmcnt(0) between a wave's pieces
// MODE 3: one M0 per wave and unit (a wave owns CONTIGUOUS pieces), pieces by the instruction's immediate offset
// MODE 4: MODE 1 + 32 wait states behind every piece
// MODE 5: MODE 3 with the wave's base address per lane (64-bit VGPR pair, saddr = off)
template <int MODE>
__device__ __forceinline__ void dma_unit(const char* __restrict__ src, unsigned lds_dst, int wave, unsigned l16) {
  constexpr int NP = UNIT / 1024;     // 17
  if constexpr (MODE == 3 || MODE == 5) {
    // M0 = base of the wave's first piece + 2048, so that offsets -2048 .. 2048 reach five pieces
    // split: wave 0: 0-4 (5 pieces), wave 1: 5-8, wave 2: 9-12, wave 3: 13-16
    const int first = wave == 0 ? 0 : 4 * wave + 1;
    const unsigned m0v = lds_dst + (unsigned)(first * 1024 + 2048);
    const char* g = src + first * 1024 + 2048;
    if constexpr (MODE == 3) {
      if (wave == 0) {
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 4\n\t"
                     "global_load_lds_dwordx4 %0, %1 offset:-2048\n\tglobal_load_lds_dwordx4 %0, %1 offset:-1024\n\t"
                     "global_load_lds_dwordx4 %0, %1\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024\n\t"
                     "global_load_lds_dwordx4 %0, %1 offset:2048"
                     :: "v"(l16), "s"(g), "s"(m0v) : "memory");
      } else {
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 4\n\t"
                     "global_load_lds_dwordx4 %0, %1 offset:-2048\n\tglobal_load_lds_dwordx4 %0, %1 offset:-1024\n\t"
                     "global_load_lds_dwordx4 %0, %1\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024"
                     :: "v"(l16), "s"(g), "s"(m0v) : "memory");
      }
    } else {
      const char* gl = g + l16;
      if (wave == 0) {
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 4\n\t"
                     "global_load_lds_dwordx4 %0, off offset:-2048\n\tglobal_load_lds_dwordx4 %0, off offset:-1024\n\t"
                     "global_load_lds_dwordx4 %0, off\n\tglobal_load_lds_dwordx4 %0, off offset:1024\n\t"
                     "global_load_lds_dwordx4 %0, off offset:2048"
                     :: "v"(gl), "s"(m0v) : "memory");
      } else {
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 4\n\t"
                     "global_load_lds_dwordx4 %0, off offset:-2048\n\tglobal_load_lds_dwordx4 %0, off offset:-1024\n\t"
                     "global_load_lds_dwordx4 %0, off\n\tglobal_load_lds_dwordx4 %0, off offset:1024"
                     :: "v"(gl), "s"(m0v) : "memory");
      }
    }
    return;
  }
#pragma unroll
  for (int p = 0; p < (NP + WAVES - 1) / WAVES; ++p) {
    const int q = p * WAVES + wave;
    if (q < NP) {
      unsigned keep;
      if constexpr (MODE == 0) {
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(l16), "s"(src + q * 1024), "s"(lds_dst + (unsigned)(q * 1024)) : "memory");
      } else if constexpr (MODE == 1) {
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %0, %1"
                     :: "v"(l16), "s"(src + q * 1024), "s"(lds_dst + (unsigned)(q * 1024)) : "memory");
      } else if constexpr (MODE == 2) {
        if (p > 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_nop 7\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(l16), "s"(src + q * 1024), "s"(lds_dst + (unsigned)(q * 1024)) : "memory");
      } else {
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %0, %1" NOP32
                     :: "v"(l16), "s"(src + q * 1024), "s"(lds_dst + (unsigned)(q * 1024)) : "memory");
      }
    }
  }
}

__device__ __forceinline__ uint4 lds_read(const char* p) {
  asm volatile("" ::: "memory");
  const uint4 v = *reinterpret_cast<const uint4*>(p);
  asm volatile("" ::: "memory");
  return v;
}

__device__ __forceinline__ void record(int it, int u, int c, uint4 v, uint4 again, int cls) {
  if (cls == 0) atomicAdd(&g_st.stale, 1u);
  else if (cls == 1) atomicAdd(&g_st.late, 1u);
  else if (cls == 2) atomicAdd(&g_st.misplaced, 1u);
  else atomicAdd(&g_st.other, 1u);
  const unsigned i = atomicAdd(&g_st.nrec, 1u);
  if (i < 512) g_rec[i] = Rec{blockIdx.x, (unsigned)it, (unsigned)u, (unsigned)c, v.x, v.y, v.z, v.w, again.x, again.y};
}

// STORES: 16-byte global stores per thread and unit that the boundary does NOT wait for (counted vmcnt), as the product does
template <int MODE, int STORES, int CROSS = 0>
__global__ void __launch_bounds__(256, 2) k_pipe(const char* __restrict__ img, uint4* __restrict__ sink, int n_it, int verify_all) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned l16 = 16u * lane;
  const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;
  // canary behind the two unit buffers
  for (int i = 2 * UNIT + 16 * tid; i < LDS_BYTES; i += 16 * 256)
    *reinterpret_cast<uint4*>(smem + i) = make_uint4(0xC0FFEE00u, (unsigned)i, 0u, 0u);
  __syncthreads();
  dma_unit<MODE>(img, lds0, wave, l16);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  dma_unit<MODE>(img + UNIT, lds0 + UNIT, wave, l16);
  uint4* mysink = sink + ((size_t)blockIdx.x * 256 + tid) * 8;
  unsigned nchk = 0;
  for (int it = 0; it < n_it; ++it) {
#pragma unroll 1
    for (int u = 0; u < NU; ++u) {
      const char* buf = smem + (u & 1) * UNIT;
      // "compute" on unit u: every chunk must name unit u
      for (int c = tid; c < CHUNKS; c += 256) {
        const uint4 v = lds_read(buf + 16 * c);
        ++nchk;
        if (v.x != (0xA5000000u | (unsigned)u) || v.y != (unsigned)c) {
          // read again 4 us later: landed late, or wrong for good?
          for (int s = 0; s < 8; ++s) __builtin_amdgcn_s_sleep(127);
          const uint4 w = lds_read(buf + 16 * c);
          const bool ok_now = w.x == (0xA5000000u | (unsigned)u) && w.y == (unsigned)c;
          const unsigned prev = 0xA5000000u | (unsigned)((u + NU - 2) % NU);
          int cls = 3;
          if (v.x == prev && v.y == (unsigned)c) cls = ok_now ? 1 : 0;
          else if ((v.x >> 24) == 0xA5u) cls = 2;
          record(it, u, c, v, w, cls);
        }
      }
      if (STORES) {
#pragma unroll
        for (int s = 0; s < STORES; ++s) mysink[s] = make_uint4(nchk, (unsigned)u, (unsigned)s, 0u);
      }
      // boundary at the end of unit u: unit u+1 landed (counted: the STORES issued behind its DMA may stay in flight), barrier,
      // unit u+2 into u's buffer
      u32x4 q0 = {0u, 0u, 0u, 0u}, q1 = q0, d0 = q0, d1 = q0, d2 = q0, d3 = q0;      // (d*: kept allocated until the reads have returned)
      if constexpr (CROSS != 0) {
        const unsigned a0 = lds0 + (unsigned)((u & 1) * UNIT) + 16u * (unsigned)tid;      // chunks tid and tid + 544
        // queue pressure, as the weight-fragment bursts of the product's two workgroups give it: 32 reads of this buffer that
        // nobody waits for stand in the wave's LDS queue in front of the two checked ones
        asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\tds_read_b128 %3, %4 offset:3072\n\t"
                     "ds_read_b128 %0, %4 offset:4096\n\tds_read_b128 %1, %4 offset:5120\n\tds_read_b128 %2, %4 offset:6144\n\tds_read_b128 %3, %4 offset:7168\n\t"
                     "ds_read_b128 %0, %4 offset:8192\n\tds_read_b128 %1, %4 offset:9216\n\tds_read_b128 %2, %4 offset:10240\n\tds_read_b128 %3, %4 offset:11264\n\t"
                     "ds_read_b128 %0, %4 offset:12288\n\tds_read_b128 %1, %4 offset:1040\n\tds_read_b128 %2, %4 offset:2064\n\tds_read_b128 %3, %4 offset:3088\n\t"
                     "ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\tds_read_b128 %3, %4 offset:3072\n\t"
                     "ds_read_b128 %0, %4 offset:4096\n\tds_read_b128 %1, %4 offset:5120\n\tds_read_b128 %2, %4 offset:6144\n\tds_read_b128 %3, %4 offset:7168\n\t"
                     "ds_read_b128 %0, %4 offset:8192\n\tds_read_b128 %1, %4 offset:9216\n\tds_read_b128 %2, %4 offset:10240\n\tds_read_b128 %3, %4 offset:11264\n\t"
                     "ds_read_b128 %0, %4 offset:12288\n\tds_read_b128 %1, %4 offset:1040\n\tds_read_b128 %2, %4 offset:2064\n\tds_read_b128 %3, %4 offset:3088"
                     : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a0) : "memory");
        asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:8704" : "=&v"(q0), "=&v"(q1) : "v"(a0) : "memory");
      }
      if constexpr (CROSS == 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(STORES) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(STORES) : "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      const bool more = u + 2 < NU || it + 1 < n_it;
      if (more) dma_unit<MODE>(img + (size_t)((u + 2) % NU) * UNIT, lds0 + (unsigned)((u & 1) * UNIT), wave, l16);
      if constexpr (CROSS != 0) {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(q0), "+v"(q1), "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) :: "memory");
        const uint4 r0 = make_uint4(q0.x, q0.y, q0.z, q0.w), r1 = make_uint4(q1.x, q1.y, q1.z, q1.w);
        const unsigned want = 0xA5000000u | (unsigned)u, nxt = 0xA5000000u | (unsigned)((u + 2) % NU);
        if (r0.x != want || r0.y != (unsigned)tid) {
          if (r0.x == nxt && more) atomicAdd(&g_st.overtaken, 1u);
          record(it, u, tid, r0, r1, 3);
        }
        if (r1.x != want || r1.y != (unsigned)(tid + 544)) {
          if (r1.x == nxt && more) atomicAdd(&g_st.overtaken, 1u);
          record(it, u, tid + 544, r1, r0, 3);
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = 2 * UNIT + 16 * tid; i < LDS_BYTES; i += 16 * 256) {
    const uint4 v = *reinterpret_cast<const uint4*>(smem + i);
    if (v.x != 0xC0FFEE00u || v.y != (unsigned)i) {
      atomicAdd(&g_st.canary, 1u);
      const unsigned k = atomicAdd(&g_st.nrec, 1u);
      if (k < 512) g_rec[k] = Rec{blockIdx.x, 0xffffffffu, 0xffffffffu, (unsigned)i, v.x, v.y, v.z, v.w, 0u, 0u};
    }
  }
  if (tid == 0) atomicAdd(&g_st.checks, nchk);
}

// Part B: one piece with M0 = A, M0 := B after WS wait states (B = A + 1024, canary there).  LOADS: plain loads issued in
// front to back the vector-memory queue up.
struct StatsB { unsigned at_a, at_b, neither, both, pad[4]; };
__device__ StatsB g_b;
template <int WS, int LOADS>
__global__ void __launch_bounds__(256, 2) k_m0(const char* __restrict__ img, const uint4* __restrict__ junk, uint4* __restrict__ sink, int n_it) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned l16 = 16u * lane;
  const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;
  char* A = smem + wave * 2048;
  uint4 acc = make_uint4(0u, 0u, 0u, 0u);
  for (int it = 0; it < n_it; ++it) {
    *reinterpret_cast<uint4*>(A + l16) = make_uint4(0xC0FFEE00u, 1u, 0u, 0u);
    *reinterpret_cast<uint4*>(A + 1024 + l16) = make_uint4(0xC0FFEE00u, 2u, 0u, 0u);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    uint4 t[LOADS > 0 ? LOADS : 1];
#pragma unroll
    for (int s = 0; s < LOADS; ++s) t[s] = junk[((size_t)(blockIdx.x * 131 + it * 17 + s * 7919) * 256 + tid) & 0xfffff];
    const unsigned a0 = lds0 + (unsigned)(wave * 2048);
    if constexpr (WS == 0)
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %0, %1\n\ts_mov_b32 m0, %3"
                   :: "v"(l16), "s"(img + (it % 13) * 1024), "s"(a0), "s"(a0 + 1024u) : "memory");
    else if constexpr (WS == 8)
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %0, %1\n\ts_nop 7\n\ts_mov_b32 m0, %3"
                   :: "v"(l16), "s"(img + (it % 13) * 1024), "s"(a0), "s"(a0 + 1024u) : "memory");
    else
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %0, %1" NOP32 "\n\ts_mov_b32 m0, %3"
                   :: "v"(l16), "s"(img + (it % 13) * 1024), "s"(a0), "s"(a0 + 1024u) : "memory");
#pragma unroll
    for (int s = 0; s < LOADS; ++s) { acc.x += t[s].x; acc.y ^= t[s].y; }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const uint4 va = lds_read(A + l16);
    const uint4 vb = lds_read(A + 1024 + l16);
    const bool da = (va.x >> 24) == 0xA5u, db = (vb.x >> 24) == 0xA5u;
    if (da && !db) atomicAdd(&g_b.at_a, 1u);
    else if (!da && db) atomicAdd(&g_b.at_b, 1u);
    else if (!da && !db) atomicAdd(&g_b.neither, 1u);
    else atomicAdd(&g_b.both, 1u);
    __builtin_amdgcn_s_barrier();
  }
  if (acc.x == 0x12345u) sink[tid] = acc;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int MODE, int STORES, int CROSS = 0>
static void run_pipe(const char* img, uint4* sink, int grid, int n_it, int launches, const char* what) {
  Stats z{};
  CK(hipMemcpyToSymbol(HIP_SYMBOL(g_st), &z, sizeof z));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pipe<MODE, STORES, CROSS>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0));
  for (int l = 0; l < launches; ++l) hipLaunchKernelGGL((k_pipe<MODE, STORES, CROSS>), dim3(grid), dim3(256), LDS_BYTES, 0, img, sink, n_it, 0);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  Stats s;
  CK(hipMemcpyFromSymbol(&s, HIP_SYMBOL(g_st), sizeof s));
  printf("pipe mode %d stores %d cross %d grid %4d n_it %3d launches %5d: stale %u late %u misplaced %u other %u canary %u OVERTAKEN %u  (%.1f us/launch)  %s\n",
         MODE, STORES, CROSS, grid, n_it, launches, s.stale, s.late, s.misplaced, s.other, s.canary, s.overtaken, 1000.f * ms / launches, what);
  if (s.nrec) {
    std::vector<Rec> r(512);
    CK(hipMemcpyFromSymbol(r.data(), HIP_SYMBOL(g_rec), sizeof(Rec) * 512));
    const unsigned n = s.nrec < 24 ? s.nrec : 24;
    for (unsigned i = 0; i < n; ++i)
      printf("    block %u it %u unit %u chunk %u (piece %u, wave %u): found %08x %u %08x %08x, again %08x %u\n", r[i].block, r[i].it,
             r[i].unit, r[i].chunk, r[i].chunk / 64, (r[i].chunk / 64) % 4, r[i].f0, r[i].f1, r[i].f2, r[i].f3, r[i].again0, r[i].again1);
  }
  fflush(stdout);
}

template <int WS, int LOADS>
static void run_m0(const char* img, const uint4* junk, uint4* sink, int grid, int n_it, int launches) {
  StatsB z{};
  CK(hipMemcpyToSymbol(HIP_SYMBOL(g_b), &z, sizeof z));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_m0<WS, LOADS>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
  for (int l = 0; l < launches; ++l) hipLaunchKernelGGL((k_m0<WS, LOADS>), dim3(grid), dim3(256), LDS_BYTES, 0, img, junk, sink, n_it);
  CK(hipDeviceSynchronize());
  StatsB s;
  CK(hipMemcpyFromSymbol(&s, HIP_SYMBOL(g_b), sizeof s));
  printf("m0 change %2d wait states behind the DMA, %2d loads in front, grid %d: landed at old M0 %u, at NEW M0 %u, nowhere %u, both %u\n",
         WS, LOADS, grid, s.at_a, s.at_b, s.neither, s.both);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const int launches = argc > 1 ? atoi(argv[1]) : 2000;
  std::vector<unsigned> h((size_t)NU * UNIT / 4);
  for (int u = 0; u < NU; ++u)
    for (int c = 0; c < CHUNKS; ++c) {
      unsigned* p = h.data() + ((size_t)u * UNIT + 16 * c) / 4;
      p[0] = 0xA5000000u | (unsigned)u; p[1] = (unsigned)c; p[2] = (unsigned)(u * 4096 + c) * 2654435761u; p[3] = ~(unsigned)c;
    }
  char* img; uint4 *sink, *junk;
  CK(hipMalloc(&img, h.size() * 4));
  CK(hipMemcpy(img, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&sink, (size_t)1024 * 256 * 8 * 16));
  CK(hipMalloc(&junk, (size_t)(1 << 20) * 16));
  CK(hipMemset(junk, 1, (size_t)(1 << 20) * 16));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("device %s, %d CUs\n", prop.name, cus);

  // Part B first (cheap, decisive for the late-M0 question)
  run_m0<0, 0>(img, junk, sink, 2 * cus, 200, 20);
  run_m0<0, 16>(img, junk, sink, 2 * cus, 200, 20);
  run_m0<8, 16>(img, junk, sink, 2 * cus, 200, 20);
  run_m0<32, 16>(img, junk, sink, 2 * cus, 200, 20);

  // Part C: reads that cross the boundary
  run_pipe<0, 8, 1>(img, sink, 2 * cus, 2, launches, "reads cross the barrier, back-to-back DMA, 2 wg/CU");
  run_pipe<0, 8, 2>(img, sink, 2 * cus, 2, launches, "lgkmcnt(0) in front of the barrier, back-to-back DMA, 2 wg/CU");
  run_pipe<2, 8, 1>(img, sink, 2 * cus, 2, launches, "reads cross the barrier, serial DMA (round-4 product), 2 wg/CU");
  run_pipe<4, 8, 1>(img, sink, 2 * cus, 2, launches, "reads cross the barrier, 32 wait states per piece, 2 wg/CU");
  run_pipe<0, 8, 1>(img, sink, cus, 2, launches, "reads cross the barrier, back-to-back DMA, 1 wg/CU");
  run_pipe<0, 8, 1>(img, sink, 2 * cus, 60, launches / 20, "reads cross the barrier, back-to-back DMA, long");
  run_pipe<0, 8, 2>(img, sink, 2 * cus, 60, launches / 20, "lgkmcnt(0) in front of the barrier, long");
  // Part A: short launches (start-up, as R = 13 000 in the product: 1-2 tiles per workgroup) and long ones
  run_pipe<0, 8>(img, sink, 2 * cus, 2, launches, "product back-to-back, 2 wg/CU");
  run_pipe<0, 0>(img, sink, 2 * cus, 2, launches, "product back-to-back, vmcnt(0) boundaries");
  run_pipe<1, 8>(img, sink, 2 * cus, 2, launches, "no M0 restore");
  run_pipe<4, 8>(img, sink, 2 * cus, 2, launches, "32 wait states behind every piece");
  run_pipe<2, 8>(img, sink, 2 * cus, 2, launches, "serial pieces (shipped)");
  run_pipe<3, 8>(img, sink, 2 * cus, 2, launches, "one M0 per wave and unit, immediate offsets");
  run_pipe<5, 8>(img, sink, 2 * cus, 2, launches, "same, per-lane 64-bit addresses");
  run_pipe<0, 8>(img, sink, cus, 2, launches, "product back-to-back, 1 wg/CU");
  run_pipe<0, 8>(img, sink, 2 * cus, 60, launches / 20, "product back-to-back, long");
  run_pipe<3, 8>(img, sink, 2 * cus, 60, launches / 20, "one M0 per wave and unit, long");
  run_pipe<2, 8>(img, sink, 2 * cus, 60, launches / 20, "serial, long");
  return 0;
}
