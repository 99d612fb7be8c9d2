// Does a 128-bit buffer store still read its data VGPRs after it has issued?  hipcc pads "store of more than 64 bits ->
// write of its data registers" with wait states only when the store has NO register soffset (GCNHazardRecognizer:
// "this hazard only exists if the instruction is not using a register in the soffset field").  Each variant stores a
// known pattern and overwrites the third data register N instructions later; wrong dwords in memory = the store read
// the register after the overwrite.  Many waves per CU store at once, as in the column-transformer kernels.
//   build: hipcc -O2 --offload-arch=gfx950 store_data_hazard.hip -o store_data_hazard
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// VAR 0: register soffset, overwrite directly behind the store      VAR 4 / 1: s_nop 0 / s_nop 1 in between
// VAR 2: soffset = 0 (immediate), overwrite directly behind          VAR 3: s_nop 1 in between
template <int VAR, int NF>
__global__ void __launch_bounds__(256, 2) k_haz(unsigned* __restrict__ out, int n_it, unsigned bytes_per_block) {
  const unsigned tid = threadIdx.x;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(out + (size_t)blockIdx.x * (bytes_per_block / 4), 0, (int)bytes_per_block, 0x00020000);
  const unsigned soff = __builtin_amdgcn_readfirstlane(128u + (tid >> 20));
  for (int it = 0; it < n_it; ++it) {
    const unsigned voff = (unsigned)(it & 3) * 16384u + 16u * tid;
    const unsigned voffi = voff + 128u;                       // the immediate-soffset variants carry the 128 in voffset
    const unsigned vfill = voff + 8192u;                      // filler stores: bytes 8192 .. 16383 of the window
    u32x4 v = {0xA0000000u | tid, 0xA1000000u | tid, 0xA2000000u | tid, 0xA3000000u | tid};
    u32x4 w = {0xB0000000u | tid, 0xB1000000u | tid, 0xB2000000u | tid, 0xB3000000u | tid};
    // three filler stores (w) keep the path busy, then the checked store of v[20:23] and the overwrite of v22 / v23
#define HAZ(SOFF, VO, GAP)                                                                              \
      asm volatile("v_mov_b32 v20, %0\n\tv_mov_b32 v21, %1\n\tv_mov_b32 v22, %2\n\tv_mov_b32 v23, %3\n\ts_nop 4\n\t"          \
                   "buffer_store_dwordx4 %4, %9, %7, " SOFF " offen\n\tbuffer_store_dwordx4 %4, %9, %7, " SOFF " offen offset:2048\n\t" \
                   "buffer_store_dwordx4 %4, %9, %7, " SOFF " offen offset:3968\n\t"                  \
                   "buffer_store_dwordx4 v[20:23], " VO ", %7, " SOFF " offen\n\t" GAP                 \
                   "v_mov_b32 v22, 0\n\tv_mov_b32 v23, 0"                                              \
                   :: "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w), "v"(w), "v"(voff), "v"(voffi), "s"(rs), "s"(soff), "v"(vfill)   \
                   : "memory", "v20", "v21", "v22", "v23")
    // NF more filler stores in front (the product's row store is the eighth wide store of its wave in a row)
    for (int f = 0; f < NF; ++f) {
      if constexpr (VAR == 2 || VAR == 3) asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen offset:1024" :: "v"(w), "v"(vfill), "s"(rs) : "memory");
      else asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen offset:1024" :: "v"(w), "v"(vfill), "s"(rs), "s"(soff) : "memory");
    }
    if constexpr (VAR == 0) HAZ("%8", "%5", "");
    else if constexpr (VAR == 4) HAZ("%8", "%5", "s_nop 0\n\t");
    else if constexpr (VAR == 1) HAZ("%8", "%5", "s_nop 1\n\t");
    else if constexpr (VAR == 2) HAZ("0", "%6", "");
    else HAZ("0", "%6", "s_nop 1\n\t");
#undef HAZ
  }
}

template <int VAR, int NF = 0>
static void run(unsigned* out, std::vector<unsigned>& h, int grid, int launches, const char* what) {
  const unsigned bpb = 4 * 16384u + 16384u;     // four 16 KiB windows of checked rows (+128 B), filler windows behind
  size_t words = (size_t)grid * bpb / 4;
  unsigned long long bad = 0, seen = 0, badreg[4] = {0, 0, 0, 0}, badlane[64] = {0};
  for (int l = 0; l < launches; ++l) {
    CK(hipMemsetAsync(out, 0, words * 4));
    hipLaunchKernelGGL((k_haz<VAR, NF>), dim3(grid), dim3(256), 0, 0, out, 64, bpb);
    CK(hipMemcpy(h.data(), out, words * 4, hipMemcpyDeviceToHost));
    for (int b = 0; b < grid; ++b)
      for (int w = 0; w < 4; ++w)
        for (unsigned t = 0; t < 256; ++t) {
          const unsigned* p = h.data() + (size_t)b * (bpb / 4) + (w * 16384u + 128u + 16u * t) / 4;
          for (int d = 0; d < 4; ++d) {
            ++seen;
            if (p[d] != ((0xA0000000u + 0x01000000u * d) | t)) { ++bad; ++badreg[d]; ++badlane[t & 63]; }
          }
        }
  }
  printf("%-58s: %llu wrong dwords of %llu  (by data register: %llu %llu %llu %llu)", what, bad, seen, badreg[0], badreg[1], badreg[2], badreg[3]);
  if (bad) { printf("  lanes:"); for (int i = 0; i < 64; ++i) if (badlane[i]) printf(" %d:%llu", i, badlane[i]); }
  printf("\n");
  fflush(stdout);
}

int main(int argc, char** argv) {
  const int launches = argc > 1 ? atoi(argv[1]) : 20;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int grid = 2 * prop.multiProcessorCount * 2;
  const unsigned bpb = 4 * 16384u + 16384u;
  unsigned* out;
  CK(hipMalloc(&out, (size_t)grid * bpb));
  std::vector<unsigned> h((size_t)grid * bpb / 4);
  run<0>(out, h, grid, launches, "register soffset, data overwritten directly behind");
  run<4>(out, h, grid, launches, "register soffset, s_nop 0 in between");
  run<1>(out, h, grid, launches, "register soffset, s_nop 1 in between");
  run<2>(out, h, grid, launches, "immediate soffset, data overwritten directly behind");
  run<3>(out, h, grid, launches, "immediate soffset, s_nop 1 in between");
  run<0, 4>(out, h, grid, launches, "register soffset, directly behind, 7 stores in front");
  run<4, 4>(out, h, grid, launches, "register soffset, s_nop 0, 7 stores in front");
  run<1, 4>(out, h, grid, launches, "register soffset, s_nop 1, 7 stores in front");
  run<3, 4>(out, h, grid, launches, "immediate soffset, s_nop 1, 7 stores in front");
  run<0, 12>(out, h, grid, launches, "register soffset, directly behind, 15 stores in front");
  run<3, 12>(out, h, grid, launches, "immediate soffset, s_nop 1, 15 stores in front");
  return 0;
}
