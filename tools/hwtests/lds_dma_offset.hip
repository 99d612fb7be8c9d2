// Where does global_load_lds_dwordx4 with an instruction offset put its data?  (M0 = LDS base, voffset = 16 * lane)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const unsigned* __restrict__ src, unsigned* __restrict__ dump) {
  __shared__ __attribute__((aligned(16))) unsigned lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = 0xdeadbeefu;
  __syncthreads();
  const unsigned l16 = 16u * threadIdx.x;
  const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned*)lds;
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %0, %1 offset:2048\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\t"
               "global_load_lds_dwordx4 %0, %1 offset:-1024\n\ts_waitcnt vmcnt(0)"
               :: "v"(l16), "s"(src + 1024 /* bytes 4096 */), "s"(lds0 + 4096u) : "memory", "m0");
  __syncthreads();
  for (int i = threadIdx.x; i < 4096; i += 64) dump[i] = lds[i];
}
int main() {
  std::vector<unsigned> h(8192);
  for (int i = 0; i < 8192; ++i) h[i] = i;
  unsigned *s, *d;
  hipMalloc(&s, 8192 * 4); hipMalloc(&d, 4096 * 4);
  hipMemcpy(s, h.data(), 8192 * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, s, d);
  std::vector<unsigned> o(4096);
  hipMemcpy(o.data(), d, 4096 * 4, hipMemcpyDeviceToHost);
  int first = -1;
  for (int i = 0; i < 4096; ++i) {
    bool w = o[i] != 0xdeadbeefu;
    if (w && first < 0) first = i;
    if (!w && first >= 0) { printf("LDS dwords [%d, %d) <- src dwords [%u, %u]\n", first, i, o[first], o[i - 1]); first = -1; }
  }
  // expected if the offset applies to BOTH addresses: M0 = 4096 B -> dword 1024; offset 2048 B -> LDS dwords [1536, 1792) <- src dwords 1024 + 512 .. ;
  // offset -1024 B -> LDS dwords [768, 1024) <- src dwords 1024 - 256 ..
  return 0;
}
