// Does the instruction offset of global_load_lds_dwordx4 move the LDS destination too (as MUBUF ... lds does), or only the
// global address?  And: is one wait state enough between an SALU write of M0 and the LDS-DMA that reads it?
// build: hipcc --offload-arch=gfx950 -O3 tools/dbg/ldsdma_offset_probe.hip -o tools/dbg/libs/ldsdma_offset_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const unsigned* src, unsigned* out) {
  __shared__ __attribute__((aligned(16))) unsigned lds[2048];          // 8 KB
  const int lane = threadIdx.x;
  for (int i = lane; i < 2048; i += 64) lds[i] = 0xdeadbeefu;
  __syncthreads();
  const unsigned voff = lane * 16;
  const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned*)lds;
  // piece 0 -> LDS bytes [0, 1024); piece 1 (offset:1024) -> [1024, 2048) if the offset applies to the LDS side too, else [0, 1024) again
  asm volatile("s_add_i32 m0, %1, 0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2\n\tglobal_load_lds_dwordx4 %0, %2 offset:1024\n\ts_waitcnt vmcnt(0)"
               : : "v"(voff), "s"(base), "s"(src) : "memory");
  __syncthreads();
  for (int i = lane; i < 2048; i += 64) out[i] = lds[i];
}
int main() {
  unsigned h[2048], o[2048];
  for (int i = 0; i < 2048; ++i) h[i] = i;
  unsigned *d, *r;
  hipMalloc(&d, sizeof(h)); hipMalloc(&r, sizeof(o));
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  k<<<1, 64>>>(d, r);
  hipMemcpy(o, r, sizeof(o), hipMemcpyDeviceToHost);
  printf("lds dword 0: %u  dword 255: %u  dword 256: %x  dword 511: %x  dword 512: %x\n", o[0], o[255], o[256], o[511], o[512]);
  bool both = true, only_global = true;
  for (int i = 0; i < 512; ++i) both &= o[i] == (unsigned)i;
  for (int i = 0; i < 256; ++i) only_global &= o[i] == (unsigned)(i + 256);
  printf("%s\n", both ? "offset applies to BOTH the global and the LDS address" : only_global ? "offset applies to the GLOBAL address only" : "neither pattern");
  return 0;
}
