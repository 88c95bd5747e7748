// Does MODE.FP16_OVFL (bit 23) make v_cvt_pk_f16_f32 and v_cvt_scalef32_pk_fp8_f16 SATURATE on gfx950?
// build: hipcc --offload-arch=gfx950 -O3 tools/dbg/ovfl_probe.hip -o tools/dbg/libs/ovfl_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(2))) short s16x2;
__global__ void k(const float* in, unsigned* o16, unsigned* o8, int ovfl) {
  if (ovfl) __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1);
  const int t = threadIdx.x;
  const f32x2 f = {in[2 * t], in[2 * t + 1]};
  const f16x2 h = __builtin_convertvector(f, f16x2);
  o16[t] = __builtin_bit_cast(unsigned, h);
  s16x2 r = {0, 0};
  r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(r, h, 1.0f, false);
  o8[t] = (unsigned)__builtin_bit_cast(unsigned, r) & 0xffffu;
}
int main() {
  float h_in[8] = {1.0f, 448.0f, 449.0f, 1000.0f, 65504.0f, 65600.0f, 1e6f, -1e6f};
  float* d_in; unsigned *d16, *d8;
  hipMalloc(&d_in, sizeof(h_in)); hipMalloc(&d16, 16); hipMalloc(&d8, 16);
  hipMemcpy(d_in, h_in, sizeof(h_in), hipMemcpyHostToDevice);
  for (int ovfl = 0; ovfl < 2; ++ovfl) {
    k<<<1, 4>>>(d_in, d16, d8, ovfl);
    unsigned a[4], b[4];
    hipMemcpy(a, d16, 16, hipMemcpyDeviceToHost); hipMemcpy(b, d8, 16, hipMemcpyDeviceToHost);
    printf("FP16_OVFL=%d:", ovfl);
    for (int i = 0; i < 4; ++i) printf("  f16 pair %08x fp8 pair %04x |", a[i], b[i]);
    printf("\n");
  }
  return 0;
}
