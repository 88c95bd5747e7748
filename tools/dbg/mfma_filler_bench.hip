// What does an instruction cost BESIDE v_mfma_f32_32x32x16 on gfx950?  One kernel: every wave runs ITER x 16 MFMAs with F fillers of
// kind K behind each MFMA; the MFMAs either accumulate into ONE accumulator (the chains' form: 16 dependent MFMAs per block) or
// alternate between TWO.  Waves per SIMD: 1 (256 threads) or 2 (512 threads).  Output: cycles per MFMA per SIMD at the measured wall
// time (clock from a calibrated bare-MFMA run is not assumed: the table prints ns per MFMA-slot and the ratio to the bare stream).
// build: hipcc --offload-arch=gfx950 -O3 tools/dbg/mfma_filler_bench.hip -o tools/dbg/libs/mfma_filler_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

enum { K_NONE = 0, K_VADD, K_VCMP, K_CVT, K_PKMAX, K_FP8, K_CNDMASK, K_DSREAD, K_SNOP, K_SMOV, K_DSWRITE, K_MIXF, K_MIXB, K_NKINDS };
static const char* kname[] = {"none", "v_add_f32", "v_cmp_lt_f32 -> sgpr", "v_cvt_pk_f16_f32", "v_pk_max_i16", "v_cvt_scalef32_pk_fp8_f16", "v_cndmask (sgpr mask)",
                              "ds_read_b128", "s_nop 0", "s_mov_b32", "ds_write_b32", "fwd-chain mix (cmp,cvt,max,fp8,ds_read ..)", "bwd-chain mix (cndmask x2, cvt_bf16, ds_read)"};

template <int K>
__device__ __forceinline__ void filler(int i, float (&x)[8], unsigned (&u)[8], unsigned long long& m, unsigned lp, u32x4 (&fr)[4]) {
  float& a = x[i & 7];
  unsigned& b = u[i & 7];
  if constexpr (K == K_VADD) asm volatile("v_add_f32 %0, %0, %0" : "+v"(a));
  if constexpr (K == K_VCMP) asm volatile("v_cmp_lt_f32_e64 %0, 0, %1" : "=s"(m) : "v"(a));
  if constexpr (K == K_CVT) asm volatile("v_cvt_pk_f16_f32 %0, %1, %1" : "=v"(b) : "v"(a));
  if constexpr (K == K_PKMAX) asm volatile("v_pk_max_i16 %0, %0, 0" : "+v"(b));
  if constexpr (K == K_FP8) asm volatile("v_cvt_scalef32_pk_fp8_f16 %0, %1, 1.0" : "+v"(b) : "v"(u[(i + 1) & 7]));
  if constexpr (K == K_CNDMASK) asm volatile("v_cndmask_b32_e64 %0, 0, %0, %1" : "+v"(a) : "s"(m));
  if constexpr (K == K_DSREAD) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fr[i & 3]) : "v"(lp), "n"(1024) : "memory");
  if constexpr (K == K_SNOP) asm volatile("s_nop 0");
  if constexpr (K == K_SMOV) { unsigned t; asm volatile("s_mov_b32 %0, 0" : "=s"(t)); }
  if constexpr (K == K_DSWRITE) asm volatile("ds_write_b32 %0, %1 offset:%2" : : "v"(lp), "v"(b), "n"(2048) : "memory");
}

// F fillers of kind K behind MFMA number c of the block (mixes: a fixed recipe per gap, F ignored)
template <int K, int F>
__device__ __forceinline__ void gap(int c, float (&x)[8], unsigned (&u)[8], unsigned long long& m, unsigned lp, u32x4 (&fr)[4]) {
  if constexpr (K == K_MIXF) {
    // forward chain, per 4 gaps: [ds_read + 4 cmp] [ds_read + 2 cvt + 2 max] [ds_read + 2 fp8] [ds_read + ds_write]
    filler<K_DSREAD>(c, x, u, m, lp, fr);
    switch (c & 3) {
      case 0: for (int i = 0; i < 4; ++i) filler<K_VCMP>(i, x, u, m, lp, fr); break;
      case 1: filler<K_CVT>(0, x, u, m, lp, fr); filler<K_PKMAX>(0, x, u, m, lp, fr); filler<K_CVT>(1, x, u, m, lp, fr); filler<K_PKMAX>(1, x, u, m, lp, fr); break;
      case 2: filler<K_FP8>(0, x, u, m, lp, fr); filler<K_FP8>(1, x, u, m, lp, fr); break;
      default: filler<K_DSWRITE>(0, x, u, m, lp, fr); break;
    }
  } else if constexpr (K == K_MIXB) {
    filler<K_DSREAD>(c, x, u, m, lp, fr);
    switch (c % 3) {
      case 0: for (int i = 0; i < 4; ++i) filler<K_CNDMASK>(i, x, u, m, lp, fr); break;
      case 1: asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[0]) : "v"(x[0]), "v"(x[1])); asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[1]) : "v"(x[2]), "v"(x[3])); break;
      default: filler<K_DSWRITE>(0, x, u, m, lp, fr); break;
    }
  } else {
#pragma unroll
    for (int i = 0; i < F; ++i) filler<K>(c * F + i, x, u, m, lp, fr);
  }
}

template <int K, int F, int NACC>
__global__ __launch_bounds__(512) void bench(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 65536 / 4; i += blockDim.x) reinterpret_cast<unsigned*>(lds)[i] = i;
  __syncthreads();
  const unsigned lp = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds + (threadIdx.x >> 6) * 4096 + lane * 16;
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (lane + i)); b[i] = (_Float16)(0.002f * (lane - i)); }
  f32x16 acc[2];
  for (int i = 0; i < 16; ++i) acc[0][i] = acc[1][i] = 0.f;
  float x[8];
  unsigned u[8];
  u32x4 fr[4] = {};
  for (int i = 0; i < 8; ++i) { x[i] = 1.f + lane + i; u[i] = lane * 77 + i; }
  unsigned long long m = 0x5555555555555555ull;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      acc[NACC == 2 ? (c & 1) : 0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[NACC == 2 ? (c & 1) : 0], 0, 0, 0);
      gap<K, F>(c, x, u, m, lp, fr);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += acc[0][i] + acc[1][i];
  for (int i = 0; i < 8; ++i) s += x[i] + (float)u[i];
  for (int i = 0; i < 4; ++i) s += (float)fr[i][0];
  s += (float)(m & 1);
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int K, int F, int NACC>
static double run(int threads, float* d_out, int iters) {
  std::vector<float> ts;
  for (int r = 0; r < 7; ++r) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    bench<K, F, NACC><<<256, threads>>>(d_out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (r >= 2) ts.push_back(ms);
    hipEventDestroy(e0); hipEventDestroy(e1);
  }
  std::sort(ts.begin(), ts.end());
  return ts[ts.size() / 2];
}

template <int K, int F, int NACC>
static void report(float* d_out, int iters, double base1, double base2) {
  const double t1 = run<K, F, NACC>(256, d_out, iters), t2 = run<K, F, NACC>(512, d_out, iters);
  // MFMA slots per SIMD: 1 wave -> 16 iters; 2 waves -> 32 iters
  printf("%-46s F=%d acc=%d | 1 wave/SIMD %7.1f ns/MFMA (x%.2f) | 2 waves/SIMD %7.1f ns/MFMA-slot (x%.2f)\n", kname[K], F, NACC,
         t1 * 1e6 / (16.0 * iters), base1 > 0 ? t1 / base1 : 1.0, t2 * 1e6 / (32.0 * iters), base2 > 0 ? t2 / base2 : 1.0);
}

int main() {
  const int iters = 2000;
  float* d_out;
  hipMalloc(&d_out, 256 * 512 * 4);
  const double b1 = run<K_NONE, 0, 1>(256, d_out, iters), b2 = run<K_NONE, 0, 1>(512, d_out, iters);
  printf("bare dependent MFMAs: 1 wave/SIMD %.1f ns/MFMA, 2 waves/SIMD %.1f ns/MFMA-slot (32 cycles at 2.4 GHz = 13.3 ns)\n", b1 * 1e6 / (16.0 * iters), b2 * 1e6 / (32.0 * iters));
  report<K_NONE, 0, 2>(d_out, iters, b1, b2);
#define ROW(K) report<K, 1, 1>(d_out, iters, b1, b2); report<K, 2, 1>(d_out, iters, b1, b2); report<K, 4, 1>(d_out, iters, b1, b2); report<K, 8, 1>(d_out, iters, b1, b2); report<K, 4, 2>(d_out, iters, b1, b2);
  ROW(K_VADD) ROW(K_VCMP) ROW(K_CVT) ROW(K_PKMAX) ROW(K_FP8) ROW(K_CNDMASK) ROW(K_DSREAD) ROW(K_SNOP) ROW(K_SMOV) ROW(K_DSWRITE)
  report<K_MIXF, 0, 1>(d_out, iters, b1, b2); report<K_MIXF, 0, 2>(d_out, iters, b1, b2);
  report<K_MIXB, 0, 1>(d_out, iters, b1, b2); report<K_MIXB, 0, 2>(d_out, iters, b1, b2);
  return 0;
}
