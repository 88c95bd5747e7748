// BatchNorm (training) + ReLU over the conv stack's big layers (NCHW fp32; layer 1: 64 x 24 x 64 x 64 = 25 MB, layer 2: 64 x 24 x 32 x 32):
// today TWO launches -- per-slice fp64 partial sums (cn_stats_kernel), then normalise + ReLU from the partials (cn_apply_kernel:
// a second read of x) -- restated here as (A); against (B) ONE launch in which workgroup (slice s, channel c) keeps its 8192
// (layer 2: 2048) elements in REGISTERS, publishes its partial, waits inside the launch for the other S - 1 slices of ITS channel
// (agent-scope flag per slice, epoch-tagged: nothing is cleared between launches), and applies from the registers: x is read once.
// What this measures: whether that one-launch form is worth building into rn_convnorm.hip.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/dbg/libs/bn_coop_bench tools/dbg/bn_coop_bench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

typedef __attribute__((ext_vector_type(4))) float f32x4;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
constexpr int T = 256;

__device__ __forceinline__ void block_reduce2(double& a, double& b, double* red) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
  if (lane == 0) { red[2 * w] = a; red[2 * w + 1] = b; }
  __syncthreads();
  a = red[0] + red[2] + red[4] + red[6];
  b = red[1] + red[3] + red[5] + red[7];
  __syncthreads();
}
// element q (float4 index within channel c) of an NCHW tensor with hw4 float4 per plane
__device__ __forceinline__ long addr4(long q, int c, int C, int hw4) { return ((q / hw4) * C + c) * hw4 + q % hw4; }

// (A1) slice s of channel c: the float4 elements s * per .. (s + 1) * per of the channel (contiguous: the same slices as (B))
__global__ __launch_bounds__(T) void stats_kernel(const f32x4* __restrict__ x, double* __restrict__ part, int C, int hw4, long n4, int S) {
  __shared__ double red[8];
  const int c = blockIdx.y, s = blockIdx.x;
  const long per = n4 / S;
  float a0 = 0.f, a1 = 0.f;
  for (long q = s * per + threadIdx.x; q < (s + 1) * per; q += T) {
    const f32x4 v = x[addr4(q, c, C, hw4)];
    a0 += (v[0] + v[1]) + (v[2] + v[3]);
    a1 += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
  }
  double d0 = a0, d1 = a1;
  block_reduce2(d0, d1, red);
  if (threadIdx.x == 0) { part[((long)c * S + s) * 2] = d0; part[((long)c * S + s) * 2 + 1] = d1; }
}
__device__ __forceinline__ void scale_shift(const double* part, int c, int S, double count, float& sc, float& sh) {
  double s0 = 0.0, s1 = 0.0;
  for (int i = 0; i < S; ++i) { s0 += part[((long)c * S + i) * 2]; s1 += part[((long)c * S + i) * 2 + 1]; }
  const double md = s0 / count;
  double var = s1 / count - md * md;
  if (var < 0.0) var = 0.0;
  const float m = (float)md, is = (float)(1.0 / sqrt(var + 1e-5));
  sc = is; sh = -m * is;
}
// (A2)
__global__ __launch_bounds__(T) void apply_kernel(const f32x4* __restrict__ x, f32x4* __restrict__ y, const double* __restrict__ part, int C, int hw4,
                                                  long n4, int S, double count) {
  const int c = blockIdx.y, s = blockIdx.x;
  float sc, sh;
  scale_shift(part, c, S, count, sc, sh);
  const long per = n4 / S;
  for (long q = s * per + threadIdx.x; q < (s + 1) * per; q += T) {
    const long a = addr4(q, c, C, hw4);
    const f32x4 v = x[a];
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = fmaxf(v[e] * sc + sh, 0.f);
    y[a] = o;
  }
}
// (B) NV float4 per thread in registers: per = NV * T
template <int NV>
__global__ __launch_bounds__(T) void coop_kernel(const f32x4* __restrict__ x, f32x4* __restrict__ y, double* __restrict__ part, unsigned* __restrict__ flag,
                                                 unsigned epoch, int C, int hw4, long n4, int S, double count, unsigned* __restrict__ err) {
  __shared__ double red[8];
  __shared__ float ss[2];
  const int c = blockIdx.y, s = blockIdx.x;
  const long per = n4 / S;
  f32x4 v[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = x[addr4(s * per + (long)i * T + threadIdx.x, c, C, hw4)];
  float a0 = 0.f, a1 = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    a0 += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    a1 += (v[i][0] * v[i][0] + v[i][1] * v[i][1]) + (v[i][2] * v[i][2] + v[i][3] * v[i][3]);
  }
  double d0 = a0, d1 = a1;
  block_reduce2(d0, d1, red);
  if (threadIdx.x < 64) {
    if (threadIdx.x == 0) {
      __hip_atomic_store(&part[((long)c * S + s) * 2], d0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&part[((long)c * S + s) * 2 + 1], d1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&flag[c * S + s], epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    // wave 0: lane l waits for slice l of this channel (bounded: a launch that is not wholly resident must not hang)
    const int l = threadIdx.x;
    unsigned spins = 0;
    bool ok = false;
    while (true) {
      const unsigned f = l < S ? __hip_atomic_load(&flag[c * S + l], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) : epoch;
      if (__ballot(f != epoch) == 0ull) { ok = true; break; }
      if (++spins > (1u << 22)) break;
      __builtin_amdgcn_s_sleep(1);
    }
    if (!ok && l == 0) atomicAdd(err, 1u);
    if (l == 0) {
      double s0 = 0.0, s1 = 0.0;
      for (int i = 0; i < S; ++i) {
        s0 += __hip_atomic_load(&part[((long)c * S + i) * 2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s1 += __hip_atomic_load(&part[((long)c * S + i) * 2 + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      const double md = s0 / count;
      double var = s1 / count - md * md;
      if (var < 0.0) var = 0.0;
      const float m = (float)md, is = (float)(1.0 / sqrt(var + 1e-5));
      ss[0] = is; ss[1] = -m * is;
    }
  }
  __syncthreads();
  const float sc = ss[0], sh = ss[1];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = fmaxf(v[i][e] * sc + sh, 0.f);
    y[addr4(s * per + (long)i * T + threadIdx.x, c, C, hw4)] = o;
  }
}

template <typename F> float time_us(F fn, int reps = 40) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 5; ++i) fn();
  std::vector<float> ts;
  for (int i = 0; i < reps; ++i) {
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); fn(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms * 1e3f);
  }
  std::sort(ts.begin(), ts.end());
  return ts[ts.size() / 2];
}

template <int NV> void run(int N, int C, int HW, int S) {
  const int hw4 = HW / 4;
  const long n4 = (long)N * hw4, tot4 = n4 * C;
  if (n4 / S != (long)NV * T) { printf("shape mismatch\n"); return; }
  f32x4 *x, *ya, *yb; double *pa, *pb; unsigned *flag, *err;
  CK(hipMalloc(&x, tot4 * 16)); CK(hipMalloc(&ya, tot4 * 16)); CK(hipMalloc(&yb, tot4 * 16));
  CK(hipMalloc(&pa, C * S * 16)); CK(hipMalloc(&pb, C * S * 16)); CK(hipMalloc(&flag, C * S * 4)); CK(hipMalloc(&err, 4));
  CK(hipMemset(flag, 0, C * S * 4)); CK(hipMemset(err, 0, 4));
  std::vector<float> h(tot4 * 4);
  unsigned r = 12345u;
  for (auto& v : h) { r = r * 1664525u + 1013904223u; v = ((r >> 8) & 0xffff) / 65536.f * 3.f - 1.f; }
  CK(hipMemcpy(x, h.data(), tot4 * 16, hipMemcpyHostToDevice));
  const double count = (double)N * HW;
  dim3 grid(S, C);
  unsigned epoch = 0;
  auto A = [&]() {
    stats_kernel<<<grid, T>>>(x, pa, C, hw4, n4, S);
    apply_kernel<<<grid, T>>>(x, ya, pa, C, hw4, n4, S, count);
  };
  auto B = [&]() { coop_kernel<NV><<<grid, T>>>(x, yb, pb, flag, ++epoch, C, hw4, n4, S, count, err); };
  const float ta = time_us(A), tb = time_us(B);
  const float ts = time_us([&]() { stats_kernel<<<grid, T>>>(x, pa, C, hw4, n4, S); });
  const float tp = time_us([&]() { apply_kernel<<<grid, T>>>(x, ya, pa, C, hw4, n4, S, count); });
  A(); B(); CK(hipDeviceSynchronize());
  std::vector<float> ha(tot4 * 4), hb(tot4 * 4);
  CK(hipMemcpy(ha.data(), ya, tot4 * 16, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb.data(), yb, tot4 * 16, hipMemcpyDeviceToHost));
  unsigned herr; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
  long ndiff = 0; for (size_t i = 0; i < ha.size(); ++i) ndiff += ha[i] != hb[i];
  printf("N %d C %d HW %d (%.1f MB), %d slices x %d channels = %d workgroups: two launches %.1f us (stats alone %.1f, apply alone %.1f); one cooperative launch %.1f us; "
         "outputs differ in %ld of %zu values; give-ups %u\n", N, C, HW, tot4 * 16 / 1e6, S, C, S * C, ta, ts, tp, tb, ndiff, ha.size(), herr);
}

int main() {
  run<8>(64, 24, 64 * 64, 32);      // conv layer 1's output
  run<4>(64, 24, 64 * 64, 64);
  run<2>(64, 24, 32 * 32, 32);      // layer 2's
  run<1>(64, 24, 32 * 32, 64);
  return 0;
}
