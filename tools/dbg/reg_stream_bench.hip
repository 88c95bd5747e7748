// How fast can the weight gradient's operand stream be pulled HBM -> REGISTERS (no LDS ring), with the row-blocked images'
// own address pattern?  Each wave of a 4-wave workgroup takes every 4th 32-row group of its workgroup's row range and keeps D
// groups of 16-byte-per-lane loads in flight (a load instruction = two 512-byte runs: 32 features x 8 / 16 rows of two row blocks).
// Compared with the product kernel's LDS-DMA ring alone (tools/time_wgrad.py, ABL 1: 138 us for the step's three jobs on 192 CUs).
// build: hipcc --offload-arch=gfx950 -O3 -o tools/dbg/libs/reg_stream_bench tools/dbg/reg_stream_bench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

struct Job { const unsigned char* dz; const unsigned char* a; int dz8; };
struct Args { Job job[3]; int njobs, S, Z; unsigned* sink; };

// NZ / NA: loads per group for the dZ operand / the A operand of the workgroup's 128 x 128 block
template <int D>
__global__ __launch_bounds__(256) void stream_kernel(Args a) {
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int blk = slot % 4, u = (slot / 4) * 8 + xcd;
  if (u >= a.njobs * a.Z) return;
  const int job = u % a.njobs, z = u / a.njobs;
  const Job jb = a.job[job];
  const int nh = blk & 1, kb = blk >> 1;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, n = lane & 31, h = lane >> 5;
  const int s0 = (int)((long)z * a.S / a.Z), s1 = (int)((long)(z + 1) * a.S / a.Z);
  const int g0 = 2 * s0 + w, g1 = 2 * s1;                 // 32-row groups of this wave: g0, g0 + 4, ...
  u32x4 buf[D][12];
  u32x4 x = {0u, 0u, 0u, 0u};
  auto load_group = [&](int g, u32x4 (&b)[12]) {
    // dZ: bf16 image, 8-row blocks: MFMA a / b (q) of n block i: block 4 g + q + 2 h; e4m3 image (gate job): 16-row block 2 g + h
    if (jb.dz8) {
#pragma unroll
      for (int i = 0; i < 4; ++i) b[i] = *reinterpret_cast<const u32x4*>(jb.dz + (((long)(2 * g + h) * 256 + nh * 128 + 32 * i + n) * 16));
#pragma unroll
      for (int i = 4; i < 8; ++i) b[i] = u32x4{0u, 0u, 0u, 0u};
    } else {
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i) b[4 * q + i] = *reinterpret_cast<const u32x4*>(jb.dz + (((long)(4 * g + q + 2 * h) * 256 + nh * 128 + 32 * i + n) * 16));
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) b[8 + j] = *reinterpret_cast<const u32x4*>(jb.a + (((long)(2 * g + h) * 256 + kb * 128 + 32 * j + n) * 16));
  };
  int g = g0;
#pragma unroll
  for (int d = 0; d < D; ++d) {
    if (g + 4 * d < g1) load_group(g + 4 * d, buf[d]);
  }
  for (; g < g1; g += 4 * D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      if (g + 4 * d < g1) {
#pragma unroll
        for (int i = 0; i < 12; ++i) x ^= buf[d][i];
        if (g + 4 * (d + D) < g1) load_group(g + 4 * (d + D), buf[d]);
      }
    }
  }
  if ((x[0] ^ x[1] ^ x[2] ^ x[3]) == 0x12345678u) a.sink[threadIdx.x] = x[0];
}

template <int D> float run(const Args& a, int grid) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  std::vector<float> ts;
  for (int it = 0; it < 12; ++it) {
    hipEventRecord(e0);
    stream_kernel<D><<<grid, 256>>>(a);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (it >= 2) ts.push_back(ms * 1e3f);
  }
  std::sort(ts.begin(), ts.end());
  return ts[ts.size() / 2];
}

// Variant "dup": a wave owns a 64 x 64 tile (wn, wk) and walks EVERY group of the range: 4 dZ + 2 A loads per group (gate job: 2 + 2),
// each fragment requested by two waves of the workgroup (L1 / L2 hits for the second) -- no LDS, no cross-wave reduction.
template <int D>
__global__ __launch_bounds__(256) void stream_dup_kernel(Args a) {
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int blk = slot % 4, u = (slot / 4) * 8 + xcd;
  if (u >= a.njobs * a.Z) return;
  const int job = u % a.njobs, z = u / a.njobs;
  const Job jb = a.job[job];
  const int nh = blk & 1, kb = blk >> 1;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, n = lane & 31, h = lane >> 5, wn = w >> 1, wk = w & 1;
  const int s0 = (int)((long)z * a.S / a.Z), s1 = (int)((long)(z + 1) * a.S / a.Z);
  const int g0 = 2 * s0, g1 = 2 * s1;
  u32x4 buf[D][6];
  u32x4 x = {0u, 0u, 0u, 0u};
  auto load_group = [&](int g, u32x4 (&b)[6]) {
    if (jb.dz8) {
#pragma unroll
      for (int i = 0; i < 2; ++i) b[i] = *reinterpret_cast<const u32x4*>(jb.dz + (((long)(2 * g + h) * 256 + nh * 128 + wn * 64 + 32 * i + n) * 16));
      b[2] = b[3] = u32x4{0u, 0u, 0u, 0u};
    } else {
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int i = 0; i < 2; ++i) b[2 * q + i] = *reinterpret_cast<const u32x4*>(jb.dz + (((long)(4 * g + q + 2 * h) * 256 + nh * 128 + wn * 64 + 32 * i + n) * 16));
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) b[4 + j] = *reinterpret_cast<const u32x4*>(jb.a + (((long)(2 * g + h) * 256 + kb * 128 + wk * 64 + 32 * j + n) * 16));
  };
  int g = g0;
#pragma unroll
  for (int d = 0; d < D; ++d)
    if (g + d < g1) load_group(g + d, buf[d]);
  for (; g < g1; g += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      if (g + d < g1) {
#pragma unroll
        for (int i = 0; i < 6; ++i) x ^= buf[d][i];
        if (g + d + D < g1) load_group(g + d + D, buf[d]);
      }
    }
  }
  if ((x[0] ^ x[1] ^ x[2] ^ x[3]) == 0x12345678u) a.sink[threadIdx.x] = x[0];
}

template <int D> float run_dup(const Args& a, int grid) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  std::vector<float> ts;
  for (int it = 0; it < 12; ++it) {
    hipEventRecord(e0);
    stream_dup_kernel<D><<<grid, 256>>>(a);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (it >= 2) ts.push_back(ms * 1e3f);
  }
  std::sort(ts.begin(), ts.end());
  return ts[ts.size() / 2];
}

int main(int argc, char** argv) {
  const long M = 64L * 4096;
  unsigned char *dz[2], *a8[3]; unsigned* sink;
  for (int i = 0; i < 2; ++i) { hipMalloc(&dz[i], M * 512); hipMemset(dz[i], 1, M * 512); }
  for (int i = 0; i < 3; ++i) { hipMalloc(&a8[i], M * 256); hipMemset(a8[i], 1, M * 256); }
  hipMalloc(&sink, 4096);
  for (int total : {48, 64}) {
    for (int njobs : {3, 1}) {
      Args a;
      a.njobs = njobs; a.S = (int)(M / 64); a.Z = total / njobs; a.sink = sink;
      a.job[0] = Job{dz[0], a8[0], 0};
      a.job[1] = Job{dz[1], a8[1], 0};
      a.job[2] = Job{a8[2], a8[2], 1};
      const int grid = 8 * 4 * ((njobs * a.Z + 7) / 8);
      const double mb = njobs == 3 ? (2 * (M * 512) + 3 * (M * 256)) / 1e6 : (M * 768) / 1e6;
      printf("%d job(s), %3d row splits (%d workgroups), %.0f MB:", njobs, total, 4 * njobs * a.Z, mb);
      float t;
      t = run<1>(a, grid); printf("  D=1 %6.1f us (%.2f TB/s)", t, mb / t);
      t = run<2>(a, grid); printf("  D=2 %6.1f us (%.2f TB/s)", t, mb / t);
      t = run<3>(a, grid); printf("  D=3 %6.1f us (%.2f TB/s)", t, mb / t);
      t = run<4>(a, grid); printf("  D=4 %6.1f us (%.2f TB/s)", t, mb / t);
      t = run<6>(a, grid); printf("  D=6 %6.1f us (%.2f TB/s)", t, mb / t);
      t = run<8>(a, grid); printf("  D=8 %6.1f us (%.2f TB/s)\n", t, mb / t);
      printf("     64 x 64 tile per wave, shared fragments loaded twice:");
      t = run_dup<2>(a, grid); printf("  D=2 %6.1f us (%.2f TB/s)", t, mb / t);
      t = run_dup<4>(a, grid); printf("  D=4 %6.1f us (%.2f TB/s)", t, mb / t);
      t = run_dup<8>(a, grid); printf("  D=8 %6.1f us (%.2f TB/s)\n", t, mb / t);
    }
  }
  return 0;
}
