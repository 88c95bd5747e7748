// Cost of a software grid barrier on MI355X (8 XCDs): G co-resident workgroups, K barriers per launch.
// build: hipcc --offload-arch=gfx950 -O3 -o gpurun_out/gbb tools/dbg/grid_barrier_bench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

__device__ __forceinline__ void grid_barrier(unsigned* cnt, unsigned target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    atomicAdd(cnt, 1u);
    while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    __threadfence();
  }
  __syncthreads();
}

__global__ void bench(unsigned* cnt, unsigned* done, float* data, int K) {
  float v = 0.f;
  for (int k = 0; k < K; ++k) {
    data[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = v + k;       // something to publish
    grid_barrier(cnt, (unsigned)(k + 1) * gridDim.x);
    v += data[(size_t)((blockIdx.x + 1) % gridDim.x) * blockDim.x + threadIdx.x];   // read the neighbour's value
  }
  data[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    if (atomicAdd(done, 1u) == gridDim.x - 1) { *cnt = 0u; *done = 0u; __threadfence(); }
  }
}

int main() {
  unsigned* cnt; float* data;
  hipMalloc(&cnt, 8); hipMemset(cnt, 0, 8);
  hipMalloc(&data, 4096 * 1024 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int T : {256, 1024})
    for (int G : {64, 128, 256, 512, 1024, 2048}) {
      if ((long)G * T > 256L * 1024) continue;   // half of the chip's 256 x 2048 resident threads: every workgroup co-resident
      float t[2];
      int Ks[2] = {2, 34};
      for (int i = 0; i < 2; ++i) {
        for (int rep = 0; rep < 3; ++rep) {
          hipEventRecord(e0);
          bench<<<G, T>>>(cnt, cnt + 1, data, Ks[i]);
          hipEventRecord(e1); hipEventSynchronize(e1);
          hipEventElapsedTime(&t[i], e0, e1);
        }
      }
      // correctness: after K barriers v = sum of neighbour's (v_prev + k): every thread identical
      std::vector<float> h(T); hipMemcpy(h.data(), data, T * 4, hipMemcpyDeviceToHost);
      printf("T=%4d G=%4d  %.2f us per barrier (K=2: %.1f us, K=34: %.1f us)  check %.0f\n", T, G, (t[1] - t[0]) * 1e3 / 32, t[0] * 1e3, t[1] * 1e3, h[0]); fflush(stdout);
    }
  return 0;
}
