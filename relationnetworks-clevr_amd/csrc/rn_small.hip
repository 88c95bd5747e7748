// K4 -- small exact-fp32 kernels around the pair path: f_phi (model.py:155-160) forward and
// backward, log_softmax (model.py:162) and the (B*n x G) tail GEMMs of the pair backward.
// These are ~0.01 % of the step's flops; they run on the fp32 MFMA
// (v_mfma_f32_32x32x2_f32 = a k-ordered fp32 fmaf chain) so f_phi stays bit-comparable with an
// fp32 reference, one wave per 32x32 output tile, operands straight from global (L2-resident).
#include "rn_common.h"

// One workgroup (4 waves) per 32x32 output tile; the 4 waves split K (each runs a k-ordered fp32
// fmaf chain over its quarter), partial tiles are combined through LDS in a fixed order.
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ A, long sam, long sak,
                                                       const float* __restrict__ B, long sbk, long sbn,
                                                       float* __restrict__ C, long ldc, int M, int N, int K,
                                                       const float* __restrict__ bias, const float* __restrict__ mul,
                                                       long ldmul, const float* __restrict__ gate, long ldgate,
                                                       int flags, int tiles_n) {
  __shared__ float red[4][16][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int tile = blockIdx.x;
  const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
  const int m = tm * 32 + (lane & 31), n = tn * 32 + (lane & 31);
  const int kh = lane >> 5;
  const bool mok = m < M, nok = n < N;
  const float* ap = A + (long)(mok ? m : 0) * sam;
  const float* bp = B + (long)(nok ? n : 0) * sbn;
  const int kq = ((K + 3) / 4 + 7) / 8 * 8;              // per-wave K range, multiple of 8
  const int kbeg = w * kq;
  const int kend = (kbeg + kq < K) ? kbeg + kq : K;
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  int k = kbeg;
  for (; k + 16 <= kend; k += 16) {
    float av[8], bv[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      av[s] = ap[(long)(k + 2 * s + kh) * sak];
      bv[s] = bp[(long)(k + 2 * s + kh) * sbk];
    }
#pragma unroll
    for (int s = 0; s < 8; ++s)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(mok ? av[s] : 0.f, nok ? bv[s] : 0.f, acc, 0, 0, 0);
  }
  for (; k < kend; k += 2) {
    const bool kok = (k + kh) < kend;
    const float av = (mok && kok) ? ap[(long)(k + kh) * sak] : 0.f;
    const float bv = (nok && kok) ? bp[(long)(k + kh) * sbk] : 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
  }
#pragma unroll
  for (int e = 0; e < 16; ++e) red[w][e][lane] = acc[e];
  __syncthreads();
  // wave w finishes accumulator registers 4w..4w+3.
  // D[i][j]: j = lane&31 -> column n ; i = (reg&3) + 8*(reg>>2) + 4*(lane>>5) -> row within the tile
  if (!nok) return;
  const float bn = bias ? bias[n] : 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int reg = 4 * w + r;
    const int row = tm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * kh;
    if (row < M) {
      float v = ((red[0][reg][lane] + red[1][reg][lane]) + red[2][reg][lane]) + red[3][reg][lane] + bn;
      if (mul) v *= mul[(long)row * ldmul + n];
      if (flags & RN_RELU) v = fmaxf(v, 0.f);
      if (gate) v = gate[(long)row * ldgate + n] > 0.f ? v : 0.f;
      float* cp = C + (long)row * ldc + n;
      if (flags & RN_ACCUMULATE) v += *cp;
      *cp = v;
    }
  }
}

extern "C" int rn_gemm_f32(const float* A, long sam, long sak, const float* B, long sbk, long sbn, float* C, long ldc,
                           int M, int N, int K, const float* bias, const float* mul, long ldmul, const float* gate,
                           long ldgate, int flags, void* stream) {
  RN_CHECK_ARG(A && B && C && M > 0 && N > 0 && K > 0, "rn_gemm_f32: bad pointer/size");
  const int tiles_m = cdiv(M, 32), tiles_n = cdiv(N, 32);
  gemm_f32_kernel<<<tiles_m * tiles_n, 256, 0, (hipStream_t)stream>>>(A, sam, sak, B, sbk, sbn, C, ldc, M, N, K, bias, mul,
                                                                     ldmul, gate, ldgate, flags, tiles_n);
  RN_LAUNCH_CHECK("rn_gemm_f32");
  return 0;
}

// one wave per row
__global__ __launch_bounds__(256) void log_softmax_fwd_kernel(const float* __restrict__ z, float* __restrict__ out,
                                                              int Bn, int A) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= Bn) return;
  const float* zr = z + (long)row * A;
  float mx = -INFINITY;
  for (int c = lane; c < A; c += 64) mx = fmaxf(mx, zr[c]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  float sum = 0.f;
  for (int c = lane; c < A; c += 64) sum += expf(zr[c] - mx);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  const float lse = mx + logf(sum);
  for (int c = lane; c < A; c += 64) out[(long)row * A + c] = zr[c] - lse;
}

__global__ __launch_bounds__(256) void log_softmax_bwd_kernel(const float* __restrict__ out,
                                                              const float* __restrict__ gout, float* __restrict__ dz,
                                                              int Bn, int A) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= Bn) return;
  float gs = 0.f;
  for (int c = lane; c < A; c += 64) gs += gout[(long)row * A + c];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) gs += __shfl_xor(gs, o);
  for (int c = lane; c < A; c += 64) {
    const long i = (long)row * A + c;
    dz[i] = gout[i] - expf(out[i]) * gs;
  }
}

extern "C" int rn_log_softmax_fwd(const float* z, float* out, int B, int A, void* stream) {
  RN_CHECK_ARG(z && out && B > 0 && A > 0, "rn_log_softmax_fwd: bad pointer/size");
  log_softmax_fwd_kernel<<<cdiv(B, 4), 256, 0, (hipStream_t)stream>>>(z, out, B, A);
  RN_LAUNCH_CHECK("rn_log_softmax_fwd");
  return 0;
}

extern "C" int rn_log_softmax_bwd(const float* out, const float* gout, float* dz, int B, int A, void* stream) {
  RN_CHECK_ARG(out && gout && dz && B > 0 && A > 0, "rn_log_softmax_bwd: bad pointer/size");
  log_softmax_bwd_kernel<<<cdiv(B, 4), 256, 0, (hipStream_t)stream>>>(out, gout, dz, B, A);
  RN_LAUNCH_CHECK("rn_log_softmax_bwd");
  return 0;
}

__global__ __launch_bounds__(256) void colsum_f32_kernel(const float* __restrict__ src, long ld, float* __restrict__ out,
                                                         int R, int C) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float s = 0.f;
  for (int r = 0; r < R; ++r) s += src[(long)r * ld + c];
  out[c] = s;
}

extern "C" int rn_colsum_f32(const float* src, long ld, float* out, int R, int C, void* stream) {
  RN_CHECK_ARG(src && out && R > 0 && C > 0, "rn_colsum_f32: bad pointer/size");
  colsum_f32_kernel<<<cdiv(C, 256), 256, 0, (hipStream_t)stream>>>(src, ld, out, R, C);
  RN_LAUNCH_CHECK("rn_colsum_f32");
  return 0;
}
