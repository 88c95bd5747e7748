// BatchNorm2d + ReLU of the conv stack that feeds the relation layer (reference model.py:22-35:
// x = conv(img); x = batchNorm(x); x = relu(x), four times), fused into two HBM passes per direction.
//
// The stock path spends ~0.5 ms per training step here in a dozen separate launches (bias add, batch-norm
// statistics + apply, ReLU, their three backward kernels and a separate bias-gradient reduction), each a full
// trip over the 25 MB layer-1 activation.  The arithmetic is a per-channel reduction plus a point-wise map:
//
//   forward   pass 1: per-channel sum / sum of squares of the conv output x            (read x)
//             pass 2: mean / invstd from the slice sums (+ running statistics), y = relu((x - mean) * invstd * gamma + beta)
//                                                                                      (read x, write y)
//   backward  pass 1: S1 = sum dz, S2 = sum dz * x with dz = dy * (y > 0)              (read dy, x)
//             pass 2: dx = gamma * invstd * (dz - S1/n - xhat * dgamma/n)              (read dy, x, write dx)
//
// The ReLU mask is recomputed from x with the forward's own expression (bit-identical), so neither y nor a mask
// is read back.  The convolution bias drops out of a batch-normalised output (it shifts the batch mean by the
// same amount) -- the conv runs without it, the running mean gets it added, and its gradient is identically zero.
// NCHW fp32; a channel's data are N planes of HW contiguous floats (HW % 4 == 0).
#include <stdlib.h>

#include "rn_common.h"

namespace {
constexpr int CN_T = 256;

// block (x, c): the float4 indices q = x*CN_T*U .. of channel c, q -> plane n = q / hw4, offset q % hw4
__device__ __forceinline__ long cn_addr4(long q, int c, int C, int hw4) {
  const long n = q / hw4;
  return (n * C + c) * hw4 + (q - n * hw4);
}

template <int NV>
__device__ __forceinline__ void cn_block_reduce(double (&v)[NV], double* red) {   // fixed order -> deterministic
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    double x = v[i];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) x += __shfl_xor(x, o);
    if (lane == 0) red[w * NV + i] = x;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = red[i] + red[NV + i] + red[2 * NV + i] + red[3 * NV + i];
  __syncthreads();
}
// The S (<= 32) slice partials of a channel, added in slice order: lane l loads partial l (one round trip for all of them -- the
// plain loop is S dependent scalar loads, ~10 us in front of every workgroup of the apply kernels, which is most of what those
// kernels took), then the values are read back lane by lane: the same sums, in the same order, as the loop gave.
__device__ __forceinline__ void cn_slice_sums(const double* __restrict__ part, int c, int S, double& s0, double& s1) {
  const int l = threadIdx.x & 63;
  const double a = l < S ? part[((long)c * S + l) * 2] : 0.0, b = l < S ? part[((long)c * S + l) * 2 + 1] : 0.0;
  const u32x2 ab = __builtin_bit_cast(u32x2, a), bb = __builtin_bit_cast(u32x2, b);
  s0 = 0.0;
  s1 = 0.0;
  for (int i = 0; i < S; ++i) {                          // (i is wave-uniform: v_readlane)
    const u32x2 x = {(unsigned)__builtin_amdgcn_readlane((int)ab[0], i), (unsigned)__builtin_amdgcn_readlane((int)ab[1], i)};
    const u32x2 y = {(unsigned)__builtin_amdgcn_readlane((int)bb[0], i), (unsigned)__builtin_amdgcn_readlane((int)bb[1], i)};
    s0 += __builtin_bit_cast(double, x);
    s1 += __builtin_bit_cast(double, y);
  }
}
}  // namespace

// part[(c * S + s) * 2 + {0, 1}] = sum, sum of squares over slice s of channel c
__global__ __launch_bounds__(CN_T) void cn_stats_kernel(const f32x4* __restrict__ x, double* __restrict__ part, int C, int hw4,
                                                        long n4, int S) {
  __shared__ double red[8];
  const int c = blockIdx.y, s = blockIdx.x;
  float a0 = 0.f, a1 = 0.f;
  for (long q = (long)s * CN_T + threadIdx.x; q < n4; q += (long)S * CN_T) {
    const f32x4 v = x[cn_addr4(q, c, C, hw4)];
    a0 += (v[0] + v[1]) + (v[2] + v[3]);
    a1 += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
  }
  double v[2] = {(double)a0, (double)a1};
  cn_block_reduce<2>(v, red);
  if (threadIdx.x == 0) {
    part[((long)c * S + s) * 2] = v[0];
    part[((long)c * S + s) * 2 + 1] = v[1];
  }
}

// y = relu((x - mean) * invstd * gamma + beta).  FROM_PART (training): every block first adds up the channel's slice
// sums (fixed order) to mean / invstd; block (0, c) also publishes them for the backward pass and updates the running
// statistics (torch.nn.BatchNorm2d semantics: biased variance normalises, unbiased variance goes to running_var;
// the conv ran without its bias, so the bias is added to the running mean here).  !FROM_PART (evaluation): mean / invstd
// are caller-prepared.
template <bool FROM_PART>
__global__ __launch_bounds__(CN_T) void cn_apply_kernel(const f32x4* __restrict__ x, f32x4* __restrict__ y,
                                                        const double* __restrict__ part, int S, double count,
                                                        float* __restrict__ mean, float* __restrict__ invstd,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        const float* __restrict__ conv_bias, float eps, float momentum,
                                                        float* __restrict__ running_mean, float* __restrict__ running_var,
                                                        long long* __restrict__ num_batches, int C, int hw4, long n4) {
  const int c = blockIdx.y;
  float m, is;
  if constexpr (FROM_PART) {
    double s0, s1;
    cn_slice_sums(part, c, S, s0, s1);
    const double md = s0 / count;
    double var = s1 / count - md * md;
    if (var < 0.0) var = 0.0;
    m = (float)md;
    is = (float)(1.0 / sqrt(var + (double)eps));
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      mean[c] = m;
      invstd[c] = is;
      if (running_mean) {
        const double mb = md + (conv_bias ? (double)conv_bias[c] : 0.0);
        running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mb);
        const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
        running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unb);
      }
      if (c == 0 && num_batches) num_batches[0] += 1;
    }
  } else {
    m = mean[c];
    is = invstd[c];
  }
  const float sc = gamma[c] * is, sh = beta[c] - m * sc;
  for (long q = (long)blockIdx.x * CN_T + threadIdx.x; q < n4; q += (long)gridDim.x * CN_T) {
    const long a = cn_addr4(q, c, C, hw4);
    const f32x4 v = x[a];
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = fmaxf(v[e] * sc + sh, 0.f);
    y[a] = o;
  }
}

// part[(c * S + s) * 2 + {0, 1}] = sum dz, sum dz * x   (dz = dy where the forward output was > 0)
__global__ __launch_bounds__(CN_T) void cn_bwd_sums_kernel(const f32x4* __restrict__ dy, const f32x4* __restrict__ x,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           double* __restrict__ part, int C, int hw4, long n4, int S) {
  __shared__ double red[8];
  const int c = blockIdx.y, s = blockIdx.x;
  const float sc = gamma[c] * invstd[c], sh = beta[c] - mean[c] * sc;
  float a0 = 0.f, a1 = 0.f;
  for (long q = (long)s * CN_T + threadIdx.x; q < n4; q += (long)S * CN_T) {
    const long a = cn_addr4(q, c, C, hw4);
    const f32x4 v = x[a], g = dy[a];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float dz = (v[e] * sc + sh > 0.f) ? g[e] : 0.f;
      a0 += dz;
      a1 += dz * v[e];
    }
  }
  double v[2] = {(double)a0, (double)a1};
  cn_block_reduce<2>(v, red);
  if (threadIdx.x == 0) {
    part[((long)c * S + s) * 2] = v[0];
    part[((long)c * S + s) * 2 + 1] = v[1];
  }
}

// dgamma = invstd * (S2 - mean * S1), dbeta = S1;
// dx = gamma * invstd * (dz - S1 / n - (x - mean) * invstd * dgamma / n)
__global__ __launch_bounds__(CN_T) void cn_bwd_apply_kernel(const f32x4* __restrict__ dy, const f32x4* __restrict__ x,
                                                            f32x4* __restrict__ dx, const double* __restrict__ part, int S,
                                                            double count, const float* __restrict__ mean,
                                                            const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, float* __restrict__ zero_out, int C, int hw4, long n4) {
  const int c = blockIdx.y;
  double s1, s2;
  cn_slice_sums(part, c, S, s1, s2);
  const float m = mean[c], is = invstd[c], ga = gamma[c];
  const float dga = (float)((double)is * (s2 - (double)m * s1));
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    dgamma[c] = dga;
    dbeta[c] = (float)s1;
    if (zero_out) zero_out[c] = 0.f;
  }
  const float sc = ga * is, sh = beta[c] - m * sc;
  const float k0 = (float)(s1 / count), k1 = (float)((double)dga * is / count);
  for (long q = (long)blockIdx.x * CN_T + threadIdx.x; q < n4; q += (long)gridDim.x * CN_T) {
    const long a = cn_addr4(q, c, C, hw4);
    const f32x4 v = x[a], g = dy[a];
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float dz = (v[e] * sc + sh > 0.f) ? g[e] : 0.f;
      o[e] = sc * (dz - k0 - (v[e] - m) * k1);
    }
    dx[a] = o;
  }
}

// ------------------------------------------------------------------------------------------------ one launch per direction
// Layers whose channel holds <= 16 Ki elements (the last two conv blocks at the headline shape): ONE workgroup of 1024
// threads per channel does statistics AND apply -- the data of a channel (<= 64 KB) sit in the threads' registers (NV
// float4 each) between the two passes.  Two launches and a workspace round trip per direction and layer become one
// launch; these layers are pure launch latency.  (One workgroup per channel on the 64 Ki-element layer -- re-reading from
// L2 -- measured slower than the two-launch path: 24 workgroups do not pull 12 MB fast enough.)
namespace {
constexpr int CF_T = 1024;
template <int NV>
__device__ __forceinline__ void cf_block_reduce(double (&v)[NV], double* red) {   // 16 waves, fixed order -> deterministic
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    double x = v[i];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) x += __shfl_xor(x, o);
    if (lane == 0) red[w * NV + i] = x;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < CF_T / 64; ++k) s += red[k * NV + i];
    v[i] = s;
  }
  __syncthreads();
}
}  // namespace

template <int NV, bool KEEP>
__global__ __launch_bounds__(CF_T) void cn_fwd_fused_kernel(const f32x4* __restrict__ x, f32x4* __restrict__ y, double count,
                                                            float* __restrict__ mean, float* __restrict__ invstd,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            const float* __restrict__ conv_bias, float eps, float momentum,
                                                            float* __restrict__ running_mean, float* __restrict__ running_var,
                                                            long long* __restrict__ num_batches, int C, int hw4, long n4) {
  __shared__ double red[2 * CF_T / 64];
  const int c = blockIdx.x, t = threadIdx.x;
  f32x4 v[KEEP ? NV : 1];
  float a0 = 0.f, a1 = 0.f;
  if constexpr (KEEP) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const long q = (long)i * CF_T + t;
      v[i] = q < n4 ? x[cn_addr4(q, c, C, hw4)] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      a0 += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
      a1 += (v[i][0] * v[i][0] + v[i][1] * v[i][1]) + (v[i][2] * v[i][2] + v[i][3] * v[i][3]);
    }
  } else {
#pragma unroll 4
    for (long q = t; q < n4; q += CF_T) {
      const f32x4 u = x[cn_addr4(q, c, C, hw4)];
      a0 += (u[0] + u[1]) + (u[2] + u[3]);
      a1 += (u[0] * u[0] + u[1] * u[1]) + (u[2] * u[2] + u[3] * u[3]);
    }
  }
  double s[2] = {(double)a0, (double)a1};
  cf_block_reduce<2>(s, red);
  const double md = s[0] / count;
  double var = s[1] / count - md * md;
  if (var < 0.0) var = 0.0;
  const float m = (float)md, is = (float)(1.0 / sqrt(var + (double)eps));
  if (t == 0) {
    mean[c] = m;
    invstd[c] = is;
    if (running_mean) {
      const double mb = md + (conv_bias ? (double)conv_bias[c] : 0.0);
      running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mb);
      const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
      running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unb);
    }
    if (c == 0 && num_batches) num_batches[0] += 1;
  }
  const float sc = gamma[c] * is, sh = beta[c] - m * sc;
  if constexpr (KEEP) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const long q = (long)i * CF_T + t;
      if (q < n4) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = fmaxf(v[i][e] * sc + sh, 0.f);
        y[cn_addr4(q, c, C, hw4)] = o;
      }
    }
  } else {
#pragma unroll 4
    for (long q = t; q < n4; q += CF_T) {
      const long a = cn_addr4(q, c, C, hw4);
      const f32x4 u = x[a];
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = fmaxf(u[e] * sc + sh, 0.f);
      y[a] = o;
    }
  }
}

template <int NV, bool KEEP>
__global__ __launch_bounds__(CF_T) void cn_bwd_fused_kernel(const f32x4* __restrict__ dy, const f32x4* __restrict__ x,
                                                            f32x4* __restrict__ dx, double count, const float* __restrict__ mean,
                                                            const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, float* __restrict__ zero_out, int C, int hw4, long n4) {
  __shared__ double red[2 * CF_T / 64];
  const int c = blockIdx.x, t = threadIdx.x;
  const float m = mean[c], is = invstd[c], ga = gamma[c];
  const float sc = ga * is, sh = beta[c] - m * sc;
  f32x4 v[KEEP ? NV : 1], g[KEEP ? NV : 1];
  float a0 = 0.f, a1 = 0.f;
#pragma unroll(KEEP ? NV : 2)
  for (int i = 0; i < (KEEP ? NV : (int)((n4 + CF_T - 1) / CF_T)); ++i) {
    const long q = (long)i * CF_T + t;
    f32x4 vv = {0.f, 0.f, 0.f, 0.f}, gg = {0.f, 0.f, 0.f, 0.f};
    if (q < n4) {
      const long a = cn_addr4(q, c, C, hw4);
      vv = x[a];
      gg = dy[a];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float dz = (vv[e] * sc + sh > 0.f) ? gg[e] : 0.f;
      a0 += dz;
      a1 += dz * vv[e];
    }
    if constexpr (KEEP) { v[i] = vv; g[i] = gg; }
  }
  double s[2] = {(double)a0, (double)a1};
  cf_block_reduce<2>(s, red);
  const float dga = (float)((double)is * (s[1] - (double)m * s[0]));
  if (t == 0) {
    dgamma[c] = dga;
    dbeta[c] = (float)s[0];
    if (zero_out) zero_out[c] = 0.f;
  }
  const float k0 = (float)(s[0] / count), k1 = (float)((double)dga * is / count);
#pragma unroll(KEEP ? NV : 2)
  for (int i = 0; i < (KEEP ? NV : (int)((n4 + CF_T - 1) / CF_T)); ++i) {
    const long q = (long)i * CF_T + t;
    if (q < n4) {
      const long a = cn_addr4(q, c, C, hw4);
      f32x4 vv, gg;
      if constexpr (KEEP) { vv = v[i]; gg = g[i]; } else { vv = x[a]; gg = dy[a]; }
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float dz = (vv[e] * sc + sh > 0.f) ? gg[e] : 0.f;
        o[e] = sc * (dz - k0 - (vv[e] - m) * k1);
      }
      dx[a] = o;
    }
  }
}

static int cn_slices(long n4) {          // workgroups per channel of the reduction passes
  long s = n4 / (CN_T * 8);
  if (s < 1) s = 1;
  if (s > 32) s = 32;
  return (int)s;
}
static int cn_check(const char* who, const void* a, const void* b, int N, int C, int HW) {
  RN_CHECK_ARG(a && b && N > 0 && C > 0 && HW > 0 && HW % 4 == 0, "%s: bad pointer/size (N=%d C=%d HW=%d; HW must be a multiple of 4)", who, N, C, HW);
  RN_CHECK_ARG(((uintptr_t)a | (uintptr_t)b) % 16 == 0, "%s: tensors must be 16-byte aligned", who);
  return 0;
}

size_t rnws_bn_relu(int N, int C, int HW) {
  if (N <= 0 || C <= 0 || HW <= 0) return 0;
  return (size_t)C * cn_slices((long)N * HW / 4) * 2 * sizeof(double);
}

extern "C" int rn_bn_relu_fwd(const float* x, float* y, const float* gamma, const float* beta, const float* conv_bias,
                              float* running_mean, float* running_var, long long* num_batches, float* mean, float* invstd,
                              void* ws, float eps, float momentum, int N, int C, int HW, void* stream) {
  if (int rc = cn_check("rn_bn_relu_fwd", x, y, N, C, HW)) return rc;
  RN_CHECK_ARG(gamma && beta && mean && invstd && ws, "rn_bn_relu_fwd: NULL parameter / workspace");
  const long n4 = (long)N * HW / 4;
  const int S = cn_slices(n4), hw4 = HW / 4;
  hipStream_t s = (hipStream_t)stream;
  if (n4 <= 4 * CF_T) {                                   // one workgroup per channel: statistics + apply in one launch
#define RN_CF(NV_, KEEP_) cn_fwd_fused_kernel<NV_, KEEP_><<<C, CF_T, 0, s>>>((const f32x4*)x, (f32x4*)y, (double)N * HW, mean, invstd, gamma, beta, conv_bias, \
                                                               eps, momentum, running_mean, running_var, num_batches, C, hw4, n4)
    if (n4 <= CF_T) RN_CF(1, true);
    else RN_CF(4, true);
#undef RN_CF
    RN_LAUNCH_CHECK("rn_bn_relu_fwd(fused)");
    return 0;
  }
  cn_stats_kernel<<<dim3(S, C), CN_T, 0, s>>>((const f32x4*)x, (double*)ws, C, hw4, n4, S);
  int gx = (int)((n4 + CN_T * 4 - 1) / (CN_T * 4));
  if (gx < 1) gx = 1;
  cn_apply_kernel<true><<<dim3(gx, C), CN_T, 0, s>>>((const f32x4*)x, (f32x4*)y, (const double*)ws, S, (double)N * HW, mean, invstd, gamma,
                                                     beta, conv_bias, eps, momentum, running_mean, running_var, num_batches, C, hw4, n4);
  RN_LAUNCH_CHECK("rn_bn_relu_fwd");
  return 0;
}

// evaluation mode: `mean` / `invstd` hold (running_mean - conv_bias) and 1 / sqrt(running_var + eps)
extern "C" int rn_bn_relu_apply(const float* x, float* y, const float* gamma, const float* beta, const float* mean,
                                const float* invstd, int N, int C, int HW, void* stream) {
  if (int rc = cn_check("rn_bn_relu_apply", x, y, N, C, HW)) return rc;
  RN_CHECK_ARG(gamma && beta && mean && invstd, "rn_bn_relu_apply: NULL parameter");
  const long n4 = (long)N * HW / 4;
  int gx = (int)((n4 + CN_T * 4 - 1) / (CN_T * 4));
  if (gx < 1) gx = 1;
  cn_apply_kernel<false><<<dim3(gx, C), CN_T, 0, (hipStream_t)stream>>>((const f32x4*)x, (f32x4*)y, nullptr, 0, 1.0, const_cast<float*>(mean),
                                                                        const_cast<float*>(invstd), gamma, beta, nullptr, 0.f, 0.f, nullptr, nullptr,
                                                                        nullptr, C, HW / 4, n4);
  RN_LAUNCH_CHECK("rn_bn_relu_apply");
  return 0;
}

// pass 1 of the backward for rn_conv.hip's fused weight-gradient kernel (which is pass 2 there); *S = slices per channel in ws
int rn_cn_launch_bwd_sums(const float* dy, const float* x, const float* mean, const float* invstd, const float* gamma, const float* beta,
                          void* ws, int N, int C, int HW, void* stream, int* S) {
  if (int rc = cn_check("rn_bn_relu_bwd_conv_wgrad", dy, x, N, C, HW)) return rc;
  const long n4 = (long)N * HW / 4;
  *S = cn_slices(n4);
  cn_bwd_sums_kernel<<<dim3(*S, C), CN_T, 0, (hipStream_t)stream>>>((const f32x4*)dy, (const f32x4*)x, mean, invstd, gamma, beta, (double*)ws, C, HW / 4, n4, *S);
  RN_LAUNCH_CHECK("rn_bn_relu_bwd_conv_wgrad(sums)");
  return 0;
}

extern "C" int rn_bn_relu_bwd(const float* dy, const float* x, float* dx, const float* gamma, const float* beta, const float* mean,
                              const float* invstd, float* dgamma, float* dbeta, float* zero_out, void* ws, int N, int C, int HW, void* stream) {
  if (int rc = cn_check("rn_bn_relu_bwd", dy, x, N, C, HW)) return rc;
  RN_CHECK_ARG(dx && gamma && beta && mean && invstd && dgamma && dbeta && ws && (uintptr_t)dx % 16 == 0, "rn_bn_relu_bwd: NULL / misaligned argument");
  const long n4 = (long)N * HW / 4;
  const int S = cn_slices(n4), hw4 = HW / 4;
  hipStream_t s = (hipStream_t)stream;
  if (n4 <= 4 * CF_T) {
    const double cnt = (double)N * HW;
    if (n4 <= CF_T) cn_bwd_fused_kernel<1, true><<<C, CF_T, 0, s>>>((const f32x4*)dy, (const f32x4*)x, (f32x4*)dx, cnt, mean, invstd, gamma, beta, dgamma, dbeta, zero_out, C, hw4, n4);
    else cn_bwd_fused_kernel<4, true><<<C, CF_T, 0, s>>>((const f32x4*)dy, (const f32x4*)x, (f32x4*)dx, cnt, mean, invstd, gamma, beta, dgamma, dbeta, zero_out, C, hw4, n4);
    RN_LAUNCH_CHECK("rn_bn_relu_bwd(fused)");
    return 0;
  }
  cn_bwd_sums_kernel<<<dim3(S, C), CN_T, 0, s>>>((const f32x4*)dy, (const f32x4*)x, mean, invstd, gamma, beta, (double*)ws, C, hw4, n4, S);
  int gx = (int)((n4 + CN_T * 4 - 1) / (CN_T * 4));
  if (gx < 1) gx = 1;
  cn_bwd_apply_kernel<<<dim3(gx, C), CN_T, 0, s>>>((const f32x4*)dy, (const f32x4*)x, (f32x4*)dx, (const double*)ws, S, (double)N * HW,
                                                   mean, invstd, gamma, beta, dgamma, dbeta, zero_out, C, hw4, n4);
  RN_LAUNCH_CHECK("rn_bn_relu_bwd");
  return 0;
}
