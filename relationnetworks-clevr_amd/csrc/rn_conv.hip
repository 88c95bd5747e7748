// The four 3x3 / stride-2 / pad-1 convolutions of the conv stack in front of the relation layer (reference
// model.py:13-20: Conv2d(3 -> 24) and three Conv2d(24 -> 24), 128x128 -> 8x8): forward and input-gradient as direct
// fp32 convolutions.  They are tiny (0.34 .. 0.68 GFLOP at B = 64) and the library kernels spend 25-45 us each on
// them -- 0.22 ms on the critical path of a 1.3 ms training step.  Here a thread owns one output pixel (forward) or
// one 2x2 input block (input gradient) for ALL channels: COUT (x 4) accumulators, the 3x3xCIN weights are broadcast
// from LDS (every lane reads the same 16 bytes), 9 * CIN * COUT FMAs per thread against 9 * CIN (4 * COUT) loads.
// The weight gradient stays with MIOpen (it is off the dependency chain, on a side stream).
// NCHW fp32, H and W even.
#include "rn_common.h"

namespace {
constexpr int CV_T = 256;
}

// y[n][co][oy][ox] = sum_{ci,ky,kx} w[co][ci][ky][kx] * x[n][ci][2 oy + ky - 1][2 ox + kx - 1]
// thread = (output pixel, group of CG output channels): blockIdx.y = channel group.  The taps of input channel ci+1
// are loaded while channel ci is accumulated (the kernel is a chain of L2 round trips otherwise).
template <int CIN, int COUT, int CG>
__global__ __launch_bounds__(CV_T) void conv3x3s2_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             float* __restrict__ y, int N, int H, int W) {
  __shared__ __attribute__((aligned(16))) float ws[CIN * 9 * CG];            // [ci][tap][co in group]
  const int g0 = blockIdx.y * CG;
  for (int i = threadIdx.x; i < CIN * 9 * CG; i += CV_T) {
    const int co = i % CG, ct = i / CG;                                      // ct = ci * 9 + tap
    ws[i] = w[(long)(g0 + co) * CIN * 9 + ct];
  }
  __syncthreads();
  const int OH = H >> 1, OW = W >> 1;
  const long p = (long)blockIdx.x * CV_T + threadIdx.x;
  if (p >= (long)N * OH * OW) return;
  const int ox = (int)(p % OW), oy = (int)((p / OW) % OH), n = (int)(p / ((long)OW * OH));
  float acc[CG];
#pragma unroll
  for (int c = 0; c < CG; ++c) acc[c] = 0.f;
  const int iy0 = 2 * oy - 1, ix0 = 2 * ox - 1;
  const float* xn = x + (long)n * CIN * H * W;
  // tap offsets / validity are the same for every channel (only row / column -1 can be out of range: H, W even)
  int off[9];
  bool ok[9];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int iy = iy0 + ky, ix = ix0 + kx;
      ok[ky * 3 + kx] = iy >= 0 && ix >= 0;
      off[ky * 3 + kx] = ok[ky * 3 + kx] ? iy * W + ix : 0;
    }
  float t[9], tn[9];
#pragma unroll
  for (int tp = 0; tp < 9; ++tp) t[tp] = ok[tp] ? xn[off[tp]] : 0.f;
#pragma unroll 1
  for (int ci = 0; ci < CIN; ++ci) {
    if (ci + 1 < CIN) {
      const float* xc = xn + (long)(ci + 1) * H * W;
#pragma unroll
      for (int tp = 0; tp < 9; ++tp) tn[tp] = ok[tp] ? xc[off[tp]] : 0.f;
    }
    const float* wc = ws + ci * 9 * CG;
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) {
#pragma unroll
      for (int c = 0; c < CG; c += 4) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(wc + tp * CG + c);
        acc[c] = fmaf(wv[0], t[tp], acc[c]);
        acc[c + 1] = fmaf(wv[1], t[tp], acc[c + 1]);
        acc[c + 2] = fmaf(wv[2], t[tp], acc[c + 2]);
        acc[c + 3] = fmaf(wv[3], t[tp], acc[c + 3]);
      }
    }
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) t[tp] = tn[tp];
  }
  float* yo = y + ((long)n * COUT + g0) * OH * OW + (long)oy * OW + ox;
#pragma unroll
  for (int c = 0; c < CG; ++c) yo[(long)c * OH * OW] = acc[c];
}

// dx[n][ci][iy][ix] = sum_{co,ky,kx : iy + 1 - ky = 2 oy, ix + 1 - kx = 2 ox} dy[n][co][oy][ox] * w[co][ci][ky][kx]
// thread = (2x2 input block (2a.., 2b..), group of CG input channels): it needs dy[a..a+1][b..b+1]; pixel parity selects taps:
//   (2a, 2b): (1,1)<-dy[a][b];  (2a, 2b+1): (1,0)<-dy[a][b+1], (1,2)<-dy[a][b];  (2a+1, 2b): (0,1)<-dy[a+1][b], (2,1)<-dy[a][b];
//   (2a+1, 2b+1): (0,0)<-dy[a+1][b+1], (0,2)<-dy[a+1][b], (2,0)<-dy[a][b+1], (2,2)<-dy[a][b]
template <int CIN, int COUT, int CG>
__global__ __launch_bounds__(CV_T) void conv3x3s2_bwd_data_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                                  float* __restrict__ dx, int N, int H, int W) {
  __shared__ __attribute__((aligned(16))) float ws[COUT * 9 * CG];           // [co][tap][ci in group]
  const int g0 = blockIdx.y * CG;
  for (int i = threadIdx.x; i < COUT * 9 * CG; i += CV_T) {
    const int ci = i % CG, r = i / CG, tap = r % 9, co = r / 9;
    ws[i] = w[((long)co * CIN + g0 + ci) * 9 + tap];
  }
  __syncthreads();
  const int OH = H >> 1, OW = W >> 1;
  const long p = (long)blockIdx.x * CV_T + threadIdx.x;
  if (p >= (long)N * OH * OW) return;
  const int b = (int)(p % OW), a = (int)((p / OW) % OH), n = (int)(p / ((long)OW * OH));
  float a00[CG], a01[CG], a10[CG], a11[CG];                                // the block's four pixels, CG input channels
#pragma unroll
  for (int c = 0; c < CG; ++c) a00[c] = a01[c] = a10[c] = a11[c] = 0.f;
  const bool ra = a + 1 < OH, rb = b + 1 < OW;
  const float* dn = dy + (long)n * COUT * OH * OW + (long)a * OW + b;
  const int o01 = rb ? 1 : 0, o10 = ra ? OW : 0, o11 = (ra && rb) ? OW + 1 : 0;
  float d00 = dn[0], d01 = rb ? dn[o01] : 0.f, d10 = ra ? dn[o10] : 0.f, d11 = (ra && rb) ? dn[o11] : 0.f;
#pragma unroll 1
  for (int co = 0; co < COUT; ++co) {
    float e00 = 0.f, e01 = 0.f, e10 = 0.f, e11 = 0.f;
    if (co + 1 < COUT) {                                                     // next output channel's 2x2 patch
      const float* dc = dn + (long)(co + 1) * OH * OW;
      e00 = dc[0]; e01 = rb ? dc[o01] : 0.f; e10 = ra ? dc[o10] : 0.f; e11 = (ra && rb) ? dc[o11] : 0.f;
    }
    const float* wc = ws + co * 9 * CG;
#pragma unroll
    for (int c = 0; c < CG; c += 4) {
      f32x4 wv[9];
#pragma unroll
      for (int tp = 0; tp < 9; ++tp) wv[tp] = *reinterpret_cast<const f32x4*>(wc + tp * CG + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        a00[c + e] = fmaf(wv[4][e], d00, a00[c + e]);
        a01[c + e] = fmaf(wv[3][e], d01, fmaf(wv[5][e], d00, a01[c + e]));
        a10[c + e] = fmaf(wv[1][e], d10, fmaf(wv[7][e], d00, a10[c + e]));
        a11[c + e] = fmaf(wv[0][e], d11, fmaf(wv[2][e], d10, fmaf(wv[6][e], d01, fmaf(wv[8][e], d00, a11[c + e]))));
      }
    }
    d00 = e00; d01 = e01; d10 = e10; d11 = e11;
  }
  float* xo = dx + ((long)n * CIN + g0) * H * W + (long)(2 * a) * W + 2 * b;
#pragma unroll
  for (int c = 0; c < CG; ++c) {
    float* xc = xo + (long)c * H * W;
    *reinterpret_cast<u32x2*>(xc) = u32x2{__builtin_bit_cast(unsigned, a00[c]), __builtin_bit_cast(unsigned, a01[c])};
    *reinterpret_cast<u32x2*>(xc + W) = u32x2{__builtin_bit_cast(unsigned, a10[c]), __builtin_bit_cast(unsigned, a11[c])};
  }
}

static int cv_check(const char* who, const void* a, const void* b, const void* c, int N, int Cin, int Cout, int H, int W) {
  RN_CHECK_ARG(a && b && c && N > 0 && H > 0 && W > 0, "%s: bad pointer/size", who);
  RN_CHECK_ARG(H % 2 == 0 && W % 2 == 0 && Cout == 24 && (Cin == 3 || Cin == 24),
               "%s: built for 3x3 / stride 2 / pad 1, even H and W, 24 output and 3 or 24 input channels (got Cin=%d Cout=%d H=%d W=%d)", who,
               Cin, Cout, H, W);
  return 0;
}

extern "C" int rn_conv3x3s2_fwd(const float* x, const float* w, float* y, int N, int Cin, int Cout, int H, int W, void* stream) {
  if (int rc = cv_check("rn_conv3x3s2_fwd", x, w, y, N, Cin, Cout, H, W)) return rc;
  const long px = (long)N * (H / 2) * (W / 2);
  const int gx = (int)((px + CV_T - 1) / CV_T);
  // big layers: 24 channels per thread (fewest input reads); small ones: 8 per thread, 3x the waves
  if (Cin == 3) conv3x3s2_fwd_kernel<3, 24, 24><<<dim3(gx, 1), CV_T, 0, (hipStream_t)stream>>>(x, w, y, N, H, W);
  else if (px >= 200000) conv3x3s2_fwd_kernel<24, 24, 24><<<dim3(gx, 1), CV_T, 0, (hipStream_t)stream>>>(x, w, y, N, H, W);
  else conv3x3s2_fwd_kernel<24, 24, 8><<<dim3(gx, 3), CV_T, 0, (hipStream_t)stream>>>(x, w, y, N, H, W);
  RN_LAUNCH_CHECK("rn_conv3x3s2_fwd");
  return 0;
}

extern "C" int rn_conv3x3s2_bwd_data(const float* dy, const float* w, float* dx, int N, int Cin, int Cout, int H, int W, void* stream) {
  if (int rc = cv_check("rn_conv3x3s2_bwd_data", dy, w, dx, N, Cin, Cout, H, W)) return rc;
  RN_CHECK_ARG(Cin == 24 && (uintptr_t)dx % 8 == 0, "rn_conv3x3s2_bwd_data: built for 24 input channels");
  const long px = (long)N * (H / 2) * (W / 2);
  const int gx = (int)((px + CV_T - 1) / CV_T);
  conv3x3s2_bwd_data_kernel<24, 24, 8><<<dim3(gx, 3), CV_T, 0, (hipStream_t)stream>>>(dy, w, dx, N, H, W);
  RN_LAUNCH_CHECK("rn_conv3x3s2_bwd_data");
  return 0;
}
