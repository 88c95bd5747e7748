// The four 3x3 / stride-2 / pad-1 convolutions of the conv stack in front of the relation layer (reference
// model.py:13-20: Conv2d(3 -> 24) and three Conv2d(24 -> 24), 128x128 -> 8x8): forward and input-gradient as direct
// fp32 convolutions.  They are tiny (0.34 .. 0.68 GFLOP at B = 64) and the library kernels spend 25-45 us each on
// them -- 0.22 ms on the critical path of a 1.3 ms training step.  Here a thread owns one output pixel (forward) or
// one 2x2 input block (input gradient) for ALL channels: COUT (x 4) accumulators, the 3x3xCIN weights are broadcast
// from LDS (every lane reads the same 16 bytes), 9 * CIN * COUT FMAs per thread against 9 * CIN (4 * COUT) loads.
// The weight gradient is a (24 x 9 CIN) x positions product on the fp32 matrix pipe (conv3x3s2_wgrad_kernel below).
// NCHW fp32, H and W even.
#include <stdlib.h>

#include "rn_common.h"

namespace {
constexpr int CV_T = 256;
constexpr long CV_KS_MAX = 100000;   // up to this many output pixels (2x2 input blocks) the reduction axis is split over the waves
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

// ---- the small layers (16x16 and 8x8 outputs at the headline shape): with one thread per pixel there are only 16-64
// workgroups of single-wave-per-SIMD code whose channel loop is a serial chain of loads and FMAs (19 us for 0.04 GFLOP).
// Here the reduction axis is split over the 4 waves of a workgroup: 64 pixels x 4 channel quarters, partial sums meet in
// LDS, wave q finalises a quarter of the outputs.  4x the workgroups, a quarter of the chain per thread.
template <int CIN, int COUT, int CG>
__global__ __launch_bounds__(CV_T) void conv3x3s2_fwd_ks_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                float* __restrict__ y, int N, int H, int W) {
  constexpr int KS = 4, CQ = CIN / KS;
  static_assert(CIN % KS == 0 && CG % KS == 0, "quarters");
  __shared__ __attribute__((aligned(16))) float ws[CIN * 9 * CG];            // [ci][tap][co in group]
  __shared__ float red[KS][CG][64];
  const int g0 = blockIdx.y * CG;
  for (int i = threadIdx.x; i < CIN * 9 * CG; i += CV_T) {
    const int co = i % CG, ct = i / CG;
    ws[i] = w[(long)(g0 + co) * CIN * 9 + ct];
  }
  __syncthreads();
  const int OH = H >> 1, OW = W >> 1;
  const int lane = threadIdx.x & 63, kq = threadIdx.x >> 6;
  const long p = (long)blockIdx.x * 64 + lane;
  const bool live = p < (long)N * OH * OW;
  const long pp = live ? p : 0;
  const int ox = (int)(pp % OW), oy = (int)((pp / OW) % OH), n = (int)(pp / ((long)OW * OH));
  float acc[CG];
#pragma unroll
  for (int c = 0; c < CG; ++c) acc[c] = 0.f;
  const int iy0 = 2 * oy - 1, ix0 = 2 * ox - 1;
  const float* xn = x + ((long)n * CIN + kq * CQ) * H * W;
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
  float t[CQ][9];                                                            // all of this quarter's taps in flight at once
#pragma unroll
  for (int c = 0; c < CQ; ++c)
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) t[c][tp] = ok[tp] ? xn[(long)c * H * W + off[tp]] : 0.f;
#pragma unroll
  for (int c = 0; c < CQ; ++c) {
    const float* wc = ws + (kq * CQ + c) * 9 * CG;
#pragma unroll
    for (int tp = 0; tp < 9; ++tp)
#pragma unroll
      for (int o = 0; o < CG; o += 4) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(wc + tp * CG + o);
        acc[o] = fmaf(wv[0], t[c][tp], acc[o]);
        acc[o + 1] = fmaf(wv[1], t[c][tp], acc[o + 1]);
        acc[o + 2] = fmaf(wv[2], t[c][tp], acc[o + 2]);
        acc[o + 3] = fmaf(wv[3], t[c][tp], acc[o + 3]);
      }
  }
#pragma unroll
  for (int o = 0; o < CG; ++o) red[kq][o][lane] = acc[o];
  __syncthreads();
  if (live) {
#pragma unroll
    for (int o = kq * (CG / KS); o < (kq + 1) * (CG / KS); ++o)
      y[((long)n * COUT + g0 + o) * OH * OW + (long)oy * OW + ox] = ((red[0][o][lane] + red[1][o][lane]) + red[2][o][lane]) + red[3][o][lane];
  }
}

template <int CIN, int COUT, int CG>
__global__ __launch_bounds__(CV_T) void conv3x3s2_bwd_data_ks_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                                     float* __restrict__ dx, int N, int H, int W) {
  constexpr int KS = 4, CQ = COUT / KS;
  static_assert(COUT % KS == 0 && CG % KS == 0, "quarters");
  __shared__ __attribute__((aligned(16))) float ws[COUT * 9 * CG];           // [co][tap][ci in group]
  __shared__ float red[KS][4 * CG][64];
  const int g0 = blockIdx.y * CG;
  for (int i = threadIdx.x; i < COUT * 9 * CG; i += CV_T) {
    const int ci = i % CG, r = i / CG, tap = r % 9, co = r / 9;
    ws[i] = w[((long)co * CIN + g0 + ci) * 9 + tap];
  }
  __syncthreads();
  const int OH = H >> 1, OW = W >> 1;
  const int lane = threadIdx.x & 63, kq = threadIdx.x >> 6;
  const long p = (long)blockIdx.x * 64 + lane;
  const bool live = p < (long)N * OH * OW;
  const long pp = live ? p : 0;
  const int b = (int)(pp % OW), a = (int)((pp / OW) % OH), n = (int)(pp / ((long)OW * OH));
  float a00[CG], a01[CG], a10[CG], a11[CG];
#pragma unroll
  for (int c = 0; c < CG; ++c) a00[c] = a01[c] = a10[c] = a11[c] = 0.f;
  const bool ra = a + 1 < OH, rb = b + 1 < OW;
  const float* dn = dy + ((long)n * COUT + kq * CQ) * OH * OW + (long)a * OW + b;
  const int o01 = rb ? 1 : 0, o10 = ra ? OW : 0, o11 = (ra && rb) ? OW + 1 : 0;
  float d[CQ][4];
#pragma unroll
  for (int c = 0; c < CQ; ++c) {
    const float* dc = dn + (long)c * OH * OW;
    d[c][0] = dc[0];
    d[c][1] = rb ? dc[o01] : 0.f;
    d[c][2] = ra ? dc[o10] : 0.f;
    d[c][3] = (ra && rb) ? dc[o11] : 0.f;
  }
#pragma unroll
  for (int cc = 0; cc < CQ; ++cc) {
    const float d00 = d[cc][0], d01 = d[cc][1], d10 = d[cc][2], d11 = d[cc][3];
    const float* wc = ws + (kq * CQ + cc) * 9 * CG;
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
  }
#pragma unroll
  for (int c = 0; c < CG; ++c) {
    red[kq][4 * c + 0][lane] = a00[c]; red[kq][4 * c + 1][lane] = a01[c];
    red[kq][4 * c + 2][lane] = a10[c]; red[kq][4 * c + 3][lane] = a11[c];
  }
  __syncthreads();
  if (live) {
#pragma unroll
    for (int c = kq * (CG / KS); c < (kq + 1) * (CG / KS); ++c) {
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = ((red[0][4 * c + e][lane] + red[1][4 * c + e][lane]) + red[2][4 * c + e][lane]) + red[3][4 * c + e][lane];
      float* xc = dx + ((long)n * CIN + g0 + c) * H * W + (long)(2 * a) * W + 2 * b;
      *reinterpret_cast<u32x2*>(xc) = u32x2{__builtin_bit_cast(unsigned, v[0]), __builtin_bit_cast(unsigned, v[1])};
      *reinterpret_cast<u32x2*>(xc + W) = u32x2{__builtin_bit_cast(unsigned, v[2]), __builtin_bit_cast(unsigned, v[3])};
    }
  }
}

// ---- the large layers on the fp32 matrix pipe.  The convolution as a (24 x 9 CIN) x (9 CIN x pixels) product with
// v_mfma_f32_32x32x2_f32 (a k-ordered fp32 FMA chain per output, like the scalar loop): A = the weights (rows = output channels,
// padded to 32), B = the input taps of 32 consecutive output pixels of one row, D[co][pixel] -- a lane then owns ONE pixel and the
// stores of a channel are 128 contiguous bytes.  The reduction index is ordered (channel pair, tap) with the two k-values of an
// instruction = channels c and c + CP / 2 at the same tap: the per-lane LDS address of the tap is then a constant offset from a
// lane base (one ds_read_b32 with an immediate offset per MFMA), and a lane's weights (9 CP / 2 floats) stay in registers for
// the life of the workgroup.  Unit = (image, 4 output rows, 32-pixel segment): the 9 input rows x 65 columns of all channels
// are staged in LDS (zero padded), wave w computes output row w.  CP = channels padded to even (24 -> 24, 3 -> 4).
template <int CIN, int CP>
__global__ __launch_bounds__(256) void conv3x3s2_fwd_mfma_kernel(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y,
                                                                 int N, int H, int W, int units, int segs) {
  constexpr int NK = 9 * CP / 2, CS = 66, HALF = CP / 2;
  extern __shared__ __attribute__((aligned(16))) float cm_smem[];     // [CP][9][CS]
  const int OH = H >> 1, OW = W >> 1;
  const int t = threadIdx.x, l = t & 63, wv = t >> 6, p = l & 31, h = l >> 5;
  // this lane's weights: row co = p, k-values (channel cp + HALF h, tap)
  float wr[NK];
#pragma unroll
  for (int kk = 0; kk < NK; ++kk) {
    const int cp = kk / 9, tap = kk - 9 * cp, ci = cp + HALF * h;
    const bool ok = p < 24 && ci < CIN;
    const float u = w[((long)(ok ? p : 0) * CIN + (ok ? ci : 0)) * 9 + tap];
    wr[kk] = ok ? u : 0.f;
  }
  const int lane_base = (HALF * h * 9 + 2 * wv) * CS + 2 * p;
  const int rpi = (OH + 3) / 4;                                       // row blocks per image
  for (int u = blockIdx.x; u < units; u += gridDim.x) {
    const int seg = u % segs, rb = (u / segs) % rpi, n = u / (segs * rpi);
    const int oy0 = 4 * rb, ox0 = 32 * seg;
    const int iy0 = 2 * oy0 - 1, ix0 = 2 * ox0 - 1;                   // staged (row 0, column 0)
    __syncthreads();                                                  // the previous unit's reads are done
    // staging: wave wv takes rows wv, wv + 4, ..; lanes walk the columns; EVERY load of the unit is issued before the first LDS
    // write waits on one (unconditional loads from clamped addresses: one memory round trip per unit, not one per batch)
    constexpr int NR = (CP * 9 + 3) / 4;                                // rows per wave
    constexpr int NB = NR > 32 ? 4 : 1, RB = (NR + NB - 1) / NB;        // (24 channels: four batches of 14 loads -- the lane's 108 weights live in registers too)
#pragma unroll 1
    for (int b = 0; b < NB; ++b) {
      float v[RB];
#pragma unroll
      for (int q = 0; q < RB; ++q) {
        const int row = wv + 4 * (b * RB + q), ci = row / 9, iy = iy0 + (row - ci * 9), ix = ix0 + l;
        const bool ok = b * RB + q < NR && row < CP * 9 && ci < CIN && iy >= 0 && iy < H && ix >= 0 && ix < W;
        const int off = ok ? ((n * CIN + ci) * H + iy) * W + ix : 0;     // (32-bit element offsets: N Cin H W < 2^31, checked on the host)
        const float u = x[off];
        v[q] = ok ? u : 0.f;
      }
#pragma unroll
      for (int q = 0; q < RB; ++q) {
        const int row = wv + 4 * (b * RB + q);
        if (b * RB + q < NR && row < CP * 9) cm_smem[row * CS + l] = v[q];
      }
    }
    float e = 0.f;                                                      // column 64 of row t
    {
      const int row = t < CP * 9 ? t : 0, ci = row / 9, iy = iy0 + (row - ci * 9), ix = ix0 + 64;
      const bool ok = t < CP * 9 && ci < CIN && iy >= 0 && iy < H && ix < W;
      const float u = x[ok ? ((n * CIN + ci) * H + iy) * W + ix : 0];
      e = ok ? u : 0.f;
    }
    if (t < CP * 9) cm_smem[t * CS + 64] = e;
    __syncthreads();
    typedef __attribute__((ext_vector_type(16))) float f32x16_;
    f32x16_ acc[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[0][i] = acc[1][i] = 0.f;
    const float* xb = cm_smem + lane_base;
#pragma unroll
    for (int kk = 0; kk < NK; ++kk) {
      const int cp = kk / 9, tap = kk - 9 * cp, ky = tap / 3, kx = tap - 3 * ky;
      acc[kk & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wr[kk], xb[(cp * 9 + ky) * CS + kx], acc[kk & 1], 0, 0, 0);
    }
    const int oy = oy0 + wv, ox = ox0 + p;
    if (oy < OH && ox < OW) {
      // D row co = 8 (i / 4) + 4 h + i % 4 (i < 12: the 24 real channels), D column = this lane's pixel
      float* yo = y + (((long)n * 24) * OH + oy) * OW + ox;
#pragma unroll
      for (int i = 0; i < 12; ++i) yo[(long)(8 * (i >> 2) + 4 * h + (i & 3)) * OH * OW] = acc[0][i] + acc[1][i];
    }
  }
}

// ... the same product for 64 / 32 / 16-column inputs (layers 2..4 at the headline shape) with the input rows moved HBM -> LDS by
// LDS-DMA (global_load_lds_dwordx4: a wave instruction moves 1 KB = 4 / 8 / 16 rows, no registers, so all requests of a unit are
// in flight at once -- the register-staged version needs four round trips because a lane's 108 weights live in registers too).
// A wave's 32 pixels are RW = 32 / OW consecutive output rows (contiguous in y); a unit is 4 RW output rows of an image (the
// whole image when it has fewer).  Rows are packed at W floats: the taps left of column 0 and above row 0 are masked to zero
// instead of being stored.
template <int W>
__global__ __launch_bounds__(256) void conv3x3s2_fwd_mfma_dma_kernel(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y,
                                                                     int N, int H, int units, int RU) {
  constexpr int CIN = 24, NK = 9 * 12, HALF = 12, OW = W / 2, RW = 32 / OW, LPR = W / 4, RPI = 64 / LPR, PAD = 4;
  typedef __attribute__((address_space(3))) unsigned char lds_u8_;
  extern __shared__ __attribute__((aligned(16))) float cd_smem[];     // [PAD floats][24][RT rows][W]
  const int OH = H >> 1, RT = 2 * RU + 1, NROW = CIN * RT, NINST = (NROW + RPI - 1) / RPI;   // RU: output rows per unit (<= 4 RW)
  const int t = threadIdx.x, l = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6), p = l & 31, h = l >> 5;
  float wr[NK];
#pragma unroll
  for (int kk = 0; kk < NK; ++kk) {
    const int cp = kk / 9, tap = kk - 9 * cp, ci = cp + HALF * h;
    const float u = w[((long)(p < 24 ? p : 0) * CIN + ci) * 9 + tap];
    wr[kk] = p < 24 ? u : 0.f;
  }
  const int prow = p / OW, pcol = p - prow * OW;                      // this lane's pixel inside the wave's tile
  const unsigned lds0 = (unsigned)(size_t)(lds_u8_*)cd_smem + PAD * 4;
  const int rpi = OH / RU;
  const bool active = wv * RW < RU;
  for (int u = blockIdx.x; u < units; u += gridDim.x) {
    const int rb = u % rpi, n = u / rpi;
    const int oy0 = RU * rb, iy0 = 2 * oy0 - 1;
    __syncthreads();                                                  // the previous unit's reads are done
    const float* xn = x + (long)n * CIN * H * W;
    asm volatile("" : "+s"(xn));
    for (int inst = wv; inst < NINST; inst += 4) {
      int R = RPI * inst + l / LPR;
      R = R < NROW ? R : NROW - 1;                                    // (the last instruction's tail re-requests the last row)
      const int ci = R / RT, iy = iy0 + (R - ci * RT);
      const unsigned voff = (unsigned)((((ci * H + (iy < 0 ? 0 : (iy < H ? iy : H - 1))) * W) << 2) + ((l % LPR) << 4));
      const unsigned dst = lds0 + inst * 1024;
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep)
                   : "v"(voff), "s"(xn), "s"(dst)
                   : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (active) {
      typedef __attribute__((ext_vector_type(16))) float f32x16_;
      f32x16_ acc[2];
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[0][i] = acc[1][i] = 0.f;
      const int orow = wv * RW + prow;                                // output row inside the unit
      const float* xb = cd_smem + PAD + (HALF * h * RT + 2 * orow) * W + 2 * pcol - 1;
      const bool left = pcol == 0, top = oy0 + orow == 0;
#pragma unroll
      for (int kk = 0; kk < NK; ++kk) {
        const int cp = kk / 9, tap = kk - 9 * cp, ky = tap / 3, kx = tap - 3 * ky;
        float v = xb[(cp * RT + ky) * W + kx];
        if (kx == 0) v = left ? 0.f : v;                              // the column left of the image
        if (ky == 0) v = top ? 0.f : v;                               // the row above it
        acc[kk & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wr[kk], v, acc[kk & 1], 0, 0, 0);
      }
      float* yo = y + (((long)n * 24) * OH + oy0 + wv * RW) * OW + p;   // (the wave's RW rows are contiguous)
#pragma unroll
      for (int i = 0; i < 12; ++i) yo[(long)(8 * (i >> 2) + 4 * h + (i & 3)) * OH * OW] = acc[0][i] + acc[1][i];
    }
  }
}

// ... and the input gradient of that layer the same way.  By the parity of the input pixel (2a + s, 2b + t) the transposed
// convolution is four products with 1 / 2 / 2 / 4 taps (see conv3x3s2_bwd_data_kernel): D_class[ci][b] = sum over (co, tap in
// class) of w[co][ci][tap] * dy[co][a + da][b + db].  A = the weights (rows = input channels; a lane's 108 values in registers,
// k-pair = output channels c, c + 12 at one tap), B = dy taps of the 32 columns b of row a from LDS, four accumulator tiles per
// wave.  Unit = (image, 4 rows a): the 5 x 32 dy values of all 24 channels arrive by LDS-DMA (a wave instruction moves eight
// 128-byte rows); the column right of the image (lane b = 31, db = 1) is masked, the row below it zeroed by the requesting wave.
// A lane writes its 2 x 2 input block as two 8-byte stores per channel (256 contiguous bytes per half-wave).
// OW = dy columns (32 / 16 / 8: layers 2..4 at the headline shape): a wave's 32 lanes are RW = 32 / OW consecutive rows a; a unit is
// RU rows of an image (4 RW, or the whole image when it has fewer: then only RU / RW waves compute).  RU + 1 dy rows per channel are
// staged, packed at OW floats; the column right of the image and the row below it are masked in registers.
template <int OW>
__global__ __launch_bounds__(256) void conv3x3s2_bwd_data_mfma_dma_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                                          float* __restrict__ dx, int N, int OH, int units, int RU) {
  constexpr int C = 24, HALF = 12, NK = 9 * HALF, W = 2 * OW, RW = 32 / OW, LPR = OW / 4, RPI = 64 / LPR;
  typedef __attribute__((address_space(3))) unsigned char lds_u8_;
  extern __shared__ __attribute__((aligned(16))) float bd_smem[];     // [co][RU + 1 rows][OW] (+ the last request's tail)
  float* ds = bd_smem;
  const int H = 2 * OH, RT = RU + 1, NROW = C * RT, NINST = (NROW + RPI - 1) / RPI;
  const int t = threadIdx.x, l = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6), p = l & 31, h = l >> 5;
  float wr[NK];                                                       // w[co = cp + 12 h][ci = p][tap], index cp * 9 + tap
#pragma unroll
  for (int kk = 0; kk < NK; ++kk) {
    const int cp = kk / 9, tap = kk - 9 * cp;
    const float u = w[((long)(cp + HALF * h) * C + (p < C ? p : 0)) * 9 + tap];
    wr[kk] = p < C ? u : 0.f;
  }
  const int pr = p / OW, pb = p - pr * OW;                            // this lane's pixel inside the wave's tile
  const int arow = wv * RW + pr;                                      // its row a inside the unit
  const int lane_base = (HALF * h * RT + arow) * OW + pb;
  const unsigned lds0 = (unsigned)(size_t)(lds_u8_*)ds;
  const int rpi = OH / RU;
  const bool active = wv * RW < RU;
  typedef __attribute__((ext_vector_type(16))) float f32x16_;
  for (int u = blockIdx.x; u < units; u += gridDim.x) {
    const int rb = u % rpi, n = u / rpi, a0 = RU * rb;
    __syncthreads();
    const float* dn = dy + (long)n * C * OH * OW;
    asm volatile("" : "+s"(dn));
    for (int inst = wv; inst < NINST; inst += 4) {
      int R = RPI * inst + l / LPR;
      R = R < NROW ? R : NROW - 1;
      const int co = R / RT, a = a0 + (R - co * RT);
      const unsigned voff = (unsigned)((((co * OH + (a < OH ? a : OH - 1)) * OW) << 2) + ((l % LPR) << 4));
      const unsigned dst = lds0 + inst * 1024;
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep)
                   : "v"(voff), "s"(dn), "s"(dst)
                   : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (active) {
      f32x16_ ee, eo, oe, oo;
#pragma unroll
      for (int i = 0; i < 16; ++i) ee[i] = eo[i] = oe[i] = oo[i] = 0.f;
      const float* db_ = ds + lane_base;
      const int a = a0 + arow;
      const bool edge = pb == OW - 1, bot = a + 1 >= OH;               // b + 1 / a + 1 outside the image
#pragma unroll
      for (int cp = 0; cp < HALF; ++cp) {
        const float d00 = db_[(cp * RT) * OW];
        float d10 = db_[(cp * RT + 1) * OW], d01 = db_[(cp * RT) * OW + 1], d11 = db_[(cp * RT + 1) * OW + 1];
        d01 = edge ? 0.f : d01;
        d10 = bot ? 0.f : d10;
        d11 = (edge || bot) ? 0.f : d11;
        const float* wc = wr + cp * 9;
        ee = __builtin_amdgcn_mfma_f32_32x32x2f32(wc[4], d00, ee, 0, 0, 0);
        eo = __builtin_amdgcn_mfma_f32_32x32x2f32(wc[3], d01, eo, 0, 0, 0);
        oe = __builtin_amdgcn_mfma_f32_32x32x2f32(wc[1], d10, oe, 0, 0, 0);
        oo = __builtin_amdgcn_mfma_f32_32x32x2f32(wc[0], d11, oo, 0, 0, 0);
        eo = __builtin_amdgcn_mfma_f32_32x32x2f32(wc[5], d00, eo, 0, 0, 0);
        oe = __builtin_amdgcn_mfma_f32_32x32x2f32(wc[7], d00, oe, 0, 0, 0);
        oo = __builtin_amdgcn_mfma_f32_32x32x2f32(wc[2], d10, oo, 0, 0, 0);
        oo = __builtin_amdgcn_mfma_f32_32x32x2f32(wc[6], d01, oo, 0, 0, 0);
        oo = __builtin_amdgcn_mfma_f32_32x32x2f32(wc[8], d00, oo, 0, 0, 0);
      }
      float* xo = dx + (((long)n * C) * H + 2 * a) * W + 2 * pb;
#pragma unroll
      for (int i = 0; i < 12; ++i) {
        float* xc = xo + (long)(8 * (i >> 2) + 4 * h + (i & 3)) * H * W;
        // (the four values go through VGPRs explicitly: hipcc 7.2 otherwise forms the 8-byte store pairs inside the accumulator
        // registers and stores element 0's pair for a whole group of four rows)
        float v0 = ee[i], v1 = eo[i], v2 = oe[i], v3 = oo[i];
        asm volatile("" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
        *reinterpret_cast<u32x2*>(xc) = u32x2{__builtin_bit_cast(unsigned, v0), __builtin_bit_cast(unsigned, v1)};
        *reinterpret_cast<u32x2*>(xc + W) = u32x2{__builtin_bit_cast(unsigned, v2), __builtin_bit_cast(unsigned, v3)};
      }
    }
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
  // the small 24 -> 24 layers (32- and 16-column inputs): the LDS-DMA matrix-pipe kernel with 2 / 4 output rows per wave
  {
    bool small = Cin == 24 && (W == 32 || W == 16) && H == W && N >= 16 && (uintptr_t)x % 16 == 0;
    if (const char* e = rn_diag_env("RN_CONV_DMA_SMALL")) small = small && atoi(e) != 0;
    if (small) {
      const int OHs = H / 2, RU = W == 32 ? 8 : 8;                     // output rows per unit (W = 16: the whole 8-row image, two waves)
      const int units_s = N * (OHs / RU);
      const int rows_per_inst = 256 / W, nrow = 24 * (2 * RU + 1);     // (whole 1 KB instructions: the last one's tail needs room)
      const size_t shm = ((size_t)((nrow + rows_per_inst - 1) / rows_per_inst) * rows_per_inst * W + 4) * sizeof(float);
      if (W == 32) {
        (void)hipFuncSetAttribute((const void*)conv3x3s2_fwd_mfma_dma_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        conv3x3s2_fwd_mfma_dma_kernel<32><<<units_s, 256, shm, (hipStream_t)stream>>>(x, w, y, N, H, units_s, RU);
      } else {
        conv3x3s2_fwd_mfma_dma_kernel<16><<<units_s, 256, shm, (hipStream_t)stream>>>(x, w, y, N, H, units_s, RU);
      }
      RN_LAUNCH_CHECK("rn_conv3x3s2_fwd(mfma, small)");
      return 0;
    }
  }
  // (the 3-channel layer stays on the scalar kernel: K = 27 is too short for the staging to pay -- 15.6 us against 13.1)
  bool mfma = Cin == 24 && px >= 32768 && W >= 64 && (long)N * Cin * H * W < (1L << 31);
  if (const char* e = rn_diag_env("RN_CONV_MFMA")) mfma = mfma && atoi(e) != 0;      // (diagnostics builds: A/B against the scalar kernels)
  if (mfma) {
    // the large layers: fp32 matrix pipe (units of 4 output rows x 32 pixels; all channels of the 9 input rows in LDS)
    const int segs = (W / 2 + 31) / 32, units = N * ((H / 2 + 3) / 4) * segs;
    const int grid = units < 1024 ? units : 1024;
    bool dma = W == 64 && (H / 2) % 4 == 0 && (uintptr_t)x % 16 == 0;
    if (const char* e = rn_diag_env("RN_CONV_DMA")) dma = dma && atoi(e) != 0;
    if (dma) {
      const size_t shm = ((size_t)24 * 9 * 64 + 4) * sizeof(float);
      (void)hipFuncSetAttribute((const void*)conv3x3s2_fwd_mfma_dma_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
      conv3x3s2_fwd_mfma_dma_kernel<64><<<grid, 256, shm, (hipStream_t)stream>>>(x, w, y, N, H, units, 4);
    } else {
      const size_t shm = (size_t)24 * 9 * 66 * sizeof(float);
      (void)hipFuncSetAttribute((const void*)conv3x3s2_fwd_mfma_kernel<24, 24>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
      conv3x3s2_fwd_mfma_kernel<24, 24><<<grid, 256, shm, (hipStream_t)stream>>>(x, w, y, N, H, W, units, segs);
    }
    RN_LAUNCH_CHECK("rn_conv3x3s2_fwd(mfma)");
    return 0;
  }
  if (Cin == 3) conv3x3s2_fwd_kernel<3, 24, 24><<<dim3(gx, 1), CV_T, 0, (hipStream_t)stream>>>(x, w, y, N, H, W);
  else if (px >= 200000) conv3x3s2_fwd_kernel<24, 24, 24><<<dim3(gx, 1), CV_T, 0, (hipStream_t)stream>>>(x, w, y, N, H, W);
  else if (px >= CV_KS_MAX) conv3x3s2_fwd_kernel<24, 24, 8><<<dim3(gx, 3), CV_T, 0, (hipStream_t)stream>>>(x, w, y, N, H, W);
  else conv3x3s2_fwd_ks_kernel<24, 24, 8><<<dim3((int)((px + 63) / 64), 3), CV_T, 0, (hipStream_t)stream>>>(x, w, y, N, H, W);
  RN_LAUNCH_CHECK("rn_conv3x3s2_fwd");
  return 0;
}

extern "C" int rn_conv3x3s2_bwd_data(const float* dy, const float* w, float* dx, int N, int Cin, int Cout, int H, int W, void* stream) {
  if (int rc = cv_check("rn_conv3x3s2_bwd_data", dy, w, dx, N, Cin, Cout, H, W)) return rc;
  RN_CHECK_ARG(Cin == 24 && (uintptr_t)dx % 8 == 0, "rn_conv3x3s2_bwd_data: built for 24 input channels");
  const long px = (long)N * (H / 2) * (W / 2);
  const int gx = (int)((px + CV_T - 1) / CV_T);
  // the 32 x 32 -> 64 x 64 layer on the fp32 matrix pipe with LDS-DMA staging (register-staged it was 21.2 us against 24.0 for the
  // scalar kernel and nothing on the 14 x 14 grid's layers: only this shape is taken)
  bool mfma = W == 64 && (H / 2) % 4 == 0 && px >= 32768 && (uintptr_t)dy % 16 == 0;
  if (const char* e = rn_diag_env("RN_CONV_DMA")) mfma = mfma && atoi(e) != 0;
  // ... and the 16 x 16 / 8 x 8 gradients, 2 / 4 rows per wave: 8.4 / 7.3 us alone against the scalar kernels' 8.0 / 5.3, but +0.9 % on the
  // step -- 128 / 64 workgroups leave the chip to the streams beside the conv backward (the same trade as the forward's last layer)
  bool small = (W == 32 || W == 16) && H == W && N >= 16 && (uintptr_t)dy % 16 == 0;
  if (const char* e = rn_diag_env("RN_BD_DMA_SMALL")) small = small && atoi(e) != 0;
  if (mfma || small) {
    const int OH = H / 2, OW = W / 2, RU = OH < 4 * (32 / OW) ? OH : 4 * (32 / OW);
    const int units = N * (OH / RU), rpi = 256 / OW, nrow = 24 * (RU + 1);
    const size_t shm = ((size_t)((nrow + rpi - 1) / rpi) * rpi * OW + 4) * sizeof(float);
    const int grid = units < 1024 ? units : 1024;
    if (OW == 32) conv3x3s2_bwd_data_mfma_dma_kernel<32><<<grid, 256, shm, (hipStream_t)stream>>>(dy, w, dx, N, OH, units, RU);
    else if (OW == 16) conv3x3s2_bwd_data_mfma_dma_kernel<16><<<grid, 256, shm, (hipStream_t)stream>>>(dy, w, dx, N, OH, units, RU);
    else conv3x3s2_bwd_data_mfma_dma_kernel<8><<<grid, 256, shm, (hipStream_t)stream>>>(dy, w, dx, N, OH, units, RU);
    RN_LAUNCH_CHECK("rn_conv3x3s2_bwd_data(mfma)");
    return 0;
  }
  if (px >= CV_KS_MAX) conv3x3s2_bwd_data_kernel<24, 24, 8><<<dim3(gx, 3), CV_T, 0, (hipStream_t)stream>>>(dy, w, dx, N, H, W);
  else conv3x3s2_bwd_data_ks_kernel<24, 24, 8><<<dim3((int)((px + 63) / 64), 3), CV_T, 0, (hipStream_t)stream>>>(dy, w, dx, N, H, W);
  RN_LAUNCH_CHECK("rn_conv3x3s2_bwd_data");
  return 0;
}

// ------------------------------------------------------------------------------------------------ weight gradient
// dw[co][ci][ky][kx] = sum_{n,oy,ox} dy[n][co][oy][ox] * x[n][ci][2 oy + ky - 1][2 ox + kx - 1]
// = a (24 x 9 CIN) x (N Ho Wo) matrix product with the positions as the reduction axis: v_mfma_f32_32x32x2_f32 (a
// k-ordered fp32 FMA chain per output, like the scalar loop), M = co padded to 32, N = (ci, ky, kx) in tiles of 32,
// K = two neighbouring output pixels per instruction.  The library needs 80-90 us per layer for this (layout transposes
// + an implicit-GEMM kernel); the last one sits at the very end of the backward pass.
// Unit of work = (image, 4 output rows): the 9 input rows (zero padded, column index + 1) and the 4 dy rows of all channels
// are staged in LDS row by row (a wave per row: no index arithmetic per element).  One A read (dy) and one B read (an x
// tap: a per-lane gather at a compile-time offset per step) per MFMA.
//   CIN = 24: 7 waves, wave = one 32-column tile of (ci, ky, kx) over all 4 rows -- nothing to combine across waves;
//   CIN = 3 : 4 waves, wave = one output row of the single tile; the four partial sums meet in LDS at the end.
// Blocks are persistent over units; one partial per block goes to the workspace and conv_wgrad_reduce_kernel sums the
// partials in a fixed order.
// BN (with BN = true): dy is the gradient of relu(batch_norm(conv)) and the kernel forms the convolution's output gradient from it
// and the conv output while staging, dconv = gamma invstd (dz - S1/n - (xc - mean) invstd dgamma / n) with dz = dy where the block's
// output was > 0 -- cn_bwd_apply_kernel's expressions (rn_convnorm.hip), so the first block's dconv (25 MB written and read back
// at the very end of the backward pass) never goes to memory.  S1, S2 come from that file's pass 1, summed here in its slice order.
namespace {
typedef __attribute__((ext_vector_type(16))) float cv_f32x16;
struct CwBn {
  const float* xc;                                      // the convolution's output (N, 24, Ho, Wo)
  const double* part;                                   // [24][S][2] slice sums of pass 1
  int S;
  double count;
  const float *mean, *invstd, *gamma, *beta;
  float *dgamma, *dbeta, *zero_out;
};
}
template <int CIN, bool BN>
__global__ __launch_bounds__(CIN == 3 ? 256 : 448) void conv3x3s2_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                               float* __restrict__ part, int H, int W, int units, int cpi, CwBn bn) {
  constexpr int NC = CIN * 9, NW = CIN == 3 ? 4 : 7;
  constexpr bool NSPLIT = CIN != 3;                     // waves split the (ci, ky, kx) tiles, not the rows
  extern __shared__ __attribute__((aligned(16))) float cw_smem[];
  const int Ho = H / 2, Wo = W / 2, CS = W + 2, DS = 4 * Wo + 1;
  float* xs = cw_smem;                                  // [CIN][9][CS]
  float* dys = cw_smem + CIN * 9 * CS;                  // [32][DS]
  const int t = threadIdx.x, l = t & 63, w = t >> 6, ncol = l & 31, half = l >> 5;
  const int col = NSPLIT ? 32 * w + ncol : ncol;
  const int boff = col < NC ? (col / 9) * 9 * CS + ((col % 9) / 3) * CS + col % 3 + 2 * half : 0;
  const int aoff = ncol * DS + half;
  for (int i = t; i < CIN * 9 * CS + 32 * DS; i += NW * 64) cw_smem[i] = 0.f;      // pads (and channels 24..31 of dy) stay zero
  __shared__ float coef[24][5];                         // sc, sh, mean, S1/n, invstd dgamma / n
  if constexpr (BN) {
    __shared__ double sums[24][32][2];
    for (int i = t; i < 24 * bn.S * 2; i += NW * 64) sums[i / (2 * bn.S)][(i / 2) % bn.S][i & 1] = bn.part[i];   // one round trip
    __syncthreads();
    if (t < 24) {
      double s1 = 0.0, s2 = 0.0;
      for (int i = 0; i < bn.S; ++i) {
        s1 += sums[t][i][0];
        s2 += sums[t][i][1];
      }
      const float m = bn.mean[t], is = bn.invstd[t], ga = bn.gamma[t];
      const float dga = (float)((double)is * (s2 - (double)m * s1));
      const float sc = ga * is;
      coef[t][0] = sc;
      coef[t][1] = bn.beta[t] - m * sc;
      coef[t][2] = m;
      coef[t][3] = (float)(s1 / bn.count);
      coef[t][4] = (float)((double)dga * is / bn.count);
      if (blockIdx.x == 0) {
        bn.dgamma[t] = dga;
        bn.dbeta[t] = (float)s1;
        if (bn.zero_out) bn.zero_out[t] = 0.f;
      }
    }
  }
  cv_f32x16 acc[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[0][i] = acc[1][i] = 0.f;
  for (int u = blockIdx.x; u < units; u += gridDim.x) {
    const int n = u / cpi, oy0 = 4 * (u - n * cpi);
    __syncthreads();                                    // the previous unit's reads (or the zero fill) are done
    // rows in batches of 8 per wave: the 8 loads are issued back to back (one HBM round trip per batch, not per row)
    for (int row0 = w; row0 < CIN * 9; row0 += 8 * NW)
      for (int c = l; c < W; c += 64) {
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int row = row0 + q * NW, ci = row / 9, iy = 2 * oy0 - 1 + (row - ci * 9);
          const bool ok = row < CIN * 9 && iy >= 0 && iy < H;
          v[q] = ok ? x[(((long)n * CIN + ci) * H + iy) * W + c] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q)
          if (row0 + q * NW < CIN * 9) xs[(row0 + q * NW) * CS + 1 + c] = v[q];
      }
    for (int row0 = w; row0 < 24 * 4; row0 += 8 * NW)
      for (int c = l; c < Wo; c += 64) {
        float v[8], xv[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int row = row0 + q * NW, co = row >> 2, rr = row & 3;
          const bool ok = row < 24 * 4 && oy0 + rr < Ho;
          const long a = (((long)n * 24 + co) * Ho + oy0 + rr) * Wo + c;
          v[q] = ok ? dy[a] : 0.f;
          if constexpr (BN) xv[q] = ok ? bn.xc[a] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int row = row0 + q * NW;
          if (row < 24 * 4) {
            float o = v[q];
            if constexpr (BN) {
              const float* cf = coef[row >> 2];
              const float dz = (xv[q] * cf[0] + cf[1] > 0.f) ? v[q] : 0.f;
              o = cf[0] * (dz - cf[3] - (xv[q] - cf[2]) * cf[4]);
              o = oy0 + (row & 3) < Ho ? o : 0.f;
            }
            dys[(row >> 2) * DS + (row & 3) * Wo + c] = o;
          }
        }
      }
    __syncthreads();
    if constexpr (NSPLIT) {
#pragma unroll
      for (int rr = 0; rr < 4; ++rr)
#pragma unroll 4
        for (int ox0 = 0; ox0 < Wo; ox0 += 4) {         // two independent accumulators: even / odd pixel pairs
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(dys[aoff + rr * Wo + ox0], xs[boff + 2 * rr * CS + 2 * ox0], acc[0], 0, 0, 0);
          if (ox0 + 2 < Wo)
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(dys[aoff + rr * Wo + ox0 + 2], xs[boff + 2 * rr * CS + 2 * ox0 + 4], acc[1], 0, 0, 0);
        }
    } else {
#pragma unroll 4
      for (int ox0 = 0; ox0 < Wo; ox0 += 4) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(dys[aoff + w * Wo + ox0], xs[boff + 2 * w * CS + 2 * ox0], acc[0], 0, 0, 0);
        if (ox0 + 2 < Wo)
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(dys[aoff + w * Wo + ox0 + 2], xs[boff + 2 * w * CS + 2 * ox0 + 4], acc[1], 0, 0, 0);
      }
    }
  }
  // D row co = 8 (i / 4) + 4 half + i % 4 (i < 12 for the 24 real channels), D column = this lane's (ci, ky, kx)
  float* dst = part + (long)blockIdx.x * 24 * NC;
  if constexpr (NSPLIT) {
    if (col < NC) {
#pragma unroll
      for (int i = 0; i < 12; ++i) dst[(8 * (i >> 2) + 4 * half + (i & 3)) * NC + col] = acc[0][i] + acc[1][i];
    }
  } else {
    __syncthreads();
    float* red = cw_smem;                               // [4 waves][12][64]
#pragma unroll
    for (int i = 0; i < 12; ++i) red[(w * 12 + i) * 64 + l] = acc[0][i] + acc[1][i];
    __syncthreads();
    for (int idx = t; idx < 12 * 64; idx += 256) {
      const int i = idx >> 6, ln = idx & 63, c = ln & 31, co = 8 * (i >> 2) + 4 * (ln >> 5) + (i & 3);
      if (c < NC) dst[co * NC + c] = ((red[idx] + red[12 * 64 + idx]) + red[2 * 12 * 64 + idx]) + red[3 * 12 * 64 + idx];
    }
  }
}

// The first layer at the headline shape (3 x 128 x 128 images) with every row of a unit moved HBM -> LDS by LDS-DMA: the
// register-staged kernel above spends its time in 4-5 dependent load round trips per unit (19 us for 38 MB); here all ~40-60
// 1 KB requests of a unit are in flight at once.  x rows are packed at 128 floats, two per request, each request 2 floats
// further on (bank spread for the tap gather); the tap left of column 0 and the row above row 0 are masked.  A channel's 4 dy
// rows are contiguous in memory = one request, at 257-float pitch.  BN: the conv output comes in the same way and one pass
// over LDS turns dy into dconv in place.
template <bool BN>
__global__ __launch_bounds__(256) void conv3x3s2_wgrad1_dma_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                   float* __restrict__ part, int H, int units, int cpi, CwBn bn) {
  constexpr int CIN = 3, NC = 27, W = 128, Wo = 64, PAD = 4, XP = 258, NXI = 14, DS = 257;
  constexpr int XS = PAD + NXI * XP, DYS = 32 * DS;     // floats: [x rows][dy: 32 channels][conv output: 24 channels]
  typedef __attribute__((address_space(3))) unsigned char lds_u8_;
  extern __shared__ __attribute__((aligned(16))) float cw_smem[];
  float* xs = cw_smem;
  float* dys = cw_smem + XS;
  float* xcs = dys + DYS;
  __shared__ float coef[24][5];                         // sc, sh, mean, S1/n, invstd dgamma / n
  const int Ho = H / 2;
  const int t = threadIdx.x, l = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6), ncol = l & 31, half = l >> 5;
  const int colc = ncol < NC ? ncol : 0, ci = colc / 9, ky = (colc % 9) / 3, kx = colc % 3;
  const int xr = ci * 9 + 2 * w + ky;                   // this lane's tap row for the wave's output row
  const int boff = PAD + (xr >> 1) * XP + (xr & 1) * W + kx - 1 + 2 * half;
  const int aoff = ncol * DS + half + w * Wo;
  const bool dead = ncol >= NC, edge = half == 0 && kx == 0;
  for (int i = t; i < 8 * DS; i += 256) dys[24 * DS + i] = 0.f;       // channels 24..31 of the A operand
  if constexpr (BN) {
    double* sums = (double*)xcs;                        // [24][S][2], before the first unit uses the region
    for (int i = t; i < 24 * bn.S * 2; i += 256) sums[i] = bn.part[i];
    __syncthreads();
    if (t < 24) {
      double s1 = 0.0, s2 = 0.0;
      for (int i = 0; i < bn.S; ++i) {
        s1 += sums[(t * bn.S + i) * 2];
        s2 += sums[(t * bn.S + i) * 2 + 1];
      }
      const float m = bn.mean[t], is = bn.invstd[t], ga = bn.gamma[t];
      const float dga = (float)((double)is * (s2 - (double)m * s1));
      const float sc = ga * is;
      coef[t][0] = sc;
      coef[t][1] = bn.beta[t] - m * sc;
      coef[t][2] = m;
      coef[t][3] = (float)(s1 / bn.count);
      coef[t][4] = (float)((double)dga * is / bn.count);
      if (blockIdx.x == 0) {
        bn.dgamma[t] = dga;
        bn.dbeta[t] = (float)s1;
        if (bn.zero_out) bn.zero_out[t] = 0.f;
      }
    }
  }
  const unsigned lds_x = (unsigned)(size_t)(lds_u8_*)xs + PAD * 4, lds_dy = (unsigned)(size_t)(lds_u8_*)dys, lds_xc = (unsigned)(size_t)(lds_u8_*)xcs;
  cv_f32x16 acc[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[0][i] = acc[1][i] = 0.f;
  for (int u = blockIdx.x; u < units; u += gridDim.x) {
    const int n = __builtin_amdgcn_readfirstlane(u / cpi), oy0 = __builtin_amdgcn_readfirstlane(4 * (u - n * cpi));
    __syncthreads();                                    // the previous unit's reads (or the prologue) are done
    {
      const float* xn = x + (long)n * CIN * H * W;
      asm volatile("" : "+s"(xn));
      for (int inst = w; inst < NXI; inst += 4) {
        int r = 2 * inst + (l >> 5);
        r = r < 27 ? r : 26;
        const int c = r / 9, iy = 2 * oy0 - 1 + (r - 9 * c);
        const unsigned voff = (unsigned)((((c * H + (iy < 0 ? 0 : iy)) * W) << 2) + ((l & 31) << 4));
        const unsigned dst = lds_x + inst * (XP * 4);
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(xn), "s"(dst) : "memory");
      }
      const long doff = ((long)n * 24 * Ho + oy0) * Wo;
      const float* dn = dy + doff;
      asm volatile("" : "+s"(dn));
      const unsigned lane_off = (unsigned)(l << 4);
      for (int co = w; co < 24; co += 4) {
        const unsigned voff = (unsigned)((co * Ho * Wo) << 2) + lane_off;
        const unsigned dst = lds_dy + co * (DS * 4);
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(dn), "s"(dst) : "memory");
      }
      if constexpr (BN) {
        const float* cn = bn.xc + doff;
        asm volatile("" : "+s"(cn));
        for (int co = w; co < 24; co += 4) {
          const unsigned voff = (unsigned)((co * Ho * Wo) << 2) + lane_off;
          const unsigned dst = lds_xc + co * (DS * 4);
          unsigned keep;
          asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                       : "=&s"(keep) : "v"(voff), "s"(cn), "s"(dst) : "memory");
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if constexpr (BN) {
#pragma unroll 8
      for (int co = 0; co < 24; ++co) {
        const float g = dys[co * DS + t], v = xcs[co * DS + t];
        const float* cf = coef[co];
        const float dz = (v * cf[0] + cf[1] > 0.f) ? g : 0.f;
        dys[co * DS + t] = cf[0] * (dz - cf[3] - (v - cf[2]) * cf[4]);
      }
      __syncthreads();
    }
    const bool zall = dead || (ky == 0 && oy0 == 0 && w == 0);        // (the row above the image)
#pragma unroll 4
    for (int ox0 = 0; ox0 < Wo; ox0 += 4) {
      float b0 = xs[boff + 2 * ox0], b1 = xs[boff + 2 * ox0 + 4];
      if (ox0 == 0) b0 = edge ? 0.f : b0;                            // (the column left of the image)
      b0 = zall ? 0.f : b0;
      b1 = zall ? 0.f : b1;
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(dys[aoff + ox0], b0, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(dys[aoff + ox0 + 2], b1, acc[1], 0, 0, 0);
    }
  }
  // D row co = 8 (i / 4) + 4 half + i % 4 (i < 12 for the 24 real channels), D column = this lane's (ci, ky, kx)
  float* dst = part + (long)blockIdx.x * 24 * NC;
  __syncthreads();
  float* red = cw_smem;                                 // [4 waves][12][64]
#pragma unroll
  for (int i = 0; i < 12; ++i) red[(w * 12 + i) * 64 + l] = acc[0][i] + acc[1][i];
  __syncthreads();
  for (int idx = t; idx < 12 * 64; idx += 256) {
    const int i = idx >> 6, ln = idx & 63, c = ln & 31, co = 8 * (i >> 2) + 4 * (ln >> 5) + (i & 3);
    if (c < NC) dst[co * NC + c] = ((red[idx] + red[12 * 64 + idx]) + red[2 * 12 * 64 + idx]) + red[3 * 12 * 64 + idx];
  }
}

// ... and the 24 -> 24 layers (64- / 32- / 16-column inputs) the same way: 256 / W x rows and 128 / W channels of dy rows per 1-KB
// request, 7 waves = the 7 column tiles of (ci, ky, kx) as in the register-staged kernel.  (The 64-column layer's weight gradient
// is the last thing on the side stream of the backward pass: 31 us there, staged through registers.)
template <int W>
__global__ __launch_bounds__(448) void conv3x3s2_wgrad24_dma_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                    float* __restrict__ part, int H, int units, int cpi) {
  constexpr int CIN = 24, NC = 216, Wo = W / 2, PAD = 4, RPI = 256 / W, CPI = 128 / W, XP = 258, DP = 257;
  constexpr int NXI = (NC + RPI - 1) / RPI, NDI = 24 / CPI, XS = PAD + NXI * XP;
  typedef __attribute__((address_space(3))) unsigned char lds_u8_;
  extern __shared__ __attribute__((aligned(16))) float cw_smem[];
  float* xs = cw_smem;
  float* dys = cw_smem + XS;                            // [32 / CPI requests][DP]: channels 24..31 stay zero
  const int Ho = H / 2;
  const int t = threadIdx.x, l = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6), ncol = l & 31, half = l >> 5;
  const int col = 32 * w + ncol, colc = col < NC ? col : 0, ci = colc / 9, ky = (colc % 9) / 3, kx = colc % 3;
  int boff[4];
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int r = ci * 9 + 2 * rr + ky;
    boff[rr] = PAD + (r / RPI) * XP + (r % RPI) * W + kx - 1 + 2 * half;
  }
  const int aoff = (ncol / CPI) * DP + (ncol % CPI) * (4 * Wo) + half;
  const bool dead = col >= NC, edge = half == 0 && kx == 0;
  for (int i = t; i < (32 / CPI - NDI) * DP; i += 448) dys[NDI * DP + i] = 0.f;
  const unsigned lds_x = (unsigned)(size_t)(lds_u8_*)xs + PAD * 4, lds_dy = (unsigned)(size_t)(lds_u8_*)dys;
  cv_f32x16 acc[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[0][i] = acc[1][i] = 0.f;
  for (int u = blockIdx.x; u < units; u += gridDim.x) {
    const int n = __builtin_amdgcn_readfirstlane(u / cpi), oy0 = __builtin_amdgcn_readfirstlane(4 * (u - n * cpi));
    __syncthreads();                                    // the previous unit's reads (or the zero fill) are done
    {
      const float* xn = x + (long)n * CIN * H * W;
      asm volatile("" : "+s"(xn));
      for (int inst = w; inst < NXI; inst += 7) {
        int r = RPI * inst + l / (W / 4);
        r = r < NC ? r : NC - 1;
        const int c = r / 9, iy = 2 * oy0 - 1 + (r - 9 * c);
        const unsigned voff = (unsigned)((((c * H + (iy < 0 ? 0 : iy)) * W) << 2) + ((l % (W / 4)) << 4));
        const unsigned dst = lds_x + inst * (XP * 4);
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(xn), "s"(dst) : "memory");
      }
      const float* dn = dy + ((long)n * 24 * Ho + oy0) * Wo;
      asm volatile("" : "+s"(dn));
      for (int inst = w; inst < NDI; inst += 7) {
        const int c = CPI * inst + l / (W / 2);
        const unsigned voff = (unsigned)((c * Ho * Wo) << 2) + (unsigned)((l % (W / 2)) << 4);
        const unsigned dst = lds_dy + inst * (DP * 4);
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(dn), "s"(dst) : "memory");
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const bool zall = dead || (ky == 0 && oy0 == 0 && rr == 0);      // (the row above the image)
#pragma unroll 4
      for (int ox0 = 0; ox0 < Wo; ox0 += 4) {
        float b0 = xs[boff[rr] + 2 * ox0], b1 = xs[boff[rr] + 2 * ox0 + 4];
        if (ox0 == 0) b0 = edge ? 0.f : b0;                            // (the column left of the image)
        b0 = zall ? 0.f : b0;
        b1 = zall ? 0.f : b1;
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(dys[aoff + rr * Wo + ox0], b0, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(dys[aoff + rr * Wo + ox0 + 2], b1, acc[1], 0, 0, 0);
      }
    }
  }
  float* dst = part + (long)blockIdx.x * 24 * NC;
  if (col < NC) {
#pragma unroll
    for (int i = 0; i < 12; ++i) dst[(8 * (i >> 2) + 4 * half + (i & 3)) * NC + col] = acc[0][i] + acc[1][i];
  }
}

// (the layer-1 reduction is the last kernel before the optimiser: 32 outputs x 8 slices per block, 16 loads in flight per thread --
// 4 dependent round trips over 512 partials instead of 16)
__global__ __launch_bounds__(256) void conv_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int nout, int nparts) {
  __shared__ float red[7][32];
  const int o = blockIdx.x * 32 + (threadIdx.x & 31), s = threadIdx.x >> 5;
  float a = 0.f;
  if (o < nout) {
#pragma unroll 16
    for (int b = s; b < nparts; b += 8) a += part[(long)b * nout + o];
  }
  if (s) red[s - 1][threadIdx.x & 31] = a;
  __syncthreads();
  if (s == 0 && o < nout) {
#pragma unroll
    for (int q = 0; q < 7; ++q) a += red[q][threadIdx.x];
    dw[o] = a;
  }
}

namespace {
struct CwPlan { int units, cpi, grid; size_t shm; };
static CwPlan cw_plan(int N, int Cin, int H, int W) {
  CwPlan p;
  const int Ho = H / 2, Wo = W / 2;
  p.cpi = (Ho + 3) / 4;
  p.units = N * p.cpi;
  p.grid = p.units < 512 ? p.units : 512;
  p.shm = ((size_t)Cin * 9 * (W + 2) + 32 * (4 * Wo + 1)) * sizeof(float);
  if (p.shm < 4 * 12 * 64 * sizeof(float)) p.shm = 4 * 12 * 64 * sizeof(float);
  return p;
}
}  // namespace

size_t rnws_conv_bwd_weight(int N, int Cin, int H, int W) {
  if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0) return 0;
  return (size_t)cw_plan(N, Cin, H, W).grid * 24 * Cin * 9 * sizeof(float);
}

namespace {
static bool cw_dma24_on() {
  if (const char* e = rn_diag_env("RN_WGRAD24_DMA")) return atoi(e) != 0;
  return true;
}
template <bool BN>
static int cw_launch(const char* who, const float* x, const float* dy, float* dw, void* ws, int N, int Cin, int H, int W, const CwBn& bn, hipStream_t s) {
  const CwPlan p = cw_plan(N, Cin, H, W);
  RN_CHECK_ARG(p.shm <= 150 * 1024, "%s: W=%d needs %zu bytes of LDS", who, W, p.shm);
  float* part = (float*)ws;
  bool dma = Cin == 3 && W == 128 && H % 8 == 0 && ((uintptr_t)x | (uintptr_t)dy | (uintptr_t)bn.xc) % 16 == 0;
  if (const char* e = rn_diag_env("RN_WGRAD1_DMA")) dma = dma && atoi(e) != 0;
  if (dma) {
    const size_t shm = (size_t)(4 + 14 * 258 + 32 * 257 + (BN ? 24 * 257 : 0)) * sizeof(float);
    (void)hipFuncSetAttribute((const void*)conv3x3s2_wgrad1_dma_kernel<BN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    conv3x3s2_wgrad1_dma_kernel<BN><<<p.grid, 256, shm, s>>>(x, dy, part, H, p.units, p.cpi, bn);
  } else if (!BN && Cin == 24 && H == W && (W == 64 || W == 32 || W == 16) && ((uintptr_t)x | (uintptr_t)dy) % 16 == 0 && cw_dma24_on()) {
    const int rpi = 256 / W, cpi_d = 128 / W;
    const size_t shm = (size_t)(4 + ((216 + rpi - 1) / rpi) * 258 + (32 / cpi_d) * 257) * sizeof(float);
#define RN_CW24(W_)                                                                                                                          \
  {                                                                                                                                          \
    (void)hipFuncSetAttribute((const void*)conv3x3s2_wgrad24_dma_kernel<W_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);          \
    conv3x3s2_wgrad24_dma_kernel<W_><<<p.grid, 448, shm, s>>>(x, dy, part, H, p.units, p.cpi);                                               \
  }
    if (W == 64) RN_CW24(64) else if (W == 32) RN_CW24(32) else RN_CW24(16)
#undef RN_CW24
  } else if (Cin == 3) {
    if (p.shm > 48 * 1024) (void)hipFuncSetAttribute((const void*)conv3x3s2_wgrad_kernel<3, BN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.shm);
    conv3x3s2_wgrad_kernel<3, BN><<<p.grid, 256, p.shm, s>>>(x, dy, part, H, W, p.units, p.cpi, bn);
  } else {
    if (p.shm > 48 * 1024) (void)hipFuncSetAttribute((const void*)conv3x3s2_wgrad_kernel<24, BN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.shm);
    conv3x3s2_wgrad_kernel<24, BN><<<p.grid, 448, p.shm, s>>>(x, dy, part, H, W, p.units, p.cpi, bn);
  }
  RN_LAUNCH_CHECK(who);
  const int nout = 24 * Cin * 9;
  conv_wgrad_reduce_kernel<<<(nout + 31) / 32, 256, 0, s>>>(part, dw, nout, p.grid);
  RN_LAUNCH_CHECK("rn_conv3x3s2_bwd_weight(reduce)");
  return 0;
}
}  // namespace

extern "C" int rn_conv3x3s2_bwd_weight(const float* x, const float* dy, float* dw, void* ws, int N, int Cin, int Cout, int H, int W,
                                       void* stream) {
  if (int rc = cv_check("rn_conv3x3s2_bwd_weight", x, dy, dw, N, Cin, Cout, H, W)) return rc;
  RN_CHECK_ARG(ws, "rn_conv3x3s2_bwd_weight: NULL workspace");
  return cw_launch<false>("rn_conv3x3s2_bwd_weight", x, dy, dw, ws, N, Cin, H, W, CwBn{}, (hipStream_t)stream);
}

extern "C" int rn_bn_relu_bwd_conv_wgrad(const float* dy, const float* xc, const float* inp, const float* gamma, const float* beta,
                                         const float* mean, const float* invstd, float* dgamma, float* dbeta, float* zero_out, float* dw,
                                         void* ws_bn, void* ws_conv, int N, int Cin, int H, int W, void* stream) {
  if (int rc = cv_check("rn_bn_relu_bwd_conv_wgrad", inp, dy, dw, N, Cin, 24, H, W)) return rc;
  RN_CHECK_ARG(xc && gamma && beta && mean && invstd && dgamma && dbeta && ws_bn && ws_conv, "rn_bn_relu_bwd_conv_wgrad: NULL argument");
  CwBn bn;
  bn.xc = xc; bn.part = (const double*)ws_bn; bn.count = (double)N * (H / 2) * (W / 2);
  bn.mean = mean; bn.invstd = invstd; bn.gamma = gamma; bn.beta = beta;
  bn.dgamma = dgamma; bn.dbeta = dbeta; bn.zero_out = zero_out;
  if (int rc = rn_cn_launch_bwd_sums(dy, xc, mean, invstd, gamma, beta, ws_bn, N, 24, (H / 2) * (W / 2), stream, &bn.S)) return rc;
  return cw_launch<true>("rn_bn_relu_bwd_conv_wgrad", inp, dy, dw, ws_conv, N, Cin, H, W, bn, (hipStream_t)stream);
}
