// Fused R-CBIR feature extraction (reference extract.py:49-74).
//
// The reference hooks the INPUT of g layer k -- a (B n^2, in_k) fp32 matrix: 402 MB at B = 64 for the injected layer of the "IR"
// models -- strips the question columns, L2-normalises every pair row (F.normalize, eps 1e-12) and reduces max / mean over each
// question's n^2 pairs (extract.py:64-71).  Here that matrix never exists: a workgroup takes a tile of 64 pair rows (one (b, i), 64
// objects j), builds its [x_j | x_i] rows in LDS, runs g layers 0 .. k-1 on them with the activation tile RESIDENT in LDS (two
// padded 64 x 256 fp32 buffers, ping-pong), normalises the rows of the last tile and leaves ONE partial (max, sum) row pair per
// tile; rn_extract finishes per question.  Written per call: B n ceil(n / 64) x 2 F floats (8 MB at B = 64, n = 64, F = 256).
//
// Arithmetic: fp32 on the matrix pipe (v_mfma_f32_32x32x2_f32 = exact fmaf chains, 157 TFLOP/s peak): features are a retrieval
// signature and the op is not on the training path -- parity with the reference (1e-5) is worth more here than the 16-bit rate.
//   * un-swapped operands, D[row][feature]: wave w owns output features 64 w .. 64 w + 63 of all 64 rows (2 x 2 accumulators);
//     A = activations from LDS (row stride 257 floats: lane = row -> conflict-free ds_read_b32), B = weights straight from
//     L2 through a TRANSPOSED fp32 image Wt[k][feature] (128 contiguous bytes per half wave and k), eight k-pairs requested ahead;
//   * the question enters as a per-question bias row (W_l[:, -Q:] q_b + b_l, prepared by the caller: one small rn_gemm_f32) -- at
//     layer 0 for the original models, at layer 2 for the IR models (model.py:131-142) -- so no layer ever sees a K of 180 / 384.
#include "rn_common.h"

namespace {
constexpr int XT = 64;                                   // pair rows per tile
constexpr int XG = 256;                                  // g width
constexpr int XS = XG + 1;                               // LDS row stride (floats)
constexpr int XBUF = XT * XS;                            // floats per buffer
constexpr int XL_MAX = 4;

struct XArgs {
  const float* Wt[XL_MAX];                               // layer l: (K_l, 256) fp32, Wt[k][f] = W_l[f][k] (object / activation columns only)
  const float* bias[XL_MAX];                             // (256) fp32, or (B, 256) when per_q[l]
  int per_q[XL_MAX];
};

__global__ __launch_bounds__(256) void extract_tile_kernel(const float* __restrict__ x, long sxb, long sxn, long sxk, XArgs a, int nlayers,
                                                           int n, int k, int F, float* __restrict__ pmax, float* __restrict__ psum,
                                                           int tiles_per_i) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* buf[2] = {lds, lds + XBUF};
  float* inv_s = lds + 2 * XBUF;                         // 64 reciprocal-free row norms
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, ln = lane & 31, h = lane >> 5;
  const int tile = blockIdx.x;
  const int jc = tile % tiles_per_i, bi = tile / tiles_per_i, b = bi / n, i = bi - b * n;
  const int j0 = jc * XT, nv = (n - j0) < XT ? (n - j0) : XT;                 // valid rows of this tile
  const int K0 = 2 * k, K0p = (K0 + 15) / 16 * 16;                            // layer-0 reduction length, padded to whole groups of 8 k-pairs

  // ---- the tile's pair rows [x_j | x_i | 0 ...] (model.py:117-127), invalid rows all zero
  float* cur = buf[1];
  float* nxt = buf[0];
  for (int e = t; e < XT * K0p; e += 256) {
    const int r = e / K0p, c = e - r * K0p;
    float v = 0.f;
    if (r < nv && c < K0) {
      const int obj = c < k ? j0 + r : i, cc = c < k ? c : c - k;
      v = x[b * sxb + obj * sxn + cc * sxk];
    }
    cur[r * XS + c] = v;
  }
  __syncthreads();

  // ---- g layers 0 .. nlayers-1 on the resident tile
  for (int l = 0; l < nlayers; ++l) {
    const int K = l == 0 ? K0 : XG, Kp = l == 0 ? K0p : XG;
    const float* Wt = a.Wt[l] + 64 * w + ln;
    f32x16 acc[2][2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[mb][nb][e] = 0.f;
    // eight k-pairs of weights in flight while the previous eight feed the matrix pipe (rows beyond K: clamped -- they meet zeros)
    float wv[2][8][2];
    auto wload = [&](int slot, int s8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        int kk = 2 * (s8 + u) + h;
        kk = kk < K ? kk : K - 1;
        wv[slot][u][0] = Wt[(long)kk * XG];
        wv[slot][u][1] = Wt[(long)kk * XG + 32];
      }
    };
    const int nsteps = Kp / 2;
    wload(0, 0);
    for (int s8 = 0; s8 < nsteps; s8 += 16) {
      if (s8 + 8 < nsteps) wload(1, s8 + 8);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const float a0 = cur[ln * XS + 2 * (s8 + u) + h], a1 = cur[(32 + ln) * XS + 2 * (s8 + u) + h];
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, wv[0][u][0], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, wv[0][u][1], acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, wv[0][u][0], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, wv[0][u][1], acc[1][1], 0, 0, 0);
      }
      if (s8 + 8 < nsteps) {
        if (s8 + 16 < nsteps) wload(0, s8 + 16);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const float a0 = cur[ln * XS + 2 * (s8 + 8 + u) + h], a1 = cur[(32 + ln) * XS + 2 * (s8 + 8 + u) + h];
          acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, wv[1][u][0], acc[0][0], 0, 0, 0);
          acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, wv[1][u][1], acc[0][1], 0, 0, 0);
          acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, wv[1][u][0], acc[1][0], 0, 0, 0);
          acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, wv[1][u][1], acc[1][1], 0, 0, 0);
        }
      }
    }
    // epilogue: bias (this question's row where the question is injected), ReLU (model.py:141-145), into the other buffer.
    // D[row 32 mb + 8 j + 4 h + r][feature 64 w + 32 nb + ln]; invalid rows stay zero.
    const float* bl = a.bias[l] + (a.per_q[l] ? (long)b * XG : 0) + 64 * w + ln;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      const float bv = bl[32 * nb];
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = 32 * mb + 8 * (e >> 2) + 4 * h + (e & 3);
          nxt[row * XS + 64 * w + 32 * nb + ln] = row < nv ? fmaxf(acc[mb][nb][e] + bv, 0.f) : 0.f;
        }
    }
    __syncthreads();
    float* sw = cur; cur = nxt; nxt = sw;
  }

  // ---- F.normalize(row, p=2, eps=1e-12) over the first F columns, then max / sum over the tile's valid rows (extract.py:68-71)
  {
    const int r = t >> 2, part = t & 3;
    float ss = 0.f;
    for (int c = part; c < F; c += 4) { const float v = cur[r * XS + c]; ss = fmaf(v, v, ss); }
    ss += __shfl_xor(ss, 1);
    ss += __shfl_xor(ss, 2);
    if (part == 0) inv_s[r] = fmaxf(sqrtf(ss), 1e-12f);
  }
  __syncthreads();
  for (int c = t; c < F; c += 256) {
    float mx = -3.0e38f, sm = 0.f;
    for (int r = 0; r < nv; ++r) {
      const float u = cur[r * XS + c] / inv_s[r];
      mx = fmaxf(mx, u);
      sm += u;
    }
    pmax[(long)tile * F + c] = mx;
    psum[(long)tile * F + c] = sm;
  }
}

// per question: max / mean over its n * tiles_per_i tile partials, in tile order (deterministic)
__global__ __launch_bounds__(256) void extract_finish_kernel(const float* __restrict__ pmax, const float* __restrict__ psum,
                                                             float* __restrict__ maxf, float* __restrict__ avgf, int tiles_per_q, int F,
                                                             float inv_pairs) {
  const int b = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
  if (c >= F) return;
  const float* pm = pmax + (long)b * tiles_per_q * F + c;
  const float* ps = psum + (long)b * tiles_per_q * F + c;
  float mx = -3.0e38f, sm = 0.f;
  int tq = 0;
  for (; tq + 8 <= tiles_per_q; tq += 8) {
    float m8[8], s8[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { m8[u] = pm[(long)(tq + u) * F]; s8[u] = ps[(long)(tq + u) * F]; }
#pragma unroll
    for (int u = 0; u < 8; ++u) { mx = fmaxf(mx, m8[u]); sm += s8[u]; }
  }
  for (; tq < tiles_per_q; ++tq) { mx = fmaxf(mx, pm[(long)tq * F]); sm += ps[(long)tq * F]; }
  maxf[(long)b * F + c] = mx;
  avgf[(long)b * F + c] = sm * inv_pairs;
}
}  // namespace

size_t rnws_extract(int B, int n, int F) {
  return (B > 0 && n > 0 && F > 0) ? (size_t)2 * B * n * cdiv(n, XT) * F * sizeof(float) : 0;
}

extern "C" int rn_extract_features(const float* x, long sxb, long sxn, long sxk, const float* const* Wt, const float* const* bias,
                                   const int* bias_per_question, int nlayers, int F, float* maxf, float* avgf, void* ws, int B, int n, int k,
                                   void* stream) {
  RN_CHECK_ARG(x && maxf && avgf && ws && B > 0 && n > 0 && k > 0, "rn_extract_features: bad pointer/size");
  RN_CHECK_ARG(nlayers >= 0 && nlayers <= XL_MAX && 2 * k <= XG, "rn_extract_features: nlayers=%d must be in [0, %d], 2k=%d <= %d", nlayers, XL_MAX, 2 * k, XG);
  RN_CHECK_ARG(nlayers == 0 ? (F > 0 && F <= 2 * k) : (F > 0 && F <= XG), "rn_extract_features: F=%d out of range for the input of layer %d", F, nlayers);
  RN_CHECK_ARG(nlayers == 0 || (Wt && bias && bias_per_question), "rn_extract_features: layer tables are NULL");
  XArgs a;
  memset(&a, 0, sizeof(a));
  for (int l = 0; l < nlayers; ++l) {
    RN_CHECK_ARG(Wt[l] && bias[l], "rn_extract_features: layer %d weight / bias is NULL", l);
    a.Wt[l] = Wt[l];
    a.bias[l] = bias[l];
    a.per_q[l] = bias_per_question[l] != 0;
  }
  const int tpi = cdiv(n, XT);
  const long tiles = (long)B * n * tpi;
  RN_CHECK_ARG(tiles < (1l << 31), "rn_extract_features: too many tiles");
  float* pmax = (float*)ws;
  float* psum = pmax + tiles * F;
  const size_t shm = (size_t)(2 * XBUF + XT) * sizeof(float);
  RN_LDS_OPT_IN(extract_tile_kernel, "rn_extract_features");
  hipStream_t s = (hipStream_t)stream;
  extract_tile_kernel<<<(int)tiles, 256, shm, s>>>(x, sxb, sxn, sxk, a, nlayers, n, k, F, pmax, psum, tpi);
  RN_LAUNCH_CHECK("rn_extract_features(tiles)");
  extract_finish_kernel<<<dim3(cdiv(F, 256), B), 256, 0, s>>>(pmax, psum, maxf, avgf, n * tpi, F, 1.f / ((float)n * (float)n));
  RN_LAUNCH_CHECK("rn_extract_features(finish)");
  return 0;
}
