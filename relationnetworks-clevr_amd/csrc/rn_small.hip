// K4 -- small exact-fp32 kernels around the pair path: f_phi (model.py:155-160) forward and
// backward, log_softmax (model.py:162) and the (B*n x G) tail GEMMs of the pair backward.
// These are ~0.01 % of the step's flops; they run on the fp32 MFMA
// (v_mfma_f32_32x32x2_f32 = a k-ordered fp32 fmaf chain) so f_phi stays bit-comparable with an
// fp32 reference, one wave per 32x32 output tile, operands straight from global (L2-resident).
#include "rn_common.h"
#include "../../include/rn_hip_debug.h"

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

// ------------------------------------------------------------------------------------------------ fused f_phi
// f_phi (model.py:155-162) is three tiny matrix products on B rows (17 MFLOP at B = 64): its cost is the number
// of dependent launches on the critical path between the forward and the backward chain, not arithmetic.  One
// launch forward, two backward, plain fp32 FMA (k-ordered sums):
//   forward : f1 = relu(xg W1^T + b1); f2 = relu((f1 W2^T + b2) * mask); out = log_softmax(f2 W3^T + b3)
//   backward: dz3 = gout - exp(out) * sum(gout);  dz2 = (dz3 W3) * mask * (f2 > 0);  dz1 = (dz2 W2) * (f1 > 0);
//             dxg = dz1 W1;   dW_l = dz_l^T act_{l-1}, db_l = colsum(dz_l)
namespace {
constexpr int FP_RB = 4;          // rows per workgroup
constexpr int FP_MAXW = 1024;     // widest activation the LDS staging holds
constexpr int FP_KS = 4;          // k-slices per output feature with transposed weights (block = FP_KS * 256 threads)
static_assert(FP_KS == FP_RB, "the slice-k thread finalises row k");
constexpr int FP_KQ = 16;         // ... and with 16-byte column loads (N % 4 == 0): 16 k-slices x 64 four-column lanes, the same 1024 threads
static_assert(FP_KQ * 64 == FP_KS * 256 && FP_RB * 256 == FP_KS * 256, "one block shape for both thread maps");
// Labels outside [0, A) are CLAMPED to the nearest class, in every kernel that reads a label (these fused ones, the
// stand-alone nll kernels, the LSTM's token lookup): a device kernel cannot raise like F.nll_loss / nn.Embedding do
// without a host synchronisation per step.  The host wrappers document it; train.py's load_tensor_data produces
// 0-based labels in range by construction (utils.py:149).  There is no ignore_index.
__device__ __forceinline__ int fp_label(long long l, int A) { return l < 0 ? 0 : (l >= A ? A - 1 : (int)l); }

// acc[r] += sum_k W[f][k] * in[r][k]; thread = output feature, W (out, in) row-major; the rows' inputs sit in LDS
__device__ __forceinline__ void fp_rows(const float* __restrict__ W, const float* in_s, int K, int f, float (&acc)[FP_RB]) {
  const float* wr = W + (long)f * K;
  for (int k = 0; k < K; k += 4) {
    const f32x4 w = *reinterpret_cast<const f32x4*>(wr + k);
#pragma unroll
    for (int r = 0; r < FP_RB; ++r) {
      const f32x4 x = *reinterpret_cast<const f32x4*>(in_s + r * FP_MAXW + k);
      acc[r] = fmaf(w[3], x[3], fmaf(w[2], x[2], fmaf(w[1], x[1], fmaf(w[0], x[0], acc[r]))));
    }
  }
}
// acc[r] += sum_{i0 <= i < i1} in[r][i] * W[i][j]; thread = column j (coalesced over the workgroup), W (in, out)
// row-major.  The loop is a chain of L2 round trips: 32 independent loads are issued before their FMAs.
__device__ __forceinline__ void fp_cols(const float* __restrict__ W, const float* in_s, int i0, int i1, int J, int j, float (&acc)[FP_RB]) {
  int i = i0;
  for (; i + 32 <= i1; i += 32) {
    float w[32];
#pragma unroll
    for (int u = 0; u < 32; ++u) w[u] = W[(long)(i + u) * J + j];
#pragma unroll
    for (int u = 0; u < 32; u += 4) {
#pragma unroll
      for (int r = 0; r < FP_RB; ++r) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(in_s + r * FP_MAXW + i + u);
        acc[r] = fmaf(x[3], w[u + 3], fmaf(x[2], w[u + 2], fmaf(x[1], w[u + 1], fmaf(x[0], w[u], acc[r]))));
      }
    }
  }
  for (; i + 4 <= i1; i += 4) {
    float w[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) w[u] = W[(long)(i + u) * J + j];
#pragma unroll
    for (int r = 0; r < FP_RB; ++r) {
      const f32x4 x = *reinterpret_cast<const f32x4*>(in_s + r * FP_MAXW + i);
      acc[r] = fmaf(x[3], w[3], fmaf(x[2], w[2], fmaf(x[1], w[1], fmaf(x[0], w[0], acc[r]))));
    }
  }
  for (; i < i1; ++i) {
    const float w = W[(long)i * J + j];
#pragma unroll
    for (int r = 0; r < FP_RB; ++r) acc[r] = fmaf(in_s[r * FP_MAXW + i], w, acc[r]);
  }
}

// acc[r][0..3] += sum_{i0 <= i < i1} in[r][i] * W[i][j .. j+3]: thread = FOUR columns, one 16-byte load per weight row -- a wave
// fetches a whole 1-KB row of a 256-wide layer per instruction, and a thread's share of the layer (K / 16 rows) is ONE batch of
// independent loads: one L2 round trip per layer instead of two, a quarter of the load instructions.
__device__ __forceinline__ void fp_cols4(const float* __restrict__ W, const float* in_s, int i0, int i1, int J, int j, f32x4 (&acc)[FP_RB]) {
  int i = i0;
  for (; i + 16 <= i1; i += 16) {
    f32x4 w[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) w[u] = *reinterpret_cast<const f32x4*>(W + (long)(i + u) * J + j);
#pragma unroll
    for (int u = 0; u < 16; u += 4) {
#pragma unroll
      for (int r = 0; r < FP_RB; ++r) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(in_s + r * FP_MAXW + i + u);
#pragma unroll
        for (int c = 0; c < 4; ++c)
          acc[r][c] = fmaf(x[3], w[u + 3][c], fmaf(x[2], w[u + 2][c], fmaf(x[1], w[u + 1][c], fmaf(x[0], w[u][c], acc[r][c]))));
      }
    }
  }
  for (; i + 4 <= i1; i += 4) {
    f32x4 w[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) w[u] = *reinterpret_cast<const f32x4*>(W + (long)(i + u) * J + j);
#pragma unroll
    for (int r = 0; r < FP_RB; ++r) {
      const f32x4 x = *reinterpret_cast<const f32x4*>(in_s + r * FP_MAXW + i);
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[r][c] = fmaf(x[3], w[3][c], fmaf(x[2], w[2][c], fmaf(x[1], w[1][c], fmaf(x[0], w[0][c], acc[r][c]))));
    }
  }
  for (; i < i1; ++i) {
    const f32x4 w = *reinterpret_cast<const f32x4*>(W + (long)i * J + j);
#pragma unroll
    for (int r = 0; r < FP_RB; ++r) {
      const float x = in_s[r * FP_MAXW + i];
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[r][c] = fmaf(x, w[c], acc[r][c]);
    }
  }
}

// One layer on the block's FP_RB rows: v[r][f] = bias[f] + sum_k in[r][k] * Wop[k][f], handed to epi(r, f, v).
// TR (transposed weights, (in, out) row-major): blockDim = FP_KS * 256, thread (ks, f) sums its quarter of k, the
// partials meet in LDS and thread (ks, f) finalises row ks -- four times fewer dependent L2 round trips per layer.
// !TR: blockDim = 256, thread = feature over the whole k range.
template <bool TR, class Epi>
__device__ __forceinline__ void fp_layer_pass(const float* __restrict__ W, const float* __restrict__ bias, const float* in_s, int K,
                                              int N, float* red, Epi epi) {
  const int tf = threadIdx.x & 255, ks = threadIdx.x >> 8;
  for (int fb = 0; fb < N; fb += 256) {
    const int f = fb + tf;
    float acc[FP_RB];
#pragma unroll
    for (int r = 0; r < FP_RB; ++r) acc[r] = 0.f;
    if (TR && (N & 3) == 0) {
      // 16 k-slices x 64 lanes of four columns (thread t: slice t / 64, columns 4 (t % 64) ..); the 16 partials of an output meet
      // in LDS and thread (row t / 256, feature t % 256) adds them in slice order
      // (the slice is wave-uniform: row addresses are scalar, the lane adds one constant column offset)
      const int j4 = threadIdx.x & 63, kq = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), f0 = fb + 4 * j4;
      const int chunk = ((K + FP_KQ - 1) / FP_KQ + 3) & ~3, i0 = min(K, kq * chunk), i1 = min(K, i0 + chunk);
      f32x4 a4[FP_RB];
#pragma unroll
      for (int r = 0; r < FP_RB; ++r) a4[r] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (f0 < N) fp_cols4(W, in_s, i0, i1, N, f0, a4);
#pragma unroll
      for (int r = 0; r < FP_RB; ++r) *reinterpret_cast<f32x4*>(red + (kq * FP_RB + r) * 256 + 4 * j4) = a4[r];
      __syncthreads();
      if (f < N) {
        float v = red[ks * 256 + tf];
#pragma unroll
        for (int q = 1; q < FP_KQ; ++q) v += red[(q * FP_RB + ks) * 256 + tf];
        epi(ks, f, v + (bias ? bias[f] : 0.f));
      }
      __syncthreads();
    } else if constexpr (TR) {
      const int chunk = ((K + FP_KS - 1) / FP_KS + 3) & ~3, i0 = min(K, ks * chunk), i1 = min(K, i0 + chunk);
      if (f < N) fp_cols(W, in_s, i0, i1, N, f, acc);
#pragma unroll
      for (int r = 0; r < FP_RB; ++r) red[(ks * FP_RB + r) * 256 + tf] = acc[r];
      __syncthreads();
      if (f < N) {
        float v = red[ks * 256 + tf];
#pragma unroll
        for (int q = 1; q < FP_KS; ++q) v += red[(q * FP_RB + ks) * 256 + tf];
        epi(ks, f, v + (bias ? bias[f] : 0.f));
      }
      __syncthreads();
    } else if (f < N) {
      fp_rows(W, in_s, K, f, acc);
      const float b = bias ? bias[f] : 0.f;
#pragma unroll
      for (int r = 0; r < FP_RB; ++r) epi(r, f, acc[r] + b);
    }
  }
}
}  // namespace

// Layer 3 + log_softmax (+ the mean NLL of the batch) on the block's FP_RB rows: sa = the rows of f2 in LDS, sb <- the logits.
// Shared by the one-launch kernel below and by the head launch of the wide path.
template <bool TR>
__device__ __forceinline__ void fp_head(const float* __restrict__ W3, const float* __restrict__ b3, const float* sa, float* sb, float* red,
                                        float* lrow, float* __restrict__ out, int B, int F2, int A, int r0,
                                        const long long* __restrict__ label, float* __restrict__ loss, float* loss_part,
                                        unsigned* done_count) {
  const int t = threadIdx.x;
  fp_layer_pass<TR>(W3, b3, sa, F2, A, red, [&](int r, int f, float z) { sb[r * FP_MAXW + f] = z; });
  __syncthreads();
  if (t < FP_RB && r0 + t < B) {                       // log_softmax of one row (A <= 1024 logits in LDS)
    const float* z = sb + t * FP_MAXW;
    float mx = z[0];
    for (int a = 1; a < A; ++a) mx = fmaxf(mx, z[a]);
    float s = 0.f;
    for (int a = 0; a < A; ++a) s += expf(z[a] - mx);
    const float ls = mx + logf(s);
    for (int a = 0; a < A; ++a) out[(long)(r0 + t) * A + a] = z[a] - ls;
    if (label) lrow[t] = -(z[fp_label(label[r0 + t], A)] - ls);
  } else if (t < FP_RB) {
    lrow[t] = 0.f;
  }
  // mean NLL of the batch (train.py:41) in the same launch: block partials in a fixed order, combined by whichever
  // block finishes last (also in a fixed order: deterministic); the counter re-arms itself for the next launch
  if (label) {
    __shared__ int last_s;
    __syncthreads();
    if (t == 0) {
      loss_part[blockIdx.x] = ((lrow[0] + lrow[1]) + lrow[2]) + lrow[3];
      __threadfence();
      last_s = atomicAdd(done_count, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (last_s && t < 64) {
      // the last block adds the block partials in block order; up to 64 of them are fetched by one wave at once (one round trip
      // instead of a chain of them on the tail of the launch) and read back lane by lane: the same sum as the plain loop
      __threadfence();
      float tot = 0.f;
      for (unsigned i0 = 0; i0 < gridDim.x; i0 += 64) {
        const unsigned i = i0 + t;
        const float v = i < gridDim.x ? reinterpret_cast<volatile float*>(loss_part)[i] : 0.f;
        const unsigned cnt = gridDim.x - i0 < 64u ? gridDim.x - i0 : 64u;
        for (unsigned j = 0; j < cnt; ++j) tot += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), (int)j));
      }
      if (t == 0) {
        *loss = tot / (float)B;
        *done_count = 0u;
      }
    }
  }
}

// The backward dz chain of the SAME rows in the same launch (training step with the loss folded in: the log-prob gradient is
// -1/B at the label whatever the rest of the step does, so nothing has to be waited for): W1..3 natural (out, in) weights (NULL:
// no tail), dz3 / dz2 / dz1 (B x A / F2 / F1) and dxg (B x G) outputs -- for d loss = 1; rn_f_phi_bwd_grads finishes the job.
struct FpBwdTail {
  const float *W1, *W2, *W3;
  float *dz3, *dz2, *dz1, *dxg;
  __host__ __device__ FpBwdTail() : W1(nullptr), W2(nullptr), W3(nullptr), dz3(nullptr), dz2(nullptr), dz1(nullptr), dxg(nullptr) {}
};

// TR: W_l are given TRANSPOSED, (in, out) row-major -- the thread-per-output-feature walk is then coalesced
template <bool TR>
__global__ __launch_bounds__(TR ? FP_KS * 256 : 256) void f_phi_fwd_kernel(
    const float* __restrict__ xg, const float* __restrict__ W1, const float* __restrict__ b1, const float* __restrict__ W2,
    const float* __restrict__ b2, const float* __restrict__ W3, const float* __restrict__ b3, const float* __restrict__ mask,
    float* __restrict__ f1, float* __restrict__ f2, float* __restrict__ out, int B, int G, int F1, int F2, int A,
    const long long* __restrict__ label = nullptr, float* __restrict__ loss = nullptr, float* loss_part = nullptr,
    unsigned* done_count = nullptr, const float* __restrict__ xg_part = nullptr, int parts = 0, float* __restrict__ xg_out = nullptr,
    FpBwdTail bt = FpBwdTail()) {
  __shared__ __attribute__((aligned(16))) float sa[FP_RB * FP_MAXW], sb[FP_RB * FP_MAXW], red[FP_KQ * FP_RB * 256];
  __shared__ float lrow[FP_RB];
  const int t = threadIdx.x, r0 = blockIdx.x * FP_RB;
  if (xg_part) {
    // the pair sum (model.py:151-152) of this block's rows from the forward chain's per-tile partial rows -- xg_part[(b * parts + p)][G],
    // added in the order p = 0, 1, ... (deterministic) -- instead of a reduction launch of its own in front of this one (16 partial
    // rows per question at the headline shape: 64 KB per block).  xg_out gets the sums: the backward pass reads them.
    for (int c = t; c < FP_RB * G; c += blockDim.x) {
      const int r = c / G, k = c - r * G;
      float v = 0.f;
      if (r0 + r < B) {
        const float* src = xg_part + (long)(r0 + r) * parts * G + k;
        int p = 0;
        for (; p + 32 <= parts; p += 32) {             // 32 loads in flight: a batch is one round trip to L2 / HBM
          float u[32];
#pragma unroll
          for (int i = 0; i < 32; ++i) u[i] = src[(long)(p + i) * G];
#pragma unroll
          for (int i = 0; i < 32; ++i) v += u[i];
        }
        for (; p + 8 <= parts; p += 8) {
          float u[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) u[i] = src[(long)(p + i) * G];
#pragma unroll
          for (int i = 0; i < 8; ++i) v += u[i];
        }
        for (; p < parts; ++p) v += src[(long)p * G];
        xg_out[(long)(r0 + r) * G + k] = v;
      }
      sa[r * FP_MAXW + k] = v;
    }
  } else {
    for (int c = t; c < FP_RB * G; c += blockDim.x) {
      const int r = c / G, k = c - r * G;
      sa[r * FP_MAXW + k] = (r0 + r < B) ? xg[(long)(r0 + r) * G + k] : 0.f;
    }
  }
  __syncthreads();
  fp_layer_pass<TR>(W1, b1, sa, G, F1, red, [&](int r, int f, float z) {
    const float v = fmaxf(z, 0.f);
    sb[r * FP_MAXW + f] = v;
    if (r0 + r < B) f1[(long)(r0 + r) * F1 + f] = v;
  });
  __syncthreads();
  fp_layer_pass<TR>(W2, b2, sb, F1, F2, red, [&](int r, int f, float z) {
    const bool ok = r0 + r < B;
    const float m = (mask && ok) ? mask[(long)(r0 + r) * F2 + f] : 1.f;
    const float v = fmaxf(z * m, 0.f);
    sa[r * FP_MAXW + f] = v;
    if (ok) f2[(long)(r0 + r) * F2 + f] = v;
  });
  __syncthreads();
  fp_head<TR>(W3, b3, sa, sb, red, lrow, out, B, F2, A, r0, label, loss, loss_part, done_count);
  if constexpr (TR) {
    if (bt.W1 && label) {
      // ---- backward dz chain of this block's rows (what f_phi_bwd_dz_kernel does from global memory in a launch of its own)
      __syncthreads();                                   // (sa: layer 3's input has been read; sb: the logits have been read)
      if (t < FP_RB) {
        const int b = r0 + t;
        const float gl = -1.f / (float)B;
        const int lb = b < B ? fp_label(label[b], A) : -1;
        const float* z = sb + t * FP_MAXW;
        float mx = z[0];
        for (int a = 1; a < A; ++a) mx = fmaxf(mx, z[a]);
        float ssum = 0.f;
        for (int a = 0; a < A; ++a) ssum += expf(z[a] - mx);
        const float ls = mx + logf(ssum);
        for (int a = 0; a < A; ++a) {
          // exp of the STORED log-prob (z - ls rounded to fp32), exactly as the stand-alone kernel computes it
          const float v = b < B ? ((a == lb ? gl : 0.f) - expf(z[a] - ls) * gl) : 0.f;
          sa[t * FP_MAXW + a] = v;
          if (b < B) bt.dz3[(long)b * A + a] = v;
        }
      }
      __syncthreads();
      fp_layer_pass<true>(bt.W3, nullptr, sa, A, F2, red, [&](int r, int j, float z) {
        float v = 0.f;
        if (r0 + r < B) {
          const long o = (long)(r0 + r) * F2 + j;
          v = (f2[o] > 0.f) ? z * (mask ? mask[o] : 1.f) : 0.f;
          bt.dz2[o] = v;
        }
        sb[r * FP_MAXW + j] = v;
      });
      __syncthreads();
      fp_layer_pass<true>(bt.W2, nullptr, sb, F2, F1, red, [&](int r, int j, float z) {
        float v = 0.f;
        if (r0 + r < B) {
          const long o = (long)(r0 + r) * F1 + j;
          v = (f1[o] > 0.f) ? z : 0.f;
          bt.dz1[o] = v;
        }
        sa[r * FP_MAXW + j] = v;
      });
      __syncthreads();
      fp_layer_pass<true>(bt.W1, nullptr, sa, F1, G, red, [&](int r, int j, float z) {
        if (r0 + r < B) bt.dxg[(long)(r0 + r) * G + j] = z;
      });
    }
  }
}

// W_l natural (out, in) row-major: exactly the (in, out) operand of the backward products
__global__ __launch_bounds__(FP_KS * 256) void f_phi_bwd_dz_kernel(const float* __restrict__ gout, const float* __restrict__ out,
                                                           const float* __restrict__ f2, const float* __restrict__ f1,
                                                           const float* __restrict__ W1, const float* __restrict__ W2,
                                                           const float* __restrict__ W3, const float* __restrict__ mask,
                                                           float* __restrict__ dz3, float* __restrict__ dz2, float* __restrict__ dz1,
                                                           float* __restrict__ dxg, int B, int G, int F1, int F2, int A,
                                                           const long long* __restrict__ label = nullptr,
                                                           const float* __restrict__ gloss = nullptr) {
  __shared__ __attribute__((aligned(16))) float sa[FP_RB * FP_MAXW], sb[FP_RB * FP_MAXW], red[FP_KQ * FP_RB * 256];
  const int t = threadIdx.x, r0 = blockIdx.x * FP_RB;
  if (t < FP_RB) {
    const int b = r0 + t;
    // label mode: gout = d(mean NLL)/d out = -gloss / B at the label, 0 elsewhere -- never materialised
    const float gl = label ? -gloss[0] / (float)B : 0.f;
    const int lb = (label && b < B) ? fp_label(label[b], A) : -1;
    float s = label ? gl : 0.f;
    if (b < B && !label)
      for (int a = 0; a < A; ++a) s += gout[(long)b * A + a];
    for (int a = 0; a < A; ++a) {
      const float go = label ? (a == lb ? gl : 0.f) : (b < B ? gout[(long)b * A + a] : 0.f);
      const float v = (b < B) ? go - expf(out[(long)b * A + a]) * s : 0.f;
      sa[t * FP_MAXW + a] = v;
      if (b < B) dz3[(long)b * A + a] = v;
    }
  }
  __syncthreads();
  fp_layer_pass<true>(W3, nullptr, sa, A, F2, red, [&](int r, int j, float z) {
    float v = 0.f;
    if (r0 + r < B) {
      const long o = (long)(r0 + r) * F2 + j;
      v = (f2[o] > 0.f) ? z * (mask ? mask[o] : 1.f) : 0.f;
      dz2[o] = v;
    }
    sb[r * FP_MAXW + j] = v;
  });
  __syncthreads();
  fp_layer_pass<true>(W2, nullptr, sb, F2, F1, red, [&](int r, int j, float z) {
    float v = 0.f;
    if (r0 + r < B) {
      const long o = (long)(r0 + r) * F1 + j;
      v = (f1[o] > 0.f) ? z : 0.f;
      dz1[o] = v;
    }
    sa[r * FP_MAXW + j] = v;
  });
  __syncthreads();
  fp_layer_pass<true>(W1, nullptr, sa, F1, G, red, [&](int r, int j, float z) {
    if (r0 + r < B) dxg[(long)(r0 + r) * G + j] = z;
  });
}

// ---- the WIDE f_phi of the state-description models (config.json *-sd: 512 -> 512 -> 1024 -> 28, 3 MB of fp32 weights).  The
// one-launch kernels split ROWS over workgroups, so every workgroup pulls every weight through one CU: at B = 4 that is ONE
// workgroup and 177 us forward + 58 us backward (a CU takes 20 - 40 GB/s from L2 / HBM).  Here a layer is a launch of its own,
// split over OUTPUT FEATURES (64 per workgroup: 8 - 16 CUs share a layer's weights, 128 - 256 KB each) and rows (FP_RB per
// workgroup, as before).  The thread map differs, the arithmetic does not: the same 16 k-slices, the same fmaf chain inside a
// slice (fp_cols4), the same slice-order sum, the same bias add -- bit for bit what f_phi_fwd_kernel<true> / f_phi_bwd_dz_kernel
// compute (GPU test: the wide shapes through both).
namespace {
constexpr int FPW_COLS = 64;
enum { FPW_RELU = 0, FPW_MASK_RELU = 1, FPW_GATE_MASK = 2, FPW_GATE = 3, FPW_PLAIN = 4 };
// the dispatch: anything wider than the 256-wide image models (whose f_phi is rn_fphi.hip's / the one-launch kernels' business)
int g_fp_wide_mode = 0;             // rn_debug_f_phi_wide (tests): -1 never, 0 by size, 1 always
inline bool fp_is_wide(int G, int F1, int F2) {
  return g_fp_wide_mode ? g_fp_wide_mode > 0 : (long)G * F1 + (long)F1 * F2 > 2L * 256 * 256;
}
}  // namespace
extern "C" int rn_debug_f_phi_wide(int mode) {
  const int was = g_fp_wide_mode;
  g_fp_wide_mode = mode < 0 ? -1 : (mode > 0 ? 1 : 0);
  return was;
}

// out[r][f] = epi(bias[f] + sum_k in[r][k] * WT[k][f]); WT: (K, N) row-major, N % 4 == 0; grid (rows / FP_RB, N / 64)
//   FPW_RELU: relu(z) | FPW_MASK_RELU: relu(z * mask) | FPW_GATE_MASK: act > 0 ? z * mask : 0 | FPW_GATE: act > 0 ? z : 0 | FPW_PLAIN: z
template <int EPI>
__global__ __launch_bounds__(256) void fp_wide_layer_kernel(const float* __restrict__ in, const float* __restrict__ WT,
                                                            const float* __restrict__ bias, const float* __restrict__ act,
                                                            const float* __restrict__ mask, float* __restrict__ out, int B, int K, int N) {
  __shared__ __attribute__((aligned(16))) float in_s[FP_RB * FP_MAXW], red[FP_KQ * FP_RB * FPW_COLS];
  const int t = threadIdx.x, r0 = blockIdx.x * FP_RB, c0 = blockIdx.y * FPW_COLS;
  for (int c = t; c < FP_RB * K; c += 256) {
    const int r = c / K, k = c - r * K;
    in_s[r * FP_MAXW + k] = (r0 + r < B) ? in[(long)(r0 + r) * K + k] : 0.f;
  }
  __syncthreads();
  {
    // thread = (k-slice, four columns): 16 slices x 16 column quads; a wave holds four slices
    const int lane = t & 63, j4 = lane & 15, kq = (t >> 6) * 4 + (lane >> 4), f0 = c0 + 4 * j4;
    const int chunk = ((K + FP_KQ - 1) / FP_KQ + 3) & ~3, i0 = min(K, kq * chunk), i1 = min(K, i0 + chunk);
    f32x4 a4[FP_RB];
#pragma unroll
    for (int r = 0; r < FP_RB; ++r) a4[r] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (f0 < N) fp_cols4(WT, in_s, i0, i1, N, f0, a4);
#pragma unroll
    for (int r = 0; r < FP_RB; ++r) *reinterpret_cast<f32x4*>(red + (kq * FP_RB + r) * FPW_COLS + 4 * j4) = a4[r];
  }
  __syncthreads();
  const int r = t >> 6, f = c0 + (t & 63);
  if (f < N && r0 + r < B) {
    float v = red[r * FPW_COLS + (t & 63)];
#pragma unroll
    for (int q = 1; q < FP_KQ; ++q) v += red[(q * FP_RB + r) * FPW_COLS + (t & 63)];
    const float z = v + (bias ? bias[f] : 0.f);
    const long o = (long)(r0 + r) * N + f;
    float y;
    if constexpr (EPI == FPW_RELU) y = fmaxf(z, 0.f);
    else if constexpr (EPI == FPW_MASK_RELU) y = fmaxf(z * (mask ? mask[o] : 1.f), 0.f);
    else if constexpr (EPI == FPW_GATE_MASK) y = (act[o] > 0.f) ? z * (mask ? mask[o] : 1.f) : 0.f;
    else if constexpr (EPI == FPW_GATE) y = (act[o] > 0.f) ? z : 0.f;
    else y = z;
    out[o] = y;
  }
}

// the head of the wide forward: f2 rows -> LDS, then layer 3 + log_softmax (+ mean NLL) exactly as the one-launch kernel
__global__ __launch_bounds__(FP_KS * 256) void fp_wide_head_kernel(const float* __restrict__ f2, const float* __restrict__ W3T,
                                                                   const float* __restrict__ b3, float* __restrict__ out, int B, int F2,
                                                                   int A, const long long* __restrict__ label, float* __restrict__ loss,
                                                                   float* loss_part, unsigned* done_count) {
  __shared__ __attribute__((aligned(16))) float sa[FP_RB * FP_MAXW], sb[FP_RB * FP_MAXW], red[FP_KQ * FP_RB * 256];
  __shared__ float lrow[FP_RB];
  const int t = threadIdx.x, r0 = blockIdx.x * FP_RB;
  for (int c = t; c < FP_RB * F2; c += blockDim.x) {
    const int r = c / F2, k = c - r * F2;
    sa[r * FP_MAXW + k] = (r0 + r < B) ? f2[(long)(r0 + r) * F2 + k] : 0.f;
  }
  __syncthreads();
  fp_head<true>(W3T, b3, sa, sb, red, lrow, out, B, F2, A, r0, label, loss, loss_part, done_count);
}

// dz3 = d loss / d logits from the stored log-probs (f_phi_bwd_dz_kernel's first step): thread = one row
__global__ __launch_bounds__(64) void fp_wide_dz3_kernel(const float* __restrict__ gout, const float* __restrict__ out,
                                                         float* __restrict__ dz3, int B, int A, const long long* __restrict__ label,
                                                         const float* __restrict__ gloss) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  const float gl = label ? -gloss[0] / (float)B : 0.f;
  const int lb = label ? fp_label(label[b], A) : -1;
  float s = label ? gl : 0.f;
  if (!label)
    for (int a = 0; a < A; ++a) s += gout[(long)b * A + a];
  for (int a = 0; a < A; ++a) {
    const float go = label ? (a == lb ? gl : 0.f) : gout[(long)b * A + a];
    dz3[(long)b * A + a] = go - expf(out[(long)b * A + a]) * s;
  }
}

// forward of the wide path: three launches (W*T transposed weights); label / loss / part / cnt NULL: no loss
static void fp_wide_fwd(const float* xg, const float* W1T, const float* b1, const float* W2T, const float* b2, const float* W3T,
                        const float* b3, const float* mask, float* f1, float* f2, float* out, int B, int G, int F1, int F2, int A,
                        const long long* label, float* loss, float* part, unsigned* cnt, hipStream_t s) {
  const int rb = cdiv(B, FP_RB);
  fp_wide_layer_kernel<FPW_RELU><<<dim3(rb, cdiv(F1, FPW_COLS)), 256, 0, s>>>(xg, W1T, b1, nullptr, nullptr, f1, B, G, F1);
  fp_wide_layer_kernel<FPW_MASK_RELU><<<dim3(rb, cdiv(F2, FPW_COLS)), 256, 0, s>>>(f1, W2T, b2, nullptr, mask, f2, B, F1, F2);
  fp_wide_head_kernel<<<rb, FP_KS * 256, 0, s>>>(f2, W3T, b3, out, B, F2, A, label, loss, part, cnt);
}

// ... and of the backward dz chain (W* natural (out, in) weights = the (in, out) operands of these products): four launches
static void fp_wide_bwd_dz(const float* gout, const float* out, const float* f2, const float* f1, const float* W1, const float* W2,
                           const float* W3, const float* mask, float* dz3, float* dz2, float* dz1, float* dxg, int B, int G, int F1,
                           int F2, int A, const long long* label, const float* gloss, hipStream_t s) {
  const int rb = cdiv(B, FP_RB);
  fp_wide_dz3_kernel<<<cdiv(B, 64), 64, 0, s>>>(gout, out, dz3, B, A, label, gloss);
  fp_wide_layer_kernel<FPW_GATE_MASK><<<dim3(rb, cdiv(F2, FPW_COLS)), 256, 0, s>>>(dz3, W3, nullptr, f2, mask, dz2, B, A, F2);
  fp_wide_layer_kernel<FPW_GATE><<<dim3(rb, cdiv(F1, FPW_COLS)), 256, 0, s>>>(dz2, W2, nullptr, f1, nullptr, dz1, B, F2, F1);
  fp_wide_layer_kernel<FPW_PLAIN><<<dim3(rb, cdiv(G, FPW_COLS)), 256, 0, s>>>(dz1, W1, nullptr, nullptr, nullptr, dxg, B, F1, G);
}

// block = one output row i of dW1 (F1 rows) | dW2 (F2) | dW3 (A): dW[i][j] = sum_b dz[b][i] * act[b][j]; db[i] = sum_b dz[b][i]
__global__ __launch_bounds__(256) void f_phi_bwd_grads_kernel(const float* __restrict__ dz1, const float* __restrict__ dz2,
                                                              const float* __restrict__ dz3, const float* __restrict__ xg,
                                                              const float* __restrict__ f1, const float* __restrict__ f2,
                                                              float* __restrict__ dW1, float* __restrict__ db1, float* __restrict__ dW2,
                                                              float* __restrict__ db2, float* __restrict__ dW3, float* __restrict__ db3,
                                                              int B, int G, int F1, int F2, int A) {
  constexpr int BT = 1024;                 // batch rows staged per pass (any B: the passes accumulate in registers)
  __shared__ float dzs[BT];
  int i = blockIdx.x;
  const float *dz, *act;
  float *dW, *db;
  int I, J;
  if (i < F1) { dz = dz1; act = xg; dW = dW1; db = db1; I = F1; J = G; }
  else if (i < F1 + F2) { i -= F1; dz = dz2; act = f1; dW = dW2; db = db2; I = F2; J = F1; }
  else { i -= F1 + F2; dz = dz3; act = f2; dW = dW3; db = db3; I = A; J = F2; }
  constexpr int JT = FP_MAXW / 256;        // columns per thread (J <= FP_MAXW)
  float acc[JT], sb = 0.f;
#pragma unroll
  for (int u = 0; u < JT; ++u) acc[u] = 0.f;
  for (int b0 = 0; b0 < B; b0 += BT) {
    const int nb = B - b0 < BT ? B - b0 : BT;
    __syncthreads();
    for (int b = threadIdx.x; b < nb; b += 256) dzs[b] = dz[(long)(b0 + b) * I + i];
    __syncthreads();
#pragma unroll
    for (int u = 0; u < JT; ++u) {
      const int j = threadIdx.x + 256 * u;
      if (j < J) {
        float s = acc[u];
#pragma unroll 16
        for (int b = 0; b < nb; ++b) s = fmaf(dzs[b], act[(long)(b0 + b) * J + j], s);
        acc[u] = s;
      }
    }
    if (threadIdx.x == 0)
      for (int b = 0; b < nb; ++b) sb += dzs[b];
  }
#pragma unroll
  for (int u = 0; u < JT; ++u) {
    const int j = threadIdx.x + 256 * u;
    if (j < J) dW[(long)i * J + j] = acc[u];
  }
  if (threadIdx.x == 0) db[i] = sb;
}

static int fp_check(const char* who, int B, int G, int F1, int F2, int A) {
  RN_CHECK_ARG(B > 0 && G > 0 && F1 > 0 && F2 > 0 && A > 0, "%s: bad sizes", who);
  RN_CHECK_ARG(G <= FP_MAXW && F1 <= FP_MAXW && F2 <= FP_MAXW && A <= FP_MAXW && G % 4 == 0 && F1 % 4 == 0 && F2 % 4 == 0,
               "%s: widths must be multiples of 4 and <= %d (G=%d F1=%d F2=%d A=%d)", who, FP_MAXW, G, F1, F2, A);
  return 0;
}

extern "C" int rn_f_phi_fwd(const float* xg, const float* W1, const float* b1, const float* W2, const float* b2, const float* W3,
                            const float* b3, const float* mask, float* f1, float* f2, float* out, int transposed, int B, int G, int F1,
                            int F2, int A, void* stream) {
  RN_CHECK_ARG(xg && W1 && b1 && W2 && b2 && W3 && b3 && f1 && f2 && out, "rn_f_phi_fwd: NULL pointer");
  if (int rc = fp_check("rn_f_phi_fwd", B, G, F1, F2, A)) return rc;
  RN_CHECK_ARG(((uintptr_t)W1 | (uintptr_t)W2 | (uintptr_t)W3) % 16 == 0, "rn_f_phi_fwd: weights must be 16-byte aligned");
  if (transposed && fp_is_wide(G, F1, F2)) fp_wide_fwd(xg, W1, b1, W2, b2, W3, b3, mask, f1, f2, out, B, G, F1, F2, A, nullptr, nullptr, nullptr, nullptr, (hipStream_t)stream);
  else if (transposed) f_phi_fwd_kernel<true><<<cdiv(B, FP_RB), FP_KS * 256, 0, (hipStream_t)stream>>>(xg, W1, b1, W2, b2, W3, b3, mask, f1, f2, out, B, G, F1, F2, A);
  else f_phi_fwd_kernel<false><<<cdiv(B, FP_RB), 256, 0, (hipStream_t)stream>>>(xg, W1, b1, W2, b2, W3, b3, mask, f1, f2, out, B, G, F1, F2, A);
  RN_LAUNCH_CHECK("rn_f_phi_fwd");
  return 0;
}

// f_phi + log_softmax + mean NLL (train.py:40-41) in one launch.  sync_ws: rn_f_phi_nll_ws_bytes(B) bytes, ZEROED ONCE by the
// caller and then owned by these calls (block partials + a completion counter that re-arms itself).
size_t rnws_f_phi_nll(int B) { return B > 0 ? ((size_t)cdiv(B, FP_RB) + 4) * sizeof(float) : 0; }

extern "C" int rn_f_phi_fwd_nll(const float* xg, const float* W1, const float* b1, const float* W2, const float* b2, const float* W3,
                                const float* b3, const float* mask, const long long* label, float* f1, float* f2, float* out,
                                float* loss, void* sync_ws, int transposed, int B, int G, int F1, int F2, int A, void* stream) {
  RN_CHECK_ARG(xg && W1 && b1 && W2 && b2 && W3 && b3 && f1 && f2 && out && label && loss && sync_ws, "rn_f_phi_fwd_nll: NULL pointer");
  if (int rc = fp_check("rn_f_phi_fwd_nll", B, G, F1, F2, A)) return rc;
  RN_CHECK_ARG(((uintptr_t)W1 | (uintptr_t)W2 | (uintptr_t)W3) % 16 == 0, "rn_f_phi_fwd_nll: weights must be 16-byte aligned");
  unsigned* cnt = (unsigned*)sync_ws;
  float* part = (float*)sync_ws + 4;
  if (transposed && fp_is_wide(G, F1, F2)) fp_wide_fwd(xg, W1, b1, W2, b2, W3, b3, mask, f1, f2, out, B, G, F1, F2, A, label, loss, part, cnt, (hipStream_t)stream);
  else if (transposed) f_phi_fwd_kernel<true><<<cdiv(B, FP_RB), FP_KS * 256, 0, (hipStream_t)stream>>>(xg, W1, b1, W2, b2, W3, b3, mask, f1, f2, out, B, G, F1, F2, A, label, loss, part, cnt);
  else f_phi_fwd_kernel<false><<<cdiv(B, FP_RB), 256, 0, (hipStream_t)stream>>>(xg, W1, b1, W2, b2, W3, b3, mask, f1, f2, out, B, G, F1, F2, A, label, loss, part, cnt);
  RN_LAUNCH_CHECK("rn_f_phi_fwd_nll");
  return 0;
}

// ... and with the pair sum in front (rn_pair_sum_fwd on the forward chains' partial rows folded into this launch): xg (B, G) is an
// OUTPUT here, xg_part (B * parts_per_row, G) fp32 the chain's partials.  label / loss / sync_ws NULL: no loss.
extern "C" int rn_f_phi_fwd_from_partials(const float* xg_part, int parts_per_row, float* xg, const float* W1, const float* b1,
                                          const float* W2, const float* b2, const float* W3, const float* b3, const float* mask,
                                          const long long* label, float* f1, float* f2, float* out, float* loss, void* sync_ws,
                                          int transposed, int B, int G, int F1, int F2, int A, void* stream) {
  RN_CHECK_ARG(xg_part && parts_per_row > 0 && xg && W1 && b1 && W2 && b2 && W3 && b3 && f1 && f2 && out, "rn_f_phi_fwd_from_partials: NULL pointer / bad count");
  RN_CHECK_ARG((label != nullptr) == (loss != nullptr) && (label != nullptr) == (sync_ws != nullptr), "rn_f_phi_fwd_from_partials: label, loss and sync_ws go together");
  if (int rc = fp_check("rn_f_phi_fwd_from_partials", B, G, F1, F2, A)) return rc;
  RN_CHECK_ARG(((uintptr_t)W1 | (uintptr_t)W2 | (uintptr_t)W3) % 16 == 0, "rn_f_phi_fwd_from_partials: weights must be 16-byte aligned");
  unsigned* cnt = label ? (unsigned*)sync_ws : nullptr;
  float* part = label ? (float*)sync_ws + 4 : nullptr;
  if (transposed) f_phi_fwd_kernel<true><<<cdiv(B, FP_RB), FP_KS * 256, 0, (hipStream_t)stream>>>(nullptr, W1, b1, W2, b2, W3, b3, mask, f1, f2, out, B, G, F1, F2, A, label, loss, part, cnt, xg_part, parts_per_row, xg);
  else f_phi_fwd_kernel<false><<<cdiv(B, FP_RB), 256, 0, (hipStream_t)stream>>>(nullptr, W1, b1, W2, b2, W3, b3, mask, f1, f2, out, B, G, F1, F2, A, label, loss, part, cnt, xg_part, parts_per_row, xg);
  RN_LAUNCH_CHECK("rn_f_phi_fwd_from_partials");
  return 0;
}

// rn_f_phi_fwd_from_partials (transposed weights, loss folded in) + the backward dz chain for d loss = 1 in the SAME launch:
// W1..3 = the natural (out, in) weights, bwd_ws = rn_f_phi_bwd_ws_bytes(...) bytes (receives the dz rows), dxg (B, G) out.
// rn_f_phi_bwd_grads then produces the six parameter gradients from bwd_ws; a loss gradient other than 1 takes rn_f_phi_bwd_nll.
extern "C" int rn_f_phi_fwd_bwd_from_partials(const float* xg_part, int parts_per_row, float* xg, const float* W1T, const float* b1,
                                              const float* W2T, const float* b2, const float* W3T, const float* b3, const float* W1,
                                              const float* W2, const float* W3, const float* mask, const long long* label, float* f1,
                                              float* f2, float* out, float* loss, void* sync_ws, void* bwd_ws, float* dxg, int B, int G,
                                              int F1, int F2, int A, void* stream) {
  RN_CHECK_ARG(xg_part && parts_per_row > 0 && xg && W1T && b1 && W2T && b2 && W3T && b3 && W1 && W2 && W3 && label && f1 && f2 && out && loss &&
                   sync_ws && bwd_ws && dxg,
               "rn_f_phi_fwd_bwd_from_partials: NULL pointer / bad count");
  if (int rc = fp_check("rn_f_phi_fwd_bwd_from_partials", B, G, F1, F2, A)) return rc;
  RN_CHECK_ARG(((uintptr_t)W1T | (uintptr_t)W2T | (uintptr_t)W3T | (uintptr_t)W1 | (uintptr_t)W2 | (uintptr_t)W3) % 16 == 0,
               "rn_f_phi_fwd_bwd_from_partials: weights must be 16-byte aligned");
  unsigned* cnt = (unsigned*)sync_ws;
  float* part = (float*)sync_ws + 4;
  FpBwdTail bt;
  bt.W1 = W1; bt.W2 = W2; bt.W3 = W3;
  bt.dz1 = (float*)bwd_ws;
  bt.dz2 = bt.dz1 + (size_t)B * F1;
  bt.dz3 = bt.dz2 + (size_t)B * F2;
  bt.dxg = dxg;
  f_phi_fwd_kernel<true><<<cdiv(B, FP_RB), FP_KS * 256, 0, (hipStream_t)stream>>>(nullptr, W1T, b1, W2T, b2, W3T, b3, mask, f1, f2, out, B, G, F1, F2, A,
                                                                                  label, loss, part, cnt, xg_part, parts_per_row, xg, bt);
  RN_LAUNCH_CHECK("rn_f_phi_fwd_bwd_from_partials");
  return 0;
}

extern "C" int rn_f_phi_bwd_grads(const void* bwd_ws, const float* xg, const float* f1, const float* f2, float* dW1, float* db1, float* dW2,
                                  float* db2, float* dW3, float* db3, int B, int G, int F1, int F2, int A, void* stream) {
  RN_CHECK_ARG(bwd_ws && xg && f1 && f2 && dW1 && db1 && dW2 && db2 && dW3 && db3, "rn_f_phi_bwd_grads: NULL pointer");
  if (int rc = fp_check("rn_f_phi_bwd_grads", B, G, F1, F2, A)) return rc;
  const float* dz1 = (const float*)bwd_ws;
  const float* dz2 = dz1 + (size_t)B * F1;
  const float* dz3 = dz2 + (size_t)B * F2;
  f_phi_bwd_grads_kernel<<<F1 + F2 + A, 256, 0, (hipStream_t)stream>>>(dz1, dz2, dz3, xg, f1, f2, dW1, db1, dW2, db2, dW3, db3, B, G, F1, F2, A);
  RN_LAUNCH_CHECK("rn_f_phi_bwd_grads");
  return 0;
}

size_t rnws_f_phi_bwd(int B, int F1, int F2, int A) { return (size_t)B * (F1 + F2 + A) * sizeof(float); }

extern "C" int rn_f_phi_bwd(const float* gout, const float* out, const float* f2, const float* f1, const float* xg, const float* W1,
                            const float* W2, const float* W3, const float* mask, float* dW1, float* db1, float* dW2, float* db2,
                            float* dW3, float* db3, float* dxg, void* ws, int B, int G, int F1, int F2, int A, void* stream) {
  RN_CHECK_ARG(gout && out && f2 && f1 && xg && W1 && W2 && W3 && dW1 && db1 && dW2 && db2 && dW3 && db3 && dxg && ws, "rn_f_phi_bwd: NULL pointer");
  if (int rc = fp_check("rn_f_phi_bwd", B, G, F1, F2, A)) return rc;
  float* dz1 = (float*)ws;
  float* dz2 = dz1 + (size_t)B * F1;
  float* dz3 = dz2 + (size_t)B * F2;
  hipStream_t s = (hipStream_t)stream;
  if (fp_is_wide(G, F1, F2)) fp_wide_bwd_dz(gout, out, f2, f1, W1, W2, W3, mask, dz3, dz2, dz1, dxg, B, G, F1, F2, A, nullptr, nullptr, s);
  else f_phi_bwd_dz_kernel<<<cdiv(B, FP_RB), FP_KS * 256, 0, s>>>(gout, out, f2, f1, W1, W2, W3, mask, dz3, dz2, dz1, dxg, B, G, F1, F2, A);
  f_phi_bwd_grads_kernel<<<F1 + F2 + A, 256, 0, s>>>(dz1, dz2, dz3, xg, f1, f2, dW1, db1, dW2, db2, dW3, db3, B, G, F1, F2, A);
  RN_LAUNCH_CHECK("rn_f_phi_bwd");
  return 0;
}

// Backward of rn_f_phi_fwd_nll: gloss = d L / d loss (one device float); the log-prob gradient -gloss/B at the labels is
// formed inside the first kernel.
extern "C" int rn_f_phi_bwd_nll(const float* gloss, const long long* label, const float* out, const float* f2, const float* f1,
                                const float* xg, const float* W1, const float* W2, const float* W3, const float* mask, float* dW1,
                                float* db1, float* dW2, float* db2, float* dW3, float* db3, float* dxg, void* ws, int B, int G, int F1,
                                int F2, int A, void* stream) {
  RN_CHECK_ARG(gloss && label && out && f2 && f1 && xg && W1 && W2 && W3 && dW1 && db1 && dW2 && db2 && dW3 && db3 && dxg && ws, "rn_f_phi_bwd_nll: NULL pointer");
  if (int rc = fp_check("rn_f_phi_bwd_nll", B, G, F1, F2, A)) return rc;
  float* dz1 = (float*)ws;
  float* dz2 = dz1 + (size_t)B * F1;
  float* dz3 = dz2 + (size_t)B * F2;
  hipStream_t s = (hipStream_t)stream;
  if (fp_is_wide(G, F1, F2)) fp_wide_bwd_dz(nullptr, out, f2, f1, W1, W2, W3, mask, dz3, dz2, dz1, dxg, B, G, F1, F2, A, label, gloss, s);
  else f_phi_bwd_dz_kernel<<<cdiv(B, FP_RB), FP_KS * 256, 0, s>>>(nullptr, out, f2, f1, W1, W2, W3, mask, dz3, dz2, dz1, dxg, B, G, F1, F2, A, label, gloss);
  f_phi_bwd_grads_kernel<<<F1 + F2 + A, 256, 0, s>>>(dz1, dz2, dz3, xg, f1, f2, dW1, db1, dW2, db2, dW3, db3, B, G, F1, F2, A);
  RN_LAUNCH_CHECK("rn_f_phi_bwd_nll");
  return 0;
}

// ------------------------------------------------------------------------------- clip + Adam on the flat gradient
// The tail of a training step (train.py:45-48: clip_grad_norm, Adam with coupled weight decay) as two launches over
// the flat gradient buffer of the data-parallel bucket instead of ~25 small ones:
//   1. per-block sums of squares of the (already all-reduced) gradient;
//   2. every block adds the partials up in the same fixed order (-> the same norm everywhere), forms the
//      torch.nn.utils.clip_grad_norm_ coefficient min(1, max_norm / (norm + 1e-6)) and applies torch.optim.Adam
//      (amsgrad=False) to its chunk.  Parameters stay separate tensors: a chunk table maps flat ranges to them.
namespace {
constexpr int OPT_NB = 256;        // blocks of the norm pass
constexpr int OPT_CHUNK = 1024;    // elements per chunk of the update pass (a chunk never crosses a parameter)
}  // namespace
struct RnAdamChunk { float* param; long flat_off; int count; int pad; };

__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ g, long n, double* __restrict__ partial,
                                                            int* __restrict__ step_dev = nullptr) {
  __shared__ double red[4];
  if (step_dev && blockIdx.x == 0 && threadIdx.x == 0) step_dev[0] += 1;   // device-side update count (rn_clip_adam_step_dev): the next launch reads it
  float a = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) a = fmaf(g[i], g[i], a);
  double x = (double)a;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) x += __shfl_xor(x, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = x;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void clip_adam_kernel(const RnAdamChunk* __restrict__ chunks, float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v,
                                                        const double* __restrict__ partial, int npartial, float gscale,
                                                        float max_norm, float lr, float beta1, float beta2, float eps, float wd, float bc1,
                                                        float bc2_sqrt, float* __restrict__ norm_out, const float* __restrict__ hyper = nullptr,
                                                        const int* __restrict__ step_dev = nullptr) {
  __shared__ float coef_s, hs[8];
  __shared__ double nred[4];
  // the norm: every block adds the npartial (<= 256) partials in the same fixed tree (-> the same value everywhere) -- one
  // load per thread and a shuffle tree instead of one thread walking the list (which was three quarters of this kernel)
  {
    double x = (int)threadIdx.x < npartial ? partial[threadIdx.x] : 0.0;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) x += __shfl_xor(x, o);
    if ((threadIdx.x & 63) == 0) nred[threadIdx.x >> 6] = x;
  }
  if (hyper) {
    // hyper-parameters and the update count from device memory: the launch can sit in a captured graph and still follow an LR
    // schedule.  hyper = {grad_scale, max_norm, lr, beta1, beta2, eps, weight_decay}; bias corrections as on the host (double)
    if (threadIdx.x == 0) {
      const double t = (double)step_dev[0];
      hs[0] = hyper[0]; hs[1] = hyper[1]; hs[2] = hyper[2]; hs[3] = hyper[3]; hs[4] = hyper[4]; hs[5] = hyper[5]; hs[6] = hyper[6];
      hs[7] = (float)(1.0 - pow((double)hyper[3], t));
      coef_s = (float)sqrt(1.0 - pow((double)hyper[4], t));
    }
    __syncthreads();
    gscale = hs[0]; max_norm = hs[1]; lr = hs[2]; beta1 = hs[3]; beta2 = hs[4]; eps = hs[5]; wd = hs[6]; bc1 = hs[7]; bc2_sqrt = coef_s;
    __syncthreads();
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double s = (nred[0] + nred[1]) + (nred[2] + nred[3]);
    const float total = gscale * (float)sqrt(s);           // norm of the SCALED gradient (gscale = 1 / world after a sum all-reduce)
    float c = 1.f;
    if (max_norm > 0.f) {
      c = max_norm / (total + 1e-6f);
      c = c < 1.f ? c : 1.f;
    }
    coef_s = c;
    if (blockIdx.x == 0 && norm_out) norm_out[0] = total;
  }
  __syncthreads();
  const float coef = coef_s * gscale;
  const RnAdamChunk c = chunks[blockIdx.x];
  const float step = lr / bc1;
  for (int i = threadIdx.x; i < c.count; i += 256) {
    const long f = c.flat_off + i;
    const float gc = g[f] * coef;                          // clipped gradient (left in the buffer, like clip_grad_norm_)
    g[f] = gc;
    const float p = c.param[i];
    const float gd = fmaf(wd, p, gc);                      // coupled L2 weight decay
    const float mn = fmaf(beta1, m[f], (1.f - beta1) * gd);
    const float vn = fmaf(beta2, v[f], (1.f - beta2) * gd * gd);
    m[f] = mn;
    v[f] = vn;
    c.param[i] = p - step * mn / (sqrtf(vn) / bc2_sqrt + eps);
  }
}

extern "C" int rn_clip_adam_chunk(void) { return OPT_CHUNK; }
size_t rnws_clip_adam(void) { return OPT_NB * sizeof(double); }

extern "C" int rn_clip_adam_step(const void* chunks, int nchunks, float* g, float* m, float* v, long n, void* ws, float grad_scale,
                                 float max_norm, float lr, float beta1, float beta2, float eps, float weight_decay, int step, float* norm_out,
                                 void* stream) {
  RN_CHECK_ARG(chunks && nchunks > 0 && g && m && v && n > 0 && ws && step >= 1, "rn_clip_adam_step: bad argument");
  hipStream_t s = (hipStream_t)stream;
  sumsq_partial_kernel<<<OPT_NB, 256, 0, s>>>(g, n, (double*)ws);
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  clip_adam_kernel<<<nchunks, 256, 0, s>>>((const RnAdamChunk*)chunks, g, m, v, (const double*)ws, OPT_NB, grad_scale, max_norm, lr, beta1, beta2,
                                           eps, weight_decay, (float)bc1, (float)sqrt(bc2), norm_out);
  RN_LAUNCH_CHECK("rn_clip_adam_step");
  return 0;
}

// The same two launches with every per-step scalar in device memory: hyper (7 floats: grad_scale, max_norm, lr, beta1, beta2, eps,
// weight_decay) and the 1-based update count step_dev[0], which the first launch increments -- capturable in a hipGraph (the
// host rewrites `hyper` only when a scheduler changes it).
extern "C" int rn_clip_adam_step_dev(const void* chunks, int nchunks, float* g, float* m, float* v, long n, void* ws, const float* hyper,
                                     int* step_dev, float* norm_out, void* stream) {
  RN_CHECK_ARG(chunks && nchunks > 0 && g && m && v && n > 0 && ws && hyper && step_dev, "rn_clip_adam_step_dev: bad argument");
  hipStream_t s = (hipStream_t)stream;
  sumsq_partial_kernel<<<OPT_NB, 256, 0, s>>>(g, n, (double*)ws, step_dev);
  clip_adam_kernel<<<nchunks, 256, 0, s>>>((const RnAdamChunk*)chunks, g, m, v, (const double*)ws, OPT_NB, 1.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 1.f, 1.f,
                                           norm_out, hyper, step_dev);
  RN_LAUNCH_CHECK("rn_clip_adam_step_dev");
  return 0;
}

// ------------------------------------------------------------------------------------------------ mean NLL loss
// F.nll_loss(log_probs, label) of the training loop (train.py:41), mean reduction: forward one launch, backward one
// launch that writes the WHOLE (B, A) gradient (-g / B at the label, 0 elsewhere) -- the stock path spends four launches
// (gather-reduce, two fills, scatter) on the critical path between the forward and the backward chain.
__global__ __launch_bounds__(256) void nll_mean_fwd_kernel(const float* __restrict__ logp, const long long* __restrict__ label,
                                                           float* __restrict__ loss, int B, int A) {
  __shared__ double red[4];
  double a = 0.0;
  for (int b = threadIdx.x; b < B; b += 256) {
    long long y = label[b];
    y = y < 0 ? 0 : (y >= A ? A - 1 : y);
    a -= (double)logp[(long)b * A + y];
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) a += __shfl_xor(a, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) loss[0] = (float)(((red[0] + red[1]) + (red[2] + red[3])) / B);
}
__global__ __launch_bounds__(256) void nll_mean_bwd_kernel(const long long* __restrict__ label, const float* __restrict__ gloss,
                                                           float* __restrict__ gout, int B, int A) {
  const float g = -gloss[0] / (float)B;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < (long)B * A; i += (long)gridDim.x * 256) {
    const int b = (int)(i / A), c = (int)(i - (long)b * A);
    long long y = label[b];
    y = y < 0 ? 0 : (y >= A ? A - 1 : y);
    gout[i] = (c == y) ? g : 0.f;
  }
}
extern "C" int rn_nll_mean_fwd(const float* logp, const long long* label, float* loss, int B, int A, void* stream) {
  RN_CHECK_ARG(logp && label && loss && B > 0 && A > 0, "rn_nll_mean_fwd: bad pointer/size");
  nll_mean_fwd_kernel<<<1, 256, 0, (hipStream_t)stream>>>(logp, label, loss, B, A);
  RN_LAUNCH_CHECK("rn_nll_mean_fwd");
  return 0;
}
extern "C" int rn_nll_mean_bwd(const long long* label, const float* gloss, float* gout, int B, int A, void* stream) {
  RN_CHECK_ARG(label && gloss && gout && B > 0 && A > 0, "rn_nll_mean_bwd: bad pointer/size");
  int blocks = cdiv((long)B * A, 256);
  if (blocks > 256) blocks = 256;
  nll_mean_bwd_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(label, gloss, gout, B, A);
  RN_LAUNCH_CHECK("rn_nll_mean_bwd");
  return 0;
}

// ------------------------------------------------------------------------------------------------ diagnostics
// One-thread kernel that writes the constant-rate wall clock into *slot: captured into the step's hipGraph between
// the real kernels it gives a concurrent multi-stream timeline (tools/step_timeline.py) -- rocprofv3's kernel trace
// serialises the queues and cannot.
__global__ void debug_stamp_kernel(unsigned long long* slot) { *slot = wall_clock64(); }

extern "C" int rn_debug_stamp(unsigned long long* slot, void* stream) {
  RN_CHECK_ARG(slot, "rn_debug_stamp: NULL slot");
  debug_stamp_kernel<<<1, 1, 0, (hipStream_t)stream>>>(slot);
  RN_LAUNCH_CHECK("rn_debug_stamp");
  return 0;
}

// ---- the matrix pipe's SUSTAINED rate on the device at hand: every wave issues `iters` x 16 v_mfma_f32_32x32x16_{f16|bf16} on two
// alternating accumulators from constant registers -- nothing else, one workgroup per CU, 1 or 2 waves per SIMD.  The chip clocks
// to its power budget: this bare stream is what "MFMA-bound" can mean on a given box at a given moment (measured on the pool:
// 19.5-20 ns per 32-cycle MFMA slot = 1.7 PFLOP/s where the nominal figure is 2.5), and bench.py quotes the chains against both.
template <bool BF>
__global__ __launch_bounds__(512) void mfma_stream_kernel(float* out, int iters, int zero_operands) {
  typedef __attribute__((ext_vector_type(8))) _Float16 h8;
  typedef __attribute__((ext_vector_type(8))) __bf16 b8;
  typedef __attribute__((ext_vector_type(16))) float f16v;
  const int lane = threadIdx.x & 63;
  f16v acc[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[0][i] = acc[1][i] = 0.f;
  h8 ah, bh;
  b8 ab, bb;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    // zero_operands: the same instruction stream on all-zero inputs -- the data pattern most "peak" figures are measured with; the
    // chip then draws less power per MFMA and clocks higher (MI355X_MICROARCH.md "DVFS give-back": +19 % TF/s on zero-filled inputs)
    const float x = zero_operands ? 0.f : 0.37f * (float)((lane * 7 + i * 3) % 17 - 8), y = zero_operands ? 0.f : 0.21f * (float)((lane * 5 + i * 11) % 13 - 6);
    ah[i] = (_Float16)x; bh[i] = (_Float16)y; ab[i] = (__bf16)x; bb[i] = (__bf16)y;
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      if constexpr (BF) acc[c & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc[c & 1], 0, 0, 0);
      else acc[c & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[c & 1], 0, 0, 0);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[0][i] + acc[1][i];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

extern "C" int rn_probe_mfma_stream_ops(float* out, int workgroups, int waves_per_simd, int iters, int dtype, int zero_operands, void* stream) {
  RN_CHECK_ARG(out && workgroups > 0 && (waves_per_simd == 1 || waves_per_simd == 2) && iters > 0 && (dtype == RN_BF16 || dtype == RN_F16),
               "rn_probe_mfma_stream: out (workgroups * 256 * waves_per_simd floats), waves_per_simd 1 | 2, dtype RN_F16 | RN_BF16");
  if (dtype == RN_BF16) mfma_stream_kernel<true><<<workgroups, 256 * waves_per_simd, 0, (hipStream_t)stream>>>(out, iters, zero_operands);
  else mfma_stream_kernel<false><<<workgroups, 256 * waves_per_simd, 0, (hipStream_t)stream>>>(out, iters, zero_operands);
  RN_LAUNCH_CHECK("rn_probe_mfma_stream");
  return 0;
}
extern "C" int rn_probe_mfma_stream(float* out, int workgroups, int waves_per_simd, int iters, int dtype, void* stream) {
  return rn_probe_mfma_stream_ops(out, workgroups, waves_per_simd, iters, dtype, 0, stream);
}

// ------------------------------------------------------------------------------------------------ batch hand-off
namespace {
struct CopyMany {
  unsigned char* dst[4];
  const unsigned char* src[4];
  size_t bytes[4];
  int blk0[5];                                            // first workgroup of segment i (blk0[n] = grid)
  int n;
};
constexpr int CM_PER_BLOCK = 256 * 4 * 16;               // bytes per workgroup: 4 x 16 bytes per thread, all loads before the stores
}
__global__ __launch_bounds__(256) void copy_many_kernel(CopyMany c) {
  int i = 0;
  while (i + 1 < c.n && (int)blockIdx.x >= c.blk0[i + 1]) ++i;
  const size_t off = (size_t)((int)blockIdx.x - c.blk0[i]) * CM_PER_BLOCK, n = c.bytes[i], n16 = n & ~(size_t)15;
  const unsigned char* s = c.src[i];
  unsigned char* d = c.dst[i];
  u32x4 v[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const size_t a = off + ((size_t)q * 256 + threadIdx.x) * 16;
    if (a < n16) v[q] = *reinterpret_cast<const u32x4*>(s + a);
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const size_t a = off + ((size_t)q * 256 + threadIdx.x) * 16;
    if (a < n16) *reinterpret_cast<u32x4*>(d + a) = v[q];
  }
  if (off + CM_PER_BLOCK >= n && n16 + threadIdx.x < n && off <= n16) d[n16 + threadIdx.x] = s[n16 + threadIdx.x];   // (< 16 bytes)
}

// ------------------------------------------------------------------------------------------------ dropout mask
// The f_phi dropout mask (model.py:158: F.dropout on the (B, f_fc2) activations) from a counter-based generator of the library's
// own: mask[i] = hash(seed, draw, i) >= p ? 1 / (1 - p) : 0 with `draw` a DEVICE counter that the launch itself advances (the last
// block to finish: every block has read it by then).  Why not torch's generator: a captured graph that draws from it makes every
// replay launch two fill kernels for the Philox seed / offset IN FRONT of the graph -- 10 us of the 16 between two steps
// (tools/dbg/graph_gaps.py).  state[0] = draws so far, state[1] = completion count (zero between launches).
__device__ __forceinline__ unsigned long long dm_mix(unsigned long long z) {      // splitmix64's finaliser
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__global__ __launch_bounds__(256) void dropout_mask_kernel(float* __restrict__ mask, long n, float p, float keep_scale,
                                                           unsigned long long seed, unsigned long long* __restrict__ state) {
  const unsigned long long draw = state[0];
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    const unsigned long long z = dm_mix(dm_mix(seed + 0x9E3779B97F4A7C15ull * (draw + 1)) + 0x9E3779B97F4A7C15ull * (unsigned long long)(i + 1));
    const float u = (float)(z >> 40) * (1.0f / 16777216.0f);                      // 24 bits -> [0, 1)
    mask[i] = u >= p ? keep_scale : 0.f;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    if (atomicAdd(&state[1], 1ull) == (unsigned long long)gridDim.x - 1) {
      state[0] = draw + 1;
      state[1] = 0ull;
    }
  }
}

extern "C" int rn_dropout_mask(float* mask, long n, float p, unsigned long long seed, unsigned long long* state, void* stream) {
  RN_CHECK_ARG(mask && state && n > 0 && p >= 0.f && p < 1.f, "rn_dropout_mask: bad argument (n=%ld, p=%g: 0 <= p < 1)", n, (double)p);
  RN_CHECK_ARG((uintptr_t)state % 8 == 0, "rn_dropout_mask: state must be two aligned 64-bit words");
  dropout_mask_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(mask, n, p, 1.0f / (1.0f - p), seed, state);
  RN_LAUNCH_CHECK("rn_dropout_mask");
  return 0;
}

extern "C" int rn_copy_many(void* const* dst, const void* const* src, const size_t* bytes, int n, void* stream) {
  RN_CHECK_ARG(dst && src && bytes && n > 0 && n <= 4, "rn_copy_many: bad argument (n=%d, 1..4)", n);
  CopyMany c;
  memset(&c, 0, sizeof(c));
  c.n = n;
  int grid = 0;
  for (int i = 0; i < n; ++i) {
    RN_CHECK_ARG(dst[i] && src[i] && ((uintptr_t)dst[i] | (uintptr_t)src[i]) % 16 == 0 && bytes[i] > 0, "rn_copy_many: segment %d: NULL / misaligned / empty", i);
    c.dst[i] = (unsigned char*)dst[i];
    c.src[i] = (const unsigned char*)src[i];
    c.bytes[i] = bytes[i];
    c.blk0[i] = grid;
    grid += (int)((bytes[i] + CM_PER_BLOCK - 1) / CM_PER_BLOCK);
  }
  c.blk0[n] = grid;
  copy_many_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(c);
  RN_LAUNCH_CHECK("rn_copy_many");
  return 0;
}
