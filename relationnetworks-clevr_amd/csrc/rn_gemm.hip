// K2 -- g_theta layer GEMM on the gfx950 matrix cores (model.py:141-145) and its dgrad.
//
//   out[m, n] = epi( sum_k A[m, k] * W[n, k] )        A: (M, K) row-major, W: (N, K) row-major
//
// Workgroup = 256 threads (4 waves, one per SIMD; two workgroups per CU), tile 128(M) x 256(N),
// K streamed in 128-byte slabs (64 bf16 / 32 fp32 per row) through one padded LDS buffer with
// the next slab prefetched into registers while the current one feeds the MFMAs.
//
// MFMA operand assignment is "swapped": the weight fragment is the A-operand (rows = output
// feature n) and the activation fragment is the B-operand (cols = pair row m), so every lane
// ends up holding 4 CONSECUTIVE output features of one pair row per accumulator group ->
// row-contiguous 8-byte (bf16) / 16-byte (fp32) epilogue stores instead of 2-byte scatters.
//
//   bf16: v_mfma_f32_32x32x16_bf16, lane l supplies row (l&31), k = 8*(l>>5)..+8 of both operands
//   fp32: v_mfma_f32_32x32x2_f32 (exact fp32 fmaf chain), 4 MFMAs per 16-byte fragment; the
//         k order inside a slab is permuted identically for both operands (sum order only).
//   D layout (both): col j = l&31 (pair row), row i = (reg&3) + 8*(reg>>2) + 4*(l>>5) (feature).
#include "rn_common.h"
#include <type_traits>

enum { EPI_BIAS_RELU = 0, EPI_GATE = 1 };

template <typename T> struct Mma;
template <> struct Mma<bf16> {
  typedef bf16x8 Frag;
  static __device__ __forceinline__ void mma(const Frag& a, const Frag& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct Mma<float> {
  typedef f32x4 Frag;
  static __device__ __forceinline__ void mma(const Frag& a, const Frag& b, f32x16& c) {
#pragma unroll
    for (int s = 0; s < 4; ++s) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], c, 0, 0, 0);
  }
};

// Two tile shapes (workgroup tile = 2 x 2 waves of MT x NT 32 x 32 MFMA tiles): 128 x 256 for the pair matrices of the image
// models (thousands of workgroups), 64 x 64 for the state-description models (BASELINE configs[0]: M = B * 144 = 576 pair rows at
// B = 4, 512 features -- 10 of the big tiles, i.e. 10 of 256 CUs at 55 us a layer; 72 small ones).  The k order of every output
// element is the same in both: results are bitwise identical whichever tile computes them.
// RN_F32X3 -- fp32 storage, products on the bf16 matrix pipe: every operand value x is split as it is staged into hi = bf16(x),
// lo = bf16(x - hi) (16 mantissa bits together) and a product is hi*hi + hi*lo + lo*hi in fp32 accumulators -- three
// v_mfma_f32_32x32x16_bf16 (96 cycles per 16 k) where the exact path spends eight v_mfma_f32_32x32x2_f32 (512 cycles); what is
// dropped (lo*lo and the rounding of lo) is 2^-16 of a product.  The 16-bit arithmetic of the 512-wide state-description models,
// whose layers do not fit the register-resident chains (VERDICT r4 "missing" 5); log-probs within 1e-5 of the fp32 reference.
struct F32x3 {};
template <> struct Mma<F32x3> {
  typedef bf16x8 Frag;
};
constexpr int TN_BIG = 256;            // the entry points accept output widths that are multiples of the small tile's 64
// big tiles on fewer than this many workgroups -> small tiles.  Measured (tools/dbg/time_gemm_tiles.py, profiles/r05_ablations/
// gemm_tiles.txt): the small tiles win or tie up to M = 36,864 at every width / dtype (fp32 512-wide 187 vs 205 us, 256-wide 58 vs 78;
// bf16 14.9 vs 19.8) and tie at 73,728 -- four big tiles per CU is where the big ones start to pay
constexpr int GEMM_SMALL_BELOW = 1024;     // exact fp32 (the matrix pipe is the limit: the small tiles' extra fragment reads are free)
constexpr int GEMM_SMALL_BELOW_16 = 512;   // bf16 / bf16x3: the big tiles pay from two per CU (bf16x3 512-wide, M = 36,864: 91 vs 103 us)
static int g_gemm_small_below = -1;        // (rn_debug_gemm_small_below: the sweep tool moves the switch; < 0: the values above)
constexpr int SLAB_B = 128;            // bytes of K per row per slab
constexpr int ROW_B = SLAB_B + 16;     // padded LDS row stride: conflict-free ds_read_b128

template <typename T, int EPI, int MT, int NT, bool X3 = false>
__global__ __launch_bounds__(256, 2) void gemm_rowtile_kernel(const T* __restrict__ A, int lda,
                                                              const T* __restrict__ W, int ldw,
                                                              const float* __restrict__ bias,
                                                              const T* __restrict__ gate, int ldg,
                                                              T* __restrict__ C, int ldc, int M, int K) {
  constexpr int TM = 64 * MT, TN = 64 * NT;
  constexpr int CH = Elem<T>::kPer16B;
  constexpr int BKE = SLAB_B / (int)sizeof(T);
  static_assert(!X3 || sizeof(T) == 4, "the split arithmetic is a mode of the fp32 storage");
  typedef typename std::conditional<X3, bf16x8, typename Mma<T>::Frag>::type Frag;
  __shared__ __attribute__((aligned(16))) unsigned char lds[(TM + TN) * ROW_B];
  unsigned char* ldsA = lds;
  unsigned char* ldsW = lds + TM * ROW_B;

  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int wm = w & 1, wn = w >> 1;
  const long m0 = (long)blockIdx.x * TM;
  const int n0 = blockIdx.y * TN;
  const int srow = t >> 3, scc = t & 7;        // staging: 8 consecutive lanes cover one 128-byte row slab

  u32x4 ra[2 * MT], rw[2 * NT];
  const T* a_ptr[2 * MT];
  const T* w_ptr[2 * NT];
#pragma unroll
  for (int s = 0; s < 2 * MT; ++s) {
    long r = m0 + srow + 32 * s;
    if (r > M - 1) r = M - 1;                   // clamp: rows >= M are computed but never stored
    a_ptr[s] = A + r * lda + scc * CH;
  }
#pragma unroll
  for (int s = 0; s < 2 * NT; ++s) w_ptr[s] = W + (long)(n0 + srow + 32 * s) * ldw + scc * CH;

  auto gload = [&](int kt) {
#pragma unroll
    for (int s = 0; s < 2 * MT; ++s) ra[s] = *reinterpret_cast<const u32x4*>(a_ptr[s] + kt * BKE);
#pragma unroll
    for (int s = 0; s < 2 * NT; ++s) rw[s] = *reinterpret_cast<const u32x4*>(w_ptr[s] + kt * BKE);
  };
  // X3: a slab row is [32 hi bf16 | 32 lo bf16] (the same 128 bytes): chunk scc (4 floats) -> 8 bytes at 8 scc of each half
  auto split_store = [&](unsigned char* row, const u32x4 v) {
    const f32x4 x = __builtin_bit_cast(f32x4, v);        // (whole-vector cast: element-wise casts of v[i] were folded to v[0] by hipcc 7.2)
    const bf16 h0 = (bf16)x[0], h1 = (bf16)x[1], h2 = (bf16)x[2], h3 = (bf16)x[3];
    const bf16 l0 = (bf16)(x[0] - (float)h0), l1 = (bf16)(x[1] - (float)h1), l2 = (bf16)(x[2] - (float)h2), l3 = (bf16)(x[3] - (float)h3);
    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_;
    const bf16x4_ hi = {h0, h1, h2, h3}, lo = {l0, l1, l2, l3};
    *reinterpret_cast<bf16x4_*>(row + scc * 8) = hi;
    *reinterpret_cast<bf16x4_*>(row + 64 + scc * 8) = lo;
  };
  auto lstore = [&]() {
    if constexpr (X3) {
#pragma unroll
      for (int s = 0; s < 2 * MT; ++s) split_store(ldsA + (srow + 32 * s) * ROW_B, ra[s]);
#pragma unroll
      for (int s = 0; s < 2 * NT; ++s) split_store(ldsW + (srow + 32 * s) * ROW_B, rw[s]);
    } else {
#pragma unroll
      for (int s = 0; s < 2 * MT; ++s) *reinterpret_cast<u32x4*>(ldsA + (srow + 32 * s) * ROW_B + scc * 16) = ra[s];
#pragma unroll
      for (int s = 0; s < 2 * NT; ++s) *reinterpret_cast<u32x4*>(ldsW + (srow + 32 * s) * ROW_B + scc * 16) = rw[s];
    }
  };

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int nk = K / BKE;
  gload(0);
  lstore();
  __syncthreads();
  const unsigned char* fa_base = ldsA + (wm * 32 * MT + (lane & 31)) * ROW_B + (lane >> 5) * 16;
  const unsigned char* fw_base = ldsW + (wn * 32 * NT + (lane & 31)) * ROW_B + (lane >> 5) * 16;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) gload(kt + 1);
    if constexpr (X3) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {                     // 32 k per slab = two 16-k MFMA steps; hi at +0, lo at +64 of a row
        Frag fah[MT], fal[MT], fwh[NT], fwl[NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          fah[mt] = *reinterpret_cast<const Frag*>(fa_base + mt * 32 * ROW_B + ks * 32);
          fal[mt] = *reinterpret_cast<const Frag*>(fa_base + mt * 32 * ROW_B + 64 + ks * 32);
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          fwh[nt] = *reinterpret_cast<const Frag*>(fw_base + nt * 32 * ROW_B + ks * 32);
          fwl[nt] = *reinterpret_cast<const Frag*>(fw_base + nt * 32 * ROW_B + 64 + ks * 32);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {                // smallest terms first
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fwl[nt], fah[mt], acc[mt][nt], 0, 0, 0);
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fwh[nt], fal[mt], acc[mt][nt], 0, 0, 0);
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fwh[nt], fah[mt], acc[mt][nt], 0, 0, 0);
          }
      }
    } else {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      Frag fa[MT], fw[NT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) fa[mt] = *reinterpret_cast<const Frag*>(fa_base + mt * 32 * ROW_B + ks * 32);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) fw[nt] = *reinterpret_cast<const Frag*>(fw_base + nt * 32 * ROW_B + ks * 32);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) Mma<T>::mma(fw[nt], fa[mt], acc[mt][nt]);
    }
    }
    __syncthreads();
    if (kt + 1 < nk) {
      lstore();
      __syncthreads();
    }
  }

  // ---- epilogue: lane holds, per (mt, nt, g), features nb..nb+3 of pair row `row`
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const long row = m0 + wm * 32 * MT + mt * 32 + (lane & 31);
    if (row >= M) continue;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int nb = n0 + wn * 32 * NT + nt * 32 + 8 * g + 4 * (lane >> 5);
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = acc[mt][nt][4 * g + r];
        if constexpr (EPI == EPI_BIAS_RELU) {
          const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + nb);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r] + bv[r], 0.f);
        } else {
          T gt[4];
          if constexpr (sizeof(T) == 2) *reinterpret_cast<u32x2*>(gt) = *reinterpret_cast<const u32x2*>(gate + row * ldg + nb);
          else *reinterpret_cast<u32x4*>(gt) = *reinterpret_cast<const u32x4*>(gate + row * ldg + nb);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = is_pos<T>(gt[r]) ? v[r] : 0.f;
        }
        T o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = Elem<T>::from_f32(v[r]);
        if constexpr (sizeof(T) == 2) *reinterpret_cast<u32x2*>(C + row * ldc + nb) = *reinterpret_cast<const u32x2*>(o);
        else *reinterpret_cast<u32x4*>(C + row * ldc + nb) = *reinterpret_cast<const u32x4*>(o);
      }
    }
  }
}

template <int EPI>
static int gemm_launch(const void* A, int lda, const void* W, int ldw, const float* bias, const void* gate, int ldg,
                       void* C, int ldc, int dtype, int M, int N, int K, hipStream_t s, const char* who) {
  RN_CHECK_ARG(A && W && C && M > 0, "%s: bad pointer/size", who);
  RN_CHECK_ARG(dtype == RN_BF16 || dtype == RN_F32 || dtype == RN_F32X3, "%s: bad dtype %d", who, dtype);
  const int CH = dtype == RN_BF16 ? 8 : 4;
  RN_CHECK_ARG(N % 64 == 0, "%s: output width %d must be a multiple of 64", who, N);
  RN_CHECK_ARG(K % 64 == 0 && K > 0, "%s: reduction length %d must be a multiple of 64", who, K);
  RN_CHECK_ARG(lda % CH == 0 && ldw % CH == 0 && ldc % CH == 0 && lda >= K && ldw >= K && ldc >= N,
               "%s: leading dimensions (lda=%d ldw=%d ldc=%d) must be 16-byte multiples and cover K=%d / N=%d", who, lda,
               ldw, ldc, K, N);
  RN_CHECK_ARG(((uintptr_t)A | (uintptr_t)W | (uintptr_t)C | (uintptr_t)gate) % 16 == 0, "%s: pointers must be 16-byte aligned", who);
  // 128 x 256 tiles when they give every CU work; 64 x 64 ones for the short matrices (same sums, bit for bit)
  const bool small = N % TN_BIG != 0 || (long)cdiv(M, 128) * (N / TN_BIG) < (g_gemm_small_below >= 0 ? g_gemm_small_below : (dtype == RN_F32 ? GEMM_SMALL_BELOW : GEMM_SMALL_BELOW_16));
#define RN_GEMM_LAUNCH(T, MT, NT, X3)                                                                                          \
  gemm_rowtile_kernel<T, EPI, MT, NT, X3><<<dim3(cdiv(M, 64 * MT), N / (64 * NT)), 256, 0, s>>>(                                \
      (const T*)A, lda, (const T*)W, ldw, bias, (const T*)gate, ldg, (T*)C, ldc, M, K)
  if (dtype == RN_BF16) {
    if (small) RN_GEMM_LAUNCH(bf16, 1, 1, false);
    else RN_GEMM_LAUNCH(bf16, 2, 4, false);
  } else if (dtype == RN_F32X3) {
    if (small) RN_GEMM_LAUNCH(float, 1, 1, true);
    else RN_GEMM_LAUNCH(float, 2, 4, true);
  } else {
    if (small) RN_GEMM_LAUNCH(float, 1, 1, false);
    else RN_GEMM_LAUNCH(float, 2, 4, false);
  }
#undef RN_GEMM_LAUNCH
  RN_LAUNCH_CHECK(who);
  return 0;
}

extern "C" int rn_debug_gemm_small_below(int n) {
  const int was = g_gemm_small_below;
  g_gemm_small_below = n < 0 ? -1 : n;
  return was;
}

extern "C" int rn_g_linear_fwd(const void* A, int lda, const void* Wp, int ldw, const float* bias, void* H, int ldh,
                               int dtype, int M, int N, int K, void* stream) {
  RN_CHECK_ARG(bias && ((uintptr_t)bias % 16 == 0), "rn_g_linear_fwd: bias must be a 16-byte aligned fp32 vector");
  return gemm_launch<EPI_BIAS_RELU>(A, lda, Wp, ldw, bias, nullptr, 0, H, ldh, dtype, M, N, K, (hipStream_t)stream,
                                    "rn_g_linear_fwd");
}

extern "C" int rn_g_linear_bwd_dgrad(const void* dZ, int lddz, const void* Wt, int ldwt, const void* Hprev, int ldhp,
                                     void* dZprev, int lddzp, int dtype, int M, int N, int Kin, void* stream) {
  RN_CHECK_ARG(Hprev, "rn_g_linear_bwd_dgrad: Hprev is NULL");
  const int CH = dtype == RN_BF16 ? 8 : 4;
  RN_CHECK_ARG(ldhp % CH == 0 && ldhp >= Kin, "rn_g_linear_bwd_dgrad: bad ldhp=%d", ldhp);
  // out (M, Kin) = dZ (M, N) @ Wt(Kin, N)^T, gated by Hprev > 0
  return gemm_launch<EPI_GATE>(dZ, lddz, Wt, ldwt, nullptr, Hprev, ldhp, dZprev, lddzp, dtype, M, Kin, N,
                               (hipStream_t)stream, "rn_g_linear_bwd_dgrad");
}
