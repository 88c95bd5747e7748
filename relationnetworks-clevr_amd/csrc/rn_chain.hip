// Fused g_theta chains (model.py:130-152 and their autograd): one launch runs ALL g layers for a
// 128-row tile of the pair matrix and keeps the 256-wide tile in LDS between layers.
//
//   forward  (MODE_FWD): tile <- P rows;            per layer: tile = relu(tile @ W_l^T + b_l)
//            each activation is written to HBM exactly once (for the backward pass) and never read
//            back; the pair sum (model.py:151-152) is taken from the last tile while it is on chip.
//   backward (MODE_BWD): tile <- dxg[b] * (H_L > 0);  per layer: tile = (tile @ W_l[:, :256]) * (H_{l-1} > 0)
//            i.e. pair-sum broadcast + last ReLU gate + the whole dgrad chain; every dZ_l is written
//            once (wgrad reads it), every H_l is read once (as the gate).
//
//   Weights (<= 128 KB per layer, bf16) stream from L2 in 64-wide K slabs through a double-buffered,
//   padded LDS stage (register-staged prefetch one slab ahead).
//
// Workgroup = 512 threads (8 waves = 2 per SIMD), tile 128(M) x 256(N); wave (wm, wn) in a 2 x 4
// grid owns a 64 x 64 sub-tile = 2 x 2 MFMA tiles (v_mfma_f32_32x32x16_bf16), 64 accumulator
// registers.  Operand assignment is swapped (weights = A-operand) exactly as in rn_gemm.hip, so a
// lane ends up with 4 consecutive features of one pair row -> one ds_write_b64 into the LDS tile.
// The tile is then copied LDS -> HBM with 16-byte, fully row-contiguous stores.
#include <stdlib.h>

#include "rn_common.h"

namespace {
constexpr int CT_G = 256, CT_MAXL = 8;
constexpr int ACT_RS = CT_G * 2 + 16;        // 528 B: tile row stride (conflict-free b128 reads)
enum { MODE_FWD = 0, MODE_BWD = 1 };

struct ChainArgs {
  const bf16* W[CT_MAXL];                    // fwd: packed (256, K[l]);  bwd: transposed (256 kin, 256 n) of step s
  const float* bias[CT_MAXL];                // fwd only
  const bf16* gate[CT_MAXL];                 // bwd only: activation gating the output of step s, (M, 256)
  bf16* out[CT_MAXL];                        // fwd: H_l (may be null);  bwd: dZ after step s
  int K[CT_MAXL];                            // reduction length of step l (multiple of BK, <= 256)
  // backward prologue
  const bf16* HL;                            // last activation (M, 256)
  const float* dxg;                          // (B, 256) fp32
  bf16* out0;                                // dZ of the last layer (M, 256)
  int rows_per_b;                            // n*n
};
}  // namespace

// TM = tile rows (128 with 512 threads: one workgroup per CU; 64 with 256 threads: two co-resident
// workgroups per CU), BK = K-slab width.
template <int TM, int NT, int BK, int MODE>
__global__ __launch_bounds__(NT) void g_chain_kernel(const bf16* __restrict__ P, int ldp, ChainArgs a, int L,
                                                     float* __restrict__ xg_part, unsigned long long* __restrict__ trace) {
  constexpr int W_RS = BK * 2 + 16;             // weight slab row stride (144 / 80 B: conflict-free b128 reads)
  constexpr int ACT_BYTES = TM * ACT_RS;
  constexpr int WBUF_BYTES = CT_G * W_RS;
  constexpr int CPRW = BK / 8;                  // 16-byte chunks per weight-slab row
  constexpr int KSTEPS = BK / 16;
  constexpr int GM = TM / 64;                   // wave grid GM x GN, every wave owns 64 x 64
  static_assert(CT_G * CPRW / NT == 4 && NT / CPRW == 64 && (NT / 64) / GM == 4, "tile geometry");
  constexpr int BIAS_BYTES = (MODE == MODE_FWD) ? CT_MAXL / 2 * CT_G * 4 : 0;      // fwd: up to 4 layers of bias in LDS
  __shared__ __attribute__((aligned(16))) unsigned char lds[ACT_BYTES + 2 * WBUF_BYTES + BIAS_BYTES];
  unsigned char* act = lds;
  unsigned char* wbuf = lds + ACT_BYTES;
  float* bias_s = reinterpret_cast<float*>(lds + ACT_BYTES + 2 * WBUF_BYTES);

  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int wm = w % GM, wn = w / GM;
  const long m0 = (long)blockIdx.x * TM;
  // optional phase timestamps (s_memtime) of wave 0 of a few workgroups: tools/trace_chain.py
  int tp = 0;
  const bool tracing = trace != nullptr && t == 0 && (blockIdx.x % 397) == 0;
  auto stamp = [&]() {
    if (tracing) trace[(blockIdx.x / 397) * 32 + (tp++)] = __builtin_amdgcn_s_memtime();
  };
  stamp();

  if constexpr (MODE == MODE_FWD) {
    const bool bias_in_lds = L <= CT_MAXL / 2;
    if (bias_in_lds)
      for (int c = t; c < L * CT_G; c += NT) bias_s[c] = a.bias[c >> 8][c & 255];
    // ---- stage the P tile: TM rows x K0 columns -> tile[:, 0:K0]
    const int K0 = a.K[0];
    const int cpr = K0 >> 3;                              // 16-byte chunks per row
    const int total = TM * cpr;
    for (int c = t; c < total; c += NT) {
      const int r = c / cpr, cc = c - r * cpr;
      *reinterpret_cast<u32x4*>(act + r * ACT_RS + cc * 16) =
          *reinterpret_cast<const u32x4*>(P + (m0 + r) * ldp + cc * 8);
    }
  } else {
    // ---- dZ_L tile = dxg[b] * (H_L > 0)   (backward of the pair sum + last ReLU), also stored to HBM
    const int b = (int)(m0 / a.rows_per_b);               // a tile never straddles two questions
    const float* gb = a.dxg + (long)b * CT_G;
#pragma unroll
    for (int i = 0; i < TM * 32 / NT; ++i) {
      const int c = t + NT * i;
      const int r = c >> 5, cc = c & 31;
      const bf16x8 h = *reinterpret_cast<const bf16x8*>(a.HL + (m0 + r) * CT_G + cc * 8);
      const f32x4 g0 = *reinterpret_cast<const f32x4*>(gb + cc * 8);
      const f32x4 g1 = *reinterpret_cast<const f32x4*>(gb + cc * 8 + 4);
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        o[e] = (float)h[e] > 0.f ? (bf16)g0[e] : (bf16)0.f;
        o[e + 4] = (float)h[e + 4] > 0.f ? (bf16)g1[e] : (bf16)0.f;
      }
      *reinterpret_cast<bf16x8*>(act + r * ACT_RS + cc * 16) = o;
      *reinterpret_cast<bf16x8*>(a.out0 + (m0 + r) * CT_G + cc * 8) = o;
    }
  }
  // weight slab staging: 256 rows x 2*BK bytes, 4 chunks per thread; CPRW lanes cover one row slab
  const int srow = t / CPRW, scc = t % CPRW;
  u32x4 rw[4];
  auto gload = [&](const bf16* Wl, int ldw, int slab) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
      rw[s] = *reinterpret_cast<const u32x4*>(Wl + (long)(srow + 64 * s) * ldw + slab * BK + scc * 8);
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
      *reinterpret_cast<u32x4*>(wbuf + buf * WBUF_BYTES + (srow + 64 * s) * W_RS + scc * 16) = rw[s];
  };
  gload(a.W[0], a.K[0], 0);
  lstore(0);
  __syncthreads();
  stamp();

  const unsigned char* fa_base = act + (wm * 64 + (lane & 31)) * ACT_RS + (lane >> 5) * 16;
  const int fw_off = (wn * 64 + (lane & 31)) * W_RS + (lane >> 5) * 16;
  // LDS -> HBM copy of the finished tile: TM rows x 512 B, 16-byte chunks, row-contiguous.  It is issued
  // AFTER the next weight-slab loads of the following layer: vmcnt retires in order, so a load that is
  // younger than these stores could only be waited for together with them (HBM write latency).
  auto copy_out = [&](bf16* Ol) {
    if (Ol) {
#pragma unroll
      for (int i = 0; i < TM * 32 / NT; ++i) {
        const int c = t + NT * i;
        const int r = c >> 5, cc = c & 31;
        *reinterpret_cast<u32x4*>(Ol + (m0 + r) * CT_G + cc * 8) = *reinterpret_cast<const u32x4*>(act + r * ACT_RS + cc * 16);
      }
    }
  };
  int cur = 0;
  for (int l = 0; l < L; ++l) {
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    // backward: fetch this step's ReLU gate (the lane's 2x2x4 groups of 4 features) early, use it in the epilogue
    u32x2 gt[2][2][4];
    if constexpr (MODE == MODE_BWD) {
      const bf16* gl = a.gate[l];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int g = 0; g < 4; ++g)
            gt[mt][nt][g] = *reinterpret_cast<const u32x2*>(
                gl + (m0 + wm * 64 + mt * 32 + (lane & 31)) * CT_G + wn * 64 + nt * 32 + 8 * g + 4 * (lane >> 5));
    }
    const int ns = a.K[l] / BK;
    for (int s = 0; s < ns; ++s) {
      const bool last_slab = (s == ns - 1);
      const bool has_next = !(last_slab && l == L - 1);
      if (has_next) {
        if (last_slab) gload(a.W[l + 1], a.K[l + 1], 0);
        else gload(a.W[l], a.K[l], s + 1);
      }
      if (s == 0 && l > 0) copy_out(a.out[l - 1]);      // previous layer's tile (still intact in LDS until this layer's epilogue)
      const unsigned char* fw_base = wbuf + cur * WBUF_BYTES + fw_off;
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) {
        bf16x8 fa[2], fw[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
          fa[mt] = *reinterpret_cast<const bf16x8*>(fa_base + mt * 32 * ACT_RS + s * (2 * BK) + ks * 32);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) fw[nt] = *reinterpret_cast<const bf16x8*>(fw_base + nt * 32 * W_RS + ks * 32);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[nt], fa[mt], acc[mt][nt], 0, 0, 0);
      }
#ifdef RN_CHAIN_TRACE_SLABS
      if (l == 1) stamp();
#endif
      if (has_next) lstore(cur ^ 1);
#ifdef RN_CHAIN_TRACE_SLABS
      if (l == 1) stamp();
#endif
      __syncthreads();                  // (A) all reads of wbuf[cur] / this tile slab done; next slab visible
      cur ^= 1;
#ifdef RN_CHAIN_TRACE_SLABS
      if (l == 1) stamp();
#endif
    }
    stamp();
    // ---- epilogue -> bf16 -> tile in place (all waves are past barrier A)
    {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int row = wm * 64 + mt * 32 + (lane & 31);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int nb = wn * 64 + nt * 32 + 8 * g + 4 * (lane >> 5);
            bf16x4 o;
            if constexpr (MODE == MODE_FWD) {
              const f32x4 bv = (L <= CT_MAXL / 2) ? *reinterpret_cast<const f32x4*>(bias_s + l * CT_G + nb)
                                                  : *reinterpret_cast<const f32x4*>(a.bias[l] + nb);
#pragma unroll
              for (int r = 0; r < 4; ++r) o[r] = (bf16)fmaxf(acc[mt][nt][4 * g + r] + bv[r], 0.f);
            } else {
              union { u32x2 u; bf16x4 h; } gv;
              gv.u = gt[mt][nt][g];
#pragma unroll
              for (int r = 0; r < 4; ++r) o[r] = (float)gv.h[r] > 0.f ? (bf16)acc[mt][nt][4 * g + r] : (bf16)0.f;
            }
            *reinterpret_cast<bf16x4*>(act + row * ACT_RS + nb * 2) = o;
          }
        }
      }
    }
    __syncthreads();                    // (B) the new tile is visible
    stamp();
    if (l == L - 1) copy_out(a.out[l]);
  }
  stamp();
  // ---- forward: pair-sum partial of this tile = column sums of the bf16 tile (fp32, fixed order)
  if (MODE == MODE_FWD && xg_part) {
    constexpr int NH = NT / 256, RPH = TM / NH;           // NH row groups of RPH rows, one thread per column
    const int c = t & 255, h = t >> 8;
    float s = 0.f;
#pragma unroll 8
    for (int r = 0; r < RPH; ++r) s += (float)*reinterpret_cast<const bf16*>(act + (h * RPH + r) * ACT_RS + c * 2);
    if constexpr (NH == 2) {
      float* red = reinterpret_cast<float*>(wbuf);      // weight buffers are idle now (past barrier A/B)
      if (h == 1) red[c] = s;
      __syncthreads();
      if (h == 0) xg_part[(long)blockIdx.x * CT_G + c] = s + red[c];
    } else {
      xg_part[(long)blockIdx.x * CT_G + c] = s;
    }
  }
}

static unsigned long long* g_trace = nullptr;      // diagnostics only
extern "C" void rn_debug_set_chain_trace(void* buf) { g_trace = (unsigned long long*)buf; }

static int chain_tile_rows() {
  const char* te = getenv("RN_CHAIN_TILE");       // 128 (default; measured 279 us vs 326 us for 64) or 64
  return (te && atoi(te) == 64) ? 64 : 128;
}
extern "C" int rn_g_chain_tile(void) { return chain_tile_rows(); }

extern "C" int rn_g_chain_fwd(const void* P, int ldp, const void* const* Wp, const float* const* bias,
                              void* const* H, const int* K, float* xg_part, int dtype, int M, int L, int G,
                              void* stream) {
  RN_CHECK_ARG(P && Wp && bias && K && M > 0, "rn_g_chain_fwd: bad pointer/size");
  RN_CHECK_ARG(dtype == RN_BF16, "rn_g_chain_fwd: only the bf16 storage mode has a fused chain (dtype=%d)", dtype);
  RN_CHECK_ARG(G == CT_G && L >= 1 && L <= CT_MAXL, "rn_g_chain_fwd: needs G == 256 and 1 <= L <= %d (G=%d L=%d)", CT_MAXL, G, L);
  const int TM = chain_tile_rows();
  RN_CHECK_ARG(M % TM == 0, "rn_g_chain_fwd: M=%d must be a multiple of %d", M, TM);
  RN_CHECK_ARG(ldp % 8 == 0 && ldp >= K[0] && ((uintptr_t)P % 16 == 0), "rn_g_chain_fwd: bad P layout");
  ChainArgs a;
  memset(&a, 0, sizeof(a));
  for (int l = 0; l < L; ++l) {
    RN_CHECK_ARG(Wp[l] && bias[l], "rn_g_chain_fwd: layer %d weight/bias is NULL", l);
    RN_CHECK_ARG(K[l] % 64 == 0 && K[l] >= 64 && K[l] <= 256 && (l == 0 || K[l] == CT_G),
                 "rn_g_chain_fwd: layer %d reduction length %d unsupported", l, K[l]);
    RN_CHECK_ARG(((uintptr_t)Wp[l] | (uintptr_t)bias[l] | (uintptr_t)(H ? H[l] : nullptr)) % 16 == 0,
                 "rn_g_chain_fwd: layer %d pointers must be 16-byte aligned", l);
    a.W[l] = (const bf16*)Wp[l];
    a.bias[l] = bias[l];
    a.out[l] = H ? (bf16*)H[l] : nullptr;
    a.K[l] = K[l];
  }
  hipStream_t s = (hipStream_t)stream;
  if (TM == 128) g_chain_kernel<128, 512, 64, MODE_FWD><<<M / 128, 512, 0, s>>>((const bf16*)P, ldp, a, L, xg_part, g_trace);
  else g_chain_kernel<64, 256, 32, MODE_FWD><<<M / 64, 256, 0, s>>>((const bf16*)P, ldp, a, L, xg_part, g_trace);
  RN_LAUNCH_CHECK("rn_g_chain_fwd");
  return 0;
}

extern "C" int rn_g_chain_bwd(const void* HL, const float* dxg, const void* const* Wt, const void* const* Hgate,
                              void* const* dZ, int dtype, int M, int rows_per_question, int L, int G, void* stream) {
  RN_CHECK_ARG(HL && dxg && Wt && Hgate && dZ && M > 0, "rn_g_chain_bwd: bad pointer/size");
  RN_CHECK_ARG(dtype == RN_BF16, "rn_g_chain_bwd: only the bf16 storage mode has a fused chain (dtype=%d)", dtype);
  RN_CHECK_ARG(G == CT_G && L >= 2 && L <= CT_MAXL, "rn_g_chain_bwd: needs G == 256 and 2 <= L <= %d (G=%d L=%d)", CT_MAXL, G, L);
  RN_CHECK_ARG(M % 128 == 0 && rows_per_question % 128 == 0 && M % rows_per_question == 0,
               "rn_g_chain_bwd: M=%d and rows per question=%d must be multiples of 128", M, rows_per_question);
  ChainArgs a;
  memset(&a, 0, sizeof(a));
  RN_CHECK_ARG(dZ[0] && (((uintptr_t)HL | (uintptr_t)dxg | (uintptr_t)dZ[0]) % 16 == 0), "rn_g_chain_bwd: bad HL/dxg/dZ[0]");
  a.HL = (const bf16*)HL;
  a.dxg = dxg;
  a.out0 = (bf16*)dZ[0];
  a.rows_per_b = rows_per_question;
  for (int s = 0; s + 1 < L; ++s) {                      // L-1 dgrad steps: step s goes through layer L-1-s
    RN_CHECK_ARG(Wt[s] && Hgate[s] && dZ[s + 1], "rn_g_chain_bwd: step %d has a NULL pointer", s);
    RN_CHECK_ARG(((uintptr_t)Wt[s] | (uintptr_t)Hgate[s] | (uintptr_t)dZ[s + 1]) % 16 == 0,
                 "rn_g_chain_bwd: step %d pointers must be 16-byte aligned", s);
    a.W[s] = (const bf16*)Wt[s];
    a.gate[s] = (const bf16*)Hgate[s];
    a.out[s] = (bf16*)dZ[s + 1];
    a.K[s] = CT_G;
  }
  g_chain_kernel<128, 512, 64, MODE_BWD><<<M / 128, 512, 0, (hipStream_t)stream>>>(nullptr, 0, a, L - 1, nullptr, g_trace);
  RN_LAUNCH_CHECK("rn_g_chain_bwd");
  return 0;
}
