// Fused g_theta forward chain (model.py:130-152): one launch runs ALL g layers for a 128-row tile
// of the pair matrix and keeps the 256-wide activation tile in LDS between layers, so each
// activation is written to HBM exactly once (for the backward pass) and never read back, and
// the pair sum (model.py:151-152) is taken from the last tile while it is still on chip.
//
//   HBM traffic / step:  read P (M x K0) + write H_1..H_L (M x 256 each), vs. read+write of every
//   activation in the per-layer kernels (rn_gemm.hip).  Weights (<= 128 KB per layer, bf16) stream
//   from L2 in 64-wide K slabs through a double-buffered, padded LDS stage.
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
constexpr int ACT_RS = CT_G * 2 + 16;        // 528 B: act tile row stride (conflict-free b128 reads)

struct ChainArgs {
  const bf16* W[CT_MAXL];
  const float* bias[CT_MAXL];
  bf16* H[CT_MAXL];                          // may be null: activation not stored (inference)
  int K[CT_MAXL];                            // padded reduction length of layer l (multiple of 64, <= 256)
};
}  // namespace

// TM = tile rows (128 with 512 threads: one workgroup per CU; 64 with 256 threads: two co-resident
// workgroups per CU that overlap each other's load / epilogue / store phases), BK = K-slab width.
template <int TM, int NT, int BK>
__global__ __launch_bounds__(NT) void g_chain_fwd_kernel(const bf16* __restrict__ P, int ldp, ChainArgs a, int L,
                                                         float* __restrict__ xg_part, int abl) {
  constexpr int W_RS = BK * 2 + 16;             // weight slab row stride (144 / 80 B: conflict-free b128 reads)
  constexpr int ACT_BYTES = TM * ACT_RS;
  constexpr int WBUF_BYTES = CT_G * W_RS;
  constexpr int CPRW = BK / 8;                  // 16-byte chunks per weight-slab row
  constexpr int KSTEPS = BK / 16;
  constexpr int GM = TM / 64;                   // wave grid GM x GN, every wave owns 64 x 64
  static_assert(CT_G * CPRW / NT == 4 && NT / CPRW == 64 && (NT / 64) / GM == 4, "tile geometry");
  __shared__ __attribute__((aligned(16))) unsigned char lds[ACT_BYTES + 2 * WBUF_BYTES];
  unsigned char* act = lds;
  unsigned char* wbuf = lds + ACT_BYTES;

  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int wm = w % GM, wn = w / GM;
  const long m0 = (long)blockIdx.x * TM;

  // ---- stage the P tile: 128 rows x K0 columns -> act[:, 0:K0]
  {
    const int K0 = a.K[0];
    const int cpr = K0 >> 3;                              // 16-byte chunks per row
    const int total = TM * cpr;
    for (int c = t; c < total; c += NT) {
      const int r = c / cpr, cc = c - r * cpr;
      *reinterpret_cast<u32x4*>(act + r * ACT_RS + cc * 16) =
          *reinterpret_cast<const u32x4*>(P + (m0 + r) * ldp + cc * 8);
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

  const unsigned char* fa_base = act + (wm * 64 + (lane & 31)) * ACT_RS + (lane >> 5) * 16;
  const int fw_off = (wn * 64 + (lane & 31)) * W_RS + (lane >> 5) * 16;
  int cur = 0;
  for (int l = 0; l < L; ++l) {
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const int ns = a.K[l] / BK;
    for (int s = 0; s < ns; ++s) {
      const bool last_slab = (s == ns - 1);
      const bool has_next = !(last_slab && l == L - 1) && !(abl & 1);   // abl: timing ablations only (RN_CHAIN_ABLATE)
      if (has_next) {
        if (last_slab) gload(a.W[l + 1], a.K[l + 1], 0);
        else gload(a.W[l], a.K[l], s + 1);
      }
      const unsigned char* fw_base = wbuf + cur * WBUF_BYTES + fw_off;
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) {
        bf16x8 fa[2], fw[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
          fa[mt] = *reinterpret_cast<const bf16x8*>(fa_base + mt * 32 * ACT_RS + s * (2 * BK) + ks * 32);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) fw[nt] = *reinterpret_cast<const bf16x8*>(fw_base + nt * 32 * W_RS + ks * 32);
        if (!(abl & 4)) {
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[nt], fa[mt], acc[mt][nt], 0, 0, 0);
        } else {
          asm volatile("" ::"v"(fw[0]), "v"(fw[1]), "v"(fa[0]), "v"(fa[1]));
        }
      }
      if (has_next) lstore(cur ^ 1);
      __syncthreads();                  // (A) all reads of wbuf[cur] / this act slab done; next slab visible
      cur ^= 1;
    }
    // ---- epilogue: bias + ReLU -> bf16 -> act tile in place (all waves are past barrier A)
    const float* bl = a.bias[l];
    if (!(abl & 2))
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int row = wm * 64 + mt * 32 + (lane & 31);
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int nb = wn * 64 + nt * 32 + 8 * g + 4 * (lane >> 5);
          const f32x4 bv = *reinterpret_cast<const f32x4*>(bl + nb);
          bf16x4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = (bf16)fmaxf(acc[mt][nt][4 * g + r] + bv[r], 0.f);
          *reinterpret_cast<bf16x4*>(act + row * ACT_RS + nb * 2) = o;
        }
      }
    }
    __syncthreads();                    // (B) the new activation tile is visible
    // ---- copy the tile LDS -> HBM: TM rows x 512 B, 8 x 16-byte chunks per thread, row-contiguous
    bf16* Hl = a.H[l];
    if (Hl) {
#pragma unroll
      for (int i = 0; i < TM * 32 / NT; ++i) {
        const int c = t + NT * i;
        const int r = c >> 5, cc = c & 31;
        *reinterpret_cast<u32x4*>(Hl + (m0 + r) * CT_G + cc * 8) = *reinterpret_cast<const u32x4*>(act + r * ACT_RS + cc * 16);
      }
    }
  }
  // ---- pair-sum partial of this tile: column sums of the bf16 tile (fp32 accumulate, fixed order)
  if (xg_part) {
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
    a.H[l] = H ? (bf16*)H[l] : nullptr;
    a.K[l] = K[l];
  }
  const char* ab = getenv("RN_CHAIN_ABLATE");
  const int abl = ab ? atoi(ab) : 0;
  if (TM == 128) g_chain_fwd_kernel<128, 512, 64><<<M / 128, 512, 0, (hipStream_t)stream>>>((const bf16*)P, ldp, a, L, xg_part, abl);
  else g_chain_fwd_kernel<64, 256, 32><<<M / 64, 256, 0, (hipStream_t)stream>>>((const bf16*)P, ldp, a, L, xg_part, abl);
  RN_LAUNCH_CHECK("rn_g_chain_fwd");
  return 0;
}
