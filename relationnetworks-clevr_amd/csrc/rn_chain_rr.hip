// Register-resident fused g_theta chain (model.py:130-152): the headline-shape kernel.
//
// A workgroup is 4 waves, ONE per SIMD, each with the whole 512-register file.  A wave owns 64 pair
// rows for ALL four layers: with swapped MFMA operands (weights = A, activations = B) the 32x32 output
// block D[feature][row] leaves a lane holding features {8j + 4h + r} of row n -- after bias + ReLU +
// bf16 packing those registers ARE the B operand of the next layer's MFMA, provided the next layer's
// weights are packed with the matching K permutation (rn_pack_matrix_frag).  The activation therefore
// never goes through LDS between layers; LDS traffic is the weight stream only (one 1-KB A fragment
// feeds 2 MFMAs = 16 B/clk/wave, a quarter of the LDS read rate) plus the staging of the rows that are
// copied to HBM for the backward pass.
//
//   per wave:  act[2][16 k-steps][2 row blocks] x 4 VGPR (ping-pong, 256 regs)  +  2 x 2 accumulators (64)
//              + 8-deep A-fragment ring (32)
//   per stage: one 32-feature output block `ob` of one layer = 16 KB of weights = 16 fragments x 2 MFMAs.
//              The epilogue of block ob-1 (bias, ReLU, pack, stage, copy-out) is interleaved with the MFMAs
//              of block ob, at most ~5 single-issue instructions per MFMA gap (MI355X_MICROARCH.md).
//   weights:   fragment-major images stream L2 -> LDS by LDS-DMA into an 8-slot ring, 7 stages ahead; one
//              counted s_waitcnt vmcnt + s_barrier per stage (vmcnt retires in order and counts the stores).
//              8 blocks per layer == 8 slots, so every LDS address is a compile-time constant.
//   stores:    H_l rows are staged per block as [64 rows][64 B] and written with 16-byte row-contiguous
//              stores (a lane-per-row store would touch 32 cache lines per instruction).
//   pair sum:  the LAST layer runs with the operands un-swapped, D[row][feature]: a lane then owns one feature
//              of 16 rows and the pair sum is an in-lane fp32 add of the un-rounded activations; one partial
//              row per wave (32 pair rows) goes to xg_part.
#include <stdlib.h>

#include "rn_common.h"

namespace {
constexpr int RR_G = 256, RR_L = 4, RR_TM = 256, RR_NT = 512;
constexpr int RR_NW = RR_NT / 64, RR_WR = RR_TM / RR_NW;        // 8 waves, 32 pair rows each
constexpr int RR_DPW = 16 / RR_NW;                      // LDS-DMA pieces (1 KB) per wave and stage
constexpr int RR_RD = 4;                                // A-fragment read-ahead (register ring)
constexpr int RR_NSLOT = 8, RR_LA = RR_NSLOT - 1;      // ring slots / stages of look-ahead
constexpr int RR_STAGE = 16 * 1024;                    // one output block of weights: 16 fragments x 1 KB
constexpr int RR_SRS = 80;                             // staging row stride: 64 B of features + 16
constexpr int RR_STG = RR_WR * RR_SRS;                 // per wave
// small tables first: every ds_* address is then one of a few lane-constant VGPRs + a 16-bit immediate
constexpr int RR_OFF_BIAS = 0;
constexpr int RR_OFF_STG = RR_OFF_BIAS + RR_L * RR_G * 4;
constexpr int RR_OFF_RING = RR_OFF_STG + RR_NW * RR_STG;
constexpr int RR_LDS = RR_OFF_RING + RR_NSLOT * RR_STAGE;
static_assert(RR_LDS <= 160 * 1024, "LDS budget");

typedef u32x4 Frag;                                     // 8 bf16 (raw bits)
typedef __attribute__((address_space(1))) unsigned char gbl_u8;
typedef __attribute__((address_space(1))) const unsigned char gbl_cu8;
typedef __attribute__((ext_vector_type(2))) short s16x2;

struct RRArgs {
  const bf16* W[RR_L];                                  // fragment-major (rn_pack_matrix_frag), 128 KB each
  const float* bias[RR_L];
  bf16* out[RR_L];                                      // H_l (M, 256) or null
};

template <int N> struct IC { static constexpr int value = N; };

typedef __attribute__((ext_vector_type(2))) float f32x2;
__device__ __forceinline__ unsigned relu_pack_bf16(float a, float b) {
  const f32x2 f = {a, b};
  const bf16x2 v = __builtin_convertvector(f, bf16x2);     // one v_cvt_pk_bf16_f32
  // ReLU on the packed pair: a negative bf16 is a negative int16
  s16x2 x = __builtin_bit_cast(s16x2, v);
  const s16x2 z = {0, 0};
  x = __builtin_elementwise_max(x, z);
  return __builtin_bit_cast(unsigned, x);
}
__device__ __forceinline__ float bf16lo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float bf16hi(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }
}  // namespace

// dst[((ob * 16 + ks) * 64 + lane) * 8 + e] = src[32 ob + lane % 32][kidx], 0 beyond (R, C)
//   natural  : kidx = 16 ks + 8 h + e                         (operand read straight from memory rows)
//   permuted : kidx = 32 (ks / 2) + 4 h + 8 (2 (ks % 2) + e / 4) + e % 4   (operand = previous MFMA output)
__global__ __launch_bounds__(256) void pack_frag_kernel(const float* __restrict__ src, long sr, long sc, int R, int C,
                                                        bf16* __restrict__ dst, int natural) {
  const int g = blockIdx.x * 256 + threadIdx.x;
  const int e = g & 7, lane = (g >> 3) & 63, ks = (g >> 9) & 15, ob = g >> 13;
  const int h = lane >> 5, m = 32 * ob + (lane & 31);
  const int kidx = natural ? 16 * ks + 8 * h + e : 32 * (ks >> 1) + 4 * h + 8 * (2 * (ks & 1) + (e >> 2)) + (e & 3);
  const float v = (m < R && kidx < C) ? src[(long)m * sr + (long)kidx * sc] : 0.f;
  dst[g] = (bf16)v;
}

extern "C" int rn_pack_matrix_frag(const float* src, long sr, long sc, int R, int C, void* dst, int natural, void* stream) {
  RN_CHECK_ARG(src && dst && R > 0 && R <= RR_G && C > 0 && C <= RR_G, "rn_pack_matrix_frag: needs 0 < R, C <= 256 (R=%d C=%d)", R, C);
  pack_frag_kernel<<<RR_G * RR_G / 256, 256, 0, (hipStream_t)stream>>>(src, sr, sc, R, C, (bf16*)dst, natural);
  RN_LAUNCH_CHECK("rn_pack_matrix_frag");
  return 0;
}

template <int NK0, bool STORE, bool XG, int ABL = 0>
__global__ __launch_bounds__(RR_NT) void g_chain_rr_kernel(const bf16* __restrict__ P, int ldp, RRArgs a,
                                                           float* __restrict__ xg_part, int ntiles) {
  static_assert(NK0 % 4 == 0 && NK0 >= 4 && NK0 <= 16, "layer-0 reduction length");
  __shared__ __attribute__((aligned(16))) unsigned char lds[RR_LDS];
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int n = lane & 31, h = lane >> 5;
  unsigned char* const stg = lds + RR_OFF_STG + w * RR_STG;
  float* const bias_s = reinterpret_cast<float*>(lds + RR_OFF_BIAS);
  // counted waits: at the top of stage s the weights of stage s+1 must have landed.  Younger than them are
  // 5 stages of weight requests (2 per wave and stage) and, when activations are stored, 2 stores per stage;
  // the first stages of a tile have no copy-out yet (the tail of the previous tile did it), hence the
  // smaller layer-0 count.  Waiting for MORE than necessary is always safe.
  constexpr int VM_L0 = STORE ? 16 : 10, VM_LX = STORE ? 20 : 10;

  const unsigned lane16 = (unsigned)lane * 16u;
  auto dma_piece = [&](const bf16* Wl, int ob2, int slot, int i) {
    // uniform base (SGPR pair) + one lane-constant 32-bit offset: no per-piece address registers
    unsigned z = 0;
    asm volatile("" : "+s"(z));                        // opaque 0: the base is computed AT the use (SALU), not hoisted and spilled
    gbl_cu8* ub = (gbl_cu8*)(reinterpret_cast<const unsigned char*>(Wl) + (z + ob2 * RR_STAGE + (RR_DPW * w + i) * 1024));
    asm volatile("" : "+s"(ub));                       // ... and stays an SGPR base (no per-piece VGPR address)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ub + lane16),
                                     (__attribute__((address_space(3))) void*)(lds + RR_OFF_RING + slot * RR_STAGE + (RR_DPW * w + i) * 1024), 16, 0, 0);
  };
  // the ring spans 128 KB but a ds_read immediate reaches 64 KB: three lane-constant bases, made opaque so
  // that the compiler does not materialise (and keep, and spill) one address register per far fragment
  typedef __attribute__((address_space(3))) unsigned char lds_u8;
  typedef __attribute__((address_space(3))) const Frag lds_frag;
  lds_u8* rbase[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    rbase[r] = (lds_u8*)lds + r * 65536 + lane16;
    asm volatile("" : "+v"(rbase[r]));
  }
  auto rd_frag = [&](int slot, int ks) -> Frag {
    const int abs = RR_OFF_RING + slot * RR_STAGE + ks * 1024, r = abs >> 16;
    return *reinterpret_cast<lds_frag*>(rbase[r] + (abs & 0xffff));
  };
  const unsigned prow_off = (unsigned)(n * ldp + 8 * h) * 2u;         // this lane's byte offset inside a wave's 32 pair rows
  auto load_row_frag = [&](long m0w, int ks) -> Frag {                // layer-0 operand straight from the pair rows
    gbl_cu8* base = (gbl_cu8*)reinterpret_cast<const unsigned char*>(P + m0w * ldp);
    asm volatile("" : "+s"(base));
    return *reinterpret_cast<__attribute__((address_space(1))) const Frag*>(base + prow_off + 32 * ks);
  };

  Frag actA[16], actB[16], ring[RR_RD];
  f32x16 acc[2];
  u32x4 co[2];
  f32x16 cinit;                                                       // bias of the current block, laid out like the accumulator

  int tile = blockIdx.x;
  if (tile >= ntiles) return;
#pragma unroll
  for (int s = 0; s < RR_LA; ++s)
#pragma unroll
    for (int i = 0; i < RR_DPW; ++i) dma_piece(a.W[0], s, s, i);
#pragma unroll
  for (int ks = 0; ks < NK0; ++ks) actA[ks] = load_row_frag((long)tile * RR_TM + RR_WR * w, ks);
  if (t < RR_G) {
#pragma unroll
    for (int l = 0; l < RR_L; ++l) bias_s[l * RR_G + t] = a.bias[l][t];
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();
#pragma unroll
  for (int r = 0; r < RR_RD; ++r) ring[r] = rd_frag(0, r);

  for (; tile < ntiles; tile += gridDim.x) {
    const long m0w = (long)tile * RR_TM + RR_WR * w;                 // this wave's first pair row
    const int tnext = tile + (int)gridDim.x < ntiles ? tile + (int)gridDim.x : tile;
    const long m0n = (long)tnext * RR_TM + RR_WR * w;
    float xs[8];                                                      // pair-sum partials: feature 32 ob + lane % 32, this lane's 16 rows
#pragma unroll
    for (int i = 0; i < 8; ++i) xs[i] = 0.f;

    // ---- epilogue pieces ---------------------------------------------------------------------------------
    // group j of output block (pl, pob): features 32 pob + 8 j + 4 h + {0..3} of row n
    auto bias_read = [&](int l, int ob) {                             // -> C operand of the block's first MFMA
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(bias_s + l * RR_G + 32 * ob + 8 * j + 4 * h);
#pragma unroll
        for (int r = 0; r < 4; ++r) cinit[4 * j + r] = b[r];
      }
    };
    auto epi_group = [&](int pl, int pob, int j, int phase_lo, int phase_hi, Frag* dst, f32x4 (&v)[4], u32x2 (&pk)[4]) {
      (void)v;
      if (phase_lo <= 1 && 1 <= phase_hi) {                           // (the bias came in through the accumulator)
        pk[j][0] = relu_pack_bf16(acc[pob & 1][4 * j + 0], acc[pob & 1][4 * j + 1]);
        pk[j][1] = relu_pack_bf16(acc[pob & 1][4 * j + 2], acc[pob & 1][4 * j + 3]);
      }
      if (phase_lo <= 2 && 2 <= phase_hi) {
        *reinterpret_cast<u32x2*>(stg + n * RR_SRS + 16 * j + 8 * h) = pk[j];
        if (dst) {
          dst[2 * pob + (j >> 1)][(j & 1) * 2 + 0] = pk[j][0];
          dst[2 * pob + (j >> 1)][(j & 1) * 2 + 1] = pk[j][1];
        }
      }
    };
    auto co_read = [&]() {
#pragma unroll
      for (int q = 0; q < 2; ++q) co[q] = *reinterpret_cast<const u32x4*>(stg + (16 * q + (lane >> 2)) * RR_SRS + (lane & 3) * 16);
    };
    const unsigned orow_off = (unsigned)((lane >> 2) * RR_G + (lane & 3) * 8) * 2u;
    auto co_store = [&](int cl, int cob, int q) {
      if constexpr (STORE) {
        gbl_u8* base = (gbl_u8*)reinterpret_cast<unsigned char*>(a.out[cl] + m0w * RR_G);
        asm volatile("" : "+s"(base));
        *reinterpret_cast<__attribute__((address_space(1))) u32x4*>(base + orow_off + (16 * q * RR_G + 32 * cob) * 2) = co[q];
      }
    };
    // LAST layer: operands un-swapped (activations = A, weights = B), so D[row][feature] leaves a lane with ONE
    // feature (32 pob + lane % 32) of 16 rows (8 (i / 4) + 4 h + i % 4): the pair sum (model.py:151-152) is an
    // in-lane fp32 add of the un-rounded activations -- no cross-lane traffic, no LDS.
    float b3 = 0.f;
    auto epi3_group = [&](int pob, int j, int phase_lo, int phase_hi, f32x4 (&v)[4]) {
      if (phase_lo <= 0 && 0 <= phase_hi) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[j][r] = fmaxf(acc[pob & 1][4 * j + r] + b3, 0.f);
      }
      if (phase_lo <= 1 && 1 <= phase_hi) {
        if constexpr (XG) xs[pob] += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
      }
      if (phase_lo <= 2 && 2 <= phase_hi) {
        if constexpr (STORE) {
#pragma unroll
          for (int r = 0; r < 4; ++r) *reinterpret_cast<bf16*>(stg + (8 * j + 4 * h + r) * RR_SRS + n * 2) = (bf16)v[j][r];
        }
      }
    };
    // ---- one layer = 8 stages ------------------------------------------------------------------------------
    auto layer = [&](auto lc, Frag (&in)[16], Frag (&out)[16]) {
      constexpr int l = decltype(lc)::value;
      constexpr int NK = (l == 0) ? NK0 : 16;
      constexpr int CPG = NK / 4;                                     // MFMA gaps per epilogue group
      constexpr int PF_PER = (NK0 + 7) / 8;                           // next-tile row loads per stage (last layer)
#pragma unroll
      for (int ob = 0; ob < 8; ++ob) {
        const int sidx = l * 8 + ob;
        const bool has_prev = sidx > 0;                               // (l, ob) == (0, 0): the tail of the last tile did it
        const int pl = ob ? l : l - 1, pob = ob ? ob - 1 : 7;
        const bool has_co = sidx >= 2 && STORE;
        const int cl = (sidx - 2) >> 3, cob = (sidx - 2) & 7;
        const int didx = sidx + RR_LA;                                // stage whose weights are requested now
        const int dl = (didx >> 3) & 3, dob = didx & 7;
        const int nob = (sidx + 1) & 7;                               // next stage (the read-ahead crosses into it)
        if (l < RR_L - 1) bias_read(l, ob);
        if (has_prev && pl == RR_L - 1) b3 = bias_s[pl * RR_G + 32 * pob + n];
        if (!(ABL & 1)) {
          if (l == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM_L0) : "memory");
          else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM_LX) : "memory");
          __builtin_amdgcn_s_barrier();
        }
        asm volatile("" ::: "memory");
        if (has_co) co_read();
        __builtin_amdgcn_sched_barrier(0);
        Frag* dst = nullptr;
        if (has_prev && pl < RR_L - 1) dst = ob ? out : in;
        f32x4 v[4];
        u32x2 pk[4];
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
          const int c = ks;
          const bf16x8 fw = __builtin_bit_cast(bf16x8, ring[ks % RR_RD]), fx = __builtin_bit_cast(bf16x8, in[ks]);
          const bf16x8 fa = (l == RR_L - 1) ? fx : fw, fb = (l == RR_L - 1) ? fw : fx;
          if (ks == 0 && l < RR_L - 1) {
            acc[ob & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, cinit, 0, 0, 0);
          } else if (ks == 0) {
            const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            acc[ob & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, z, 0, 0, 0);
          } else {
            acc[ob & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[ob & 1], 0, 0, 0);
          }
          // ---- fillers of this MFMA gap
          {                                                           // refill the ring slot just consumed, RR_RD fragments ahead
            const int f = ks + RR_RD;
            if (ABL & 4) {
            } else if (f < NK) ring[ks % RR_RD] = rd_frag(ob, f);
            else ring[ks % RR_RD] = rd_frag(nob, f - NK);
          }
          if (!(ABL & 2) && (c & 1) && (c >> 1) < RR_DPW) dma_piece(a.W[dl], dob, dob, c >> 1);
          if (has_prev && !(ABL & 8) && pl == RR_L - 1) {
            const int j = c / CPG, ph = c % CPG;
            if (ph < 3) epi3_group(pob, j, ph, ph, v);
          } else if (has_prev && !(ABL & 8)) {
            const int j = c / CPG, ph = c % CPG;
            if (CPG >= 3) {
              if (ph < 3) epi_group(pl, pob, j, ph, ph, dst, v, pk);
            } else if (CPG == 2) {
              if (ph == 0) epi_group(pl, pob, j, 0, 1, dst, v, pk);
              else epi_group(pl, pob, j, 2, 2, dst, v, pk);
            } else {
              epi_group(pl, pob, j, 0, 2, dst, v, pk);
            }
          }
          if (has_co && (c == 4 || c == 8)) {
            const int q = (c >> 2) - 1;
            co_store(cl, cob, q);
          }
          if (l == RR_L - 1 && (c & 1) == 0 && (c >> 1) < PF_PER) {   // next tile's pair rows -> the idle half of the ping-pong
            const int i = ob * PF_PER + (c >> 1);
            if (i < NK0) out[i] = load_row_frag(m0n, i);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    };
    layer(IC<0>{}, actA, actB);
    layer(IC<1>{}, actB, actA);
    layer(IC<2>{}, actA, actB);
    layer(IC<3>{}, actB, actA);

    // ---- tail: blocks (3, 6) and (3, 7) leave the chip
    {
      f32x4 v[4];
      if constexpr (STORE) {
        co_read();
#pragma unroll
        for (int q = 0; q < 2; ++q) co_store(RR_L - 1, 6, q);
      }
      b3 = bias_s[(RR_L - 1) * RR_G + 32 * 7 + n];
#pragma unroll
      for (int j = 0; j < 4; ++j) epi3_group(7, j, 0, 2, v);
      if constexpr (STORE) {
        co_read();
#pragma unroll
        for (int q = 0; q < 2; ++q) co_store(RR_L - 1, 7, q);
      }
      if constexpr (XG) {
#pragma unroll
        for (int ob = 0; ob < 8; ++ob) {
          const float tot = xs[ob] + __shfl_xor(xs[ob], 32);          // the two 16-row halves of this wave's 32 rows
          if (h == 0) xg_part[((long)tile * RR_NW + w) * RR_G + 32 * ob + n] = tot;
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // trailing (unused) weight requests
}

static int rr_num_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) n = p.multiProcessorCount;
    if (n <= 0) n = 256;
  }
  return n;
}

extern "C" int rn_g_chain_rr_tile(void) { return RR_TM; }

extern "C" int rn_g_chain_fwd_rr(const void* P, int ldp, const void* const* Wf, const float* const* bias, void* const* H,
                                 int K0, float* xg_part, int M, int L, int G, void* stream) {
  RN_CHECK_ARG(P && Wf && bias && M > 0, "rn_g_chain_fwd_rr: bad pointer/size");
  RN_CHECK_ARG(G == RR_G && L == RR_L, "rn_g_chain_fwd_rr: needs G == 256 and L == 4 (G=%d L=%d)", G, L);
  RN_CHECK_ARG(M % RR_TM == 0, "rn_g_chain_fwd_rr: M=%d must be a multiple of %d", M, RR_TM);
  RN_CHECK_ARG(K0 == 192 || K0 == 256, "rn_g_chain_fwd_rr: layer-0 reduction length %d unsupported (192 or 256)", K0);
  RN_CHECK_ARG(ldp % 8 == 0 && ldp >= K0 && ((uintptr_t)P % 16 == 0), "rn_g_chain_fwd_rr: bad P layout");
  RRArgs a;
  bool store = false;
  for (int l = 0; l < RR_L; ++l) {
    RN_CHECK_ARG(Wf[l] && bias[l], "rn_g_chain_fwd_rr: layer %d weight/bias is NULL", l);
    RN_CHECK_ARG(((uintptr_t)Wf[l] | (uintptr_t)bias[l] | (uintptr_t)(H ? H[l] : nullptr)) % 16 == 0,
                 "rn_g_chain_fwd_rr: layer %d pointers must be 16-byte aligned", l);
    a.W[l] = (const bf16*)Wf[l];
    a.bias[l] = bias[l];
    a.out[l] = H ? (bf16*)H[l] : nullptr;
    store = store || a.out[l];
  }
  RN_CHECK_ARG(store || xg_part, "rn_g_chain_fwd_rr: nothing to compute (no H, no xg_part)");
  const int ntiles = M / RR_TM;
  const int grid = ntiles < rr_num_cus() ? ntiles : rr_num_cus();
  hipStream_t s = (hipStream_t)stream;
  const bf16* Pb = (const bf16*)P;
#define RN_RR(NK0) \
  do { \
    if (store && xg_part) g_chain_rr_kernel<NK0, true, true><<<grid, RR_NT, 0, s>>>(Pb, ldp, a, xg_part, ntiles); \
    else if (store) g_chain_rr_kernel<NK0, true, false><<<grid, RR_NT, 0, s>>>(Pb, ldp, a, xg_part, ntiles); \
    else g_chain_rr_kernel<NK0, false, true><<<grid, RR_NT, 0, s>>>(Pb, ldp, a, xg_part, ntiles); \
  } while (0)
  const char* ae = getenv("RN_RR_ABL");                    // diagnostics: timing-only ablations (results are wrong)
  const int abl = ae ? atoi(ae) : 0;
  if (abl && K0 == 192) {
    switch (abl) {
      case 1: g_chain_rr_kernel<12, false, true, 1><<<grid, RR_NT, 0, s>>>(Pb, ldp, a, xg_part, ntiles); break;
      case 2: g_chain_rr_kernel<12, false, true, 2><<<grid, RR_NT, 0, s>>>(Pb, ldp, a, xg_part, ntiles); break;
      case 3: g_chain_rr_kernel<12, false, true, 3><<<grid, RR_NT, 0, s>>>(Pb, ldp, a, xg_part, ntiles); break;
      case 4: g_chain_rr_kernel<12, false, true, 4><<<grid, RR_NT, 0, s>>>(Pb, ldp, a, xg_part, ntiles); break;
      case 7: g_chain_rr_kernel<12, false, true, 7><<<grid, RR_NT, 0, s>>>(Pb, ldp, a, xg_part, ntiles); break;
      case 8: g_chain_rr_kernel<12, false, true, 8><<<grid, RR_NT, 0, s>>>(Pb, ldp, a, xg_part, ntiles); break;
      default: g_chain_rr_kernel<12, false, true, 15><<<grid, RR_NT, 0, s>>>(Pb, ldp, a, xg_part, ntiles); break;
    }
  } else if (K0 == 192) RN_RR(12);
  else RN_RR(16);
#undef RN_RR
  RN_LAUNCH_CHECK("rn_g_chain_fwd_rr");
  return 0;
}
