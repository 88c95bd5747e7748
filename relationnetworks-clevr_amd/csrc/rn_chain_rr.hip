// Register-resident fused g_theta chains (model.py:130-152 and their backward): the headline-shape kernels.
//
// A workgroup is 8 waves, two per SIMD, each wave with 256 registers and 32 pair rows for ALL four layers (tile = 256 rows,
// persistent workgroups).  With swapped MFMA operands (weights = A, activations = B) the 32x32 output block D[feature][row]
// leaves a lane holding features {8j + 4h + r} of row n -- after bias + ReLU + 16-bit packing those registers ARE the B operand
// of the next layer's MFMA, provided the next layer's weights are packed with the matching K permutation
// (rn_pack_matrix_frag_many).  The activation therefore never goes through LDS between layers; LDS carries the weight stream
// (fragment-major 16-KB blocks, L2 -> LDS by LDS-DMA into a ring, one counted s_waitcnt vmcnt + s_barrier per stage) and the
// staging of what is copied to HBM for the backward pass.
//
//   per wave:  act[2][16 k-steps] x 4 VGPR (ping-pong, 128 regs) + 2 accumulators (32) + a fragment read-ahead ring
//   per stage: one 32-feature output block `ob` of one layer = 16 fragments x 1 MFMA (32 rows).  The epilogue of block ob-1
//              (bias, ReLU, pack, masks, staging, copy-out) is sliced over the MFMA gaps of block ob (sched_barrier per gap).
//   stores:    H_l copies leave as ROW-BLOCKED images (the layout of their only reader, rn_wgrad_blocked.hip), e4m3 or 16-bit.
//   pair sum:  the LAST layer runs with the operands un-swapped, D[row][feature]: a lane then owns one feature of 16 rows and the
//              pair sum is an in-lane fp32 add of the un-rounded activations; one partial row per tile goes to xg_part.
//
// Kernels here: g_chain_rr_f16s_kernel (forward, the "f16s" arithmetic on the FACTORED first layer -- the only forward chain;
// the round-1..3 bf16 and pair-matrix variants were removed in round 4, the per-layer kernels of rn_gemm.hip cover what they
// covered) and g_chain_rr_bwd_kernel (backward, bf16, from the forward's lane masks; RED: with the pair-axis reductions on chip).
#include "rn_common.h"

namespace {
constexpr int RR_G = 256, RR_L = 4, RR_TM = 256, RR_NT = 512;
constexpr int RR_PRIO_DEFAULT = 0;                     // static priority for waves 4..7 (see the kernels): measured, not adopted unless it wins
constexpr int RR_NW = RR_NT / 64, RR_WR = RR_TM / RR_NW;        // 8 waves, 32 pair rows each
constexpr int RR_DPW = 16 / RR_NW;                      // LDS-DMA pieces (1 KB) per wave and stage
constexpr int RR_RD = 4;                                // A-fragment read-ahead (register ring)
constexpr int RR_NSLOT = 8, RR_LA = RR_NSLOT - 1;      // ring slots / stages of look-ahead
constexpr int RR_STAGE = 16 * 1024;                    // one output block of weights: 16 fragments x 1 KB
constexpr int RR_SRS = 80;                             // staging row stride: 64 B of features + 16
constexpr int RR_SRS8 = 72;                            // ... of the e4m3 rows (64 B + 8: two-way on the dword writes = free, MI355X_MICROARCH.md)
constexpr int RR_STG = RR_WR * RR_SRS;                 // per wave
// small tables first: every ds_* address is then one of a few lane-constant VGPRs + a 16-bit immediate
constexpr int RR_OFF_BIAS = 0;
constexpr int RR_OFF_VC = RR_OFF_BIAS + RR_L * RR_G * 4;     // per-wave layer-0 bias row of the factored first layer (ALG0)
constexpr int RR_OFF_STG = RR_OFF_VC + RR_NW * RR_G * 4;
constexpr int RR_OFF_RING = RR_OFF_STG + RR_NW * RR_STG;
constexpr int RR_LDS = RR_OFF_RING + RR_NSLOT * RR_STAGE;
static_assert(RR_LDS <= 160 * 1024, "LDS budget");

typedef u32x4 Frag;                                     // 8 bf16 (raw bits)
typedef __attribute__((address_space(1))) unsigned char gbl_u8;
typedef __attribute__((address_space(1))) const unsigned char gbl_cu8;
typedef __attribute__((ext_vector_type(2))) short s16x2;

typedef unsigned long long u64;
typedef __attribute__((address_space(3))) unsigned char lds_u8;
// the copies that leave for the weight gradient are written once and read once, half a step later: non-temporal stores keep them
// from walking the other kernels' lines out of L2 / the Infinity Cache (variant builds: -DRR_FWD_STORE_T / -DRR_BWD_STORE_T = plain)
#ifdef RR_FWD_STORE_T
#define RR_FWD_STORE(v, p) (*(p) = (v))
#else
#define RR_FWD_STORE(v, p) __builtin_nontemporal_store(v, p)
#endif
#ifdef RR_BWD_STORE_T
#define RR_BWD_STORE(v, p) (*(p) = (v))
#else
#define RR_BWD_STORE(v, p) __builtin_nontemporal_store(v, p)
#endif
typedef __attribute__((address_space(3))) const Frag lds_frag;

struct RRBwdArgs {                                      // the per-layer buffers are equally spaced (checked on the host):
  const bf16* W;                                        // step s: fragment-major W_{3-s}^T at W + s * w_stride
  const u64* mask;                                      // lane masks of layer l (forward kernel) at mask + l * mask_stride
  bf16* dZ;                                             // dZ[s] = gradient of the pre-activation of layer 3-s, (M, 256), at dZ + s * dz_stride
  long w_stride, mask_stride, dz_stride;                // in elements
  const float* dxg;                                     // (B, 256) fp32
  int rows_per_b;
  int prio;
};
typedef __attribute__((ext_vector_type(16))) unsigned u32x16;
// 16 lane masks (32 dwords) -> SGPRs.  Inline asm: a compiler-visible scalar load would make every LDS wait a
// full lgkmcnt(0) drain while it is in flight (SMEM returns out of order).  The destination is ready only after
// mask_wait() -- nothing may read it before.
__device__ __forceinline__ void mask_load(const u64* p, u32x16& a, u32x16& b) {
  asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx16 %1, %2, 0x40" : "=&s"(a), "=&s"(b) : "s"(p) : "memory");
}
__device__ __forceinline__ void mask_wait(u32x16& a, u32x16& b) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a), "+s"(b)::"memory"); }
__device__ __forceinline__ bool gate_bit(const u32x16& a, const u32x16& b, int i) {
  const u32x16& v = i < 8 ? a : b;
  const u64 m = ((u64)v[2 * (i & 7) + 1] << 32) | v[2 * (i & 7)];
  return __builtin_amdgcn_inverse_ballot_w64(m);
}

// What both kernels share: the weight stream (LDS-DMA ring) and its fragment reads.
struct RRCore {
  unsigned char* lds;
  int lane, w, n, h;
  unsigned lane16;
  lds_u8* rbase[3];
  __device__ __forceinline__ void init(unsigned char* l) {
    lds = l;
    lane = threadIdx.x & 63;
    w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    n = lane & 31;
    h = lane >> 5;
    lane16 = (unsigned)lane * 16u;
    // the ring spans 128 KB but a ds_read immediate reaches 64 KB: three lane-constant bases, made opaque so
    // that the compiler does not materialise (and keep, and spill) one address register per far fragment
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      rbase[r] = (lds_u8*)lds + r * 65536 + lane16;
      asm volatile("" : "+v"(rbase[r]));
    }
  }
  // The requests of ONE ring stage that this wave issues: NPC consecutive 1-KB pieces from `src` (wave-uniform: SGPRs) to the LDS
  // byte address wl + DST (wl: the wave's slice of the ring, an SGPR; DST: compile-time) -- SGPR base + one lane-constant 32-bit
  // offset, LDS address in M0.  The instruction offset moves BOTH addresses (tools/dbg/ldsdma_offset_probe.hip), so ONE M0 write
  // serves all pieces of a stage, and M0 needs no save / restore: gfx9 DS instructions do not read it and the compiler has no other
  // use for it in these kernels (tools/kernel_resources.py checks that nothing else in them touches M0).  Round 5: 6 instructions
  // per stage where the per-piece form (opaque zero + 64-bit adds + M0 save / restore) had 26 -- the chains are ISSUE-bound.
  // Inline asm on purpose: hipcc marks the LDS-DMA builtin as a FLAT operation that may touch both memories and from then on turns
  // EVERY vmcnt / lgkmcnt wait into a full drain (0) while one is pending.  An asm statement is invisible to its bookkeeping; its
  // own counted waits only get a little stricter.
  template <int NPC, int DST>
  __device__ __forceinline__ void dma_run(const unsigned char* src, unsigned wl) const {
    static_assert(NPC == 1 || NPC == 2 || NPC == 4, "pieces per wave and stage");
    if constexpr (NPC == 1)
      asm volatile("s_add_i32 m0, %1, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %3" : : "v"(lane16), "s"(wl), "n"(DST), "s"(src) : "memory");
    else if constexpr (NPC == 2)
      asm volatile("s_add_i32 m0, %1, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %3\n\tglobal_load_lds_dwordx4 %0, %3 offset:1024"
                   : : "v"(lane16), "s"(wl), "n"(DST), "s"(src) : "memory");
    else
      asm volatile("s_add_i32 m0, %1, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %3\n\tglobal_load_lds_dwordx4 %0, %3 offset:1024\n\t"
                   "global_load_lds_dwordx4 %0, %3 offset:2048\n\tglobal_load_lds_dwordx4 %0, %3 offset:3072"
                   : : "v"(lane16), "s"(wl), "n"(DST), "s"(src) : "memory");
  }
  // this wave's slice of a ring stage as an LDS byte address (SGPR): `off` = its offset inside the stage
  __device__ __forceinline__ unsigned lds_addr(int off) const {
    return (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(size_t)(lds_u8*)lds + (unsigned)off));
  }
  __device__ __forceinline__ Frag rd_at(int abs) const { return *reinterpret_cast<lds_frag*>(rbase[abs >> 16] + (abs & 0xffff)); }
  __device__ __forceinline__ Frag rd_frag(int slot, int ks, int ring_off = RR_OFF_RING) const {
    const int abs = ring_off + slot * RR_STAGE + ks * 1024, r = abs >> 16;
    return *reinterpret_cast<lds_frag*>(rbase[r] + (abs & 0xffff));
  }
};

// four lane masks (SGPR pairs, fresh from v_cmp) -> 32 contiguous bytes.  Scalar stores go through the scalar
// data cache: the kernel ends with s_dcache_wb.  hipcc neither counts nor pads them (s_nop: VALU-written SGPR
// read by SMEM).
__device__ __forceinline__ void mask_store4(u64* p, u64 m0, u64 m1, u64 m2, u64 m3) {
#ifndef RR_MASK_X2
  // two 16-byte scalar stores instead of four 8-byte ones (the compiler puts the four masks into two aligned SGPR quads; -DRR_MASK_X2: the old form)
  const u32x4 qa = {(unsigned)m0, (unsigned)(m0 >> 32), (unsigned)m1, (unsigned)(m1 >> 32)};
  const u32x4 qb = {(unsigned)m2, (unsigned)(m2 >> 32), (unsigned)m3, (unsigned)(m3 >> 32)};
  asm volatile("s_nop 4\n\ts_store_dwordx4 %0, %2, 0x0\n\ts_store_dwordx4 %1, %2, 0x10" : : "s"(qa), "s"(qb), "s"(p) : "memory");
  return;
#endif
  asm volatile("s_nop 4\n\ts_store_dwordx2 %0, %4, 0x0\n\ts_store_dwordx2 %1, %4, 0x8\n\ts_store_dwordx2 %2, %4, 0x10\n\t"
               "s_store_dwordx2 %3, %4, 0x18"
               :
               : "s"(m0), "s"(m1), "s"(m2), "s"(m3), "s"(p)
               : "memory");
}

// ... the same with the byte offset as an IMMEDIATE (one base pointer per tile and layer instead of a 64-bit add per store) and
// WITHOUT the hazard padding: the caller issues it at least one MFMA gap (>= 5 instructions of this wave) behind the v_cmp that
// wrote the masks.
template <int OFF, bool PAD = false>
__device__ __forceinline__ void mask_store4_at(const u64* base, unsigned soff, u64 m0, u64 m1, u64 m2, u64 m3) {
  // address = layer base (a kernel argument: already in SGPRs) + ONE 32-bit wave-tile offset shared by the four layers + immediate
  const u32x4 qa = {(unsigned)m0, (unsigned)(m0 >> 32), (unsigned)m1, (unsigned)(m1 >> 32)};
  const u32x4 qb = {(unsigned)m2, (unsigned)(m2 >> 32), (unsigned)m3, (unsigned)(m3 >> 32)};
  if constexpr (PAD)
    asm volatile("s_nop 4\n\ts_store_dwordx4 %0, %2, %3 offset:%4\n\ts_store_dwordx4 %1, %2, %3 offset:%5" : : "s"(qa), "s"(qb), "s"(base), "s"(soff), "n"(OFF), "n"(OFF + 16) : "memory");
  else
    asm volatile("s_store_dwordx4 %0, %2, %3 offset:%4\n\ts_store_dwordx4 %1, %2, %3 offset:%5" : : "s"(qa), "s"(qb), "s"(base), "s"(soff), "n"(OFF), "n"(OFF + 16) : "memory");
}

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

// ---- copy-out of the rows that only the weight-gradient kernel reads (H_0..2, dZ_1..3): ROW-BLOCKED images (rn_wgrad_blocked.hip)
//   16-bit: element (m, f) at ((m / 8) * 256 + f) * 8 + m % 8;   e4m3: byte (m, f) at ((m / 16) * 256 + f) * 16 + m % 16
// i.e. 16 bytes = 8 (16) pair rows of ONE feature = one lane's MFMA operand in the product that contracts over the rows.  The
// epilogue holds a pair row per lane, so the staged block goes back through the LDS TRANSPOSE reads: a lane receives one
// feature's column of 4 (tr_b16) / 8 (tr_b8) rows per read, two reads make its 16 bytes, and a store instruction covers 32
// (16-bit: 512 contiguous bytes per half wave) / 64 (e4m3: 1 KB) consecutive features of one row block.
typedef __attribute__((address_space(3))) s16x4* lds_tr16;
typedef __attribute__((ext_vector_type(2))) int rr_i32x2;
typedef __attribute__((address_space(3))) rr_i32x2* lds_tr8;
// 16-bit block staged as [32 rows][RR_SRS], 32 features = 64 B per row: -> co[t] = rows 8 rb .. 8 rb + 7 (rb = 2 t + lane / 32) of
// feature 16 ((lane / 16) % 2) + lane % 16
template <int SRS = RR_SRS>
__device__ __forceinline__ void co_read_blk16(const unsigned char* stg, int lane, u32x4 (&co)[2]) {
  const int g = lane >> 4, li = lane & 15;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int row = 8 * (2 * t + (g >> 1)) + 4 * u + (li >> 2);
      const s16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr16)(stg + row * SRS + 32 * (g & 1) + 8 * (li & 3)));
      const u32x2 rr = __builtin_bit_cast(u32x2, r);
      co[t][2 * u] = rr[0];
      co[t][2 * u + 1] = rr[1];
    }
}
// byte offset of this lane's 16 bytes of store t inside the wave's 32 rows of a 16-bit image (32-feature block `cob`)
__device__ __forceinline__ unsigned co_off_blk16(int lane, int cob, int t) {
  const int g = lane >> 4;
  return (unsigned)(((2 * t + (g >> 1)) * RR_G + 32 * cob + 16 * (g & 1) + (lane & 15)) * 16);
}
// e4m3 pair of blocks staged as [32 rows][RR_SRS8], 64 features = 64 B per row: -> co[t] = rows 16 t .. 16 t + 15 of feature `lane`
__device__ __forceinline__ void co_read_blk8(const unsigned char* stg, int lane, u32x4 (&co)[2]) {
  const int g = lane >> 4, li = lane & 15;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int row = 16 * t + 8 * u + (li >> 1);
      const rr_i32x2 r = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_tr8)(stg + row * RR_SRS8 + 16 * g + 8 * (li & 1)));
      co[t][2 * u] = (unsigned)r[0];
      co[t][2 * u + 1] = (unsigned)r[1];
    }
}
__device__ __forceinline__ unsigned co_off_blk8(int lane, int cob, int t) {
  return (unsigned)((t * RR_G + 32 * (cob & ~1) + lane) * 16);
}
}  // namespace

// dst[((ob * 16 + ks) * 64 + lane) * 8 + e] = src[32 ob + lane % 32][kidx], 0 beyond (R, C)
//   natural  : kidx = 16 ks + 8 h + e                         (operand read straight from memory rows)
//   permuted : kidx = 32 (ks / 2) + 4 h + 8 (2 (ks % 2) + e / 4) + e % 4   (operand = previous MFMA output)
// all fragment-major images of a step in ONE launch (7 per training step: 4 forward + 3 transposed)
namespace {
constexpr int RR_MAXPACK = 16;
struct PackMany { const float* src[RR_MAXPACK]; long sr[RR_MAXPACK], sc[RR_MAXPACK]; int R[RR_MAXPACK], C[RR_MAXPACK], natural[RR_MAXPACK]; bf16* dst[RR_MAXPACK]; };
}  // namespace
__global__ __launch_bounds__(256) void pack_frag_many_kernel(PackMany a) {
  const int i = blockIdx.y;
  const int g = blockIdx.x * 256 + threadIdx.x;
  if (a.natural[i] == 2) {                               // fp32 transpose: dst[c][r] = src[r][c]
    const int R = a.R[i], Cc = a.C[i];
    if (g < R * Cc) {
      const int c = g / R, r = g - c * R;
      reinterpret_cast<float*>(a.dst[i])[g] = a.src[i][(long)r * a.sr[i] + (long)c * a.sc[i]];
    }
    return;
  }
  const int e = g & 7, lane = (g >> 3) & 63, ks = (g >> 9) & 15, ob = g >> 13;
  const int h = lane >> 5, m = 32 * ob + (lane & 31);
  const int mode = a.natural[i];
  const int kidx = (mode & 1) ? 16 * ks + 8 * h + e : 32 * (ks >> 1) + 4 * h + 8 * (2 * (ks & 1) + (e >> 2)) + (e & 3);
  const float v = (m < a.R[i] && kidx < a.C[i]) ? a.src[i][(long)m * a.sr[i] + (long)kidx * a.sc[i]] : 0.f;
  if (mode & 12) {                                       // fp16 split halves of the f16s mode: hi = fp16(v), lo = fp16(v - hi)
    const int V = (mode >> 8) & 0xff;
    if ((mode & 4) && V > 1) {
      // TILE-DITHERED hi images (layers whose second pass is dropped): image d = RNE(v + ((d + 1/2) / V - 1/2) ulp16(v)).  The
      // mean of the V roundings is within ulp / (2 V) of v, and the pair sum (model.py:151-152) averages over the 256-row tiles,
      // each of which multiplies image (tile mod V): the weight rounding error stops being systematic over a question's pairs.
      const float av = fabsf(v);
      const int e = av >= 6.103515625e-05f ? ((__builtin_bit_cast(int, av) >> 23) - 127) : -14;     // binade (fp16 subnormals: 2^-14)
      const float ulp = __builtin_bit_cast(float, (e - 10 + 127) << 23);
      for (int d = 0; d < V; ++d) reinterpret_cast<f16*>(a.dst[i])[(long)d * RR_G * RR_G + g] = (f16)(v + (((float)d + 0.5f) / (float)V - 0.5f) * ulp);
      return;
    }
    const f16 hi = (f16)v;
    reinterpret_cast<f16*>(a.dst[i])[g] = (mode & 4) ? hi : (f16)(v - (float)hi);
  } else {
    a.dst[i][g] = (bf16)v;
  }
}

extern "C" int rn_pack_matrix_frag_many(const float* const* src, const long* sr, const long* sc, const int* R, const int* C,
                                        void* const* dst, const int* natural, int count, void* stream) {
  RN_CHECK_ARG(src && sr && sc && R && C && dst && natural && count > 0 && count <= RR_MAXPACK, "rn_pack_matrix_frag_many: bad arguments (count=%d, max %d)", count, RR_MAXPACK);
  PackMany a;
  memset(&a, 0, sizeof(a));
  for (int i = 0; i < count; ++i) {
    RN_CHECK_ARG(src[i] && dst[i] && R[i] > 0 && R[i] <= RR_G && C[i] > 0 && C[i] <= RR_G, "rn_pack_matrix_frag_many: entry %d: needs 0 < R, C <= 256", i);
    a.src[i] = src[i]; a.sr[i] = sr[i]; a.sc[i] = sc[i]; a.R[i] = R[i]; a.C[i] = C[i]; a.natural[i] = natural[i]; a.dst[i] = (bf16*)dst[i];
  }
  pack_frag_many_kernel<<<dim3(RR_G * RR_G / 256, count), 256, 0, (hipStream_t)stream>>>(a);
  RN_LAUNCH_CHECK("rn_pack_matrix_frag_many");
  return 0;
}

// ---- counted waits ------------------------------------------------------------------------------------------
// At the top of stage s the weights of stage s+1 must have landed.  vmcnt retires in order and counts every
// VMEM operation of the wave, so the wait names how many operations YOUNGER than those weight requests may
// stay in flight: the requests of the five stages in between plus whatever else those stages issue (stores,
// next-tile row loads).  The models below give a LOWER bound of that number per site -- waiting for more than
// necessary is always safe, waiting for less is a race.
template <bool SKIP0, bool RED = false>
struct BwdVm {
  // RED (pair reductions inside the kernel): a 6-slot ring -- the two slots it gives up are one exchange buffer
  static constexpr int NSLOT = RED ? 6 : RR_NSLOT, LA = NSLOT - 1, YS = LA - 2;   // YS: whole stages younger than the awaited weights
  static constexpr int ops(int sidx) {
    int k = RR_DPW + ((sidx >= 2 && (!RED || sidx < 18)) ? 2 : 0);    // weight requests + the two copy-out stores (RED: dZ of layer 0 is not stored)
    if (RED && sidx >= 12 && sidx < 20) k += 1;                       // the gate dword of layer-0 block sidx - 12
    if (RED && sidx >= 17) k += 1;                                    // the Ri partial of block sidx - 17
    return k;
  }
  static constexpr int younger(int sidx) {                            // the tile prologue (56 operations; 40 without the dZ[0] copy) is younger too
    int k = sidx < YS ? (SKIP0 ? 32 : 40) : 0;
    for (int t = sidx - YS > 0 ? sidx - YS : 0; t < sidx; ++t) k += ops(t);
    return k < 63 ? k : 63;
  }
};

#define RN_LAYER(L_, IN_, OUT_)                                                                                      \
  stage(IC<L_>{}, IC<0>{}, IN_, OUT_); stage(IC<L_>{}, IC<1>{}, IN_, OUT_); stage(IC<L_>{}, IC<2>{}, IN_, OUT_);      \
  stage(IC<L_>{}, IC<3>{}, IN_, OUT_); stage(IC<L_>{}, IC<4>{}, IN_, OUT_); stage(IC<L_>{}, IC<5>{}, IN_, OUT_);      \
  stage(IC<L_>{}, IC<6>{}, IN_, OUT_); stage(IC<L_>{}, IC<7>{}, IN_, OUT_)

// ================================================================================================ forward, f16s
// The parity-grade 16-bit arithmetic: fp16 activations in the operand registers (fp32 accumulate); layer 0 multiplies the hi AND
// the lo half of its fp16-split weights, layers 1..3 one tile-dithered image each (below) -- the systematic weight rounding error
// that keeps single-pass bf16 at ~1e-2 of the fp32 reference drops out.  A ring stage is hi | lo = 32 KB (4 slots, 3 stages ahead).
namespace {
constexpr int F_STAGE = 32 * 1024, F_NSLOT = 4, F_LA = 3, F_RDK = 2;
static_assert(F_NSLOT * F_STAGE == RR_NSLOT * RR_STAGE, "same ring bytes");
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
struct RRArgsF {
  const f16* Whi[RR_L];                                 // layer 0: one image; layers 1..3: vmask + 1 tile-dithered images, 128 KB apart
  const f16* Wlo[RR_L];                                 // (layer 0 only; the two-pass inference variant: every layer)
  int vmask;
  const float* bias[RR_L];
  bf16* out[RR_L];
  u64* mask[RR_L];
  int prio;
};
// The forward kernel runs with MODE.FP16_OVFL = 1 (rr_fp16_ovfl_on, first instruction): fp16 results that overflow are CLAMPED to
// +-65504 by the conversion itself, and -- same mode bit -- v_cvt_scalef32_pk_fp8_f16 saturates at 448 (byte 0x7e) instead of
// producing the NaN byte 0x7f (tools/dbg/ovfl_probe.hip pins both on gfx950).  The saturation that round 4 bought with three packed
// minimums per four values (65504 on the operand pair, 448 on both pairs in front of the e4m3 conversion) is therefore free:
// same bits, 16 VALU instructions less per stage -- the kernel is bound by VALU issue beside the MFMAs (forward chain alone,
// same box, alternating: 126.3 us with the float clamps of round 4 -> 119.3 with packed minimums -> this).
__device__ __forceinline__ void rr_fp16_ovfl_on() { __builtin_amdgcn_s_setreg(1 | (23 << 6), 1); }   // hwreg(HW_REG_MODE, 23, 1)
__device__ __forceinline__ unsigned relu_pack_f16(float a, float b) {     // (FP16_OVFL: saturates instead of overflowing to inf)
#if defined(RR_RELU_PACK_OLD)                               // (variant builds: the two v_min_f32 in front of the conversion)
  const f32x2 f = {fminf(a, 65504.f), fminf(b, 65504.f)};
#else
  const f32x2 f = {a, b};
#endif
  s16x2 x = __builtin_bit_cast(s16x2, __builtin_convertvector(f, f16x2));
  const s16x2 z = {0, 0};
  x = __builtin_elementwise_max(x, z);                      // ReLU on the packed pair: a negative fp16 is a negative int16
#if defined(RR_NO_FP16_OVFL) && !defined(RR_RELU_PACK_OLD)
  return rn_pk_min_u16(__builtin_bit_cast(unsigned, x), 0x7BFF);
#else
  return __builtin_bit_cast(unsigned, x);
#endif
}
// two packed NON-NEGATIVE fp16 pairs -> four e4m3 bytes; the clamp at 448 is the conversion's own under FP16_OVFL
__device__ __forceinline__ unsigned rr_fp8x4_from_f16(unsigned lo, unsigned hi) {
#if defined(RR_NO_FP16_OVFL) || defined(RR_RELU_PACK_OLD)
  return rn_fp8x4_from_f16(lo, hi);
#else
  rn_s16x2 r = {0, 0};
  r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(r, __builtin_bit_cast(rn_f16x2, lo), RN_H8_SCALE, false);
  r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(r, __builtin_bit_cast(rn_f16x2, hi), RN_H8_SCALE, true);
  return __builtin_bit_cast(unsigned, r);
#endif
}
template <int NK0, bool STORE, bool ST3, bool XG, bool ALG0 = false, int INJ = 0, bool H8 = false, bool GATE = false, bool LO = false>
struct F16Vm {
  static constexpr int PF_PER = (NK0 + 7) / 8;
  static constexpr int VC_STAGE = (RR_L - 1) * 8 + 4;                // ALG0: the stage that requests the next tile's bias row
  static constexpr int VQ_STAGE = (RR_L - 1) * 8 + 2;                // INJ: ... and the next tile's question row of layer INJ
  // Only the fragments a stage's MFMAs read are requested: layer 0 multiplies NK0 of a block's 16 fragments (hi and lo), the
  // one-pass layers the hi image only (each 1-KB request costs ~100 issue cycles beside the MFMAs: streaming whole 32-KB
  // stages measured 32 of the kernel's 173 us).
  static constexpr int nfr(int l) { return l == 0 ? NK0 : 16; }       // fragments per image of a layer-l stage
  static constexpr int nimg(int l) { return (l == 0 || LO) ? 2 : 1; }  // LO: hi + lo on every layer (the batch-invariant inference arithmetic)
  static constexpr int dpw(int l) { return nfr(l) * nimg(l) / RR_NW; }   // requests per wave for a stage of layer l
  static_assert((NK0 * 2) % RR_NW == 0, "layer-0 requests divide evenly over the waves");
  static constexpr int ops(int sidx) {
    int k = dpw(((sidx + F_LA) >> 3) & 3);                            // (stage sidx requests the weights of stage sidx + F_LA)
    if (STORE && sidx >= 2 && (((sidx - 2) >> 3) < RR_L - 1 || ST3) && !(GATE && ((sidx - 2) >> 3) == RR_L - 2))    // (GATE: H_2 leaves with the gate, below)
      k += (H8 && ((sidx - 2) >> 3) < RR_L - 1) ? (((sidx - 2) & 1) ? 2 : 0) : 2;
    if ((sidx >> 3) == RR_L - 1)
      for (int c = 0; c < PF_PER; ++c) k += ((sidx & 7) * PF_PER + c < NK0) ? 1 : 0;
    if (ALG0 && sidx == VC_STAGE) k += 1;
    if (INJ > 0 && sidx == VQ_STAGE) k += 1;
    if (GATE && (sidx >> 3) == RR_L - 1 && (sidx & 7) >= 1) k += 1;   // the H_2 | gate cells of block (sidx & 7) - 1
    return k;
  }
  static constexpr int tail() { return (STORE && ST3 ? 4 : 0) + (XG ? (ALG0 ? 1 : 8) : 0) + (GATE ? 1 : 0); }
  // weights of stage s+1 were requested in stage s+1-F_LA: younger are the stages s-(F_LA-2) .. s-1
  static constexpr int younger(int sidx, bool first) {
    int k = 0;
    for (int t = sidx - (F_LA - 2); t < sidx; ++t) k += t >= 0 ? ops(t) : (first ? 0 : ops(t + 8 * RR_L));
    if (sidx < F_LA - 2 && !first) k += tail();
    if (first && sidx <= F_LA - 2) k += dpw(0) * (F_LA - 2 - sidx) + NK0 + (ALG0 ? 1 : 0);
    return k < 63 ? k : 63;
  }
};
}  // namespace

// ALG0 -- the first layer factored through the pair structure (the only instantiated form):
//   W0 [x_j | x_i | q] + b0 = W0a x_j + (W0b x_i + W0c q + b0): the bracket is constant over a wave's 32 pair rows (same question,
//   same i) and comes in as the layer's BIAS row -- Vc[b*n + i][256], fp32, from rn_pair_tables; only the x_j part is left as an
//   MFMA, K = 64 (P = the packed fp16 object rows Xp[b*n + j][64], a 0.5 MB table that lives in L2) instead of K = 192 on a
//   100 MB pair matrix that then never exists.
// INJ > 0 -- the question injected at layer INJ (the reference's "IR" variants, model.py:131-142): the layer's input is
//   [H_{INJ-1} | q[b]], so W_INJ [H | q] + b = W_INJ[:, 0:256] H + (W_INJ[:, 256:] q[b] + b): the bracket is a per-QUESTION constant
//   -- Vq[b][256], fp32, one small product on the host side -- and takes the place of the layer's bias row in LDS for the tile (a
//   256-row tile lies inside one question: n*n % 256 == 0).  The next tile's row is fetched during the last layer.
// Passes.  Layer 0 runs hi + lo (its K = 64 / 192 product is a twelfth .. a quarter of a later layer's).  Layers 1..3 run ONE pass
// on TILE-DITHERED hi images (rn_pack_matrix_frag_many): tile t multiplies image t mod V, V = 4 roundings of the weights whose
// mean is within ulp / 8 of the fp32 weight.  What keeps single-pass 16-bit arithmetic at 3e-3 of the fp32 reference is not the
// size of the weight rounding error (2^-12) but that it is the SAME for all 4096 pairs of a question and survives the pair sum;
// dithered over the question's tiles it averages out like the activation rounding does.  Measured (CPU emulation of this
// arithmetic on the released checkpoints, 24 questions, tools/dbg/emulate_lo_sets.py; the bar is 1e-3): hi + lo on layers 0..2
// (round 2) 1.1e-4 / on all four 5e-5 (ir-fp); one pass everywhere 2.6e-3 / 6.3e-4; THIS scheme 1.7e-4 (original-fp) / 1.9e-4
// (ir-fp) -- with 448 instead of 704 / 832 MFMAs per wave and tile.
// H8 -- the H_l copies for the weight gradient leave as OCP e4m3 bytes (rn_common.h), converted from the fp16 operand pair the
//   group has just built (clamp + two v_cvt_scalef32_pk_fp8_f16); TWO blocks share a staging row (64 B) and leave together after
//   the odd one: a row must receive 64 contiguous bytes per store instruction (32-byte pieces ran the kernel at half speed).
// ABL (RN_DIAG builds only: timing ablations, WRONG results) -- 1: no bias rows (the block's first MFMA starts from zero), 2: no
// barriers, 4: no waits for the weight stream, 8: no copy-out (no staging reads, no H stores), 16: no mask stores, 32: no epilogue
// at all, 64: no weight requests
// RAG (ALG0, question at layer 0) -- object counts that are no multiple of 32 (the 14 x 14 grid: n = 196): the j axis of every
// (question, i) group is PADDED to njp = 32 ceil(n / 32) pair rows, so that a wave still lies inside one group (one Vc bias
// row) and the factored first layer applies -- no 443-MB pair matrix, K = 64 instead of 192 on layer 0, no stored H_3.  Row m of
// the padded pair space = (b, i, j) with j = m mod njp; rows with j >= n read an all-zero object row (index n_zero of Xp) and
// are INVALID: their ReLU lane-mask bits are cleared in every layer (the backward chain, the gate job and the pair reductions
// then see zero gradients for them without knowing about the padding) and they are left out of the pair sum.  A 256-row tile
// may straddle two questions at a wave boundary: it leaves TWO partial rows (rn_pair_sum_tiles adds them up per question).
// GATE: the last layer's ReLU gate leaves IN THE SIGN BITS of the e4m3 H_2 image (post-ReLU bytes are never negative): byte (m, f)
// of out[2] = e4m3(H_2[m, f]) | gate_3[m, f] << 7 -- ONE 67-MB image is both operands of the gate job of rn_g_wgrad_blocked
// (dZ_3 = gate_3 x dxg[question], never stored), instead of an H_2 image plus a {0, 1} byte image of the gate (round 3: 67 MB
// written here and read there for 8 MB of information).  H_2 therefore leaves one layer late: block pob's bytes are converted from
// the fp16 operand registers of the last layer (alive until its end) and staged during the stage that closes output block pob of
// the LAST layer, read back transposed -- lane (n, h) receives rows 16 h .. 16 h + 15 of feature n -- and or-ed with the gate cell:
// in the un-swapped last layer a lane owns feature n of rows 8 j + 4 h + r, i.e. per 16-row group two of the cell's four dwords,
// the other two sit in the partner lane (n, 1 - h) -- two v_permlane32_swap hand lane half 0 the whole cell of rows 0..15 and half
// 1 that of rows 16..31.  One 16-byte store per lane and block, all in the MFMA shadow.
// LO (inference only): hi + lo split weights on EVERY layer instead of one pass on the tile's dithered image -- no arithmetic depends
// on where a pair row sits (tile index), so a question's log-probs are the same whatever its position in the batch and whatever
// the order of its objects (up to fp32 summation order), at twice the MFMA count of layers 1..3.  What eval() runs by default.
template <int NK0, bool STORE, bool ST3, bool MASK, bool XG, bool ALG0 = false, int INJ = 0, bool H8 = false, int ABL = 0, bool RAG = false,
          bool GATE = false, bool LO = false>
__global__ __launch_bounds__(RR_NT) void g_chain_rr_f16s_kernel(const f16* __restrict__ P, int ldp, RRArgsF a,
                                                                float* __restrict__ xg_part, int ntiles,
                                                                const float* __restrict__ Vc = nullptr, int n_obj = 0,
                                                                const float* __restrict__ Vq = nullptr, int rows_per_b = 1,
                                                                int njp = 0, int n_zero = 0) {
  static_assert(!RAG || (ALG0 && INJ == 0 && XG), "padded j axis: the factored first layer with the question at layer 0");
  static_assert(INJ >= 0 && INJ < RR_L - 1, "the injected layer is one of the swapped-operand layers");
  static_assert(!H8 || STORE, "e4m3 copies of H_0..2 (a stored H_3 stays bf16: the pair sum reads it)");
  static_assert(!GATE || (H8 && MASK && !ST3), "the gate image belongs to the training output set with e4m3 copies");
  static_assert(!LO || (!STORE && !MASK), "two passes on every layer: the inference variant");
  typedef F16Vm<NK0, STORE, ST3, XG, ALG0, INJ, H8, GATE, LO> Vm;
#if !defined(RR_NO_FP16_OVFL) && !defined(RR_RELU_PACK_OLD)
  rr_fp16_ovfl_on();
#endif
  __shared__ __attribute__((aligned(16))) unsigned char lds[RR_LDS];
  RRCore k;
  k.init(lds);
  const int t = threadIdx.x, lane = k.lane, w = k.w, n = k.n, h = k.h;
  unsigned char* const stg = lds + RR_OFF_STG + w * RR_STG;
  float* const bias_s = reinterpret_cast<float*>(lds + RR_OFF_BIAS);
  const unsigned prow_off = (unsigned)(n * ldp + 8 * h) * 2u;
  auto load_row_frag = [&](long m0w, int ks) -> Frag {
    gbl_cu8* base = (gbl_cu8*)reinterpret_cast<const unsigned char*>(P + m0w * ldp);
    asm volatile("" : "+s"(base));
    return *reinterpret_cast<__attribute__((address_space(1))) const Frag*>(base + prow_off + 32 * ks);
  };
  // The requests of this wave for a stage of layer l2: pieces q = dpw w .. dpw w + dpw - 1 of the stage's nimg x nfr fragments -- image
  // q / nfr (hi, lo; wave-uniform: dpw divides nfr), fragments q % nfr ..: consecutive KBs in the image and in the ring stage.
  // wsrc(l2, tile_): the wave's first source byte of block 0; wdst[l2 != 0]: its slice of a ring stage (LDS byte address).
  static_assert(Vm::nfr(0) % Vm::dpw(0) == 0 && Vm::nfr(1) % Vm::dpw(1) == 0, "a wave's pieces lie inside one image");
  auto wsrc = [&](int l2, int tile_) -> const unsigned char* {
    const int q = Vm::dpw(l2) * w, im = q / Vm::nfr(l2), fr = q - im * Vm::nfr(l2);
    const f16* img = (l2 == 0 || LO) ? (im ? a.Wlo[l2] : a.Whi[l2]) : a.Whi[l2] + (long)(tile_ & a.vmask) * (RR_G * RR_G);   // (the tile's dithered image)
    return reinterpret_cast<const unsigned char*>(img) + fr * 1024;
  };
  auto wdst_of = [&](int l2) -> unsigned {
    const int q = Vm::dpw(l2) * w, im = q / Vm::nfr(l2), fr = q - im * Vm::nfr(l2);
    return k.lds_addr(im * RR_STAGE + fr * 1024);
  };
  const unsigned wdst[2] = {wdst_of(0), wdst_of(1)};
  const unsigned char* wsp[RR_L];                                     // per layer, for the tile at hand (layer 0: any tile)
  auto rd = [&](int slot, int ks, int p) -> Frag { return k.rd_at(RR_OFF_RING + slot * F_STAGE + p * RR_STAGE + ks * 1024); };

  auto op_row = [&](long m0w_) -> long {
    if constexpr (!ALG0) return m0w_;
    const int nn = n_obj * n_obj, b = (int)(m0w_ / nn), r = (int)(m0w_ - (long)b * nn), i = r / n_obj;
    return (long)b * n_obj + (r - i * n_obj);
  };
  auto vc_row = [&](long m0w_) -> long {
    if constexpr (RAG) return m0w_ / njp;                             // (b, i) group of the wave: b * n + i
    const int nn = n_obj * n_obj, b = (int)(m0w_ / nn), r = (int)(m0w_ - (long)b * nn);
    return (long)b * n_obj + r / n_obj;
  };
  // RAG: this lane's object row of Xp for the wave that starts at padded pair row m0w_ (j = jw + n; beyond the n objects: the zero row)
  auto rag_load = [&](long m0w_, int ks) -> Frag {
    const int bi = (int)(m0w_ / njp), jw = (int)(m0w_ - (long)bi * njp), j = jw + n;
    const int row = j < n_obj ? (bi / n_obj) * n_obj + j : n_zero;
    return *reinterpret_cast<const Frag*>(reinterpret_cast<const unsigned char*>(P) + ((long)row * ldp + 8 * h) * 2 + 32 * ks);
  };
  auto in_frag = [&](long m0w_, int ks) -> Frag {
    if constexpr (RAG) return rag_load(m0w_, ks);
    else return load_row_frag(op_row(m0w_), ks);
  };
  float* const vc_s = reinterpret_cast<float*>(lds + RR_OFF_VC) + w * RR_G;
  auto vc_load = [&](long m0w_) -> f32x4 { return *reinterpret_cast<const f32x4*>(Vc + vc_row(m0w_) * RR_G + lane * 4); };
  f32x4 vcreg = {0.f, 0.f, 0.f, 0.f};
  f32x4 vqreg = {0.f, 0.f, 0.f, 0.f};
  auto vq_load = [&](int tile_) -> f32x4 {
    return *reinterpret_cast<const f32x4*>(Vq + (long)(((long)tile_ * RR_TM) / rows_per_b) * RR_G + lane * 4);
  };

  Frag actA[16], actB[16], ring[F_RDK][2];
  f32x16 acc[2];
  u32x4 co[2];
  f32x16 cinit = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  int tile = blockIdx.x;
  if (tile >= ntiles) return;
  // Two waves share each SIMD; the hardware arbitrates their issue slots by priority, then age, and the second-dispatched
  // half loses every stage head.  One static s_setprio for that half evens the pair out (MI355X_MICROARCH.md, "Two waves
  // per SIMD", item 4).  RN_RR_PRIO=0 turns it off.
  if (a.prio && k.w >= RR_NW / 2) __builtin_amdgcn_s_setprio(1);
  wsp[0] = wsrc(0, tile);
  static_assert(F_LA == 3, "prologue: three stages of layer 0");
  k.template dma_run<Vm::dpw(0), RR_OFF_RING + 0 * F_STAGE>(wsp[0] + 0 * RR_STAGE, wdst[0]);
  k.template dma_run<Vm::dpw(0), RR_OFF_RING + 1 * F_STAGE>(wsp[0] + 1 * RR_STAGE, wdst[0]);
  k.template dma_run<Vm::dpw(0), RR_OFF_RING + 2 * F_STAGE>(wsp[0] + 2 * RR_STAGE, wdst[0]);
#pragma unroll
  for (int ks = 0; ks < NK0; ++ks) actA[ks] = in_frag((long)tile * RR_TM + RR_WR * w, ks);
  if constexpr (ALG0) *reinterpret_cast<f32x4*>(vc_s + lane * 4) = vc_load((long)tile * RR_TM + RR_WR * w);
  if (t < RR_G) {
#pragma unroll
    for (int l = 0; l < RR_L; ++l)
      bias_s[l * RR_G + t] = (INJ > 0 && l == INJ) ? Vq[(long)(((long)tile * RR_TM) / rows_per_b) * RR_G + t] : a.bias[l][t];
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();
#pragma unroll
  for (int r = 0; r < F_RDK; ++r)
#pragma unroll
    for (int p = 0; p < 2; ++p) ring[r][p] = rd(0, r, p);
  bool first = true;

  for (; tile < ntiles; tile += gridDim.x) {
    const long m0w = (long)tile * RR_TM + RR_WR * w;
    const long wt = (long)tile * RR_NW + w;
    const int tnext = tile + (int)gridDim.x < ntiles ? tile + (int)gridDim.x : tile;
    const long m0n = (long)tnext * RR_TM + RR_WR * w;
    // the wave's weight sources of this tile (layers 1..3: the tile's dithered image; layer 0 -- also what the last F_LA stages
    // request for the NEXT tile -- has one image pair).  Opaque per tile: `pointer + constant` is then formed at the use (two SALU
    // operations) instead of being hoisted out of the tile loop into 56 live address pairs.
#pragma unroll
    for (int l2 = 0; l2 < RR_L; ++l2) {
      wsp[l2] = wsrc(l2, tile);
      asm volatile("" : "+s"(wsp[l2]));
    }
    float xs[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) xs[i] = 0.f;
    // RAG: valid rows of this wave = the first nv (a multiple of 4; 32 for all but a group's last wave).  Lane masks: swapped
    // layers 0..2 have lane = row (both lane halves), the last layer accumulator group j' of lane half hh = rows 8 j' + 4 hh + r
    u64 vm012 = ~0ull, vm3[4] = {~0ull, ~0ull, ~0ull, ~0ull};
    float keep3[4] = {1.f, 1.f, 1.f, 1.f};
    if constexpr (RAG) {
      const int jw = (int)(m0w % njp), nv = (n_obj - jw) < RR_WR ? (n_obj - jw) : RR_WR;
      const u64 lo = nv >= 32 ? 0xffffffffull : ((1ull << nv) - 1ull);
      vm012 = lo | (lo << 32);
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        vm3[jj] = (8 * jj < nv ? 0x00000000ffffffffull : 0ull) | (8 * jj + 4 < nv ? 0xffffffff00000000ull : 0ull);
        keep3[jj] = (8 * jj + 4 * h < nv) ? 1.f : 0.f;
      }
    }

    auto bias_read = [&](int l, int ob) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float* src = (ALG0 && l == 0) ? vc_s : bias_s + l * RR_G;
        const f32x4 b = *reinterpret_cast<const f32x4*>(src + 32 * ob + 8 * j + 4 * h);
#pragma unroll
        for (int r = 0; r < 4; ++r) cinit[4 * j + r] = b[r];
      }
    };
    // lane masks of a group: compared in one MFMA gap (mask_cmp), stored in the next (mask_put) -- the VALU -> SMEM hazard of the
    // compare results is then covered by the instructions in between (no s_nop), and a store is base + IMMEDIATE (the wave-tile's
    // 1-KB record of layer pl starts at mbase[pl])
    unsigned moff = (unsigned)(wt * 1024);                            // byte offset of the wave-tile's 1-KB record in every layer's mask buffer
    asm volatile("" : "+s"(moff));
    u64 mk[4];
    auto mask_cmp = [&](int pl, int j, float x0, float x1, float x2, float x3) {
      if constexpr (MASK && !(ABL & 16)) {
        const u64 vm = !RAG ? ~0ull : (pl == RR_L - 1 ? vm3[j] : vm012);
        mk[0] = __ballot(x0 > 0.f) & vm; mk[1] = __ballot(x1 > 0.f) & vm; mk[2] = __ballot(x2 > 0.f) & vm; mk[3] = __ballot(x3 > 0.f) & vm;
      }
    };
    auto mask_put = [&](auto plc, auto pobc, int j, auto padc) {     // (j: an unrolled loop's constant -- the switch folds)
      if constexpr (MASK && !(ABL & 16)) {
        constexpr int pl = decltype(plc)::value < 0 ? 0 : decltype(plc)::value, pob = decltype(pobc)::value;   // (-1: the never-executed instance of stage 0)
        constexpr bool pad = decltype(padc)::value != 0;
        switch (j) {
          case 0: mask_store4_at<(pob * 16 + 0) * 8, pad>(a.mask[pl], moff, mk[0], mk[1], mk[2], mk[3]); break;
          case 1: mask_store4_at<(pob * 16 + 4) * 8, pad>(a.mask[pl], moff, mk[0], mk[1], mk[2], mk[3]); break;
          case 2: mask_store4_at<(pob * 16 + 8) * 8, pad>(a.mask[pl], moff, mk[0], mk[1], mk[2], mk[3]); break;
          default: mask_store4_at<(pob * 16 + 12) * 8, pad>(a.mask[pl], moff, mk[0], mk[1], mk[2], mk[3]); break;
        }
      }
    };
    // phases of group j: 0 masks, 1 fp16 operand of the next layer, 2 bf16 copy for HBM, 3 staging write
    auto epi_group = [&](auto plc, auto pobc, int j, int ph, Frag* dst, u32x2 (&pk)[4]) {
      constexpr int pl = decltype(plc)::value, pob = decltype(pobc)::value;
      const f32x16& c = acc[pob & 1];
      if (ph == 0) {
        mask_cmp(pl, j, c[4 * j], c[4 * j + 1], c[4 * j + 2], c[4 * j + 3]);
        mask_put(plc, pobc, j, IC<1>{});                               // (right behind the compare: padded -- keeping the masks in SGPRs
      }                                                                //  over an MFMA gap spills: the kernel sits at the 102-SGPR limit)
      if (ph == 1 && dst) {
        const unsigned f0 = relu_pack_f16(c[4 * j + 0], c[4 * j + 1]), f1 = relu_pack_f16(c[4 * j + 2], c[4 * j + 3]);
        dst[2 * pob + (j >> 1)][(j & 1) * 2 + 0] = f0;
        dst[2 * pob + (j >> 1)][(j & 1) * 2 + 1] = f1;
        if constexpr (H8) { pk[j][0] = f0; pk[j][1] = f1; }
      }
      if (GATE && pl == RR_L - 2) return;                             // (H_2 is staged by the last layer, with its gate)
      if (ph == 2 && STORE) {
        if constexpr (H8) {
          pk[j][0] = rr_fp8x4_from_f16(pk[j][0], pk[j][1]);
        } else {
          pk[j][0] = relu_pack_bf16(c[4 * j + 0], c[4 * j + 1]);
          pk[j][1] = relu_pack_bf16(c[4 * j + 2], c[4 * j + 3]);
        }
      }
      if (ph == 3 && STORE) {
        if constexpr (H8) *reinterpret_cast<unsigned*>(stg + n * RR_SRS8 + 32 * (pob & 1) + 8 * j + 4 * h) = pk[j][0];
        else *reinterpret_cast<u32x2*>(stg + n * RR_SRS + 16 * j + 8 * h) = pk[j];
      }
    };
    // cl < RR_L - 1: the row-blocked image for the weight gradient (transposing read-back); the last layer's rows (a stored
    // H_3: the pair sum reads it) stay row-major
    auto co_read = [&](int cl) {
      if (cl < RR_L - 1) {
        if constexpr (H8) co_read_blk8(stg, lane, co);
        else co_read_blk16(stg, lane, co);
        return;
      }
#pragma unroll
      for (int q = 0; q < 2; ++q) co[q] = *reinterpret_cast<const u32x4*>(stg + (16 * q + (lane >> 2)) * RR_SRS + (lane & 3) * 16);
    };
    const unsigned orow_off = (unsigned)((lane >> 2) * RR_G + (lane & 3) * 8) * 2u;
    auto co_store = [&](int cl, int cob, int q) {
      if (cl < RR_L - 1) {
        // e4m3: blocks cob - 1, cob (one byte per element); 16-bit: block cob.  m0w * 256 elements = the wave's first row block
        gbl_u8* baseb = (gbl_u8*)(reinterpret_cast<unsigned char*>(a.out[cl]) + m0w * RR_G * (H8 ? 1 : 2));
        asm volatile("" : "+s"(baseb));
        const unsigned off = H8 ? co_off_blk8(lane, cob, q) : co_off_blk16(lane, cob, q);
        RR_FWD_STORE(co[q], reinterpret_cast<__attribute__((address_space(1))) u32x4*>(baseb + off));
        return;
      }
      gbl_u8* base = (gbl_u8*)reinterpret_cast<unsigned char*>(a.out[cl] + m0w * RR_G);
      asm volatile("" : "+s"(base));
      RR_FWD_STORE(co[q], reinterpret_cast<__attribute__((address_space(1))) u32x4*>(base + orow_off + (16 * q * RR_G + 32 * cob) * 2));
    };
    float b3 = 0.f;
    unsigned gd[4];                                                    // GATE: the lane's four gate bytes of accumulator group j
    u32x4 hcell;                                                       // GATE: rows 16 h .. 16 h + 15 of feature n of H_2 block pob, e4m3
    const unsigned gate_lane = (unsigned)((h * RR_G + n) * 16);
    // H_2 block pob, accumulator group j: the lane's four features 8 j + 4 h + r of row n, from the fp16 operand pair of the last layer
    auto h2_stage = [&](int pob, int j, unsigned f0, unsigned f1) {
      if constexpr (GATE) *reinterpret_cast<unsigned*>(stg + n * RR_SRS8 + 32 * (pob & 1) + 8 * j + 4 * h) = rr_fp8x4_from_f16(f0, f1);
    };
    auto h2_read = [&](int pob) {
      if constexpr (GATE) {
        const int li = lane & 15, fg = (lane >> 4) & 1;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const rr_i32x2 r = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_tr8)(stg + (16 * h + 8 * u + (li >> 1)) * RR_SRS8 + 32 * (pob & 1) + 16 * fg + 8 * (li & 1)));
          hcell[2 * u] = (unsigned)r[0];
          hcell[2 * u + 1] = (unsigned)r[1];
        }
      }
    };
    auto gate_store = [&](int pob) {
      if constexpr (GATE) {
        // gd[0], gd[1]: dwords h, 2 + h of the cell (rows 0..15, feature n); gd[2], gd[3]: of the cell of rows 16..31
        const u32x2 s0 = __builtin_amdgcn_permlane32_swap(gd[0], gd[2], false, false);
        const u32x2 s1 = __builtin_amdgcn_permlane32_swap(gd[1], gd[3], false, false);
        const u32x4 cell = {s0[0] | hcell[0], s0[1] | hcell[1], s1[0] | hcell[2], s1[1] | hcell[3]};
        gbl_u8* base = (gbl_u8*)(reinterpret_cast<unsigned char*>(a.out[RR_L - 2]) + m0w * RR_G);
        asm volatile("" : "+s"(base));
        RR_FWD_STORE(cell, reinterpret_cast<__attribute__((address_space(1))) u32x4*>(base + gate_lane + 512 * pob));
      }
    };
    auto epi3_group = [&](auto pobc, int j, int ph, f32x4 (&v)[4], auto padc) {
      constexpr int pob = decltype(pobc)::value;
      if (ph == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[j][r] = fmaxf(acc[pob & 1][4 * j + r] + b3, 0.f);
      }
      if (ph == 3) {
        if constexpr (GATE) {
          unsigned g = (v[j][0] > 0.f ? 0x80u : 0u) | (v[j][1] > 0.f ? 0x8000u : 0u) | (v[j][2] > 0.f ? 0x800000u : 0u) | (v[j][3] > 0.f ? 0x80000000u : 0u);
          if constexpr (RAG) g = keep3[j] != 0.f ? g : 0u;              // (padded rows: gate 0, like their mask bits)
          gd[j] = g;
        }
      }
      if (ph == 1) {
        if constexpr (XG && RAG) xs[pob] += keep3[j] * ((v[j][0] + v[j][1]) + (v[j][2] + v[j][3]));
        else if constexpr (XG) xs[pob] += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
        mask_cmp(RR_L - 1, j, v[j][0], v[j][1], v[j][2], v[j][3]);
        mask_put(IC<RR_L - 1>{}, pobc, j, IC<1>{});
      }
      if (ph == 2) {
        if constexpr (STORE && ST3) {
#pragma unroll
          for (int r = 0; r < 4; ++r) *reinterpret_cast<bf16*>(stg + (8 * j + 4 * h + r) * RR_SRS + n * 2) = (bf16)v[j][r];
        }
      }
    };
    auto stage = [&](auto lc, auto obc, Frag (&in)[16], Frag (&out)[16]) {
      constexpr int l = decltype(lc)::value, ob = decltype(obc)::value;
      constexpr int NK = (l == 0) ? NK0 : 16;
      constexpr int NP = (l == 0 || LO) ? 2 : 1;                       // passes of this stage: hi (+ lo)
      constexpr int nl = (l * 8 + ob + 1 == 8 * RR_L) ? 0 : (l * 8 + ob + 1) >> 3;   // layer of the next stage
      constexpr int NPn = (nl == 0 || LO) ? 2 : 1;
      constexpr int CPG = NP * NK / 4;                                 // MFMA gaps per epilogue group (8 / 6 / 4)
      constexpr int sidx = l * 8 + ob;
      constexpr bool has_prev = sidx > 0;
      constexpr int pl = ob ? l : l - 1, pob = ob ? ob - 1 : 7;
      constexpr int cl = (sidx - 2) >> 3, cob = (sidx - 2) & 7;
      constexpr bool has_co = !(ABL & 8) && STORE && sidx >= 2 && (cl < RR_L - 1 || ST3) && (!H8 || cl == RR_L - 1 || (cob & 1)) && !(GATE && cl == RR_L - 2);
      constexpr int didx = sidx + F_LA;
      constexpr int dl = (didx >> 3) & 3, dob = didx & 7;
      constexpr int slot = ob & 3, nslot = (ob + 1) & 3, dslot = (ob + F_LA) & 3;
      if (l < RR_L - 1 && !(ABL & 1)) bias_read(l, ob);
      if (has_prev && pl == RR_L - 1) b3 = bias_s[pl * RR_G + 32 * pob + n];
      if (ABL & 4) {
      } else if (Vm::younger(sidx, true) != Vm::younger(sidx, false)) {
        if (first) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(Vm::younger(sidx, true)) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(Vm::younger(sidx, false)) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(Vm::younger(sidx, false)) : "memory");
      }
      if (!(ABL & 2)) __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (has_co) co_read(cl);
      __builtin_amdgcn_sched_barrier(0);
      Frag* dst = nullptr;
      if (has_prev && pl < RR_L - 1) dst = ob ? out : in;
      f32x4 v[4];
      u32x2 pk[4];
#pragma unroll
      for (int ks = 0; ks < NK; ++ks) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          const int c = NP * ks + p;
          const f16x8 fw = __builtin_bit_cast(f16x8, ring[ks % F_RDK][p]), fx = __builtin_bit_cast(f16x8, in[ks]);
          const f16x8 fa = (l == RR_L - 1) ? fx : fw, fb = (l == RR_L - 1) ? fw : fx;
          if (c == 0 && l < RR_L - 1) {
            acc[ob & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, cinit, 0, 0, 0);
          } else if (c == 0) {
            const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            acc[ob & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, z, 0, 0, 0);
          } else {
            acc[ob & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc[ob & 1], 0, 0, 0);
          }
          {
            const int f = ks + F_RDK;
            if (f < NK) ring[ks % F_RDK][p] = rd(slot, f, p);
            else if (p < NPn) ring[ks % F_RDK][p] = rd(nslot, f - NK, p);
            // a one-pass stage in front of a two-pass one (the next tile's first layer): its lo read-ahead rides here
            if (NP == 1 && NPn == 2 && f >= NK) ring[ks % F_RDK][1] = rd(nslot, f - NK, 1);
          }
          // (stage didx >= 8 L belongs to the next tile: layer 0, whose images do not depend on the tile)
          static_assert(F_LA < 8, "the look-ahead reaches no further than the next tile's first layer");
          if (c == 1 && !(ABL & 64)) k.template dma_run<Vm::dpw(dl), RR_OFF_RING + dslot * F_STAGE>(wsp[dl] + dob * RR_STAGE, wdst[dl != 0]);
          if (has_prev && !(ABL & 32)) {
            const int j = c / CPG, ph = c % CPG;
            if (pl == RR_L - 1) {
              if (ph < 4) epi3_group(IC<pob>{}, j, ph, v, IC<0>{});
              if (GATE && c < 4) {                                      // (H_2 block pob: complete since the last layer began)
                const Frag& f = in[2 * pob + ((c & 3) >> 1)];
                h2_stage(pob, c & 3, f[(c & 1) * 2], f[(c & 1) * 2 + 1]);
              }
              if (c == 6) h2_read(pob);
              if (j == 3 && ph == 3) gate_store(pob);
            } else if (CPG >= 4) {
              if (ph < 4) epi_group(IC<pl>{}, IC<pob>{}, j, ph, dst, pk);
            } else {                                                  // NK = 4: two gaps per group, two phases per gap
              epi_group(IC<pl>{}, IC<pob>{}, j, 2 * ph, dst, pk);
              epi_group(IC<pl>{}, IC<pob>{}, j, 2 * ph + 1, dst, pk);
            }
          }
          constexpr int CO2 = NP * NK > 8 ? 8 : NP * NK - 1;
          if (has_co && c == 4) co_store(cl, cob, 0);
          if (has_co && c == CO2) co_store(cl, cob, 1);
          if (l == RR_L - 1 && (c & 1) == 0 && (c >> 1) < Vm::PF_PER) {
            const int i = ob * Vm::PF_PER + (c >> 1);
            if (i < NK0) out[i] = in_frag(m0n, i);
          }
          if (ALG0 && sidx == Vm::VC_STAGE && c == 1) vcreg = vc_load(m0n);
          if (INJ > 0 && sidx == Vm::VQ_STAGE && c == 1) vqreg = vq_load(tnext);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    };
    RN_LAYER(0, actA, actB);
    RN_LAYER(1, actB, actA);
    RN_LAYER(2, actA, actB);
    RN_LAYER(3, actB, actA);
    first = false;
    {
      f32x4 v[4];
      if constexpr (STORE && ST3) {
        co_read(RR_L - 1);
#pragma unroll
        for (int q = 0; q < 2; ++q) co_store(RR_L - 1, 6, q);
      }
      b3 = bias_s[(RR_L - 1) * RR_G + 32 * 7 + n];
      if constexpr (GATE) {                                             // (constant indices: a loop here keeps the operand arrays out of registers)
        h2_stage(7, 0, actB[14][0], actB[14][1]);
        h2_stage(7, 1, actB[14][2], actB[14][3]);
        h2_stage(7, 2, actB[15][0], actB[15][1]);
        h2_stage(7, 3, actB[15][2], actB[15][3]);
      }
      h2_read(7);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        epi3_group(IC<7>{}, j, 0, v, IC<1>{});                           // (straight-line tail: the store right behind the compare -> padded)
        epi3_group(IC<7>{}, j, 1, v, IC<1>{});
        epi3_group(IC<7>{}, j, 2, v, IC<1>{});
        epi3_group(IC<7>{}, j, 3, v, IC<1>{});
      }
      gate_store(7);
      if constexpr (STORE && ST3) {
        co_read(RR_L - 1);
#pragma unroll
        for (int q = 0; q < 2; ++q) co_store(RR_L - 1, 7, q);
      }
      if constexpr (XG && ALG0) {
        // the factored-first-layer paths (a tile lies inside one question): the eight waves' partial rows meet in the staging
        // area, idle since the last copy-out, and ONE row per tile leaves -- (M / 256, 256) instead of (M / 32, 256) for the
        // reduction launch behind this kernel.  Wave w adds feature block w in wave order (deterministic).
        float* const xs_s = reinterpret_cast<float*>(lds + RR_OFF_STG);
        if constexpr (GATE) {                                           // (every wave has read its staged H_2 block back)
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          asm volatile("" ::: "memory");
        }
#pragma unroll
        for (int ob = 0; ob < 8; ++ob) {
          const float tot = xs[ob] + __shfl_xor(xs[ob], 32);          // the two 16-row halves of this wave's 32 rows
          if (h == 0) xs_s[w * RR_G + 32 * ob + n] = tot;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if constexpr (RAG) {
          // the tile's waves belong to at most two questions (a question = n_obj groups of njp / 32 waves): one partial row each
          if (h == 0) {
            const long rpq = (long)n_obj * njp;
            const long q0 = ((long)tile * RR_TM) / rpq;
            const int bw = (int)((((q0 + 1) * rpq - (long)tile * RR_TM) + RR_WR - 1) / RR_WR);      // waves of the first question (>= 1)
            float v0 = 0.f, v1 = 0.f;
#pragma unroll
            for (int q = 0; q < RR_NW; ++q) {
              const float x = xs_s[q * RR_G + 32 * w + n];
              if (q < bw) v0 += x;
              else v1 += x;
            }
            xg_part[((long)tile * 2) * RR_G + 32 * w + n] = v0;
            xg_part[((long)tile * 2 + 1) * RR_G + 32 * w + n] = v1;
          }
        } else if (h == 0) {
          float v = xs_s[32 * w + n];
#pragma unroll
          for (int q = 1; q < RR_NW; ++q) v += xs_s[q * RR_G + 32 * w + n];
          xg_part[(long)tile * RR_G + 32 * w + n] = v;
        }
      } else if constexpr (XG) {
#pragma unroll
        for (int ob = 0; ob < 8; ++ob) {
          const float tot = xs[ob] + __shfl_xor(xs[ob], 32);          // the two 16-row halves of this wave's 32 rows
          if (h == 0) xg_part[((long)tile * RR_NW + w) * RR_G + 32 * ob + n] = tot;
        }
      }
      if constexpr (ALG0) *reinterpret_cast<f32x4*>(vc_s + lane * 4) = vcreg;
      if constexpr (INJ > 0) {
        if (w == 0) {
          *reinterpret_cast<f32x4*>(bias_s + INJ * RR_G + lane * 4) = vqreg;
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  if constexpr (MASK) asm volatile("s_dcache_wb" ::: "memory");
}

// ================================================================================================== backward
// dZ[0] = dxg[b] * (H_3 > 0);  dZ[s+1] = (dZ[s] @ W_{3-s}) * (H_{2-s} > 0), s = 0..2 -- the ReLU gates come from the
// forward kernel's lane masks (32 bytes per pair row and layer instead of a 512-byte activation row).
// SKIP0: dZ[0] (the last layer's gradient) is not copied to HBM -- the gate job of rn_g_wgrad_blocked works from the masks and
// dxg instead, its only consumer besides this kernel's own first step.
// ABL (RN_DIAG builds only): timing ablations with wrong results.
// RED: the gradient of layer 0's pre-activation never leaves the chip -- its only readers are the pair-axis reductions of the
//   expansion backward (Rj = sum over i, Ri = sum over j), and those are formed here, in fp32, from the un-rounded accumulators:
//   * a tile is 8 waves = 8 consecutive i of ONE (question, block of 32 j): wave-tile (b, i, jg) = forward wave (b*n + i) * (n/32) + jg;
//   * the last dgrad step runs with the MFMA operands UN-swapped (D[row][feature], like the forward's last layer): a lane then owns
//     one feature of 16 rows (= 16 j), so Ri is an in-lane sum (two partial rows per wave-tile -- one per lane half -- go to ri_part);
//   * the ReLU gate of layer 0 is stored as lane masks of the SWAPPED layout (lane = row): for the un-swapped layout the dword
//     that holds a feature's 32 row bits is fetched per lane (a coalesced 128-byte read per wave and block, issued two dgrad
//     steps ahead) and tested bit by bit -- two VALU operations per value instead of one v_cndmask on an SGPR mask;
//   * Rj needs the sum over the tile's 8 waves: every wave writes its gated fp32 block into a double-buffered LDS exchange area,
//     and after the next stage's barrier wave w adds up the eighth it OWNS (accumulator group w / 2, register pair w % 2) over
//     the 8 waves in wave order -- two accumulators per block, carried in registers over the tiles_per_unit consecutive tiles of
//     a unit (same question and j block, consecutive i) and written once per unit to rj_part.  Every sum has a fixed order:
//     results are bitwise reproducible.
//   LDS: the weight ring shrinks to 6 slots (5 stages ahead: 5 us of cover for an L2 fetch) and moves up by two; exchange buffer 0
//   takes the two slots' place, buffer 1 the bias / staging area, which the copy-outs of layers 2 and 1 have left two barriers before its
//   first use (and which the next tile touches two barriers after its last).
// (Round 2 also had the pair reduction of the LAST gradient, dZ of layer 0, inside this kernel -- every block through LDS in
// fp32 in the swapped layout, added over the tile's four i and each wave's 32 j: the exchange made the last step LDS-bound
// (+24 us here for -40 us in rn_pair_reduce_bwd) and the step gained 1 %: removed in round 3.)
namespace {
constexpr int RR_XBUF_BYTES = RR_NW * 4 * 2 * 512;                    // [wave][accumulator group][register pair][lane] x 8 bytes = 32 KB
constexpr int RR_OFF_XB1 = 0, RR_OFF_XB0 = RR_OFF_RING, RR_OFF_RING_RED = RR_OFF_XB0 + RR_XBUF_BYTES;   // (both buffers below 64 KB: one address register + immediates)
static_assert(RR_OFF_XB1 + RR_XBUF_BYTES <= RR_OFF_RING && RR_OFF_RING_RED + 6 * RR_STAGE <= RR_LDS, "exchange buffers + 6-slot ring");
struct RRRedArgs {
  float* rj_part;                                                     // (units, 32, 256) fp32
  float* ri_part;                                                     // (M / 16, 256) fp32: row ((b*n + i) * (njp/32) + jg) * 2 + lane half
  int n_obj, tiles_per_unit;
  int njp;                                                            // pair rows per (question, i) group: n_obj, or 32 ceil(n_obj / 32) (padded j axis)
  // The units are WALKED in the order p = (v * njp/32 + jg) * B + b (question fastest) for unit ((b * njp/32 + jg) * nu + v).  Walk
  // positions [0, units_whole) run as tiles_per_unit tiles and leave ONE Rj record (the unit's own); the units behind them run one
  // tile at a time (a balanced tail) and leave one record PER TILE: tile 0 in the unit's own record, tile t > 0 in record
  // nunits + (p - units_whole) (tpu - 1) + t - 1.  With the question fastest the tail's extra records spread evenly over the
  // questions (rn_pair_reduce_parts is one workgroup per question: the natural order made three questions carry the whole tail).
  int units_whole;
};
}  // namespace
template <int ABL, bool SKIP0 = false, bool RED = false>
__global__ __launch_bounds__(RR_NT) void g_chain_rr_bwd_kernel(RRBwdArgs a, int ntiles, RRRedArgs ra = RRRedArgs{nullptr, nullptr, 0, 1, 0, 0}) {
#ifdef RN_DIAG
  static_assert(!RED || SKIP0, "in-kernel pair reductions: the variant without a stored dZ[0]");
#else
  static_assert(!RED || (SKIP0 && ABL == 0), "in-kernel pair reductions: the product variant without a stored dZ[0]");
#endif
  typedef BwdVm<SKIP0, RED> Vm;
  __shared__ __attribute__((aligned(16))) unsigned char lds[RR_LDS];
  RRCore k;
  k.init(lds);
  const int lane = k.lane, w = k.w, n = k.n, h = k.h;
  unsigned char* const stg = lds + RR_OFF_STG + w * RR_STG;
  constexpr int NS = RR_L - 1;                                        // dgrad steps
  constexpr int NSL = Vm::NSLOT, LA = Vm::LA, RING = RED ? RR_OFF_RING_RED : RR_OFF_RING;
  // staging row stride of the 16-bit blocks (64 B of features + padding).  80 B keeps the row-major 16-byte reads of a stored dZ_0
  // aligned; without them (RED) 72 B is the better padding: the 8-byte epilogue writes of 16 lanes and the transposing reads of a
  // lane group then fall on distinct banks (stride 18 dwords: 0, 18, 4, 22, ... mod 32)
#ifndef RN_RED_SRS
#define RN_RED_SRS 72
#endif
  constexpr int SRS = RED ? RN_RED_SRS : RR_SRS;

  Frag actA[16], actB[16], ring[RR_RD];
  f32x16 acc[2];
  u32x4 co[2];
  u32x16 gA[2], gB[2];                                               // two sets of 16 lane masks (SGPRs)
  // RED state: the lane's gate dwords of layer 0 (one per block), this wave's share of the Rj sums, the exchange reads in flight
  unsigned gdw[8];
  float racc[8][2];
  f32x2 xv[4];
  float rs = 0.f;

  const int nunits = RED ? ntiles / ra.tiles_per_unit : ntiles;       // (non-RED: a unit is a tile)
  // RED: work items = the whole units, then the tiles of the remaining units one by one -- with U units on C CUs the last
  // U mod C units would keep U mod C workgroups busy for a whole unit while the others idle (14 x 14 grid: 1120 units of 5 tiles
  // on 256 CUs = 25 tile times; 1024 whole units + 480 single tiles = 22)
  const int nitems = RED ? ra.units_whole + (nunits - ra.units_whole) * ra.tiles_per_unit : ntiles;
  int item = blockIdx.x;
  if (item >= nitems) return;
  // Two waves share each SIMD; the hardware arbitrates their issue slots by priority, then age, and the second-dispatched
  // half loses every stage head.  One static s_setprio for that half evens the pair out (MI355X_MICROARCH.md, "Two waves
  // per SIMD", item 4).  RN_RR_PRIO=0 turns it off.
  if (a.prio && k.w >= RR_NW / 2) __builtin_amdgcn_s_setprio(1);
  // this wave's slice of every weight block: RR_DPW consecutive KBs, in the image and in the ring stage
  const unsigned wdst = k.lds_addr(RR_DPW * w * 1024);
  const unsigned char* wsp[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) wsp[s] = reinterpret_cast<const unsigned char*>(a.W + s * a.w_stride) + RR_DPW * w * 1024;
  static_assert(LA == 5 || LA == 7, "prologue: LA stages of the first step");
  k.template dma_run<RR_DPW, RING + 0 * RR_STAGE>(wsp[0] + 0 * RR_STAGE, wdst);
  k.template dma_run<RR_DPW, RING + 1 * RR_STAGE>(wsp[0] + 1 * RR_STAGE, wdst);
  k.template dma_run<RR_DPW, RING + 2 * RR_STAGE>(wsp[0] + 2 * RR_STAGE, wdst);
  k.template dma_run<RR_DPW, RING + 3 * RR_STAGE>(wsp[0] + 3 * RR_STAGE, wdst);
  k.template dma_run<RR_DPW, RING + 4 * RR_STAGE>(wsp[0] + 4 * RR_STAGE, wdst);
  if constexpr (LA == 7) {
    k.template dma_run<RR_DPW, RING + 5 * RR_STAGE>(wsp[0] + 5 * RR_STAGE, wdst);
    k.template dma_run<RR_DPW, RING + 6 * RR_STAGE>(wsp[0] + 6 * RR_STAGE, wdst);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
#pragma unroll
  for (int r = 0; r < RR_RD; ++r) ring[r] = k.rd_frag(0, r, RING);
  // dword of a block's (un-swapped, last-layer) mask image that holds row n: mask i = 4 (n / 8) + n % 4, half (n / 4) % 2
  const int rowsel = 2 * (4 * (n >> 3) + (n & 3)) + ((n >> 2) & 1);
  // RED: the same dword index, read the other way -- in a SWAPPED layer's mask image it holds the 32 row bits of feature n
  // (padded j axis: njp / 32 wave-tiles per (question, i) -- the rows j >= n_obj have cleared mask bits in every layer and come out
  //  as exact zeros --; n_obj need not be a multiple of 8 either: the last tile of a (question, j block) then has fewer than 8 i,
  //  and its spare waves re-do the LAST valid i -- same loads, same stores of the same values (the counted waits stay valid) -- but
  //  hand ZEROS to the Rj exchange)
  const int jgs = RED ? ra.njp / RR_WR : 1, tpbj = RED ? (ra.n_obj + RR_NW - 1) / RR_NW : 1;
  unsigned char* const xwb = lds + (w * 8) * 512 + lane * 8;          // this wave's slice of an exchange buffer
  const unsigned char* const xrb = lds + w * 512 + lane * 8;          // ... and the piece it owns of every wave's slice

  for (; item < nitems; item += gridDim.x) {
   int tile0 = item, tcount = 1;
   long rec = item;                                                   // Rj record of this item
   if constexpr (RED) {
#pragma unroll
     for (int ob = 0; ob < 8; ++ob) racc[ob][0] = racc[ob][1] = 0.f;
     const int nu = tpbj / ra.tiles_per_unit, nb = nunits / (jgs * nu);
     const RnRedItem it = rn_red_item(item, nunits, ra.tiles_per_unit, ra.units_whole, nb, jgs, nu);   // (rn_common.h: shared with the reader)
     tile0 = it.tile0; tcount = it.tcount; rec = it.rec;
   }
   for (int tu = 0; tu < tcount; ++tu) {
    const int tile = tile0 + tu;
    // (opaque per tile: a block's address = pointer + constant is then formed at its use -- two SALU operations -- instead of being
    //  hoisted out of the tile loop into 24 live address pairs)
#pragma unroll
    for (int s = 0; s < NS; ++s) asm volatile("" : "+s"(wsp[s]));
    // wave-tile of this wave = index of the forward wave whose 32 pair rows it takes over
    long wt = (long)tile * RR_NW + w;
    bool wvalid = true;
    if constexpr (RED) {
      const int bj = tile / tpbj, ig = tile - bj * tpbj, bq = bj / jgs, jg = bj - bq * jgs;
      int iw = ig * RR_NW + w;
      wvalid = iw < ra.n_obj;
      iw = wvalid ? iw : ra.n_obj - 1;
      wt = ((long)bq * ra.n_obj + iw) * jgs + jg;
    }
    const long m0w = wt * RR_WR;
    const long b = (m0w + n) / a.rows_per_b;                          // question of THIS lane's pair row (a wave may straddle two)
    // zi < NS: dZ of layers 3..1 -- read only by the weight gradient: row-blocked image (transposing read-back); zi == NS: dZ of
    // layer 0, read by the pair reduction: row-major
    auto co_read = [&](int zi) {
      if (zi < NS) {
        co_read_blk16<SRS>(stg, lane, co);
        return;
      }
#pragma unroll
      for (int q = 0; q < 2; ++q) co[q] = *reinterpret_cast<const u32x4*>(stg + (16 * q + (lane >> 2)) * SRS + (lane & 3) * 16);
    };
    const unsigned orow_off = (unsigned)((lane >> 2) * RR_G + (lane & 3) * 8) * 2u;
    auto co_store = [&](int zi, int cob, int q) {
      if (ABL & 4) return;
      gbl_u8* base = (gbl_u8*)reinterpret_cast<unsigned char*>(a.dZ + zi * a.dz_stride + ((ABL & 32) ? (long)(blockIdx.x * RR_TM + RR_WR * w) : m0w) * RR_G);
      asm volatile("" : "+s"(base));
      if (zi < NS) {
        RR_BWD_STORE(co[q], reinterpret_cast<__attribute__((address_space(1))) u32x4*>(base + co_off_blk16(lane, cob, q)));
        return;
      }
      // non-temporal: these rows are read back by ANOTHER kernel much later; allocated in L2 they only evict the
      // weight images that every workgroup re-reads for every tile (measured: 222 -> 150 us)
      if (ABL & 64) *reinterpret_cast<__attribute__((address_space(1))) u32x4*>(base + orow_off + (16 * q * RR_G + 32 * cob) * 2) = co[q];
      else RR_BWD_STORE(co[q], reinterpret_cast<__attribute__((address_space(1))) u32x4*>(base + orow_off + (16 * q * RR_G + 32 * cob) * 2));
    };
    // ---- RED: epilogue of block pob of the LAST step (accumulator D[row 8 j + 4 h + r][feature 32 pob + n]), group j, phase ph
    unsigned rb = 0;
    float gx[4];
    auto red_epi = [&](int pob, int j, int ph) {
      const f32x16& c = acc[pob & 1];
      if (ph == 0) {
        if (j == 0) rb = gdw[pob] >> (4 * h);                         // bit 8 j + r = gate of row 8 j + 4 h + r
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          // (the element is copied out first: __builtin_bit_cast straight on `c[i]` reads element 0 with hipcc 7.2; and the
          //  0 / -1 word is made opaque, else the pair v_bfe_i32 + v_and becomes v_and + v_cmp + s_nop + v_cndmask)
          const float v = c[4 * j + r];
          unsigned t = (unsigned)__builtin_amdgcn_sbfe((int)rb, 8 * j + r, 1);
          asm("" : "+v"(t));
          gx[r] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v) & t);
        }
      } else if (ph == 1) {
        const float s4 = (gx[0] + gx[1]) + (gx[2] + gx[3]);
        rs = j == 0 ? s4 : rs + s4;
      } else {
        unsigned char* xw = xwb + ((pob & 1) ? RR_OFF_XB1 : RR_OFF_XB0);
        *reinterpret_cast<f32x2*>(xw + (2 * j) * 512) = wvalid ? f32x2{gx[0], gx[1]} : f32x2{0.f, 0.f};
        *reinterpret_cast<f32x2*>(xw + (2 * j + 1) * 512) = wvalid ? f32x2{gx[2], gx[3]} : f32x2{0.f, 0.f};
        if (j == 3) {
          // Ri partials of this wave-tile: the two lane halves hold rows 8 j + 4 h + {0..3} of the same feature and leave one
          // partial row EACH (a full-wave 256-byte store; v_permlane32_swap on two copies of one value is mis-compiled by hipcc 7.2)
          ra.ri_part[(wt * 2 + h) * RR_G + 32 * pob + n] = rs;
        }
      }
    };
    auto x_read = [&](int blk, int src) {
      xv[src & 3] = *reinterpret_cast<const f32x2*>(xrb + ((blk & 1) ? RR_OFF_XB1 : RR_OFF_XB0) + src * (8 * 512));
    };
    // ---- prologue: dZ[0] in operand layout (natural feature order) + its copy to HBM
    if (ABL & 1) {
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) actA[ks] = Frag{(unsigned)tile, 0u, 1u, 0u};
    } else if ((ABL & 128) && item != (int)blockIdx.x) {              // (timing only: the prologue runs for the workgroup's first tile only)
    } else {
      const float* dxb = a.dxg + b * RR_G + 8 * h;
      const unsigned* m3 = reinterpret_cast<const unsigned*>(a.mask + (RR_L - 1) * a.mask_stride) + wt * 8 * 32 + rowsel;
#pragma unroll
      for (int ob = 0; ob < 8; ++ob) {
        const unsigned bits = m3[ob * 32] >> (8 * h);                 // features 32 ob + 8 h + {0..7} and + 16 of row n
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const int ks = 2 * ob + s;
          const f32x4 d0 = *reinterpret_cast<const f32x4*>(dxb + 16 * ks), d1 = *reinterpret_cast<const f32x4*>(dxb + 16 * ks + 4);
          const float d[8] = {d0[0], d0[1], d0[2], d0[3], d1[0], d1[1], d1[2], d1[3]};
#pragma unroll
          for (int p = 0; p < 4; ++p) {
            const f32x2 f = {d[2 * p], d[2 * p + 1]};
            const unsigned u = __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2));
            const unsigned t0 = (unsigned)__builtin_amdgcn_sbfe((int)bits, 16 * s + 2 * p, 1);
            const unsigned t1 = (unsigned)__builtin_amdgcn_sbfe((int)bits, 16 * s + 2 * p + 1, 1);
            actA[ks][p] = u & ((t0 & 0xffffu) | (t1 & 0xffff0000u));
          }
          if constexpr (!SKIP0) *reinterpret_cast<u32x4*>(stg + n * SRS + 32 * s + 16 * h) = actA[ks];
        }
        if constexpr (!SKIP0) {
          co_read(0);
#pragma unroll
          for (int q = 0; q < 2; ++q) co_store(0, ob, q);
        }
      }
    }
    // ---- one stage = one 32-feature block of one dgrad step ---------------------------------------------------
    auto stage = [&](auto sc, auto obc, Frag (&in)[16], Frag (&out)[16]) {
      constexpr int s = decltype(sc)::value, ob = decltype(obc)::value;
      constexpr int sidx = s * 8 + ob;
      constexpr bool has_prev = sidx > 0;
      constexpr int ps = ob ? s : s - 1, pob = ob ? ob - 1 : 7;       // block whose epilogue runs here
      constexpr int cs = (sidx - 2) >> 3, cob = (sidx - 2) & 7;       // block copied out here
      constexpr bool has_co = sidx >= 2 && !(RED && cs == NS - 1);    // (RED: the last step's blocks leave through the exchange)
      constexpr bool has_x = RED && sidx >= 2 && cs == NS - 1;        // the exchange of block cob is complete behind this stage's barrier
      constexpr int didx = sidx + LA;
      constexpr int dl = (didx >> 3) % NS, dob = didx & 7;
      constexpr int slot = sidx % NSL, nslot = (sidx + 1) % NSL, dslot = didx % NSL;
      constexpr bool un_swapped = RED && s == NS - 1;
      constexpr bool gate_this = !(RED && s == NS - 1), gate_prev = has_prev && !(RED && ps == NS - 1);   // gates that come as SGPR lane masks
      // gate of THIS block (layer 2 - s): requested now, used by the epilogue that runs in the next stage; the
      // gate of the previous block, requested one stage ago, must have arrived
      u32x16(&gn)[2] = (sidx & 1) ? gB : gA;
      u32x16(&gp)[2] = (sidx & 1) ? gA : gB;
      if (gate_prev && !(ABL & 2)) mask_wait(gp[0], gp[1]);
      if (gate_this && !(ABL & 2)) mask_load(a.mask + (NS - 1 - s) * a.mask_stride + (wt * 8 + ob) * 16, gn[0], gn[1]);
      // (the exchange buffers are read by OTHER waves behind this barrier: the writes must have left the LDS queue)
      if constexpr (has_x) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (ABL & 16) {                                                 // timing only: 16 more operations may stay in flight (a race)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(Vm::younger(sidx) + 16 < 63 ? Vm::younger(sidx) + 16 : 63) : "memory");
        __builtin_amdgcn_s_barrier();
      } else if (!(ABL & 8)) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(Vm::younger(sidx)) : "memory");
        __builtin_amdgcn_s_barrier();
      }
      asm volatile("" ::: "memory");
      if (has_co) co_read(cs + 1);
      if constexpr (has_x) {
#pragma unroll
        for (int src = 0; src < 4; ++src) x_read(cob, src);
      }
      if constexpr (RED && sidx >= 12 && sidx < 20) gdw[sidx - 12] = reinterpret_cast<const unsigned*>(a.mask)[(wt * 8 + sidx - 12) * 32 + rowsel];
      __builtin_amdgcn_sched_barrier(0);
      Frag* dst = nullptr;
      if (has_prev && ps < NS - 1) dst = ob ? out : in;
      float x[4][4];
      u32x2 pk[4];
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        const int c = ks;
        const bf16x8 fw = __builtin_bit_cast(bf16x8, ring[ks % RR_RD]), fx = __builtin_bit_cast(bf16x8, in[ks]);
        const bf16x8 fa = un_swapped ? fx : fw, fb = un_swapped ? fw : fx;
        if (ks == 0) {
          const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          acc[ob & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, z, 0, 0, 0);
        } else {
          acc[ob & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[ob & 1], 0, 0, 0);
        }
        {
          const int f = ks + RR_RD;
          if (f < 16) ring[ks % RR_RD] = k.rd_frag(slot, f, RING);
          else ring[ks % RR_RD] = k.rd_frag(nslot, f - 16, RING);
        }
        if (c == 1) k.template dma_run<RR_DPW, RING + dslot * RR_STAGE>(wsp[dl] + dob * RR_STAGE, wdst);
        if (has_prev && c >= 2 && c < 14) {                           // epilogue of the previous block, 3 gaps per group
          const int j = (c - 2) / 3, ph = (c - 2) % 3;
          if constexpr (RED && ps == NS - 1) {
            red_epi(pob, j, ph);
          } else if (ph == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) x[j][r] = ((ABL & 2) || gate_bit(gp[0], gp[1], 4 * j + r)) ? acc[pob & 1][4 * j + r] : 0.f;
          } else if (ph == 1) {
            const f32x2 f0 = {x[j][0], x[j][1]}, f1 = {x[j][2], x[j][3]};
            pk[j][0] = __builtin_bit_cast(unsigned, __builtin_convertvector(f0, bf16x2));
            pk[j][1] = __builtin_bit_cast(unsigned, __builtin_convertvector(f1, bf16x2));
          } else {
            *reinterpret_cast<u32x2*>(stg + n * SRS + 16 * j + 8 * h) = pk[j];
            if (dst) {
              dst[2 * pob + (j >> 1)][(j & 1) * 2 + 0] = pk[j][0];
              dst[2 * pob + (j >> 1)][(j & 1) * 2 + 1] = pk[j][1];
            }
          }
        }
        if (has_co && (c == 4 || c == 8)) {
          co_store(cs + 1, cob, (c >> 2) - 1);
        }
        if constexpr (has_x) {                                        // this wave's share of the Rj sums, the 8 waves in wave order
          if (c >= 4 && c < 12) {
            racc[cob][0] += xv[(c - 4) & 3][0];
            racc[cob][1] += xv[(c - 4) & 3][1];
            if (c < 8) x_read(cob, c);                                // (sources 4..7 take the registers of 0..3 as those are added)
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    RN_LAYER(0, actA, actB);
    RN_LAYER(1, actB, actA);
    RN_LAYER(2, actA, actB);
    if constexpr (RED) {
      // ---- tail: block (2, 7)'s epilogue; then the exchanges of blocks (2, 6) [written during the last stage] and (2, 7).
      // Buffer 1 still holds block 5, read at the top of the last stage: nobody may overwrite it before everybody is here.
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int ph = 0; ph < 3; ++ph) red_epi(7, j, ph);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
#pragma unroll
      for (int blk = 6; blk < 8; ++blk)
#pragma unroll
        for (int s4 = 0; s4 < RR_NW; s4 += 4) {
#pragma unroll
          for (int src = s4; src < s4 + 4; ++src) x_read(blk, src);
#pragma unroll
          for (int src = s4; src < s4 + 4; ++src) {
            racc[blk][0] += xv[src & 3][0];
            racc[blk][1] += xv[src & 3][1];
          }
        }
    } else {
    // ---- tail: blocks (2, 6) and (2, 7)
      co_read(NS);
#pragma unroll
      for (int q = 0; q < 2; ++q) co_store(NS, 6, q);
      u32x16(&gp)[2] = ((NS * 8 - 1) & 1) ? gB : gA;                    // requested in the last stage
      if (!(ABL & 2)) mask_wait(gp[0], gp[1]);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float x[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) x[r] = ((ABL & 2) || gate_bit(gp[0], gp[1], 4 * j + r)) ? acc[1][4 * j + r] : 0.f;
        const f32x2 f0 = {x[0], x[1]}, f1 = {x[2], x[3]};
        u32x2 pk;
        pk[0] = __builtin_bit_cast(unsigned, __builtin_convertvector(f0, bf16x2));
        pk[1] = __builtin_bit_cast(unsigned, __builtin_convertvector(f1, bf16x2));
        *reinterpret_cast<u32x2*>(stg + n * SRS + 16 * j + 8 * h) = pk;
      }
      co_read(NS);
#pragma unroll
      for (int q = 0; q < 2; ++q) co_store(NS, 7, q);
    }
   }
   if constexpr (RED) {
     // the unit's Rj partial: this wave owns accumulator group w / 2, register pair w % 2 -> rows 8 (w / 2) + 4 h + 2 (w % 2) + e
     float* dst = ra.rj_part + (rec * RR_WR + 8 * (w >> 1) + 4 * h + 2 * (w & 1)) * RR_G + n;
#pragma unroll
     for (int ob = 0; ob < 8; ++ob) {
       dst[32 * ob] = racc[ob][0];
       dst[RR_G + 32 * ob] = racc[ob][1];
     }
   }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
#undef RN_LAYER

#ifdef RN_DIAG
static int g_diag_abl = 0;
extern "C" int rn_diag_set_abl(int v) { g_diag_abl = v; return 0; }
#endif
static int rr_prio() {
  const char* e = rn_diag_env("RN_RR_PRIO");               // (RN_DIAG builds: static s_setprio for the second-dispatched waves, measured +-0.3 %)
  return e ? (e[0] != '0') : RR_PRIO_DEFAULT;
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
size_t rnws_rr_mask(int M) { return M > 0 ? (size_t)M * 32 : 0; }

static int rr_check_inject(const char* who, const float* Vq, int inj, int n) {
  RN_CHECK_ARG(inj == 0 || inj == 2, "%s: the question can be injected at layer 0 (tables) or 2 (got %d)", who, inj);
  RN_CHECK_ARG(inj == 0 || (Vq && ((uintptr_t)Vq % 16 == 0)), "%s: injection at layer %d needs the 16-byte aligned question rows Vq", who, inj);
  RN_CHECK_ARG(inj == 0 || ((long)n * n) % RR_TM == 0, "%s: injection at layer %d needs n*n %% %d == 0 (n=%d)", who, inj, RR_TM, n);
  return 0;
}

// Argument checks of the f16s entry point: Whi[0] / Wlo[0] = the two fragment-major fp16 images of layer 0,
// Whi[1..3] = `dither` tile-dithered hi images each (128 KB apart); Wlo[1..3] are not read.
// dither == 0: the TWO-PASS inference arithmetic -- Whi[l] / Wlo[l] = the plain hi / lo split images of every layer.
static int rr_f16s_args(const char* who, RRArgsF& a, const void* const* Whi, const void* const* Wlo, int dither, const float* const* bias,
                        void* const* H, void* const* mask, int* nh_, int* nm_) {
  RN_CHECK_ARG(dither == 0 || dither == 1 || dither == 2 || dither == 4 || dither == 8,
               "%s: dither (hi images per layer) must be 1, 2, 4 or 8, or 0 = hi + lo on every layer (got %d)", who, dither);
  const bool two = dither == 0;
  memset(&a, 0, sizeof(a));
  a.prio = rr_prio();
  a.vmask = two ? 0 : dither - 1;
  int nh = 0, nm = 0;
  for (int l = 0; l < RR_L; ++l) {
    RN_CHECK_ARG(Whi[l] && ((l > 0 && !two) || Wlo[l]) && bias[l], "%s: layer %d weight/bias is NULL", who, l);
    RN_CHECK_ARG(((uintptr_t)Whi[l] | (uintptr_t)((l == 0 || two) ? Wlo[l] : nullptr) | (uintptr_t)bias[l] | (uintptr_t)(H ? H[l] : nullptr) | (uintptr_t)(mask ? mask[l] : nullptr)) % 16 == 0,
                 "%s: layer %d pointers must be 16-byte aligned", who, l);
    a.Whi[l] = (const f16*)Whi[l];
    a.Wlo[l] = (l == 0 || two) ? (const f16*)Wlo[l] : nullptr;
    a.bias[l] = bias[l];
    a.out[l] = H ? (bf16*)H[l] : nullptr;
    a.mask[l] = mask ? (u64*)mask[l] : nullptr;
    nh += a.out[l] != nullptr;
    nm += a.mask[l] != nullptr;
  }
  *nh_ = nh;
  *nm_ = nm;
  return 0;
}

// The forward chain (f16s arithmetic, factored first layer): Xp16 = fp16 object rows (B*n [+ 1], 64).
extern "C" int rn_g_chain_fwd_rr_f16s_alg0(const void* Xp16, const float* Vc, int n, int njp, const void* const* Whi, const void* const* Wlo, int dither,
                                           const float* const* bias, void* const* H, int h_dtype, void* const* mask, int gate_in_h2,
                                           float* xg_part, const float* Vq, int inject_layer, int M, int L, int G, void* stream) {
  RN_CHECK_ARG(Xp16 && Vc && Whi && Wlo && bias && M > 0 && xg_part, "rn_g_chain_fwd_rr_f16s_alg0: bad pointer/size");
  RN_CHECK_ARG(!H || h_dtype == RN_BF16 || h_dtype == RN_FP8, "rn_g_chain_fwd_rr_f16s_alg0: h_dtype must be RN_BF16 or RN_FP8 (got %d)", h_dtype);
  const bool h8 = H && h_dtype == RN_FP8;
  if (int rc = rr_check_inject("rn_g_chain_fwd_rr_f16s_alg0", Vq, inject_layer, n)) return rc;
  RN_CHECK_ARG(G == RR_G && L == RR_L, "rn_g_chain_fwd_rr_f16s_alg0: needs G == 256 and L == 4 (G=%d L=%d)", G, L);
  const bool rag = njp != n;                                          // padded j axis (see g_chain_rr_f16s_kernel, RAG)
  RN_CHECK_ARG(n > 0 && njp >= n && njp - n < RR_WR && njp % RR_WR == 0 && (!rag || n % 4 == 0) && M % ((long)n * njp) == 0 && M % RR_TM == 0,
               "rn_g_chain_fwd_rr_f16s_alg0: needs njp = 32 ceil(n / 32) (n %% 4 == 0 when padded) and M a multiple of n*njp and of %d (n=%d njp=%d M=%d)",
               RR_TM, n, njp, M);
  RN_CHECK_ARG(!rag || inject_layer == 0, "rn_g_chain_fwd_rr_f16s_alg0: a padded j axis goes with the question at layer 0");
  RN_CHECK_ARG(((uintptr_t)Xp16 | (uintptr_t)Vc) % 16 == 0, "rn_g_chain_fwd_rr_f16s_alg0: tables must be 16-byte aligned");
  RRArgsF a;
  int nh = 0, nm = 0;
  if (int rc = rr_f16s_args("rn_g_chain_fwd_rr_f16s_alg0", a, Whi, Wlo, dither, bias, H, mask, &nh, &nm)) return rc;
  const bool h012 = nh == 3 && !a.out[RR_L - 1] && nm == RR_L;
  RN_CHECK_ARG((nh == 0 && nm == 0) || h012, "rn_g_chain_fwd_rr_f16s_alg0: H / masks: none (inference) or H_0..2 + all four masks (training)");
  RN_CHECK_ARG(!gate_in_h2 || (h8 && h012), "rn_g_chain_fwd_rr_f16s_alg0: the gate in the sign bits of H_2 goes with the e4m3 training output set");
  const bool gate = gate_in_h2 != 0;
  const bool two = dither == 0;
  RN_CHECK_ARG(!two || nh == 0, "rn_g_chain_fwd_rr_f16s_alg0: dither == 0 (hi + lo on every layer) is the inference arithmetic: no H / mask outputs");
  const int ntiles = M / RR_TM;
  const int grid = ntiles < rr_num_cus() ? ntiles : rr_num_cus();
  hipStream_t s = (hipStream_t)stream;
  const int rpb = n * n;
  const f16* Xp = (const f16*)Xp16;
  if (two) {
    const int nz = rag ? (M / (n * njp)) * n : 0;
    if (rag) g_chain_rr_f16s_kernel<4, false, false, false, true, true, 0, false, 0, true, false, true><<<grid, RR_NT, 0, s>>>(Xp, 64, a, xg_part, ntiles, Vc, n, nullptr, 1, njp, nz);
    else if (inject_layer == 2) g_chain_rr_f16s_kernel<4, false, false, false, true, true, 2, false, 0, false, false, true><<<grid, RR_NT, 0, s>>>(Xp, 64, a, xg_part, ntiles, Vc, n, Vq, rpb);
    else g_chain_rr_f16s_kernel<4, false, false, false, true, true, 0, false, 0, false, false, true><<<grid, RR_NT, 0, s>>>(Xp, 64, a, xg_part, ntiles, Vc, n);
  } else if (rag) {
    const int nz = (M / (n * njp)) * n;                               // the all-zero object row behind the B * n real ones
    if (nh == 0) g_chain_rr_f16s_kernel<4, false, false, false, true, true, 0, false, 0, true><<<grid, RR_NT, 0, s>>>(Xp, 64, a, xg_part, ntiles, Vc, n, nullptr, 1, njp, nz);
    else if (h8 && gate) g_chain_rr_f16s_kernel<4, true, false, true, true, true, 0, true, 0, true, true><<<grid, RR_NT, 0, s>>>(Xp, 64, a, xg_part, ntiles, Vc, n, nullptr, 1, njp, nz);
    else if (h8) g_chain_rr_f16s_kernel<4, true, false, true, true, true, 0, true, 0, true><<<grid, RR_NT, 0, s>>>(Xp, 64, a, xg_part, ntiles, Vc, n, nullptr, 1, njp, nz);
    else g_chain_rr_f16s_kernel<4, true, false, true, true, true, 0, false, 0, true><<<grid, RR_NT, 0, s>>>(Xp, 64, a, xg_part, ntiles, Vc, n, nullptr, 1, njp, nz);
  } else if (inject_layer == 2) {
    if (nh == 0) g_chain_rr_f16s_kernel<4, false, false, false, true, true, 2><<<grid, RR_NT, 0, s>>>(Xp, 64, a, xg_part, ntiles, Vc, n, Vq, rpb);
    else if (h8 && gate) g_chain_rr_f16s_kernel<4, true, false, true, true, true, 2, true, 0, false, true><<<grid, RR_NT, 0, s>>>(Xp, 64, a, xg_part, ntiles, Vc, n, Vq, rpb);
    else if (h8) g_chain_rr_f16s_kernel<4, true, false, true, true, true, 2, true><<<grid, RR_NT, 0, s>>>(Xp, 64, a, xg_part, ntiles, Vc, n, Vq, rpb);
    else g_chain_rr_f16s_kernel<4, true, false, true, true, true, 2><<<grid, RR_NT, 0, s>>>(Xp, 64, a, xg_part, ntiles, Vc, n, Vq, rpb);
  } else {
    if (nh == 0) g_chain_rr_f16s_kernel<4, false, false, false, true, true><<<grid, RR_NT, 0, s>>>(Xp, 64, a, xg_part, ntiles, Vc, n);
#ifdef RN_DIAG
#define RN_ABL(v) else if (h8 && g_diag_abl == v) g_chain_rr_f16s_kernel<4, true, false, true, true, true, 0, true, v><<<grid, RR_NT, 0, s>>>(Xp, 64, a, xg_part, ntiles, Vc, n);
    RN_ABL(1) RN_ABL(2) RN_ABL(4) RN_ABL(6) RN_ABL(8) RN_ABL(16) RN_ABL(24) RN_ABL(64) RN_ABL(32) RN_ABL(96) RN_ABL(102) RN_ABL(88)
#undef RN_ABL
#endif
    else if (h8 && gate) g_chain_rr_f16s_kernel<4, true, false, true, true, true, 0, true, 0, false, true><<<grid, RR_NT, 0, s>>>(Xp, 64, a, xg_part, ntiles, Vc, n);
    else if (h8) g_chain_rr_f16s_kernel<4, true, false, true, true, true, 0, true><<<grid, RR_NT, 0, s>>>(Xp, 64, a, xg_part, ntiles, Vc, n);
    else g_chain_rr_f16s_kernel<4, true, false, true, true, true><<<grid, RR_NT, 0, s>>>(Xp, 64, a, xg_part, ntiles, Vc, n);
  }
  RN_LAUNCH_CHECK("rn_g_chain_fwd_rr_f16s_alg0");
  return 0;
}

extern "C" int rn_g_chain_bwd_rr(const float* dxg, const void* const* mask, const void* const* Wtf, void* const* dZ, int M,
                                 int rows_per_question, int L, int G, void* stream) {
  RN_CHECK_ARG(dxg && mask && Wtf && dZ && M > 0, "rn_g_chain_bwd_rr: bad pointer/size");
  RN_CHECK_ARG(G == RR_G && L == RR_L, "rn_g_chain_bwd_rr: needs G == 256 and L == 4 (G=%d L=%d)", G, L);
  RN_CHECK_ARG(M % RR_TM == 0 && rows_per_question > 0 && M % rows_per_question == 0,
               "rn_g_chain_bwd_rr: M=%d must be a multiple of %d and of rows per question=%d", M, RR_TM, rows_per_question);
  RRBwdArgs a;
  memset(&a, 0, sizeof(a));
  a.prio = rr_prio();
  const bool skip0 = dZ[0] == nullptr;                     // the last layer's gradient is not stored (the gate job of rn_g_wgrad_blocked replaces it)
  for (int l = 0; l < RR_L; ++l) {
    RN_CHECK_ARG(mask[l] && (dZ[l] || (l == 0 && skip0)) && (l == RR_L - 1 || Wtf[l]), "rn_g_chain_bwd_rr: entry %d has a NULL pointer", l);
    RN_CHECK_ARG(((uintptr_t)mask[l] | (uintptr_t)dZ[l] | (uintptr_t)(l < RR_L - 1 ? Wtf[l] : nullptr)) % 16 == 0,
                 "rn_g_chain_bwd_rr: entry %d pointers must be 16-byte aligned", l);
  }
  // the kernel addresses the per-layer buffers as base + l * stride (three pointers instead of eleven)
  a.mask = (const u64*)mask[0];
  a.dz_stride = (bf16*)dZ[2] - (bf16*)dZ[1];
  a.dZ = (bf16*)dZ[1] - a.dz_stride;
  a.W = (const bf16*)Wtf[0];
  a.mask_stride = (const u64*)mask[1] - (const u64*)mask[0];
  a.w_stride = (const bf16*)Wtf[1] - (const bf16*)Wtf[0];
  for (int l = 0; l < RR_L; ++l) {
    RN_CHECK_ARG((const u64*)mask[l] == a.mask + l * a.mask_stride && ((l == 0 && skip0) || (bf16*)dZ[l] == a.dZ + l * a.dz_stride) &&
                     (l == RR_L - 1 || (const bf16*)Wtf[l] == a.W + l * a.w_stride),
                 "rn_g_chain_bwd_rr: mask / dZ / Wtf buffers must be equally spaced (slices of one allocation each)");
  }
  RN_CHECK_ARG((uintptr_t)dxg % 16 == 0, "rn_g_chain_bwd_rr: dxg must be 16-byte aligned");
  a.dxg = dxg;
  a.rows_per_b = rows_per_question;
  const int ntiles = M / RR_TM;
  const int grid = ntiles < rr_num_cus() ? ntiles : rr_num_cus();
  if (skip0) {
    g_chain_rr_bwd_kernel<0, true><<<grid, RR_NT, 0, (hipStream_t)stream>>>(a, ntiles);
    RN_LAUNCH_CHECK("rn_g_chain_bwd_rr");
    return 0;
  }
#ifdef RN_DIAG
  switch (g_diag_abl) {                                    // timing-only ablations of the stored-dZ[0] variant (results are wrong)
#define RN_ABL(v) case v: g_chain_rr_bwd_kernel<v><<<grid, RR_NT, 0, (hipStream_t)stream>>>(a, ntiles); break;
    RN_ABL(1) RN_ABL(2) RN_ABL(3) RN_ABL(4) RN_ABL(7) RN_ABL(15) RN_ABL(16) RN_ABL(17) RN_ABL(32) RN_ABL(64)
#undef RN_ABL
    default: g_chain_rr_bwd_kernel<0><<<grid, RR_NT, 0, (hipStream_t)stream>>>(a, ntiles); break;
  }
#else
  g_chain_rr_bwd_kernel<0><<<grid, RR_NT, 0, (hipStream_t)stream>>>(a, ntiles);
#endif
  RN_LAUNCH_CHECK("rn_g_chain_bwd_rr");
  return 0;
}

// The backward chain with the pair-axis reductions of layer 0's gradient formed on chip (g_chain_rr_bwd_kernel, RED).
// tiles of the reducing backward chain: B questions x njp / 32 j blocks x ceil(n / 8) groups of 8 i
static long rr_red_tiles(int M, int n, int njp) { return (long)(M / ((long)n * njp)) * (njp / RR_WR) * ((n + RR_NW - 1) / RR_NW); }

extern "C" int rn_g_chain_bwd_rr_red_tpu(int M, int n, int njp) {
  if (M <= 0 || n <= 0 || njp < n || njp - n >= RR_WR || njp % RR_WR != 0 || M % ((long)n * njp) != 0) return 0;
  const long ntiles = rr_red_tiles(M, n, njp);
  const int tpbj = (n + RR_NW - 1) / RR_NW;                           // tiles per (question, block of 32 j)
  int tpu = 1;                                                        // the largest divisor of tpbj that still gives every CU a unit
  for (int d = 2; d <= tpbj; ++d)
    if (tpbj % d == 0 && ntiles / d >= rr_num_cus()) tpu = d;
  return tpu;
}

// Whole units of the balanced schedule: with U units on C CUs, the first C floor(U / C); the tiles of the other U mod C units are
// handed out one by one.  (tiles_per_unit == 1: a unit IS a tile -- all whole.)
extern "C" int rn_g_chain_bwd_rr_red_whole(int M, int n, int njp, int tiles_per_unit) {
  if (rn_g_chain_bwd_rr_red_tpu(M, n, njp) <= 0 || tiles_per_unit <= 0 || ((n + RR_NW - 1) / RR_NW) % tiles_per_unit != 0) return -1;
  const int nunits = (int)(rr_red_tiles(M, n, njp) / tiles_per_unit);
  return tiles_per_unit == 1 ? nunits : (nunits / rr_num_cus()) * rr_num_cus();
}

// The work items of a launch as the kernel decodes them (diagnostics: the CPU tests check that every tile is run exactly once and that
// rn_pair_reduce_parts reads exactly the records that are written).  out: (items, 3) ints = first tile, tiles, record.
extern "C" int rn_probe_red_schedule(int M, int n, int njp, int tiles_per_unit, int units_whole, int* out, int max_items) {
  RN_CHECK_ARG(rn_g_chain_bwd_rr_red_tpu(M, n, njp) > 0 && tiles_per_unit > 0 && ((n + RR_NW - 1) / RR_NW) % tiles_per_unit == 0, "rn_probe_red_schedule: bad shape");
  const int tpbj = (n + RR_NW - 1) / RR_NW, jgs = njp / RR_WR, nu = tpbj / tiles_per_unit;
  const int nunits = (int)(rr_red_tiles(M, n, njp) / tiles_per_unit), nb = nunits / (jgs * nu);
  RN_CHECK_ARG(units_whole >= 0 && units_whole <= nunits, "rn_probe_red_schedule: units_whole=%d must lie in [0, %d]", units_whole, nunits);
  const long nitems = units_whole + (long)(nunits - units_whole) * tiles_per_unit;
  RN_CHECK_ARG(out && nitems <= max_items, "rn_probe_red_schedule: %ld items do not fit (max_items=%d)", nitems, max_items);
  for (int i = 0; i < (int)nitems; ++i) {
    const RnRedItem it = rn_red_item(i, nunits, tiles_per_unit, units_whole, nb, jgs, nu);
    out[3 * i] = it.tile0; out[3 * i + 1] = it.tcount; out[3 * i + 2] = (int)it.rec;
  }
  return (int)nitems;
}

extern "C" int rn_g_chain_bwd_rr_red(const float* dxg, const void* const* mask, const void* const* Wtf, void* const* dZ, int M, int n,
                                     int njp, int L, int G, float* rj_part, float* ri_part, int tiles_per_unit, int units_whole, void* stream) {
  RN_CHECK_ARG(dxg && mask && Wtf && dZ && rj_part && ri_part && M > 0, "rn_g_chain_bwd_rr_red: bad pointer/size");
  RN_CHECK_ARG(G == RR_G && L == RR_L, "rn_g_chain_bwd_rr_red: needs G == 256 and L == 4 (G=%d L=%d)", G, L);
  RN_CHECK_ARG(rn_g_chain_bwd_rr_red_tpu(M, n, njp) > 0,
               "rn_g_chain_bwd_rr_red: needs njp = 32 ceil(n / 32) pair rows per (question, i) and M a multiple of n*njp (M=%d n=%d njp=%d)", M, n, njp);
  const int ntiles = (int)rr_red_tiles(M, n, njp), tpbj = (n + RR_NW - 1) / RR_NW;
  RN_CHECK_ARG(tiles_per_unit > 0 && tpbj % tiles_per_unit == 0, "rn_g_chain_bwd_rr_red: tiles_per_unit=%d must divide ceil(n / 8) = %d", tiles_per_unit, tpbj);
  RRBwdArgs a;
  memset(&a, 0, sizeof(a));
  a.prio = rr_prio();
  RN_CHECK_ARG(dZ[0] == nullptr && dZ[1] && dZ[2], "rn_g_chain_bwd_rr_red: dZ[0] must be NULL (gate job), dZ[1], dZ[2] the stored images; dZ[3] is not written");
  for (int l = 0; l < RR_L; ++l) {
    RN_CHECK_ARG(mask[l] && (l == RR_L - 1 || Wtf[l]), "rn_g_chain_bwd_rr_red: entry %d has a NULL pointer", l);
    RN_CHECK_ARG(((uintptr_t)mask[l] | (uintptr_t)(l == 1 || l == 2 ? dZ[l] : nullptr) | (uintptr_t)(l < RR_L - 1 ? Wtf[l] : nullptr)) % 16 == 0,
                 "rn_g_chain_bwd_rr_red: entry %d pointers must be 16-byte aligned", l);
  }
  a.mask = (const u64*)mask[0];
  a.dz_stride = (bf16*)dZ[2] - (bf16*)dZ[1];
  a.dZ = (bf16*)dZ[1] - a.dz_stride;
  a.W = (const bf16*)Wtf[0];
  a.mask_stride = (const u64*)mask[1] - (const u64*)mask[0];
  a.w_stride = (const bf16*)Wtf[1] - (const bf16*)Wtf[0];
  for (int l = 0; l < RR_L; ++l) {
    RN_CHECK_ARG((const u64*)mask[l] == a.mask + l * a.mask_stride && (l == RR_L - 1 || (const bf16*)Wtf[l] == a.W + l * a.w_stride),
                 "rn_g_chain_bwd_rr_red: mask / Wtf buffers must be equally spaced (slices of one allocation each)");
  }
  RN_CHECK_ARG(((uintptr_t)dxg | (uintptr_t)rj_part | (uintptr_t)ri_part) % 16 == 0, "rn_g_chain_bwd_rr_red: dxg / partials must be 16-byte aligned");
  a.dxg = dxg;
  a.rows_per_b = n * njp;
  const int nunits = ntiles / tiles_per_unit;
  RN_CHECK_ARG(units_whole >= 0 && units_whole <= nunits, "rn_g_chain_bwd_rr_red: units_whole=%d must lie in [0, %d]", units_whole, nunits);
  RRRedArgs ra{rj_part, ri_part, n, tiles_per_unit, njp, units_whole};
  const long nitems = units_whole + (long)(nunits - units_whole) * tiles_per_unit;
  const int grid = nitems < rr_num_cus() ? (int)nitems : rr_num_cus();
#ifdef RN_DIAG
  switch (g_diag_abl) {                                    // timing-only ablations (results are wrong): 128 the tile prologue (gated dxg operand) only once per workgroup, 2 no mask loads, 4 no dZ stores, 8 no waits / barriers, 32 stores to L2-resident addresses
#define RN_ABL(v) case v: g_chain_rr_bwd_kernel<v, true, true><<<grid, RR_NT, 0, (hipStream_t)stream>>>(a, ntiles, ra); break;
    RN_ABL(2) RN_ABL(4) RN_ABL(6) RN_ABL(8) RN_ABL(14) RN_ABL(32) RN_ABL(64) RN_ABL(128) RN_ABL(142)
#undef RN_ABL
    default: g_chain_rr_bwd_kernel<0, true, true><<<grid, RR_NT, 0, (hipStream_t)stream>>>(a, ntiles, ra); break;
  }
#else
  g_chain_rr_bwd_kernel<0, true, true><<<grid, RR_NT, 0, (hipStream_t)stream>>>(a, ntiles, ra);
#endif
  RN_LAUNCH_CHECK("rn_g_chain_bwd_rr_red");
  return 0;
}
