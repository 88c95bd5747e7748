// wgrad of a g_theta layer (autograd of model.py:141-145):
//   dW[n, k] = sum_m dZ[m, n] * A[m, k]        db[n] = sum_m dZ[m, n]
// A reduction over M = B*n*n (262,144 .. 1.2M) pair rows with a 256 x K output: split over M
// across workgroups (grid.z), fp32 partial tiles in workspace, then a fixed-order reduction ->
// bitwise deterministic (no atomics).
//
// Workgroup = 512 threads (8 waves), output tile 256(n) x (NKT*32)(k); wave w owns features
// 32w..32w+31 and all NKT k-tiles.  Both MFMA operands are read "down a column" of a row-major
// [m][.] LDS tile; for bf16 that is the gfx950 LDS transpose read (ds_read_b64_tr_b16), for fp32
// (v_mfma_f32_32x32x2_f32, one value per lane) a plain ds_read_b32.
#include <stdlib.h>

#include "rn_common.h"
#include "../../include/rn_hip_debug.h"

template <typename T> struct WG;
template <> struct WG<bf16> { static constexpr int ROWS = 64; };
template <> struct WG<float> { static constexpr int ROWS = 32; };

template <typename T, int NKT, bool USE_TR>
__global__ __launch_bounds__(512) void wgrad_kernel(const T* __restrict__ dZ, int lddz, const T* __restrict__ A,
                                                    int lda, float* __restrict__ part, float* __restrict__ part_db,
                                                    int M, int N, int Kpad, int rows_per_split) {
  constexpr int CH = Elem<T>::kPer16B;
  constexpr int ROWS = WG<T>::ROWS;
  // LDS row stride (bytes), both tiles.  bf16: +64 B so that the 4 rows x 64 B a 32-lane group of
  // ds_read_b64_tr_b16 touches land on 4 distinct 16-bank quarters (stride = 16 dwords mod 64); fp32: +16 B.
  constexpr int RSZ = 256 * (int)sizeof(T) + (sizeof(T) == 2 ? 64 : 16);
  constexpr int CPZ = 256 / CH;                            // 16-byte chunks per dZ tile row
  constexpr int CPA = NKT * 32 / CH;                       // chunks per A tile row
  constexpr int NZ = ROWS * CPZ / 512;                     // dZ chunks per thread per step (4)
  constexpr int NA = (ROWS * CPA + 511) / 512;             // A chunks per thread per step (<= 4)
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * ROWS * RSZ];   // 66-68 KB static
  unsigned char* ldsZ = lds;
  unsigned char* ldsA = lds + ROWS * RSZ;

  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int n0 = blockIdx.x * 256;
  const int k0 = blockIdx.y * NKT * 32;
  const int kvalid = (Kpad - k0) < NKT * 32 ? (Kpad - k0) : NKT * 32;   // columns of A that exist
  const long r_begin = (long)blockIdx.z * rows_per_split;
  long r_end = r_begin + rows_per_split;
  if (r_end > M) r_end = M;

  // fixed staging assignment
  int zrow[NZ], zcc[NZ], arow[NA], acc_[NA];
  bool aok[NA];
#pragma unroll
  for (int s = 0; s < NZ; ++s) {
    const int c = t + 512 * s;
    zrow[s] = c / CPZ;
    zcc[s] = c % CPZ;
  }
#pragma unroll
  for (int s = 0; s < NA; ++s) {
    const int c = t + 512 * s;
    arow[s] = c / CPA;
    acc_[s] = c % CPA;
    aok[s] = (c < ROWS * CPA) && (acc_[s] * CH < kvalid);
  }
  u32x4 rz[NZ], ra[NA];
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  auto gload = [&](long r0) {
#pragma unroll
    for (int s = 0; s < NZ; ++s) {
      const long r = r0 + zrow[s];
      rz[s] = (r < r_end) ? *reinterpret_cast<const u32x4*>(dZ + r * lddz + n0 + zcc[s] * CH) : zero4;
    }
#pragma unroll
    for (int s = 0; s < NA; ++s) {
      const long r = r0 + arow[s];
      ra[s] = (aok[s] && r < r_end) ? *reinterpret_cast<const u32x4*>(A + r * lda + k0 + acc_[s] * CH) : zero4;
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int s = 0; s < NZ; ++s) *reinterpret_cast<u32x4*>(ldsZ + zrow[s] * RSZ + zcc[s] * 16) = rz[s];
#pragma unroll
    for (int s = 0; s < NA; ++s)
      if (t + 512 * s < ROWS * CPA) *reinterpret_cast<u32x4*>(ldsA + arow[s] * RSZ + acc_[s] * 16) = ra[s];
  };

  f32x16 acc[NKT];
#pragma unroll
  for (int i = 0; i < NKT; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  float dbsum = 0.f;
  const bool do_db = (blockIdx.y == 0) && (t < 256);

  const int i16 = lane & 15, cb = (lane >> 4) & 1, rb = lane >> 5;
  gload(r_begin);
  lstore();
  __syncthreads();
  for (long r0 = r_begin; r0 < r_end; r0 += ROWS) {
    const bool more = (r0 + ROWS) < r_end;
    if (more) gload(r0 + ROWS);
    if constexpr (sizeof(T) == 2) {
#pragma unroll
      for (int kk0 = 0; kk0 < ROWS; kk0 += 16) {
        bf16x8 fz, fa[NKT];
        if constexpr (USE_TR) {
          // lane supplies the address of row (kk0 + rb*8 + 4h + i16/4), cols (cb*16 + 4*(i16%4)..+4) of its
          // 16-lane block and receives column i16 of that 4x16 block: 4 consecutive m for one n / k.
          const int roff = (kk0 + rb * 8 + (i16 >> 2)) * RSZ;
          const int coff = (cb * 16 + 4 * (i16 & 3)) * 2;
          typedef __attribute__((address_space(3))) s16x4* lptr;
          const unsigned char* pz = ldsZ + roff + (w * 32) * 2 + coff;
          s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(pz));
          s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(pz + 4 * RSZ));
          union { struct { s16x4 a, b; } s; bf16x8 v; } u;
          u.s.a = lo; u.s.b = hi; fz = u.v;
#pragma unroll
          for (int kt = 0; kt < NKT; ++kt) {
            const unsigned char* pa = ldsA + roff + (kt * 32) * 2 + coff;
            u.s.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(pa));
            u.s.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(pa + 4 * RSZ));
            fa[kt] = u.v;
          }
        } else {
          const int rbase = kk0 + rb * 8;
#pragma unroll
          for (int e = 0; e < 8; ++e)
            fz[e] = *reinterpret_cast<const bf16*>(ldsZ + (rbase + e) * RSZ + (w * 32 + (lane & 31)) * 2);
#pragma unroll
          for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int e = 0; e < 8; ++e)
              fa[kt][e] = *reinterpret_cast<const bf16*>(ldsA + (rbase + e) * RSZ + (kt * 32 + (lane & 31)) * 2);
        }
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) acc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fz, fa[kt], acc[kt], 0, 0, 0);
      }
    } else {
#pragma unroll 4
      for (int kk0 = 0; kk0 < ROWS; kk0 += 2) {
        const int roff = (kk0 + rb) * RSZ;
        const float fz = *reinterpret_cast<const float*>(ldsZ + roff + (w * 32 + (lane & 31)) * 4);
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
          const float fa = *reinterpret_cast<const float*>(ldsA + roff + (kt * 32 + (lane & 31)) * 4);
          acc[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(fz, fa, acc[kt], 0, 0, 0);
        }
      }
    }
    if (do_db) {
#pragma unroll 8
      for (int r = 0; r < ROWS; ++r) dbsum += Elem<T>::to_f32(*reinterpret_cast<const T*>(ldsZ + r * RSZ + t * sizeof(T)));
    }
    __syncthreads();
    if (more) {
      lstore();
      __syncthreads();
    }
  }

  // ---- write the fp32 partial tile: part[z][n][k], lane = 32 consecutive k of one feature row
  float* pz = part + (long)blockIdx.z * N * Kpad;
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) {
    const int kcol = k0 + kt * 32 + (lane & 31);
    if (kcol < Kpad) {
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int nrow = n0 + w * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
        pz[(long)nrow * Kpad + kcol] = acc[kt][reg];
      }
    }
  }
  if (do_db) part_db[(long)blockIdx.z * N + n0 + t] = dbsum;
}

// Ordered reduction of the per-split partial tiles.  Z slabs of E4 float4's each; a workgroup owns
// 16 consecutive float4 outputs and splits the Z slabs 16 ways (16 independent 16-byte loads in
// flight per thread), then combines the 16 slices through LDS in a fixed order -> deterministic.
// Blocks [0, nbw) reduce dW (trimming the K padding), blocks [nbw, ..) reduce db.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part,
                                                           const float* __restrict__ part_db, float* __restrict__ dW,
                                                           float* __restrict__ db, int N, int Kpad, int Ktrue, int Z,
                                                           int nbw) {
  __shared__ f32x4 red[16][16];
  const int o = threadIdx.x & 15, zs = threadIdx.x >> 4;
  const bool is_w = (int)blockIdx.x < nbw;
  const long E4 = is_w ? (long)N * Kpad / 4 : N / 4;
  const long g4 = (long)(is_w ? blockIdx.x : blockIdx.x - nbw) * 16 + o;
  const f32x4* src = reinterpret_cast<const f32x4*>(is_w ? part : part_db);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (g4 < E4) {
#pragma unroll 4
    for (int z = zs; z < Z; z += 16) acc += src[(long)z * E4 + g4];
  }
  red[zs][o] = acc;
  __syncthreads();
  if (zs == 0 && g4 < E4) {
    f32x4 sum = red[0][o];
#pragma unroll
    for (int i = 1; i < 16; ++i) sum += red[i][o];
    if (is_w) {
      const long e = g4 * 4;
      const int n = (int)(e / Kpad), k = (int)(e - (long)n * Kpad);     // Kpad % 4 == 0: the 4 lanes share n
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (k + i < Ktrue) dW[(long)n * Ktrue + k + i] = sum[i];
    } else if (db) {
      *reinterpret_cast<f32x4*>(db + g4 * 4) = sum;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// Streaming wgrad for the headline shapes (bf16, N == 256, K == 256 or 192, M % 64 == 0).  The product is
// HBM-bound by 2x (each operand row is read exactly once: 268 MB per layer against 34 GFLOP), so the kernel is
// built around bytes in flight, not around the MFMA:
//   * no register staging: 64-row operand tiles stream HBM -> LDS by LDS-DMA into a 4-stage ring, three stages
//     (96 KB per CU) in flight, one counted s_waitcnt vmcnt + s_barrier per 64 rows;
//   * output split 2 x KBLK: a workgroup owns a 128 (n) x KB (k) block, KB = 128 (K = 256) or 64 (K = 192), and
//     1/Z of the rows -- the fp32 partial set is 4x smaller than with full-width blocks (16 MB per layer instead
//     of 64 MB written and read back);
//   * the 2 * KBLK workgroups that read the same rows (other column halves) sit on ONE XCD (blockIdx % 8), a few
//     dispatch slots apart: the second reader hits that XCD's L2;
//   * the linear LDS image LDS-DMA writes is made conflict-free for ds_read_b64_tr_b16 by an XOR swizzle of the
//     16-byte chunks, applied to the per-lane SOURCE address (the rows of a 4-row group land 64 B apart mod 256 B).
// Same partial-tile format and the same fixed-order reduction as wgrad_kernel -> bitwise deterministic.
// GEN -- the LAST g layer's gradient operand is not read but generated: dZ_3 = (ReLU gate of layer 3) x dxg[question]
// (model.py:151-152: the pair sum broadcasts one gradient row to all pairs of a question), so the 134 MB bf16 matrix the
// backward chain would write and this kernel read back is replaced by the forward kernel's lane masks (1 KB per 64-row
// step and column half, LDS-DMA into an 8-slot ring three steps ahead of their use) + the question's dxg row: every
// thread builds two 16-byte chunks of the step's dZ tile (8 mask bits select among 8 pre-rounded bf16 values) and writes
// them to the swizzled position the LDS-DMA would have used.  The A operand (H_2) streams as before.
// A8 -- the A operand (the stored activations H_{l-1}) arrives as OCP e4m3 bytes (rn_common.h, RN_H8_SCALE): half the bytes
// of the operand that is two thirds of this kernel's traffic once dZ_3 is generated (GEN) and a third otherwise.  The tile is
// 64 rows x 128 B; ONE ds_read_b64_tr_b8 hands a lane its column's 8 consecutive rows (the 16 lanes of a group supply an
// 8-row x 16-byte block), four v_cvt_scalef32_pk_bf16_fp8 turn them into the bf16 MFMA operand -- exact, e4m3 is a subset
// of bf16.  lda counts BYTES (= elements) then.
// Measured and dropped (round 2): the same loop software-pipelined by half a step (the transpose reads of a half issued ahead of
// the previous half's MFMAs, inline-asm reads with hand-placed lgkmcnt waits because hipcc turns its own into full drains at
// the loop header): 69.5 / 62.6 us (bf16 / e4m3 A) against 64.8 / 62.1 for this loop on the same chip, the generated-operand
// variant 87 against 70 -- with two waves per SIMD the LDS issue rate, not the read/MFMA phase order, sets the 57 us of
// compute under the stream, and a fourth stage of look-ahead costs the second readers their L2 hits (84 / 78 us).
template <int KTW, bool GEN = false, bool A8 = false>   // 32-wide k tiles per wave: 2 -> KB = 128, 1 -> KB = 64
__global__ __launch_bounds__(512) void wgrad_stream_kernel(const bf16* __restrict__ dZ, int lddz, const bf16* __restrict__ A,
                                                           int lda, float* __restrict__ part, float* __restrict__ part_db,
                                                           int S, int Z, int NB, int Kpad, int abl,
                                                           const unsigned* __restrict__ gmask = nullptr,
                                                           const float* __restrict__ dxg = nullptr, int steps_per_q = 1) {
  static_assert(!GEN || KTW == 2, "generated operand: K == 256 only");
  static_assert(!A8 || KTW == 2, "fp8 operand: K == 256 only");
  constexpr int KB = KTW * 64;                             // k block width
  constexpr int ES = A8 ? 1 : 2;                           // bytes per A element
  constexpr int ZB = 64 * 256, AB = 64 * KB * ES;          // bytes per stage: dZ tile (64 x 128 cols), A tile (64 x KB cols)
  constexpr int STG = ZB + AB, NSTG = 4, LA = 3;           // three stages (96 KB) in flight; a 5-stage ring measured slower
  constexpr int PPW = GEN ? AB / 1024 / 8 : (ZB + AB) / 1024 / 8;   // 1-KB LDS-DMA pieces per wave and stage (4 / 3; GEN: 2 + a mask slice)
  constexpr int RBA = KB * ES;                             // A tile row bytes (256 / 128; fp8: 128)
  constexpr int MSLOTS = 8, MLA = LA + 3;                  // GEN: mask ring, requested MLA steps ahead of the step that multiplies them
  __shared__ __attribute__((aligned(16))) unsigned char lds[NSTG * STG + (GEN ? MSLOTS * 1024 : 0)];
  typedef __attribute__((address_space(3))) unsigned char lds_u8;
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  // XCD-aware decode: consecutive ids round-robin over the 8 XCDs; the NB blocks of one row range share an XCD
  const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
  const int blk = slot % NB, z = (slot / NB) * 8 + xcd;
  if (z >= Z) return;
  const int nh = blk & 1, kb = blk >> 1;
  const long s0 = (long)z * S / Z, s1 = (long)(z + 1) * S / Z;  // 64-row steps of this workgroup

  // ---- LDS-DMA: per-lane source offsets (swizzled), uniform bases per piece
  const int zr = lane >> 4, zc = (lane & 15) ^ (4 * (zr & 3));
  const unsigned zoff = (unsigned)(zr * lddz * 2 + zc * 16);
  unsigned aoff;
  if (A8) {
    const int ar = lane >> 3, ac = lane & 7;
    aoff = (unsigned)(ar * lda + ((((ac >> 1) ^ ((ar >> 1) & 3)) << 1) | (ac & 1)) * 16);
  } else if (KB == 128) {
    aoff = (unsigned)(zr * lda * 2 + zc * 16);
  } else {
    const int ar = lane >> 3, ac = (lane & 7) ^ (4 * ((ar >> 1) & 1));
    aoff = (unsigned)(ar * lda * 2 + ac * 16);
  }
  const unsigned ldsb = (unsigned)(size_t)(lds_u8*)lds;
  auto issue = [&](long s) {                               // stage of step s -> ring slot s % NSTG
    const unsigned sb = ldsb + (unsigned)(s % NSTG) * STG;
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int q = GEN ? 16 + w * PPW + i : w * PPW + i;  // wave-uniform (GEN: the 16 A pieces only)
      const unsigned char* ub;
      unsigned off;
      if (q < 16) {                                        // dZ piece: rows 4q .. 4q+3, 256 B each
        ub = reinterpret_cast<const unsigned char*>(dZ + (s * 64 + 4 * q) * lddz + nh * 128);
        off = zoff;
      } else {                                             // A piece: 1 KB = 4 rows x 256 B or 8 rows x 128 B
        const int qa = q - 16;
        ub = reinterpret_cast<const unsigned char*>(A) + ((s * 64 + qa * (1024 / RBA)) * lda + kb * KB) * ES;
        off = aoff;
      }
      const unsigned dst = sb + q * 1024;
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep)
                   : "v"(off), "s"(ub), "s"(dst)
                   : "memory");
    }
  };

  // ---- GEN: lane masks by LDS-DMA (8 lanes x 16 B per wave = 128 B of the step's 1 KB), tile generation
  // mask image of layer 3 (un-swapped epilogue of the forward kernel): per 32-row block and 32-feature block 32 dwords,
  // dword 2 (4 (r / 8) + r % 4) + (r / 4) % 2 = the 32 feature bits of row r
  auto issue_mask = [&](long x) {
    const unsigned char* ub = reinterpret_cast<const unsigned char*>(gmask) + x * 2048 + (w >> 2) * 1024 + nh * 512 + (w & 3) * 128;
    const unsigned dst = ldsb + (unsigned)(NSTG * STG) + (unsigned)(x % MSLOTS) * 1024 + w * 128;
    const unsigned off = (unsigned)lane * 16u;
    unsigned keep;
    unsigned long long ex;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b64 %1, exec\n\ts_mov_b32 m0, %4\n\ts_mov_b64 exec, 0xff\n\t"
                 "global_load_lds_dwordx4 %2, %3\n\ts_mov_b64 exec, %1\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep), "=&s"(ex)
                 : "v"(off), "s"(ub), "s"(dst)
                 : "memory");
  };
  const int gc = t & 15, gr = t >> 4;                      // this thread's chunk column and row (rows gr and gr + 32)
  const int gdw = 2 * (4 * (gr >> 3) + (gr & 3)) + ((gr >> 2) & 1);
  unsigned dxbf[4] = {0u, 0u, 0u, 0u};                     // bf16 pairs of dxg[question][nh * 128 + 8 gc + 0..7]
  int cur_q = -1, q_left = 0;                              // steps are generated in order: the question changes every steps_per_q
  auto gen_load = [&](long x, unsigned (&mb)[2]) {         // the two mask dwords of this thread's rows gr and gr + 32
    const unsigned char* ms = lds + NSTG * STG + ((unsigned)x % MSLOTS) * 1024 + (gc >> 2) * 128 + gdw * 4;
    mb[0] = *reinterpret_cast<const unsigned*>(ms);
    mb[1] = *reinterpret_cast<const unsigned*>(ms + 512);
  };
  auto gen_store = [&](long x, const unsigned (&mb)[2]) {
    if (q_left == 0) {
      if (cur_q < 0) {
        cur_q = (int)(x / steps_per_q);
        q_left = steps_per_q - (int)(x - (long)cur_q * steps_per_q);
      } else {
        ++cur_q;
        q_left = steps_per_q;
      }
      const float* dp = dxg + (long)cur_q * 256 + nh * 128 + 8 * gc;
      const f32x4 d0 = *reinterpret_cast<const f32x4*>(dp), d1 = *reinterpret_cast<const f32x4*>(dp + 4);
      typedef __attribute__((ext_vector_type(2))) float f32x2_;
      typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_;
      const f32x2_ p0 = {d0[0], d0[1]}, p1 = {d0[2], d0[3]}, p2 = {d1[0], d1[1]}, p3 = {d1[2], d1[3]};
      dxbf[0] = __builtin_bit_cast(unsigned, __builtin_convertvector(p0, bf16x2_));
      dxbf[1] = __builtin_bit_cast(unsigned, __builtin_convertvector(p1, bf16x2_));
      dxbf[2] = __builtin_bit_cast(unsigned, __builtin_convertvector(p2, bf16x2_));
      dxbf[3] = __builtin_bit_cast(unsigned, __builtin_convertvector(p3, bf16x2_));
    }
    --q_left;
    unsigned char* zt = lds + ((unsigned)x % NSTG) * STG;
#pragma unroll
    for (int blk2 = 0; blk2 < 2; ++blk2) {
      const unsigned bits = mb[blk2] >> (8 * (gc & 3));
      u32x4 o;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const unsigned t0 = (unsigned)__builtin_amdgcn_sbfe((int)bits, 2 * p, 1), t1 = (unsigned)__builtin_amdgcn_sbfe((int)bits, 2 * p + 1, 1);
        o[p] = dxbf[p] & ((t0 & 0xffffu) | (t1 & 0xffff0000u));      // v_bfe_i32 x2, v_bfi_b32, v_and_b32
      }
      const int R = 32 * blk2 + gr;
      *reinterpret_cast<u32x4*>(zt + R * 256 + ((gc ^ (4 * (R & 3))) * 16)) = o;
    }
  };

  // ---- fragment addresses (ds_read_b64_tr_b16: the lane supplies row kk + rb*8 + i16/4 (+4), 4 columns)
  const int i16 = lane & 15, cb = (lane >> 4) & 1, rb = lane >> 5;
  const int fr = rb * 8 + (i16 >> 2);                      // row inside a 16-row k step (second read: + 4)
  const int nb = w & 3, kg = w >> 2;                       // wave: n block (32 features), k group
  const int zcol = nb * 32 + cb * 16 + 4 * (i16 & 3);      // column inside the 128-wide dZ tile
  const unsigned zaddr = (unsigned)(fr * 256 + (((zcol >> 3) ^ (4 * (fr & 3))) * 16) + (zcol & 4) * 2);
  unsigned aaddr[KTW];
#pragma unroll
  for (int kt = 0; kt < KTW; ++kt) {
    if constexpr (A8) {
      const int fr8 = rb * 8 + (i16 >> 1), pair = kg * KTW + kt;
      aaddr[kt] = (unsigned)(ZB + fr8 * RBA + ((((pair ^ ((fr8 >> 1) & 3)) << 1) | cb) * 16) + 8 * (i16 & 1));
    } else {
      const int acol = kg * (KTW * 32) + kt * 32 + cb * 16 + 4 * (i16 & 3);
      const int sw = KB == 128 ? 4 * (fr & 3) : 4 * ((fr >> 1) & 1);
      aaddr[kt] = (unsigned)(ZB + fr * RBA + (((acol >> 3) ^ sw) * 16) + (acol & 4) * 2);
    }
  }

  f32x16 acc[KTW];
#pragma unroll
  for (int i = 0; i < KTW; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  f32x16 acc_db;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc_db[e] = 0.f;
  bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (bf16)1.0f;
  const bool do_db = (kb == 0) && (kg == 0);

  if constexpr (GEN) {
    for (long x = s0; x < s0 + MLA && x < s1; ++x) issue_mask(x);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (long x = s0; x < s0 + LA && x < s1; ++x) {
      unsigned mb[2];
      gen_load(x, mb);
      gen_store(x, mb);
      issue(x);
    }
  } else if (!(abl & 2)) {
    for (long s = s0; s < s0 + LA && s < s1; ++s) issue(s);
  }
  for (long s = s0; s < s1; ++s) {
    // stage s has landed when at most the requests of the younger stages in flight are outstanding
    if constexpr (GEN) {
      // per step and wave: [mask slice of step s+MLA][2 A pieces of step s+LA]; needed now: the A pieces of step s and the
      // mask slice of step s+LA (requested BEFORE them) -- both older than the last two steps' 3 + 3 requests; the first
      // two steps see only the prologue's A pieces behind theirs (2 + 2); the last five steps (whose predecessors issued
      // fewer requests) drain
      if (s + 5 >= s1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (s < s0 + 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPW) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPW + 2) : "memory");
    } else {
      const long younger = (s1 - 1 - s) < (LA - 1) ? (s1 - 1 - s) : (LA - 1);
      if (younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPW) : "memory");
      else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();                          // ... for every wave's pieces; and slot (s-1) % NSTG is free
    asm volatile("" ::: "memory");
    unsigned mb[2] = {0u, 0u};
    if constexpr (GEN) {
      // order per step and wave: [mask slice of step s+MLA] ... [2 A pieces of step s+LA]; the tile of step s+LA is built
      // between them, its mask read ahead of and its VALU work + writes behind this step's transpose reads (the LDS
      // returns in order: the build then rides on the read latency instead of preceding it)
      if (s + MLA < s1 && !(abl & 8)) issue_mask(s + MLA);
      if (s + LA < s1 && !(abl & 16)) gen_load(s + LA, mb);
    } else if (s + LA < s1 && !(abl & 2)) issue(s + LA);
    const unsigned char* st = lds + (s % NSTG) * STG;
    typedef __attribute__((address_space(3))) s16x4* lptr;
    if (abl & 1) continue;                                 // diagnostics: stream only
    // all 24 (16) transpose reads of the step first, then the MFMAs back to back: read -> wait -> MFMA per k step
    // exposes the LDS latency four times per step with every wave of the workgroup in the same phase
    union Fr { struct { s16x4 a, b; } s; bf16x8 v; };
    Fr uz[4], ua[4][KTW];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      uz[kk].s.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(st + kk * 16 * 256 + zaddr));
      uz[kk].s.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(st + kk * 16 * 256 + 4 * 256 + zaddr));
#pragma unroll
      for (int kt = 0; kt < KTW; ++kt) {
        if constexpr (A8) {
          typedef __attribute__((ext_vector_type(2))) int i32x2_;
          typedef __attribute__((address_space(3))) i32x2_* lptr8;
          const i32x2_ r8 = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lptr8)(st + kk * 16 * RBA + aaddr[kt]));
          ua[kk][kt].s.a = __builtin_bit_cast(s16x4, r8);
        } else {
          ua[kk][kt].s.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(st + kk * 16 * RBA + aaddr[kt]));
          ua[kk][kt].s.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(st + kk * 16 * RBA + 4 * RBA + aaddr[kt]));
        }
      }
    }
    if constexpr (GEN) {
      __builtin_amdgcn_sched_barrier(0);
      if (s + LA < s1) {
        if (!(abl & 4)) gen_store(s + LA, mb);
        issue(s + LA);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
      for (int kt = 0; kt < KTW; ++kt) {
        if constexpr (A8) {
          const u32x2 raw = __builtin_bit_cast(u32x2, ua[kk][kt].s.a);
          acc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(uz[kk].v, rn_bf16x8_from_fp8(raw[0], raw[1]), acc[kt], 0, 0, 0);
        } else {
          acc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(uz[kk].v, ua[kk][kt].v, acc[kt], 0, 0, 0);
        }
      }
      // db = dZ^T 1: one more MFMA against a tile of ones (every output column carries the column sums)
      if (do_db) acc_db = __builtin_amdgcn_mfma_f32_32x32x16_bf16(uz[kk].v, ones, acc_db, 0, 0, 0);
    }
  }

  // ---- fp32 partial tile: part[z][n][k]; a lane holds 32 consecutive k... one k column of 16 feature rows
  float* pz = part + (long)z * 256 * Kpad;
#pragma unroll
  for (int kt = 0; kt < KTW; ++kt) {
    const int kcol = kb * KB + kg * (KTW * 32) + kt * 32 + (lane & 31);
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int nrow = nh * 128 + nb * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
      pz[(long)nrow * Kpad + kcol] = acc[kt][reg];
    }
  }
  if (do_db && (lane & 31) == 0) {
#pragma unroll
    for (int reg = 0; reg < 16; ++reg)
      part_db[(long)z * 256 + nh * 128 + nb * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)] = acc_db[reg];
  }
}

static void wgrad_plan(int M, int N, int K, int* nkt, int* gy, int* Z, int* rps) {
  const int ktiles = K / 32;
  const int chunks = (ktiles + 7) / 8;
  *nkt = (ktiles + chunks - 1) / chunks;
  *gy = (ktiles + *nkt - 1) / *nkt;
  const int gx = N / 256;
  int z = 256 / (gx * *gy);
  if (z < 1) z = 1;
  const int steps = (M + 63) / 64;
  if (z > steps) z = steps;
  int r = ((M + z - 1) / z + 63) / 64 * 64;
  z = (M + r - 1) / r;
  *Z = z;
  *rps = r;
}

extern "C" size_t rn_wgrad_ws_bytes(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0 || N % 256 || K % 32) return 0;
  int nkt, gy, Z, rps;
  wgrad_plan(M, N, K, &nkt, &gy, &Z, &rps);
  return ((size_t)Z * N * K + (size_t)Z * N) * sizeof(float);
}

// The streaming kernel covers bf16, N == 256, K in {256, 192}, whole 64-row steps (RN_WGRAD_V1=1 forces the general kernel).
static bool wgrad_stream_ok(int dtype, int M, int N, int K, int lddz, int lda) {
  const char* e = getenv("RN_WGRAD_V1");
  if (e && e[0] == '1') return false;
  const char* k192 = getenv("RN_WGRAD_STREAM_192");          // K == 192 (three 64-wide k blocks) measured slower than the general kernel
  if (K == 192 && !(k192 && k192[0] == '1')) return false;
  return dtype == RN_BF16 && N == 256 && (K == 256 || K == 192) && M % 64 == 0 && M / 64 >= 64 && lddz % 8 == 0 && lda % 8 == 0;
}

// Number of row splits the streaming kernel uses for this product (0: the general kernel runs instead).  With Z splits the
// workspace holds, behind the Z x N x K weight partials, Z x N fp32 column sums of dZ over rows [z M / Z, (z + 1) M / Z) -- the
// bias-gradient partials.  When a split never straddles two questions they are also the per-question sums of dZ that the
// question-injected layer's backward needs (Rq): no second pass over dZ.
extern "C" int rn_wgrad_stream_splits(int dtype, int a_dtype, int M, int N, int K, int lddz, int lda) {
  if (a_dtype == RN_FP8) {
    if (!(dtype == RN_BF16 && N == 256 && K == 256 && M % 64 == 0 && M / 64 >= 64)) return 0;
  } else if (!wgrad_stream_ok(dtype, M, N, K, lddz, lda)) {
    return 0;
  }
  const char* ze = getenv("RN_WGRAD_ZS");
  const int NB = K == 256 ? 4 : 6;
  return ze ? atoi(ze) : 256 / NB;
}

// Weight gradient of the LAST g layer from the forward kernel's lane masks (see wgrad_stream_kernel, GEN): dW = dZ^T A,
// db = colsum(dZ) with dZ[(b, pair), f] = (mask bit) ? bf16(dxg[b][f]) : 0 -- bitwise what rn_g_chain_bwd_rr would store.
extern "C" int rn_g_linear_bwd_wgrad_gated(const void* mask, const float* dxg, int rows_per_question, const void* A, int lda,
                                           int a_dtype, float* dW, float* db, void* ws, int M, int N, int K, void* stream) {
  RN_CHECK_ARG(mask && dxg && A && dW && db && ws && M > 0, "rn_g_linear_bwd_wgrad_gated: bad pointer/size");
  RN_CHECK_ARG(a_dtype == RN_BF16 || a_dtype == RN_FP8, "rn_g_linear_bwd_wgrad_gated: A must be bf16 or e4m3 (a_dtype=%d)", a_dtype);
  RN_CHECK_ARG(N == 256 && K == 256 && M % 64 == 0 && M / 64 >= 64 && lda % (a_dtype == RN_FP8 ? 16 : 8) == 0 && lda >= K,
               "rn_g_linear_bwd_wgrad_gated: needs N == K == 256, M %% 64 == 0, M >= 4096 (M=%d N=%d K=%d)", M, N, K);
  RN_CHECK_ARG(rows_per_question > 0 && rows_per_question % 64 == 0 && M % rows_per_question == 0,
               "rn_g_linear_bwd_wgrad_gated: rows_per_question=%d must be a multiple of 64 dividing M", rows_per_question);
  RN_CHECK_ARG(((uintptr_t)mask | (uintptr_t)dxg | (uintptr_t)A) % 16 == 0, "rn_g_linear_bwd_wgrad_gated: pointers must be 16-byte aligned");
  const char* ze = getenv("RN_WGRAD_ZS");                   // diagnostics: M-split (workgroups = 4 x Zs); default fills the chip
  const int NB = 4, S = M / 64, Zs = (ze && atoi(ze) > 0 && atoi(ze) <= 256 / NB) ? atoi(ze) : 256 / NB;
  float* part = (float*)ws;
  float* part_db = part + (size_t)Zs * N * K;
  hipStream_t s = (hipStream_t)stream;
  const char* ae = getenv("RN_WGRAD_ABL");
  if (a_dtype == RN_FP8)
    wgrad_stream_kernel<2, true, true><<<8 * NB * cdiv(Zs, 8), 512, 0, s>>>(nullptr, 0, (const bf16*)A, lda, part, part_db, S, Zs, NB, K, ae ? atoi(ae) : 0,
                                                                            (const unsigned*)mask, dxg, rows_per_question / 64);
  else
    wgrad_stream_kernel<2, true><<<8 * NB * cdiv(Zs, 8), 512, 0, s>>>(nullptr, 0, (const bf16*)A, lda, part, part_db, S, Zs, NB, K, ae ? atoi(ae) : 0,
                                                                      (const unsigned*)mask, dxg, rows_per_question / 64);
  RN_LAUNCH_CHECK("rn_g_linear_bwd_wgrad_gated(stream)");
  const int nbw = cdiv((long)N * K / 4, 16), nbb = cdiv(N / 4, 16);
  wgrad_reduce_kernel<<<nbw + nbb, 256, 0, s>>>(part, part_db, dW, db, N, K, K, Zs, nbw);
  RN_LAUNCH_CHECK("rn_g_linear_bwd_wgrad_gated(reduce)");
  return 0;
}

// USE_TR can be turned off (RN_WGRAD_NO_TR=1) to fall back to scalar LDS column reads.
static bool wgrad_use_tr() {
  const char* e = getenv("RN_WGRAD_NO_TR");       // read per call so tests can flip it
  return !(e && e[0] == '1');
}

template <typename T, int NKT>
static void wgrad_dispatch(bool tr, dim3 grid, hipStream_t s, const T* dZ, int lddz, const T* A, int lda,
                           float* part, float* part_db, int M, int N, int K, int rps) {
  if constexpr (sizeof(T) == 2) {
    if (tr) wgrad_kernel<T, NKT, true><<<grid, 512, 0, s>>>(dZ, lddz, A, lda, part, part_db, M, N, K, rps);
    else wgrad_kernel<T, NKT, false><<<grid, 512, 0, s>>>(dZ, lddz, A, lda, part, part_db, M, N, K, rps);
  } else {
    wgrad_kernel<T, NKT, false><<<grid, 512, 0, s>>>(dZ, lddz, A, lda, part, part_db, M, N, K, rps);
  }
}

template <typename T>
static int wgrad_launch_t(const T* dZ, int lddz, const T* A, int lda, float* part, float* part_db, int M, int N, int K,
                          int nkt, int gy, int Z, int rps, hipStream_t s) {
  dim3 grid(N / 256, gy, Z);
  const bool tr = wgrad_use_tr();
  switch (nkt) {
#define RN_CASE(n) case n: wgrad_dispatch<T, n>(tr, grid, s, dZ, lddz, A, lda, part, part_db, M, N, K, rps); break;
    RN_CASE(1) RN_CASE(2) RN_CASE(3) RN_CASE(4) RN_CASE(5) RN_CASE(6) RN_CASE(7) RN_CASE(8)
#undef RN_CASE
    default: rn_set_error("rn_g_linear_bwd_wgrad: bad k-tile count %d", nkt); return -1;
  }
  return 0;
}

extern "C" int rn_g_linear_bwd_wgrad(const void* dZ, int lddz, const void* A, int lda, int a_dtype, float* dW, float* db, void* ws,
                                     int dtype, int M, int N, int K, int Ktrue, void* stream) {
  RN_CHECK_ARG(dZ && A && dW && ws && M > 0, "rn_g_linear_bwd_wgrad: bad pointer/size");
  RN_CHECK_ARG(dtype == RN_BF16 || dtype == RN_F32, "rn_g_linear_bwd_wgrad: bad dtype %d", dtype);
  if (a_dtype == RN_FP8) {
    // e4m3 copy of the activations (the forward chains' h_dtype = RN_FP8): the streaming kernel only
    RN_CHECK_ARG(dtype == RN_BF16 && N == 256 && K == 256 && Ktrue == K && M % 64 == 0 && M / 64 >= 64 && lddz % 8 == 0 && lddz >= N &&
                     lda % 16 == 0 && lda >= K && ((uintptr_t)dZ | (uintptr_t)A) % 16 == 0,
                 "rn_g_linear_bwd_wgrad: an e4m3 A needs bf16 dZ, N == K == 256, M %% 64 == 0, M >= 4096 (M=%d N=%d K=%d)", M, N, K);
    const char* ze = getenv("RN_WGRAD_ZS");
    const int NB = 4, S = M / 64, Zs = ze ? atoi(ze) : 256 / NB;
    float* part = (float*)ws;
    float* part_db = part + (size_t)Zs * N * K;
    const char* ae = getenv("RN_WGRAD_ABL");
    wgrad_stream_kernel<2, false, true><<<8 * NB * cdiv(Zs, 8), 512, 0, (hipStream_t)stream>>>((const bf16*)dZ, lddz, (const bf16*)A, lda, part, part_db, S,
                                                                                                Zs, NB, K, ae ? atoi(ae) : 0);
    RN_LAUNCH_CHECK("rn_g_linear_bwd_wgrad(stream, e4m3 A)");
    const int nbw = cdiv((long)N * K / 4, 16), nbb = cdiv(N / 4, 16);
    wgrad_reduce_kernel<<<nbw + nbb, 256, 0, (hipStream_t)stream>>>(part, part_db, dW, db, N, K, Ktrue, Zs, nbw);
    RN_LAUNCH_CHECK("rn_g_linear_bwd_wgrad(reduce)");
    return 0;
  }
  RN_CHECK_ARG(a_dtype == dtype, "rn_g_linear_bwd_wgrad: A must have dZ's type or be e4m3 (a_dtype=%d)", a_dtype);
  const int CH = dtype == RN_BF16 ? 8 : 4;
  RN_CHECK_ARG(N % 256 == 0 && K % 32 == 0 && Ktrue > 0 && Ktrue <= K, "rn_g_linear_bwd_wgrad: N=%d K=%d Ktrue=%d unsupported",
               N, K, Ktrue);
  RN_CHECK_ARG(lddz % CH == 0 && lda % CH == 0 && lddz >= N && lda >= K, "rn_g_linear_bwd_wgrad: bad leading dimensions");
  RN_CHECK_ARG(((uintptr_t)dZ | (uintptr_t)A) % 16 == 0, "rn_g_linear_bwd_wgrad: pointers must be 16-byte aligned");
  int nkt, gy, Z, rps;
  wgrad_plan(M, N, K, &nkt, &gy, &Z, &rps);
  float* part = (float*)ws;
  hipStream_t s = (hipStream_t)stream;
  if (wgrad_stream_ok(dtype, M, N, K, lddz, lda)) {
    const char* ze = getenv("RN_WGRAD_ZS");
    const int NB = K == 256 ? 4 : 6, S = M / 64;                     // (2 n halves) x (2 | 3 k blocks); M-split so that ~256 workgroups run
    const int Zs = ze ? atoi(ze) : 256 / NB;
    float* part_db = part + (size_t)Zs * N * K;
    const int grid = 8 * NB * cdiv(Zs, 8);
    const char* ae = getenv("RN_WGRAD_ABL");
    const int abl = ae ? atoi(ae) : 0;
    if (K == 256) wgrad_stream_kernel<2><<<grid, 512, 0, s>>>((const bf16*)dZ, lddz, (const bf16*)A, lda, part, part_db, S, Zs, NB, K, abl);
    else wgrad_stream_kernel<1><<<grid, 512, 0, s>>>((const bf16*)dZ, lddz, (const bf16*)A, lda, part, part_db, S, Zs, NB, K, abl);
    RN_LAUNCH_CHECK("rn_g_linear_bwd_wgrad(stream)");
    const int nbw = cdiv((long)N * K / 4, 16), nbb = cdiv(N / 4, 16);
    wgrad_reduce_kernel<<<nbw + nbb, 256, 0, s>>>(part, part_db, dW, db, N, K, Ktrue, Zs, nbw);
    RN_LAUNCH_CHECK("rn_g_linear_bwd_wgrad(reduce)");
    return 0;
  }
  float* part_db = part + (size_t)Z * N * K;
  int rc;
  if (dtype == RN_BF16)
    rc = wgrad_launch_t<bf16>((const bf16*)dZ, lddz, (const bf16*)A, lda, part, part_db, M, N, K, nkt, gy, Z, rps, s);
  else
    rc = wgrad_launch_t<float>((const float*)dZ, lddz, (const float*)A, lda, part, part_db, M, N, K, nkt, gy, Z, rps, s);
  if (rc) return rc;
  RN_LAUNCH_CHECK("rn_g_linear_bwd_wgrad");
  const int nbw = cdiv((long)N * K / 4, 16), nbb = cdiv(N / 4, 16);
  wgrad_reduce_kernel<<<nbw + nbb, 256, 0, s>>>(part, part_db, dW, db, N, K, Ktrue, Z, nbw);
  RN_LAUNCH_CHECK("rn_g_linear_bwd_wgrad(reduce)");
  return 0;
}

// ---- diagnostic: raw lane mapping of ds_read_b64_tr_b16 (checked by tests/test_gpu_kernels.py)
__global__ void probe_tr16_kernel(const unsigned short* in, unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short l[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) l[i] = in[i];
  __syncthreads();
  typedef __attribute__((address_space(3))) s16x4* lptr;
  // lane supplies the address of 4 consecutive u16 at element offset lane*4 (a linear image)
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(l + threadIdx.x * 4));
  for (int e = 0; e < 4; ++e) out[threadIdx.x * 4 + e] = (unsigned short)v[e];
}

extern "C" int rn_probe_tr16(const unsigned short* in4096, unsigned short* out256, void* stream) {
  probe_tr16_kernel<<<1, 64, 0, (hipStream_t)stream>>>(in4096, out256);
  RN_LAUNCH_CHECK("rn_probe_tr16");
  return 0;
}

// ---- diagnostic: raw lane mapping of ds_read_b64_tr_b8 (linear image: lane l supplies &lds[8 l])
__global__ void probe_tr8_kernel(const unsigned char* in, unsigned char* out) {
  __shared__ __attribute__((aligned(16))) unsigned char l[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) l[i] = in[i];
  __syncthreads();
  typedef __attribute__((ext_vector_type(2))) int i32x2_;
  typedef __attribute__((address_space(3))) i32x2_* lptr8;
  const i32x2_ v = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lptr8)(l + threadIdx.x * 8));
  reinterpret_cast<i32x2_*>(out)[threadIdx.x] = v;
}

extern "C" int rn_probe_tr8(const unsigned char* in4096, unsigned char* out512, void* stream) {
  probe_tr8_kernel<<<1, 64, 0, (hipStream_t)stream>>>(in4096, out512);
  RN_LAUNCH_CHECK("rn_probe_tr8");
  return 0;
}

// ---- diagnostic: the e4m3 conversions the kernels use.  in: n floats (n % 4 == 0); out8_*: n bytes through the bf16 / fp16
// down-conversion at scale `scale`; back: n floats = the up-conversion of out8_bf at the same scale
__global__ void probe_fp8_cvt_kernel(const float* in, float scale, unsigned* out8_bf, unsigned* out8_h, float* back, int n4) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= n4) return;
  const f32x4 v = reinterpret_cast<const f32x4*>(in)[i];
  typedef __attribute__((ext_vector_type(2))) float f32x2_;
  const f32x2_ a = {v[0], v[1]}, b = {v[2], v[3]};
  const bf16x2 ba = __builtin_convertvector(a, bf16x2), bb = __builtin_convertvector(b, bf16x2);
  const rn_f16x2 ha = __builtin_convertvector(a, rn_f16x2), hb = __builtin_convertvector(b, rn_f16x2);
  unsigned q;
  if (scale == RN_H8_SCALE && v[0] >= 0.f && v[1] >= 0.f && v[2] >= 0.f && v[3] >= 0.f) {   // the kernels' own helpers (with their clamp)
    q = rn_fp8x4_from_bf16(__builtin_bit_cast(unsigned, ba), __builtin_bit_cast(unsigned, bb));
    out8_h[i] = rn_fp8x4_from_f16(__builtin_bit_cast(unsigned, ha), __builtin_bit_cast(unsigned, hb));
  } else {                                                                                   // the raw instructions
    rn_s16x2 r = {0, 0};
    r = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(r, ba, scale, false);
    r = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(r, bb, scale, true);
    q = __builtin_bit_cast(unsigned, r);
    rn_s16x2 r2 = {0, 0};
    r2 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(r2, ha, scale, false);
    r2 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(r2, hb, scale, true);
    out8_h[i] = __builtin_bit_cast(unsigned, r2);
  }
  out8_bf[i] = q;
  const bf16x2 u0 = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(q, scale, false), u1 = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(q, scale, true);
  f32x4 o = {(float)u0[0], (float)u0[1], (float)u1[0], (float)u1[1]};
  reinterpret_cast<f32x4*>(back)[i] = o;
}

extern "C" int rn_probe_fp8_cvt(const float* in, float scale, void* out8_bf16, void* out8_f16, float* back, int n, void* stream) {
  RN_CHECK_ARG(in && out8_bf16 && out8_f16 && back && n > 0 && n % 4 == 0, "rn_probe_fp8_cvt: bad arguments");
  probe_fp8_cvt_kernel<<<cdiv(n / 4, 64), 64, 0, (hipStream_t)stream>>>(in, scale, (unsigned*)out8_bf16, (unsigned*)out8_f16, back, n / 4);
  RN_LAUNCH_CHECK("rn_probe_fp8_cvt");
  return 0;
}
