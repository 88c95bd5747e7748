// Weight gradients of the 256-wide g_theta layers (autograd of model.py:141-145) on ROW-BLOCKED operand images:
//   dW[n, k] = sum_m dZ[m, n] * A[m, k]        db[n] = sum_m dZ[m, n]        M = B*n*n pair rows (262,144 .. 1.2 M)
//
// Both operands of this product are read "down a column" (the reduction runs over the pair rows), while the chains that
// produce them hold a pair row per lane.  Each of H_0..2 / dZ_1..3 has exactly ONE reader -- this kernel -- so the chains
// store them for it (rn_chain_rr.hip, copy-out through one LDS transpose read):
//   16-bit image (bf16 dZ, bf16 H):  element (m, f) at ((m / 8) * 256 + f) * 8 + m % 8     -- 16 bytes = 8 rows of one feature
//   e4m3 image (H copies, gates):    byte    (m, f) at ((m / 16) * 256 + f) * 16 + m % 16  -- 16 bytes = 16 rows of one feature
// i.e. 16 bytes ARE one lane's MFMA operand (32x32x16: a lane supplies 8 consecutive k of one row/column): the tiles stream
// HBM -> LDS by LDS-DMA as they lie (linear, no swizzle: 32 consecutive features = 512 contiguous bytes = every bank once per
// ds_read_b128 lane group) and every operand is ONE plain ds_read_b128 -- no ds_read_b64_tr_*, a third of the LDS instructions
// of the row-major version, whose 16-24 transpose reads per wave and 64-row step with two waves per SIMD in lock step were its
// floor (57 us of reads + MFMAs under a 44-us stream, round 2).
//
// Mapping.  A workgroup is 4 waves, one per SIMD, and owns a 128 (n) x 128 (k) block of dW over 1/Z of the rows; a wave owns
// 64 x 64 of it: per 16-row k-step 2 dZ + 2 A fragments feed 4 MFMAs on 4 independent accumulators.  The NB = 4 workgroups of
// one row range sit on one XCD (second reader = L2 hit).  Per 64-row step a stage of 16 / 24 / 32 KB lands in a ring of
// 8 / 6 / 4 stages.  One counted s_waitcnt vmcnt + s_barrier per step makes stage s+1 visible -- ONE STAGE AHEAD of its use --
// so the fragments of the next step's first half are read before the barrier that ends this step: the barrier costs its skew,
// not an LDS round trip.  The step is hand-scheduled: 16 MFMA gaps, each with its share of the fragment reads, of the stage
// requests and of the e4m3 -> bf16 conversions of the NEXT half step (with one wave per SIMD nothing else covers an instruction
// that waits).
//   reduction order inside a 32-row group (both operands agree, any bijection is a valid contraction order): MFMA a takes
//   rows {0..7} u {16..23}, MFMA b rows {8..15} u {24..31} -- an e4m3 image hands a lane both in one 16-byte read.
//   db = dZ^T 1: one more MFMA against a tile of ones, spread evenly: of the 4 k-steps of a 64-row step each of the 4 waves
//   that hold an n block (2 workgroups x 2 waves) takes one -- 4 partial rows per split, 18 instead of 16 MFMAs per wave.
// GATE jobs -- the LAST layer's gradient is never stored: dZ_3[(b, pair), f] = gate_3 x dxg[b][f] (model.py:151-152: the pair sum
//   broadcasts one row to all pairs of a question).  The gate comes IN THE SIGN BITS of the e4m3 H_2 image (the forward chain
//   writes byte (m, f) = e4m3(H_2[m, f]) | gate_3[m, f] << 7; post-ReLU bytes are never negative): ONE image is both operands.
//   A fragment dword d becomes the gate operand (d >> 1) & 0x40404040 -- e4m3 {0, 2.0} -- or the activation operand
//   d & 0x7f7f7f7f in the MFMA gaps of the half step before its use (3 VALU instructions per gap), both go to the fp8 matrix pipe
//   (v_mfma_f32_32x32x16_fp8_fp8: products of {0, 2} and e4m3 values are exact, fp32 accumulate); the accumulators of a question
//   are scaled by half its dxg row -- in fp32, un-rounded -- when the question ends.  (Round 3: a separate {0, 1} byte image of the
//   gate, 67 MB written by the forward chain and read here for 8 MB of information; before that the tile rebuilt inside this
//   kernel from the forward's lane masks -- 80 bit-test VALU instructions per wave and step beside 18 MFMAs: 75 us against 45.)
// Same fp32 partial format and fixed-order reduction as the general kernel: bitwise deterministic.  Up to 4 jobs (the three
// layers of a step) run as ONE launch + ONE reduction launch.
#include "rn_common.h"
#include "../../include/rn_hip_debug.h"

#ifndef KB_OCC
#define KB_OCC 1          // workgroups per CU the launch bounds allow (2: with rings of 3 / 4 stages two fit, 8 waves per CU)
#endif
#ifndef KB_RING24
#define KB_RING24 6       // ring stages of a 24-KB step (bf16 dZ + e4m3 A) ...
#define KB_RING16 8       // ... and of a 16-KB one (gate job)
#endif
namespace {
constexpr int KB_NT = 256, KB_NB = 4, KB_MAXJOBS = 4;
typedef __attribute__((address_space(3))) unsigned char lds_u8;

struct KbJob {
  const unsigned char* dZ;      // blocked image: 16-bit dZ, or (gate != 0) the e4m3 image A itself (the gate = its sign bits)
  const unsigned char* A;       // blocked image of the layer's input
  const float* dxg;             // gate jobs: (M / rows_per_question, 256) fp32
  float* part;                  // [Z][256][256] fp32
  float* part_db;               // [Z][4][256] fp32
  float* dW;
  float* db;
  int steps_per_q, gate;
  int Z, wide;                  // row splits of THIS job; wide: one workgroup per split holds the whole 256 x 256 dW (kbw_run)
};
struct KbArgs {
  KbJob job[KB_MAXJOBS];
  int njobs, S;                 // S = M / 64
  int nq, nw, grid_q;           // quad jobs (4 workgroups per split), wide jobs, workgroups of the quad part of the grid
  int qjob[KB_MAXJOBS], wjob[KB_MAXJOBS];
};

template <bool Z8, bool A8> struct KbGeo {
  static_assert(!Z8 || A8, "a gate operand goes with the e4m3 activation image (fp8 x fp8 MFMA)");
  static constexpr int ZB = 64 * 128 * (Z8 ? 1 : 2);      // dZ tile of a 64-row step
  static constexpr int AB = 64 * 128 * (A8 ? 1 : 2);      // A tile
  static constexpr int STG = ZB + AB;
  static constexpr int NSTG = STG <= 16384 ? KB_RING16 : (STG <= 24576 ? KB_RING24 : 4);
  static constexpr int LA = NSTG - 1;                     // stage s + LA is requested in step s
  static constexpr int PZ = ZB / 1024 / 4, PA = AB / 1024 / 4;   // 1-KB LDS-DMA pieces per wave and step
  static constexpr int LDS = NSTG * STG;
};
// the ring of a launch: e4m3 activation images (stored jobs: 24-KB steps, gate jobs: 16-KB steps) or bf16 ones (32-KB steps)
template <bool A8> constexpr int kb_lds_bytes() {
  return A8 ? (KbGeo<false, true>::LDS > KbGeo<true, true>::LDS ? KbGeo<false, true>::LDS : KbGeo<true, true>::LDS) : KbGeo<false, false>::LDS;
}

__device__ __forceinline__ void kb_dma(const unsigned char* uniform_src, unsigned lane_off, unsigned lds_dst) {
  unsigned keep;
  // NON-TEMPORAL requests: the launch streams 470 MB beside the latency-bound kernels that close the backward pass; with the default
  // policy it walks their working set out of the XCD's L2 and the Infinity Cache (the step with this launch's requests ablated:
  // 0.62 ms instead of 0.73 -- its arithmetic costs the step nothing, its stream 100 us).  nt: -2.5 % on the step, the launch
  // alone unchanged (176 vs 178 us; the partner workgroup's second read of a line still hits).
#ifndef KB_DMA_POLICY
#define KB_DMA_POLICY " nt"    // (variant builds: "", " sc1", " sc1 nt", ..)
#endif
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" KB_DMA_POLICY "\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(lane_off), "s"(uniform_src), "s"(lds_dst)
               : "memory");
}

template <bool Z8, bool A8> struct KbFrag {
  u32x4 dz[Z8 ? 1 : 2][2];                                // 16-bit: [MFMA a / b][n block]; e4m3: [0][n block] = 16 rows (a | b)
  u32x4 a[A8 ? 1 : 2][2];                                 // likewise, k blocks
};

// ABL (RN_DIAG builds only, tools/): timing ablations with WRONG results -- 1: the stream alone (no fragment reads, no MFMAs),
// 2: compute alone (no requests), 4: no barriers, 8 / 16: no A / dZ fragment reads, 64: no conversions, 256: every step requests the
// addresses of steps 0..3 (all L2 hits: the L2 -> LDS path alone), 512: each workgroup requests only the operand half its partner
// does not (dZ where kb == 0, A where nh == 0: no second reader)
template <bool Z8, bool A8, int ABL = 0>
__device__ __forceinline__ void kb_run(unsigned char* lds, const KbJob& jb_, int S, int Z, int z, int nh, int kb) {
  typedef KbGeo<Z8, A8> G;
  const int t = threadIdx.x, lane = t & 63, n = lane & 31, h = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6), wn = w >> 1, wk = w & 1;
  // 64-row steps of this workgroup (scalars: the loop control and every request address stay on the SALU)
  const int s0 = __builtin_amdgcn_readfirstlane((int)((long)z * S / Z)), s1 = __builtin_amdgcn_readfirstlane((int)((long)(z + 1) * S / Z));
  // the job's pointers once, into SGPRs (the by-value kernel argument is indexed by a run-time job number: without this
  // the loop re-reads them from the kernarg segment and every such scalar load drains the LDS queue)
  KbJob jb = jb_;
  asm volatile("" : "+s"(jb.dZ), "+s"(jb.A), "+s"(jb.dxg), "+s"(jb.steps_per_q));
  const unsigned ldsb = (unsigned)(size_t)(lds_u8*)lds;
  const unsigned lane16 = (unsigned)lane * 16u;

  // ---- the stream: stage of step s -> ring slot s % NSTG.  Wave w: dZ pieces PZ w .., A pieces PA w .. (16-bit: 8-row block
  // q / 2, e4m3: 16-row block q / 2; 64-feature half q % 2)
  constexpr int NPIECE = G::PZ + G::PA;                   // requests per wave and step
  auto issue_piece = [&](int s, int slot, int idx) {
    const unsigned sb = ldsb + (unsigned)slot * G::STG;
    if (ABL & 256) s &= 3;
    if ((ABL & 512) && (idx < G::PZ ? kb != 0 : nh != 0)) return;
    if (idx < G::PZ) {
      const int q = G::PZ * w + idx;
      const long rb = Z8 ? s * 4 + (q >> 1) : s * 8 + (q >> 1);
      kb_dma(jb.dZ + ((rb * 256 + nh * 128 + (q & 1) * 64) * 16), lane16, sb + q * 1024);
    } else {
      const int q = G::PA * w + (idx - G::PZ);
      const long rb = A8 ? s * 4 + (q >> 1) : s * 8 + (q >> 1);
      kb_dma(jb.A + ((rb * 256 + kb * 128 + (q & 1) * 64) * 16), lane16, sb + G::ZB + q * 1024);
    }
  };
  auto issue = [&](int s, int slot) {
#pragma unroll
    for (int i = 0; i < NPIECE; ++i) issue_piece(s, slot, i);
  };

  // ---- fragment reads: 32-row group R of the stage at `st`; piece q of 3
  const unsigned zoff = (unsigned)(((Z8 ? h : 2 * h) * 128 + wn * 64 + n) * 16);
  const unsigned aoff = (unsigned)(G::ZB + ((A8 ? h : 2 * h) * 128 + wk * 64 + n) * 16);
  auto read_piece = [&](const unsigned char* st, int R, KbFrag<Z8, A8>& f, int q) {
    if ((ABL & 8) && q == 2) return;                      // (ablation: no A fragment reads)
    if ((ABL & 16) && q < 2) return;                      // (ablation: no dZ fragment reads)
    if (q < 2) {                                          // the dZ fragments of both n blocks: MFMA a (q = 0), MFMA b (q = 1)
      if constexpr (Z8) {
        if (q == 0) {
#pragma unroll
          for (int i = 0; i < 2; ++i) f.dz[0][i] = *reinterpret_cast<const u32x4*>(st + zoff + ((2 * R) * 128 + 32 * i) * 16);
        }
      } else {
#pragma unroll
        for (int i = 0; i < 2; ++i) f.dz[q][i] = *reinterpret_cast<const u32x4*>(st + zoff + ((4 * R + q) * 128 + 32 * i) * 16);
      }
    } else if constexpr (A8) {
#pragma unroll
      for (int j = 0; j < 2; ++j) f.a[0][j] = *reinterpret_cast<const u32x4*>(st + aoff + ((2 * R) * 128 + 32 * j) * 16);
    } else {
#pragma unroll
      for (int ab = 0; ab < 2; ++ab)
#pragma unroll
        for (int j = 0; j < 2; ++j) f.a[ab][j] = *reinterpret_cast<const u32x4*>(st + aoff + ((4 * R + ab) * 128 + 32 * j) * 16);
    }
  };
  auto read_group = [&](const unsigned char* st, int R, KbFrag<Z8, A8>& f) {
#pragma unroll
    for (int q = 0; q < 3; ++q) read_piece(st, R, f, q);
  };
  // bf16 A operand of MFMA `ab`, k block j: half `hf` (2 dwords) of the bf16x8 -- e4m3: two conversions; 16-bit: as read
  auto conv_half = [&](const KbFrag<Z8, A8>& f, int ab, int j, int hf, u32x4& dst) {
    if constexpr (Z8) {
      return;                                             // (fp8 x fp8: both operands go to the matrix pipe as they are)
    } else {
      if constexpr (A8) {
        const unsigned d = f.a[0][j][2 * ab + hf];
        dst[2 * hf] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(d, RN_H8_SCALE, false));
        dst[2 * hf + 1] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(d, RN_H8_SCALE, true));
      } else {
        dst[2 * hf] = f.a[ab][j][2 * hf];
        dst[2 * hf + 1] = f.a[ab][j][2 * hf + 1];
      }
      // pinned HERE: the compiler otherwise sinks the conversions down to their use (the next half step's MFMAs, where they
      // would precede those instead of riding in this half step's MFMA gaps)
      asm volatile("" : "+v"(dst[2 * hf]), "+v"(dst[2 * hf + 1]));
    }
  };

  // gate jobs: dword 2 ab + hf of the 16-row cells of n block / k block ij -> the gate operand {0, 2.0} and the activation operand
  auto gate_half = [&](KbFrag<Z8, A8>& f, int ab, int ij, int hf) {
    if constexpr (Z8) {
      unsigned z = f.dz[0][ij][2 * ab + hf], x = f.a[0][ij][2 * ab + hf];
      z = (z >> 1) & 0x40404040u;
      x &= 0x7f7f7f7fu;
      asm volatile("" : "+v"(z), "+v"(x));                // (pinned like the conversions)
      f.dz[0][ij][2 * ab + hf] = z;
      f.a[0][ij][2 * ab + hf] = x;
    }
  };

  f32x16 acc[2][2], acc_db[2];                            // Z8: the current question's sums (of 2 gate x A, of 2 gate)
  f32x16 tot[2][2], tot_db[2];                            // Z8: ... scaled by the question's dxg row and added up
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    acc[0][0][e] = 0.f; acc[0][1][e] = 0.f; acc[1][0][e] = 0.f; acc[1][1][e] = 0.f;
    acc_db[0][e] = 0.f; acc_db[1][e] = 0.f;
    if constexpr (Z8) {
      tot[0][0][e] = 0.f; tot[0][1][e] = 0.f; tot[1][0][e] = 0.f; tot[1][1][e] = 0.f;
      tot_db[0][e] = 0.f; tot_db[1][e] = 0.f;
    }
  }
  bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (bf16)1.0f;
  const long ones8 = 0x3838383838383838L;                 // eight e4m3 1.0
  const int db_kk = 2 * kb + wk;                          // the k-step of a 64-row step whose column sums this wave adds
  // Z8: the question of step s ends -> tot += dxg[question] (x) acc, acc = 0.  Accumulator register `reg` of n block i is
  // feature nh 128 + wn 64 + i 32 + 8 (reg / 4) + 4 h + reg % 4.  (A plain load: its s_waitcnt vmcnt(0) drains the request
  // ring, once per question -- with question-aligned splits at the headline shape that is once, behind the loop.)
  int q_cur = 0, q_left = 0;
  if constexpr (Z8) {
    q_cur = s0 / jb.steps_per_q;
    q_left = jb.steps_per_q - (s0 - q_cur * jb.steps_per_q);
  }
  auto flush_question = [&]() {
    const float* dp = jb.dxg + ((long)q_cur * 256 + nh * 128 + wn * 64 + 4 * h);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const f32x4 d = *reinterpret_cast<const f32x4*>(dp + i * 32 + 8 * g4);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int reg = 4 * g4 + r;
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            tot[i][j][reg] += (0.5f * d[r]) * acc[i][j][reg];
            acc[i][j][reg] = 0.f;
          }
          tot_db[i][reg] += (0.5f * d[r]) * acc_db[i][reg];
          acc_db[i][reg] = 0.f;
        }
      }
    ++q_cur;
    q_left = jb.steps_per_q;
  };

  // ---- prologue: LA stages requested; stages s0 and s0 + 1 visible; first fragments in registers
  for (int s = s0; s < s0 + G::LA && s < s1; ++s) issue(s, s % G::NSTG);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"((G::LA - 2) * NPIECE) : "memory");   // (over-waits when fewer stages exist)
  if (s1 - s0 < G::LA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  KbFrag<Z8, A8> f0, f1;
  u32x4 bfc[2], bfn[2];                                   // converted A operands: current half step, next half step
  read_group(lds + (unsigned)(s0 % G::NSTG) * G::STG, 0, f0);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    conv_half(f0, 0, j, 0, bfc[j]);
    conv_half(f0, 0, j, 1, bfc[j]);
    gate_half(f0, 0, j, 0);
    gate_half(f0, 0, j, 1);
  }

  int slot = s0 % G::NSTG;
  for (int s = s0; s < s1; ++s) {
    if constexpr (Z8) {
      if (q_left == 0) flush_question();
      --q_left;
    }
    // stage s + 1 (requested LA - 1 steps ago) must have landed: younger requests = the LA - 2 stages behind it
    if (s + G::LA - 1 < s1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((G::LA - 2) * NPIECE) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (!(ABL & 4)) __builtin_amdgcn_s_barrier();         // ... for every wave's pieces; and every wave is done with slot (s - 1) % NSTG
    asm volatile("" ::: "memory");
    const int pslot = slot == 0 ? G::NSTG - 1 : slot - 1;
    const bool do_issue = s + G::LA < s1 && !(ABL & 2);   // the requests themselves ride in the MFMA gaps below
    const unsigned char* st = lds + (unsigned)slot * G::STG;
    const int nslot = slot + 1 == G::NSTG ? 0 : slot + 1;
    const unsigned char* stn = lds + (unsigned)nslot * G::STG;
    if (ABL & 1) {
      if (do_issue) issue(s + G::LA, pslot);
      slot = nslot;
      continue;
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      const int hh = g >> 2, q = g & 3, R = hh >> 1, ab = hh & 1, i = q >> 1, j = q & 1;
      const KbFrag<Z8, A8>& fr = R ? f1 : f0;
      if constexpr (Z8) {
        const long za = (long)(((unsigned long)fr.dz[0][i][2 * ab + 1] << 32) | fr.dz[0][i][2 * ab]);
        const long aa = (long)(((unsigned long)fr.a[0][j][2 * ab + 1] << 32) | fr.a[0][j][2 * ab]);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(za, aa, acc[i][j], 0, 0, 0);
      } else {
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fr.dz[ab][i]), __builtin_bit_cast(bf16x8, bfc[j]), acc[i][j], 0, 0, 0);
      }
      // ---- fillers of this gap
      if (g < 3) read_piece(st, 1, f1, g);                // this step's second 32-row group
      if (g >= 8 && g < 11) read_piece(stn, 0, f0, g - 8);   // the next step's first group (behind the last step: a slot nobody uses)
      {                                                   // one request of stage s + LA per gap, in the gaps without fragment reads
        const int pi = g >= 3 && g < 8 ? g - 3 : (g >= 11 ? g - 6 : -1);
        if (pi >= 0 && pi < NPIECE && do_issue) issue_piece(s + G::LA, pslot, pi);
      }
      {                                                   // the next half step's A operands: k block q / 2, half q % 2
        const int nh_ = (hh + 1) & 3, nR = nh_ >> 1, nab = nh_ & 1;
        if (!(ABL & 64)) conv_half(nR ? f1 : f0, nab, q >> 1, q & 1, bfn[q >> 1]);
        if (!(ABL & 64)) gate_half(nR ? f1 : f0, nab, q >> 1, q & 1);
      }
      if (q == 3) {
        if (db_kk == hh) {
#pragma unroll
          for (int ii = 0; ii < 2; ++ii) {
            if constexpr (Z8) {
              const long za = (long)(((unsigned long)fr.dz[0][ii][2 * ab + 1] << 32) | fr.dz[0][ii][2 * ab]);
              acc_db[ii] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(za, ones8, acc_db[ii], 0, 0, 0);
            } else {
              acc_db[ii] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fr.dz[ab][ii]), ones, acc_db[ii], 0, 0, 0);
            }
          }
        }
        if constexpr (!Z8) {
          bfc[0] = bfn[0];
          bfc[1] = bfn[1];
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    slot = nslot;
  }
  if constexpr (Z8) flush_question();

  // ---- fp32 partial tile: part[z][n][k]; a lane holds one k column of 16 feature rows
  float* pz = jb.part + (long)z * 256 * 256;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int kcol = kb * 128 + wk * 64 + j * 32 + n;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int nrow = nh * 128 + wn * 64 + i * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
        pz[(long)nrow * 256 + kcol] = Z8 ? tot[i][j][reg] : acc[i][j][reg];
      }
    }
  if (n == 0) {
    float* pd = jb.part_db + ((long)z * 4 + db_kk) * 256;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) pd[nh * 128 + wn * 64 + i * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h] = Z8 ? tot_db[i][reg] : acc_db[i][reg];
  }
}

// ---- WIDE units (round 6): a stored-gradient job on e4m3 activations, ONE workgroup per row split ----------------------------
// The quad mapping above reads every operand byte twice (two of the four workgroups of a row range need it) and, with a 64 x 64
// tile per wave, pays one fragment read and two e4m3 -> bf16 conversions per MFMA and one barrier per 16 MFMAs -- on one wave per
// SIMD that issue stream, not the matrix pipe, was its arithmetic floor (compute alone 127 us against 86 for the MFMAs).  Here a
// workgroup owns the WHOLE 256 x 256 dW of its rows: wave (wn, wk) holds a 128 x 128 block as 16 accumulator tiles = 256 AGPRs
// (the wave has the SIMD's 512 registers to itself), per 32-row group 8 dZ + 4 A fragment reads and 32 conversions feed 32
// MFMAs (0.375 reads, 1 conversion per MFMA: half), a barrier every 32 MFMAs, and every operand byte enters the chip once --
// a 32-row group is 16 KB of the dZ image + 8 KB of the A image, both CONTIGUOUS (all 256 features), 24 one-KB LDS-DMA pieces.
// db: a lane's dZ fragment is 8 rows of ONE feature, so the column sums are in-lane -- v_dot2c_f32_bf16 against (1, 1), four per
// fragment, in the MFMA gaps (the quad mapping spends 2 of 18 MFMAs on them); the two waves that hold the same features take
// MFMA a's / MFMA b's row blocks.  Same partial format as the quad units ([z][256][256] + 4 db rows per split), same fixed-order
// reduction.  Cost: a 256-KB partial tile per workgroup instead of 64 KB.
struct KbwFrag {
  u32x4 dz[2][4];                                         // [MFMA a / b][n block]: 8 rows (block 2 h + ab of the group) of one feature
  u32x4 a[4];                                             // [k block]: 16 rows (block h) of one feature, e4m3
};
#ifndef KBW_RING
#define KBW_RING 6        // ring stages of the wide units (24 KB each): KBW_RING - 1 stages = 120 KB per CU in flight
#endif
constexpr int KBW_ZB = 32 * 256 * 2, KBW_AB = 32 * 256, KBW_STG = KBW_ZB + KBW_AB, KBW_NSTG = KBW_RING, KBW_LA = KBW_NSTG - 1, KBW_NPIECE = 6;
static_assert(KBW_NSTG * KBW_STG <= kb_lds_bytes<true>(), "the wide units' ring must fit the launch's LDS");

template <int DBAB, int ABL = 0>
__device__ __forceinline__ void kbw_run(unsigned char* lds, const KbJob& jb_, int S, int z) {
  const int t = threadIdx.x, lane = t & 63, n = lane & 31, h = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6), wn = w >> 1, wk = w & 1;
  KbJob jb = jb_;
  asm volatile("" : "+s"(jb.dZ), "+s"(jb.A));
  // 32-row groups of this workgroup (row splits are counted in 64-row steps like the quad units': an even number of groups)
  const int s0 = 2 * __builtin_amdgcn_readfirstlane((int)((long)z * S / jb.Z)), s1 = 2 * __builtin_amdgcn_readfirstlane((int)((long)(z + 1) * S / jb.Z));
  const unsigned ldsb = (unsigned)(size_t)(lds_u8*)lds;
  const unsigned lane16 = (unsigned)lane * 16u;
  // the stream: group s -> ring slot s % NSTG; wave w requests dZ pieces 4 w .. 4 w + 3 and A pieces 2 w, 2 w + 1 (1 KB each)
  auto issue_piece = [&](int s, int slot, int idx) {
    const unsigned sb = ldsb + (unsigned)slot * KBW_STG;
    if (ABL & 256) s &= 3;
    if (idx < 4) {
      const int q = 4 * w + idx;
      kb_dma(jb.dZ + ((long)s * KBW_ZB + q * 1024), lane16, sb + q * 1024);
    } else {
      const int q = 2 * w + (idx - 4);
      kb_dma(jb.A + ((long)s * KBW_AB + q * 1024), lane16, sb + KBW_ZB + q * 1024);
    }
  };
  const unsigned zoff = (unsigned)(((2 * h) * 256 + wn * 128 + n) * 16);
  const unsigned aoff = (unsigned)(KBW_ZB + (h * 256 + wk * 128 + n) * 16);
  auto read_piece = [&](const unsigned char* st, KbwFrag& f, int p) {       // 12 pieces of a group
    if (p < 8) {
      if (ABL & 16) return;
      f.dz[p >> 2][p & 3] = *reinterpret_cast<const u32x4*>(st + zoff + ((p >> 2) * 256 + 32 * (p & 3)) * 16);
    } else {
      if (ABL & 8) return;
      f.a[p - 8] = *reinterpret_cast<const u32x4*>(st + aoff + (32 * (p - 8)) * 16);
    }
  };
  // conversion c of 16 of MFMA ab's operands: k block c / 4, source dword 2 ab + (c / 2) % 2, its low / high byte pair
  auto conv = [&](const KbwFrag& f, int ab, int c, u32x4 (&bf)[4]) {
    if (ABL & 64) return;
    const int j = c >> 2, hf = (c >> 1) & 1, hi = c & 1;
    bf[j][2 * hf + hi] = __builtin_bit_cast(unsigned, hi ? __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(f.a[j][2 * ab + hf], RN_H8_SCALE, true)
                                                        : __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(f.a[j][2 * ab + hf], RN_H8_SCALE, false));
    asm volatile("" : "+v"(bf[j][2 * hf + hi]));          // pinned in its gap (the compiler otherwise sinks it to its use)
  };

  f32x16 acc[4][4];
  float acc_db[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // ---- prologue: LA groups requested; groups s0, s0 + 1 visible; the first fragments + MFMA a's converted operands in registers
  for (int s = s0; s < s0 + KBW_LA && s < s1; ++s)
#pragma unroll
    for (int i = 0; i < KBW_NPIECE; ++i) issue_piece(s, s % KBW_NSTG, i);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"((KBW_LA - 2) * KBW_NPIECE) : "memory");
  if (s1 - s0 < KBW_LA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  KbwFrag f0, f1;
  u32x4 bf[2][4];                                         // converted A operands of MFMA a / b, k blocks 0..3
  {
    const unsigned char* st = lds + (unsigned)(s0 % KBW_NSTG) * KBW_STG;
#pragma unroll
    for (int p = 0; p < 12; ++p) read_piece(st, f0, p);
#pragma unroll
    for (int c = 0; c < 16; ++c) conv(f0, 0, c, bf[0]);
  }

  int slot = s0 % KBW_NSTG;
  // one 32-row group: 32 MFMAs on F (in registers), the reads of the next group into Fn, one group's requests, the conversions
  // of MFMA b's operands (first half) and of the NEXT group's MFMA a operands (second half), this wave's share of db
  auto group = [&](KbwFrag& F, KbwFrag& Fn, int s) {
    if (s + KBW_LA - 1 < s1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((KBW_LA - 2) * KBW_NPIECE) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (!(ABL & 4)) __builtin_amdgcn_s_barrier();         // group s + 1 visible; every wave is done with slot (s - 1) % NSTG
    asm volatile("" ::: "memory");
    const int pslot = slot == 0 ? KBW_NSTG - 1 : slot - 1;
    const int nslot = slot + 1 == KBW_NSTG ? 0 : slot + 1;
    const bool do_issue = s + KBW_LA < s1 && !(ABL & 2);
    const unsigned char* stn = lds + (unsigned)nslot * KBW_STG;
    slot = nslot;
    if (ABL & 1) {
      if (do_issue)
#pragma unroll
        for (int i = 0; i < KBW_NPIECE; ++i) issue_piece(s + KBW_LA, pslot, i);
      return;
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < 32; ++g) {
      const int ab = g >> 4, i = (g >> 2) & 3, j = g & 3;
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, F.dz[ab][i]), __builtin_bit_cast(bf16x8, bf[ab][j]), acc[i][j], 0, 0, 0);
      if (g < 12) read_piece(stn, Fn, g);                 // (behind the last group: a slot nobody uses)
      if (g >= 12 && g < 12 + KBW_NPIECE && do_issue) issue_piece(s + KBW_LA, pslot, g - 12);
      if (g < 16) conv(F, 1, g, bf[1]);
      else conv(Fn, 0, g - 16, bf[0]);
      if (g >= 16) {                                      // db: dword (g - 16) % 4 of the fragment of n block (g - 16) / 4
        const int c = g - 16;
        // (written as asm: hipcc 7.2 folds __builtin_amdgcn_fdot2_f32_bf16(bit_cast<bf16x2>(vector[c]), ..) to element 0 of the vector
        // for every c -- one dword added four times -- and an asm statement stays in its gap)
        asm volatile("v_dot2c_f32_bf16 %0, 0x3f803f80, %1" : "+v"(acc_db[c >> 2]) : "v"(F.dz[DBAB][c >> 2][c & 3]));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  for (int s = s0; s < s1; s += 2) {
    group(f0, f1, s);
    group(f1, f0, s + 1);
  }

  // ---- fp32 partial tile part[z][n][k] (a lane holds one k column of 16 feature rows) and this wave's 2 of the split's 4 db rows
  float* pz = jb.part + (long)z * 256 * 256;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int kcol = wk * 128 + j * 32 + n;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int nrow = wn * 128 + i * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
        pz[(long)nrow * 256 + kcol] = acc[i][j][reg];
      }
    }
  float* pd = jb.part_db + ((long)z * 4 + wk * 2 + h) * 256 + wn * 128 + n;
#pragma unroll
  for (int i = 0; i < 4; ++i) pd[i * 32] = acc_db[i];
}

template <bool A8, int ABL = 0>
__global__ __launch_bounds__(KB_NT, KB_OCC) void wgrad_blocked_kernel(KbArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[kb_lds_bytes<A8>()];
  // XCD-aware decode: consecutive ids round-robin over the 8 XCDs; the NB blocks of one unit (a job's row range) share an XCD,
  // and the njobs x Z units are dealt out over the XCDs evenly (unit u -> XCD u % 8; jobs interleaved)
  if constexpr (A8) {
    if ((int)blockIdx.x >= a.grid_q) {                    // wide units: consecutive ids round-robin over the XCDs as they come
      const int v = (int)blockIdx.x - a.grid_q;
      const KbJob& jw = a.job[a.wjob[v % a.nw]];
      if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) & 1) kbw_run<1, ABL>(lds, jw, a.S, v / a.nw);
      else kbw_run<0, ABL>(lds, jw, a.S, v / a.nw);
      return;
    }
  }
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int blk = slot % KB_NB, u = (slot / KB_NB) * 8 + xcd;
  const int job = a.qjob[u % a.nq], z = u / a.nq;
  const KbJob& jb = a.job[job];
  if (z >= jb.Z) return;
  if constexpr (A8) {
    if (jb.gate) {
      kb_run<true, true, ABL>(lds, jb, a.S, jb.Z, z, blk & 1, blk >> 1);
      return;
    }
  }
  kb_run<false, A8, ABL>(lds, jb, a.S, jb.Z, z, blk & 1, blk >> 1);
}

// Ordered reduction of the per-split partials of every job: blocks [0, nbw) of a job reduce dW, the rest db (4 partial rows
// per split).  16 consecutive float4 outputs per workgroup, the Z slabs split 16 ways, combined through LDS in a fixed order.
__global__ __launch_bounds__(256) void wgrad_blocked_reduce_kernel(KbArgs a, int nbw) {
  __shared__ f32x4 red[16][16];
  const KbJob& jb = a.job[blockIdx.y];
  const int o = threadIdx.x & 15, zs = threadIdx.x >> 4;
  const bool is_w = (int)blockIdx.x < nbw;
  const long E4 = is_w ? 256 * 256 / 4 : 256 / 4;
  const int Zr = is_w ? jb.Z : 4 * jb.Z;
  const long g4 = (long)(is_w ? blockIdx.x : blockIdx.x - nbw) * 16 + o;
  const f32x4* src = reinterpret_cast<const f32x4*>(is_w ? jb.part : jb.part_db);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (g4 < E4) {
#pragma unroll 4
    for (int z = zs; z < Zr; z += 16) acc += src[(long)z * E4 + g4];
  }
  red[zs][o] = acc;
  __syncthreads();
  if (zs == 0 && g4 < E4) {
    f32x4 sum = red[0][o];
#pragma unroll
    for (int i = 1; i < 16; ++i) sum += red[i][o];
    float* dst = is_w ? jb.dW : jb.db;
    if (dst) *reinterpret_cast<f32x4*>(dst + g4 * 4) = sum;
  }
}

// Row splits per job: njobs x Z x 4 workgroups ~ one per CU, all jobs of a launch streaming at the same time -- about 64 / njobs
// (three jobs: 21 splits each = 252 workgroups that run ONE ring prologue and write ONE 64-KB partial tile each instead of
// three; 16 MB of partials per step instead of 50).  aligned != 0 (the question-injected layer's backward reads the db partials
// as per-question sums of dZ): a count that never lets a split straddle two questions -- Z = B * d, d | steps per question, as
// close to the target as such a count gets, or B itself -- when one exists within 256 splits.
int kb_total_units() {
  // The row-split budget: 4 x RN_KB_TOTAL workgroups, one per CU.  The launch runs on a side stream beside the latency-bound kernels
  // that close the backward pass (pair reduction, dx / dq, the conv stack's backward); with a workgroup on every CU those kernels
  // wait for CU slots and stretch 3-5x.  Rounds 4-5 (quad units only, launch behind the partial sums): 48 = 192 workgroups, three
  // quarters of the chip (64 -> 0.983 ms, 54 -> 0.955, 48 -> 0.931, 42 -> 0.944, 36 -> 0.976 on the whole step).  Round 6, with
  // the wide units (the launch lost 34 us and left the step's critical path): re-swept on the whole step, alternating on one box --
  // 32 / 36 / 40 / 44 / 48 / 52 / 56 / 64: 96.8 / 96.9 / 97.7 / 95.9 / 96.1 / 97.2 / 91.2 / 91.8 k q/s with the launch where it was,
  // and launched right behind the backward chain (functional.SCHED "wgrad_late": 0 for models without question injection)
  // 32 / 36 / 40 / 44: 99.9 / 99.4 / 100.5 / 92.1 -- 40 = 160 workgroups (profiles/r06_ablations/ab_kb_total_wide*.txt).
#ifndef RN_KB_TOTAL
#define RN_KB_TOTAL 40
#endif
  int total = RN_KB_TOTAL;
  if (const char* e = rn_diag_env("RN_KB_TOTAL")) total = atoi(e) > 0 ? atoi(e) : total;      // (diagnostics builds)
  return total;
}
int kb_splits(int M, int rows_per_question, int njobs, int aligned) {
  const int S = M / 64;
  if (S < 1 || njobs < 1) return 0;
  const int total = kb_total_units();
  const int target = total / njobs > 0 ? total / njobs : 1;
  const int Zd = S >= target ? target : S;
  if (!aligned || rows_per_question <= 0 || rows_per_question % 64 || M % rows_per_question) return Zd;
  const int B = M / rows_per_question, spq = rows_per_question / 64;
  if (B > 256) return Zd;
  if (B >= target) return B;
  int best = B;
  for (int d = 1; d <= spq && B * d <= target; ++d)
    if (spq % d == 0) best = B * d;
  return best;
}
// Row splits of a launch that mixes WIDE jobs (stored gradient x e4m3 image: one workgroup per split) and QUAD jobs (the gate
// job: four workgroups per split) on the same budget of 4 x kb_total_units() workgroups, so that both kinds of workgroup take the
// same time: a wide workgroup does 64 MFMAs per wave on 48 KB per 64 rows, a quad one 18 on 16 KB -- RN_KBW_RATIO_X8 / 8 quad
// rows per wide row (swept on the step).
#ifndef RN_KBW_RATIO_X8
#define RN_KBW_RATIO_X8 24
#endif
void kb_mixed_splits(int M, int nw, int nq, int* Zw, int* Zq) {
  const int S = M / 64, T = 4 * kb_total_units();
  int r8 = RN_KBW_RATIO_X8;
  if (const char* e = rn_diag_env("RN_KBW_RATIO_X8")) r8 = atoi(e) > 0 ? atoi(e) : r8;        // (diagnostics builds)
  // nw Zw + 4 nq Zq = T, Zw = (r8 / 8) Zq
  int zq = nq ? (8 * T + (nw * r8 + 32 * nq) / 2) / (nw * r8 + 32 * nq) : 0;
  if (nq && zq < 1) zq = 1;
  int zw = nw ? (T - 4 * nq * zq) / nw : 0;
  if (nw && zw < 1) zw = 1;
  *Zw = zw > S ? S : zw;
  *Zq = zq > S ? S : zq;
}
// the most row splits any job of a launch can get (workspace sizing: the mix of job kinds is not known to the size query)
int kb_max_units(int M, int rows_per_question, int njobs, int aligned) {
  const int Zu = kb_splits(M, rows_per_question, njobs, aligned);
  if (aligned) return njobs * Zu;
  int most = njobs * Zu;
  for (int nq = 0; nq <= njobs; ++nq) {
    int zw, zq;
    kb_mixed_splits(M, njobs - nq, nq, &zw, &zq);
    const int u = (njobs - nq) * zw + nq * zq;
    most = u > most ? u : most;
  }
  return most;
}
}  // namespace

extern "C" int rn_wgrad_blocked_splits(int M, int rows_per_question, int njobs, int aligned) {
  return (M > 0 && M % 64 == 0 && njobs > 0 && njobs <= KB_MAXJOBS) ? kb_splits(M, rows_per_question, njobs, aligned) : 0;
}

// (diagnostics / tests: the row splits a launch of `nw` wide and `nq` quad jobs gets -- host arithmetic only)
extern "C" int rn_debug_wgrad_blocked_mix(int M, int nw, int nq, int* Zw, int* Zq) {
  RN_CHECK_ARG(M > 0 && M % 64 == 0 && nw >= 0 && nq >= 0 && nw + nq > 0 && nw + nq <= KB_MAXJOBS && Zw && Zq, "rn_debug_wgrad_blocked_mix: bad argument");
  kb_mixed_splits(M, nw, nq, Zw, Zq);
  return 0;
}

size_t rnws_wgrad_blocked(int M, int rows_per_question, int njobs, int aligned) {
  const int Z = rn_wgrad_blocked_splits(M, rows_per_question, njobs, aligned);
  if (Z <= 0) return 0;
  return (size_t)kb_max_units(M, rows_per_question, njobs, aligned) * ((size_t)256 * 256 + (size_t)4 * 256) * sizeof(float);
}

// (aligned launches only -- uniform splits: the db partials of job `job` as per-question column sums)
extern "C" size_t rn_wgrad_blocked_db_partials_offset(int M, int rows_per_question, int njobs, int aligned, int job) {
  const int Z = rn_wgrad_blocked_splits(M, rows_per_question, njobs, aligned);
  if (Z <= 0 || job < 0 || job >= njobs) return 0;
  return ((size_t)njobs * Z * 256 * 256 + (size_t)job * Z * 4 * 256) * sizeof(float);
}

static int kb_launch(const void* const* dZ, const int* dz_dtype, const void* const* A, int a_dtype, const float* dxg, int rows_per_question,
                     int aligned, float* const* dW, float* const* db, int njobs, void* ws, int M, void* stream, int abl) {
  RN_CHECK_ARG(dZ && dz_dtype && A && dW && db && ws && njobs > 0 && njobs <= KB_MAXJOBS, "rn_g_wgrad_blocked: bad pointer / job count (%d, max %d)", njobs, KB_MAXJOBS);
  RN_CHECK_ARG(a_dtype == RN_BF16 || a_dtype == RN_FP8, "rn_g_wgrad_blocked: A must be bf16 or e4m3 (a_dtype=%d)", a_dtype);
  const int Zu = rn_wgrad_blocked_splits(M, rows_per_question, njobs, aligned);
  RN_CHECK_ARG(Zu > 0, "rn_g_wgrad_blocked: needs M %% 64 == 0 (M=%d)", M);
  KbArgs a;
  memset(&a, 0, sizeof(a));
  a.njobs = njobs;
  a.S = M / 64;
  // kinds: a stored gradient on an e4m3 image runs as WIDE units unless the caller reads the db partials per question (aligned)
  bool no_wide = aligned != 0 || a_dtype != RN_FP8;
  if (const char* e = rn_diag_env("RN_KB_NO_WIDE")) no_wide = no_wide || atoi(e) != 0;          // (diagnostics builds)
  for (int j = 0; j < njobs; ++j) {
    RN_CHECK_ARG(dz_dtype[j] == RN_BF16 || dz_dtype[j] == RN_FP8, "rn_g_wgrad_blocked: job %d: dz_dtype must be RN_BF16 (a stored image) or RN_FP8 (gate job) (dz_dtype=%d)", j, dz_dtype[j]);
    a.job[j].wide = (!no_wide && dz_dtype[j] == RN_BF16) ? 1 : 0;
    if (a.job[j].wide) a.wjob[a.nw++] = j;
    else a.qjob[a.nq++] = j;
  }
  int Zw = 0, Zq = Zu;
  if (a.nw) kb_mixed_splits(M, a.nw, a.nq, &Zw, &Zq);
  float* part = (float*)ws;
  size_t units = 0;
  for (int j = 0; j < njobs; ++j) units += a.job[j].wide ? Zw : Zq;
  float* part_db = part + units * 256 * 256;
  size_t u0 = 0;
  for (int j = 0; j < njobs; ++j) {
    RN_CHECK_ARG((dZ[j] || dz_dtype[j] == RN_FP8) && A[j] && dW[j], "rn_g_wgrad_blocked: job %d: dZ / A / dW is NULL", j);
    RN_CHECK_ARG(((uintptr_t)dZ[j] | (uintptr_t)A[j] | (uintptr_t)dW[j] | (uintptr_t)db[j]) % 16 == 0, "rn_g_wgrad_blocked: job %d: pointers must be 16-byte aligned", j);
    RN_CHECK_ARG(dz_dtype[j] != RN_FP8 || !dZ[j] || dZ[j] == A[j], "rn_g_wgrad_blocked: job %d is a gate job: its gate is the sign bits of A (dZ must be NULL or A)", j);
    a.job[j].Z = a.job[j].wide ? Zw : Zq;
    a.job[j].dZ = (const unsigned char*)dZ[j];
    a.job[j].A = (const unsigned char*)A[j];
    a.job[j].part = part + u0 * 256 * 256;
    a.job[j].part_db = part_db + u0 * 4 * 256;
    u0 += a.job[j].Z;
    a.job[j].dW = dW[j];
    a.job[j].db = db[j];
    if (dz_dtype[j] == RN_FP8) {
      RN_CHECK_ARG(a_dtype == RN_FP8, "rn_g_wgrad_blocked: job %d: a gate job needs an e4m3 A image", j);
      a.job[j].dZ = (const unsigned char*)A[j];
      RN_CHECK_ARG(dxg && (uintptr_t)dxg % 16 == 0, "rn_g_wgrad_blocked: job %d is a gate job: needs the 16-byte aligned dxg", j);
      RN_CHECK_ARG(rows_per_question > 0 && rows_per_question % 64 == 0 && M % rows_per_question == 0,
                   "rn_g_wgrad_blocked: gate job: rows_per_question=%d must be a multiple of 64 dividing M", rows_per_question);
      a.job[j].dxg = dxg;
      a.job[j].steps_per_q = rows_per_question / 64;
      a.job[j].gate = 1;
    }
  }
  hipStream_t s = (hipStream_t)stream;
  a.grid_q = a.nq ? 8 * KB_NB * cdiv(a.nq * Zq, 8) : 0;
  if (!a.nq) a.nq = 1;                                        // (never indexed: every workgroup is a wide one; no division by zero)
  if (!a.nw) a.nw = 1;
  int nwide_wg = 0;
  for (int j = 0; j < njobs; ++j) nwide_wg += a.job[j].wide ? a.job[j].Z : 0;
  const int grid_all = a.grid_q + nwide_wg;
#ifdef RN_DIAG
  switch (abl) {
#define RN_ABL(v) case v: if (a_dtype == RN_FP8) wgrad_blocked_kernel<true, v><<<grid_all, KB_NT, 0, s>>>(a); else wgrad_blocked_kernel<false, v><<<grid_all, KB_NT, 0, s>>>(a); break;
    RN_ABL(257) RN_ABL(513) RN_ABL(1) RN_ABL(2) RN_ABL(3) RN_ABL(4) RN_ABL(6) RN_ABL(66) RN_ABL(8) RN_ABL(24) RN_ABL(10) RN_ABL(26)
#undef RN_ABL
    default:
#else
  (void)abl;
  {
#endif
#ifdef KB_FORCE_ABL                                         // (variant builds: what would the step cost with this launch ablated?)
    if (a_dtype == RN_FP8) wgrad_blocked_kernel<true, KB_FORCE_ABL><<<grid_all, KB_NT, 0, s>>>(a);
    else
#endif
    if (a_dtype == RN_FP8) wgrad_blocked_kernel<true><<<grid_all, KB_NT, 0, s>>>(a);
    else wgrad_blocked_kernel<false><<<grid_all, KB_NT, 0, s>>>(a);
  }
  RN_LAUNCH_CHECK("rn_g_wgrad_blocked");
  const int nbw = cdiv(256 * 256 / 4, 16), nbb = cdiv(256 / 4, 16);
  wgrad_blocked_reduce_kernel<<<dim3(nbw + nbb, njobs), 256, 0, s>>>(a, nbw);
  RN_LAUNCH_CHECK("rn_g_wgrad_blocked(reduce)");
  return 0;
}

extern "C" int rn_g_wgrad_blocked(const void* const* dZ, const int* dz_dtype, const void* const* A, int a_dtype, const float* dxg,
                                  int rows_per_question, int aligned, float* const* dW, float* const* db, int njobs, void* ws, int M, void* stream) {
  return kb_launch(dZ, dz_dtype, A, a_dtype, dxg, rows_per_question, aligned, dW, db, njobs, ws, M, stream, 0);
}
#ifdef RN_DIAG
extern "C" int rn_diag_wgrad_blocked(const void* const* dZ, const int* dz_dtype, const void* const* A, int a_dtype, const float* dxg,
                                     int rows_per_question, int aligned, float* const* dW, float* const* db, int njobs, void* ws, int M, void* stream, int abl) {
  return kb_launch(dZ, dz_dtype, A, a_dtype, dxg, rows_per_question, aligned, dW, db, njobs, ws, M, stream, abl);
}
#endif

// The ReLU gate of the last g layer merged into the SIGN BITS of an e4m3 row-blocked image (tests / tools: the forward chain
// writes H_2 that way itself): the forward kernel's layer-3 lane masks (un-swapped epilogue: per 32-row block and 32-feature block
// 32 dwords, dword 2 (4 (r / 8) + r % 4) + (r / 4) % 2 = the 32 feature bits of row r) -> bit 7 of byte (row, feature).  A
// workgroup = one 32-row block, a thread = one feature: 32 bit tests, two 16-byte read-modify-writes.
__global__ __launch_bounds__(256) void relu_gate_image_kernel(const unsigned* __restrict__ mask, unsigned char* __restrict__ img) {
  const long wt = blockIdx.x;
  const int f = threadIdx.x, fb = f & 31;
  const u32x4* mp = reinterpret_cast<const u32x4*>(mask + (wt * 8 + (f >> 5)) * 32);
  u32x4 m[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) m[c] = mp[c];
#pragma unroll
  for (int hf = 0; hf < 2; ++hf) {                        // rows 16 hf .. 16 hf + 15 = dwords 16 hf .. 16 hf + 15
    u32x4 o;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      unsigned v = 0u;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = 4 * p + e, d = 16 * hf + 8 * (r >> 3) + 2 * (r & 3) + ((r & 7) >> 2);
        const unsigned tbit = (unsigned)__builtin_amdgcn_sbfe((int)m[d >> 2][d & 3], fb, 1);
        v |= tbit & (0x80u << (8 * e));
      }
      o[p] = v;
    }
    u32x4* cell = reinterpret_cast<u32x4*>(img + ((2 * wt + hf) * 256 + f) * 16);
    const u32x4 old = *cell;
#pragma unroll
    for (int p = 0; p < 4; ++p) o[p] |= old[p] & 0x7f7f7f7fu;
    *cell = o;
  }
}

extern "C" int rn_relu_gate_image(const void* mask, void* img, int M, void* stream) {
  RN_CHECK_ARG(mask && img && M > 0 && M % 32 == 0 && ((uintptr_t)mask | (uintptr_t)img) % 16 == 0, "rn_relu_gate_image: needs 16-byte aligned buffers and M %% 32 == 0 (M=%d)", M);
  relu_gate_image_kernel<<<M / 32, 256, 0, (hipStream_t)stream>>>((const unsigned*)mask, (unsigned char*)img);
  RN_LAUNCH_CHECK("rn_relu_gate_image");
  return 0;
}

// Health of an e4m3 activation copy (rn_fp8_copy_health): how much of a layer's post-ReLU activation does NOT survive as an e4m3
// byte.  The copies use a fixed scale of 1 (rn_common.h): values below 2^-10 flush to zero, values above 448 are clamped.  The
// forward chain's lane masks say which elements were positive BEFORE rounding (swapped layers 0..2: word 4 j + r of a block, lane
// n + 32 h -> row n, feature 8 j + 4 h + r), so a set gate bit over a zero byte is a flushed element.  A workgroup = one 32-row
// block, a thread = one feature; integer counters (order-independent), accumulated with atomics:
//   out[0] = positive elements (gate bits), out[1] = ... whose byte is 0, out[2] = bytes at the clamp (0x7e = 448), out[3] = largest byte.
__global__ __launch_bounds__(256) void fp8_copy_health_kernel(const unsigned* __restrict__ mask, const unsigned char* __restrict__ img,
                                                              unsigned long long* __restrict__ out) {
  __shared__ unsigned red[4][4];
  const long wt = blockIdx.x;
  const int f = threadIdx.x, fb = f & 31;
  const unsigned rowbits = mask[(wt * 8 + (f >> 5)) * 32 + 2 * (4 * (fb >> 3) + (fb & 3)) + ((fb >> 2) & 1)];   // bit n = row n of feature f
  unsigned pos = 0, flushed = 0, sat = 0, mx = 0;
#pragma unroll
  for (int hf = 0; hf < 2; ++hf) {
    const u32x4 cell = *reinterpret_cast<const u32x4*>(img + ((2 * wt + hf) * 256 + f) * 16);     // rows 16 hf .. + 15 of feature f
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const unsigned byte = (cell[r >> 2] >> (8 * (r & 3))) & 0x7fu, bit = (rowbits >> (16 * hf + r)) & 1u;   // (bit 7 of H_2: the next layer's gate)
      pos += bit;
      flushed += bit & (byte == 0u ? 1u : 0u);
      sat += byte == 0x7eu ? 1u : 0u;
      mx = byte > mx ? byte : mx;
    }
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    pos += __shfl_xor(pos, o); flushed += __shfl_xor(flushed, o); sat += __shfl_xor(sat, o);
    const unsigned m2 = __shfl_xor(mx, o); mx = m2 > mx ? m2 : mx;
  }
  if ((threadIdx.x & 63) == 0) { const int w = threadIdx.x >> 6; red[0][w] = pos; red[1][w] = flushed; red[2][w] = sat; red[3][w] = mx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(out + 0, (unsigned long long)(red[0][0] + red[0][1] + red[0][2] + red[0][3]));
    atomicAdd(out + 1, (unsigned long long)(red[1][0] + red[1][1] + red[1][2] + red[1][3]));
    atomicAdd(out + 2, (unsigned long long)(red[2][0] + red[2][1] + red[2][2] + red[2][3]));
    unsigned m4 = red[3][0]; for (int w = 1; w < 4; ++w) m4 = red[3][w] > m4 ? red[3][w] : m4;
    atomicMax(out + 3, (unsigned long long)m4);
  }
}

extern "C" int rn_fp8_copy_health(const void* mask, const void* img, unsigned long long* out4, int M, void* stream) {
  RN_CHECK_ARG(mask && img && out4 && M > 0 && M % 32 == 0 && ((uintptr_t)mask | (uintptr_t)img) % 16 == 0 && (uintptr_t)out4 % 8 == 0,
               "rn_fp8_copy_health: needs 16-byte aligned buffers and M %% 32 == 0 (M=%d)", M);
  fp8_copy_health_kernel<<<M / 32, 256, 0, (hipStream_t)stream>>>((const unsigned*)mask, (const unsigned char*)img, out4);
  RN_LAUNCH_CHECK("rn_fp8_copy_health");
  return 0;
}

// Row-blocked image of a row-major (M, 256) matrix (tests / tools; the chains write the images themselves).
//   src_dtype RN_BF16: 16-bit elements, 8-row blocks;  RN_FP8: bytes, 16-row blocks.  M % 16 == 0.
__global__ __launch_bounds__(256) void to_blocked_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst, long M, int es, int back) {
  const long g = (long)blockIdx.x * 256 + threadIdx.x;    // one element
  if (g >= M * 256) return;
  const int rb = es == 2 ? 8 : 16;
  const long m = g / 256;
  const int f = (int)(g - m * 256);
  const long blocked = ((m / rb) * 256 + f) * rb + m % rb;
  const long from = back ? blocked : g, to = back ? g : blocked;
  if (es == 2) reinterpret_cast<unsigned short*>(dst)[to] = reinterpret_cast<const unsigned short*>(src)[from];
  else dst[to] = src[from];
}

extern "C" int rn_rows_to_blocked(const void* src, void* dst, int dtype, int M, int back, void* stream) {
  RN_CHECK_ARG(src && dst && M > 0 && M % 16 == 0 && (dtype == RN_BF16 || dtype == RN_FP8), "rn_rows_to_blocked: needs M %% 16 == 0 and bf16 / e4m3 (M=%d dtype=%d)", M, dtype);
  to_blocked_kernel<<<cdiv((long)M * 256, 256), 256, 0, (hipStream_t)stream>>>((const unsigned char*)src, (unsigned char*)dst, M, dtype == RN_BF16 ? 2 : 1, back);
  RN_LAUNCH_CHECK("rn_rows_to_blocked");
  return 0;
}

// Per-question column sums of a 16-bit row-blocked image: Rq[b, f] = sum over the rows of question b of dZ[., f] -- what the
// question-injected layer's backward needs when the weight-gradient kernel's row splits straddle questions.  One workgroup per
// (question, 64 features): 4 phases of row blocks per feature, combined in a fixed order.
__global__ __launch_bounds__(256) void blocked_question_sums_kernel(const unsigned char* __restrict__ img, float* __restrict__ Rq, int blocks_per_q) {
  __shared__ float red[4][64];
  const int b = blockIdx.x >> 2, f = (blockIdx.x & 3) * 64 + (threadIdx.x & 63), ph = threadIdx.x >> 6;
  float acc = 0.f;
  // a pass over 134 MB on the weight-gradient stream, IN FRONT of that launch (ir-*): one 16-byte load in flight per thread ran it
  // at 1.5 TB/s (90 us at the headline shape, the weight gradient started that much later); sixteen in flight, same add order
  const unsigned char* src = img + (((long)b * blocks_per_q + ph) * 256 + f) * 16;
  constexpr int QS_U = 16;
  int rb = ph;
  for (; rb + 4 * (QS_U - 1) < blocks_per_q; rb += 4 * QS_U) {
    u32x4 v[QS_U];
#pragma unroll
    for (int u = 0; u < QS_U; ++u) v[u] = *reinterpret_cast<const u32x4*>(src + (long)u * 4 * 256 * 16);
#pragma unroll
    for (int u = 0; u < QS_U; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc += __builtin_bit_cast(float, v[u][e] << 16) + __builtin_bit_cast(float, v[u][e] & 0xffff0000u);
    src += (long)QS_U * 4 * 256 * 16;
  }
  for (; rb < blocks_per_q; rb += 4) {
    const u32x4 v = *reinterpret_cast<const u32x4*>(src);
#pragma unroll
    for (int e = 0; e < 4; ++e) acc += __builtin_bit_cast(float, v[e] << 16) + __builtin_bit_cast(float, v[e] & 0xffff0000u);
    src += 4 * 256 * 16;
  }
  red[ph][threadIdx.x & 63] = acc;
  __syncthreads();
  if (ph == 0) Rq[(long)b * 256 + f] = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}

extern "C" int rn_blocked_question_sums(const void* img, float* Rq, int M, int rows_per_question, void* stream) {
  RN_CHECK_ARG(img && Rq && M > 0 && rows_per_question > 0 && rows_per_question % 8 == 0 && M % rows_per_question == 0,
               "rn_blocked_question_sums: needs rows_per_question %% 8 == 0 dividing M (M=%d, rows_per_question=%d)", M, rows_per_question);
  blocked_question_sums_kernel<<<(M / rows_per_question) * 4, 256, 0, (hipStream_t)stream>>>((const unsigned char*)img, Rq, rows_per_question / 8);
  RN_LAUNCH_CHECK("rn_blocked_question_sums");
  return 0;
}
