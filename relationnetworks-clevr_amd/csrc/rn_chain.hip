// Fused g_theta chains (model.py:130-152 and their autograd): persistent workgroups walk the
// 128-row tiles of the pair matrix; ALL g layers run on a tile while its 256-wide activation stays
// in LDS.
//
//   forward  (MODE_FWD): tile <- P rows;            per layer: tile = relu(tile @ W_l^T + b_l)
//            each activation is written to HBM exactly once (for the backward pass) and never read
//            back; the pair sum (model.py:151-152) is taken from the last tile while it is on chip.
//   backward (MODE_BWD): tile <- dxg[b] * (H_L > 0);  per layer: tile = (tile @ W_l[:, :256]) * (H_{l-1} > 0)
//            i.e. pair-sum broadcast + last ReLU gate + the whole dgrad chain; every dZ_l is written
//            once (wgrad reads it), every H_l is read once (as the gate).
//
// Arithmetic (template PREC):
//   PREC_BF16: bf16 tile x bf16 weights, v_mfma_f32_32x32x16_bf16, 64-wide K slabs;
//   PREC_F16S: fp16 tile x fp16 (hi + lo) split weights, two v_mfma_f32_32x32x16_f16 per product,
//              32-wide K slabs (hi and lo images side by side); forward only.  Removes the weight
//              rounding error that keeps single-pass bf16 at ~1e-2 of the fp32 reference.
//
// One workgroup per CU (LDS-limited): 512 threads, 8 waves (2 per SIMD) in a 2 x 4 grid, 64 x 64 per
// wave = 2 x 2 MFMA tiles, 64 accumulators.  Operands are swapped (weights = MFMA A-operand) exactly
// as in rn_gemm.hip, so a lane ends up with 4 consecutive features of one pair row -> one
// ds_write_b64 into the LDS tile; the tile is copied LDS -> HBM with 16-byte, row-contiguous stores.
//
// Weights (<= 128 KB per layer) stream from L2 in K slabs into a 2-deep LDS ring:
//   GLDS = true : LDS-DMA (global_load_lds_dwordx4): no VGPR staging, no ds_write; the slab image is
//                 linear in LDS, bank conflicts are avoided by an XOR swizzle applied to the per-lane
//                 SOURCE address and to the fragment read;
//   GLDS = false: register-staged prefetch into a padded image (row stride 144 B; PREC_BF16 only).
// The end-of-slab barrier waits with a COUNTED vmcnt (only the slab's own loads; younger tile stores
// and the next tile's source prefetch stay in flight) and a raw s_barrier.
// The next tile's source rows (P / H_L) are prefetched into registers during the last slab.
#include "rn_common.h"

namespace {
constexpr int CT_G = 256, CT_MAXL = 8, CT_TM = 128, CT_NT = 512;
constexpr int ACT_RS = CT_G * 2 + 16;        // 528 B: tile row stride (conflict-free b128 reads)
constexpr int ACT_BYTES = CT_TM * ACT_RS;    // 67584
enum { MODE_FWD = 0, MODE_BWD = 1 };
enum { PREC_BF16 = 0, PREC_F16S = 1 };

struct ChainArgs {
  const void* W[CT_MAXL];                    // fwd: packed (256, K[l]) (f16s: hi);  bwd: transposed (256 kin, 256 n) of step s
  const void* Wlo[CT_MAXL];                  // f16s: lo halves
  const float* bias[CT_MAXL];                // fwd only
  const bf16* gate[CT_MAXL];                 // bwd only: activation gating the output of step s, (M, 256)
  bf16* out[CT_MAXL];                        // fwd: H_l as bf16 (may be null);  bwd: dZ after step s
  int K[CT_MAXL];                            // reduction length of step l (multiple of 64, <= 256)
  // backward prologue
  const bf16* HL;                            // last activation (M, 256)
  const float* dxg;                          // (B, 256) fp32
  bf16* out0;                                // dZ of the last layer (M, 256)
  int rows_per_b;                            // n*n
};

template <int PREC> struct Prec;
template <> struct Prec<PREC_BF16> {
  typedef bf16 T;
  typedef bf16x8 Frag;
  typedef bf16x4 Quad;
  static __device__ __forceinline__ void mma(const Frag& a, const Frag& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct Prec<PREC_F16S> {
  typedef f16 T;
  typedef f16x8 Frag;
  typedef f16x4 Quad;
  static __device__ __forceinline__ void mma(const Frag& a, const Frag& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
};
}  // namespace

template <int MODE, int PREC, bool GLDS, bool PREFETCH, int RDEPTH>
__global__ __launch_bounds__(CT_NT) void g_chain_kernel(const void* __restrict__ Pv, int ldp, ChainArgs a, int L,
                                                        float* __restrict__ xg_part, int ntiles,
                                                        unsigned long long* __restrict__ trace) {
  typedef typename Prec<PREC>::T T;                         // element type of P, the LDS tile and the weights
  typedef typename Prec<PREC>::Frag Frag;
  typedef typename Prec<PREC>::Quad Quad;
  static_assert(!(PREC == PREC_F16S && (MODE != MODE_FWD || !GLDS)), "f16s: forward, LDS-DMA only");
  constexpr int TM = CT_TM, NT = CT_NT, NW = NT / 64;
  constexpr int NPASS = (PREC == PREC_F16S) ? 2 : 1;        // weight images per slab (hi, lo)
  constexpr int BK = 64 / NPASS;                            // K-slab width: 64 (bf16) / 32 (f16s); 32 KB per stage either way
  constexpr int KS = BK / 16;                               // K16 steps per slab (4 / 2)
  constexpr int RD = RDEPTH < KS ? RDEPTH : KS;             // K16 steps of fragments read ahead per batch
  constexpr int NST = TM * 32 / NT;                         // 16-byte chunks per thread of a full tile (8)
  constexpr int IPW = 32 / NW;                              // LDS-DMA instructions per wave per slab (4)
  constexpr int W_RS = GLDS ? BK * 2 : BK * 2 + 16;        // weight slab row stride: linear / padded
  constexpr int IMG_BYTES = CT_G * W_RS;                    // one weight image (256 rows)
  constexpr int WBUF_BYTES = NPASS * IMG_BYTES;             // one ring stage
  constexpr int CPR = BK / 8;                               // 16-byte chunks per slab row (8 / 4)
  constexpr int RPI = 64 / CPR;                             // slab rows covered by one 1-KB LDS-DMA instruction (8 / 16)
  constexpr int BIAS_BYTES = CT_MAXL / 2 * CT_G * 4;        // up to 4 layers of bias (fwd)
  constexpr int TAB_BYTES = 6 * CT_MAXL * 8;               // per-layer pointer / size table (see below)
  __shared__ __attribute__((aligned(16))) unsigned char lds[ACT_BYTES + 2 * WBUF_BYTES + BIAS_BYTES + CT_G * 4 + TAB_BYTES];
  unsigned char* act = lds;
  unsigned char* wbuf = lds + ACT_BYTES;
  float* bias_s = reinterpret_cast<float*>(lds + ACT_BYTES + 2 * WBUF_BYTES);
  float* red = bias_s + CT_MAXL / 2 * CT_G;
  // The per-layer arguments are indexed with a run-time layer number.  Straight from the kernel-argument
  // struct that becomes a VMEM load + s_waitcnt vmcnt(0) in front of every weight slab (which drains the
  // whole in-order VM queue); a copy in LDS is read with ds_read (lgkmcnt) and made scalar again.
  unsigned long long* tab = reinterpret_cast<unsigned long long*>(red + CT_G);
  enum { T_W = 0, T_WLO = 1, T_BIAS = 2, T_GATE = 3, T_OUT = 4, T_K = 5 };
  if (threadIdx.x < CT_MAXL) {
    const int i = threadIdx.x;
    tab[T_W * CT_MAXL + i] = (unsigned long long)a.W[i];
    tab[T_WLO * CT_MAXL + i] = (unsigned long long)a.Wlo[i];
    tab[T_BIAS * CT_MAXL + i] = (unsigned long long)a.bias[i];
    tab[T_GATE * CT_MAXL + i] = (unsigned long long)a.gate[i];
    tab[T_OUT * CT_MAXL + i] = (unsigned long long)a.out[i];
    tab[T_K * CT_MAXL + i] = (unsigned long long)a.K[i];
  }
  __syncthreads();
  auto tget = [&](int what, int l) -> unsigned long long {       // wave-uniform table read -> SGPR pair
    const unsigned long long v = tab[what * CT_MAXL + l];
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
  };
  const T* P = static_cast<const T*>(Pv);

  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int wu = __builtin_amdgcn_readfirstlane(w);         // provably wave-uniform copy for the LDS-DMA base
  const int wm = w & 1, wn = w >> 1;
  // swizzle of the linear weight image: 64-B rows need (row >> 2) & 3, 128-B rows (row >> 1) & 7
  auto swz = [](int row) { return CPR == 8 ? ((row >> 1) & 7) : ((row >> 2) & 3); };
  // optional phase timestamps (s_memtime) of wave 0 for the SECOND tile of a few workgroups: tools/trace_chain.py
  int tp = 0;
  bool tracing = false;
  auto stamp = [&]() {
    if (tracing) trace[(blockIdx.x / 50) * 32 + (tp++)] = __builtin_amdgcn_s_memtime();
  };

  const bool bias_in_lds = (MODE == MODE_FWD) && L <= CT_MAXL / 2;
  if (bias_in_lds)
    for (int c = t; c < L * CT_G; c += NT) bias_s[c] = reinterpret_cast<const float*>(tget(T_BIAS, c >> 8))[c & 255];

  // ---- tile source prefetch (registers): fwd = P rows (K0 columns), bwd = H_L rows (256 columns)
  u32x4 rp[NST];
  const int cpr0 = (MODE == MODE_FWD) ? ((int)tget(T_K, 0) >> 3) : 32;          // 16-byte chunks per source row
  auto prefetch_tile = [&](long m0n) {
#pragma unroll
    for (int i = 0; i < NST; ++i) {                           // always NST loads per thread (the slab barrier counts them)
      int c = t + NT * i;
      c = c < TM * cpr0 ? c : TM * cpr0 - 1;
      const int r = c / cpr0, cc = c - r * cpr0;
      if constexpr (MODE == MODE_FWD) rp[i] = *reinterpret_cast<const u32x4*>(P + (m0n + r) * ldp + cc * 8);
      else rp[i] = *reinterpret_cast<const u32x4*>(a.HL + (m0n + r) * CT_G + cc * 8);
    }
  };
  // Barrier at the end of a K slab.  LDS-DMA path: wait only for this slab's weight loads -- vmcnt retires
  // in order, `younger` = VM operations issued after them (tile stores, next-tile prefetch) that may stay
  // in flight -- plus all LDS traffic, then a raw s_barrier (a __syncthreads() would drain vmcnt to 0).
  auto slab_barrier = [&](int younger) {
    if constexpr (GLDS) {
      if (younger < 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // no weight load pending
      else if (younger >= 2 * NST) asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");
      else if (younger >= NST) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    } else {
      __syncthreads();
    }
  };
  static_assert(NST == 8, "slab_barrier's vmcnt immediates assume 8 chunks per thread");
  auto stage_tile = [&](long m0) {
    if constexpr (!PREFETCH) prefetch_tile(m0);               // no register prefetch: load the rows right here
#pragma unroll
    for (int i = 0; i < NST; ++i) {
      const int c = t + NT * i;
      if (c < TM * cpr0) {
        const int r = c / cpr0, cc = c - r * cpr0;
        if constexpr (MODE == MODE_FWD) {
          *reinterpret_cast<u32x4*>(act + r * ACT_RS + cc * 16) = rp[i];
        } else {
          // dZ_L = dxg[b] * (H_L > 0): backward of the pair sum + last ReLU; also stored to HBM for wgrad
          union { u32x4 u; bf16x8 h; } hv;
          hv.u = rp[i];
          const float* gb = a.dxg + ((m0 + r) / a.rows_per_b) * CT_G;     // question of THIS row (tiles may straddle)
          const f32x4 g0 = *reinterpret_cast<const f32x4*>(gb + cc * 8);
          const f32x4 g1 = *reinterpret_cast<const f32x4*>(gb + cc * 8 + 4);
          bf16x8 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            o[e] = (float)hv.h[e] > 0.f ? (bf16)g0[e] : (bf16)0.f;
            o[e + 4] = (float)hv.h[e + 4] > 0.f ? (bf16)g1[e] : (bf16)0.f;
          }
          *reinterpret_cast<bf16x8*>(act + r * ACT_RS + cc * 16) = o;
          *reinterpret_cast<bf16x8*>(a.out0 + (m0 + r) * CT_G + cc * 8) = o;
        }
      }
    }
  };

  // ---- weight slab loaders
  const int srow = t >> 3, scc = t & 7;                     // register path: 8 lanes cover one 128-byte row slab
  u32x4 rw[GLDS ? 1 : 4];
  auto w_issue = [&](int l, int slab, int buf) {
    const int ldw = (int)tget(T_K, l);
    const T* Whi_l = reinterpret_cast<const T*>(tget(T_W, l));
    const T* Wlo_l = NPASS == 2 ? reinterpret_cast<const T*>(tget(T_WLO, l)) : Whi_l;
    if constexpr (GLDS) {
      // LDS-DMA instruction q = IPW*wave + s of a stage fills bytes [q*1024, +1024): image q / (32/NPASS), slab
      // rows RPI*(q % ..) .. ; lane i lands at row + i / CPR, chunk position i % CPR and must therefore
      // FETCH chunk (i % CPR) ^ swz(row)
#pragma unroll
      for (int s = 0; s < IPW; ++s) {
        const int q = IPW * w + s;
        const int img = q / (32 / NPASS), qi = q % (32 / NPASS);
        const int row = qi * RPI + lane / CPR;
        const int chunk = (lane % CPR) ^ swz(row);
        const T* Wl = img == 0 ? Whi_l : Wlo_l;
        const T* g = Wl + (long)row * ldw + slab * BK + chunk * 8;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                         (__attribute__((address_space(3))) void*)(wbuf + buf * WBUF_BYTES + (IPW * wu + s) * 1024),
                                         16, 0, 0);
      }
    } else {
      const T* Wl = Whi_l;
#pragma unroll
      for (int s = 0; s < 4; ++s)
        rw[s] = *reinterpret_cast<const u32x4*>(Wl + (long)(srow + 64 * s) * ldw + slab * BK + scc * 8);
    }
  };
  auto w_commit = [&](int buf) {                             // register path only: regs -> padded LDS image
    if constexpr (!GLDS) {
#pragma unroll
      for (int s = 0; s < 4; ++s)
        *reinterpret_cast<u32x4*>(wbuf + buf * WBUF_BYTES + (srow + 64 * s) * W_RS + scc * 16) = rw[s];
    }
  };

  // fragment addressing
  const unsigned char* fa_base = act + (wm * 64 + (lane & 31)) * ACT_RS + (lane >> 5) * 16;
  int fw_row_off[2], fw_swz[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int r = wn * 64 + nt * 32 + (lane & 31);
    fw_row_off[nt] = r * W_RS;
    fw_swz[nt] = GLDS ? swz(r) : 0;
  }
  auto copy_out = [&](bf16* Ol, long m0) {                   // LDS tile -> HBM (bf16), 16-byte chunks, row-contiguous
    if (Ol) {
#pragma unroll
      for (int i = 0; i < NST; ++i) {
        const int c = t + NT * i;
        const int r = c >> 5, cc = c & 31;
        if constexpr (PREC == PREC_F16S) {                   // the backward pass is bf16: convert the fp16 tile on the way out
          const f16x8 v = *reinterpret_cast<const f16x8*>(act + r * ACT_RS + cc * 16);
          bf16x8 o;
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = (bf16)(float)v[e];
          *reinterpret_cast<bf16x8*>(Ol + (m0 + r) * CT_G + cc * 8) = o;
        } else {
          *reinterpret_cast<u32x4*>(Ol + (m0 + r) * CT_G + cc * 8) = *reinterpret_cast<const u32x4*>(act + r * ACT_RS + cc * 16);
        }
      }
    }
  };

  int tile = blockIdx.x;
  if (tile >= ntiles) return;
  if constexpr (PREFETCH) prefetch_tile((long)tile * TM);
  w_issue(0, 0, 0);
  w_commit(0);
  int cur = 0;
  for (; tile < ntiles; tile += gridDim.x) {
    const long m0 = (long)tile * TM;
    const bool has_next_tile = tile + (int)gridDim.x < ntiles;
    tracing = trace != nullptr && t == 0 && (blockIdx.x % 50) == 0 && tile == (int)(blockIdx.x + gridDim.x);
    stamp();
    stage_tile(m0);
    __syncthreads();                    // tile, bias and weight slab `cur` visible (drains the LDS-DMA too)
    stamp();
    for (int l = 0; l < L; ++l) {
      f32x16 acc[2][2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
      // backward: fetch this step's ReLU gate (the lane's groups of 4 features) early, use it in the epilogue
      u32x2 gt[2][2][4];
      if constexpr (MODE == MODE_BWD) {
        const bf16* gl = reinterpret_cast<const bf16*>(tget(T_GATE, l));
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g)
              gt[mt][nt][g] = *reinterpret_cast<const u32x2*>(
                  gl + (m0 + wm * 64 + mt * 32 + (lane & 31)) * CT_G + wn * 64 + nt * 32 + 8 * g + 4 * (lane >> 5));
      }
      const int ns = (int)tget(T_K, l) / BK;
      for (int s = 0; s < ns; ++s) {
        const bool last_slab = (s == ns - 1);
        const bool last_of_tile = last_slab && l == L - 1;
        const bool has_next = !last_of_tile || has_next_tile;
        if (has_next) {
          if (last_of_tile) w_issue(0, 0, cur ^ 1);                           // next tile, first slab
          else if (last_slab) w_issue(l + 1, 0, cur ^ 1);
          else w_issue(l, s + 1, cur ^ 1);
        }
        int younger = 0;
        if (s == 0 && l > 0) {
          // stores go AFTER the weight loads: vmcnt retires in order
          bf16* prev_out = reinterpret_cast<bf16*>(tget(T_OUT, l - 1));
          copy_out(prev_out, m0);                     // previous layer's tile (intact in LDS until this layer's epilogue)
          if (prev_out) younger += NST;
        }
        if (PREFETCH && last_of_tile && has_next_tile) {
          // next tile's source rows: issued in the LAST slab, after the last weight load of this tile, so no
          // later wait has to drain them (in-order vmcnt); they land under the epilogue / copy-out / pair sum
          prefetch_tile((long)(tile + gridDim.x) * TM);
          younger += NST;
        }
        const unsigned char* wb = wbuf + cur * WBUF_BYTES + (GLDS ? 0 : (lane >> 5) * 16);
        // Fragment reads are issued in batches of RD K16 steps AHEAD of the MFMAs that consume them (the
        // compiler otherwise reads 2-4 fragments, waits lgkmcnt(0), issues 2 MFMAs, ... and every wait exposes
        // a full LDS round trip while the partner wave of the SIMD sits in the same phase).
#pragma unroll
        for (int kb = 0; kb < KS; kb += RD) {
          Frag fa[RD][2], fw[RD][NPASS][2];
#pragma unroll
          for (int k2 = 0; k2 < RD; ++k2) {
            const int ks = kb + k2;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
              fa[k2][mt] = *reinterpret_cast<const Frag*>(fa_base + mt * 32 * ACT_RS + s * (2 * BK) + ks * 32);
#pragma unroll
            for (int p = 0; p < NPASS; ++p)
#pragma unroll
              for (int nt = 0; nt < 2; ++nt) {
                if constexpr (GLDS)
                  fw[k2][p][nt] = *reinterpret_cast<const Frag*>(wb + p * IMG_BYTES + fw_row_off[nt] +
                                                                 (((2 * ks + (lane >> 5)) ^ fw_swz[nt]) << 4));
                else
                  fw[k2][p][nt] = *reinterpret_cast<const Frag*>(wb + fw_row_off[nt] + ks * 32);
              }
          }
          __builtin_amdgcn_sched_barrier(0);        // keep the whole batch of ds_reads above the MFMAs
#pragma unroll
          for (int k2 = 0; k2 < RD; ++k2)
#pragma unroll
            for (int p = 0; p < NPASS; ++p)
#pragma unroll
              for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) Prec<PREC>::mma(fw[k2][p][nt], fa[k2][mt], acc[mt][nt]);
        }
        if (has_next) w_commit(cur ^ 1);
        slab_barrier(has_next ? younger : -1);       // (A) all reads of wbuf[cur] / this tile slab done; next slab visible
        cur ^= 1;
      }
      stamp();
      // ---- epilogue -> T -> tile in place (all waves are past barrier A)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int row = wm * 64 + mt * 32 + (lane & 31);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int nb = wn * 64 + nt * 32 + 8 * g + 4 * (lane >> 5);
            Quad o;
            if constexpr (MODE == MODE_FWD) {
              const f32x4 bv = bias_in_lds ? *reinterpret_cast<const f32x4*>(bias_s + l * CT_G + nb)
                                           : *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(tget(T_BIAS, l)) + nb);
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                float v = fmaxf(acc[mt][nt][4 * g + r] + bv[r], 0.f);
                if constexpr (PREC == PREC_F16S) v = fminf(v, 65504.f);       // saturate instead of overflowing fp16 to inf
                o[r] = (T)v;
              }
            } else {
              union { u32x2 u; bf16x4 h; } gv;
              gv.u = gt[mt][nt][g];
#pragma unroll
              for (int r = 0; r < 4; ++r) o[r] = (float)gv.h[r] > 0.f ? (T)acc[mt][nt][4 * g + r] : (T)0.f;
            }
            *reinterpret_cast<Quad*>(act + row * ACT_RS + nb * 2) = o;
          }
        }
      }
      __syncthreads();                    // (B) the new tile is visible
      stamp();
    }
    copy_out(reinterpret_cast<bf16*>(tget(T_OUT, L - 1)), m0);
    // ---- forward: pair-sum partial of this tile = column sums of the stored tile (fp32, fixed order)
    if (MODE == MODE_FWD && xg_part) {
      const int c = t & 255, h = t >> 8;
      float s = 0.f;
#pragma unroll 8
      for (int r = 0; r < 64; ++r) s += (float)*reinterpret_cast<const T*>(act + (h * 64 + r) * ACT_RS + c * 2);
      if (h == 1) red[c] = s;
      __syncthreads();
      if (h == 0) xg_part[(long)tile * CT_G + c] = s + red[c];
    }
    stamp();
    __syncthreads();                      // (C) every reader of the tile is done before the next tile is staged
  }
}

static unsigned long long* g_trace = nullptr;      // diagnostics only
extern "C" void rn_debug_set_chain_trace(void* buf) { g_trace = (unsigned long long*)buf; }

static int num_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) n = p.multiProcessorCount;
    if (n <= 0) n = 256;
  }
  return n;
}
static bool env_on(const char* name, bool dflt) {
  const char* e = rn_diag_env(name);
  if (!e || !e[0]) return dflt;
  return e[0] != '0';
}

template <int MODE>
static void chain_launch(int grid, hipStream_t s, const void* P, int ldp, const ChainArgs& a, int L, float* xg_part,
                         int ntiles) {
  const bool gl = env_on("RN_CHAIN_GLDS", true);            // RN_CHAIN_GLDS=0: register-staged weight slabs
  const bool pf = env_on("RN_CHAIN_PREFETCH", true);        // RN_CHAIN_PREFETCH=0: no next-tile source prefetch
  const char* re = rn_diag_env("RN_CHAIN_RDEPTH");               // fragment read-ahead in K16 steps: 1 (default), 2 or 4
  const int rd = re ? atoi(re) : 1;                          // measured: 1 -> 272 us, 2 -> 287 us, 4 -> 340 us (lock-step phases)
#define RN_GO(G, PFV, R) g_chain_kernel<MODE, PREC_BF16, G, PFV, R><<<grid, CT_NT, 0, s>>>(P, ldp, a, L, xg_part, ntiles, g_trace)
  if (!gl) RN_GO(false, true, 1);
  else if (pf && rd >= 4) RN_GO(true, true, 4);
  else if (pf && rd == 2) RN_GO(true, true, 2);
  else if (pf) RN_GO(true, true, 1);
  else if (rd >= 4) RN_GO(true, false, 4);
  else if (rd == 2) RN_GO(true, false, 2);
  else RN_GO(true, false, 1);
#undef RN_GO
}

extern "C" int rn_g_chain_tile(void) { return CT_TM; }

static int chain_fwd_args(const char* who, ChainArgs& a, const void* P, int ldp, const void* const* W,
                          const void* const* Wlo, const float* const* bias, void* const* H, const int* K, int M, int L,
                          int G) {
  RN_CHECK_ARG(P && W && bias && K && M > 0, "%s: bad pointer/size", who);
  RN_CHECK_ARG(G == CT_G && L >= 1 && L <= CT_MAXL, "%s: needs G == 256 and 1 <= L <= %d (G=%d L=%d)", who, CT_MAXL, G, L);
  RN_CHECK_ARG(M % CT_TM == 0, "%s: M=%d must be a multiple of %d", who, M, CT_TM);
  RN_CHECK_ARG(ldp % 8 == 0 && ldp >= K[0] && ((uintptr_t)P % 16 == 0), "%s: bad P layout", who);
  memset(&a, 0, sizeof(a));
  for (int l = 0; l < L; ++l) {
    RN_CHECK_ARG(W[l] && bias[l] && (!Wlo || Wlo[l]), "%s: layer %d weight/bias is NULL", who, l);
    RN_CHECK_ARG(K[l] % 64 == 0 && K[l] >= 64 && K[l] <= 256 && (l == 0 || K[l] == CT_G),
                 "%s: layer %d reduction length %d unsupported", who, l, K[l]);
    RN_CHECK_ARG(((uintptr_t)W[l] | (uintptr_t)(Wlo ? Wlo[l] : nullptr) | (uintptr_t)bias[l] |
                  (uintptr_t)(H ? H[l] : nullptr)) % 16 == 0,
                 "%s: layer %d pointers must be 16-byte aligned", who, l);
    a.W[l] = W[l];
    a.Wlo[l] = Wlo ? Wlo[l] : nullptr;
    a.bias[l] = bias[l];
    a.out[l] = H ? (bf16*)H[l] : nullptr;
    a.K[l] = K[l];
  }
  return 0;
}

extern "C" int rn_g_chain_fwd(const void* P, int ldp, const void* const* Wp, const float* const* bias,
                              void* const* H, const int* K, float* xg_part, int dtype, int M, int L, int G,
                              void* stream) {
  RN_CHECK_ARG(dtype == RN_BF16, "rn_g_chain_fwd: only the bf16 storage mode has a fused chain (dtype=%d)", dtype);
  ChainArgs a;
  if (int rc = chain_fwd_args("rn_g_chain_fwd", a, P, ldp, Wp, nullptr, bias, H, K, M, L, G)) return rc;
  const int ntiles = M / CT_TM;
  const int grid = ntiles < num_cus() ? ntiles : num_cus();
  chain_launch<MODE_FWD>(grid, (hipStream_t)stream, P, ldp, a, L, xg_part, ntiles);
  RN_LAUNCH_CHECK("rn_g_chain_fwd");
  return 0;
}

extern "C" int rn_g_chain_fwd_f16s(const void* P, int ldp, const void* const* Whi, const void* const* Wlo,
                                   const float* const* bias, void* const* H, const int* K, float* xg_part, int M, int L,
                                   int G, void* stream) {
  RN_CHECK_ARG(Wlo, "rn_g_chain_fwd_f16s: Wlo is NULL");
  ChainArgs a;
  if (int rc = chain_fwd_args("rn_g_chain_fwd_f16s", a, P, ldp, Whi, Wlo, bias, H, K, M, L, G)) return rc;
  const int ntiles = M / CT_TM;
  const int grid = ntiles < num_cus() ? ntiles : num_cus();
  g_chain_kernel<MODE_FWD, PREC_F16S, true, true, 1><<<grid, CT_NT, 0, (hipStream_t)stream>>>(P, ldp, a, L, xg_part, ntiles,
                                                                                           g_trace);
  RN_LAUNCH_CHECK("rn_g_chain_fwd_f16s");
  return 0;
}

extern "C" int rn_g_chain_bwd(const void* HL, const float* dxg, const void* const* Wt, const void* const* Hgate,
                              void* const* dZ, int dtype, int M, int rows_per_question, int L, int G, void* stream) {
  RN_CHECK_ARG(HL && dxg && Wt && Hgate && dZ && M > 0, "rn_g_chain_bwd: bad pointer/size");
  RN_CHECK_ARG(dtype == RN_BF16, "rn_g_chain_bwd: only the bf16 storage mode has a fused chain (dtype=%d)", dtype);
  RN_CHECK_ARG(G == CT_G && L >= 2 && L <= CT_MAXL, "rn_g_chain_bwd: needs G == 256 and 2 <= L <= %d (G=%d L=%d)", CT_MAXL, G, L);
  RN_CHECK_ARG(M % CT_TM == 0 && rows_per_question > 0 && M % rows_per_question == 0,
               "rn_g_chain_bwd: M=%d must be a multiple of %d and of rows per question=%d", M, CT_TM, rows_per_question);
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
    a.W[s] = Wt[s];
    a.gate[s] = (const bf16*)Hgate[s];
    a.out[s] = (bf16*)dZ[s + 1];
    a.K[s] = CT_G;
  }
  const int ntiles = M / CT_TM;
  const int grid = ntiles < num_cus() ? ntiles : num_cus();
  chain_launch<MODE_BWD>(grid, (hipStream_t)stream, nullptr, 0, a, L - 1, nullptr, ntiles);
  RN_LAUNCH_CHECK("rn_g_chain_bwd");
  return 0;
}
