// f_phi as a FEATURE-SPLIT fp32 MFMA chain in ONE launch (model.py:155-162, + the mean NLL of train.py:41 and, in the training
// step, the backward dz chain for d loss = 1): rn_f_phi_split.
//
// Why.  f_phi is three products on B <= 64 rows: 1.8e7 flop against 0.54 MB of fp32 weights.  The row-split kernel of rn_small.hip
// (16 workgroups x 4 rows) makes every workgroup pull EVERY weight matrix through one CU's L1 -- 1.1 MB per workgroup for forward
// + backward, ~2.5 us per 256-KB layer whatever does the arithmetic: 34-38 us on the critical path between the two g_theta chains.
// Here a 256-wide layer is split over its OUTPUT FEATURES instead: workgroup w owns features 16 w .. 16 w + 15 of every layer, i.e.
// a 16-KB slab of each weight matrix (fetched straight into MFMA operand registers while the wave waits for its input), and what
// moves between the layers is the activation, handed from the 16 producers to the 16 consumers INSIDE the launch.
//
// Structure: FOUR INDEPENDENT PIPELINES, no workgroup barrier after the prologue.  The rows of the batch never mix in an MLP: wave v
// of workgroup w computes rows 16 v .. 16 v + 15 of feature slab w, and needs exactly the rows 16 v .. 16 v + 15 of the previous
// layer -- produced by wave v of the 16 workgroups.  The row-wise part in the middle (logits, log-softmax, loss, dz3, dz2) is ONE
// row per wave: wave (w, v) owns row 16 v + w, which again only pipeline v reads.
// Hand-off (CDNA4 guide, Guideline 16, form R2): the data IS the flag.  Every value is published as an 8-byte GRANULE
// {value, tag = epoch of this launch} with write-through (sc1) stores; a consumer sweeps the granules it needs with 16-byte sc1
// loads until every tag is the launch's epoch.  No flag, no drain, no fence, no L2 write-back, no kernel boundary (a launch of the
// replayed step costs >= 6 us; a flag + drain hop measured ~5 us here; a granule hop ~2.5 us).  The plain fp32 tensors the later
// kernels read (xg, f1, f2, the dz rows) are written beside the granules with ordinary stores.
//
// Arithmetic: v_mfma_f32_16x16x4_f32 -- exact fp32, bit-for-bit a k-ordered fmaf chain (north_star: "f_phi as a small MFMA GEMM
// chain"; the 1e-3 bar is met by ~1e-7): 64 MFMAs per wave and layer on two interleaved accumulators (k-groups of 16: even, odd;
// added at the end -- a fixed order, bitwise reproducible).
//   A operand = activation rows (lane (i, g) holds act[row i][16 q + 4 g .. + 3], q = 0..15: two 16-byte granule loads each),
//   B operand = the weight slab rows in the same k order (lane (j, g): W[16 w + j][16 q + 4 g .. + 3]).
// Stages of pipeline v (G* = granule array, 64 x 256 x 8 bytes each, in the sync workspace):
//   GX   wave (w, v): row 16 v + w of the pair sums from the forward chain's per-tile partials (model.py:151-152), partial order
//   GF1  f1[rows, slab] = relu(xg W1^T + b1)                        (feature-split; sweeps GX rows 16 v ..)
//   GF2  f2[rows, slab] = relu((f1 W2^T + b2) * mask)               (feature-split; sweeps GF1)
//   GD2  wave (w, v): row 16 v + w -- logits, log-softmax, NLL, dz3, dz2 = (dz3 W3) * mask * (f2 > 0)   (sweeps one row of GF2)
//   GD1  dz1[rows, slab] = (dz2 W2[:, slab]) * (f1[rows, slab] > 0) (feature-split; the gate never left the registers)
//   --   dxg[rows, slab] = dz1 W1[:, slab]                          (feature-split; plain stores: the next kernel reads them)
//   the row losses travel as 64 granules; wave (0, 0) adds them in row order -> the mean NLL.
// State: `sync` (zeroed ONCE by the caller, then owned by these launches): an epoch word that the launch's last wave advances --
// every tag of launch e is e, so nothing is cleared between launches (a replayed hipGraph has no memset node) --, a completion
// counter, an error word (a sweep that is not answered within ~1 s gives up and stores its stage there instead of hanging: results
// are then garbage and rn_f_phi_split_status reports it), and the granule arrays.
// Residency: 16 workgroups of 256 threads -- resident together on any MI355X that is not wedged; every spin is bounded anyway.
#include "rn_common.h"

namespace {
constexpr int FS_NW = 16, FS_NV = 4, FS_NT = 64, FS_W = 256, FS_ROWS = 64, FS_AMAX = 32;   // 16 feature slabs x 4 row pipelines = 64 one-wave workgroups
constexpr unsigned FS_SPIN_MAX = 1u << 20;
enum { FS_GX = 0, FS_GF1 = 1, FS_GF2 = 2, FS_GD2 = 3, FS_GD1 = 4, FS_NGRAN = 5 };
// sync workspace: words [0] epoch of the last completed launch, [1] completion counter, [2] error (0 = none); byte FS_OFF_LOSS: 64
// loss granules; byte FS_OFF_GRAN + s * FS_GRAN_BYTES: granule array s
constexpr int FS_OFF_LOSS = 256, FS_OFF_GRAN = 1024, FS_GRAN_BYTES = FS_ROWS * FS_W * 8;
constexpr size_t FS_SYNC_BYTES = FS_OFF_GRAN + (size_t)FS_NGRAN * FS_GRAN_BYTES;
constexpr int FS_AUX_SC1 = 16;                            // buffer-instruction cache policy: sc1 = device scope (write-through / L1 bypass)
constexpr int FS_RSRC_FLAGS = 0x00020000;                 // gfx9-family raw-buffer descriptor word 3 (32-bit data format)

struct FsArgs {
  const float* xg_part;                                   // (B * parts, 256) partial pair sums, or NULL: xg is an input
  int parts;
  float* xg;                                              // (B, 256)
  const float *W1, *b1, *W2, *b2, *W3, *b3;               // nn.Linear (out, in) weights
  const float *W1T, *W2T;                                 // (in, out) copies: the slabs of the backward products
  const float* mask;                                      // (B, 256) dropout mask incl. 1 / (1 - p), or NULL
  const long long* label;                                 // (B,) or NULL (then no loss / backward)
  float *f1, *f2, *out, *loss;
  float *dz1, *dz2, *dz3, *dxg;
  unsigned char* sync;
  int B, A;
};

typedef __attribute__((address_space(1))) unsigned gu32;
typedef __attribute__((address_space(1))) unsigned long long gu64;
__device__ __forceinline__ unsigned fs_ld(const void* p) { return __hip_atomic_load((gu32*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void fs_st(void* p, unsigned v) { __hip_atomic_store((gu32*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// ONE aligned 8-byte write-through store = one granule {value, tag}
__device__ __forceinline__ void fs_granule(void* p, float v, unsigned epoch) {
  __hip_atomic_store((gu64*)p, ((unsigned long long)epoch << 32) | __builtin_bit_cast(unsigned, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t fs_rsrc(const void* p, long bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, FS_RSRC_FLAGS);
}
__device__ __forceinline__ u32x4 fs_ld16(__amdgpu_buffer_rsrc_t r, int off) { return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, FS_AUX_SC1); }
__device__ __forceinline__ void fs_give_up(unsigned char* sync, int stage) { fs_st(sync + 8, 1u + (unsigned)stage); }

// The wave's 16 activation rows of granule array `gr` as MFMA A operands: a[q] = act[row0 + i][16 q + 4 g .. + 3] for lane (i, g).
// Sweeps (32 sixteen-byte sc1 loads in flight per lane) until every tag is this launch's; rows >= B are not produced: zeros.
__device__ __forceinline__ void fs_sweep_rows(unsigned char* sync, int stage, int row0, int B, unsigned epoch, f32x4 (&a)[16]) {
  const int lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4, row = row0 + i;
  const __amdgpu_buffer_rsrc_t r = fs_rsrc(sync + FS_OFF_GRAN + (size_t)stage * FS_GRAN_BYTES, FS_GRAN_BYTES);
  const bool live = row < B;
  // light poll first: ONE granule per producer (lane l < 16: the granule of the wave's last live row in slab l's last column) until
  // all 16 carry this launch's tag -- a full sweep is 32 KB per wave (~1.3 us), a light pass 16 loads; the full sweep below still
  // verifies EVERY tag (stores are not ordered), it just starts when it is likely to succeed
  {
    const int lastrow = (row0 + 15 < B ? row0 + 15 : B - 1);
    const unsigned char* sp = sync + FS_OFF_GRAN + (size_t)stage * FS_GRAN_BYTES + ((size_t)lastrow * FS_W + 16 * (lane & 15) + 15) * 8 + 4;
    for (unsigned spins = 0; spins < FS_SPIN_MAX; ++spins) {
      const unsigned tg = lane < FS_NW ? fs_ld(sp) : epoch;
      if (__ballot(tg != epoch) == 0ull) break;
      __builtin_amdgcn_s_sleep(1);
    }
  }
  for (unsigned spins = 0;; ++spins) {
    u32x4 lo[16], hi[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int off = (row * FS_W + 16 * q + 4 * g) * 8;
      lo[q] = fs_ld16(r, off);
      hi[q] = fs_ld16(r, off + 16);
    }
    bool ok = true;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      ok = ok && lo[q][1] == epoch && lo[q][3] == epoch && hi[q][1] == epoch && hi[q][3] == epoch;
      const u32x4 v = {lo[q][0], lo[q][2], hi[q][0], hi[q][2]};
      a[q] = live ? __builtin_bit_cast(f32x4, v) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (__ballot(live && !ok) == 0ull) return;
    if (spins > FS_SPIN_MAX) {                            // (never on a healthy chip: say so instead of hanging)
      if (lane == 0) fs_give_up(sync, stage);
      return;
    }
    __builtin_amdgcn_s_sleep(1);
  }
}
// ONE row of a granule array: lane l gets columns 4 l .. 4 l + 3
__device__ __forceinline__ f32x4 fs_sweep_row(unsigned char* sync, int stage, int row, unsigned epoch) {
  const int lane = threadIdx.x & 63;
  const __amdgpu_buffer_rsrc_t r = fs_rsrc(sync + FS_OFF_GRAN + (size_t)stage * FS_GRAN_BYTES, FS_GRAN_BYTES);
  for (unsigned spins = 0;; ++spins) {
    const u32x4 lo = fs_ld16(r, (row * FS_W + 4 * lane) * 8), hi = fs_ld16(r, (row * FS_W + 4 * lane) * 8 + 16);
    const bool ok = lo[1] == epoch && lo[3] == epoch && hi[1] == epoch && hi[3] == epoch;
    const u32x4 v = {lo[0], lo[2], hi[0], hi[2]};
    if (__ballot(!ok) == 0ull) return __builtin_bit_cast(f32x4, v);
    if (spins > FS_SPIN_MAX) {
      if (lane == 0) fs_give_up(sync, stage);
      return __builtin_bit_cast(f32x4, v);
    }
    __builtin_amdgcn_s_sleep(1);
  }
}
// ... and the publishing side of a row: lane l holds columns 4 l .. 4 l + 3 (two 16-byte write-through stores = four granules)
__device__ __forceinline__ void fs_publish_row(unsigned char* sync, int stage, int row, unsigned epoch, f32x4 v) {
  const int lane = threadIdx.x & 63;
  const __amdgpu_buffer_rsrc_t r = fs_rsrc(sync + FS_OFF_GRAN + (size_t)stage * FS_GRAN_BYTES, FS_GRAN_BYTES);
  // (a whole-vector bit cast: __builtin_bit_cast on an ext-vector ELEMENT lvalue reads element 0 whatever the index -- hipcc 7.2)
  const u32x4 u = __builtin_bit_cast(u32x4, v);
  const u32x4 lo = {u[0], epoch, u[1], epoch};
  const u32x4 hi = {u[2], epoch, u[3], epoch};
  __builtin_amdgcn_raw_buffer_store_b128(lo, r, (row * FS_W + 4 * lane) * 8, 0, FS_AUX_SC1);
  __builtin_amdgcn_raw_buffer_store_b128(hi, r, (row * FS_W + 4 * lane) * 8 + 16, 0, FS_AUX_SC1);
}

// acc[i'] (row row0 + 4 g + i', feature slab column j; lane = (j, g)) = sum_k a[..][k] * slab[j][k]
__device__ __forceinline__ f32x4 fs_mfma(const f32x4 (&a)[16], const f32x4 (&w)[16]) {
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int q = 0; q < 16; q += 2) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q][m], w[q][m], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q + 1][m], w[q + 1][m], acc1, 0, 0, 0);
    }
  }
  return acc0 + acc1;
}
__device__ __forceinline__ void fs_load_slab(const float* W, int wg, f32x4 (&w)[16]) {
  const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
  const f32x4* row = reinterpret_cast<const f32x4*>(W + (long)(FS_NW * wg + j) * FS_W);
#pragma unroll
  for (int q = 0; q < 16; ++q) w[q] = row[4 * q + g];
}
// a feature-split stage's outputs: plain tensor + granules (lane (j, g): rows row0 + 4 g + i', column f)
__device__ __forceinline__ void fs_store_slab(unsigned char* sync, int stage, float* plain, int row0, int f, int B, unsigned epoch, f32x4 v) {
  const int g = (threadIdx.x & 63) >> 4;
  unsigned char* gr = sync + FS_OFF_GRAN + (size_t)stage * FS_GRAN_BYTES;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = row0 + 4 * g + i;
    if (row < B) {
      fs_granule(gr + ((size_t)row * FS_W + f) * 8, v[i], epoch);
      plain[(long)row * FS_W + f] = v[i];
    }
  }
}

template <bool BWD>
__global__ __launch_bounds__(FS_NT) void f_phi_split_kernel(FsArgs a) {
  // ONE LDS object: W3 (A rows at a padded stride), then per wave the f2 row and the dz3 row of the row-wise stage
  constexpr int W3S = FS_W + 4;                           // row stride of the W3 image (floats)
  __shared__ __attribute__((aligned(16))) float lds[FS_AMAX * W3S + (FS_W + FS_AMAX)];
  float* const w3s = lds;
  // one wave per workgroup: a CU then pulls ONE pipeline's hand-off bytes (the per-CU rate of handed-off data, ~65 GB/s, is what a
  // hop costs: 4 waves on a CU measured 9 us per hop, one wave 3); block b = (slab b / 4, pipeline b % 4) -> a pipeline's 16 members sit on two XCDs
  const int t = threadIdx.x, wg = blockIdx.x >> 2, lane = t & 63, wv = blockIdx.x & 3;
  float* const f2s = lds + FS_AMAX * W3S;
  float* const zs = f2s + FS_W;
  const int B = a.B, A = a.A;
  const unsigned epoch = fs_ld(a.sync) + 1u;
  const int j = lane & 15, g = lane >> 4, f = FS_NW * wg + j, row0 = 16 * wv, myrow = row0 + wg;

  f32x4 w[16];
  fs_load_slab(a.W1, wg, w);                               // (in flight during the pair sums and the first sweep)
  // ---- GX first (nothing else is waited for by other workgroups): row 16 v + w of the pair sums, partials added in order
  if (myrow < B) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (a.xg_part) {
      const f32x4* src = reinterpret_cast<const f32x4*>(a.xg_part + (long)myrow * a.parts * FS_W) + lane;
      int p = 0;
      for (; p + 16 <= a.parts; p += 16) {                // 16 loads in flight: one round trip per batch
        f32x4 u[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) u[i] = src[(long)(p + i) * (FS_W / 4)];
#pragma unroll
        for (int i = 0; i < 16; ++i) v += u[i];
      }
      for (; p < a.parts; ++p) v += src[(long)p * (FS_W / 4)];
      reinterpret_cast<f32x4*>(a.xg + (long)myrow * FS_W)[lane] = v;
    } else {
      v = reinterpret_cast<const f32x4*>(a.xg + (long)myrow * FS_W)[lane];
    }
    fs_publish_row(a.sync, FS_GX, myrow, epoch, v);
  }
  // ---- W3 -> LDS (this workgroup's own; one wave: no barrier beyond its own LDS wait)
  for (int c = t; c < FS_AMAX * (FS_W / 4); c += FS_NT) {
    const int r = c / (FS_W / 4), k4 = c - r * (FS_W / 4);
    const f32x4 v = r < A ? reinterpret_cast<const f32x4*>(a.W3 + (long)r * FS_W)[k4] : f32x4{0.f, 0.f, 0.f, 0.f};
    *reinterpret_cast<f32x4*>(w3s + r * W3S + 4 * k4) = v;
  }
  __syncthreads();
  if (row0 < B) {                                          // (a pipeline without rows has nothing to do; its waves still check out below)
    f32x4 act[16];
    // ---- GF1
    f32x4 f1r;
    {
      fs_sweep_rows(a.sync, FS_GX, row0, B, epoch, act);
      const f32x4 acc = fs_mfma(act, w);
      fs_load_slab(a.W2, wg, w);
      const float b = a.b1[f];
#pragma unroll
      for (int i = 0; i < 4; ++i) f1r[i] = fmaxf(acc[i] + b, 0.f);
      fs_store_slab(a.sync, FS_GF1, a.f1, row0, f, B, epoch, f1r);
    }
    // ---- GF2
    {
      fs_sweep_rows(a.sync, FS_GF1, row0, B, epoch, act);
      const f32x4 acc = fs_mfma(act, w);
      if constexpr (BWD) fs_load_slab(a.W2T, wg, w);
      const float b = a.b2[f];
      f32x4 v;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = row0 + 4 * g + i;
        const float m = (a.mask && row < B) ? a.mask[(long)row * FS_W + f] : 1.f;
        v[i] = fmaxf((acc[i] + b) * m, 0.f);
      }
      fs_store_slab(a.sync, FS_GF2, a.f2, row0, f, B, epoch, v);
    }
    // ---- row 16 v + w: logits, log-softmax, NLL, dz3, dz2
    if (myrow < B) {
      const f32x4 fv = fs_sweep_row(a.sync, FS_GF2, myrow, epoch);          // f2[myrow][4 lane ..]
      *reinterpret_cast<f32x4*>(f2s + 4 * lane) = fv;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                     // (this wave's own LDS row: no barrier needed)
      const int c = lane & 31, kh = lane >> 5;                               // class c, k half kh: 128 products, k-ordered
      float z = 0.f;
      {
        const float* wr = w3s + c * W3S + 128 * kh;
        const float* x = f2s + 128 * kh;
        for (int k = 0; k < 128; k += 4) {
          const f32x4 wq = *reinterpret_cast<const f32x4*>(wr + k), xq = *reinterpret_cast<const f32x4*>(x + k);
          z = fmaf(wq[3], xq[3], fmaf(wq[2], xq[2], fmaf(wq[1], xq[1], fmaf(wq[0], xq[0], z))));
        }
      }
      z += __shfl_xor(z, 32);                                                // the two halves (same value in both afterwards)
      z = c < A ? z + a.b3[c] : 0.f;
      const float zz = c < A ? z : -INFINITY;
      float mx = zz;
#pragma unroll
      for (int o = 16; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
      float s = c < A ? expf(z - mx) : 0.f;
#pragma unroll
      for (int o = 16; o >= 1; o >>= 1) s += __shfl_xor(s, o);
      const float ls = mx + logf(s), lp = z - ls;
      if (c < A && kh == 0) a.out[(long)myrow * A + c] = lp;
      if (a.label) {
        const long long lraw = a.label[myrow];
        const int lb = lraw < 0 ? 0 : (lraw >= A ? A - 1 : (int)lraw);
        float nl = (c == lb && kh == 0) ? -lp : 0.f;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) nl += __shfl_xor(nl, o);
        if (lane == 0) fs_granule(a.sync + FS_OFF_LOSS + 8 * myrow, nl, epoch);
        if constexpr (BWD) {
          const float gl = -1.f / (float)B;                                  // d(mean NLL) / d log-prob at the label
          const float d = c < A ? ((c == lb ? gl : 0.f) - expf(lp) * gl) : 0.f;
          if (kh == 0) {
            zs[c] = d;
            if (c < A) a.dz3[(long)myrow * A + c] = d;
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          f32x4 dd = {0.f, 0.f, 0.f, 0.f};
          for (int cc = 0; cc < A; ++cc) {
            const float z3 = zs[cc];
            const f32x4 wq = *reinterpret_cast<const f32x4*>(w3s + cc * W3S + 4 * lane);
#pragma unroll
            for (int e = 0; e < 4; ++e) dd[e] = fmaf(z3, wq[e], dd[e]);
          }
          const f32x4 mv = a.mask ? *reinterpret_cast<const f32x4*>(a.mask + (long)myrow * FS_W + 4 * lane) : f32x4{1.f, 1.f, 1.f, 1.f};
#pragma unroll
          for (int e = 0; e < 4; ++e) dd[e] = fv[e] > 0.f ? dd[e] * mv[e] : 0.f;
          reinterpret_cast<f32x4*>(a.dz2 + (long)myrow * FS_W)[lane] = dd;
          fs_publish_row(a.sync, FS_GD2, myrow, epoch, dd);
        }
      }
    }
    if constexpr (BWD) {
      // ---- GD1: dz1[rows, slab] = (dz2 W2[:, slab]) * (f1[rows, slab] > 0)
      {
        fs_sweep_rows(a.sync, FS_GD2, row0, B, epoch, act);
        const f32x4 acc = fs_mfma(act, w);
        fs_load_slab(a.W1T, wg, w);
        f32x4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = f1r[i] > 0.f ? acc[i] : 0.f;
        fs_store_slab(a.sync, FS_GD1, a.dz1, row0, f, B, epoch, v);
      }
      // ---- dxg[rows, slab] = dz1 W1[:, slab]: plain stores (the next kernel on the stream reads them)
      {
        fs_sweep_rows(a.sync, FS_GD1, row0, B, epoch, act);
        const f32x4 acc = fs_mfma(act, w);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = row0 + 4 * g + i;
          if (row < B) a.dxg[(long)row * FS_W + f] = acc[i];
        }
      }
    }
    // ---- the batch's mean NLL: the row losses in row order (wave (0, 0))
    if (a.label && wg == 0 && wv == 0) {
      float v = 0.f;
      for (unsigned spins = 0;; ++spins) {
        const unsigned long long gq = lane < B ? __hip_atomic_load((gu64*)(a.sync + FS_OFF_LOSS + 8 * lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                               : ((unsigned long long)epoch << 32);
        v = __builtin_bit_cast(float, (unsigned)gq);
        if (__ballot((unsigned)(gq >> 32) != epoch) == 0ull) break;
        if (spins > FS_SPIN_MAX) {
          if (lane == 0) fs_give_up(a.sync, FS_NGRAN);
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
      float tot = 0.f;
      for (int i = 0; i < B; ++i) tot += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), i));
      if (lane == 0) *a.loss = tot / (float)B;
    }
  }
  // ---- the launch's last wave advances the epoch (the counter re-arms itself)
  if (lane == 0) {
    const unsigned n = __hip_atomic_fetch_add((gu32*)(a.sync + 4), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (n == FS_NW * FS_NV - 1) {
      fs_st(a.sync + 4, 0u);
      fs_st(a.sync, epoch);
    }
  }
}
}  // namespace

size_t rnws_f_phi_split(void) { return FS_SYNC_BYTES; }

extern "C" int rn_f_phi_split_ok(int B, int G, int F1, int F2, int A) {
  return B > 0 && B <= FS_ROWS && G == FS_W && F1 == FS_W && F2 == FS_W && A > 0 && A <= FS_AMAX;
}

extern "C" int rn_f_phi_split(const float* xg_part, int parts_per_row, float* xg, const float* W1, const float* b1, const float* W2,
                              const float* b2, const float* W3, const float* b3, const float* W1T, const float* W2T, const float* mask,
                              const long long* label, float* f1, float* f2, float* out, float* loss, void* bwd_ws, float* dxg,
                              void* sync_ws, int B, int G, int F1, int F2, int A, void* stream) {
  RN_CHECK_ARG(rn_f_phi_split_ok(B, G, F1, F2, A), "rn_f_phi_split: needs B <= %d, G = F1 = F2 = %d, A <= %d (B=%d G=%d F1=%d F2=%d A=%d)", FS_ROWS, FS_W,
               FS_AMAX, B, G, F1, F2, A);
  RN_CHECK_ARG(xg && W1 && b1 && W2 && b2 && W3 && b3 && f1 && f2 && out && sync_ws, "rn_f_phi_split: NULL pointer");
  RN_CHECK_ARG(!xg_part || parts_per_row > 0, "rn_f_phi_split: parts_per_row must be positive with xg_part");
  RN_CHECK_ARG((label != nullptr) == (loss != nullptr), "rn_f_phi_split: label and loss go together");
  const bool bwd = bwd_ws != nullptr;
  RN_CHECK_ARG(!bwd || (label && dxg && W1T && W2T), "rn_f_phi_split: the backward dz chain needs label, dxg and the (in, out) weight copies");
  RN_CHECK_ARG(((uintptr_t)xg_part | (uintptr_t)xg | (uintptr_t)W1 | (uintptr_t)W2 | (uintptr_t)W3 | (uintptr_t)W1T | (uintptr_t)W2T | (uintptr_t)mask |
                (uintptr_t)f1 | (uintptr_t)f2 | (uintptr_t)bwd_ws | (uintptr_t)dxg | (uintptr_t)sync_ws) % 16 == 0,
               "rn_f_phi_split: pointers must be 16-byte aligned");
  FsArgs a;
  memset(&a, 0, sizeof(a));
  a.xg_part = xg_part; a.parts = parts_per_row; a.xg = xg;
  a.W1 = W1; a.b1 = b1; a.W2 = W2; a.b2 = b2; a.W3 = W3; a.b3 = b3; a.W1T = W1T; a.W2T = W2T;
  a.mask = mask; a.label = label; a.f1 = f1; a.f2 = f2; a.out = out; a.loss = loss;
  if (bwd) {                                                     // (the dz rows where rn_f_phi_bwd_grads expects them)
    a.dz1 = (float*)bwd_ws;
    a.dz2 = a.dz1 + (size_t)B * F1;
    a.dz3 = a.dz2 + (size_t)B * F2;
    a.dxg = dxg;
  }
  a.sync = (unsigned char*)sync_ws;
  a.B = B; a.A = A;
  if (bwd) f_phi_split_kernel<true><<<FS_NW * FS_NV, FS_NT, 0, (hipStream_t)stream>>>(a);
  else f_phi_split_kernel<false><<<FS_NW * FS_NV, FS_NT, 0, (hipStream_t)stream>>>(a);
  RN_LAUNCH_CHECK("rn_f_phi_split");
  return 0;
}

// error word of a sync workspace (device -> host copy on `stream`, synchronised): 0 = every sweep of every launch so far was
// answered; s + 1 = a wave gave up waiting in stage s (results of that launch are garbage)
extern "C" int rn_f_phi_split_status(const void* sync_ws, void* stream) {
  unsigned v = 0;
  hipError_t e = hipMemcpyAsync(&v, (const unsigned char*)sync_ws + 8, sizeof(v), hipMemcpyDeviceToHost, (hipStream_t)stream);
  if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
  if (e != hipSuccess) {
    rn_set_error("rn_f_phi_split_status: %s", hipGetErrorString(e));
    return -(int)e - 1000;
  }
  return (int)v;
}
