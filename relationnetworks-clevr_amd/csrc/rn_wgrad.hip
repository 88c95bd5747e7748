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
#include "rn_common.h"
#include "../../include/rn_hip_debug.h"

template <typename T> struct WG;
template <> struct WG<bf16> { static constexpr int ROWS = 64; };
template <> struct WG<float> { static constexpr int ROWS = 32; };

// X3 (T = float only; dtype RN_F32X3, the "bf16x3" precision): fp32 operands split into hi + lo bf16 as they are staged -- two bf16
// tiles per operand in LDS, read with the bf16 path's transposing reads -- and every 16-row step is three bf16 MFMAs
// (lo x hi, hi x lo, hi x hi) instead of eight fp32 ones; 2^-16 of a product is dropped (rn_gemm.hip has the forward's twin).
template <typename T, int NKT, bool USE_TR, bool X3 = false>
__global__ __launch_bounds__(512) void wgrad_kernel(const T* __restrict__ dZ, int lddz, const T* __restrict__ A,
                                                    int lda, float* __restrict__ part, float* __restrict__ part_db,
                                                    int M, int N, int Kpad, int rows_per_split) {
  constexpr int CH = Elem<T>::kPer16B;
  constexpr int ROWS = WG<T>::ROWS;
  // LDS row stride (bytes), both tiles.  bf16: +64 B so that the 4 rows x 64 B a 32-lane group of
  // ds_read_b64_tr_b16 touches land on 4 distinct 16-bank quarters (stride = 16 dwords mod 64); fp32: +16 B.
  static_assert(!X3 || (sizeof(T) == 4 && USE_TR), "the split arithmetic is a mode of the fp32 storage");
  constexpr int RSZ = X3 ? 256 * 2 + 64 : 256 * (int)sizeof(T) + (sizeof(T) == 2 ? 64 : 16);
  constexpr int CPZ = 256 / CH;                            // 16-byte chunks per dZ tile row
  constexpr int CPA = NKT * 32 / CH;                       // chunks per A tile row
  constexpr int NZ = ROWS * CPZ / 512;                     // dZ chunks per thread per step (4)
  constexpr int NA = (ROWS * CPA + 511) / 512;             // A chunks per thread per step (<= 4)
  __shared__ __attribute__((aligned(16))) unsigned char lds[(X3 ? 4 : 2) * ROWS * RSZ];   // 66-74 KB static
  unsigned char* ldsZ = lds;
  unsigned char* ldsA = lds + ROWS * RSZ;
  unsigned char* ldsZl = lds + 2 * ROWS * RSZ;             // X3: the lo tiles (ldsZ / ldsA hold the hi ones)
  unsigned char* ldsAl = lds + 3 * ROWS * RSZ;

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
  typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_;
  auto split_store = [&](unsigned char* hi_p, unsigned char* lo_p, const u32x4 v) {   // 4 floats -> 4 hi + 4 lo bf16 (8 bytes each)
    const f32x4 x = __builtin_bit_cast(f32x4, v);
    const bf16 h0 = (bf16)x[0], h1 = (bf16)x[1], h2 = (bf16)x[2], h3 = (bf16)x[3];
    const bf16x4_ hi = {h0, h1, h2, h3};
    const bf16x4_ lo = {(bf16)(x[0] - (float)h0), (bf16)(x[1] - (float)h1), (bf16)(x[2] - (float)h2), (bf16)(x[3] - (float)h3)};
    *reinterpret_cast<bf16x4_*>(hi_p) = hi;
    *reinterpret_cast<bf16x4_*>(lo_p) = lo;
  };
  auto lstore = [&]() {
    if constexpr (X3) {
#pragma unroll
      for (int s = 0; s < NZ; ++s) split_store(ldsZ + zrow[s] * RSZ + zcc[s] * 8, ldsZl + zrow[s] * RSZ + zcc[s] * 8, rz[s]);
#pragma unroll
      for (int s = 0; s < NA; ++s)
        if (t + 512 * s < ROWS * CPA) split_store(ldsA + arow[s] * RSZ + acc_[s] * 8, ldsAl + arow[s] * RSZ + acc_[s] * 8, ra[s]);
    } else {
#pragma unroll
      for (int s = 0; s < NZ; ++s) *reinterpret_cast<u32x4*>(ldsZ + zrow[s] * RSZ + zcc[s] * 16) = rz[s];
#pragma unroll
      for (int s = 0; s < NA; ++s)
        if (t + 512 * s < ROWS * CPA) *reinterpret_cast<u32x4*>(ldsA + arow[s] * RSZ + acc_[s] * 16) = ra[s];
    }
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
    if constexpr (X3) {
#pragma unroll
      for (int kk0 = 0; kk0 < ROWS; kk0 += 16) {
        const int roff = (kk0 + rb * 8 + (i16 >> 2)) * RSZ;                    // (the bf16 path's transposing reads, on the hi and the lo tiles)
        const int coff = (cb * 16 + 4 * (i16 & 3)) * 2;
        typedef __attribute__((address_space(3))) s16x4* lptr;
        union { struct { s16x4 a, b; } s; bf16x8 v; } u;
        auto frag = [&](const unsigned char* p) {
          u.s.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(p));
          u.s.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(p + 4 * RSZ));
          return u.v;
        };
        const bf16x8 fzh = frag(ldsZ + roff + (w * 32) * 2 + coff), fzl = frag(ldsZl + roff + (w * 32) * 2 + coff);
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
          const bf16x8 fah = frag(ldsA + roff + (kt * 32) * 2 + coff), fal = frag(ldsAl + roff + (kt * 32) * 2 + coff);
          acc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fzl, fah, acc[kt], 0, 0, 0);
          acc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fzh, fal, acc[kt], 0, 0, 0);
          acc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fzh, fah, acc[kt], 0, 0, 0);
        }
      }
    } else if constexpr (sizeof(T) == 2) {
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
      if constexpr (X3) {
#pragma unroll 8
        for (int r = 0; r < ROWS; ++r)
          dbsum += (float)*reinterpret_cast<const bf16*>(ldsZ + r * RSZ + t * 2) + (float)*reinterpret_cast<const bf16*>(ldsZl + r * RSZ + t * 2);
      } else {
#pragma unroll 8
        for (int r = 0; r < ROWS; ++r) dbsum += Elem<T>::to_f32(*reinterpret_cast<const T*>(ldsZ + r * RSZ + t * sizeof(T)));
      }
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

size_t rnws_wgrad(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0 || N % 256 || K % 32) return 0;
  int nkt, gy, Z, rps;
  wgrad_plan(M, N, K, &nkt, &gy, &Z, &rps);
  return ((size_t)Z * N * K + (size_t)Z * N) * sizeof(float);
}


template <typename T, int NKT>
static void wgrad_dispatch(dim3 grid, hipStream_t s, const T* dZ, int lddz, const T* A, int lda,
                           float* part, float* part_db, int M, int N, int K, int rps, bool x3 = false) {
  if constexpr (sizeof(T) == 4) {
    if (x3) {
      wgrad_kernel<T, NKT, true, true><<<grid, 512, 0, s>>>(dZ, lddz, A, lda, part, part_db, M, N, K, rps);
      return;
    }
  }
  if constexpr (sizeof(T) == 2) {
    wgrad_kernel<T, NKT, true><<<grid, 512, 0, s>>>(dZ, lddz, A, lda, part, part_db, M, N, K, rps);
  } else {
    wgrad_kernel<T, NKT, false><<<grid, 512, 0, s>>>(dZ, lddz, A, lda, part, part_db, M, N, K, rps);
  }
}

template <typename T>
static int wgrad_launch_t(const T* dZ, int lddz, const T* A, int lda, float* part, float* part_db, int M, int N, int K,
                          int nkt, int gy, int Z, int rps, hipStream_t s, bool x3 = false) {
  dim3 grid(N / 256, gy, Z);
  switch (nkt) {
#define RN_CASE(n) case n: wgrad_dispatch<T, n>(grid, s, dZ, lddz, A, lda, part, part_db, M, N, K, rps, x3); break;
    RN_CASE(1) RN_CASE(2) RN_CASE(3) RN_CASE(4) RN_CASE(5) RN_CASE(6) RN_CASE(7) RN_CASE(8)
#undef RN_CASE
    default: rn_set_error("rn_g_linear_bwd_wgrad: bad k-tile count %d", nkt); return -1;
  }
  return 0;
}

extern "C" int rn_g_linear_bwd_wgrad(const void* dZ, int lddz, const void* A, int lda, float* dW, float* db, void* ws,
                                     int dtype, int M, int N, int K, int Ktrue, void* stream) {
  RN_CHECK_ARG(dZ && A && dW && ws && M > 0, "rn_g_linear_bwd_wgrad: bad pointer/size");
  RN_CHECK_ARG(dtype == RN_BF16 || dtype == RN_F32 || dtype == RN_F32X3, "rn_g_linear_bwd_wgrad: bad dtype %d", dtype);
  const int CH = dtype == RN_BF16 ? 8 : 4;
  RN_CHECK_ARG(N % 256 == 0 && K % 32 == 0 && Ktrue > 0 && Ktrue <= K, "rn_g_linear_bwd_wgrad: N=%d K=%d Ktrue=%d unsupported",
               N, K, Ktrue);
  RN_CHECK_ARG(lddz % CH == 0 && lda % CH == 0 && lddz >= N && lda >= K, "rn_g_linear_bwd_wgrad: bad leading dimensions");
  RN_CHECK_ARG(((uintptr_t)dZ | (uintptr_t)A) % 16 == 0, "rn_g_linear_bwd_wgrad: pointers must be 16-byte aligned");
  int nkt, gy, Z, rps;
  wgrad_plan(M, N, K, &nkt, &gy, &Z, &rps);
  float* part = (float*)ws;
  hipStream_t s = (hipStream_t)stream;
  float* part_db = part + (size_t)Z * N * K;
  int rc;
  if (dtype == RN_BF16)
    rc = wgrad_launch_t<bf16>((const bf16*)dZ, lddz, (const bf16*)A, lda, part, part_db, M, N, K, nkt, gy, Z, rps, s);
  else
    rc = wgrad_launch_t<float>((const float*)dZ, lddz, (const float*)A, lda, part, part_db, M, N, K, nkt, gy, Z, rps, s, dtype == RN_F32X3);
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
