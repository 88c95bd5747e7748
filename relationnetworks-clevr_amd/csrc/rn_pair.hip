// Pair-axis kernels (HBM-bound byte movers and reductions):
//   K1  pair_build        model.py:112-127      materialise P = [x_j | x_i | q | 0]
//       qst_broadcast     model.py:135-140      question columns of a wide activation
//       pack_matrix       model.py:96-99        nn.Linear weights -> padded MFMA operand
//   K3  segsum / pair_sum model.py:151-152      x_g = sum over the n*n pairs
//       pair_sum_bwd      (autograd of :151 + last ReLU gate)
//       pair_reduce_bwd   (autograd of :117-127, algebraic form)
// All stores are 16-byte, fully coalesced; no MFMA here (K1 is HBM-write bound).
#include <stdarg.h>

#include "rn_common.h"

// ---------------------------------------------------------------- error state
static thread_local char g_err[512] = "";
void rn_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* rn_last_error(void) { return g_err; }
extern "C" int rn_abi_version(void) { return RN_ABI_VERSION; }

extern "C" int rn_stream_abandon_capture(void* stream) {
  hipStream_t s = (hipStream_t)stream;
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  hipError_t e = hipStreamIsCapturing(s, &st);
  if (e == hipSuccess && st == hipStreamCaptureStatusNone) return 0;
  hipGraph_t g = nullptr;
  (void)hipStreamEndCapture(s, &g);                      // (an invalidated capture returns an error here AND leaves capture mode)
  if (g) (void)hipGraphDestroy(g);
  (void)hipGetLastError();
  st = hipStreamCaptureStatusNone;
  e = hipStreamIsCapturing(s, &st);
  (void)hipGetLastError();
  if (e == hipSuccess && st == hipStreamCaptureStatusNone) return 0;
  rn_set_error("rn_stream_abandon_capture: the stream is still capturing (status %d, %s)", (int)st, hipGetErrorString(e));
  return e != hipSuccess ? (int)e : 1;
}

// ------------------------------------------------------------------ workspace sizes
extern "C" size_t rn_workspace_bytes(int op, int a, int b, int c, int d) {
  switch (op) {
    case RN_WS_RR_MASK: return rnws_rr_mask(a);
    case RN_WS_PAIR_SUM: return rnws_pair_sum(a, b, c);
    case RN_WS_WGRAD: return rnws_wgrad(a, b, c);
    case RN_WS_WGRAD_BLOCKED: return rnws_wgrad_blocked(a, b, c, d);
    case RN_WS_PAIR_REDUCE: return rnws_pair_reduce(a, b, c);
    case RN_WS_WGRAD0: return rnws_wgrad0(a, b, c);
    case RN_WS_PAIR_FEATURES: return rnws_pair_features(a, b, c);
    case RN_WS_EXTRACT: return rnws_extract(a, b, c);
    case RN_WS_F_PHI_BWD: return rnws_f_phi_bwd(a, b, c, d);
    case RN_WS_F_PHI_NLL: return rnws_f_phi_nll(a);
    case RN_WS_CLIP_ADAM: return rnws_clip_adam();
    case RN_WS_CONV_BWD_WEIGHT: return rnws_conv_bwd_weight(a, b, c, d);
    case RN_WS_BN_RELU: return rnws_bn_relu(a, b, c);
    case RN_WS_F_PHI_SPLIT: return rnws_f_phi_split();
    default: rn_set_error("rn_workspace_bytes: unknown op %d", op); return 0;
  }
}

// ------------------------------------------------------------------ K1 pair build
// One workgroup = one question b and IB consecutive "i" objects.  For a fixed (b,i) the
// n rows (b,i,0..n-1) are one contiguous n*ld*sizeof(T) byte span of P, and only the first
// k columns change with j.  LDS holds:  pre[n][PW] (first PW = roundup(k,CH) columns of every
// j-row) and suf[ld] (the j-independent remainder: x_i | q | 0).  The copy loop then streams
// 16-byte chunks LDS -> HBM, consecutive lanes -> consecutive 16 B.
template <typename T, bool NT>
__global__ __launch_bounds__(256) void pair_build_kernel(const float* __restrict__ x, long sxb, long sxn, long sxk,
                                                         const float* __restrict__ q, long sqb, T* __restrict__ P,
                                                         int n, int k, int Q, int ld, int IB) {
  constexpr int CH = Elem<T>::kPer16B;
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4_;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int PW = (k + CH - 1) / CH * CH;
  T* pre = reinterpret_cast<T*>(smem_raw);          // [n][PW]   (columns k..PW-1 are never written: masked out below)
  T* suf = pre + (size_t)n * PW;                    // [IB][ld]  (columns 0..k-1 likewise)
  const int b = blockIdx.y;
  const int t = threadIdx.x;
  const int i0 = blockIdx.x * IB;
  const int ni = min(IB, n - i0);
  const float* xb = x + (long)b * sxb;
  const int cpr = ld / CH;                          // 16-byte chunks per row
  // One staging phase for the whole block -- the j-dependent head pre[j][e] = x[b,j,e] (e < k) and the j-independent
  // remainder (x_i | q | 0) of every i this block writes -- then ONE barrier; the copy loops below never wait again.
  // Every global load of the phase is issued before the first LDS write waits on one: one memory latency, not two.
  // Straight-line: unconditional loads from clamped addresses (a branch per load would put a full wait after each one),
  // (row, column) of the walk advanced incrementally instead of divided out per element.
  const int ns = ni * (ld - k), sw = ld - k;
  const bool efast = sxk <= sxn;      // consecutive lanes walk the contiguous axis of x (object-major input: k; the conv view: n)
  const int inner = efast ? k : n, outer = efast ? n : k;
  const long s_in = efast ? sxk : sxn, s_out = efast ? sxn : sxk;
  constexpr int US = 6, UP = 8;
  float rs[US], rp[UP];
  int sa = t / sw, sr = t - sa * sw;                 // suffix element t + 256 u  ->  (i - i0, column - k)
  const int sqa = 256 / sw, sqr = 256 - sqa * sw;
  int sa_[US], sr_[US];
#pragma unroll
  for (int u = 0; u < US; ++u) {
    sa_[u] = sa; sr_[u] = sr;
    const int c = k + sr, ii = min(sa, ni - 1);
    const float* src = c < 2 * k ? xb + (long)(i0 + ii) * sxn + (long)(c - k) * sxk
                                 : (c < 2 * k + Q ? q + (long)b * sqb + (c - 2 * k) : xb);
    rs[u] = *src;
    if (c >= 2 * k + Q) rs[u] = 0.f;
    sr += sqr; sa += sqa;
    if (sr >= sw) { sr -= sw; ++sa; }
  }
  int pa = t / inner, pr = t - pa * inner;           // head element t + 256 u  ->  (outer, inner) index of x[b]
  const int pqa = 256 / inner, pqr = 256 - pqa * inner;
  for (int base = 0; base < n * k; base += 256 * UP) {
    int pa_[UP], pr_[UP];
#pragma unroll
    for (int u = 0; u < UP; ++u) {
      pa_[u] = pa; pr_[u] = pr;
      rp[u] = xb[(long)min(pa, outer - 1) * s_out + (long)pr * s_in];
      pr += pqr; pa += pqa;
      if (pr >= inner) { pr -= inner; ++pa; }
    }
    if (base == 0) {
#pragma unroll
      for (int u = 0; u < US; ++u)
        if (sa_[u] < ni) suf[sa_[u] * ld + k + sr_[u]] = Elem<T>::from_f32(rs[u]);
    }
#pragma unroll
    for (int u = 0; u < UP; ++u)
      if (pa_[u] < outer) pre[(efast ? pa_[u] : pr_[u]) * PW + (efast ? pr_[u] : pa_[u])] = Elem<T>::from_f32(rp[u]);
  }
  for (int idx = t + 256 * US; idx < ns; idx += 256) {      // (more than 6 * 256 suffix elements: very wide rows)
    const int ii = idx / sw, c = k + idx % sw;
    float v = 0.f;
    if (c < 2 * k) v = xb[(long)(i0 + ii) * sxn + (long)(c - k) * sxk];
    else if (c < 2 * k + Q) v = q[(long)b * sqb + (c - 2 * k)];
    suf[ii * ld + c] = Elem<T>::from_f32(v);
  }
  __syncthreads();
  const int hc = PW / CH;                            // head chunks per row
  // A thread keeps ONE chunk column c for a whole (b, i) span: the j-independent chunks (c >= hc: x_i | q | 0, all but
  // the first few of a row) are then a register constant -- no LDS read, no index arithmetic per store -- and only the
  // head chunks are read from LDS per row (the chunk that straddles column k is merged in registers: lanes of x_j under
  // the mask, x_i above it).  rpp rows per pass, consecutive lanes -> consecutive 16 B of consecutive rows (one contiguous
  // rpp * ld * sizeof(T) byte burst per pass).
  // (round 6, measured and not adopted: the (b, i) span as one flat run of chunks, 256 per pass -- every lane stores in every pass
  //  where rows-per-pass leaves 14 % of the store slots empty at ld = 192, but the j-independent chunk then comes from LDS per store
  //  instead of from a register: 20.1 / 20.4 us against 19.5 / 20.0 alternating on one box; the kernel is not store-issue bound)
  const int rpp = 256 / cpr;                         // rows per pass (threads beyond rpp * cpr idle in the copy)
  if (rpp > 0) {
    const int jr = t / cpr, c = t - jr * cpr;
    if (jr >= rpp) return;
    const bool head = c < hc;
    u32x4_ m = {0u, 0u, 0u, 0u};                     // bytes of this chunk that come from x_j
    if (head) {
      unsigned char* mb = reinterpret_cast<unsigned char*>(&m);
#pragma unroll
      for (int e = 0; e < 16; ++e) mb[e] = (c * CH + e / (int)sizeof(T) < k) ? 0xff : 0;
    }
    const T* prow = pre + c * CH;
    for (int ii = 0; ii < ni; ++ii) {
      u32x4_* dst = reinterpret_cast<u32x4_*>(P + ((long)(b * n + i0 + ii) * n) * ld) + c;
      const u32x4_ sv = *reinterpret_cast<const u32x4_*>(suf + ii * ld + c * CH) & ~m;
      for (int j = jr; j < n; j += rpp) {
        u32x4_ v = sv;
        if (head) v |= *reinterpret_cast<const u32x4_*>(prow + j * PW) & m;
        if (NT) __builtin_nontemporal_store(v, dst + (long)j * cpr); else dst[(long)j * cpr] = v;
      }
    }
  } else {                                           // rows wider than 256 chunks: the generic walk
    const int total = n * cpr;
    for (int ii = 0; ii < ni; ++ii) {
      u32x4_* dst = reinterpret_cast<u32x4_*>(P + ((long)(b * n + i0 + ii) * n) * ld);
      for (int g = t; g < total; g += 256) {
        const int j = g / cpr, c = g - j * cpr;
        u32x4_ v = *reinterpret_cast<const u32x4_*>(suf + ii * ld + c * CH);
        if (c < hc) {
          u32x4_ m;
          unsigned char* mb = reinterpret_cast<unsigned char*>(&m);
          for (int e = 0; e < 16; ++e) mb[e] = (c * CH + e / (int)sizeof(T) < k) ? 0xff : 0;
          v = (*reinterpret_cast<const u32x4_*>(pre + j * PW + c * CH) & m) | (v & ~m);
        }
        dst[g] = v;
      }
    }
  }
}

extern "C" int rn_pair_build_fwd(const float* x, long sxb, long sxn, long sxk, const float* q, long sqb, void* P,
                                 int dtype, int B, int n, int k, int Q, int ld, void* stream) {
  RN_CHECK_ARG(x && P && B > 0 && n > 0 && k > 0 && Q >= 0, "rn_pair_build_fwd: bad pointer/size");
  RN_CHECK_ARG(Q == 0 || q, "rn_pair_build_fwd: Q > 0 but q is NULL");
  RN_CHECK_ARG(ld % 64 == 0 && ld >= 2 * k + Q, "rn_pair_build_fwd: ld=%d must be a multiple of 64 and >= 2k+Q=%d", ld,
               2 * k + Q);
  RN_CHECK_ARG(dtype == RN_BF16 || dtype == RN_F32 || dtype == RN_F16, "rn_pair_build_fwd: bad dtype %d", dtype);
  // rows of one i per workgroup: ~56 KB of stores behind each staging phase (measured optimum 2 at the headline shape: 18.7 us
  // vs 20.2 / 19.1 for 1 / 4), but no fewer than ~1000 workgroups
  const long span = (long)n * ld * (dtype == RN_F32 ? 4 : 2);
  int IB = (int)((57344 + span / 2) / span);
  IB = IB < 2 ? 2 : (IB > 8 ? 8 : IB);
  const int cap = (int)((long)B * n / 1024);
  if (IB > cap) IB = cap < 1 ? 1 : cap;
  bool nt = true;
#ifdef RN_DIAG
  if (const char* e = rn_diag_env("RN_K1_IB")) IB = atoi(e) > 0 ? atoi(e) : IB;
  if (const char* e = rn_diag_env("RN_K1_NT")) nt = atoi(e) != 0;
#endif
  dim3 grid(cdiv(n, IB), B);
  hipStream_t s = (hipStream_t)stream;
  const int esz = dtype == RN_F32 ? 4 : 2, ch = 16 / esz;
  const size_t lds = ((size_t)n * ((k + ch - 1) / ch * ch) + (size_t)IB * ld) * esz;
  RN_CHECK_ARG(lds <= RN_LDS_MAX, "rn_pair_build_fwd: n*k too large for LDS staging (%zu B > %d B)", lds, RN_LDS_MAX);
#define RN_K1_LAUNCH(T, NT)                                                                      \
  do {                                                                                           \
    if (lds > 64 * 1024) RN_LDS_OPT_IN((pair_build_kernel<T, NT>), "rn_pair_build_fwd");         \
    pair_build_kernel<T, NT><<<grid, 256, lds, s>>>(x, sxb, sxn, sxk, q, sqb, (T*)P, n, k, Q, ld, IB); \
  } while (0)
  if (dtype == RN_BF16) { if (nt) RN_K1_LAUNCH(bf16, true); else RN_K1_LAUNCH(bf16, false); }
  else if (dtype == RN_F16) { if (nt) RN_K1_LAUNCH(f16, true); else RN_K1_LAUNCH(f16, false); }
  else { if (nt) RN_K1_LAUNCH(float, true); else RN_K1_LAUNCH(float, false); }
#undef RN_K1_LAUNCH
  RN_LAUNCH_CHECK("rn_pair_build_fwd");
  return 0;
}

// ------------------------------------------------------------ question broadcast
template <typename T>
__global__ __launch_bounds__(256) void qst_broadcast_kernel(const float* __restrict__ q, long sqb, T* __restrict__ A,
                                                            int npairs, int Q, int col0, int ld) {
  constexpr int CH = Elem<T>::kPer16B;
  const int b = blockIdx.y;
  __shared__ __attribute__((aligned(16))) T qs[1024];
  for (int c = threadIdx.x; c < Q; c += 256) qs[c] = Elem<T>::from_f32(q[(long)b * sqb + c]);
  __syncthreads();
  const int cq = Q / CH;
  const long total = (long)npairs * cq;
  for (long g = (long)blockIdx.x * 256 + threadIdx.x; g < total; g += (long)gridDim.x * 256) {
    const long p = g / cq;
    const int c = (int)(g - p * cq);
    *reinterpret_cast<Chunk16<T>*>(A + ((long)b * npairs + p) * ld + col0 + c * CH) =
        *reinterpret_cast<const Chunk16<T>*>(qs + c * CH);
  }
}

extern "C" int rn_qst_broadcast(const float* q, long sqb, void* A, int dtype, int B, int n, int Q, int col0, int ld,
                                void* stream) {
  RN_CHECK_ARG(q && A && B > 0 && n > 0, "rn_qst_broadcast: bad pointer/size");
  RN_CHECK_ARG(Q > 0 && Q <= 1024 && Q % 8 == 0 && col0 % 8 == 0 && ld % 8 == 0 && col0 + Q <= ld,
               "rn_qst_broadcast: Q=%d col0=%d ld=%d must be multiples of 8, Q <= 1024, col0+Q <= ld", Q, col0, ld);
  RN_CHECK_ARG(dtype == RN_BF16 || dtype == RN_F32, "rn_qst_broadcast: bad dtype %d", dtype);
  const int npairs = n * n;
  dim3 grid(cdiv((long)npairs * Q / 8, 256 * 4) < 1 ? 1 : cdiv((long)npairs * Q / 8, 256 * 4), B);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == RN_BF16) qst_broadcast_kernel<bf16><<<grid, 256, 0, s>>>(q, sqb, (bf16*)A, npairs, Q, col0, ld);
  else qst_broadcast_kernel<float><<<grid, 256, 0, s>>>(q, sqb, (float*)A, npairs, Q, col0, ld);
  RN_LAUNCH_CHECK("rn_qst_broadcast");
  return 0;
}

// ------------------------------------------------------------------ weight pack
template <typename T>
__global__ __launch_bounds__(256) void pack_matrix_kernel(const float* __restrict__ src, long sr, long sc, int R, int C,
                                                          T* __restrict__ dst, int ld, int Rpad) {
  const long total = (long)Rpad * ld;
  for (long g = (long)blockIdx.x * 256 + threadIdx.x; g < total; g += (long)gridDim.x * 256) {
    const int r = (int)(g / ld), c = (int)(g - (long)r * ld);
    const float v = (r < R && c < C) ? src[(long)r * sr + (long)c * sc] : 0.f;
    dst[g] = Elem<T>::from_f32(v);
  }
}

extern "C" int rn_pack_matrix(const float* src, long sr, long sc, int R, int C, void* dst, int dtype, int ld, int Rpad,
                              void* stream) {
  RN_CHECK_ARG(src && dst && R > 0 && C > 0 && ld >= C && Rpad >= R, "rn_pack_matrix: bad pointer/size");
  RN_CHECK_ARG(dtype == RN_BF16 || dtype == RN_F32 || dtype == RN_F16, "rn_pack_matrix: bad dtype %d", dtype);
  const long total = (long)Rpad * ld;
  int blocks = cdiv(total, 256);
  if (blocks > 2048) blocks = 2048;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == RN_BF16) pack_matrix_kernel<bf16><<<blocks, 256, 0, s>>>(src, sr, sc, R, C, (bf16*)dst, ld, Rpad);
  else if (dtype == RN_F16) pack_matrix_kernel<f16><<<blocks, 256, 0, s>>>(src, sr, sc, R, C, (f16*)dst, ld, Rpad);
  else pack_matrix_kernel<float><<<blocks, 256, 0, s>>>(src, sr, sc, R, C, (float*)dst, ld, Rpad);
  RN_LAUNCH_CHECK("rn_pack_matrix");
  return 0;
}

// ----------------------------------------------------- segmented sum (K3 and Ri)
// in: (nseg*seglen rows, ld) of T, width G.  Block (slice s, segment) sums up to 256 rows of
// its segment; thread = (row lane, 16-byte column chunk); fp32 accumulation; the row lanes are
// combined through LDS in a fixed order -> deterministic.
template <typename T>
__global__ __launch_bounds__(256) void segsum_kernel(const T* __restrict__ in, int ld, float* __restrict__ out,
                                                     int seglen, int G, int S) {
  constexpr int CH = Elem<T>::kPer16B;
  __shared__ float red[256 * 8];
  const int seg = blockIdx.y, s = blockIdx.x;
  const int cpr = G / CH;              // chunks per row (<= 256)
  const int lanes = 256 / cpr;         // row lanes
  const int t = threadIdx.x;
  const int c = t % cpr, rl = t / cpr;
  float acc[CH];
#pragma unroll
  for (int e = 0; e < CH; ++e) acc[e] = 0.f;
  const int r0 = s * 256;
  const int r1 = (r0 + 256 < seglen) ? r0 + 256 : seglen;
  if (rl < lanes) {
    const T* base = in + ((long)seg * seglen) * ld + c * CH;
    int r = r0 + rl;
    // 8 independent 16-byte loads per memory round trip (the kernel is a latency chain between the forward chain and
    // f_phi); rows are still added in ascending order
    for (; r + 7 * lanes < r1; r += 8 * lanes) {
      Chunk16<T> v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const Chunk16<T>*>(base + (long)(r + u * lanes) * ld);
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int e = 0; e < CH; ++e) acc[e] += Elem<T>::to_f32(v[u].v[e]);
    }
    for (; r < r1; r += lanes) {
      const Chunk16<T> v = *reinterpret_cast<const Chunk16<T>*>(base + (long)r * ld);
#pragma unroll
      for (int e = 0; e < CH; ++e) acc[e] += Elem<T>::to_f32(v.v[e]);
    }
  }
#pragma unroll
  for (int e = 0; e < CH; ++e) red[t * CH + e] = acc[e];
  __syncthreads();
  // thread t < G sums column t over the row lanes
  for (int col = t; col < G; col += 256) {
    const int cc = col / CH, e = col % CH;
    float sum = 0.f;
    for (int l = 0; l < lanes; ++l) sum += red[(l * cpr + cc) * CH + e];
    out[((long)seg * S + s) * G + col] = sum;
  }
}

__global__ __launch_bounds__(256) void segsum_finish_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                            long total, int G, int S) {
  for (long g = (long)blockIdx.x * 256 + threadIdx.x; g < total; g += (long)gridDim.x * 256) {
    const long seg = g / G;
    const int c = (int)(g - seg * G);
    float sum = 0.f;
    for (int s = 0; s < S; ++s) sum += part[(seg * S + s) * G + c];
    out[g] = sum;
  }
}

static int segsum_launch(const void* in, int ld, float* out, void* ws, int dtype, int nseg, int seglen, int G,
                         hipStream_t s, const char* who) {
  RN_CHECK_ARG(in && out && nseg > 0 && seglen > 0, "%s: bad pointer/size", who);
  RN_CHECK_ARG(dtype == RN_BF16 || dtype == RN_F32, "%s: bad dtype %d", who, dtype);
  const int CH = dtype == RN_BF16 ? 8 : 4;
  RN_CHECK_ARG(G % CH == 0 && G / CH <= 256 && ld % CH == 0, "%s: G=%d / ld=%d unsupported", who, G, ld);
  const int S = cdiv(seglen, 256);
  RN_CHECK_ARG(S == 1 || ws, "%s: workspace required", who);
  float* part = S == 1 ? out : (float*)ws;
  dim3 grid(S, nseg);
  if (dtype == RN_BF16) segsum_kernel<bf16><<<grid, 256, 0, s>>>((const bf16*)in, ld, part, seglen, G, S);
  else segsum_kernel<float><<<grid, 256, 0, s>>>((const float*)in, ld, part, seglen, G, S);
  RN_LAUNCH_CHECK(who);
  if (S > 1) {
    const long total = (long)nseg * G;
    int blocks = cdiv(total, 256);
    if (blocks > 1024) blocks = 1024;
    segsum_finish_kernel<<<blocks, 256, 0, s>>>(part, out, total, G, S);
    RN_LAUNCH_CHECK(who);
  }
  return 0;
}

size_t rnws_pair_sum(int B, int npairs, int G) {
  return (size_t)B * cdiv(npairs, 256) * G * sizeof(float);
}

extern "C" int rn_pair_sum_fwd(const void* HL, int ldh, float* xg, void* ws, int dtype, int B, int npairs, int G,
                               void* stream) {
  return segsum_launch(HL, ldh, xg, ws, dtype, B, npairs, G, (hipStream_t)stream, "rn_pair_sum_fwd");
}

// ------------------------------------------ backward of the pair sum + last ReLU gate
template <typename T>
__global__ __launch_bounds__(256) void pair_sum_bwd_kernel(const float* __restrict__ dxg, const T* __restrict__ HL,
                                                           int ldh, T* __restrict__ dZ, int lddz, int npairs, int G,
                                                           long total) {
  constexpr int CH = Elem<T>::kPer16B;
  const int cpr = G / CH;
  for (long g = (long)blockIdx.x * 256 + threadIdx.x; g < total; g += (long)gridDim.x * 256) {
    const long row = g / cpr;
    const int c = (int)(g - row * cpr);
    const int b = (int)(row / npairs);
    const Chunk16<T> h = *reinterpret_cast<const Chunk16<T>*>(HL + row * ldh + c * CH);
    Chunk16<T> o;
#pragma unroll
    for (int e = 0; e < CH; ++e) {
      const float gv = dxg[(long)b * G + c * CH + e];
      o.v[e] = Elem<T>::from_f32(is_pos<T>(h.v[e]) ? gv : 0.f);
    }
    *reinterpret_cast<Chunk16<T>*>(dZ + row * lddz + c * CH) = o;
  }
}

extern "C" int rn_pair_sum_bwd(const float* dxg, const void* HL, int ldh, void* dZ, int lddz, int dtype, int B,
                               int npairs, int G, void* stream) {
  RN_CHECK_ARG(dxg && HL && dZ && B > 0 && npairs > 0, "rn_pair_sum_bwd: bad pointer/size");
  RN_CHECK_ARG(dtype == RN_BF16 || dtype == RN_F32, "rn_pair_sum_bwd: bad dtype %d", dtype);
  const int CH = dtype == RN_BF16 ? 8 : 4;
  RN_CHECK_ARG(G % CH == 0 && ldh % CH == 0 && lddz % CH == 0, "rn_pair_sum_bwd: G/ld must be multiples of %d", CH);
  const long total = (long)B * npairs * (G / CH);
  int blocks = cdiv(total, 256 * 4);
  if (blocks < 1) blocks = 1;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == RN_BF16)
    pair_sum_bwd_kernel<bf16><<<blocks, 256, 0, s>>>(dxg, (const bf16*)HL, ldh, (bf16*)dZ, lddz, npairs, G, total);
  else
    pair_sum_bwd_kernel<float><<<blocks, 256, 0, s>>>(dxg, (const float*)HL, ldh, (float*)dZ, lddz, npairs, G, total);
  RN_LAUNCH_CHECK("rn_pair_sum_bwd");
  return 0;
}

// ---- x_g from the two-rows-per-tile partials of the padded-j forward chain (rn_g_chain_fwd_rr_f16s_alg0, njp > n)
// Block = question; thread = (tile lane of 4, feature): a lane adds every fourth tile of the question, four loads in flight, and the
// four lanes meet in LDS in lane order -- a fixed summation order.  (Round 5: one thread per feature walking all ~172 tiles of a
// 14 x 14 question with a 64-bit division per step was 172 dependent round trips -- 82 us between the forward chain and f_phi.)
__global__ __launch_bounds__(1024) void pair_sum_tiles_kernel(const float* __restrict__ part, float* __restrict__ xg, long rpq, int G) {
  __shared__ float red[4][256];
  const int b = blockIdx.x, tl = threadIdx.x >> 8;
  const long r0 = (long)b * rpq;                                                        // first pair row of question b
  const long t0 = r0 / 256, t1 = (r0 + rpq - 1) / 256;                                  // tiles that hold rows of question b
  for (int f0 = 0; f0 < G; f0 += 256) {
    const int f = f0 + (threadIdx.x & 255);
    float acc = 0.f;
    if (f < G) {
      // a tile's row 0 belongs to the question of its FIRST pair row, row 1 to the next one: question b owns row 0 of the tiles
      // that start inside it and row 1 of the one that starts before it
      auto row = [&](long t) { return (2 * t + (t * 256 >= r0 ? 0 : 1)) * G + f; };
      long t = t0 + tl;
      for (; t + 12 <= t1; t += 16) {
        const float a0 = part[row(t)], a1 = part[row(t + 4)], a2 = part[row(t + 8)], a3 = part[row(t + 12)];
        acc += (a0 + a1) + (a2 + a3);
      }
      for (; t <= t1; t += 4) acc += part[row(t)];
    }
    __syncthreads();
    red[tl][threadIdx.x & 255] = acc;
    __syncthreads();
    if (tl == 0 && f < G) xg[(long)b * G + f] = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
  }
}

extern "C" int rn_pair_sum_tiles(const float* part, float* xg, int M, int rows_per_question, int G, void* stream) {
  RN_CHECK_ARG(part && xg && M > 0 && rows_per_question >= 256 && M % rows_per_question == 0 && M % 256 == 0 && G > 0,
               "rn_pair_sum_tiles: needs M %% 256 == 0 and rows_per_question >= 256 dividing M (M=%d, rows_per_question=%d)", M, rows_per_question);
  pair_sum_tiles_kernel<<<M / rows_per_question, 1024, 0, (hipStream_t)stream>>>(part, xg, rows_per_question, G);
  RN_LAUNCH_CHECK("rn_pair_sum_tiles");
  return 0;
}

// ---------------------------------------- backward of the pair expansion (reductions)
// Rj[b,j,:] = sum_i dZ[(b,i,j),:]   Ri[b,i,:] = sum_j dZ[(b,i,j),:]   Rq[b,:] = sum_i Ri[b,i,:]
// ONE pass over dZ (HBM-bound: it is read exactly once).  Workgroup = (block of 16 j, question b); thread =
// (i-lane il, 16-byte column chunk c) and walks i = il, il + NIL, ...: for every i it loads its 16 rows
// (j0 .. j0+15; 16 independent 16-byte loads in flight), adds them into Rj accumulators kept in registers for
// the whole walk and into the Ri partial of that i, which is complete (for this j block) after the 16 rows and
// is written straight out.  Cross-thread traffic only at the end: the NIL i-lanes combine their Rj sums through
// LDS in a fixed order -> deterministic.  A tiny finish kernel adds the j-block partials of Ri and forms Rq.
template <typename T> struct Piece4;                                   // 4 consecutive elements of a row
template <> struct Piece4<bf16> { typedef u32x2 Raw; static __device__ __forceinline__ void unpack(const Raw& r, float (&x)[4]) {
  x[0] = __builtin_bit_cast(float, r[0] << 16); x[1] = __builtin_bit_cast(float, r[0] & 0xffff0000u);
  x[2] = __builtin_bit_cast(float, r[1] << 16); x[3] = __builtin_bit_cast(float, r[1] & 0xffff0000u); } };
template <> struct Piece4<float> { typedef f32x4 Raw; static __device__ __forceinline__ void unpack(const Raw& r, float (&x)[4]) {
  x[0] = r[0]; x[1] = r[1]; x[2] = r[2]; x[3] = r[3]; } };

template <typename T, bool WANT_RJ>
__global__ __launch_bounds__(256) void pair_reduce_kernel(const T* __restrict__ dZ, int ld, float* __restrict__ Rj,
                                                          float* __restrict__ ri_part, int n, int njp, int G, long part_stride) {
  constexpr int CH = 4;                                               // columns per thread: 64 Rj accumulators, 16 small loads in flight
  constexpr int JB = 16;
  typedef typename Piece4<T>::Raw Raw;
  extern __shared__ __attribute__((aligned(16))) float red[];        // [JB][G] fp32 (Rj hand-over, one i-lane per round)
  const int cpr = G / CH;                                             // threads per row (<= 256)
  const int NIL = 256 / cpr;                                          // i-lanes
  const int t = threadIdx.x, c = t % cpr, il = t / cpr;
  const int b = blockIdx.y, jb = blockIdx.x, j0 = jb * JB;
  const int nj = (n - j0) < JB ? (n - j0) : JB;
  float rj[WANT_RJ ? JB : 1][CH];
  if constexpr (WANT_RJ) {
#pragma unroll
    for (int j = 0; j < JB; ++j)
#pragma unroll
      for (int e = 0; e < CH; ++e) rj[j][e] = 0.f;
  }
  if (il < NIL) {
    // Software pipeline over i: the 16 loads of the NEXT i are in flight while this i is added up (one workgroup per
    // CU: nobody else hides the HBM latency).  All 16 loads are issued back to back -- a short last block re-reads
    // its last row and adds zeros: no branch may sit between the loads.
    constexpr int PD = 3;                                           // i's in flight per thread; <= 256 registers so that the workgroup fits NEXT TO a wgrad workgroup
    Raw v[PD][JB];
    auto issue = [&](Raw (&dst)[JB], int i) {
      const T* base = dZ + (((long)b * n + i) * njp + j0) * ld + c * CH;      // (njp pair rows per (b, i) group; the first n are read)
#pragma unroll
      for (int j = 0; j < JB; ++j) dst[j] = *reinterpret_cast<const Raw*>(base + (long)(j < nj ? j : nj - 1) * ld);
    };
    auto consume = [&](const Raw (&src)[JB], int i) {
      float ri[CH] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < JB; ++j) {
        const float keep = j < nj ? 1.f : 0.f;
        float x[4];
        Piece4<T>::unpack(src[j], x);
#pragma unroll
        for (int e = 0; e < CH; ++e) {
          ri[e] += x[e] * keep;
          if constexpr (WANT_RJ) rj[j][e] += x[e] * keep;
        }
      }
      *reinterpret_cast<f32x4*>(ri_part + jb * part_stride + ((long)b * n + i) * G + c * CH) = f32x4{ri[0], ri[1], ri[2], ri[3]};
    };
    // ring of PD stages: stage s holds i = il + (s + PD * round) * NIL; PD - 1 stages are in flight while one is consumed
#pragma unroll
    for (int s0 = 0; s0 < PD - 1; ++s0)
      if (il + s0 * NIL < n) issue(v[s0], il + s0 * NIL);
    for (int ib = il; ib < n; ib += PD * NIL) {
#pragma unroll
      for (int s0 = 0; s0 < PD; ++s0) {
        const int i = ib + s0 * NIL, inext = i + (PD - 1) * NIL;
        if (i < n) {
          if (inext < n) issue(v[(s0 + PD - 1) % PD], inext);
          __builtin_amdgcn_sched_barrier(0);
          consume(v[s0], i);
        }
      }
    }
  }
  if constexpr (WANT_RJ) {
    // i-lanes 1 .. NIL-1 hand their sums to lane 0 through LDS, one lane per round through ONE (JB x G) fp32 buffer:
    // 16 KB instead of (NIL-1) x 16 KB keeps the workgroup resident NEXT TO a wgrad workgroup (128 KB of the CU's
    // 160 KB) -- with the 48-KB version this kernel only ran in the gaps between the wgrad launches of the side stream.
    for (int l = 1; l < NIL; ++l) {
      if (il == l) {
#pragma unroll
        for (int j = 0; j < JB; ++j)
          *reinterpret_cast<f32x4*>(red + (long)j * G + c * CH) = f32x4{rj[j][0], rj[j][1], rj[j][2], rj[j][3]};
      }
      __syncthreads();
      if (il == 0) {
#pragma unroll
        for (int j = 0; j < JB; ++j) {
          const f32x4 r = *reinterpret_cast<const f32x4*>(red + (long)j * G + c * CH);
          rj[j][0] += r[0]; rj[j][1] += r[1]; rj[j][2] += r[2]; rj[j][3] += r[3];
        }
      }
      __syncthreads();
    }
    if (il == 0) {
#pragma unroll
      for (int j = 0; j < JB; ++j)
        if (j < nj) *reinterpret_cast<f32x4*>(Rj + ((long)b * n + j0 + j) * G + c * CH) = f32x4{rj[j][0], rj[j][1], rj[j][2], rj[j][3]};
    }
  }
}

// Ri = sum over the j-block partial slabs; Rq[b] = sum_i Ri[b,i].  Block = (question b, group of 64 float4 columns);
// thread = (i-lane of 4, float4 column): walks i = lane, lane + 4, ... (fixed order), the 4 i-lanes combine their Rq
// partials through LDS -> deterministic.
__global__ __launch_bounds__(256) void pair_reduce_finish_kernel(const f32x4* __restrict__ ri_part, long part_stride4, int njb,
                                                                 f32x4* __restrict__ Ri, f32x4* __restrict__ Rq, int n, int G4) {
  __shared__ f32x4 red[3][64];
  const int b = blockIdx.y, c = blockIdx.x * 64 + (threadIdx.x & 63), il = threadIdx.x >> 6;
  f32x4 q = {0.f, 0.f, 0.f, 0.f};
  if (c < G4) {
    // This loop is a chain of L2 / HBM round trips on the critical path of the backward pass (it runs beside the wgrad
    // stream, where a round trip costs microseconds): four i's x four slabs = 16 independent loads per trip instead of
    // one.  Every sum keeps its order (slabs ascending per i, i ascending into Rq): bitwise the sequential result.
    for (int i0 = il; i0 < n; i0 += 16) {
      long o[4];
      bool ok[4];
      f32x4 r[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + 4 * u;
        ok[u] = i < n;
        o[u] = ((long)b * n + (ok[u] ? i : i0)) * G4 + c;
      }
      int p = 0;
      for (; p + 4 <= njb; p += 4) {
        f32x4 v[4][4];
#pragma unroll
        for (int w = 0; w < 4; ++w)
#pragma unroll
          for (int u = 0; u < 4; ++u) v[w][u] = ri_part[(long)(p + w) * part_stride4 + o[u]];
#pragma unroll
        for (int w = 0; w < 4; ++w)
#pragma unroll
          for (int u = 0; u < 4; ++u) r[u] = (p == 0 && w == 0) ? v[w][u] : r[u] + v[w][u];
      }
      for (; p < njb; ++p) {
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = ri_part[(long)p * part_stride4 + o[u]];
#pragma unroll
        for (int u = 0; u < 4; ++u) r[u] = (p == 0) ? v[u] : r[u] + v[u];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (ok[u]) {
          if (Ri) Ri[o[u]] = r[u];
          q += r[u];
        }
      }
    }
  }
  if (il > 0) red[il - 1][threadIdx.x & 63] = q;
  __syncthreads();
  if (il == 0 && c < G4 && Rq) Rq[(long)b * G4 + c] = ((q + red[0][threadIdx.x]) + red[1][threadIdx.x]) + red[2][threadIdx.x];
}


size_t rnws_pair_reduce(int B, int n, int G) {
  return (size_t)cdiv(n, 16) * B * n * G * sizeof(float);            // Ri partials, one slab per block of 16 j
}

extern "C" int rn_pair_reduce_bwd(const void* dZ, int lddz, float* Rj, float* Ri, float* Rq, void* ws, int dtype, int B,
                                  int n, int njp, int G, void* stream) {
  RN_CHECK_ARG(dZ && B > 0 && n > 0 && njp >= n && ws, "rn_pair_reduce_bwd: bad pointer/size (njp=%d must be >= n=%d)", njp, n);
  RN_CHECK_ARG(dtype == RN_BF16 || dtype == RN_F32, "rn_pair_reduce_bwd: bad dtype %d", dtype);
  const int CH = 4;
  RN_CHECK_ARG(G % CH == 0 && G / CH <= 256 && 256 % (G / CH) == 0 && lddz % (dtype == RN_BF16 ? 8 : 4) == 0, "rn_pair_reduce_bwd: G=%d unsupported", G);
  RN_CHECK_ARG(((uintptr_t)dZ | (uintptr_t)Rj | (uintptr_t)Ri | (uintptr_t)Rq | (uintptr_t)ws) % 16 == 0, "rn_pair_reduce_bwd: pointers must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  const int njb = cdiv(n, 16);
  const long part_stride = (long)B * n * G;
  float* part = (float*)ws;
  dim3 grid(njb, B);
  const size_t shm = Rj ? (size_t)16 * G * sizeof(float) : 0;
  RN_CHECK_ARG(shm <= 160 * 1024, "rn_pair_reduce_bwd: G=%d needs %zu bytes of LDS", G, shm);
#define RN_PR(T, RJ)                                                                                                   \
  do {                                                                                                                 \
    if (shm > 64 * 1024) (void)hipFuncSetAttribute((const void*)pair_reduce_kernel<T, RJ>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm); \
    pair_reduce_kernel<T, RJ><<<grid, 256, shm, s>>>((const T*)dZ, lddz, Rj, part, n, njp, G, part_stride);            \
  } while (0)
  if (dtype == RN_BF16) { if (Rj) RN_PR(bf16, true); else RN_PR(bf16, false); }
  else { if (Rj) RN_PR(float, true); else RN_PR(float, false); }
#undef RN_PR
  RN_LAUNCH_CHECK("rn_pair_reduce_bwd");
  if (Ri || Rq) {
    RN_CHECK_ARG(G % 4 == 0, "rn_pair_reduce_bwd: G=%d must be a multiple of 4", G);
    const int G4 = G / 4;
    pair_reduce_finish_kernel<<<dim3(cdiv(G4, 64), B), 256, 0, s>>>((const f32x4*)part, part_stride / 4, njb, (f32x4*)Ri, (f32x4*)Rq, n, G4);
    RN_LAUNCH_CHECK("rn_pair_reduce_bwd(finish)");
  }
  return 0;
}

// ---- the same three reductions from the partials of the backward chain that reduces on chip (rn_g_chain_bwd_rr_red):
//   rj_part (records, 32, G): unit ((b * n/32 + jg) * nu + u) = sum over the unit's i of dZ_0[(b, i, 32 jg + j), :] -- one record for
//            a unit whose walk position p = (u * njp/32 + jg) * B + b is < units_whole; a later unit left one record per tile: tile
//            0 in its own record, tile t > 0 in record nunits + (p - units_whole) (tpu - 1) + t - 1  (the balanced tail of
//            rn_g_chain_bwd_rr_red; question fastest in p: every question carries the same share of the tail)
//   ri_part (B * n * n/16, G):      rows ((b * n + i) * n/32 + jg) * 2 + {0, 1} = sums over the 16 + 16 j of a wave's lane halves
// Block = (question, 128 columns); thread = (row lane of 32, float4 column): rows r = lane, lane + 32, ... serve as i for Ri and as
// j for Rj; the Rq partials of the row lanes meet in LDS and are added in lane order -- every sum has a fixed order.
__global__ __launch_bounds__(1024) void pair_reduce_parts_kernel(const f32x4* __restrict__ rj_part, const f32x4* __restrict__ ri_part,
                                                                 f32x4* __restrict__ Rj, f32x4* __restrict__ Ri, f32x4* __restrict__ Rq,
                                                                 int n, int G4, int nu, int njp, int tpu, int units_whole, int nunits, int nb) {
  __shared__ f32x4 red[32][32];
  const int b = blockIdx.x, c = blockIdx.y * 32 + (threadIdx.x & 31), rl = threadIdx.x >> 5;
  const int jgs = njp / 16;                                           // partial rows per (question, i) (padded j axis: njp > n)
  f32x4 q = {0.f, 0.f, 0.f, 0.f};
  if (c < G4) {
    for (int r = rl; r < n; r += 32) {
      const f32x4* pi = ri_part + ((long)b * n + r) * jgs * G4 + c;
      const int u0 = (b * (njp / 32) + (r >> 5)) * nu;                // first unit of (question, j block)
      const f32x4* pj = rj_part + (long)(r & 31) * G4 + c;
      const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
      // eight loads in flight per step (a padded 14 x 14 question is 14 + 5 partial rows per output row: one load per round trip
      // made this kernel 48 us alone at that shape); the order of the adds is fixed
      f32x4 si = z4, sj = z4;
      for (int g0 = 0; g0 < jgs; g0 += 8) {
        f32x4 v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = g0 + e < jgs ? pi[(long)(g0 + e) * G4] : z4;
        si += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
      }
      for (int v0 = 0; v0 < nu; v0 += 8) {
        f32x4 v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = v0 + e < nu ? pj[(long)(u0 + v0 + e) * 32 * G4] : z4;
        sj += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
      }
      for (int u = 0; u < nu; ++u) {                                  // the balanced tail's extra records (a few units per question)
        const int p = rn_red_walk_pos(b, r >> 5, u, nb, njp / 32);    // walk position of the unit (rn_common.h: shared with the writer)
        if (p >= units_whole)
          for (int t = 1; t < tpu; ++t) sj += pj[rn_red_extra_rec(p, t, nunits, tpu, units_whole) * 32 * G4];
      }
      Ri[((long)b * n + r) * G4 + c] = si;
      Rj[((long)b * n + r) * G4 + c] = sj;
      q += si;
    }
  }
  red[rl][threadIdx.x & 31] = q;
  __syncthreads();
  if (rl == 0 && c < G4 && Rq) {
    f32x4 t = red[0][threadIdx.x];
#pragma unroll
    for (int l = 1; l < 32; ++l) t += red[l][threadIdx.x];
    Rq[(long)b * G4 + c] = t;
  }
}

extern "C" int rn_pair_reduce_parts(const float* rj_part, const float* ri_part, float* Rj, float* Ri, float* Rq, int B, int n, int njp,
                                    int G, int nu, int tiles_per_unit, int units_whole, void* stream) {
  RN_CHECK_ARG(rj_part && ri_part && Rj && Ri && B > 0 && n > 0 && njp >= n && njp % 32 == 0 && nu > 0 && G % 4 == 0,
               "rn_pair_reduce_parts: bad pointer/size (B=%d n=%d njp=%d G=%d nu=%d; njp %% 32 == 0)", B, n, njp, G, nu);
  RN_CHECK_ARG(((uintptr_t)rj_part | (uintptr_t)ri_part | (uintptr_t)Rj | (uintptr_t)Ri | (uintptr_t)Rq) % 16 == 0, "rn_pair_reduce_parts: pointers must be 16-byte aligned");
  const int G4 = G / 4, nunits = B * (njp / 32) * nu;
  RN_CHECK_ARG(tiles_per_unit > 0 && units_whole >= 0 && units_whole <= nunits,
               "rn_pair_reduce_parts: tiles_per_unit=%d, units_whole=%d (of %d units)", tiles_per_unit, units_whole, nunits);
  pair_reduce_parts_kernel<<<dim3(B, cdiv(G4, 32)), 1024, 0, (hipStream_t)stream>>>((const f32x4*)rj_part, (const f32x4*)ri_part, (f32x4*)Rj,
                                                                                     (f32x4*)Ri, (f32x4*)Rq, n, G4, nu, njp, tiles_per_unit, units_whole, nunits, B);
  RN_LAUNCH_CHECK("rn_pair_reduce_parts");
  return 0;
}

// ------------------------------------------ layer-0 weight gradient from the pair reductions
// Layer 0 reads the pair matrix P = [x_j | x_i | q]; its weight gradient dZ_0^T P therefore factors through the
// reductions the input gradient needs anyway:
//   dW0[:, 0:k] = Rj^T X,  dW0[:, k:2k] = Ri^T X,  dW0[:, 2k:2k+Q] = Rq^T Q,  db0 = sum_b Rq[b]
// with X = x as a (B*n, k) matrix -- 55 MFLOP on (B*n)-row fp32 matrices instead of a 235 MB pass over dZ_0 and P.
// Kernel 1: workgroup (feature block of 32, part j|i, row split): partial (32 x k) products over 256 rows, X tile in LDS;
// kernel 2: fixed-order sum of the row splits + the (tiny) question part and the bias.  Deterministic.
namespace {
constexpr int W0_RS = 192;          // rows per split (24 KB of LDS: fits next to a 128-KB wgrad workgroup)
constexpr int W0_CMAX = 32;         // k <= 32 columns per x part
}
__global__ __launch_bounds__(256) void wgrad0_part_kernel(const float* __restrict__ Rj, const float* __restrict__ Ri,
                                                          const float* __restrict__ x, long sxb, long sxn, long sxk,
                                                          float* __restrict__ part, int n, int k, int N, int rows,
                                                          const float* __restrict__ coord, int kf) {
  __shared__ float xs[W0_RS][W0_CMAX + 1];
  const int fb = blockIdx.x, pt = blockIdx.y, ks = blockIdx.z;
  const int t = threadIdx.x, f = fb * 32 + (t & 31), cg = t >> 5;
  const int r0 = ks * W0_RS;
  const int nr = (rows - r0) < W0_RS ? (rows - r0) : W0_RS;
  for (int i = t; i < W0_RS * k; i += 256) {
    const int r = i / k, c = i - r * k;
    float v = 0.f;
    if (r < nr) {
      const int row = r0 + r, b = row / n, j = row - b * n;
      v = c < kf ? x[b * sxb + j * sxn + c * sxk] : coord[(long)(c - kf) * n + j];
    }
    xs[r][c] = v;
  }
  __syncthreads();
  const float* R = (pt ? Ri : Rj) + (long)r0 * N + f;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  int r = 0;
  // the loop is a chain of L2 round trips: 32 independent loads in flight per trip (the sums stay in row order)
  for (; r + 32 <= nr; r += 32) {
    float v[32];
#pragma unroll
    for (int u = 0; u < 32; ++u) v[u] = R[(long)(r + u) * N];
#pragma unroll
    for (int u = 0; u < 32; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] = fmaf(v[u], xs[r + u][cg + 8 * e], acc[e]);
  }
  for (; r + 8 <= nr; r += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = R[(long)(r + u) * N];
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] = fmaf(v[u], xs[r + u][cg + 8 * e], acc[e]);
  }
  for (; r < nr; ++r) {
    const float v = R[(long)r * N];
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] = fmaf(v, xs[r][cg + 8 * e], acc[e]);
  }
  // part[ks][pt][f][c]
  float* o = part + (((long)ks * 2 + pt) * N + f) * W0_CMAX;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[cg + 8 * e] = acc[e];
}

// block = feature f; thread = output column of dW0 (2k + Q of them) or the bias
__global__ __launch_bounds__(256) void wgrad0_finish_kernel(const float* __restrict__ part, int nks, const float* __restrict__ Rq,
                                                            const float* __restrict__ q, long sqb, float* __restrict__ dW0,
                                                            float* __restrict__ db0, int B, int k, int Q, int N, int kt) {
  // Rq[:, f] is what every question-column thread (and the bias thread) of this block needs: staged once in LDS; the
  // per-thread loops then issue 16 independent loads per L2 round trip instead of one (all sums keep their order)
  __shared__ float rq_s[1024];
  const int f = blockIdx.x;
  for (int b0 = 0; b0 < (Rq ? B : 0); b0 += 1024) {
    const int nb = B - b0 < 1024 ? B - b0 : 1024;
    __syncthreads();
    for (int b = threadIdx.x; b < nb; b += 256) rq_s[b] = Rq[(long)(b0 + b) * N + f];
    __syncthreads();
    for (int c = 2 * k + threadIdx.x; c <= kt; c += 256) {
      float s = (b0 == 0) ? 0.f : (c == kt ? db0[f] : dW0[(long)f * kt + c]);
      if (c == kt) {
        for (int b = 0; b < nb; ++b) s += rq_s[b];
        if (db0) db0[f] = s;
      } else {
        const float* qp = q + (long)b0 * sqb + (c - 2 * k);
        int b = 0;
        for (; b + 16 <= nb; b += 16) {
          float v[16];
#pragma unroll
          for (int u = 0; u < 16; ++u) v[u] = qp[(long)(b + u) * sqb];
#pragma unroll
          for (int u = 0; u < 16; ++u) s = fmaf(rq_s[b + u], v[u], s);
        }
        for (; b < nb; ++b) s = fmaf(rq_s[b], qp[(long)b * sqb], s);
        dW0[(long)f * kt + c] = s;
      }
    }
  }
  if (!Rq && threadIdx.x == 0 && db0) db0[f] = 0.f;
  for (int c = threadIdx.x; c < 2 * k; c += 256) {
    const int pt = c / k, cc = c - pt * k;
    const float* pp = part + ((long)pt * N + f) * W0_CMAX + cc;
    const long zs = (long)2 * N * W0_CMAX;
    float s = 0.f;
    int z = 0;
    for (; z + 8 <= nks; z += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = pp[(z + u) * zs];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; z < nks; ++z) s += pp[z * zs];
    dW0[(long)f * kt + c] = s;
  }
}

size_t rnws_wgrad0(int B, int n, int N) {
  return (size_t)cdiv((long)B * n, W0_RS) * 2 * N * W0_CMAX * sizeof(float);
}

extern "C" int rn_wgrad0_from_reductions(const float* Rj, const float* Ri, const float* Rq, const float* x, long sxb, long sxn,
                                         long sxk, const float* coord, int kf, const float* q, long sqb, float* dW0, float* db0,
                                         void* ws, int B, int n, int k, int Q, int N, void* stream) {
  if (!coord) kf = k;
  RN_CHECK_ARG(kf > 0 && kf <= k, "rn_wgrad0_from_reductions: kf=%d must be in (0, k=%d]", kf, k);
  RN_CHECK_ARG(Rj && Ri && x && dW0 && db0 && ws && B > 0 && n > 0, "rn_wgrad0_from_reductions: bad pointer/size");
  RN_CHECK_ARG(k > 0 && k <= W0_CMAX && N % 32 == 0 && (Q == 0 || (Rq && q)), "rn_wgrad0_from_reductions: k=%d (<= %d), N=%d (%% 32), Q=%d unsupported", k, W0_CMAX, N, Q);
  RN_CHECK_ARG(Rq || Q == 0, "rn_wgrad0_from_reductions: Rq is required for the bias when the question is injected");
  const int rows = B * n, nks = cdiv(rows, W0_RS), kt = 2 * k + Q;
  hipStream_t s = (hipStream_t)stream;
  wgrad0_part_kernel<<<dim3(N / 32, 2, nks), 256, 0, s>>>(Rj, Ri, x, sxb, sxn, sxk, (float*)ws, n, k, N, rows, coord, kf);
  wgrad0_finish_kernel<<<N, 256, 0, s>>>((const float*)ws, nks, Rq, q, sqb, dW0, db0, B, k, Q, N, kt);
  RN_LAUNCH_CHECK("rn_wgrad0_from_reductions");
  return 0;
}

// ------------------------------------------ R-CBIR pair features (extract.py:60-71)
// For the INPUT of a g layer, A (B*npairs, lda), first F columns: L2-normalise every pair row (F.normalize, eps 1e-12),
// then the maximum and the mean over the npairs rows of every question -- one pass over A, nothing is materialised in
// fp32.  A wave walks rows (lane = F/64 consecutive columns: one coalesced row read), the row norm is a wave all-reduce;
// per-(question, slice) partials are combined by a finish kernel in a fixed order.
template <typename T, int VPL>   // VPL columns per lane (F = 64 * VPL)
__global__ __launch_bounds__(256) void pair_features_kernel(const T* __restrict__ A, int lda, float* __restrict__ pmax,
                                                            float* __restrict__ psum, int npairs, int S) {
  __shared__ float red[2][4][64 * VPL];
  const int b = blockIdx.y, s = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int rows_per = (npairs + S - 1) / S;
  const int r0 = s * rows_per, r1 = (r0 + rows_per) < npairs ? (r0 + rows_per) : npairs;
  float mx[VPL], sm[VPL];
#pragma unroll
  for (int e = 0; e < VPL; ++e) { mx[e] = -3.0e38f; sm[e] = 0.f; }
  for (int r = r0 + w; r < r1; r += 4) {
    const T* row = A + ((long)b * npairs + r) * lda + lane * VPL;
    float v[VPL], ss = 0.f;
#pragma unroll
    for (int e = 0; e < VPL; ++e) { v[e] = Elem<T>::to_f32(row[e]); ss = fmaf(v[e], v[e], ss); }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) ss += __shfl_xor(ss, o);
    const float inv = 1.f / fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
    for (int e = 0; e < VPL; ++e) { const float u = v[e] * inv; mx[e] = fmaxf(mx[e], u); sm[e] += u; }
  }
#pragma unroll
  for (int e = 0; e < VPL; ++e) { red[0][w][lane * VPL + e] = mx[e]; red[1][w][lane * VPL + e] = sm[e]; }
  __syncthreads();
  for (int c = threadIdx.x; c < 64 * VPL; c += 256) {
    const float m = fmaxf(fmaxf(red[0][0][c], red[0][1][c]), fmaxf(red[0][2][c], red[0][3][c]));
    const float t = (red[1][0][c] + red[1][1][c]) + (red[1][2][c] + red[1][3][c]);
    pmax[((long)b * S + s) * 64 * VPL + c] = m;
    psum[((long)b * S + s) * 64 * VPL + c] = t;
  }
}
__global__ __launch_bounds__(256) void pair_features_finish_kernel(const float* __restrict__ pmax, const float* __restrict__ psum,
                                                                   float* __restrict__ maxf, float* __restrict__ avgf, int F, int S,
                                                                   float inv_n) {
  const int b = blockIdx.x;
  for (int c = threadIdx.x; c < F; c += 256) {
    float m = -3.0e38f, t = 0.f;
    for (int s = 0; s < S; ++s) {
      m = fmaxf(m, pmax[((long)b * S + s) * F + c]);
      t += psum[((long)b * S + s) * F + c];
    }
    maxf[(long)b * F + c] = m;
    avgf[(long)b * F + c] = t * inv_n;
  }
}

static int pf_slices(int npairs) {
  int s = npairs / 64;
  return s < 1 ? 1 : (s > 64 ? 64 : s);
}
size_t rnws_pair_features(int B, int npairs, int F) { return (size_t)2 * B * pf_slices(npairs) * F * sizeof(float); }

extern "C" int rn_pair_features(const void* A, int lda, int F, float* maxf, float* avgf, void* ws, int dtype, int B, int npairs,
                                void* stream) {
  RN_CHECK_ARG(A && maxf && avgf && ws && B > 0 && npairs > 0, "rn_pair_features: bad pointer/size");
  RN_CHECK_ARG(dtype == RN_BF16 || dtype == RN_F32, "rn_pair_features: bad dtype %d", dtype);
  RN_CHECK_ARG(F > 0 && F % 64 == 0 && F <= 512 && lda >= F, "rn_pair_features: F=%d must be a multiple of 64, <= 512 and <= lda=%d", F, lda);
  const int S = pf_slices(npairs), vpl = F / 64;
  float* pmax = (float*)ws;
  float* psum = pmax + (size_t)B * S * F;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(S, B);
#define RN_PF(T, V) pair_features_kernel<T, V><<<grid, 256, 0, s>>>((const T*)A, lda, pmax, psum, npairs, S)
#define RN_PFV(T)                                                                                                     \
  switch (vpl) { case 1: RN_PF(T, 1); break; case 2: RN_PF(T, 2); break; case 3: RN_PF(T, 3); break; case 4: RN_PF(T, 4); break; \
                 case 5: RN_PF(T, 5); break; case 6: RN_PF(T, 6); break; case 7: RN_PF(T, 7); break; default: RN_PF(T, 8); break; }
  if (dtype == RN_BF16) { RN_PFV(bf16) } else { RN_PFV(float) }
#undef RN_PFV
#undef RN_PF
  pair_features_finish_kernel<<<B, 256, 0, s>>>(pmax, psum, maxf, avgf, F, S, 1.f / (float)npairs);
  RN_LAUNCH_CHECK("rn_pair_features");
  return 0;
}

// ------------------------------------------ input gradients of the pair expansion from the reductions
// dx[b,j,:] = Rj[b,j,:] W0[:, 0:k] + Ri[b,j,:] W0[:, k:2k]     dq[b,:] = Rq[b,:] W0[:, 2k:2k+Q]      (W0: (N, kt) row-major)
// One launch instead of three small GEMMs on the critical path of the backward pass (dx feeds the conv stack, dq the
// question encoder), on the fp32 matrix pipe (v_mfma_f32_16x16x4_f32: exact fp32 products, fp32 accumulate).  Workgroup = 4 waves
// = a 16-row x 32-column output tile (x blocks: both products; q blocks: one); wave w takes every fourth group of 16 features, its
// operands straight from global memory in the MFMA's own lane layout (A: lane = (row, k), B: lane = (column, k) -- no LDS in the
// loop), all of a wave's loads in flight at once; the four waves' partial tiles meet in LDS and are added in wave order
// (deterministic).  Round 5's version -- a thread per output, both operands through LDS, two ds_read_b32 per FMA on 8 waves -- was
// LDS-issue bound: 18 us alone for 0.2 GFLOP on the critical path; a register-tiled VALU version (4 rows per thread, ds_read_b128)
// measured 24.6 us.
__global__ __launch_bounds__(256) void pair_dx_dq_kernel(const float* __restrict__ Rj, const float* __restrict__ Ri,
                                                         const float* __restrict__ Rq, const float* __restrict__ W0, int kt,
                                                         float* __restrict__ dx, float* __restrict__ dq, int rows, int B, int k,
                                                         int Q, int N, int xblocks, int qcol_tiles, int n, long sdb, long sdn,
                                                         long sdk, int kout) {
  __shared__ float red[4][2][4][64];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const float *R1, *R2;
  float* out;
  int off1, off2, ncols, nrows, r0, ld;
  if ((int)blockIdx.x < xblocks) {
    R1 = Rj; R2 = Ri; off1 = 0; off2 = k; ncols = k; nrows = rows; r0 = blockIdx.x * 16; out = dx; ld = k;
  } else {
    const int qb = blockIdx.x - xblocks, ct = qb % qcol_tiles;
    R1 = Rq; R2 = nullptr; off1 = 2 * k + 32 * ct; off2 = 0; ncols = min(32, Q - 32 * ct); nrows = B; r0 = (qb / qcol_tiles) * 16;
    out = dq + 32 * ct; ld = Q;
  }
  const bool two = R2 != nullptr;
  const int li = lane & 15, kq = lane >> 4;                              // operand lane: (row | column li, k = kq)
  const bool row_ok = r0 + li < nrows, c0_ok = li < ncols, c1_ok = 16 + li < ncols;
  const float* a1p = R1 + (long)(row_ok ? r0 + li : 0) * N;
  const float* a2p = (two ? R2 : R1) + (long)(row_ok ? r0 + li : 0) * N;
  f32x4 acc[2][2];                                                       // [operand][column tile]
#pragma unroll
  for (int o = 0; o < 2; ++o)
#pragma unroll
    for (int c = 0; c < 2; ++c) acc[o][c] = f32x4{0.f, 0.f, 0.f, 0.f};
  // groups of 16 features: MFMA i of a group multiplies features g + 4 kq + i (both operands agree; any order is a valid sum).
  // FOUR groups per trip: 4 x (2 x 4 + 4 x 4) = 96 loads of a lane in flight before the first MFMA.
  constexpr int GPT = 4;
  for (int g0 = 16 * w * GPT; g0 < N; g0 += 64 * GPT) {
    float xa[GPT][2][4], wb[GPT][2][2][4];
#pragma unroll
    for (int gi = 0; gi < GPT; ++gi) {
      const int g = g0 + 16 * gi;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int f = g + 4 * kq + i;
        const bool fok = f < N;
        xa[gi][0][i] = (fok && row_ok) ? a1p[f] : 0.f;
        xa[gi][1][i] = (fok && row_ok && two) ? a2p[f] : 0.f;
        const float* wr = W0 + (long)(fok ? f : 0) * kt;
        wb[gi][0][0][i] = (fok && c0_ok) ? wr[off1 + li] : 0.f;
        wb[gi][0][1][i] = (fok && c1_ok) ? wr[off1 + 16 + li] : 0.f;
        wb[gi][1][0][i] = (fok && c0_ok && two) ? wr[off2 + li] : 0.f;
        wb[gi][1][1][i] = (fok && c1_ok && two) ? wr[off2 + 16 + li] : 0.f;
      }
    }
#pragma unroll
    for (int gi = 0; gi < GPT; ++gi) {
      if (g0 + 16 * gi >= N) break;                                      // (wave-uniform)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[gi][0][i], wb[gi][0][0][i], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[gi][0][i], wb[gi][0][1][i], acc[0][1], 0, 0, 0);
        if (two) {
          acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[gi][1][i], wb[gi][1][0][i], acc[1][0], 0, 0, 0);
          acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[gi][1][i], wb[gi][1][1][i], acc[1][1], 0, 0, 0);
        }
      }
    }
  }
  // D layout: register r of lane l = output (row 4 (l >> 4) + r, column l & 15) of the tile
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[w][c][r][lane] = acc[0][c][r] + acc[1][c][r];
  __syncthreads();
  for (int o = t; o < 512; o += 256) {                                   // output o: row o / 32, column o % 32
    const int ro = o >> 5, c = o & 31, ct = c >> 4, l = (c & 15) + 16 * (ro >> 2), r = ro & 3;
    const float v = ((red[0][ct][r][l] + red[1][ct][r][l]) + red[2][ct][r][l]) + red[3][ct][r][l];
    const int row = r0 + ro;
    if (c < ncols && row < nrows) {
      if ((int)blockIdx.x < xblocks) {                                   // dx[b, j, c] at element strides (sdb, sdn, sdk); columns >= kout
        const int b = row / n, j = row - b * n;                          // (the coordinate tags: no gradient, model.py:216) are dropped
        if (c < kout) dx[b * sdb + j * sdn + c * sdk] = v;
      } else {
        out[(long)row * ld + c] = v;
      }
    }
  }
}

extern "C" int rn_pair_dx_dq(const float* Rj, const float* Ri, const float* Rq, const float* W0, float* dx, long sdb, long sdn, long sdk,
                             int kout, float* dq, int B, int n, int k, int Q, int N, void* stream) {
  RN_CHECK_ARG(kout > 0 && kout <= k, "rn_pair_dx_dq: kout=%d must be in (0, k=%d]", kout, k);
  RN_CHECK_ARG(Rj && Ri && W0 && dx && B > 0 && n > 0 && N > 0, "rn_pair_dx_dq: bad pointer/size");
  RN_CHECK_ARG(k > 0 && k <= 32 && (Q == 0 || (Rq && dq)), "rn_pair_dx_dq: k=%d (<= 32), Q=%d unsupported", k, Q);
  const int rows = B * n, xblocks = cdiv(rows, 16), qct = cdiv(Q, 32), qblocks = Q ? cdiv(B, 16) * qct : 0;
  pair_dx_dq_kernel<<<xblocks + qblocks, 256, 0, (hipStream_t)stream>>>(Rj, Ri, Rq, W0, 2 * k + Q, dx, dq, rows, B, k, Q, N, xblocks,
                                                                         qct ? qct : 1, n, sdb, sdn, sdk, kout);
  RN_LAUNCH_CHECK("rn_pair_dx_dq");
  return 0;
}

// ------------------------------------------ tables of the factored first layer (rn_g_chain_fwd_rr_alg0)
// With the question injected at layer 0 the first g layer is W0 [x_j | x_i | q] + b0 = W0a x_j + (W0b x_i + W0c q + b0):
//   Xp[b*n + j][0:64]  = x[b, j, 0:k] as bf16, zero padded          (the only MFMA operand left, K = 64)
//   Vc[b*n + i][0:N]   = b0 + W0b x[b, i] + W0c q[b]   in fp32      (constant over the pairs (i, .) of a wave: its bias row)
// W0T = W0 transposed, (2k + Q, N) fp32 (pack mode 2), so that thread = feature reads it coalesced.  Block = (16 objects,
// question); replaces the (B n^2, 192) pair matrix of rn_pair_build_fwd: 138 MB written and read back at the headline shape.
template <typename TX>   // bf16 (bf16 chain) or f16 (f16s chain) object rows
__global__ __launch_bounds__(256) void pair_tables_kernel(const float* __restrict__ x, long sxb, long sxn, long sxk,
                                                          const float* __restrict__ q, long ldq, const float* __restrict__ W0T,
                                                          const float* __restrict__ b0, TX* __restrict__ Xp, float* __restrict__ Vc,
                                                          int n, int k, int Q, int N, const float* __restrict__ coord, int kf) {
  __shared__ float xs[16][32];
  __shared__ float qs[1024];
  const int t = threadIdx.x, b = blockIdx.y, i0 = blockIdx.x * 16;
  // headline shape (one feature per thread, Q = 128): the thread's 26 + 128 weights are requested BEFORE the object rows and the
  // question are staged -- they depend on nothing staged, and behind the barrier they were a second round trip
  const bool pre = N <= 256 && Q == 128;
  float pwi[32], pwv[128], pb0 = 0.f;
  if (pre) {
    const int f = t < N ? t : 0;
#pragma unroll
    for (int c = 0; c < 32; ++c) pwi[c] = c < k ? W0T[(long)(k + c) * N + f] : 0.f;
#pragma unroll
    for (int u = 0; u < 128; ++u) pwv[u] = W0T[(long)(2 * k + u) * N + f];
    pb0 = b0[f];
  }
  for (int c = t; c < 16 * 32; c += 256) {
    const int r = c >> 5, cc = c & 31;
    float v = 0.f;
    if (i0 + r < n && cc < k) v = cc < kf ? x[b * sxb + (long)(i0 + r) * sxn + cc * sxk] : coord[(long)(cc - kf) * n + i0 + r];
    xs[r][cc] = v;                                                   // columns kf..k-1: the coordinate tags (model.py:195-201)
  }
  for (int c = t; c < Q; c += 256) qs[c] = q[b * ldq + c];
  __syncthreads();
  {                                                                  // packed object rows: 16 x 64 bf16, 4 per thread
    const int r = t >> 4, c4 = (t & 15) * 4;
    if (i0 + r < n) {
      TX* dst = Xp + ((long)b * n + i0 + r) * 64 + c4;
#pragma unroll
      for (int e = 0; e < 4; ++e) dst[e] = (TX)(c4 + e < 32 ? xs[r][c4 + e] : 0.f);
    }
  }
  if (pre) {
    if (t < N) {
      float cq = pb0;
#pragma unroll
      for (int u = 0; u < 128; ++u) cq = fmaf(pwv[u], qs[u], cq);
#pragma unroll 4
      for (int r = 0; r < 16; ++r) {
        if (i0 + r >= n) break;
        float v = cq;
#pragma unroll
        for (int c = 0; c < 32; ++c) v = fmaf(pwi[c], xs[r][c], v);
        Vc[((long)b * n + i0 + r) * N + t] = v;
      }
    }
    return;
  }
  for (int f = t; f < N; f += 256) {
    // This kernel is a chain of L2 round trips on the critical path in front of the forward chain: the x_i weights and the first
    // 128 question weights of the thread's column are requested together (160 independent loads, one wave per SIMD: the
    // registers are there) -- one round trip instead of five for the headline shape (Q = 128).
    float wi[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) wi[c] = c < k ? W0T[(long)(k + c) * N + f] : 0.f;
    float cq = b0[f];
    int qq = 0;
    for (; qq + 128 <= Q; qq += 128) {
      float wv[128];
#pragma unroll
      for (int u = 0; u < 128; ++u) wv[u] = W0T[(long)(2 * k + qq + u) * N + f];
#pragma unroll
      for (int u = 0; u < 128; ++u) cq = fmaf(wv[u], qs[qq + u], cq);
    }
    for (; qq + 32 <= Q; qq += 32) {
      float wv[32];
#pragma unroll
      for (int u = 0; u < 32; ++u) wv[u] = W0T[(long)(2 * k + qq + u) * N + f];
#pragma unroll
      for (int u = 0; u < 32; ++u) cq = fmaf(wv[u], qs[qq + u], cq);
    }
    for (; qq < Q; ++qq) cq = fmaf(W0T[(long)(2 * k + qq) * N + f], qs[qq], cq);
#pragma unroll 4
    for (int r = 0; r < 16; ++r) {
      if (i0 + r >= n) break;
      float v = cq;
#pragma unroll
      for (int c = 0; c < 32; ++c) v = fmaf(wi[c], xs[r][c], v);
      Vc[((long)b * n + i0 + r) * N + f] = v;
    }
  }
}

extern "C" int rn_pair_tables(const float* x, long sxb, long sxn, long sxk, const float* coord, int kf, const float* q, long ldq,
                              const float* W0T, const float* b0, void* Xp, int xp_dtype, float* Vc, int B, int n, int k, int Q, int N,
                              void* stream) {
  if (!coord) kf = k;
  RN_CHECK_ARG(kf > 0 && kf <= k, "rn_pair_tables: kf=%d must be in (0, k=%d]", kf, k);
  RN_CHECK_ARG(x && W0T && b0 && Xp && Vc && B > 0 && n > 0, "rn_pair_tables: bad pointer/size");
  RN_CHECK_ARG(k > 0 && k <= 32 && Q >= 0 && Q <= 1024 && (Q == 0 || q) && N > 0,
               "rn_pair_tables: k=%d (<= 32), Q=%d (0 .. 1024; q required when Q > 0), N=%d unsupported", k, Q, N);
  RN_CHECK_ARG(xp_dtype == RN_BF16 || xp_dtype == RN_F16, "rn_pair_tables: object rows are bf16 or fp16 (dtype %d)", xp_dtype);
  if (xp_dtype == RN_F16) pair_tables_kernel<f16><<<dim3(cdiv(n, 16), B), 256, 0, (hipStream_t)stream>>>(x, sxb, sxn, sxk, q, ldq, W0T, b0, (f16*)Xp, Vc, n, k, Q, N, coord, kf);
  else pair_tables_kernel<bf16><<<dim3(cdiv(n, 16), B), 256, 0, (hipStream_t)stream>>>(x, sxb, sxn, sxk, q, ldq, W0T, b0, (bf16*)Xp, Vc, n, k, Q, N, coord, kf);
  RN_LAUNCH_CHECK("rn_pair_tables");
  return 0;
}
