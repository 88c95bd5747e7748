// Shared device/host helpers for the gfx950 Relation-Network kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/rn_hip.h"

typedef __bf16 bf16;
typedef _Float16 f16;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

// ---- error plumbing (host) -------------------------------------------------
void rn_set_error(const char* fmt, ...);
#define RN_CHECK_ARG(cond, ...)                 \
  do {                                          \
    if (!(cond)) {                              \
      rn_set_error(__VA_ARGS__);                \
      return -1;                                \
    }                                           \
  } while (0)
#define RN_LAUNCH_CHECK(name)                                                    \
  do {                                                                           \
    hipError_t e_ = hipGetLastError();                                           \
    if (e_ != hipSuccess) {                                                      \
      rn_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));        \
      return (int)e_;                                                            \
    }                                                                            \
  } while (0)

// gfx950: 160 KiB of LDS per workgroup; dynamic allocations above 64 KiB must be opted into per kernel
#define RN_LDS_MAX (160 * 1024)
#define RN_LDS_OPT_IN(kernel, name)                                                                              \
  do {                                                                                                           \
    hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(&kernel),                                  \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, RN_LDS_MAX);                 \
    if (e_ != hipSuccess) {                                                                                      \
      rn_set_error("%s: cannot raise the dynamic LDS limit: %s", name, hipGetErrorString(e_));                   \
      return (int)e_;                                                                                            \
    }                                                                                                            \
  } while (0)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Diagnostic knobs (kernel-variant selection for A/B timing, timing ablations) exist in RN_DIAG builds only
// (`RN_DIAG=1 python relationnetworks-clevr_amd/_build.py`): the product library never reads the process environment.
#ifdef RN_DIAG
#include <stdlib.h>
static inline const char* rn_diag_env(const char* name) { return getenv(name); }
#else
static inline const char* rn_diag_env(const char*) { return nullptr; }
#endif

// ---- work items of the reducing backward chain (rn_chain_rr.hip) and the records its partial-sum kernel reads (rn_pair.hip): ONE
// definition for the kernel that writes and the kernel that reads (and rn_probe_red_schedule, which the CPU tests walk).
// Unit ((b * jgs + jg) * nu + v) -- question b, block jg of 32 objects j, v-th group of tiles_per_unit tiles -- sits at WALK position
// p = (v * jgs + jg) * nb + b (question fastest).  Work items: the nwhole units at walk positions [0, nwhole) as a whole (one Rj
// record: the unit's own), then the tiles of the units behind them one by one (tile 0 -> the unit's own record, tile t > 0 ->
// record nunits + (p - nwhole) (tpu - 1) + t - 1).
struct RnRedItem { int tile0, tcount; long rec; };
__host__ __device__ inline int rn_red_walk_pos(int b, int jg, int v, int nb, int jgs) { return (v * jgs + jg) * nb + b; }
__host__ __device__ inline int rn_red_unit_at(int p, int nb, int jgs, int nu) {
  const int r = p / nb, b = p - r * nb, v = r / jgs, jg = r - v * jgs;
  return (b * jgs + jg) * nu + v;
}
__host__ __device__ inline long rn_red_extra_rec(int p, int t, int nunits, int tpu, int nwhole) {   // record of tile t > 0 of the tail unit at walk position p
  return (long)nunits + (long)(p - nwhole) * (tpu - 1) + t - 1;
}
__host__ __device__ inline RnRedItem rn_red_item(int item, int nunits, int tpu, int nwhole, int nb, int jgs, int nu) {
  RnRedItem it;
  if (item < nwhole) {
    it.rec = rn_red_unit_at(item, nb, jgs, nu);
    it.tile0 = (int)it.rec * tpu;
    it.tcount = tpu;
  } else {
    const int f = item - nwhole, uo = f / tpu, t = f - uo * tpu, unit = rn_red_unit_at(nwhole + uo, nb, jgs, nu);
    it.tile0 = unit * tpu + t;
    it.tcount = 1;
    it.rec = t == 0 ? unit : rn_red_extra_rec(nwhole + uo, t, nunits, tpu, nwhole);
  }
  return it;
}

// workspace sizes of the entry points, one function per file that owns the layout; exported through rn_workspace_bytes (rn_pair.hip)
size_t rnws_rr_mask(int M);
size_t rnws_conv_bwd_weight(int N, int Cin, int H, int W);
size_t rnws_bn_relu(int N, int C, int HW);
size_t rnws_extract(int B, int n, int F);
size_t rnws_pair_sum(int B, int npairs, int G);
size_t rnws_pair_reduce(int B, int n, int G);
size_t rnws_wgrad0(int B, int n, int N);
size_t rnws_pair_features(int B, int npairs, int F);
size_t rnws_f_phi_nll(int B);
size_t rnws_f_phi_bwd(int B, int F1, int F2, int A);
size_t rnws_clip_adam(void);
size_t rnws_f_phi_split(void);
size_t rnws_wgrad(int M, int N, int K);
size_t rnws_wgrad_blocked(int M, int rows_per_question, int njobs, int aligned);

// rn_convnorm.hip: pass 1 of the batch-norm backward (slice sums into ws), launched by rn_conv.hip's fused entry
int rn_cn_launch_bwd_sums(const float* dy, const float* x, const float* mean, const float* invstd, const float* gamma, const float* beta,
                          void* ws, int N, int C, int HW, void* stream, int* S);

// ---- element traits ----------------------------------------------------------
template <typename T> struct Elem;
template <> struct Elem<bf16> {
  static constexpr int kPer16B = 8;
  static __device__ __forceinline__ float to_f32(bf16 v) { return (float)v; }
  static __device__ __forceinline__ bf16 from_f32(float v) { return (bf16)v; }   // RNE, v_cvt_pk_bf16_f32
};
template <> struct Elem<f16> {
  static constexpr int kPer16B = 8;
  static __device__ __forceinline__ float to_f32(f16 v) { return (float)v; }
  static __device__ __forceinline__ f16 from_f32(float v) { return (f16)v; }       // RNE, v_cvt_f16_f32
};
template <> struct Elem<float> {
  static constexpr int kPer16B = 4;
  static __device__ __forceinline__ float to_f32(float v) { return v; }
  static __device__ __forceinline__ float from_f32(float v) { return v; }
};

// 16-byte chunk of storage elements
template <typename T> struct Chunk16 { T v[Elem<T>::kPer16B]; } __attribute__((aligned(16)));

template <typename T> __device__ __forceinline__ bool is_pos(T v);
template <> __device__ __forceinline__ bool is_pos<bf16>(bf16 v) { return (float)v > 0.f; }
template <> __device__ __forceinline__ bool is_pos<float>(float v) { return v > 0.f; }
template <> __device__ __forceinline__ bool is_pos<f16>(f16 v) { return (float)v > 0.f; }

// ---- fp8 (OCP e4m3) copies of the stored activations ------------------------------------------------------------
// The H_l rows the forward chains keep for the weight gradient are read once, by an HBM-bound kernel: one byte per element
// instead of two.  gfx950's scaled conversions (v_cvt_scalef32_pk_*) take the scale as a power-of-two float: down-conversions
// divide by it (round to nearest even; beyond 448 the result is the NaN byte 0x7f, not a saturated one), up-conversions
// multiply (tests/test_gpu_kernels.py pins both).  The callers' values are post-ReLU, i.e. non-negative, so the clamp to
// 448 * RN_H8_SCALE -- far above what the g layers produce: 11 on the released checkpoints -- is an unsigned 16-bit minimum
// on the packed pair.
constexpr float RN_H8_SCALE = 1.0f;
typedef __attribute__((ext_vector_type(2))) short rn_s16x2;
typedef __attribute__((ext_vector_type(2))) unsigned short rn_u16x2;
__device__ __forceinline__ unsigned rn_pk_min_u16(unsigned v, unsigned short cap) {
  const rn_u16x2 c = {cap, cap};
  return __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_bit_cast(rn_u16x2, v), c));   // v_pk_min_u16
}
// two packed NON-NEGATIVE bf16 pairs (features f..f+3) -> one dword of four e4m3 bytes
__device__ __forceinline__ unsigned rn_fp8x4_from_bf16(unsigned lo, unsigned hi) {
  static_assert(RN_H8_SCALE == 1.0f, "the clamp constants are 448 in bf16 / fp16");
  lo = rn_pk_min_u16(lo, 0x43E0);
  hi = rn_pk_min_u16(hi, 0x43E0);
  rn_s16x2 r = {0, 0};
  r = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(r, __builtin_bit_cast(bf16x2, lo), RN_H8_SCALE, false);
  r = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(r, __builtin_bit_cast(bf16x2, hi), RN_H8_SCALE, true);
  return __builtin_bit_cast(unsigned, r);
}
typedef __attribute__((ext_vector_type(2))) _Float16 rn_f16x2;
__device__ __forceinline__ unsigned rn_fp8x4_from_f16(unsigned lo, unsigned hi) {
  lo = rn_pk_min_u16(lo, 0x5F00);
  hi = rn_pk_min_u16(hi, 0x5F00);
  rn_s16x2 r = {0, 0};
  r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(r, __builtin_bit_cast(rn_f16x2, lo), RN_H8_SCALE, false);
  r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(r, __builtin_bit_cast(rn_f16x2, hi), RN_H8_SCALE, true);
  return __builtin_bit_cast(unsigned, r);
}
// eight e4m3 bytes (two dwords, k ascending) -> the bf16x8 MFMA operand
__device__ __forceinline__ bf16x8 rn_bf16x8_from_fp8(unsigned d0, unsigned d1) {
  union { bf16x2 p[4]; bf16x8 v; } u;
  u.p[0] = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(d0, RN_H8_SCALE, false);
  u.p[1] = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(d0, RN_H8_SCALE, true);
  u.p[2] = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(d1, RN_H8_SCALE, false);
  u.p[3] = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(d1, RN_H8_SCALE, true);
  return u.v;
}
