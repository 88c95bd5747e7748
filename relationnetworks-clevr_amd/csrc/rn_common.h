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
