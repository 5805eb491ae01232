// Shared helpers for the ococc HIP library (gfx950 only).
// Error convention of the C ABI (include/ococc_hip.h): every entry point
// returns 0 on success or a negative OCOCC_E* code and records a message that
// ococc_last_error() hands back (thread local).  No entry point allocates,
// synchronises the device or throws; all work is queued on the caller's stream.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/ococc_hip.h"

extern thread_local char ococc_err_buf[512];

static inline int ococc_fail(int code, const char* fn, const char* msg) {
  snprintf(ococc_err_buf, sizeof(ococc_err_buf), "%s: %s", fn, msg);
  return code;
}

#define OCOCC_REQUIRE(cond, msg)                                   \
  do {                                                             \
    if (!(cond)) return ococc_fail(OCOCC_EINVAL, __func__, msg);   \
  } while (0)

#define OCOCC_HIP(call)                                                          \
  do {                                                                           \
    hipError_t e__ = (call);                                                     \
    if (e__ != hipSuccess) return ococc_fail(OCOCC_EHIP, __func__, hipGetErrorString(e__)); \
  } while (0)

#define OCOCC_CHECK_LAUNCH() OCOCC_HIP(hipGetLastError())

static inline int64_t ococc_align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }
static inline int64_t ococc_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Memory-bound grid sizing: cap the grid and grid-stride the rest.
static inline int ococc_grid_1d(int64_t work_items, int block, int cap = 4096) {
  int64_t g = (work_items + block - 1) / block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (int)g;
}

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

__device__ __forceinline__ float ococc_bf16_to_f32(unsigned short h) {
  return __uint_as_float(((unsigned int)h) << 16);
}
// two floats -> one dword of two bf16 (lo in bits 0-15): ONE v_cvt_pk_bf16_f32.  (Written as two scalar casts joined with
// shift / or, hipcc emits two conversions and a v_or_b32_sdwa.)
typedef __attribute__((ext_vector_type(2))) __bf16 ococc_bf16x2;
__device__ __forceinline__ unsigned int ococc_pack_bf16x2(float lo, float hi) {
  const ococc_bf16x2 p = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(unsigned int, p);
}
// round-to-nearest-even through the compiler's cast (v_cvt_pk_bf16_f32 keeps NaNs)
__device__ __forceinline__ unsigned short ococc_f32_to_bf16(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(unsigned short, b);
}
