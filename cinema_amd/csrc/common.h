// Shared device/host helpers for libcinema_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/cinema_hip.h"

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

void cinema_set_error(const char *fmt, ...);

// Launch state that belongs to a DEVICE (the dynamic-LDS limit hipFuncSetAttribute raises applies to the current device
// only; CU counts differ between devices) is cached per device index, in relaxed atomics: any host thread may call.
#include <atomic>
#define CINEMA_MAX_DEVICES 64
int cinema_dev();        // hipGetDevice(), clamped to [0, CINEMA_MAX_DEVICES)
int cinema_cu_count();   // compute units of the current device (256 if the query fails)

#define CINEMA_CHECK_HIP(expr)                                                              \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if (e_ != hipSuccess) {                                                                 \
      cinema_set_error("%s failed at %s:%d: %s", #expr, __FILE__, __LINE__, hipGetErrorString(e_)); \
      return (int)e_;                                                                       \
    }                                                                                       \
  } while (0)

// raise the dynamic-LDS limit of kernel `fn` to `bytes` on the current device (once per device and size)
#define CINEMA_SMEM_ATTR(fn, bytes)                                                                        \
  do {                                                                                                     \
    static std::atomic<int> done_[CINEMA_MAX_DEVICES];                                                     \
    const int dv_ = cinema_dev();                                                                          \
    if (done_[dv_].load(std::memory_order_relaxed) < (int)(bytes)) {                                       \
      CINEMA_CHECK_HIP(hipFuncSetAttribute((const void *)(fn), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                           (int)(bytes)));                                                 \
      done_[dv_].store((int)(bytes), std::memory_order_relaxed);                                           \
    }                                                                                                      \
  } while (0)

#define CINEMA_REQUIRE(cond, ...)        \
  do {                                   \
    if (!(cond)) {                       \
      cinema_set_error(__VA_ARGS__);     \
      return -1;                         \
    }                                    \
  } while (0)

__device__ __forceinline__ float to_f32(float x) { return x; }
__device__ __forceinline__ float to_f32(bf16_t x) { return (float)x; }

// 16 consecutive elements <-> float[16]; p must be 32-B (bf16) / 64-B (f32) aligned.
__device__ __forceinline__ void load16(const bf16_t *p, float (&v)[16]) {
  const u32x4 a = *reinterpret_cast<const u32x4 *>(p);
  const u32x4 b = *reinterpret_cast<const u32x4 *>(p + 8);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    v[2 * j] = __uint_as_float(a[j] << 16);
    v[2 * j + 1] = __uint_as_float(a[j] & 0xffff0000u);
    v[8 + 2 * j] = __uint_as_float(b[j] << 16);
    v[8 + 2 * j + 1] = __uint_as_float(b[j] & 0xffff0000u);
  }
}
__device__ __forceinline__ void load16(const float *p, float (&v)[16]) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x4 a = *reinterpret_cast<const f32x4 *>(p + 4 * q);
    v[4 * q] = a[0]; v[4 * q + 1] = a[1]; v[4 * q + 2] = a[2]; v[4 * q + 3] = a[3];
  }
}
__device__ __forceinline__ void store16(bf16_t *p, const float (&v)[16]) {
  bf16x8 lo, hi;
#pragma unroll
  for (int j = 0; j < 8; ++j) { lo[j] = (bf16_t)v[j]; hi[j] = (bf16_t)v[8 + j]; }
  *reinterpret_cast<bf16x8 *>(p) = lo;
  *reinterpret_cast<bf16x8 *>(p + 8) = hi;
}
__device__ __forceinline__ void store16(float *p, const float (&v)[16]) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    f32x4 a = {v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
    *reinterpret_cast<f32x4 *>(p + 4 * q) = a;
  }
}

// Reduce 16 per-lane values across the 16 lanes of a lane group (lanes sharing lane>>4).
// On return every lane holds the group total of value index reduce16_index(lane & 15).
__device__ __forceinline__ float reduce16_scatter(float (&v)[16], int lane) {
  float w8[8], w4[4], w2[2];
  const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4, b3 = lane & 8;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float send = b0 ? v[j] : v[j + 8];
    const float keep = b0 ? v[j + 8] : v[j];
    w8[j] = keep + __shfl_xor(send, 1, 64);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float send = b1 ? w8[j] : w8[j + 4];
    const float keep = b1 ? w8[j + 4] : w8[j];
    w4[j] = keep + __shfl_xor(send, 2, 64);
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const float send = b2 ? w4[j] : w4[j + 2];
    const float keep = b2 ? w4[j + 2] : w4[j];
    w2[j] = keep + __shfl_xor(send, 4, 64);
  }
  const float send = b3 ? w2[0] : w2[1];
  const float keep = b3 ? w2[1] : w2[0];
  return keep + __shfl_xor(send, 8, 64);
}
__device__ __forceinline__ int reduce16_index(int i) {
  return ((i & 1) << 3) | ((i & 2) << 1) | ((i & 4) >> 1) | ((i & 8) >> 3);
}

// The same for 8 per-lane values: every lane of a 16-lane group ends with the group total of value index
// reduce8_index(lane & 15) (lanes i and i ^ 8 hold the same one).
__device__ __forceinline__ float reduce8_scatter(float (&v)[8], int lane) {
  float w4[4], w2[2];
  const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float send = b0 ? v[j] : v[j + 4];
    const float keep = b0 ? v[j + 4] : v[j];
    w4[j] = keep + __shfl_xor(send, 1, 64);
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const float send = b1 ? w4[j] : w4[j + 2];
    const float keep = b1 ? w4[j + 2] : w4[j];
    w2[j] = keep + __shfl_xor(send, 2, 64);
  }
  const float send = b2 ? w2[0] : w2[1];
  const float keep = b2 ? w2[1] : w2[0];
  const float x = keep + __shfl_xor(send, 4, 64);
  return x + __shfl_xor(x, 8, 64);
}
__device__ __forceinline__ int reduce8_index(int i) { return ((i & 1) << 2) | (i & 2) | ((i & 4) >> 2); }

// XCD-aware, bijective remap of a 1-D block id: blocks that share an XCD (bid % 8 equal)
// get a contiguous range of logical ids, so neighbouring tiles share that XCD's L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
  const int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
  return base + (bid >> 3);
}
