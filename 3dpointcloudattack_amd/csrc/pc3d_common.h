// Shared host/device helpers for libpc3d_hip.so (gfx950 only — no other target is supported).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/pc3d.h"

namespace pc3d {

void set_error(const char* fmt, ...);

// Element strides of a [B, n_points, 3] fp32 point set in caller memory.
struct PtsView {
  const float* p;
  int64_t bs, ps, cs;
};
struct PtsViewMut {
  float* p;
  int64_t bs, ps, cs;
};

constexpr int kWave = 64;  // CDNA wavefront

#define PC3D_REQUIRE(cond, ...)            \
  do {                                     \
    if (!(cond)) {                         \
      pc3d::set_error(__VA_ARGS__);        \
      return PC3D_EINVAL;                  \
    }                                      \
  } while (0)

#define PC3D_LAUNCH_CHECK(name)                                              \
  do {                                                                       \
    hipError_t e_ = hipGetLastError();                                       \
    if (e_ != hipSuccess) {                                                  \
      pc3d::set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
      return (int)e_;                                                        \
    }                                                                        \
  } while (0)

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// Wave-level reductions through DPP/ds_swizzle-backed shuffles (64 lanes).
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

}  // namespace pc3d
