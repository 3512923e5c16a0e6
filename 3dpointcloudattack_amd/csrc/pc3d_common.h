// Shared host/device helpers for libpc3d_hip.so (gfx950 only — no other target is supported).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/pc3d.h"

namespace pc3d {

void set_error(const char* fmt, ...);

// gemm.hip: the group-max form of the point-wise GEMM (rows in groups of 32 / 64 / 128; see GemmArgs::gm_ns)
int gemm_nt_groupmax(const float* X, const float* W, const float* bias, int G, int ns, int K, int N, float* out, int64_t* arg,
                     void* stream);

// det.hip: deterministic scatter-adds (one wavefront owns an LDS accumulator tile and walks the records in order)
//   out[b, tgt[b,r], c] (+)= sum_r mask(val[b,r,c])      — tgt [B,R], val [B,R,ldv], act [B,R,lda] or null
int scatter_rows_det(const char* nm, const int32_t* tgt, const float* val, int64_t ldv, const float* act, int64_t lda, float slope,
                     int B, int R, int N, int C, float* out, int64_t ldo, int accumulate, int clamp, void* stream,
                     const uint8_t* mbits = nullptr,   // mbits [B,R,C/4]: the activation's sign as bits instead of `act`
                     int64_t out_bs = 0, int64_t out_cs = 1,    // batch / channel strides of out (0: N * ldo)
                     const float* row_bias = nullptr, const float* col_w = nullptr);   // + row_bias[b,n] * col_w[c] on the way out
//   dst[b, arg[b,i,c], c] += w[b,i,c]  (mode 0: w = g; mode 1: w = g * leaky'(outv), dst = [dP | dQ] with dQ = w)
int arg_scatter_det(const char* nm, const float* g, int64_t ldg, const float* outv, const int32_t* arg, int B, int S, int N, int C,
                    float slope, float* dst, int mode, void* stream, int slice = 0, const float* g2 = nullptr, int64_t ldg2 = 0);
// ascending in-place sort of every segment of a CSR list (off [B,NA+1], lst [B,L]): arrival order -> a fixed order
int sort_segments(const char* nm, const int32_t* off, int32_t* lst, int B, int NA, int64_t L, void* stream);
// out[b,t,:] = sum over the sorted reverse-index segment of t (off [B,NA+1], lst [B,E]) of mask(val[b,e,:])
int rev_gather_sum(const char* nm, const float* val, int64_t ldv, const float* act, int64_t lda, float slope, const int32_t* off,
                   const int32_t* lst, int B, int E, int NA, int C, float* out, int64_t ldo, void* stream);
bool own_fits(int N);   // N destination rows fit the LDS tile of the owner-wave kernels

// fps_pruned.hip: farthest-point sampling on one wavefront per cloud with exact pruning (N <= 4096)
int fps_pruned_launch(const char* nm, const float* xyz, int64_t x_bs, int64_t x_ps, int64_t x_cs, int B, int N, int S,
                      const int32_t* start, int32_t* out, void* stream);

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

// Zero-fill as a KERNEL. hipMemsetAsync must not be used by this library: captured into a hipGraph (the attack loops
// replay graphs) its memset node wrote garbage from the second replay on (ROCm 7.0, tools/exp_graph_memset.py).
__global__ __launch_bounds__(256) static void zero_fill_kernel(float4* __restrict__ p4, size_t n4, float* __restrict__ tail,
                                                               int ntail) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n4) p4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < (size_t)ntail) tail[i] = 0.f;
}
__global__ __launch_bounds__(256) static void zero_fill_scalar_kernel(float* __restrict__ p, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = 0.f;
}
// n floats at p (float4 stores when p is 16-byte aligned, as any torch allocation is).
static inline hipError_t zero_async(float* p, size_t n, hipStream_t st) {
  if (n == 0) return hipSuccess;
  if (reinterpret_cast<uintptr_t>(p) & 15) {
    hipLaunchKernelGGL(zero_fill_scalar_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p, n);
    return hipGetLastError();
  }
  const size_t n4 = n / 4;
  const int ntail = (int)(n % 4);
  const size_t blocks = (n4 + 255) / 256 + (n4 == 0 ? 1 : 0);
  hipLaunchKernelGGL(zero_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, st, reinterpret_cast<float4*>(p), n4,
                     p + n4 * 4, ntail);
  return hipGetLastError();
}

// Orders LDS traffic between the lanes of ONE wavefront (LDS operations of a wave complete in issue order; this only
// stops the compiler from moving them across).
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// XCD-aware order of a one-dimensional grid. Workgroup ids are dealt round-robin to the 8 XCDs, each with its own L2:
// workgroups that read the same rows (the channel slices of one cloud) should sit on ONE of them, or every L2 fetches
// every line. Launch xcd_grid(total) workgroups; xcd_band_id gives workgroup `wg` its place in an order in which each XCD
// owns a contiguous band of ceil(total / 8) places, or -1 for the ids that pad the grid.
__device__ __forceinline__ int xcd_band_id(int wg, int total) {
  const int per = (total + 7) >> 3, t = (wg & 7) * per + (wg >> 3);
  return t < total ? t : -1;
}
static inline int xcd_grid(int total) { return ((total + 7) >> 3) << 3; }
// ... and without touching the launch: for a kernel on a dim3(tiles, clouds) grid whose size is a multiple of 8 (every
// benchmark batch), the same re-deal computed from blockIdx (identity otherwise: correct, just not banded).
__device__ __forceinline__ void xcd_swizzle(int& bx, int& by) {
  const int gx = gridDim.x, total = gx * gridDim.y;
  bx = blockIdx.x, by = blockIdx.y;
  if (total & 7) return;
  const int l = bx + gx * by, t = (l & 7) * (total >> 3) + (l >> 3);
  by = t / gx;
  bx = t - by * gx;
}
// The same for kernels written for a (tiles, clouds) grid: launched as dim3(xcd_grid(gx * gy)), the (x, y) this workgroup
// takes — every cloud's tiles on one XCD, whose L2 then holds that cloud's rows instead of a share of every cloud's.
__device__ __forceinline__ bool xcd_block(int gx, int gy, int& bx, int& by) {
  const int t = xcd_band_id(blockIdx.x, gx * gy);
  if (t < 0) return false;
  by = t / gx;
  bx = t - by * gx;
  return true;
}

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


// Low-latency wave reductions on DPP row shifts / row broadcasts (VALU only, no LDS crossbar round trips): for
// idempotent operators (max, min) overlapping contributions are harmless. The result is wave-uniform.
#define PC3D_DPP_STEP_F(op, v, ctrl)                                                                     \
  v = op(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v),            \
                                                                  __builtin_bit_cast(int, v), ctrl, 0xf, 0xf, false)))
#define PC3D_DPP_STEP_I(op, v, ctrl) v = op(v, __builtin_amdgcn_update_dpp(v, v, ctrl, 0xf, 0xf, false))
__device__ __forceinline__ float wave_max_dpp(float v) {
  PC3D_DPP_STEP_F(fmaxf, v, 0x111);  // row_shr:1
  PC3D_DPP_STEP_F(fmaxf, v, 0x112);  // row_shr:2
  PC3D_DPP_STEP_F(fmaxf, v, 0x114);  // row_shr:4
  PC3D_DPP_STEP_F(fmaxf, v, 0x118);  // row_shr:8   -> lane 15 of every row holds the row maximum
  PC3D_DPP_STEP_F(fmaxf, v, 0x142);  // row_bcast:15
  PC3D_DPP_STEP_F(fmaxf, v, 0x143);  // row_bcast:31 -> lane 63 holds the wave maximum
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// Wave maximum of values that are never NaN, wave-uniform result: ONE v_max_f32 with a DPP operand per level (the C++ form
// compiles to move + DPP move + two canonicalising maxima per level). A DPP instruction needs two wait states after the
// VALU write of its source, v_readlane one after the last write; the trailing nops cover a VALU read of the SGPR result.
__device__ __forceinline__ float wave_max_chain(float v) {
  float r;
  asm volatile(
      "s_nop 1\n\t"
      "v_max_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_f32_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_f32_dpp %1, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_f32_dpp %1, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_f32_dpp %1, %1, %1 row_bcast:15 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_f32_dpp %1, %1, %1 row_bcast:31 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_readlane_b32 %0, %1, 63\n\t"
      "s_nop 1"
      : "=s"(r), "+v"(v));
  return r;
}

// (value, index) of lanes l and l ^ 32 combined IN BOTH lanes: the larger value, the lower index on a tie. One
// v_permlane32_swap_b32 (gfx950) per operand — a VALU instruction — instead of a ds_bpermute round trip through LDS each.
// `best` must not be NaN (the arg-max loops that call this start from -inf and update on a strict >).
__device__ __forceinline__ void argmax_xor32(float& best, int& bi) {
  const unsigned bv = __builtin_bit_cast(unsigned, best), bu = (unsigned)bi;
  const auto rv = __builtin_amdgcn_permlane32_swap(bv, bv, false, false);     // [0]: the low half's value, [1]: the high half's
  const auto ri = __builtin_amdgcn_permlane32_swap(bu, bu, false, false);
  const float lv = __builtin_bit_cast(float, (unsigned)rv[0]), hv = __builtin_bit_cast(float, (unsigned)rv[1]);
  const int li = (int)ri[0], hi = (int)ri[1];
  const bool high = hv > lv || (hv == lv && hi < li);
  best = high ? hv : lv;
  bi = high ? hi : li;
}
// v(lane l) + v(lane l ^ 32) in both lanes, on one v_permlane32_swap_b32 (low half + high half: the same sum in either lane)
__device__ __forceinline__ float sum_xor32(float v) {
  const unsigned u = __builtin_bit_cast(unsigned, v);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
}
__device__ __forceinline__ int imin_(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ int wave_min_dpp_i32(int v) {
  PC3D_DPP_STEP_I(imin_, v, 0x111);
  PC3D_DPP_STEP_I(imin_, v, 0x112);
  PC3D_DPP_STEP_I(imin_, v, 0x114);
  PC3D_DPP_STEP_I(imin_, v, 0x118);
  PC3D_DPP_STEP_I(imin_, v, 0x142);
  PC3D_DPP_STEP_I(imin_, v, 0x143);
  return __builtin_amdgcn_readlane(v, 63);
}

}  // namespace pc3d
