// K17: the two bandwidth-bound halves of CurveNet's local point-feature aggregation (model/curvenet_util.py:175-236)
// around its 1x1-conv GEMM, each as one pass over the [B,N,k,C] edge tensor:
//   edge_act : E[b,i,j,:] = leaky(A[b,idx[b,i,j],:] + Bc[b,i,:])      (gather + centre term + activation)
//   act_mean : out[b,i,:] = mean_j leaky(Z[b,i,j,:])                   (activation + neighbour mean)
// The reference builds the same edge tensor with gather / subtract / concat / permute / conv / add / LeakyReLU
// launches (eight passes) and averages after a separate LeakyReLU (two more).
// All tensors are channels-last, C % 4 == 0, accessed as float4.
#include "pc3d_common.h"

namespace pc3d {

__device__ __forceinline__ float4 leaky4(float4 v, float s) {
  return make_float4(v.x > 0.f ? v.x : v.x * s, v.y > 0.f ? v.y : v.y * s, v.z > 0.f ? v.z : v.z * s,
                     v.w > 0.f ? v.w : v.w * s);
}
// derivative of LeakyReLU taken from the OUTPUT's sign (leaky preserves sign; torch uses x > 0 ? 1 : slope)
__device__ __forceinline__ float4 leaky_grad4(float4 g, float4 out, float s) {
  return make_float4(out.x > 0.f ? g.x : g.x * s, out.y > 0.f ? g.y : g.y * s, out.z > 0.f ? g.z : g.z * s,
                     out.w > 0.f ? g.w : g.w * s);
}

// one thread per float4 of E; consecutive threads walk the channels of one edge, then the next edge
__global__ __launch_bounds__(256) void edge_act_fwd_kernel(const float4* __restrict__ A, const float4* __restrict__ Bc,
                                                           const int* __restrict__ idx, int N, int K, int C4,
                                                           float slope, float4* __restrict__ E, int64_t total) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const int c4 = (int)(t % C4);
  const int64_t e = t / C4;            // edge (b, i, j)
  const int64_t bi = e / K;            // point (b, i)
  const int64_t b = bi / N;
  const int j = min(max(idx[e], 0), N - 1);
  const float4 a = A[(b * N + j) * C4 + c4], c = Bc[bi * C4 + c4];
  E[t] = leaky4(make_float4(a.x + c.x, a.y + c.y, a.z + c.z, a.w + c.w), slope);
}

// one thread per (point, channel): gBc = sum_j g_pre, gA[idx] += g_pre. Consecutive lanes own consecutive channels, so
// every atomic instruction of a wave lands on whole contiguous rows (64 / C of them) instead of strided quarters.
__global__ __launch_bounds__(256) void edge_act_bwd_kernel(const float* __restrict__ gE, const float* __restrict__ E,
                                                           const int* __restrict__ idx, int N, int K, int C,
                                                           float slope, float* __restrict__ gA,
                                                           float* __restrict__ gBc, int64_t total) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const int c = (int)(t % C);
  const int64_t bi = t / C, b = bi / N;
  const float* ge = gE + bi * K * C + c;
  const float* ev = E + bi * K * C + c;
  const int* id = idx + bi * K;
  float* ga = gA + b * N * C + c;
  float acc = 0.f;
  int j = 0;
  for (; j + 4 <= K; j += 4) {   // four edges in flight
    float g[4], o[4];
    int r[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) g[u] = ge[(int64_t)(j + u) * C], o[u] = ev[(int64_t)(j + u) * C], r[u] = id[j + u];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float v = o[u] > 0.f ? g[u] : g[u] * slope;
      acc += v;
      atomicAdd(ga + (int64_t)min(max(r[u], 0), N - 1) * C, v);
    }
  }
  for (; j < K; ++j) {
    const float g = ge[(int64_t)j * C];
    const float v = ev[(int64_t)j * C] > 0.f ? g : g * slope;
    acc += v;
    atomicAdd(ga + (int64_t)min(max(id[j], 0), N - 1) * C, v);
  }
  gBc[t] = acc;
}

__global__ __launch_bounds__(256) void act_mean_fwd_kernel(const float4* __restrict__ Z, int K, int C4, float slope,
                                                           float4* __restrict__ out, int64_t total) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const int c4 = (int)(t % C4);
  const int64_t bi = t / C4;
  const float4* z = Z + bi * K * C4 + c4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int j = 0; j < K; ++j) {
    const float4 v = leaky4(z[(int64_t)j * C4], slope);
    acc.x += v.x, acc.y += v.y, acc.z += v.z, acc.w += v.w;
  }
  const float inv = 1.f / (float)K;
  out[t] = make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
}

// one thread per float4 of gZ
__global__ __launch_bounds__(256) void act_mean_bwd_kernel(const float4* __restrict__ Z, const float4* __restrict__ gout,
                                                           int K, int C4, float slope, float4* __restrict__ gZ,
                                                           int64_t total) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const int c4 = (int)(t % C4);
  const int64_t bi = t / C4 / K;
  const float inv = 1.f / (float)K;
  float4 g = gout[bi * C4 + c4];
  g = make_float4(g.x * inv, g.y * inv, g.z * inv, g.w * inv);
  gZ[t] = leaky_grad4(g, Z[t], slope);
}

static int lpfa_grid(int64_t total, unsigned* blocks, const char* nm) {
  const int64_t nb = (total + 255) / 256;
  PC3D_REQUIRE(nb <= 0x7fffffffLL, "%s: problem too large for one launch", nm);
  *blocks = (unsigned)nb;
  return PC3D_OK;
}

}  // namespace pc3d

using namespace pc3d;

#define LPFA_SIZES(nm)                                                                                               \
  PC3D_REQUIRE(B >= 0 && N >= 1 && K >= 1 && C >= 4 && C % 4 == 0, nm ": bad sizes B=%d N=%d K=%d C=%d (C %% 4 == 0)", \
               B, N, K, C);                                                                                          \
  if (B == 0) return PC3D_OK

extern "C" int pc3d_edge_act_f32(const float* A, const float* Bc, const int32_t* idx, int B, int N, int K, int C,
                                 float slope, float* E, void* stream) {
  LPFA_SIZES("pc3d_edge_act_f32");
  PC3D_REQUIRE(A && Bc && idx && E, "pc3d_edge_act_f32: null pointer");
  const int64_t total = (int64_t)B * N * K * (C / 4);
  unsigned blocks;
  if (int rc = lpfa_grid(total, &blocks, "pc3d_edge_act_f32")) return rc;
  hipLaunchKernelGGL(edge_act_fwd_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), (const float4*)A,
                     (const float4*)Bc, idx, N, K, C / 4, slope, (float4*)E, total);
  PC3D_LAUNCH_CHECK("pc3d_edge_act_f32");
  return PC3D_OK;
}

extern "C" int pc3d_edge_act_bwd_f32(const float* gE, const float* E, const int32_t* idx, int B, int N, int K, int C,
                                     float slope, float* gA, float* gBc, void* stream) {
  LPFA_SIZES("pc3d_edge_act_bwd_f32");
  PC3D_REQUIRE(gE && E && idx && gA && gBc, "pc3d_edge_act_bwd_f32: null pointer");
  const int64_t total = (int64_t)B * N * C;
  unsigned blocks;
  if (int rc = lpfa_grid(total, &blocks, "pc3d_edge_act_bwd_f32")) return rc;
  hipLaunchKernelGGL(edge_act_bwd_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), gE, E, idx, N, K, C, slope,
                     gA, gBc, total);
  PC3D_LAUNCH_CHECK("pc3d_edge_act_bwd_f32");
  return PC3D_OK;
}

extern "C" int pc3d_act_mean_f32(const float* Z, int B, int N, int K, int C, float slope, float* out, void* stream) {
  LPFA_SIZES("pc3d_act_mean_f32");
  PC3D_REQUIRE(Z && out, "pc3d_act_mean_f32: null pointer");
  const int64_t total = (int64_t)B * N * (C / 4);
  unsigned blocks;
  if (int rc = lpfa_grid(total, &blocks, "pc3d_act_mean_f32")) return rc;
  hipLaunchKernelGGL(act_mean_fwd_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), (const float4*)Z, K, C / 4,
                     slope, (float4*)out, total);
  PC3D_LAUNCH_CHECK("pc3d_act_mean_f32");
  return PC3D_OK;
}

extern "C" int pc3d_act_mean_bwd_f32(const float* Z, const float* gout, int B, int N, int K, int C, float slope,
                                     float* gZ, void* stream) {
  LPFA_SIZES("pc3d_act_mean_bwd_f32");
  PC3D_REQUIRE(Z && gout && gZ, "pc3d_act_mean_bwd_f32: null pointer");
  const int64_t total = (int64_t)B * N * K * (C / 4);
  unsigned blocks;
  if (int rc = lpfa_grid(total, &blocks, "pc3d_act_mean_bwd_f32")) return rc;
  hipLaunchKernelGGL(act_mean_bwd_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), (const float4*)Z,
                     (const float4*)gout, K, C / 4, slope, (float4*)gZ, total);
  PC3D_LAUNCH_CHECK("pc3d_act_mean_bwd_f32");
  return PC3D_OK;
}
