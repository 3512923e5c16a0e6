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
template <bool SCATTER>   // false: only gBc (the deterministic path scatters gA in a second, ordered launch: det.hip)
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
      if (SCATTER) atomicAdd(ga + (int64_t)min(max(r[u], 0), N - 1) * C, v);
    }
  }
  for (; j < K; ++j) {
    const float g = ge[(int64_t)j * C];
    const float v = ev[(int64_t)j * C] > 0.f ? g : g * slope;
    acc += v;
    if (SCATTER) atomicAdd(ga + (int64_t)min(max(id[j], 0), N - 1) * C, v);
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

// ---------------------------------------------------------------------------------------------------------
// The whole LPFA block (one 1x1-conv layer, C -> C) in ONE launch each way — no [B,N,k,C] tensor at all:
//     out[b,i,:] = mean_j leaky_s2( W . leaky_s1(A[b,idx[b,i,j],:] + Bc[b,i,:]) + bias )
// (edge_act + the point-wise GEMM + act_mean above move 3 x 84 MB at B=32, N=1024, k=20, C=32 and are bound by that).
// A workgroup owns P = 256 / C points: their k activated edge rows are gathered once into LDS (20 KB at k = 20), then
// thread (point, output channel) keeps its weight row in registers and walks the edges with broadcast ds_read_b128.
// (Two output channels per thread — half the LDS reads per FMA, 40 KB tiles — measured SLOWER: 36 -> 51 us forward,
// 124 -> 168 backward at B=32, N=1024, C=32: the kernels are bound by the latency of the staged gather, which more,
// smaller workgroups hide better, not by LDS bandwidth.)
// Backward: the edges and pre-activations are recomputed, dZ goes through LDS, thread (point, INPUT channel) forms
// dE = W^T dZ, applies the first activation's mask and scatters into gA (row-contiguous float atomics, as
// edge_act_bwd_kernel) / sums into gBc.
// ---------------------------------------------------------------------------------------------------------
struct LpfaFusedArgs {
  const float* A;      // [B,N,C]
  const float* Bc;     // [B,N,C]
  const int* idx;      // [B,N,K]
  const float* W;      // [C,C] (out, in)
  const float* Wt;     // [C,C] transposed (backward)
  const float* bias;   // [C]
  int N, K;
  float s1, s2;
  float* out;          // [B,N,C]
  const float* gout;   // [B,N,C]
  float* gA;           // [B,N,C] zero-filled by the entry point
  float* gBc;          // [B,N,C]
  float* dpre;         // [B,N,K,C] or null: deterministic mode — the per-edge gradients are stored here and summed into gA
                       // by the ordered LDS scatter of det.hip instead of being scattered with float atomics
};

template <int C>
__device__ __forceinline__ void lpfa_stage_edges(const LpfaFusedArgs& a, int b, int i0, float* E, int* nbr) {
  constexpr int P = 256 / C, C4 = C / 4;
  const int K = a.K;
  for (int t = threadIdx.x; t < P * K; t += 256) {
    const int p = t / K, j = t - p * K, i = i0 + p;
    nbr[t] = i < a.N ? min(max(a.idx[((int64_t)b * a.N + i) * K + j], 0), a.N - 1) : 0;
  }
  __syncthreads();
  for (int f = threadIdx.x; f < P * K * C4; f += 256) {
    const int c4 = f % C4, pj = f / C4;
    const int p = pj / K, i = i0 + p;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < a.N) {
      const float4 x = *reinterpret_cast<const float4*>(a.A + ((int64_t)b * a.N + nbr[pj]) * C + 4 * c4);
      const float4 y = *reinterpret_cast<const float4*>(a.Bc + ((int64_t)b * a.N + i) * C + 4 * c4);
      v = leaky4(make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w), a.s1);
    }
    *reinterpret_cast<float4*>(E + (int64_t)pj * C + 4 * c4) = v;
  }
  __syncthreads();
}

template <int C>
__device__ __forceinline__ float lpfa_dot(const float (&w)[C], const float* __restrict__ row) {
  float z = 0.f;
#pragma unroll
  for (int c = 0; c < C; c += 4) {
    const float4 e = *reinterpret_cast<const float4*>(row + c);
    z = __builtin_fmaf(w[c], e.x, z);
    z = __builtin_fmaf(w[c + 1], e.y, z);
    z = __builtin_fmaf(w[c + 2], e.z, z);
    z = __builtin_fmaf(w[c + 3], e.w, z);
  }
  return z;
}

template <int C>
__global__ __launch_bounds__(256) void lpfa_fused_fwd_kernel(LpfaFusedArgs a) {
  constexpr int P = 256 / C;
  extern __shared__ __attribute__((aligned(16))) float lf_lds[];
  float* E = lf_lds;                                        // [P][K][C]
  int* nbr = reinterpret_cast<int*>(lf_lds + P * a.K * C);   // [P][K]
  int bx_, by_;
  xcd_swizzle(bx_, by_);      // a cloud's workgroups on one XCD (pc3d_common.h)
  const int b = by_, i0 = bx_ * P;
  lpfa_stage_edges<C>(a, b, i0, E, nbr);
  const int p = threadIdx.x / C, c = threadIdx.x - p * C, i = i0 + p;
  float w[C];
#pragma unroll
  for (int k = 0; k < C; k += 4) {
    const float4 v = *reinterpret_cast<const float4*>(a.W + (int64_t)c * C + k);
    w[k] = v.x, w[k + 1] = v.y, w[k + 2] = v.z, w[k + 3] = v.w;
  }
  const float bias = a.bias ? a.bias[c] : 0.f;
  float acc = 0.f;
  for (int j = 0; j < a.K; ++j) {
    const float z = lpfa_dot<C>(w, E + (int64_t)(p * a.K + j) * C) + bias;
    acc += z > 0.f ? z : a.s2 * z;
  }
  if (i < a.N) a.out[((int64_t)b * a.N + i) * C + c] = acc * (1.f / (float)a.K);
}

template <int C, bool DET>
__global__ __launch_bounds__(256) void lpfa_fused_bwd_kernel(LpfaFusedArgs a) {
  constexpr int P = 256 / C;
  extern __shared__ __attribute__((aligned(16))) float lf_lds[];
  float* E = lf_lds;                                        // [P][K][C]
  float* dZ = lf_lds + P * a.K * C;                          // [P][K][C]
  int* nbr = reinterpret_cast<int*>(lf_lds + 2 * P * a.K * C);
  int bx_, by_;
  xcd_swizzle(bx_, by_);      // a cloud's workgroups on one XCD (pc3d_common.h)
  const int b = by_, i0 = bx_ * P;
  lpfa_stage_edges<C>(a, b, i0, E, nbr);
  const int p = threadIdx.x / C, c = threadIdx.x - p * C, i = i0 + p;
  float w[C];
  {   // phase 1: thread (point, OUTPUT channel): dZ_j = g / K * leaky'(z_j)
#pragma unroll
    for (int k = 0; k < C; k += 4) {
      const float4 v = *reinterpret_cast<const float4*>(a.W + (int64_t)c * C + k);
      w[k] = v.x, w[k + 1] = v.y, w[k + 2] = v.z, w[k + 3] = v.w;
    }
    const float bias = a.bias ? a.bias[c] : 0.f;
    const float gk = i < a.N ? a.gout[((int64_t)b * a.N + i) * C + c] * (1.f / (float)a.K) : 0.f;
    for (int j = 0; j < a.K; ++j) {
      const float z = lpfa_dot<C>(w, E + (int64_t)(p * a.K + j) * C) + bias;
      dZ[(int64_t)(p * a.K + j) * C + c] = z > 0.f ? gk : a.s2 * gk;
    }
  }
  __syncthreads();
  {   // phase 2: thread (point, INPUT channel): dE_j = W^T dZ_j, first activation's mask, scatter / sum
#pragma unroll
    for (int k = 0; k < C; k += 4) {
      const float4 v = *reinterpret_cast<const float4*>(a.Wt + (int64_t)c * C + k);
      w[k] = v.x, w[k + 1] = v.y, w[k + 2] = v.z, w[k + 3] = v.w;
    }
    float sum = 0.f;
    float* ga = a.gA + (int64_t)b * a.N * C + c;
    for (int j = 0; j < a.K; ++j) {
      const float dE = lpfa_dot<C>(w, dZ + (int64_t)(p * a.K + j) * C);
      const float e = E[(int64_t)(p * a.K + j) * C + c];
      const float dpre = e > 0.f ? dE : a.s1 * dE;
      sum += dpre;
      if (i < a.N) {
        if (DET) a.dpre[(((int64_t)b * a.N + i) * a.K + j) * C + c] = dpre;
        else atomicAdd(ga + (int64_t)nbr[p * a.K + j] * C, dpre);
      }
    }
    if (i < a.N) a.gBc[((int64_t)b * a.N + i) * C + c] = sum;
  }
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
                                     float slope, float* gA, float* gBc, int deterministic, const int32_t* rev_off,
                                     const int32_t* rev_lst, void* stream) {
  LPFA_SIZES("pc3d_edge_act_bwd_f32");
  PC3D_REQUIRE(gE && E && idx && gA && gBc, "pc3d_edge_act_bwd_f32: null pointer");
  const int64_t total = (int64_t)B * N * C;
  unsigned blocks;
  if (int rc = lpfa_grid(total, &blocks, "pc3d_edge_act_bwd_f32")) return rc;
  if (deterministic) {   // gBc as before; gA (overwritten: no zero fill needed) by the ordered LDS scatter over the N*K edges
    PC3D_REQUIRE((int64_t)N * K <= 0x7fffffffLL, "pc3d_edge_act_bwd_f32: N * K too large");
    hipLaunchKernelGGL(edge_act_bwd_kernel<false>, dim3(blocks), dim3(256), 0, as_stream(stream), gE, E, idx, N, K, C, slope,
                       gA, gBc, total);
    PC3D_LAUNCH_CHECK("pc3d_edge_act_bwd_f32");
    if (rev_off && rev_lst)   // sorted reverse index of idx (pc3d_rev_index_i32, clamp = 1): every gA row gathers its edges
      return rev_gather_sum("pc3d_edge_act_bwd_f32", gE, C, E, C, slope, rev_off, rev_lst, B, N * K, N, C, gA, C, stream);
    return scatter_rows_det("pc3d_edge_act_bwd_f32", idx, gE, C, E, C, slope, B, N * K, N, C, gA, C, 0, 1, stream);
  }
  hipLaunchKernelGGL(edge_act_bwd_kernel<true>, dim3(blocks), dim3(256), 0, as_stream(stream), gE, E, idx, N, K, C, slope,
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

static bool lpfa_fused_ok(int B, int N, int K, int C) {
  return B >= 0 && B <= 65535 && N >= 1 && K >= 1 && K <= 30 && (C == 16 || C == 32 || C == 64 || C == 128);   // 2 x 1 KiB x K of LDS (backward)
}
#define PC3D_LF_DISPATCH(KERNEL, ...)                                         \
  switch (C) {                                                                \
    case 16: hipLaunchKernelGGL((KERNEL<16>), __VA_ARGS__); break;            \
    case 32: hipLaunchKernelGGL((KERNEL<32>), __VA_ARGS__); break;            \
    case 64: hipLaunchKernelGGL((KERNEL<64>), __VA_ARGS__); break;            \
    default: hipLaunchKernelGGL((KERNEL<128>), __VA_ARGS__); break;           \
  }
#define PC3D_LF_DISPATCH2(KERNEL, DET, ...)                                   \
  switch (C) {                                                                \
    case 16: hipLaunchKernelGGL((KERNEL<16, DET>), __VA_ARGS__); break;       \
    case 32: hipLaunchKernelGGL((KERNEL<32, DET>), __VA_ARGS__); break;       \
    case 64: hipLaunchKernelGGL((KERNEL<64, DET>), __VA_ARGS__); break;       \
    default: hipLaunchKernelGGL((KERNEL<128, DET>), __VA_ARGS__); break;      \
  }

extern "C" int pc3d_lpfa_fused_f32(const float* A, const float* Bc, const int32_t* idx, const float* W, const float* bias,
                                   int B, int N, int K, int C, float slope1, float slope2, float* out, void* stream) {
  PC3D_REQUIRE(lpfa_fused_ok(B, N, K, C), "pc3d_lpfa_fused_f32: bad sizes B=%d N=%d K=%d C=%d (C in {16,32,64,128}, K <= 30)",
               B, N, K, C);
  if (B == 0) return PC3D_OK;
  PC3D_REQUIRE(A && Bc && idx && W && out, "pc3d_lpfa_fused_f32: null pointer");
  LpfaFusedArgs a{A, Bc, idx, W, nullptr, bias, N, K, slope1, slope2, out, nullptr, nullptr, nullptr, nullptr};
  const int P = 256 / C;
  const size_t lds = (size_t)P * K * C * sizeof(float) + (size_t)P * K * sizeof(int);
  PC3D_LF_DISPATCH(lpfa_fused_fwd_kernel, dim3(cdiv(N, P), B), dim3(256), lds, as_stream(stream), a);
  PC3D_LAUNCH_CHECK("pc3d_lpfa_fused_f32");
  return PC3D_OK;
}

extern "C" int pc3d_lpfa_fused_bwd_f32(const float* gout, const float* A, const float* Bc, const int32_t* idx,
                                       const float* W, const float* Wt, const float* bias, int B, int N, int K, int C,
                                       float slope1, float slope2, float* gA, float* gBc, float* edge_scratch,
                                       const int32_t* rev_off, const int32_t* rev_lst, void* stream) {
  PC3D_REQUIRE(lpfa_fused_ok(B, N, K, C),
               "pc3d_lpfa_fused_bwd_f32: bad sizes B=%d N=%d K=%d C=%d (C in {16,32,64,128}, K <= 30)", B, N, K, C);
  if (B == 0) return PC3D_OK;
  PC3D_REQUIRE(gout && A && Bc && idx && W && Wt && gA && gBc, "pc3d_lpfa_fused_bwd_f32: null pointer");
  hipStream_t st = as_stream(stream);
  LpfaFusedArgs a{A, Bc, idx, W, Wt, bias, N, K, slope1, slope2, nullptr, gout, gA, gBc, edge_scratch};
  const int P = 256 / C;
  const size_t lds = (size_t)2 * P * K * C * sizeof(float) + (size_t)P * K * sizeof(int);
  if (edge_scratch) {    // deterministic: per-edge gradients to the scratch, then the ordered LDS scatter (gA overwritten)
    PC3D_REQUIRE((int64_t)N * K <= 0x7fffffffLL, "pc3d_lpfa_fused_bwd_f32: N * K too large");
    PC3D_LF_DISPATCH2(lpfa_fused_bwd_kernel, true, dim3(cdiv(N, P), B), dim3(256), lds, st, a);
    PC3D_LAUNCH_CHECK("pc3d_lpfa_fused_bwd_f32");
    if (rev_off && rev_lst)   // sorted reverse index of idx (pc3d_rev_index_i32, clamp = 1): every gA row gathers its edges
      return rev_gather_sum("pc3d_lpfa_fused_bwd_f32", edge_scratch, C, nullptr, 0, 0.f, rev_off, rev_lst, B, N * K, N, C, gA, C, stream);
    return scatter_rows_det("pc3d_lpfa_fused_bwd_f32", idx, edge_scratch, C, nullptr, 0, 0.f, B, N * K, N, C, gA, C, 0, 1, stream);
  }
  if (hipError_t e = zero_async(gA, (size_t)B * N * C, st); e != hipSuccess) {
    set_error("pc3d_lpfa_fused_bwd_f32: zero fill failed: %s", hipGetErrorString(e));
    return (int)e;
  }
  PC3D_LF_DISPATCH2(lpfa_fused_bwd_kernel, false, dim3(cdiv(N, P), B), dim3(256), lds, st, a);
  PC3D_LAUNCH_CHECK("pc3d_lpfa_fused_bwd_f32");
  return PC3D_OK;
}
