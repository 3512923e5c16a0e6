// K12 — graph Laplacian of a point cloud for the AOF spectral front-end (attack/AOF/TAOF_attack.py:31-52):
// A_ij = exp(-|p_i - p_j|^2) on the symmetrised kNN graph, L = D - A, dense [B,N,N] (the eigensolver wants it dense).
// The reference materialises the [B,N,N,3] difference tensor (403 MB at B=32, N=1024), a dense A and a dense mask; here
// only the O(N k) graph edges are evaluated and scattered into a zero-filled L, then one pass fixes the diagonal.
#include "pc3d_common.h"

namespace pc3d {

struct LapArgs {
  PtsView x;
  const int32_t* idx;  // [B,N,K] (self may be included; self-loops cancel in D - A)
  int N, K;
  float* L;            // [B,N,N], zero-filled before edges_kernel
};

__global__ __launch_bounds__(256) void lap_edges_kernel(LapArgs a) {
  const int b = blockIdx.y;
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= a.N * a.K) return;
  const int i = e / a.K;
  const int j = a.idx[(int64_t)b * a.N * a.K + e];
  if (j == i) return;
  const float* pi = a.x.p + (int64_t)b * a.x.bs + (int64_t)i * a.x.ps;
  const float* pj = a.x.p + (int64_t)b * a.x.bs + (int64_t)j * a.x.ps;
  const float dx = pi[0] - pj[0], dy = pi[a.x.cs] - pj[a.x.cs], dz = pi[2 * a.x.cs] - pj[2 * a.x.cs];
  const float w = -expf(-((dx * dx + dy * dy) + dz * dz));
  float* Lb = a.L + (int64_t)b * a.N * a.N;
  // symmetrised mask: (i,j) and (j,i) both get the same value; concurrent writers write identical bits
  Lb[(int64_t)i * a.N + j] = w;
  Lb[(int64_t)j * a.N + i] = w;
}

// L_ii = -sum_{j != i} L_ij : one wave per row
__global__ __launch_bounds__(256) void lap_diag_kernel(float* L, int N) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (i >= N) return;
  float* row = L + ((int64_t)b * N + i) * N;
  float s = 0.f;
  for (int j = lane; j < N; j += 64)
    if (j != i) s += row[j];
  s = wave_sum(s);
  if (lane == 0) row[i] = -s;
}

}  // namespace pc3d

using namespace pc3d;

extern "C" int pc3d_graph_laplacian_f32(const float* xyz, int64_t x_bs, int64_t x_ps, int64_t x_cs,
                                        const int32_t* idx, int B, int N, int K, float* L, void* stream) {
  PC3D_REQUIRE(B >= 0 && N >= 1 && K >= 1, "pc3d_graph_laplacian_f32: bad sizes B=%d N=%d K=%d", B, N, K);
  PC3D_REQUIRE(B <= 65535, "pc3d_graph_laplacian_f32: B=%d exceeds grid.y limit", B);
  if (B == 0) return PC3D_OK;
  PC3D_REQUIRE(xyz && idx && L, "pc3d_graph_laplacian_f32: null pointer");
  hipStream_t st = as_stream(stream);
  hipError_t e = zero_async(L, (size_t)B * N * N, st);
  if (e != hipSuccess) {
    set_error("pc3d_graph_laplacian_f32: zero fill failed: %s", hipGetErrorString(e));
    return (int)e;
  }
  LapArgs a{{xyz, x_bs, x_ps, x_cs}, idx, N, K, L};
  hipLaunchKernelGGL(lap_edges_kernel, dim3(cdiv(N * K, 256), B), dim3(256), 0, st, a);
  PC3D_LAUNCH_CHECK("pc3d_graph_laplacian_f32/edges");
  hipLaunchKernelGGL(lap_diag_kernel, dim3(cdiv(N, 4), B), dim3(256), 0, st, L, N);
  PC3D_LAUNCH_CHECK("pc3d_graph_laplacian_f32/diag");
  return PC3D_OK;
}
