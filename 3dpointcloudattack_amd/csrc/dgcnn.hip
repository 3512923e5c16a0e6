// K3 — DGCNN dynamic-graph ops (model/dgcnn.py:194-227 knn + get_graph_feature, :299-313 EdgeConv), gfx950.
//
// (1) knn_feat: k nearest neighbours in C-dimensional FEATURE space (C in {64,64,128} after the first layer), self
//     included, for channels-last features x [B,N,C]. The reference materialises -|xi-xj|^2 as a [B,N,N] matrix
//     (GEMM + 2 broadcasts) and calls topk. Here a workgroup owns 32 queries: the 32 x N similarity strip is
//     produced by v_mfma_f32_32x32x2_f32 (this IS a dense contraction over C) straight into LDS (<= 128 KiB, never
//     HBM), then each wave extracts the top-k of 4 strips with k rounds of a wave-wide arg-max.
// (2) gather_max: out[b,i,c] = max_{j in nbr(i)} P[b,idx[b,i,j],c] (and the arg-max for the backward), the
//     neighbour reduction of an EdgeConv rewritten as two point-wise GEMMs:
//         W [x_j - x_i ; x_i] = Wa x_j + (Wb - Wa) x_i = P_j + Q_i ,   W = [Wa | Wb]
//     so  max_j leaky(bn(P_j + Q_i)) = leaky(bn(max_j P_j + Q_i))  for a positive BN scale (min_j for a negative one):
//     the [B,2C,N,k] edge tensor (up to 671 MB at B=32) and the k-fold conv FLOPs disappear.
#include "pc3d_common.h"

namespace pc3d {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int KF_T = 512;      // 8 waves
constexpr int KF_Q = 32;       // queries per workgroup
constexpr int KF_MAXN = 1024;  // reference points per strip (LDS: 32 x 1024 x 4 B = 128 KiB)

struct KnnFeatArgs {
  const float* x;  // [B,N,C] channels-last
  int N, C, K;
  int32_t* idx;    // [B,N,K]
};

__global__ __launch_bounds__(KF_T) void knn_feat_kernel(KnnFeatArgs a) {
  __shared__ float strip[KF_Q][KF_MAXN + 1];  // similarity -|qi - rj|^2 ; +1 breaks the power-of-two row stride
  __shared__ float qn[KF_Q];
  const int b = blockIdx.y, q0 = blockIdx.x * KF_Q;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const float* xb = a.x + (int64_t)b * a.N * a.C;
  // A operand = query rows (fixed for the workgroup): lane (r,h) holds x[q0+r][8t+4h .. +3]
  const int qrow = (q0 + r < a.N) ? q0 + r : a.N - 1;
  const float* qp = xb + (int64_t)qrow * a.C + 4 * h;
  const int nt = a.C / 8;
  float4 aq[16];  // C <= 128
#pragma unroll
  for (int t = 0; t < 16; ++t)
    if (t < nt) aq[t] = *reinterpret_cast<const float4*>(qp + 8 * t);
  // squared norms of the queries (lane pairs r / r+32 hold complementary halves of the row)
  {
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t)
      if (t < nt) s += aq[t].x * aq[t].x + aq[t].y * aq[t].y + aq[t].z * aq[t].z + aq[t].w * aq[t].w;
    s += __shfl_xor(s, 32, 64);
    if (wave == 0 && h == 0) qn[r] = s;
  }
  __syncthreads();
  // each wave produces reference tiles wave, wave+8, ...
  const int ntile = (a.N + 31) / 32;
  for (int tile = wave; tile < ntile; tile += KF_T / 64) {
    const int r0 = tile * 32;
    const int rrow = (r0 + r < a.N) ? r0 + r : a.N - 1;
    const float* rp = xb + (int64_t)rrow * a.C + 4 * h;
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    float rn = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t)
      if (t < nt) {
        const float4 bv = *reinterpret_cast<const float4*>(rp + 8 * t);
        rn += bv.x * bv.x + bv.y * bv.y + bv.z * bv.z + bv.w * bv.w;
        // D[row = query][col = reference]
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[t].x, bv.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[t].y, bv.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[t].z, bv.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[t].w, bv.w, acc, 0, 0, 0);
      }
    rn += __shfl_xor(rn, 32, 64);  // |r_j|^2 for column j = r
    const bool valid = (r0 + r) < a.N;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int qi = (e & 3) + 8 * (e >> 2) + 4 * h;
      // model/dgcnn.py:195-197: -xx - (-2 x.x) - xx^T  = 2 q.r - |q|^2 - |r|^2
      strip[qi][r0 + r] = valid ? (2.f * acc[e] - qn[qi] - rn) : -__builtin_inff();
    }
  }
  __syncthreads();
  // top-K per query strip: wave w handles queries w, w+8, ... ; lane holds N/64 strided candidates
  constexpr int PER = KF_MAXN / 64;
  for (int qi = wave; qi < KF_Q; qi += KF_T / 64) {
    if (q0 + qi >= a.N) break;
    float v[PER];
#pragma unroll
    for (int e = 0; e < PER; ++e) {
      const int j = e * 64 + lane;
      v[e] = (j < a.N) ? strip[qi][j] : -__builtin_inff();
    }
    int32_t* o = a.idx + ((int64_t)b * a.N + q0 + qi) * a.K;
    for (int k = 0; k < a.K; ++k) {
      float bv = -__builtin_inff();
      int be = 0;
#pragma unroll
      for (int e = 0; e < PER; ++e)
        if (v[e] > bv) bv = v[e], be = e;  // ascending index inside the lane: strict > keeps the lowest
      int bj = be * 64 + lane;
      float wv = bv;
      int wj = bj;
#pragma unroll
      for (int o2 = 32; o2 > 0; o2 >>= 1) {
        const float ov = __shfl_xor(wv, o2, 64);
        const int oj = __shfl_xor(wj, o2, 64);
        if (ov > wv || (ov == wv && oj < wj)) wv = ov, wj = oj;
      }
      if (lane == 0) o[k] = wj;
      // the owner retires the winner (static register indexing: predicated writes)
      const bool mine = (wj & 63) == lane;
      const int we = wj >> 6;
#pragma unroll
      for (int e = 0; e < PER; ++e)
        if (mine && e == we) v[e] = -__builtin_inff();
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// gather-max / gather-min over neighbour lists, channels-last. sign[c] >= 0 -> max, < 0 -> min (BN scale sign).
// ---------------------------------------------------------------------------------------------------------
struct GMaxArgs {
  const float* P;       // [B,N,C]
  const int32_t* idx;   // [B,N,K]
  const float* sign;    // [C] or null (all max)
  int N, C, K;
  float* out;           // [B,N,C]
  int32_t* arg;         // [B,N,C] winning neighbour (absolute point index) or null
};

__global__ __launch_bounds__(256) void gather_max_kernel(GMaxArgs a) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (i >= a.N) return;
  const int32_t* nb = a.idx + ((int64_t)b * a.N + i) * a.K;
  const float* Pb = a.P + (int64_t)b * a.N * a.C;
  for (int c = lane; c < a.C; c += 64) {
    const bool mx = (a.sign == nullptr) || (a.sign[c] >= 0.f);
    float best = mx ? -__builtin_inff() : __builtin_inff();
    int bj = nb[0];
    for (int k = 0; k < a.K; ++k) {
      const int j = nb[k];
      const float v = Pb[(int64_t)j * a.C + c];
      if (mx ? (v > best) : (v < best)) best = v, bj = j;
    }
    a.out[((int64_t)b * a.N + i) * a.C + c] = best;
    if (a.arg) a.arg[((int64_t)b * a.N + i) * a.C + c] = bj;
  }
}

// backward: gP[b, arg[b,i,c], c] += g[b,i,c]   (gP zero-filled first)
__global__ __launch_bounds__(256) void gather_max_bwd_kernel(const float* g, const int32_t* arg, int N, int C, float* gP) {
  const int b = blockIdx.y;
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (int64_t)N * C) return;
  const int c = (int)(e % C);
  const int j = arg[(int64_t)b * N * C + e];
  atomicAdd(gP + ((int64_t)b * N + j) * C + c, g[(int64_t)b * N * C + e]);
}

}  // namespace pc3d

using namespace pc3d;

extern "C" int pc3d_knn_feat_f32(const float* x, int B, int N, int C, int K, int32_t* idx, void* stream) {
  PC3D_REQUIRE(B >= 0 && N >= 1 && K >= 1 && K <= N, "pc3d_knn_feat_f32: bad sizes B=%d N=%d K=%d", B, N, K);
  PC3D_REQUIRE(C >= 8 && C <= 128 && C % 8 == 0, "pc3d_knn_feat_f32: C=%d must be a multiple of 8 in [8,128]", C);
  PC3D_REQUIRE(N <= KF_MAXN, "pc3d_knn_feat_f32: N=%d exceeds %d", N, KF_MAXN);
  PC3D_REQUIRE(B <= 65535, "pc3d_knn_feat_f32: B=%d exceeds grid.y limit", B);
  if (B == 0) return PC3D_OK;
  PC3D_REQUIRE(x && idx, "pc3d_knn_feat_f32: null pointer");
  KnnFeatArgs a{x, N, C, K, idx};
  hipLaunchKernelGGL(knn_feat_kernel, dim3(cdiv(N, KF_Q), B), dim3(KF_T), 0, as_stream(stream), a);
  PC3D_LAUNCH_CHECK("pc3d_knn_feat_f32");
  return PC3D_OK;
}

extern "C" int pc3d_gather_max_f32(const float* P, const int32_t* idx, const float* sign, int B, int N, int C, int K,
                                   float* out, int32_t* arg, void* stream) {
  PC3D_REQUIRE(B >= 0 && N >= 1 && C >= 1 && K >= 1, "pc3d_gather_max_f32: bad sizes");
  PC3D_REQUIRE(B <= 65535, "pc3d_gather_max_f32: B=%d exceeds grid.y limit", B);
  if (B == 0) return PC3D_OK;
  PC3D_REQUIRE(P && idx && out, "pc3d_gather_max_f32: null pointer");
  GMaxArgs a{P, idx, sign, N, C, K, out, arg};
  hipLaunchKernelGGL(gather_max_kernel, dim3(cdiv(N, 4), B), dim3(256), 0, as_stream(stream), a);
  PC3D_LAUNCH_CHECK("pc3d_gather_max_f32");
  return PC3D_OK;
}

extern "C" int pc3d_gather_max_bwd_f32(const float* g, const int32_t* arg, int B, int N, int C, float* gP,
                                       void* stream) {
  PC3D_REQUIRE(B >= 0 && N >= 1 && C >= 1, "pc3d_gather_max_bwd_f32: bad sizes");
  PC3D_REQUIRE(B <= 65535, "pc3d_gather_max_bwd_f32: B=%d exceeds grid.y limit", B);
  if (B == 0) return PC3D_OK;
  PC3D_REQUIRE(g && arg && gP, "pc3d_gather_max_bwd_f32: null pointer");
  hipStream_t st = as_stream(stream);
  hipError_t e = hipMemsetAsync(gP, 0, (size_t)B * N * C * sizeof(float), st);
  if (e != hipSuccess) {
    set_error("pc3d_gather_max_bwd_f32: memset failed: %s", hipGetErrorString(e));
    return (int)e;
  }
  hipLaunchKernelGGL(gather_max_bwd_kernel, dim3((unsigned)(((int64_t)N * C + 255) / 256), B), dim3(256), 0, st, g, arg,
                     N, C, gP);
  PC3D_LAUNCH_CHECK("pc3d_gather_max_bwd_f32");
  return PC3D_OK;
}
