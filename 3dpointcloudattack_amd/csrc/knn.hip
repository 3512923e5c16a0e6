// K2/K4 — K nearest reference points of every query (xyz, squared L2, direct-difference form), sorted ascending,
// without the [B,N,M] matrix + topk of the reference:
//   attack/CW/CW_utils/dist_utils.py:133-144 (KNNDist, k+1=6), attack/GeoA3/knn_utils.py:10-55 (knn_points,
//   K in {1,4,17}), attack/AOF/TAOF_attack.py:13-28 (k=30), model/dgcnn.py:194-200 on xyz (k=20),
//   model/curvenet_util.py:10-17.
//
// Same skeleton as the NN kernel: workgroup = (batch, 64 queries), reference cloud staged through LDS as SoA,
// the four waves scan one quarter of each LDS tile for the SAME 64 queries (one per lane). Each lane keeps a
// sorted K-list in registers (compile-time KMAX in {8,16,32}); a candidate is inserted only when it beats the
// lane's current K-th distance (wave-uniform skip otherwise). The four partial lists are merged through LDS.
// Ties: the lower reference index comes first.
#include "pc3d_common.h"

namespace pc3d {

constexpr int kKnnThreads = 256;
constexpr int kKnnWaves = 4;
constexpr int kKnnMaxTile = 4096;
constexpr float kKnnFar = 1.0e18f;

struct KnnArgs {
  PtsView q, r;
  int N, M, K;
  float* d;    // [B,N,K]
  int32_t* i;  // [B,N,K]
};

template <int KMAX>
__device__ __forceinline__ void knn_insert(float (&bd)[KMAX], int (&bi)[KMAX], float d, int idx) {
#pragma unroll
  for (int j = KMAX - 1; j > 0; --j) {
    const bool shift = d < bd[j - 1];
    const bool here = d < bd[j];
    bd[j] = shift ? bd[j - 1] : (here ? d : bd[j]);
    bi[j] = shift ? bi[j - 1] : (here ? idx : bi[j]);
  }
  const bool first = d < bd[0];
  bd[0] = first ? d : bd[0];
  bi[0] = first ? idx : bi[0];
}

template <int KMAX>
__global__ __launch_bounds__(kKnnThreads) void knn_kernel(KnnArgs a, int mt_cap) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int N = a.N, M = a.M, K = a.K;
  const int q0 = blockIdx.x * kWave;
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int mt = M < mt_cap ? M : mt_cap;
  const int slice = ((mt + kKnnWaves * 4 - 1) / (kKnnWaves * 4)) * 4;
  const int mt_pad = slice * kKnnWaves;
  float* sx = lds;
  float* sy = lds + mt_pad;
  float* sz = lds + 2 * mt_pad;

  int qi = q0 + lane;
  if (qi >= N) qi = N - 1;
  const float* qp = a.q.p + (int64_t)b * a.q.bs + (int64_t)qi * a.q.ps;
  const float qx = qp[0], qy = qp[a.q.cs], qz = qp[2 * a.q.cs];

  float bd[KMAX];
  int bi[KMAX];
#pragma unroll
  for (int j = 0; j < KMAX; ++j) {
    bd[j] = __builtin_inff();
    bi[j] = 0x7fffffff;
  }
  // only the first K slots matter: slots >= K would merely waste work, so the acceptance threshold is slot K-1
  // (KMAX == K rounded up; unused tail slots just carry larger values)

  const float* rb = a.r.p + (int64_t)b * a.r.bs;
  for (int m0 = 0; m0 < M; m0 += mt) {
    __syncthreads();
    for (int j = threadIdx.x; j < mt_pad; j += kKnnThreads) {
      const int m = m0 + j;
      float x = kKnnFar, y = kKnnFar, z = kKnnFar;
      if (j < mt && m < M) {
        const float* rp = rb + (int64_t)m * a.r.ps;
        x = rp[0], y = rp[a.r.cs], z = rp[2 * a.r.cs];
      }
      sx[j] = x, sy[j] = y, sz[j] = z;
    }
    __syncthreads();
    const int s0 = wave * slice;
    for (int j = s0; j < s0 + slice; j += 4) {
      const float4 rx = *reinterpret_cast<const float4*>(sx + j);
      const float4 ry = *reinterpret_cast<const float4*>(sy + j);
      const float4 rz = *reinterpret_cast<const float4*>(sz + j);
      const float rxa[4] = {rx.x, rx.y, rx.z, rx.w};
      const float rya[4] = {ry.x, ry.y, ry.z, ry.w};
      const float rza[4] = {rz.x, rz.y, rz.z, rz.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float dx = rxa[e] - qx, dy = rya[e] - qy, dz = rza[e] - qz;
        float d = dx * dx;
        d = __builtin_fmaf(dy, dy, d);
        d = __builtin_fmaf(dz, dz, d);
        const int m = m0 + j + e;
        const bool valid = (j + e < mt) && (m < M);  // padding never enters a list
        if (__builtin_amdgcn_ballot_w64(valid && d < bd[KMAX - 1]) != 0ull) {
          if (valid) knn_insert<KMAX>(bd, bi, d, m);
        }
      }
    }
  }

  // ---- merge the four waves' sorted lists: LDS layout [wave][k][lane] (conflict-free per k)
  __syncthreads();
  float* cd = lds;
  int* ci = reinterpret_cast<int*>(lds + kKnnWaves * KMAX * kWave);
#pragma unroll
  for (int j = 0; j < KMAX; ++j) {
    cd[(wave * KMAX + j) * kWave + lane] = bd[j];
    ci[(wave * KMAX + j) * kWave + lane] = bi[j];
  }
  __syncthreads();
  if (wave == 0) {
    int pos[kKnnWaves] = {0, 0, 0, 0};
    const int qo = q0 + lane;
    for (int k = 0; k < K; ++k) {
      float best = __builtin_inff();
      int besti = 0x7fffffff, bw = 0;
#pragma unroll
      for (int w = 0; w < kKnnWaves; ++w) {
        const int p = pos[w];
        const float d = (p < KMAX) ? cd[(w * KMAX + p) * kWave + lane] : __builtin_inff();
        const int i = (p < KMAX) ? ci[(w * KMAX + p) * kWave + lane] : 0x7fffffff;
        if (d < best || (d == best && i < besti)) {
          best = d, besti = i, bw = w;
        }
      }
#pragma unroll
      for (int w = 0; w < kKnnWaves; ++w) pos[w] += (w == bw) ? 1 : 0;
      if (qo < N) {
        // fewer than K reference points: pad with the last valid neighbour (M >= 1 guaranteed)
        if (a.d) a.d[((int64_t)b * N + qo) * K + k] = best;
        if (a.i) a.i[((int64_t)b * N + qo) * K + k] = besti;
      }
    }
  }
}

template <int KMAX>
static int knn_launch(const KnnArgs& a, int B, hipStream_t st) {
  const int mt = a.M < kKnnMaxTile ? a.M : kKnnMaxTile;
  const int slice = ((mt + kKnnWaves * 4 - 1) / (kKnnWaves * 4)) * 4;
  const size_t tile = (size_t)3 * slice * kKnnWaves * sizeof(float);
  const size_t merge = (size_t)kKnnWaves * KMAX * kWave * 8;
  const size_t lds = tile > merge ? tile : merge;
  hipLaunchKernelGGL(knn_kernel<KMAX>, dim3(cdiv(a.N, kWave), B), dim3(kKnnThreads), lds, st, a, kKnnMaxTile);
  PC3D_LAUNCH_CHECK("pc3d_knn_f32");
  return PC3D_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Backward of the K distances: d[b,i,k] = |q_i - r_idx[i,k]|^2 with upstream w[b,i,k].
//   grad_q[i]  = sum_k 2 w[i,k] (q_i - r_idx[i,k])                       (dense, overwrite)
//   grad_r[j]  = sum_{(i,k): idx[i,k]==j} 2 w[i,k] (r_j - q_i)           (scatter; atomics or ordered scan)
// ---------------------------------------------------------------------------------------------------------
struct KnnBwdArgs {
  PtsView q, r;
  int N, M, K;
  const int32_t* idx;  // [B,N,K]
  const float* w;      // [B,N,K]
  PtsViewMut gq, gr;
};

__global__ __launch_bounds__(256) void knn_bwd_q_kernel(KnnBwdArgs a) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  const int nmax = a.N > a.M ? a.N : a.M;
  if (i >= nmax) return;
  if (a.gr.p && i < a.M) {  // zero-fill grad_r (the scatter pass accumulates into it)
    float* g = a.gr.p + (int64_t)b * a.gr.bs + (int64_t)i * a.gr.ps;
    g[0] = 0.f, g[a.gr.cs] = 0.f, g[2 * a.gr.cs] = 0.f;
  }
  if (!a.gq.p || i >= a.N) return;
  const float* qp = a.q.p + (int64_t)b * a.q.bs + (int64_t)i * a.q.ps;
  const float qx = qp[0], qy = qp[a.q.cs], qz = qp[2 * a.q.cs];
  float gx = 0.f, gy = 0.f, gz = 0.f;
  const int64_t base = ((int64_t)b * a.N + i) * a.K;
  for (int k = 0; k < a.K; ++k) {
    const float w = 2.f * a.w[base + k];
    const float* rp = a.r.p + (int64_t)b * a.r.bs + (int64_t)a.idx[base + k] * a.r.ps;
    gx += w * (qx - rp[0]);
    gy += w * (qy - rp[a.r.cs]);
    gz += w * (qz - rp[2 * a.r.cs]);
  }
  float* g = a.gq.p + (int64_t)b * a.gq.bs + (int64_t)i * a.gq.ps;
  g[0] = gx, g[a.gq.cs] = gy, g[2 * a.gq.cs] = gz;
}

__global__ __launch_bounds__(256) void knn_bwd_r_atomic_kernel(KnnBwdArgs a) {
  const int t = blockIdx.x * 256 + threadIdx.x;  // over N*K
  const int b = blockIdx.y;
  if (t >= a.N * a.K) return;
  const int i = t / a.K;
  const int64_t e = (int64_t)b * a.N * a.K + t;
  const float w = 2.f * a.w[e];
  if (w == 0.f) return;
  const int j = a.idx[e];
  const float* qp = a.q.p + (int64_t)b * a.q.bs + (int64_t)i * a.q.ps;
  const float* rp = a.r.p + (int64_t)b * a.r.bs + (int64_t)j * a.r.ps;
  float* g = a.gr.p + (int64_t)b * a.gr.bs + (int64_t)j * a.gr.ps;
  atomicAdd(g, w * (rp[0] - qp[0]));
  atomicAdd(g + a.gr.cs, w * (rp[a.r.cs] - qp[a.q.cs]));
  atomicAdd(g + 2 * a.gr.cs, w * (rp[2 * a.r.cs] - qp[2 * a.q.cs]));
}

__global__ __launch_bounds__(256) void knn_bwd_r_det_kernel(KnnBwdArgs a) {
  __shared__ int s_idx[1024];
  __shared__ float s_w[1024];
  const int b = blockIdx.y;
  const int j = blockIdx.x * 256 + threadIdx.x;
  const bool live = j < a.M;
  float rx = 0.f, ry = 0.f, rz = 0.f;
  if (live) {
    const float* rp = a.r.p + (int64_t)b * a.r.bs + (int64_t)j * a.r.ps;
    rx = rp[0], ry = rp[a.r.cs], rz = rp[2 * a.r.cs];
  }
  float gx = 0.f, gy = 0.f, gz = 0.f;
  const int total = a.N * a.K;
  for (int t0 = 0; t0 < total; t0 += 1024) {
    __syncthreads();
    for (int t = threadIdx.x; t < 1024; t += 256) {
      const int e = t0 + t;
      s_idx[t] = (e < total) ? a.idx[(int64_t)b * total + e] : -1;
      s_w[t] = (e < total) ? 2.f * a.w[(int64_t)b * total + e] : 0.f;
    }
    __syncthreads();
    const int lim = (total - t0) < 1024 ? (total - t0) : 1024;
    for (int t = 0; t < lim; ++t) {
      if (s_idx[t] == j) {
        const int i = (t0 + t) / a.K;
        const float* qp = a.q.p + (int64_t)b * a.q.bs + (int64_t)i * a.q.ps;
        const float w = s_w[t];
        gx += w * (rx - qp[0]);
        gy += w * (ry - qp[a.q.cs]);
        gz += w * (rz - qp[2 * a.q.cs]);
      }
    }
  }
  if (live) {
    float* g = a.gr.p + (int64_t)b * a.gr.bs + (int64_t)j * a.gr.ps;
    g[0] = gx, g[a.gr.cs] = gy, g[2 * a.gr.cs] = gz;
  }
}

}  // namespace pc3d

using namespace pc3d;

extern "C" int pc3d_knn_bwd_f32(const float* q, int64_t q_bs, int64_t q_ps, int64_t q_cs,
                                const float* r, int64_t r_bs, int64_t r_ps, int64_t r_cs,
                                int B, int N, int M, int K, const int32_t* idx, const float* w,
                                float* grad_q, int64_t gq_bs, int64_t gq_ps, int64_t gq_cs,
                                float* grad_r, int64_t gr_bs, int64_t gr_ps, int64_t gr_cs,
                                int deterministic, void* stream) {
  PC3D_REQUIRE(B >= 0 && N >= 1 && M >= 1 && K >= 1, "pc3d_knn_bwd_f32: bad sizes B=%d N=%d M=%d K=%d", B, N, M, K);
  PC3D_REQUIRE(B <= 65535, "pc3d_knn_bwd_f32: B=%d exceeds grid.y limit", B);
  if (B == 0) return PC3D_OK;
  PC3D_REQUIRE(q && r && idx && w, "pc3d_knn_bwd_f32: null input pointer");
  KnnBwdArgs a{{q, q_bs, q_ps, q_cs}, {r, r_bs, r_ps, r_cs}, N, M, K, idx, w,
               {grad_q, gq_bs, gq_ps, gq_cs}, {grad_r, gr_bs, gr_ps, gr_cs}};
  hipStream_t st = as_stream(stream);
  const int nmax = N > M ? N : M;
  hipLaunchKernelGGL(knn_bwd_q_kernel, dim3(cdiv(nmax, 256), B), dim3(256), 0, st, a);
  PC3D_LAUNCH_CHECK("pc3d_knn_bwd_f32/q");
  if (grad_r) {
    if (deterministic)
      hipLaunchKernelGGL(knn_bwd_r_det_kernel, dim3(cdiv(M, 256), B), dim3(256), 0, st, a);
    else
      hipLaunchKernelGGL(knn_bwd_r_atomic_kernel, dim3(cdiv(N * K, 256), B), dim3(256), 0, st, a);
    PC3D_LAUNCH_CHECK("pc3d_knn_bwd_f32/r");
  }
  return PC3D_OK;
}

extern "C" int pc3d_knn_f32(const float* q, int64_t q_bs, int64_t q_ps, int64_t q_cs,
                            const float* r, int64_t r_bs, int64_t r_ps, int64_t r_cs,
                            int B, int N, int M, int K, float* dists, int32_t* idx, void* stream) {
  PC3D_REQUIRE(B >= 0 && N >= 0 && M >= 1, "pc3d_knn_f32: bad sizes B=%d N=%d M=%d", B, N, M);
  PC3D_REQUIRE(K >= 1 && K <= 32, "pc3d_knn_f32: K=%d out of range [1,32]", K);
  PC3D_REQUIRE(K <= M, "pc3d_knn_f32: K=%d exceeds the reference set size M=%d", K, M);
  PC3D_REQUIRE(B <= 65535, "pc3d_knn_f32: B=%d exceeds grid.y limit", B);
  if (B == 0 || N == 0) return PC3D_OK;
  PC3D_REQUIRE(q && r, "pc3d_knn_f32: null input pointer");
  KnnArgs a{{q, q_bs, q_ps, q_cs}, {r, r_bs, r_ps, r_cs}, N, M, K, dists, idx};
  hipStream_t st = as_stream(stream);
  if (K <= 8) return knn_launch<8>(a, B, st);
  if (K <= 16) return knn_launch<16>(a, B, st);
  if (K <= 24) return knn_launch<24>(a, B, st);
  return knn_launch<32>(a, B, st);
}
