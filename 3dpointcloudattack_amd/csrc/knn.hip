// K2/K4 — K nearest reference points of every query (xyz, squared L2, direct-difference form), sorted ascending,
// without the [B,N,M] matrix + topk of the reference:
//   attack/CW/CW_utils/dist_utils.py:133-144 (KNNDist, k+1=6), attack/GeoA3/knn_utils.py:10-55 (knn_points,
//   K in {1,4,17}), attack/AOF/TAOF_attack.py:13-28 (k=30), model/dgcnn.py:194-200 on xyz (k=20),
//   model/curvenet_util.py:10-17.
//
// The reference cloud is staged through LDS as SoA (like the NN kernel); each wave serves 4 queries whose sorted
// K-lists live across its lanes (details at knn_wave_kernel). K <= 64. Ties: the lower reference index comes first.
#include "pc3d_common.h"
#include "knn_list.h"

namespace pc3d {

constexpr float kKnnFar = 1.0e18f;

struct KnnArgs {
  PtsView q, r;
  int N, M, K;
  float* d;    // [B,N,K]
  int32_t* i;  // [B,N,K]
  int32_t* i_noself = nullptr;   // [B,N,K-1] or null: columns 1 .. K-1 of i
  int32_t* i_first = nullptr;    // [B,N,k2] or null: columns 0 .. k2-1 of i
  int k2 = 0;
  const int32_t* hint = nullptr; // [B,N,K] or null: K reference indices per query from an earlier, similar search (may be `i`)
};

// ---------------------------------------------------------------------------------------------------------
// Wave-per-query search: the 64 lanes of a wave hold 64 CANDIDATES of one LDS step and test them against 4 queries
// at once (wave-uniform thresholds), so a step costs ~10 VALU wave-instructions per query like the NN kernel, and a
// candidate that beats a query's K-th distance is inserted into that query's sorted list, which lives ACROSS the
// lanes (lane j = j-th nearest so far): ballot/popcount finds the slot, one DPP lane shift makes room.
// (A list per lane — one query per lane — makes all 64 lanes pay for an insertion whenever ANY lane accepts, which
// is almost every candidate: measured 3.3 ms at B=32, N=M=4096, K=21 against 0.56 ms for this layout.)
// The first 64 candidates seed the list (bitonic sort, or K rounds of wave-min for small K). Ties: candidates are
// visited in ascending index order and inserted AFTER equal distances, so the lower index comes first
// (torch.topk / the reference's order).
// ---------------------------------------------------------------------------------------------------------
constexpr int kKwQPW = 4;          // queries scanned together by a wave
constexpr int kKwTile = 2048;      // reference points per LDS tile (24 KiB SoA)

template <int kKwWaves, bool KGE2>    // waves per workgroup: a workgroup serves 4 * kKwWaves queries from one staged copy of the cloud
__global__ __launch_bounds__(kKwWaves * 64) void knn_wave_kernel(KnnArgs a) {
  constexpr int kKwPasses = 1;
  constexpr int kKwThreads = kKwWaves * 64;
  __shared__ __attribute__((aligned(16))) float lds[3 * kKwTile];
  const int N = a.N, M = a.M, K = a.K;
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // scalar: query loads / thresholds live in SGPRs
  const int qbase = blockIdx.x * (kKwWaves * kKwPasses * kKwQPW) + wave * (kKwPasses * kKwQPW);
  float* sx = lds;
  float* sy = lds + kKwTile;
  float* sz = lds + 2 * kKwTile;

  // per-query state, statically indexed: [pass][u]
  // The lists hold distance KEYS: a squared distance is >= +0 or NaN, so its bit pattern with the sign cleared is already
  // an order-preserving integer (NaN above +inf: never below a threshold) — thresholds then live in SGPRs and the
  // re-check of a candidate is a scalar compare (knn_list.h).
  float qx[kKwPasses][kKwQPW], qy[kKwPasses][kKwQPW], qz[kKwPasses][kKwQPW];
  int lk[kKwPasses][kKwQPW], thr[kKwPasses][kKwQPW], li[kKwPasses][kKwQPW];
#pragma unroll
  for (int p = 0; p < kKwPasses; ++p)
#pragma unroll
    for (int u = 0; u < kKwQPW; ++u) {
      int qi = qbase + p * kKwQPW + u;
      if (qi >= N) qi = N - 1;
      const float* qp = a.q.p + (int64_t)b * a.q.bs + (int64_t)qi * a.q.ps;   // wave-uniform address
      qx[p][u] = qp[0], qy[p][u] = qp[a.q.cs], qz[p][u] = qp[2 * a.q.cs];
      lk[p][u] = kKnnInfKey;
      li[p][u] = 0x7fffffff;
      thr[p][u] = kKnnInfKey;
    }

  const float* rb = a.r.p + (int64_t)b * a.r.bs;
  // ---- hint: K reference indices per query from an earlier search of (nearly) the same clouds — an attack moves its points
  // by 1e-2 per iteration, so last iteration's neighbours bound this iteration's K-th distance tightly. ANY K distinct
  // valid indices give an upper bound (their largest distance >= the K-th smallest of all), so the hint can only cost time,
  // never change the result: the list starts EMPTY, candidates at or below the bound are inserted as they come (ascending
  // index, after equal keys: the same tie order), everything else is rejected by one compare. Without a hint the expected
  // number of insertions per query is K ln(M / K) (106 at M = 4096, K = 20: three quarters of the kernel's instructions);
  // with one it is K plus the few points that moved inside the bound. The wave checks the hint (in range, pairwise
  // distinct, finite bound) for its four queries and falls back to the seeded scan if any of them fails.
  int cap[kKwPasses][kKwQPW];
  bool hinted = a.hint != nullptr;
#pragma unroll
  for (int p = 0; p < kKwPasses; ++p)
#pragma unroll
    for (int u = 0; u < kKwQPW; ++u) cap[p][u] = 0x7fffffff;
  if (hinted) {
#pragma unroll
    for (int p = 0; p < kKwPasses; ++p)
#pragma unroll
      for (int u = 0; u < kKwQPW; ++u) {
        int qi = qbase + p * kKwQPW + u;
        if (qi >= N) qi = N - 1;
        const int h = lane < K ? a.hint[((int64_t)b * N + qi) * K + lane] : -1 - lane;     // (distinct fillers above K)
        bool bad = lane < K && (unsigned)h >= (unsigned)M;
        for (int t = 0; t < K; ++t) bad |= lane > t && h == __builtin_amdgcn_readlane(h, t);
        int key = 0;
        if (lane < K && !bad) {
          const float* rp = rb + (int64_t)h * a.r.ps;
          const float dx = rp[0] - qx[p][u], dy = rp[a.r.cs] - qy[p][u], dz = rp[2 * a.r.cs] - qz[p][u];
          float t = dx * dx;
          t = __builtin_fmaf(dy, dy, t);
          t = __builtin_fmaf(dz, dz, t);
          key = __builtin_bit_cast(int, t) & 0x7fffffff;      // the scan's arithmetic, bit for bit
        }
        const int hm = wave_max_i32(key);                     // (a NaN distance has a key above +inf's)
        if (__builtin_amdgcn_ballot_w64(bad) != 0 || hm >= kKnnInfKey) hinted = false;
        cap[p][u] = hm + 1;
      }
    if (hinted) {
#pragma unroll
      for (int p = 0; p < kKwPasses; ++p)
#pragma unroll
        for (int u = 0; u < kKwQPW; ++u) thr[p][u] = cap[p][u];
    } else {
#pragma unroll
      for (int p = 0; p < kKwPasses; ++p)
#pragma unroll
        for (int u = 0; u < kKwQPW; ++u) cap[p][u] = 0x7fffffff;
    }
  }
  for (int m0 = 0; m0 < M; m0 += kKwTile) {
    const int mt = (M - m0) < kKwTile ? (M - m0) : kKwTile;
    __syncthreads();
    const int mt_pad = (mt + 63) & ~63;             // pad the last step with far sentinels (never the K nearest: K <= M)
    for (int j = threadIdx.x; j < mt_pad; j += kKwThreads) {
      float x = kKnnFar, y = kKnnFar, z = kKnnFar;
      if (j < mt) {
        const float* rp = rb + (int64_t)(m0 + j) * a.r.ps;
        x = rp[0], y = rp[a.r.cs], z = rp[2 * a.r.cs];
      }
      sx[j] = x, sy[j] = y, sz[j] = z;
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < kKwPasses; ++p) {
      if (qbase + p * kKwQPW >= N) break;          // wave-uniform: no query left in this pass
      for (int j0 = 0; j0 < mt; j0 += 64) {
        const int j = j0 + lane;
        const float cx = sx[j], cy = sy[j], cz = sz[j];
        float d[kKwQPW];
#pragma unroll
        for (int u = 0; u < kKwQPW; ++u) {
          const float dx = cx - qx[p][u], dy = cy - qy[p][u], dz = cz - qz[p][u];
          float t = dx * dx;
          t = __builtin_fmaf(dy, dy, t);
          t = __builtin_fmaf(dz, dz, t);
          d[u] = t;
        }
        if (m0 == 0 && j0 == 0 && !hinted) {
          // seed: the K nearest of the first 64 candidates — K rounds of wave-min selection (ties: lowest lane) for small K
          float rem[kKwQPW], m[kKwQPW];
#pragma unroll
          for (int u = 0; u < kKwQPW; ++u) rem[u] = (d[u] == d[u]) ? d[u] : __builtin_inff();   // NaN never a neighbour
          if (K > 10) {                             // larger K: one bitonic sort per query is cheaper than K rounds
#pragma unroll
            for (int u = 0; u < kKwQPW; ++u) {
              int sk = __builtin_bit_cast(int, rem[u]);
              int si = lane;
              wave_sort_pairs_dpp(sk, si, lane);   // no LDS round trips (knn_list.h)
              lk[p][u] = sk, li[p][u] = si;
              thr[p][u] = __builtin_amdgcn_readlane(sk, K - 1);
            }
            continue;
          }
          for (int t = 0; t < K; ++t) {
#pragma unroll
            for (int u = 0; u < kKwQPW; ++u) {      // four independent chains hide the DPP / readlane latencies
              m[u] = wave_min_dpp(rem[u]);
              const int c = __builtin_ctzll(__builtin_amdgcn_ballot_w64(rem[u] == m[u]) | (1ull << 63));
              lk[p][u] = (lane == t) ? __builtin_bit_cast(int, m[u]) : lk[p][u];
              li[p][u] = (lane == t) ? c : li[p][u];
              rem[u] = (lane == c) ? __builtin_inff() : rem[u];
            }
          }
#pragma unroll
          for (int u = 0; u < kKwQPW; ++u) thr[p][u] = __builtin_bit_cast(int, m[u]);
          continue;
        }
#pragma unroll
        for (int u = 0; u < kKwQPW; ++u)
          knn_scan_insert<KGE2>(lk[p][u], li[p][u], thr[p][u], __builtin_bit_cast(int, d[u]) & 0x7fffffff, m0 + j0, K, cap[p][u]);
      }
    }
  }
#pragma unroll
  for (int p = 0; p < kKwPasses; ++p)
#pragma unroll
    for (int u = 0; u < kKwQPW; ++u) {
      const int qi = qbase + p * kKwQPW + u;
      if (qi < N && lane < K) {
        if (a.d) a.d[((int64_t)b * N + qi) * K + lane] = __builtin_bit_cast(float, lk[p][u]);
        if (a.i) a.i[((int64_t)b * N + qi) * K + lane] = li[p][u];
        if (a.i_noself && lane >= 1) a.i_noself[((int64_t)b * N + qi) * (K - 1) + lane - 1] = li[p][u];
        if (a.i_first && lane < a.k2) a.i_first[((int64_t)b * N + qi) * a.k2 + lane] = li[p][u];
      }
    }
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
  const float* wscale = nullptr;   // [B] or null: every w[b,...] is multiplied by wscale[b] (an upstream per-sample gradient)
  int self_sum = 0;                // 1: q and r are the SAME cloud and gr == gq: the scattered term is added to the dense one
};

__global__ __launch_bounds__(256) void knn_bwd_q_kernel(KnnBwdArgs a) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  const int nmax = a.N > a.M ? a.N : a.M;
  if (i >= nmax) return;
  if (a.gr.p && i < a.M && !a.self_sum) {  // zero-fill grad_r (the scatter pass accumulates into it)
    float* g = a.gr.p + (int64_t)b * a.gr.bs + (int64_t)i * a.gr.ps;
    g[0] = 0.f, g[a.gr.cs] = 0.f, g[2 * a.gr.cs] = 0.f;
  }
  if (!a.gq.p || i >= a.N) return;
  const float* qp = a.q.p + (int64_t)b * a.q.bs + (int64_t)i * a.q.ps;
  const float qx = qp[0], qy = qp[a.q.cs], qz = qp[2 * a.q.cs];
  float gx = 0.f, gy = 0.f, gz = 0.f;
  const int64_t base = ((int64_t)b * a.N + i) * a.K;
  const float ws = a.wscale ? 2.f * a.wscale[b] : 2.f;
  for (int k = 0; k < a.K; ++k) {
    const float w = ws * a.w[base + k];
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
  const float w = (a.wscale ? 2.f * a.wscale[b] : 2.f) * a.w[e];
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
  __shared__ __attribute__((aligned(16))) int s_idx[1024];
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
      s_w[t] = (e < total) ? (a.wscale ? 2.f * a.wscale[b] : 2.f) * a.w[(int64_t)b * total + e] : 0.f;
    }
    __syncthreads();
    const int lim = (total - t0) < 1024 ? (total - t0) : 1024;
    // eight list entries per trip (two broadcast ds_read_b128; entries past lim hold -1): see nn_bwd_scatter_det_kernel
    for (int t = 0; t < lim; t += 8) {
      const int4 ia = *reinterpret_cast<const int4*>(s_idx + t), ib = *reinterpret_cast<const int4*>(s_idx + t + 4);
      const int ii[8] = {ia.x, ia.y, ia.z, ia.w, ib.x, ib.y, ib.z, ib.w};
      if (!live || (ia.x != j && ia.y != j && ia.z != j && ia.w != j && ib.x != j && ib.y != j && ib.z != j && ib.w != j)) continue;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        if (ii[e] != j) continue;
        const int i = (t0 + t + e) / a.K;
        const float* qp = a.q.p + (int64_t)b * a.q.bs + (int64_t)i * a.q.ps;
        const float w = s_w[t + e];
        gx += w * (rx - qp[0]);
        gy += w * (ry - qp[a.q.cs]);
        gz += w * (rz - qp[2 * a.q.cs]);
      }
    }
  }
  if (live) {
    float* g = a.gr.p + (int64_t)b * a.gr.bs + (int64_t)j * a.gr.ps;
    if (a.self_sum) g[0] += gx, g[a.gr.cs] += gy, g[2 * a.gr.cs] += gz;
    else g[0] = gx, g[a.gr.cs] = gy, g[2 * a.gr.cs] = gz;
  }
}

// deterministic form for long lists: every edge's contribution to grad_r as a record val[b, e, 0:3] = 2 w (r_j - q_i);
// the ordered LDS scatter of det.hip then sums the records of every reference point in edge order
__global__ __launch_bounds__(256) void knn_bwd_r_edges_kernel(KnnBwdArgs a, float* __restrict__ val) {
  const int t = blockIdx.x * 256 + threadIdx.x;  // over N*K
  const int b = blockIdx.y;
  if (t >= a.N * a.K) return;
  const int i = t / a.K;
  const int64_t e = (int64_t)b * a.N * a.K + t;
  const float w = (a.wscale ? 2.f * a.wscale[b] : 2.f) * a.w[e];
  const int j = min(max(a.idx[e], 0), a.M - 1);
  const float* qp = a.q.p + (int64_t)b * a.q.bs + (int64_t)i * a.q.ps;
  const float* rp = a.r.p + (int64_t)b * a.r.bs + (int64_t)j * a.r.ps;
  float* v = val + e * 3;
  v[0] = w * (rp[0] - qp[0]);
  v[1] = w * (rp[a.r.cs] - qp[a.q.cs]);
  v[2] = w * (rp[2 * a.r.cs] - qp[2 * a.q.cs]);
}

// ---------------------------------------------------------------------------------------------------------
// The kNN-distance outlier penalty of the kNN attack (attack/CW/CW_utils/dist_utils.py:112-160 KNNDist.forward) from the
// K1 = k + 1 sorted self-kNN distances d [B,N,K1] (entry 0 = the point itself):
//   value_i = mean_{j=1..k} d[i,j];  thr = mean_i value + alpha * std_i value (unbiased, as torch.std);
//   loss[b] = mean_i value_i [value_i > thr]            (the mask is a constant of the graph: torch.no_grad, :146-151)
// and the weights the backward hands to the kNN backward: w[b,i,j] = [value_i > thr] / (N k) for j >= 1, 0 for j = 0.
// One workgroup per sample, fixed-order reductions (deterministic); replaces ~15 ATen launches forward and as many backward.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_sum256(float v, float* part) {   // all 256 threads call it; result in every thread
  v = wave_sum(v);
  __syncthreads();                       // part may still be read from the previous call
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
  __syncthreads();
  return (part[0] + part[1]) + (part[2] + part[3]);
}

__global__ __launch_bounds__(256) void knn_outlier_loss_kernel(const float* __restrict__ d, int N, int K1, float alpha,
                                                               float* __restrict__ loss, float* __restrict__ w) {
  __shared__ float part[4];
  const int b = blockIdx.x, k = K1 - 1;
  const float* db = d + (int64_t)b * N * K1;
  const float inv_k = 1.f / (float)k;
  float s = 0.f;
  for (int i = threadIdx.x; i < N; i += 256) {
    float v = 0.f;
    for (int j = 1; j < K1; ++j) v += db[(int64_t)i * K1 + j];
    s += v * inv_k;
  }
  const float mean = block_sum256(s, part) / (float)N;
  float q = 0.f;
  for (int i = threadIdx.x; i < N; i += 256) {
    float v = 0.f;
    for (int j = 1; j < K1; ++j) v += db[(int64_t)i * K1 + j];
    const float dv = v * inv_k - mean;
    q += dv * dv;
  }
  const float var = block_sum256(q, part) / (float)(N > 1 ? N - 1 : 1);
  const float thr = mean + alpha * sqrtf(var);
  const float wv = 1.f / ((float)N * (float)k);
  float l = 0.f;
  float* wb = w + (int64_t)b * N * K1;
  for (int i = threadIdx.x; i < N; i += 256) {
    float v = 0.f;
    for (int j = 1; j < K1; ++j) v += db[(int64_t)i * K1 + j];
    v *= inv_k;
    const bool out = v > thr;
    if (out) l += v;
    wb[(int64_t)i * K1] = 0.f;
    for (int j = 1; j < K1; ++j) wb[(int64_t)i * K1 + j] = out ? wv : 0.f;
  }
  l = block_sum256(l, part);
  if (threadIdx.x == 0) loss[b] = l / (float)N;
}

static int knn_bwd_launch(const char* nm, KnnBwdArgs a, int B, float* grad_r, int64_t gr_bs, int64_t gr_ps, int64_t gr_cs,
                          int deterministic, float* det_ws, void* stream) {
  const int N = a.N, M = a.M, K = a.K;
  hipStream_t st = as_stream(stream);
  const int nmax = N > M ? N : M;
  hipLaunchKernelGGL(knn_bwd_q_kernel, dim3(cdiv(nmax, 256), B), dim3(256), 0, st, a);
  PC3D_LAUNCH_CHECK(nm);
  if (grad_r) {
    if (deterministic && det_ws && (int64_t)N * K <= 0x7fffffffLL && own_fits(M)) {
      // records + ordered scatter (det_ws: B * N * K * 3 floats); the scan below costs N * K list entries per
      // reference point: 575 us at B=64, N=M=2048, K=6 (the kNN attack's regulariser) against ~20 us here
      hipLaunchKernelGGL(knn_bwd_r_edges_kernel, dim3(cdiv(N * K, 256), B), dim3(256), 0, st, a, det_ws);
      PC3D_LAUNCH_CHECK(nm);
      return scatter_rows_det(nm, a.idx, det_ws, 3, nullptr, 0, 0.f, B, N * K, M, 3, grad_r, gr_ps, a.self_sum, 1, stream, nullptr,
                              gr_bs, gr_cs);
    }
    if (deterministic)
      hipLaunchKernelGGL(knn_bwd_r_det_kernel, dim3(cdiv(M, 256), B), dim3(256), 0, st, a);
    else
      hipLaunchKernelGGL(knn_bwd_r_atomic_kernel, dim3(cdiv(N * K, 256), B), dim3(256), 0, st, a);
    PC3D_LAUNCH_CHECK(nm);
  }
  return PC3D_OK;
}

}  // namespace pc3d

using namespace pc3d;

extern "C" int pc3d_knn_outlier_loss_f32(const float* d, int B, int N, int K1, float alpha, float* loss, float* w, void* stream) {
  PC3D_REQUIRE(B >= 0 && N >= 1 && K1 >= 2, "pc3d_knn_outlier_loss_f32: bad sizes B=%d N=%d K1=%d (self + >= 1 neighbour)", B, N, K1);
  if (B == 0) return PC3D_OK;
  PC3D_REQUIRE(d && loss && w, "pc3d_knn_outlier_loss_f32: null pointer");
  hipLaunchKernelGGL(knn_outlier_loss_kernel, dim3(B), dim3(256), 0, as_stream(stream), d, N, K1, alpha, loss, w);
  PC3D_LAUNCH_CHECK("pc3d_knn_outlier_loss_f32");
  return PC3D_OK;
}

extern "C" int pc3d_knn_self_bwd_f32(const float* x, int64_t x_bs, int64_t x_ps, int64_t x_cs, int B, int N, int K,
                                     const int32_t* idx, const float* w, const float* w_scale, float* grad, int64_t g_bs,
                                     int64_t g_ps, int64_t g_cs, int deterministic, float* det_ws, void* stream) {
  PC3D_REQUIRE(B >= 0 && N >= 1 && K >= 1, "pc3d_knn_self_bwd_f32: bad sizes B=%d N=%d K=%d", B, N, K);
  PC3D_REQUIRE(B <= 65535, "pc3d_knn_self_bwd_f32: B=%d exceeds grid.y limit", B);
  if (B == 0) return PC3D_OK;
  PC3D_REQUIRE(x && idx && w && grad, "pc3d_knn_self_bwd_f32: null pointer");
  KnnBwdArgs a{{x, x_bs, x_ps, x_cs}, {x, x_bs, x_ps, x_cs}, N, N, K, idx, w,
               {grad, g_bs, g_ps, g_cs}, {grad, g_bs, g_ps, g_cs}, w_scale, 1};
  return knn_bwd_launch("pc3d_knn_self_bwd_f32", a, B, grad, g_bs, g_ps, g_cs, deterministic, det_ws, stream);
}

extern "C" int pc3d_knn_bwd_f32(const float* q, int64_t q_bs, int64_t q_ps, int64_t q_cs,
                                const float* r, int64_t r_bs, int64_t r_ps, int64_t r_cs,
                                int B, int N, int M, int K, const int32_t* idx, const float* w,
                                float* grad_q, int64_t gq_bs, int64_t gq_ps, int64_t gq_cs,
                                float* grad_r, int64_t gr_bs, int64_t gr_ps, int64_t gr_cs,
                                int deterministic, float* det_ws, void* stream) {
  PC3D_REQUIRE(B >= 0 && N >= 1 && M >= 1 && K >= 1, "pc3d_knn_bwd_f32: bad sizes B=%d N=%d M=%d K=%d", B, N, M, K);
  PC3D_REQUIRE(B <= 65535, "pc3d_knn_bwd_f32: B=%d exceeds grid.y limit", B);
  if (B == 0) return PC3D_OK;
  PC3D_REQUIRE(q && r && idx && w, "pc3d_knn_bwd_f32: null input pointer");
  KnnBwdArgs a{{q, q_bs, q_ps, q_cs}, {r, r_bs, r_ps, r_cs}, N, M, K, idx, w,
               {grad_q, gq_bs, gq_ps, gq_cs}, {grad_r, gr_bs, gr_ps, gr_cs}};
  return knn_bwd_launch("pc3d_knn_bwd_f32", a, B, grad_r, gr_bs, gr_ps, gr_cs, deterministic, det_ws, stream);
}

static int knn_graph_launch(const char* nm, const float* pts, int64_t p_bs, int64_t p_ps, int64_t p_cs, int B, int N, int K,
                            int32_t* idx, int32_t* idx_noself, int32_t* idx_first, int k2, const int32_t* hint, void* stream) {
  PC3D_REQUIRE(B >= 0 && N >= 1 && K >= 2 && K <= 64 && K <= N && k2 >= 0 && k2 <= K, "%s: bad sizes B=%d N=%d K=%d k2=%d", nm, B, N, K, k2);
  PC3D_REQUIRE(B <= 65535, "%s: B=%d exceeds grid.y limit", nm, B);
  if (B == 0) return PC3D_OK;
  PC3D_REQUIRE(pts && idx, "%s: null pointer", nm);
  KnnArgs a{{pts, p_bs, p_ps, p_cs}, {pts, p_bs, p_ps, p_cs}, N, N, K, nullptr, idx};
  a.i_noself = idx_noself, a.i_first = k2 > 0 ? idx_first : nullptr, a.k2 = k2, a.hint = hint;
  hipLaunchKernelGGL((knn_wave_kernel<4, true>), dim3(cdiv(N, 16), B), dim3(256), 0, as_stream(stream), a);
  PC3D_LAUNCH_CHECK(nm);
  return PC3D_OK;
}

extern "C" int pc3d_knn_graph_i32(const float* pts, int64_t p_bs, int64_t p_ps, int64_t p_cs, int B, int N, int K, int32_t* idx,
                                  int32_t* idx_noself, int32_t* idx_first, int k2, void* stream) {
  return knn_graph_launch("pc3d_knn_graph_i32", pts, p_bs, p_ps, p_cs, B, N, K, idx, idx_noself, idx_first, k2, nullptr, stream);
}

extern "C" int pc3d_knn_graph_hint_i32(const float* pts, int64_t p_bs, int64_t p_ps, int64_t p_cs, int B, int N, int K, int32_t* idx,
                                       int32_t* idx_noself, int32_t* idx_first, int k2, const int32_t* hint, void* stream) {
  return knn_graph_launch("pc3d_knn_graph_hint_i32", pts, p_bs, p_ps, p_cs, B, N, K, idx, idx_noself, idx_first, k2, hint, stream);
}

static int knn_launch(const char* nm, const float* q, int64_t q_bs, int64_t q_ps, int64_t q_cs,
                      const float* r, int64_t r_bs, int64_t r_ps, int64_t r_cs,
                      int B, int N, int M, int K, float* dists, int32_t* idx, const int32_t* hint, void* stream) {
  PC3D_REQUIRE(B >= 0 && N >= 0 && M >= 1, "%s: bad sizes B=%d N=%d M=%d", nm, B, N, M);
  PC3D_REQUIRE(K >= 1 && K <= 64, "%s: K=%d out of range [1,64]", nm, K);
  PC3D_REQUIRE(K <= M, "%s: K=%d exceeds the reference set size M=%d", nm, K, M);
  PC3D_REQUIRE(B <= 65535, "%s: B=%d exceeds grid.y limit", nm, B);
  if (B == 0 || N == 0) return PC3D_OK;
  PC3D_REQUIRE(q && r, "%s: null input pointer", nm);
  KnnArgs a{{q, q_bs, q_ps, q_cs}, {r, r_bs, r_ps, r_cs}, N, M, K, dists, idx};
  a.hint = hint;
  hipStream_t st = as_stream(stream);
  // 4 waves (16 queries) per workgroup: measured best of {2,4,8,16} — larger workgroups wait at the staging barriers
  if (K >= 2) hipLaunchKernelGGL((knn_wave_kernel<4, true>), dim3(cdiv(N, 16), B), dim3(256), 0, st, a);
  else hipLaunchKernelGGL((knn_wave_kernel<4, false>), dim3(cdiv(N, 16), B), dim3(256), 0, st, a);
  PC3D_LAUNCH_CHECK(nm);
  return PC3D_OK;
}

extern "C" int pc3d_knn_f32(const float* q, int64_t q_bs, int64_t q_ps, int64_t q_cs,
                            const float* r, int64_t r_bs, int64_t r_ps, int64_t r_cs,
                            int B, int N, int M, int K, float* dists, int32_t* idx, void* stream) {
  return knn_launch("pc3d_knn_f32", q, q_bs, q_ps, q_cs, r, r_bs, r_ps, r_cs, B, N, M, K, dists, idx, nullptr, stream);
}

extern "C" int pc3d_knn_hint_f32(const float* q, int64_t q_bs, int64_t q_ps, int64_t q_cs,
                                 const float* r, int64_t r_bs, int64_t r_ps, int64_t r_cs,
                                 int B, int N, int M, int K, float* dists, int32_t* idx, const int32_t* hint, void* stream) {
  return knn_launch("pc3d_knn_hint_f32", q, q_bs, q_ps, q_cs, r, r_bs, r_ps, r_cs, B, N, M, K, dists, idx, hint, stream);
}
