// K1 — fused nearest-neighbour (min + argmin of squared L2) between two point sets, never materialising
// the [B,N,M] distance matrix.  gfx950 / wave64.
//
// Replaces: attack/CW/CW_utils/distance.py:15-32,40-50,58-70; utils/dis_utils_torch.py:8-28;
//           utils/dis_utils_numpy.py:13-38; attack/GeoA3/knn_utils.py:10-20 (K=1).
//
// Structure: one 256-thread workgroup = (direction, batch b, tile of 64*Q query points).
//   * the reference cloud of batch b is staged through LDS as SoA x[],y[],z[] in tiles of <= 4096 points;
//   * all four waves hold the SAME 64*Q queries in registers (Q per lane) and each scans one quarter of the
//     LDS tile with broadcast ds_read_b128 (4 reference points per read) -> split-M across waves gives
//     4x more waves per query tile, which is what fills 256 CUs at N=1024;
//   * the running arg-min is kept per CHUNK of 8 reference points (6 VALU ops per pair for the distance + 0.5 for
//     v_min3 + 3/8 for the compare/select = 6.9 instead of 9 with a per-pair compare/select); the position inside
//     the winning chunk is recovered after the scan by recomputing its 8 distances;
//   * distance is the direct-difference form (dx*dx + dy*dy + dz*dz with FMA) — the |a|^2+|b|^2-2ab expansion
//     loses 1e-5 relative accuracy on near-coincident clouds;
//   * per-wave (min,argmin) are merged through LDS in ascending reference order so ties resolve to the
//     LOWEST index (torch.min / numpy argmin behaviour).
#include "pc3d_common.h"

namespace pc3d {

struct NNDir {
  PtsView q, r;
  int N, M;
  float* d;
  int32_t* i;
};
struct NNArgs {
  NNDir dir[2];
};

constexpr int kNNThreads = 256;
constexpr int kNNWaves = kNNThreads / kWave;
constexpr int kNNMaxTile = 4096;           // reference points per LDS tile (48 KiB SoA)
constexpr float kFar = 1.0e18f;            // sentinel coordinate: (1e18)^2*3 < FLT_MAX, never the minimum

constexpr int kNNChunk = 8;                // reference points per arg-min bookkeeping step

// the one distance formula of this file (scan and index resolution must agree bit for bit)
__device__ __forceinline__ float nn_dist(float rx, float ry, float rz, float qx, float qy, float qz) {
  const float dx = rx - qx, dy = ry - qy, dz = rz - qz;
  float d = dx * dx;
  d = __builtin_fmaf(dy, dy, d);
  return __builtin_fmaf(dz, dz, d);
}

template <int Q>
__global__ __launch_bounds__(kNNThreads) void nn_kernel(NNArgs args, int mt_cap) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const NNDir& D = args.dir[blockIdx.z];
  const int N = D.N, M = D.M;
  const int q0 = blockIdx.x * (kWave * Q);
  if (q0 >= N) return;  // grid.x is sized for the larger direction
  const int b = blockIdx.y;
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;

  // tile geometry (uniform): mt = points staged per pass, slice = points scanned per wave, multiple of the chunk
  const int mt = M < mt_cap ? M : mt_cap;
  const int slice = ((mt + kNNWaves * kNNChunk - 1) / (kNNWaves * kNNChunk)) * kNNChunk;
  const int mt_pad = slice * kNNWaves;
  float* sx = lds;
  float* sy = lds + mt_pad;
  float* sz = lds + 2 * mt_pad;

  float qx[Q], qy[Q], qz[Q], best[Q];
  int bidx[Q];
  const float* qb = D.q.p + (int64_t)b * D.q.bs;
#pragma unroll
  for (int k = 0; k < Q; ++k) {
    int qi = q0 + k * kWave + lane;
    if (qi >= N) qi = N - 1;  // clamp: duplicates a valid query, result discarded at the store
    const float* qp = qb + (int64_t)qi * D.q.ps;
    qx[k] = qp[0];
    qy[k] = qp[D.q.cs];
    qz[k] = qp[2 * D.q.cs];
    best[k] = __builtin_inff();
    bidx[k] = 0;
  }

  const float* rb = D.r.p + (int64_t)b * D.r.bs;
  for (int m0 = 0; m0 < M; m0 += mt) {
    __syncthreads();  // previous tile fully consumed
    for (int j = threadIdx.x; j < mt_pad; j += kNNThreads) {
      const int m = m0 + j;
      float x = kFar, y = kFar, z = kFar;
      if (j < mt && m < M) {
        const float* rp = rb + (int64_t)m * D.r.ps;
        x = rp[0];
        y = rp[D.r.cs];
        z = rp[2 * D.r.cs];
      }
      sx[j] = x;
      sy[j] = y;
      sz[j] = z;
    }
    __syncthreads();

    const int s0 = wave * slice;
    for (int j = s0; j < s0 + slice; j += kNNChunk) {
      const float4 rx0 = *reinterpret_cast<const float4*>(sx + j), rx1 = *reinterpret_cast<const float4*>(sx + j + 4);
      const float4 ry0 = *reinterpret_cast<const float4*>(sy + j), ry1 = *reinterpret_cast<const float4*>(sy + j + 4);
      const float4 rz0 = *reinterpret_cast<const float4*>(sz + j), rz1 = *reinterpret_cast<const float4*>(sz + j + 4);
      const float rxa[kNNChunk] = {rx0.x, rx0.y, rx0.z, rx0.w, rx1.x, rx1.y, rx1.z, rx1.w};
      const float rya[kNNChunk] = {ry0.x, ry0.y, ry0.z, ry0.w, ry1.x, ry1.y, ry1.z, ry1.w};
      const float rza[kNNChunk] = {rz0.x, rz0.y, rz0.z, rz0.w, rz1.x, rz1.y, rz1.z, rz1.w};
#pragma unroll
      for (int k = 0; k < Q; ++k) {
        float d[kNNChunk];
#pragma unroll
        for (int e = 0; e < kNNChunk; ++e) d[e] = nn_dist(rxa[e], rya[e], rza[e], qx[k], qy[k], qz[k]);
        // chunk minimum with v_min3 (0.5 op / pair); WHICH of the 8 it was is resolved once, after the scan
        float m = __builtin_fminf(__builtin_fminf(d[0], d[1]), d[2]);
        m = __builtin_fminf(__builtin_fminf(m, d[3]), d[4]);
        m = __builtin_fminf(__builtin_fminf(m, d[5]), d[6]);
        m = __builtin_fminf(m, d[7]);
        if (m < best[k]) {     // strict: the EARLIEST chunk holding the minimum wins
          best[k] = m;
          bidx[k] = m0 + j;
        }
      }
    }
  }

  // merge the four waves' candidates (ascending wave = ascending reference index inside a tile; across
  // tiles compare indices explicitly so the lowest index wins ties)
  __syncthreads();
  float* cd = lds;                                        // [kNNWaves][64*Q]
  int* ci = reinterpret_cast<int*>(lds + kNNWaves * kWave * Q);
#pragma unroll
  for (int k = 0; k < Q; ++k) {
    cd[wave * (kWave * Q) + k * kWave + lane] = best[k];
    ci[wave * (kWave * Q) + k * kWave + lane] = bidx[k];
  }
  __syncthreads();
  for (int t = threadIdx.x; t < kWave * Q; t += kNNThreads) {
    float bd = cd[t];
    int bi = ci[t];
#pragma unroll
    for (int w = 1; w < kNNWaves; ++w) {
      const float d = cd[w * (kWave * Q) + t];
      const int i = ci[w * (kWave * Q) + t];
      if (d < bd || (d == bd && i < bi)) {
        bd = d;
        bi = i;
      }
    }
    const int qi = q0 + t;
    if (qi < N) {
      if (D.d) D.d[(int64_t)b * N + qi] = bd;
      if (D.i) {
        // bi is the first index of the winning chunk: the arg-min is the first of its 8 points whose distance,
        // recomputed with the same instructions, equals the minimum (ties -> lowest index, as torch.min)
        const float* qp = qb + (int64_t)qi * D.q.ps;
        const float x = qp[0], y = qp[D.q.cs], z = qp[2 * D.q.cs];
        int arg = bi;
#pragma unroll
        for (int e = kNNChunk - 1; e >= 0; --e) {
          const int m = bi + e;
          if (m < M) {
            const float* rp = rb + (int64_t)m * D.r.ps;
            if (nn_dist(rp[0], rp[D.r.cs], rp[2 * D.r.cs], x, y, z) == bd) arg = m;
          }
        }
        D.i[(int64_t)b * N + qi] = arg;
      }
    }
  }
}

static size_t nn_lds_bytes(int M, int Q, int* mt_cap_out) {
  const int cap = kNNMaxTile;   // 1024 / 2048 / 4096 measured equal at B=32, N=4096
  const int mt = M < cap ? M : cap;
  const int slice = ((mt + kNNWaves * kNNChunk - 1) / (kNNWaves * kNNChunk)) * kNNChunk;
  const size_t tile = (size_t)3 * slice * kNNWaves * sizeof(float);
  const size_t merge = (size_t)kNNWaves * kWave * Q * 8;
  *mt_cap_out = cap;
  return tile > merge ? tile : merge;
}

static int nn_launch(const NNArgs& a, int ndir, int B, hipStream_t st) {
  int maxN = a.dir[0].N, maxM = a.dir[0].M;
  if (ndir == 2) {
    if (a.dir[1].N > maxN) maxN = a.dir[1].N;
    if (a.dir[1].M > maxM) maxM = a.dir[1].M;
  }
  // Queries per lane: enough ILP to cover the LDS broadcast reads, but keep >= ~4 waves per SIMD's worth of
  // workgroups on 256 CUs (grid = tiles x B x ndir, 4 waves each).
  const long q_total = (long)maxN * B * ndir;
  int Q = 4;
  if (q_total / (kWave * 4) < 1024) Q = 2;
  if (q_total / (kWave * 2) < 1024) Q = 1;
  int mt_cap;
  const size_t lds = nn_lds_bytes(maxM, Q, &mt_cap);
  dim3 grid(cdiv(maxN, kWave * Q), B, ndir), block(kNNThreads);
  switch (Q) {
    case 1: hipLaunchKernelGGL(nn_kernel<1>, grid, block, lds, st, a, mt_cap); break;
    case 2: hipLaunchKernelGGL(nn_kernel<2>, grid, block, lds, st, a, mt_cap); break;
    default: hipLaunchKernelGGL(nn_kernel<4>, grid, block, lds, st, a, mt_cap); break;
  }
  PC3D_LAUNCH_CHECK("pc3d_nn");
  return PC3D_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Row reduction [B,N] -> [B]: one workgroup per row, fixed-order tree => bitwise reproducible.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rowreduce_kernel(const float* x, int N, int op, int pre, float* out) {
  __shared__ float part[4];
  const float* row = x + (int64_t)blockIdx.x * N;
  float acc = (op == 1) ? -__builtin_inff() : 0.f;
  for (int i = threadIdx.x; i < N; i += 256) {
    float v = row[i];
    if (pre == 1) v = __builtin_sqrtf(fmaxf(v, 0.f));
    acc = (op == 1) ? fmaxf(acc, v) : acc + v;
  }
  acc = (op == 1) ? wave_max(acc) : wave_sum(acc);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float r = part[0];
    for (int w = 1; w < 4; ++w) r = (op == 1) ? fmaxf(r, part[w]) : r + part[w];
    if (op == 0) r /= (float)N;
    out[blockIdx.x] = r;
  }
}

// ---------------------------------------------------------------------------------------------------------
// Backward through gathered NN pairs.
//   dA[b,i] = |a_i - b_{iA[i]}|^2 with upstream weight wA[b,i]   (scaled by sA)
//   dB[b,j] = |b_j - a_{iB[j]}|^2 with upstream weight wB[b,j]   (scaled by sB)
//   grad_a[i] = 2 sA wA[i] (a_i - b_{iA[i]})  +  sum_{j: iB[j]==i} 2 sB wB[j] (a_i - b_j)
//   grad_b[j] = 2 sB wB[j] (b_j - a_{iB[j]})  +  sum_{i: iA[i]==j} 2 sA wA[i] (b_j - a_i)
// Pass 1 writes the "own" term densely (overwrite); pass 2 adds the scattered term with float atomics, or —
// deterministic mode — every destination point scans the opposite index list in ascending order.
// ---------------------------------------------------------------------------------------------------------
struct BwdSide {
  PtsView self, other;   // self = the set whose points own the NN index list `idx` (own term)
  int n_self, n_other;
  const int32_t* idx;    // [B,n_self] -> index into other
  const float* w;        // upstream weight, element strides (w_bs, w_ps); may be NULL (=> zero)
  int64_t w_bs, w_ps;
  float scale;
  PtsViewMut g_self;     // gradient wrt self (may be null)
  PtsViewMut g_other;    // gradient wrt other (may be null)
};
struct BwdArgs {
  BwdSide s[2];
};

// own term: grid (ceil(n/256), B, 2)
__global__ __launch_bounds__(256) void nn_bwd_own_kernel(BwdArgs args) {
  const BwdSide& S = args.s[blockIdx.z];
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  if (i >= S.n_self || S.g_self.p == nullptr) return;
  float gx = 0.f, gy = 0.f, gz = 0.f;
  if (S.w != nullptr && S.idx != nullptr) {
    const float w = 2.f * S.scale * S.w[(int64_t)b * S.w_bs + (int64_t)i * S.w_ps];
    const int j = S.idx[(int64_t)b * S.n_self + i];
    const float* p = S.self.p + (int64_t)b * S.self.bs + (int64_t)i * S.self.ps;
    const float* o = S.other.p + (int64_t)b * S.other.bs + (int64_t)j * S.other.ps;
    gx = w * (p[0] - o[0]);
    gy = w * (p[S.self.cs] - o[S.other.cs]);
    gz = w * (p[2 * S.self.cs] - o[2 * S.other.cs]);
  }
  float* g = S.g_self.p + (int64_t)b * S.g_self.bs + (int64_t)i * S.g_self.ps;
  g[0] = gx;
  g[S.g_self.cs] = gy;
  g[2 * S.g_self.cs] = gz;
}

// scattered term, atomic flavour: thread per (b, i in self) adds into g_other[idx]
__global__ __launch_bounds__(256) void nn_bwd_scatter_atomic_kernel(BwdArgs args) {
  const BwdSide& S = args.s[blockIdx.z];
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  if (i >= S.n_self || S.g_other.p == nullptr || S.w == nullptr || S.idx == nullptr) return;
  const float w = 2.f * S.scale * S.w[(int64_t)b * S.w_bs + (int64_t)i * S.w_ps];
  if (w == 0.f) return;
  const int j = S.idx[(int64_t)b * S.n_self + i];
  const float* p = S.self.p + (int64_t)b * S.self.bs + (int64_t)i * S.self.ps;
  const float* o = S.other.p + (int64_t)b * S.other.bs + (int64_t)j * S.other.ps;
  float* g = S.g_other.p + (int64_t)b * S.g_other.bs + (int64_t)j * S.g_other.ps;
  atomicAdd(g, w * (o[0] - p[0]));
  atomicAdd(g + S.g_other.cs, w * (o[S.other.cs] - p[S.self.cs]));
  atomicAdd(g + 2 * S.g_other.cs, w * (o[2 * S.other.cs] - p[2 * S.self.cs]));
}

// scattered term, deterministic flavour: one workgroup per (b, 256 destinations); the index list of the
// opposite side is streamed through LDS and every destination accumulates its matches in ascending order.
__global__ __launch_bounds__(256) void nn_bwd_scatter_det_kernel(BwdArgs args) {
  __shared__ int s_idx[1024];
  __shared__ float s_w[1024];
  const BwdSide& S = args.s[blockIdx.z];
  const int b = blockIdx.y;
  const int j = blockIdx.x * 256 + threadIdx.x;  // destination in `other`
  if (blockIdx.x * 256 >= S.n_other) return;
  if (S.g_other.p == nullptr || S.w == nullptr || S.idx == nullptr) return;
  const bool live = j < S.n_other;
  float ox = 0.f, oy = 0.f, oz = 0.f;
  if (live) {
    const float* o = S.other.p + (int64_t)b * S.other.bs + (int64_t)j * S.other.ps;
    ox = o[0];
    oy = o[S.other.cs];
    oz = o[2 * S.other.cs];
  }
  float gx = 0.f, gy = 0.f, gz = 0.f;
  for (int i0 = 0; i0 < S.n_self; i0 += 1024) {
    __syncthreads();
    for (int t = threadIdx.x; t < 1024; t += 256) {
      const int i = i0 + t;
      s_idx[t] = (i < S.n_self) ? S.idx[(int64_t)b * S.n_self + i] : -1;
      s_w[t] = (i < S.n_self) ? 2.f * S.scale * S.w[(int64_t)b * S.w_bs + (int64_t)i * S.w_ps] : 0.f;
    }
    __syncthreads();
    const int lim = (S.n_self - i0) < 1024 ? (S.n_self - i0) : 1024;
    for (int t = 0; t < lim; ++t) {
      if (s_idx[t] == j) {
        const float w = s_w[t];
        const float* p = S.self.p + (int64_t)b * S.self.bs + (int64_t)(i0 + t) * S.self.ps;
        gx += w * (ox - p[0]);
        gy += w * (oy - p[S.self.cs]);
        gz += w * (oz - p[2 * S.self.cs]);
      }
    }
  }
  if (live) {
    float* g = S.g_other.p + (int64_t)b * S.g_other.bs + (int64_t)j * S.g_other.ps;
    g[0] += gx;
    g[S.g_other.cs] += gy;
    g[2 * S.g_other.cs] += gz;
  }
}

}  // namespace pc3d

using namespace pc3d;

extern "C" int pc3d_nn_f32(const float* q, int64_t q_bs, int64_t q_ps, int64_t q_cs,
                           const float* r, int64_t r_bs, int64_t r_ps, int64_t r_cs,
                           int B, int N, int M, float* min_d2, int32_t* idx, void* stream) {
  PC3D_REQUIRE(B >= 0 && N >= 0 && M >= 1, "pc3d_nn_f32: bad sizes B=%d N=%d M=%d (M must be >= 1)", B, N, M);
  PC3D_REQUIRE(B <= 65535, "pc3d_nn_f32: B=%d exceeds grid.y limit 65535", B);
  if (B == 0 || N == 0) return PC3D_OK;
  PC3D_REQUIRE(q && r, "pc3d_nn_f32: null input pointer");
  NNArgs a{};
  a.dir[0] = NNDir{{q, q_bs, q_ps, q_cs}, {r, r_bs, r_ps, r_cs}, N, M, min_d2, idx};
  return nn_launch(a, 1, B, as_stream(stream));
}

extern "C" int pc3d_nn_bidir_f32(const float* a_, int64_t a_bs, int64_t a_ps, int64_t a_cs,
                                 const float* b_, int64_t b_bs, int64_t b_ps, int64_t b_cs,
                                 int B, int N, int M,
                                 float* dA, int32_t* iA, float* dB, int32_t* iB, void* stream) {
  PC3D_REQUIRE(B >= 0 && N >= 1 && M >= 1, "pc3d_nn_bidir_f32: bad sizes B=%d N=%d M=%d", B, N, M);
  PC3D_REQUIRE(B <= 65535, "pc3d_nn_bidir_f32: B=%d exceeds grid.y limit 65535", B);
  if (B == 0) return PC3D_OK;
  PC3D_REQUIRE(a_ && b_, "pc3d_nn_bidir_f32: null input pointer");
  NNArgs a{};
  a.dir[0] = NNDir{{a_, a_bs, a_ps, a_cs}, {b_, b_bs, b_ps, b_cs}, N, M, dA, iA};
  a.dir[1] = NNDir{{b_, b_bs, b_ps, b_cs}, {a_, a_bs, a_ps, a_cs}, M, N, dB, iB};
  return nn_launch(a, 2, B, as_stream(stream));
}

extern "C" int pc3d_rowreduce_f32(const float* x, int B, int N, int op, int pre, float* out, void* stream) {
  PC3D_REQUIRE(B >= 0 && N >= 1, "pc3d_rowreduce_f32: bad sizes B=%d N=%d", B, N);
  PC3D_REQUIRE(op >= 0 && op <= 2 && (pre == 0 || pre == 1), "pc3d_rowreduce_f32: bad op=%d pre=%d", op, pre);
  if (B == 0) return PC3D_OK;
  PC3D_REQUIRE(x && out, "pc3d_rowreduce_f32: null pointer");
  hipLaunchKernelGGL(rowreduce_kernel, dim3(B), dim3(256), 0, as_stream(stream), x, N, op, pre, out);
  PC3D_LAUNCH_CHECK("pc3d_rowreduce_f32");
  return PC3D_OK;
}

extern "C" int pc3d_nn_bwd_f32(const float* a_, int64_t a_bs, int64_t a_ps, int64_t a_cs,
                               const float* b_, int64_t b_bs, int64_t b_ps, int64_t b_cs,
                               int B, int N, int M,
                               const int32_t* iA, const float* wA, int64_t wA_bs, int64_t wA_ps, float sA,
                               const int32_t* iB, const float* wB, int64_t wB_bs, int64_t wB_ps, float sB,
                               float* grad_a, int64_t ga_bs, int64_t ga_ps, int64_t ga_cs,
                               float* grad_b, int64_t gb_bs, int64_t gb_ps, int64_t gb_cs,
                               int deterministic, void* stream) {
  PC3D_REQUIRE(B >= 0 && N >= 1 && M >= 1, "pc3d_nn_bwd_f32: bad sizes B=%d N=%d M=%d", B, N, M);
  PC3D_REQUIRE(B <= 65535, "pc3d_nn_bwd_f32: B=%d exceeds grid.y limit 65535", B);
  if (B == 0) return PC3D_OK;
  PC3D_REQUIRE(a_ && b_, "pc3d_nn_bwd_f32: null input pointer");
  PC3D_REQUIRE((wA == nullptr) || (iA != nullptr), "pc3d_nn_bwd_f32: wA given without iA");
  PC3D_REQUIRE((wB == nullptr) || (iB != nullptr), "pc3d_nn_bwd_f32: wB given without iB");
  BwdArgs g{};
  PtsView A{a_, a_bs, a_ps, a_cs}, Bv{b_, b_bs, b_ps, b_cs};
  PtsViewMut GA{grad_a, ga_bs, ga_ps, ga_cs}, GB{grad_b, gb_bs, gb_ps, gb_cs};
  g.s[0] = BwdSide{A, Bv, N, M, iA, wA, wA_bs, wA_ps, sA, GA, GB};
  g.s[1] = BwdSide{Bv, A, M, N, iB, wB, wB_bs, wB_ps, sB, GB, GA};
  hipStream_t st = as_stream(stream);
  const int nmax = N > M ? N : M;
  hipLaunchKernelGGL(nn_bwd_own_kernel, dim3(cdiv(nmax, 256), B, 2), dim3(256), 0, st, g);
  PC3D_LAUNCH_CHECK("pc3d_nn_bwd_f32/own");
  const bool scatter_a = grad_a && wB, scatter_b = grad_b && wA;
  if (scatter_a || scatter_b) {
    if (deterministic)
      hipLaunchKernelGGL(nn_bwd_scatter_det_kernel, dim3(cdiv(nmax, 256), B, 2), dim3(256), 0, st, g);
    else
      hipLaunchKernelGGL(nn_bwd_scatter_atomic_kernel, dim3(cdiv(nmax, 256), B, 2), dim3(256), 0, st, g);
    PC3D_LAUNCH_CHECK("pc3d_nn_bwd_f32/scatter");
  }
  return PC3D_OK;
}
