// K8 — PointNet per-point MLP 3 -> C1(64) -> C2(128) -> C3 (1x1 conv + folded eval-BN + ReLU) fused with the
// max-pool over points, forward and backward-to-input.  gfx950, fp32-input MFMA (exact fp32 FMA chains).
//
// Replaces: model/pointnet.py:34-37 (STN3d tower) and :110-123 (PointNetfeat trunk): three Conv1d+BN(+ReLU) and
// torch.max over N, which materialise [B,64,N], [B,128,N] and [B,1024,N] activations (134 MB at B=32,N=1024).
//
// Forward: workgroup = (batch b, tile of 128 points), 4 waves. Layers 1-2 are computed once per tile into LDS
// (h2 tile [128 pts][128 ch], 66 KiB); layer 3 runs on v_mfma_f32_32x32x2_f32 with POINTS on the MFMA row index
// and CHANNELS on the column (lane) index, so the max over a tile's points is an in-register reduction
// (64 accumulator values per lane) + one cross-half shuffle; W3 rows stream from L2 as float4 per lane with a
// permuted-k order shared by both operands. Only (max, argmax) per (b, tile, channel) leaves the CU; a tiny second
// kernel folds the tiles. The [B,C3,N] activation is never written.
//
// Backward: max-pool routes each channel's gradient to ONE point, so dgrad of layer 3 is sparse:
// workgroup = (b, tile of 64 points) gathers the channels whose argmax falls in its tile in ascending channel order
// (deterministic, no atomics), recomputes h1/h2 masks for those 64 points and chains W2^T, W1^T on the VALU.
#include "pc3d_common.h"

namespace pc3d {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int PM_C1 = 64;
constexpr int PM_C2 = 128;
constexpr int PM_TP = 128;           // forward: points per workgroup
constexpr int PM_LD1 = PM_C1 + 4;    // LDS row strides (floats): +4 keeps ds_read_b128 conflict-free
constexpr int PM_LD2 = PM_C2 + 4;
constexpr int PM_BTP = 64;           // backward: points per workgroup
constexpr int PM_MAXC3 = 1024;       // backward: widest pooled layer held in LDS

struct PMFwdArgs {
  PtsView x;
  int N, C3, ntiles;
  const float* T;  // [B,3,3] or null: x'[n,:] = x[n,:] @ T  (model/pointnet.py:106-109)
  const float *W1, *b1, *W2, *b2, *W3, *b3;
  float* part_val;    // [B, ntiles, C3]
  int32_t* part_idx;  // [B, ntiles, C3]
};

__device__ __forceinline__ void load_point(const PtsView& x, const float* T, int b, int n, int N, float& px,
                                           float& py, float& pz) {
  px = py = pz = 0.f;
  if (n < N) {
    const float* p = x.p + (int64_t)b * x.bs + (int64_t)n * x.ps;
    const float x0 = p[0], x1 = p[x.cs], x2 = p[2 * x.cs];
    if (T) {
      const float* t = T + (int64_t)b * 9;
      px = __builtin_fmaf(x2, t[6], __builtin_fmaf(x1, t[3], x0 * t[0]));
      py = __builtin_fmaf(x2, t[7], __builtin_fmaf(x1, t[4], x0 * t[1]));
      pz = __builtin_fmaf(x2, t[8], __builtin_fmaf(x1, t[5], x0 * t[2]));
    } else {
      px = x0, py = x1, pz = x2;
    }
  }
}

// h1[p][c] = relu(W1[c,:].x_p + b1[c]) for `npts` points whose coordinates sit in xs[3][npts]; lanes run over c.
template <int NPTS, int NTHREADS>
__device__ __forceinline__ void layer1_to_lds(const float* xs, float* h1, const float* W1, const float* b1) {
  const int c = threadIdx.x & (PM_C1 - 1);
  const int grp = threadIdx.x >> 6;
  constexpr int per = NPTS / (NTHREADS / 64);
  const float w0 = W1[c * 3 + 0], w1 = W1[c * 3 + 1], w2 = W1[c * 3 + 2], bb = b1[c];
#pragma unroll 4
  for (int i = 0; i < per; ++i) {
    const int p = grp * per + i;
    float v = __builtin_fmaf(w2, xs[2 * NPTS + p], __builtin_fmaf(w1, xs[NPTS + p], __builtin_fmaf(w0, xs[p], bb)));
    h1[p * PM_LD1 + c] = fmaxf(v, 0.f);
  }
}

__global__ __launch_bounds__(256) void pointmlp3_max_fwd_kernel(PMFwdArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[PM_TP * PM_LD2 + 3 * PM_TP];  // 69,120 B static
  float* h1 = lds;                       // [128][68]   (dead after layer 2)
  float* h2 = lds;                       // [128][132]  (overwrites h1 behind a barrier)
  float* xs = lds + PM_TP * PM_LD2;      // [3][128]
  const int tile = blockIdx.x, b = blockIdx.y;
  const int n0 = tile * PM_TP;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;

  if (threadIdx.x < PM_TP) {
    float px, py, pz;
    load_point(a.x, a.T, b, n0 + threadIdx.x, a.N, px, py, pz);
    xs[threadIdx.x] = px;
    xs[PM_TP + threadIdx.x] = py;
    xs[2 * PM_TP + threadIdx.x] = pz;
  }
  __syncthreads();
  layer1_to_lds<PM_TP, 256>(xs, h1, a.W1, a.b1);
  __syncthreads();

  // ---- layer 2 on MFMA: D[pt][c2] = sum_k h1[pt][k] W2[c2][k]; wave owns c2 in [32*wave, +32), 4 point tiles
  {
    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    const float* wrow = a.W2 + (32 * wave + r) * PM_C1 + 4 * h;
#pragma unroll
    for (int t = 0; t < PM_C1 / 8; ++t) {
      const float4 bw = *reinterpret_cast<const float4*>(wrow + 8 * t);
      float4 av[4];
#pragma unroll
      for (int tl = 0; tl < 4; ++tl)
        av[tl] = *reinterpret_cast<const float4*>(h1 + (tl * 32 + r) * PM_LD1 + 8 * t + 4 * h);
#pragma unroll
      for (int tl = 0; tl < 4; ++tl) {
        acc[tl] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[tl].x, bw.x, acc[tl], 0, 0, 0);
        acc[tl] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[tl].y, bw.y, acc[tl], 0, 0, 0);
        acc[tl] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[tl].z, bw.z, acc[tl], 0, 0, 0);
        acc[tl] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[tl].w, bw.w, acc[tl], 0, 0, 0);
      }
    }
    __syncthreads();  // every wave is done reading h1
    const float bias = a.b2[32 * wave + r];
#pragma unroll
    for (int tl = 0; tl < 4; ++tl)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int pt = tl * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        h2[pt * PM_LD2 + 32 * wave + r] = fmaxf(acc[tl][e] + bias, 0.f);
      }
  }
  __syncthreads();

  // ---- layer 3 + max over the tile's points
  const int nblk = a.C3 / 32;
  for (int cb = wave; cb < nblk; cb += 4) {
    const int ch = cb * 32 + r;
    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    const float* wrow = a.W3 + (int64_t)ch * PM_C2 + 4 * h;
#pragma unroll
    for (int t = 0; t < PM_C2 / 8; ++t) {
      const float4 bw = *reinterpret_cast<const float4*>(wrow + 8 * t);
      float4 av[4];
#pragma unroll
      for (int tl = 0; tl < 4; ++tl)
        av[tl] = *reinterpret_cast<const float4*>(h2 + (tl * 32 + r) * PM_LD2 + 8 * t + 4 * h);
#pragma unroll
      for (int tl = 0; tl < 4; ++tl) {
        acc[tl] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[tl].x, bw.x, acc[tl], 0, 0, 0);
        acc[tl] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[tl].y, bw.y, acc[tl], 0, 0, 0);
        acc[tl] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[tl].z, bw.z, acc[tl], 0, 0, 0);
        acc[tl] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[tl].w, bw.w, acc[tl], 0, 0, 0);
      }
    }
    float best = -__builtin_inff();
    int bi = n0;
#pragma unroll
    for (int tl = 0; tl < 4; ++tl)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int pt = n0 + tl * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;  // ascending in (tl, e) for fixed h
        const float v = acc[tl][e];
        if (pt < a.N && v > best) {
          best = v;
          bi = pt;
        }
      }
    const float ov = __shfl_xor(best, 32, 64);
    const int oi = __shfl_xor(bi, 32, 64);
    if (ov > best || (ov == best && oi < bi)) {
      best = ov;
      bi = oi;
    }
    if (h == 0) {
      const int64_t o = ((int64_t)b * a.ntiles + tile) * a.C3 + ch;
      a.part_val[o] = best + a.b3[ch];
      a.part_idx[o] = bi;
    }
  }
}

// fold tiles: pooled[b,c] = max_t part[b,t,c] (first tile wins ties => lowest point index), optional ReLU
__global__ __launch_bounds__(256) void pointmlp3_fold_kernel(const float* part_val, const int32_t* part_idx,
                                                             int ntiles, int C3, int relu_last, float* pooled,
                                                             int32_t* argidx) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  if (c >= C3) return;
  const int64_t base = (int64_t)b * ntiles * C3 + c;
  float best = part_val[base];
  int bi = part_idx[base];
  for (int t = 1; t < ntiles; ++t) {
    const float v = part_val[base + (int64_t)t * C3];
    if (v > best) {
      best = v;
      bi = part_idx[base + (int64_t)t * C3];
    }
  }
  if (relu_last) best = fmaxf(best, 0.f);
  pooled[(int64_t)b * C3 + c] = best;
  argidx[(int64_t)b * C3 + c] = bi;
}

// ---------------------------------------------------------------------------------------------------------
struct PMBwdArgs {
  PtsView x;
  int N, C3;
  const float* T;
  const float *W1, *b1, *W2, *b2, *W3;
  const int32_t* argidx;  // [B,C3]
  const float* g;         // [B,C3] upstream gradient on pooled (already masked for relu_last by the caller)
  PtsViewMut gx;          // gradient wrt the (transformed) tower input x'
};

__global__ __launch_bounds__(256) void pointmlp3_max_bwd_kernel(PMBwdArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[PM_BTP * PM_LD2 + 2 * PM_BTP * PM_LD1 + 3 * PM_BTP + 2 * PM_MAXC3];
  float* g2s = lds;                          // [64][132]
  float* h1s = g2s + PM_BTP * PM_LD2;        // [64][68]
  float* g1s = h1s + PM_BTP * PM_LD1;        // [64][68]
  float* xs = g1s + PM_BTP * PM_LD1;         // [3][64]
  float* s_g = xs + 3 * PM_BTP;              // [C3] (C3 <= PM_MAXC3)
  int* s_n = reinterpret_cast<int*>(s_g + PM_MAXC3);  // [C3]
  const int tile = blockIdx.x, b = blockIdx.y;
  const int n0 = tile * PM_BTP;
  const int tid = threadIdx.x;

  int any = 0;
  for (int c = tid; c < a.C3; c += 256) {
    const int n = a.argidx[(int64_t)b * a.C3 + c] - n0;
    const float gv = a.g[(int64_t)b * a.C3 + c];
    const bool in = (n >= 0) && (n < PM_BTP) && (gv != 0.f);
    s_n[c] = in ? n : -1;
    s_g[c] = gv;
    any |= in ? 1 : 0;
  }
  if (tid < PM_BTP) {
    float px, py, pz;
    load_point(a.x, a.T, b, n0 + tid, a.N, px, py, pz);
    xs[tid] = px;
    xs[PM_BTP + tid] = py;
    xs[2 * PM_BTP + tid] = pz;
  }
  for (int i = tid; i < PM_BTP * PM_LD2; i += 256) g2s[i] = 0.f;
  any = __syncthreads_or(any);
  if (!any) {  // no critical point in this tile: gradient is exactly zero
    if (tid < 3 * PM_BTP) {
      const int p = tid & (PM_BTP - 1), c = tid >> 6;
      if (n0 + p < a.N) a.gx.p[(int64_t)b * a.gx.bs + (int64_t)(n0 + p) * a.gx.ps + c * a.gx.cs] = 0.f;
    }
    return;
  }

  // ---- sparse dgrad of layer 3: g2[n][k] = sum_{c: argmax(c)==n} g[c] W3[c][k], ascending c
  {
    const int k = tid & (PM_C2 - 1), ph = tid >> 7;
    for (int c0 = 0; c0 < a.C3; c0 += 4) {
      const int4 nn = *reinterpret_cast<const int4*>(s_n + c0);
      const int ns[4] = {nn.x, nn.y, nn.z, nn.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int n = ns[e];
        if (n >= 0 && (n >> 5) == ph) {  // wave-uniform
          const float w = a.W3[(int64_t)(c0 + e) * PM_C2 + k];
          g2s[n * PM_LD2 + k] = __builtin_fmaf(s_g[c0 + e], w, g2s[n * PM_LD2 + k]);
        }
      }
    }
  }
  layer1_to_lds<PM_BTP, 256>(xs, h1s, a.W1, a.b1);
  __syncthreads();

  // ---- ReLU mask of layer 2 (recomputed): zero g2 where W2 h1 + b2 <= 0
  {
    const int k2 = tid & (PM_C2 - 1), ph = tid >> 7;
    float w[PM_C1];
#pragma unroll
    for (int j = 0; j < PM_C1; j += 4) {
      const float4 v = *reinterpret_cast<const float4*>(a.W2 + k2 * PM_C1 + j);
      w[j] = v.x, w[j + 1] = v.y, w[j + 2] = v.z, w[j + 3] = v.w;
    }
    const float bb = a.b2[k2];
    for (int p = ph * 32; p < ph * 32 + 32; ++p) {
      float pre = bb;
#pragma unroll
      for (int j = 0; j < PM_C1; j += 4) {
        const float4 hv = *reinterpret_cast<const float4*>(h1s + p * PM_LD1 + j);
        pre = __builtin_fmaf(w[j], hv.x, pre);
        pre = __builtin_fmaf(w[j + 1], hv.y, pre);
        pre = __builtin_fmaf(w[j + 2], hv.z, pre);
        pre = __builtin_fmaf(w[j + 3], hv.w, pre);
      }
      if (!(pre > 0.f)) g2s[p * PM_LD2 + k2] = 0.f;
    }
  }
  __syncthreads();

  // ---- g1[p][j] = relu'(h1) * sum_k2 W2[k2][j] g2[p][k2]
  {
    const int j = tid & (PM_C1 - 1), pg = tid >> 6;
    float w[PM_C2];
#pragma unroll
    for (int k2 = 0; k2 < PM_C2; ++k2) w[k2] = a.W2[k2 * PM_C1 + j];
    for (int p = pg * 16; p < pg * 16 + 16; ++p) {
      float s = 0.f;
#pragma unroll
      for (int k2 = 0; k2 < PM_C2; k2 += 4) {
        const float4 gv = *reinterpret_cast<const float4*>(g2s + p * PM_LD2 + k2);
        s = __builtin_fmaf(w[k2], gv.x, s);
        s = __builtin_fmaf(w[k2 + 1], gv.y, s);
        s = __builtin_fmaf(w[k2 + 2], gv.z, s);
        s = __builtin_fmaf(w[k2 + 3], gv.w, s);
      }
      g1s[p * PM_LD1 + j] = (h1s[p * PM_LD1 + j] > 0.f) ? s : 0.f;
    }
  }
  __syncthreads();

  // ---- gx'[p][c] = sum_j W1[j][c] g1[p][j]
  if (tid < 3 * PM_BTP) {
    const int p = tid & (PM_BTP - 1), c = tid >> 6;
    float s = 0.f;
#pragma unroll 8
    for (int j = 0; j < PM_C1; ++j) s = __builtin_fmaf(a.W1[j * 3 + c], g1s[p * PM_LD1 + j], s);
    if (n0 + p < a.N) a.gx.p[(int64_t)b * a.gx.bs + (int64_t)(n0 + p) * a.gx.ps + c * a.gx.cs] = s;
  }
}

}  // namespace pc3d

using namespace pc3d;

extern "C" int pc3d_pointmlp3_tile_points(void) { return PM_TP; }

extern "C" int pc3d_pointmlp3_max_fwd_f32(const float* x, int64_t x_bs, int64_t x_ps, int64_t x_cs, int B, int N,
                                          const float* T, const float* W1, const float* b1, const float* W2,
                                          const float* b2, const float* W3, const float* b3, int C1, int C2,
                                          int C3, int relu_last, float* part_val, int32_t* part_idx,
                                          float* pooled, int32_t* argidx, void* stream) {
  PC3D_REQUIRE(B >= 0 && N >= 1, "pc3d_pointmlp3_max_fwd_f32: bad sizes B=%d N=%d", B, N);
  PC3D_REQUIRE(C1 == PM_C1 && C2 == PM_C2 && C3 >= 32 && C3 % 32 == 0,
               "pc3d_pointmlp3_max_fwd_f32: unsupported widths %d/%d/%d (need 64/128/multiple of 32)", C1, C2, C3);
  PC3D_REQUIRE(B <= 65535, "pc3d_pointmlp3_max_fwd_f32: B=%d exceeds grid.y limit", B);
  if (B == 0) return PC3D_OK;
  PC3D_REQUIRE(x && W1 && b1 && W2 && b2 && W3 && b3 && part_val && part_idx,
               "pc3d_pointmlp3_max_fwd_f32: null pointer");
  PC3D_REQUIRE((pooled == nullptr) == (argidx == nullptr),
               "pc3d_pointmlp3_max_fwd_f32: pooled and argidx must both be given or both be NULL");
  const int ntiles = cdiv(N, PM_TP);
  PMFwdArgs a{{x, x_bs, x_ps, x_cs}, N, C3, ntiles, T, W1, b1, W2, b2, W3, b3, part_val, part_idx};
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(pointmlp3_max_fwd_kernel, dim3(ntiles, B), dim3(256), 0, st, a);
  PC3D_LAUNCH_CHECK("pc3d_pointmlp3_max_fwd_f32");
  if (pooled) {  // NULL: leave the per-tile partials unfolded (a fused consumer, or kernel-only timing)
    hipLaunchKernelGGL(pointmlp3_fold_kernel, dim3(cdiv(C3, 256), B), dim3(256), 0, st, part_val, part_idx, ntiles,
                       C3, relu_last, pooled, argidx);
    PC3D_LAUNCH_CHECK("pc3d_pointmlp3_max_fwd_f32/fold");
  }
  return PC3D_OK;
}

extern "C" int pc3d_pointmlp3_max_bwd_f32(const float* x, int64_t x_bs, int64_t x_ps, int64_t x_cs, int B, int N,
                                          const float* T, const float* W1, const float* b1, const float* W2,
                                          const float* b2, const float* W3, int C1, int C2, int C3,
                                          const int32_t* argidx, const float* g_pooled, float* grad_x,
                                          int64_t gx_bs, int64_t gx_ps, int64_t gx_cs, void* stream) {
  PC3D_REQUIRE(B >= 0 && N >= 1, "pc3d_pointmlp3_max_bwd_f32: bad sizes B=%d N=%d", B, N);
  PC3D_REQUIRE(C1 == PM_C1 && C2 == PM_C2 && C3 >= 32 && C3 % 32 == 0 && C3 <= PM_MAXC3,
               "pc3d_pointmlp3_max_bwd_f32: unsupported widths %d/%d/%d", C1, C2, C3);
  PC3D_REQUIRE(B <= 65535, "pc3d_pointmlp3_max_bwd_f32: B=%d exceeds grid.y limit", B);
  if (B == 0) return PC3D_OK;
  PC3D_REQUIRE(x && W1 && b1 && W2 && b2 && W3 && argidx && g_pooled && grad_x,
               "pc3d_pointmlp3_max_bwd_f32: null pointer");
  PMBwdArgs a{{x, x_bs, x_ps, x_cs}, N, C3, T, W1, b1, W2, b2, W3, argidx, g_pooled, {grad_x, gx_bs, gx_ps, gx_cs}};
  hipLaunchKernelGGL(pointmlp3_max_bwd_kernel, dim3(cdiv(N, PM_BTP), B), dim3(256), 0, as_stream(stream), a);
  PC3D_LAUNCH_CHECK("pc3d_pointmlp3_max_bwd_f32");
  return PC3D_OK;
}
