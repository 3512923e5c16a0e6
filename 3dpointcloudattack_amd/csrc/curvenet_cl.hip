// K19 — the channels-last glue of a CurveNet CIC block (model/curvenet_util.py:302-376, :379-466), gfx950.
//
// The reference keeps every tensor channels-first and strings each block together from 1x1 convolutions, BatchNorms,
// element-wise adds / multiplies, softmaxes, two batched GEMMs and a top-k. With the walk (K16), the aggregation's
// per-cloud half (K18), the edge kernels (K17) and the dense layers (gemm_nt) already channels-last kernels of this
// library, what was left between them ran as a few hundred ATen / rocBLAS launches per forward+backward. This file
// holds those pieces, so that a block is a chain of this library's launches on [B,N,C] rows:
//   gate            G = y > 0 ? g : slope g                    (backward of "activation after a residual sum")
//   att_scale       att = sigmoid(x . w), xs = x att            (CurveGrouping's self-attention score, :452-455)
//   topk_desc       the curve_num best-scored start points      (:457, order fixed to descending — DESIGN.md A-15)
//   curve_attn      leaky(x + softmax(x K_inter) V_inter + softmax(x K_intra) V_intra)   (CurveAggregation, :425-437,
//                   per-point half; keys / values come from pc3d_curve_agg_kv_f32)
//   lpfa_prep       A = x + G1 p, Bc = G2 p + t - x             (LPFA's geometry term in "A_j + B_i" form, :204-236)
// All tensors fp32, channels-last, C % 4 == 0.
#include "pc3d_common.h"

namespace pc3d {

// ---------------------------------------------------------------------------------------------------------
// gate
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gate_kernel(const float* __restrict__ g, const float* __restrict__ y, float slope,
                                                   float* __restrict__ out, int64_t n) {
  const int64_t i4 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i4 + 3 < n) {
    const float4 gv = *reinterpret_cast<const float4*>(g + i4), yv = *reinterpret_cast<const float4*>(y + i4);
    *reinterpret_cast<float4*>(out + i4) = make_float4(yv.x > 0.f ? gv.x : slope * gv.x, yv.y > 0.f ? gv.y : slope * gv.y,
                                                       yv.z > 0.f ? gv.z : slope * gv.z, yv.w > 0.f ? gv.w : slope * gv.w);
  } else {
    for (int64_t i = i4; i < n; ++i) out[i] = y[i] > 0.f ? g[i] : slope * g[i];
  }
}

// ---------------------------------------------------------------------------------------------------------
// att_scale: one thread per point (rows are 64-256 bytes; neighbouring threads own neighbouring rows)
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void att_scale_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            int64_t M, int C, float* __restrict__ xs,
                                                            float* __restrict__ att) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= M) return;
  const float4* xr = reinterpret_cast<const float4*>(x + p * C);
  const float4* wr = reinterpret_cast<const float4*>(w);
  float s = 0.f;
  for (int c = 0; c < C / 4; ++c) {
    const float4 v = xr[c], u = wr[c];
    s += v.x * u.x;
    s += v.y * u.y;
    s += v.z * u.z;
    s += v.w * u.w;
  }
  const float a = 1.f / (1.f + expf(-s));
  att[p] = a;
  float4* o = reinterpret_cast<float4*>(xs + p * C);
  for (int c = 0; c < C / 4; ++c) {
    const float4 v = xr[c];
    o[c] = make_float4(v.x * a, v.y * a, v.z * a, v.w * a);
  }
}

// gx = g att + (g . x) att (1 - att) w
__global__ __launch_bounds__(256) void att_scale_bwd_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                            const float* __restrict__ att, const float* __restrict__ w,
                                                            int64_t M, int C, float* __restrict__ gx) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= M) return;
  const float4* gr = reinterpret_cast<const float4*>(g + p * C);
  const float4* xr = reinterpret_cast<const float4*>(x + p * C);
  const float4* wr = reinterpret_cast<const float4*>(w);
  float d = 0.f;
  for (int c = 0; c < C / 4; ++c) {
    const float4 v = xr[c], u = gr[c];
    d += v.x * u.x + v.y * u.y + v.z * u.z + v.w * u.w;
  }
  const float a = att[p];
  const float k = d * a * (1.f - a);
  float4* o = reinterpret_cast<float4*>(gx + p * C);
  for (int c = 0; c < C / 4; ++c) {
    const float4 u = gr[c], ww = wr[c];
    o[c] = make_float4(u.x * a + k * ww.x, u.y * a + k * ww.y, u.z * a + k * ww.z, u.w * a + k * ww.w);
  }
}

// ---------------------------------------------------------------------------------------------------------
// topk_desc: one workgroup per cloud sorts (score descending, index ascending) keys in LDS (bitonic network)
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned ordered_bits(float f) {   // monotone float -> unsigned (NaN above +inf, as torch ranks it)
  const unsigned b = __builtin_bit_cast(unsigned, f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

__global__ __launch_bounds__(1024) void topk_desc_kernel(const float* __restrict__ score, int N, int K, int npow2,
                                                         int32_t* __restrict__ idx) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long tk_keys[];
  const int b = blockIdx.x;
  for (int i = threadIdx.x; i < npow2; i += 1024)
    tk_keys[i] = i < N ? (((unsigned long long)(~ordered_bits(score[(int64_t)b * N + i])) << 32) | (unsigned)i) : ~0ull;
  __syncthreads();
  // A stage with partner distance j <= 64 keeps every pair of a wavefront's 64 consecutive i inside one aligned block of
  // 128 keys, and so do the stages after it down to j = 1: those need no workgroup barrier, only the wavefront's own
  // LDS order. A barrier follows a stage only when it or the NEXT stage reaches across blocks (j > 64): 9 barriers instead of
  // 55 for 1024 keys (18.4 -> 16.3 us: the launch is one workgroup per cloud, and what is left is 55 dependent LDS
  // round trips).
  for (int k = 2; k <= npow2; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < npow2 / 2; i += 1024) {
        const int lo = ((i / j) * 2 * j) + (i % j), hi = lo + j;
        const bool asc = (lo & k) == 0;
        const unsigned long long a = tk_keys[lo], c = tk_keys[hi];
        if ((a > c) == asc) tk_keys[lo] = c, tk_keys[hi] = a;
      }
      const int jn = j > 1 ? (j >> 1) : k;      // partner distance of the next stage (k: first stage of size 2k)
      // Workgroup barrier when THIS stage wrote into other wavefronts' blocks (j > 64) or the NEXT one reads from them
      // (jn > 64). Until round 4 only the second condition was tested: after the j = 128 stage of every merge of 256 or
      // more keys, the j = 64 stage read keys a neighbouring wavefront might still be writing. With the 16 wavefronts in
      // step it never showed; beside a single high-priority wavefront that delays one SIMD (fps_kernel<16,64> on the
      // geometry branch of a replayed CurveNet graph) one cloud in ~800 got other start points.
      if (jn > 64 || j > 64) __syncthreads();
      else wave_lds_sync();
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < K; i += 1024) idx[(int64_t)b * K + i] = (int32_t)(tk_keys[i] & 0xffffffffu);
}

// ---------------------------------------------------------------------------------------------------------
// lpfa_prep
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lpfa_prep_fwd_kernel(const float* __restrict__ x, const float* __restrict__ pts,
                                                            const float* __restrict__ G1, const float* __restrict__ G2,
                                                            const float* __restrict__ t, int64_t M, int C,
                                                            float* __restrict__ A, float* __restrict__ Bc) {
  const int C4 = C / 4;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= M * C4) return;
  const int c4 = (int)(i % C4);
  const int64_t p = i / C4;
  const float px = pts[p * 3], py = pts[p * 3 + 1], pz = pts[p * 3 + 2];
  const float4 xv = *reinterpret_cast<const float4*>(x + p * C + 4 * c4);
  const float xe[4] = {xv.x, xv.y, xv.z, xv.w};
  float a[4], bb[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int c = 4 * c4 + e;
    a[e] = xe[e] + (G1[c * 3] * px + G1[c * 3 + 1] * py + G1[c * 3 + 2] * pz);
    bb[e] = (G2[c * 3] * px + G2[c * 3 + 1] * py + G2[c * 3 + 2] * pz + t[c]) - xe[e];
  }
  *reinterpret_cast<float4*>(A + p * C + 4 * c4) = make_float4(a[0], a[1], a[2], a[3]);
  *reinterpret_cast<float4*>(Bc + p * C + 4 * c4) = make_float4(bb[0], bb[1], bb[2], bb[3]);
}

// gx = gA - gBc;  gpts = G1^T gA + G2^T gBc   (one thread per point)
// LP lanes per point (a power of two <= 32, chosen so that LP float4 cover a row when C <= 128): rows are read and
// written coalesced, the three sums of a point are reduced across its lanes. (A thread per point read its row with a
// stride of C floats between lanes and left the deep levels at 8 workgroups: 7 - 20 us a launch, eight launches per
// CurveNet backward.)
template <int LP>
__global__ __launch_bounds__(256) void lpfa_prep_bwd_kernel(const float* __restrict__ gA, const float* __restrict__ gBc,
                                                            const float* __restrict__ G1, const float* __restrict__ G2,
                                                            int64_t M, int C, float* __restrict__ gx,
                                                            float* __restrict__ gpts) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t p = gid / LP;
  const int l = (int)(gid % LP);
  const bool live = p < M;                 // dead lanes stay for the shuffles below
  float s[3] = {0.f, 0.f, 0.f};
  if (live) {
    for (int c4 = l; c4 < C / 4; c4 += LP) {
      const float4 a = *reinterpret_cast<const float4*>(gA + p * C + 4 * c4);
      const float4 b = *reinterpret_cast<const float4*>(gBc + p * C + 4 * c4);
      *reinterpret_cast<float4*>(gx + p * C + 4 * c4) = make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w);
      const float ae[4] = {a.x, a.y, a.z, a.w}, be[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = 4 * c4 + e;
#pragma unroll
        for (int d = 0; d < 3; ++d) s[d] += G1[c * 3 + d] * ae[e] + G2[c * 3 + d] * be[e];
      }
    }
  }
#pragma unroll
  for (int o = LP / 2; o > 0; o >>= 1) {
#pragma unroll
    for (int d = 0; d < 3; ++d) s[d] += __shfl_xor(s[d], o, 64);
  }
  if (live && l == 0) gpts[p * 3] = s[0], gpts[p * 3 + 1] = s[1], gpts[p * 3 + 2] = s[2];
}

// ---------------------------------------------------------------------------------------------------------
// curve_attn: FOUR lanes per point (a quad), each owning every fourth key — with B*N ~ 32 k points a thread per
// point would leave the chip at one wave per SIMD or less, and the kernel is a chain of dependent FMAs. The cloud's
// keys (transposed to [R][C+4]) and values [R][C+4] sit in LDS (the +4 keeps the four rows a quad reads in different
// banks); scores are recomputed per pass instead of being stored (R = cn + cl ~ 105 per point): 3 C R FMAs per point
// forward. Quad reductions are two DPP quad_perm steps.
// ---------------------------------------------------------------------------------------------------------
struct CurveAttnArgs {
  const float* x;    // [B,N,C]
  const float* Kp;   // [B,C,R]
  const float* Vp;   // [B,R,C]
  int N, cn, R;
  float slope;
  float* out;        // [B,N,C]
  // backward
  const float* gout; // [B,N,C]
  float* gx;         // [B,N,C]
  float* dS;         // [B,N,R]  d(loss)/d(score)
  float* Wt;         // [B,N,R]  softmax weights
  float* G;          // [B,N,C]  gated upstream gradient
};

constexpr int CA_PPW = 64;   // points per workgroup (256 threads, 4 lanes per point)

__device__ __forceinline__ float quad_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));  // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));  // quad_perm [2,3,0,1]
  return v;
}
__device__ __forceinline__ float quad_max(float v) {
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false)));
  return v;
}

template <int C>
__device__ __forceinline__ float ca_dot(const float (&v)[C], const float* __restrict__ row) {
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < C; c += 4) {
    const float4 k = *reinterpret_cast<const float4*>(row + c);
    s += v[c] * k.x;
    s += v[c + 1] * k.y;
    s += v[c + 2] * k.z;
    s += v[c + 3] * k.w;
  }
  return s;
}

template <int C>
__device__ __forceinline__ void ca_stage(const CurveAttnArgs& a, int b, float* KT, float* V) {
  constexpr int CP = C + 4;
  const int R = a.R;
  for (int e = threadIdx.x; e < R * C; e += 256) {
    const int j = e / C, c = e - j * C;
    KT[j * CP + c] = a.Kp[((int64_t)b * C + c) * R + j];
    V[j * CP + c] = a.Vp[(int64_t)b * R * C + e];
  }
  __syncthreads();
}

template <int C>
__global__ __launch_bounds__(256) void curve_attn_fwd_kernel(CurveAttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float ca_lds[];
  constexpr int CP = C + 4;
  float* KT = ca_lds;
  float* V = ca_lds + a.R * CP;
  const int b = blockIdx.y;
  ca_stage<C>(a, b, KT, V);
  const int q = threadIdx.x & 3;
  const int p = blockIdx.x * CA_PPW + (threadIdx.x >> 2);
  const bool valid = p < a.N;
  const float* xr = a.x + ((int64_t)b * a.N + (valid ? p : a.N - 1)) * C;
  float x[C];
#pragma unroll
  for (int c = 0; c < C; c += 4) {
    const float4 v = *reinterpret_cast<const float4*>(xr + c);
    x[c] = v.x, x[c + 1] = v.y, x[c + 2] = v.z, x[c + 3] = v.w;
  }
  float m[2] = {-INFINITY, -INFINITY};
  for (int j = q; j < a.R; j += 4) {
    const float s = ca_dot<C>(x, KT + j * CP);
    if (j < a.cn) m[0] = fmaxf(m[0], s);
    else m[1] = fmaxf(m[1], s);
  }
  m[0] = quad_max(m[0]), m[1] = quad_max(m[1]);
  float o[C];
#pragma unroll
  for (int c = 0; c < C; ++c) o[c] = x[c];
#pragma unroll
  for (int seg = 0; seg < 2; ++seg) {
    const int j0 = seg ? a.cn : 0, j1 = seg ? a.R : a.cn;
    if (j1 <= j0) continue;
    float den = 0.f;
    float acc[C];
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = 0.f;
    for (int j = j0 + q; j < j1; j += 4) {
      const float e = expf(ca_dot<C>(x, KT + j * CP) - m[seg]);
      den += e;
#pragma unroll
      for (int c = 0; c < C; c += 4) {
        const float4 v = *reinterpret_cast<const float4*>(V + j * CP + c);
        acc[c] += e * v.x, acc[c + 1] += e * v.y, acc[c + 2] += e * v.z, acc[c + 3] += e * v.w;
      }
    }
    const float inv = 1.f / quad_sum(den);
#pragma unroll
    for (int c = 0; c < C; ++c) o[c] += quad_sum(acc[c]) * inv;
  }
  if (!valid) return;
  float* orow = a.out + ((int64_t)b * a.N + p) * C;
#pragma unroll
  for (int c = 0; c < C; c += 4)
    if (((c >> 2) & 3) == q)     // the quad shares the row's float4 stores
      *reinterpret_cast<float4*>(orow + c) =
          make_float4(o[c] > 0.f ? o[c] : a.slope * o[c], o[c + 1] > 0.f ? o[c + 1] : a.slope * o[c + 1],
                      o[c + 2] > 0.f ? o[c + 2] : a.slope * o[c + 2], o[c + 3] > 0.f ? o[c + 3] : a.slope * o[c + 3]);
}

// backward, per-point half: gx and, for the per-cloud reductions of the second kernel, dS / softmax weights / the gated
// gradient.  With G = leaky'(out) gout, w = softmax per segment, dw_j = G . V_j, t = sum_seg w dw:
//   dS_j = w_j (dw_j - t_seg),   gx = G + sum_j dS_j K[:, j]
template <int C>
__global__ __launch_bounds__(256) void curve_attn_bwd_point_kernel(CurveAttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float ca_lds[];
  constexpr int CP = C + 4;
  float* KT = ca_lds;
  float* V = ca_lds + a.R * CP;
  const int b = blockIdx.y;
  ca_stage<C>(a, b, KT, V);
  const int q = threadIdx.x & 3;
  const int p = blockIdx.x * CA_PPW + (threadIdx.x >> 2);
  const bool valid = p < a.N;
  const int64_t row = (int64_t)b * a.N + (valid ? p : a.N - 1);
  float x[C], G[C];
#pragma unroll
  for (int c = 0; c < C; c += 4) {
    const float4 v = *reinterpret_cast<const float4*>(a.x + row * C + c);
    const float4 g = *reinterpret_cast<const float4*>(a.gout + row * C + c);
    const float4 o = *reinterpret_cast<const float4*>(a.out + row * C + c);
    x[c] = v.x, x[c + 1] = v.y, x[c + 2] = v.z, x[c + 3] = v.w;
    G[c] = o.x > 0.f ? g.x : a.slope * g.x, G[c + 1] = o.y > 0.f ? g.y : a.slope * g.y;
    G[c + 2] = o.z > 0.f ? g.z : a.slope * g.z, G[c + 3] = o.w > 0.f ? g.w : a.slope * g.w;
  }
  float m[2] = {-INFINITY, -INFINITY};
  for (int j = q; j < a.R; j += 4) {
    const float s = ca_dot<C>(x, KT + j * CP);
    if (j < a.cn) m[0] = fmaxf(m[0], s);
    else m[1] = fmaxf(m[1], s);
  }
  m[0] = quad_max(m[0]), m[1] = quad_max(m[1]);
  float den[2] = {0.f, 0.f}, u[2] = {0.f, 0.f};
  for (int j = q; j < a.R; j += 4) {
    const int seg = j < a.cn ? 0 : 1;
    const float e = expf(ca_dot<C>(x, KT + j * CP) - m[seg]);
    den[seg] += e;
    u[seg] += e * ca_dot<C>(G, V + j * CP);
  }
  float inv[2], t[2];
#pragma unroll
  for (int seg = 0; seg < 2; ++seg) {
    const float d = quad_sum(den[seg]);
    inv[seg] = d > 0.f ? 1.f / d : 0.f;
    t[seg] = quad_sum(u[seg]) * inv[seg];
  }
  float dx[C];
#pragma unroll
  for (int c = 0; c < C; ++c) dx[c] = 0.f;
  float* dsr = a.dS + row * a.R;
  float* wr = a.Wt + row * a.R;
  for (int j = q; j < a.R; j += 4) {
    const int seg = j < a.cn ? 0 : 1;
    const float w = expf(ca_dot<C>(x, KT + j * CP) - m[seg]) * inv[seg];
    const float ds = w * (ca_dot<C>(G, V + j * CP) - t[seg]);
#pragma unroll
    for (int c = 0; c < C; c += 4) {
      const float4 k = *reinterpret_cast<const float4*>(KT + j * CP + c);
      dx[c] += ds * k.x, dx[c + 1] += ds * k.y, dx[c + 2] += ds * k.z, dx[c + 3] += ds * k.w;
    }
    if (valid) dsr[j] = ds, wr[j] = w;
  }
#pragma unroll
  for (int c = 0; c < C; ++c) dx[c] = G[c] + quad_sum(dx[c]);
  if (!valid) return;
#pragma unroll
  for (int c = 0; c < C; c += 4)
    if (((c >> 2) & 3) == q) {
      *reinterpret_cast<float4*>(a.gx + row * C + c) = make_float4(dx[c], dx[c + 1], dx[c + 2], dx[c + 3]);
      *reinterpret_cast<float4*>(a.G + row * C + c) = make_float4(G[c], G[c + 1], G[c + 2], G[c + 3]);
    }
}

// backward, per-cloud half: gKp[c][j] = sum_p x[p][c] dS[p][j],  gVp[j][c] = sum_p w[p][j] G[p][c]  over the points of
// one slice of a cloud; tiles of 32 points through LDS, each thread accumulates 4 x 4 blocks of both products. The
// slices' partial sums are plain stores into part[b][slice][2 C R] (no float atomics: deterministic), folded by
// curve_attn_fold_kernel.
constexpr int CA_PT = 32;   // points per LDS tile

template <int C>
__global__ __launch_bounds__(256) void curve_attn_bwd_cloud_kernel(CurveAttnArgs a, float* __restrict__ part,
                                                                   int per_split) {
  extern __shared__ __attribute__((aligned(16))) float ca_lds[];
  const int R = a.R, R4 = (R + 3) / 4, Rp = 4 * R4;
  float* Xt = ca_lds;               // [CA_PT][C]
  float* Gt = Xt + CA_PT * C;       // [CA_PT][C]
  float* St = Gt + CA_PT * C;       // [CA_PT][Rp]
  float* Wt = St + CA_PT * Rp;      // [CA_PT][Rp]
  const int b = blockIdx.y;
  const int p0 = blockIdx.x * per_split, p1 = min(p0 + per_split, a.N);
  constexpr int CG = C / 4;
  const int items = CG * R4;        // 4 x 4 blocks (channel group, score group)
  float accK[2][16], accV[2][16];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int e = 0; e < 16; ++e) accK[u][e] = 0.f, accV[u][e] = 0.f;
  for (int q0 = p0; q0 < p1; q0 += CA_PT) {
    const int np = min(CA_PT, p1 - q0);
    __syncthreads();
    for (int e = threadIdx.x; e < CA_PT * (C / 4); e += 256) {      // rows of x / G are contiguous: float4 copies
      const int q = e / (C / 4), c4 = e - q * (C / 4);
      const int64_t src = ((int64_t)b * a.N + q0 + q) * C + 4 * c4;
      const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
      *reinterpret_cast<float4*>(Xt + q * C + 4 * c4) = q < np ? *reinterpret_cast<const float4*>(a.x + src) : z;
      *reinterpret_cast<float4*>(Gt + q * C + 4 * c4) = q < np ? *reinterpret_cast<const float4*>(a.G + src) : z;
    }
    for (int e = threadIdx.x; e < CA_PT * Rp; e += 256) {
      const int q = e / Rp, j = e - q * Rp;
      const bool ok = q < np && j < R;
      const int64_t src = ((int64_t)b * a.N + q0 + q) * R + j;
      St[e] = ok ? a.dS[src] : 0.f;
      Wt[e] = ok ? a.Wt[src] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int it = threadIdx.x + 256 * u;
      if (it >= items) continue;
      const int cg = it % CG, jg = it / CG;
      for (int q = 0; q < CA_PT; ++q) {
        const float4 xv = *reinterpret_cast<const float4*>(Xt + q * C + 4 * cg);
        const float4 gv = *reinterpret_cast<const float4*>(Gt + q * C + 4 * cg);
        const float4 sv = *reinterpret_cast<const float4*>(St + q * Rp + 4 * jg);
        const float4 wv = *reinterpret_cast<const float4*>(Wt + q * Rp + 4 * jg);
        const float xe[4] = {xv.x, xv.y, xv.z, xv.w}, ge[4] = {gv.x, gv.y, gv.z, gv.w};
        const float se[4] = {sv.x, sv.y, sv.z, sv.w}, we[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
        for (int ci = 0; ci < 4; ++ci)
#pragma unroll
          for (int ji = 0; ji < 4; ++ji) {
            accK[u][ci * 4 + ji] += xe[ci] * se[ji];
            accV[u][ji * 4 + ci] += we[ji] * ge[ci];
          }
      }
    }
  }
  float* pk = part + ((int64_t)b * gridDim.x + blockIdx.x) * 2 * C * R;   // [C][R] then [R][C]
  float* pv = pk + C * R;
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int it = threadIdx.x + 256 * u;
    if (it >= items) continue;
    const int cg = it % CG, jg = it / CG;
#pragma unroll
    for (int ci = 0; ci < 4; ++ci)
#pragma unroll
      for (int ji = 0; ji < 4; ++ji) {
        const int c = 4 * cg + ci, j = 4 * jg + ji;
        if (j < R) {
          pk[c * R + j] = accK[u][ci * 4 + ji];
          pv[j * C + c] = accV[u][ji * 4 + ci];
        }
      }
  }
}

// gKp[b] | gVp[b] = sum over the slices (fixed order) of part[b][slice]
__global__ __launch_bounds__(256) void curve_attn_fold_kernel(const float* __restrict__ part, int nsplit, int CR,
                                                              float* __restrict__ gKp, float* __restrict__ gVp) {
  const int b = blockIdx.y;
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= 2 * CR) return;
  const float* src = part + (int64_t)b * nsplit * 2 * CR + e;
  float s = 0.f;
  for (int k = 0; k < nsplit; ++k) s += src[(int64_t)k * 2 * CR];
  if (e < CR) gKp[(int64_t)b * CR + e] = s;
  else gVp[(int64_t)b * CR + e - CR] = s;
}

}  // namespace pc3d

using namespace pc3d;

extern "C" int pc3d_gate_f32(const float* g, const float* y, int64_t n, float slope, float* out, void* stream) {
  PC3D_REQUIRE(n >= 0, "pc3d_gate_f32: bad size");
  if (n == 0) return PC3D_OK;
  PC3D_REQUIRE(g && y && out, "pc3d_gate_f32: null pointer");
  PC3D_REQUIRE(((reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(out)) & 15) == 0,
               "pc3d_gate_f32: buffers must be 16-byte aligned");
  const int64_t blocks = (n + 1023) / 1024;
  PC3D_REQUIRE(blocks <= 0x7fffffffLL, "pc3d_gate_f32: too many elements");
  hipLaunchKernelGGL(gate_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), g, y, slope, out, n);
  PC3D_LAUNCH_CHECK("pc3d_gate_f32");
  return PC3D_OK;
}

extern "C" int pc3d_att_scale_f32(const float* x, const float* w, int64_t M, int C, float* xs, float* att, void* stream) {
  PC3D_REQUIRE(M >= 0 && C >= 4 && C % 4 == 0, "pc3d_att_scale_f32: bad sizes (C %% 4 == 0)");
  if (M == 0) return PC3D_OK;
  PC3D_REQUIRE(x && w && xs && att, "pc3d_att_scale_f32: null pointer");
  hipLaunchKernelGGL(att_scale_fwd_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, as_stream(stream), x, w, M, C,
                     xs, att);
  PC3D_LAUNCH_CHECK("pc3d_att_scale_f32");
  return PC3D_OK;
}

extern "C" int pc3d_att_scale_bwd_f32(const float* g, const float* x, const float* att, const float* w, int64_t M, int C,
                                      float* gx, void* stream) {
  PC3D_REQUIRE(M >= 0 && C >= 4 && C % 4 == 0, "pc3d_att_scale_bwd_f32: bad sizes (C %% 4 == 0)");
  if (M == 0) return PC3D_OK;
  PC3D_REQUIRE(g && x && att && w && gx, "pc3d_att_scale_bwd_f32: null pointer");
  hipLaunchKernelGGL(att_scale_bwd_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, as_stream(stream), g, x, att,
                     w, M, C, gx);
  PC3D_LAUNCH_CHECK("pc3d_att_scale_bwd_f32");
  return PC3D_OK;
}

extern "C" int pc3d_topk_desc_f32(const float* score, int B, int N, int K, int32_t* idx, void* stream) {
  PC3D_REQUIRE(B >= 0 && N >= 1 && K >= 1 && K <= N && N <= 8192, "pc3d_topk_desc_f32: bad sizes B=%d N=%d K=%d (N <= 8192)",
               B, N, K);
  if (B == 0) return PC3D_OK;
  PC3D_REQUIRE(score && idx, "pc3d_topk_desc_f32: null pointer");
  int npow2 = 2;
  while (npow2 < N) npow2 <<= 1;
  hipLaunchKernelGGL(topk_desc_kernel, dim3(B), dim3(1024), (size_t)npow2 * 8, as_stream(stream), score, N, K, npow2, idx);
  PC3D_LAUNCH_CHECK("pc3d_topk_desc_f32");
  return PC3D_OK;
}

extern "C" int pc3d_lpfa_prep_f32(const float* x, const float* pts, const float* G1, const float* G2, const float* t,
                                  int64_t M, int C, float* A, float* Bc, void* stream) {
  PC3D_REQUIRE(M >= 0 && C >= 4 && C % 4 == 0, "pc3d_lpfa_prep_f32: bad sizes (C %% 4 == 0)");
  if (M == 0) return PC3D_OK;
  PC3D_REQUIRE(x && pts && G1 && G2 && t && A && Bc, "pc3d_lpfa_prep_f32: null pointer");
  const int64_t blocks = (M * (C / 4) + 255) / 256;
  PC3D_REQUIRE(blocks <= 0x7fffffffLL, "pc3d_lpfa_prep_f32: too many elements");
  hipLaunchKernelGGL(lpfa_prep_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), x, pts, G1, G2, t, M,
                     C, A, Bc);
  PC3D_LAUNCH_CHECK("pc3d_lpfa_prep_f32");
  return PC3D_OK;
}

extern "C" int pc3d_lpfa_prep_bwd_f32(const float* gA, const float* gBc, const float* G1, const float* G2, int64_t M,
                                      int C, float* gx, float* gpts, void* stream) {
  PC3D_REQUIRE(M >= 0 && C >= 4 && C % 4 == 0, "pc3d_lpfa_prep_bwd_f32: bad sizes (C %% 4 == 0)");
  if (M == 0) return PC3D_OK;
  PC3D_REQUIRE(gA && gBc && G1 && G2 && gx && gpts, "pc3d_lpfa_prep_bwd_f32: null pointer");
  const int c4 = C / 4, lp = c4 >= 32 ? 32 : c4 >= 16 ? 16 : c4 >= 8 ? 8 : c4 >= 4 ? 4 : c4 >= 2 ? 2 : 1;
  const int64_t blocks = (M * lp + 255) / 256;
  PC3D_REQUIRE(blocks <= 0x7fffffffLL, "pc3d_lpfa_prep_bwd_f32: too many elements");
  const dim3 grid((unsigned)blocks), block(256);
  switch (lp) {
    case 32: hipLaunchKernelGGL(lpfa_prep_bwd_kernel<32>, grid, block, 0, as_stream(stream), gA, gBc, G1, G2, M, C, gx, gpts); break;
    case 16: hipLaunchKernelGGL(lpfa_prep_bwd_kernel<16>, grid, block, 0, as_stream(stream), gA, gBc, G1, G2, M, C, gx, gpts); break;
    case 8: hipLaunchKernelGGL(lpfa_prep_bwd_kernel<8>, grid, block, 0, as_stream(stream), gA, gBc, G1, G2, M, C, gx, gpts); break;
    case 4: hipLaunchKernelGGL(lpfa_prep_bwd_kernel<4>, grid, block, 0, as_stream(stream), gA, gBc, G1, G2, M, C, gx, gpts); break;
    case 2: hipLaunchKernelGGL(lpfa_prep_bwd_kernel<2>, grid, block, 0, as_stream(stream), gA, gBc, G1, G2, M, C, gx, gpts); break;
    default: hipLaunchKernelGGL(lpfa_prep_bwd_kernel<1>, grid, block, 0, as_stream(stream), gA, gBc, G1, G2, M, C, gx, gpts); break;
  }
  PC3D_LAUNCH_CHECK("pc3d_lpfa_prep_bwd_f32");
  return PC3D_OK;
}

static bool curve_attn_ok(int B, int N, int C, int cn, int cl) {
  return B >= 0 && B <= 65535 && N >= 1 && cn >= 0 && cl >= 0 && cn + cl >= 1 && cn + cl <= 128 &&
         (C == 8 || C == 16 || C == 32 || C == 64) &&
         (size_t)2 * (cn + cl) * (C + 4) * sizeof(float) <= 64 * 1024;   // keys + values of one cloud in LDS
}

#define PC3D_CA_DISPATCH(C_, KERNEL, ...)                                   \
  switch (C_) {                                                             \
    case 8: hipLaunchKernelGGL((KERNEL<8>), __VA_ARGS__); break;            \
    case 16: hipLaunchKernelGGL((KERNEL<16>), __VA_ARGS__); break;          \
    case 32: hipLaunchKernelGGL((KERNEL<32>), __VA_ARGS__); break;          \
    default: hipLaunchKernelGGL((KERNEL<64>), __VA_ARGS__); break;          \
  }

extern "C" int pc3d_curve_attn_f32(const float* x, const float* Kp, const float* Vp, int B, int N, int C, int cn, int cl,
                                   float slope, float* out, void* stream) {
  PC3D_REQUIRE(curve_attn_ok(B, N, C, cn, cl),
               "pc3d_curve_attn_f32: bad sizes B=%d N=%d C=%d cn=%d cl=%d (C in {8,16,32,64}, cn + cl <= 128, 8 (cn + cl)(C + 4) bytes <= 64 KiB)", B, N, C, cn, cl);
  if (B == 0) return PC3D_OK;
  PC3D_REQUIRE(x && Kp && Vp && out, "pc3d_curve_attn_f32: null pointer");
  CurveAttnArgs a{};
  a.x = x, a.Kp = Kp, a.Vp = Vp, a.N = N, a.cn = cn, a.R = cn + cl, a.slope = slope, a.out = out;
  const size_t lds = (size_t)2 * a.R * (C + 4) * sizeof(float);
  PC3D_CA_DISPATCH(C, curve_attn_fwd_kernel, dim3(cdiv(N, CA_PPW), B), dim3(256), lds, as_stream(stream), a);
  PC3D_LAUNCH_CHECK("pc3d_curve_attn_f32");
  return PC3D_OK;
}

// slices of a cloud for the per-cloud half of the backward: ~128 points each, at least one LDS tile, enough of them
// that B * nsplit covers the chip
static int curve_attn_per_split(int B, int N) {
  // The split sets the ORDER in which a cloud's per-point terms are summed, so it is chosen from N and a nominal batch of
  // 32 clouds — never from the actual B: a cloud's gradient is the same alone and inside any batch or shard.
  (void)B;
  int per_split = 128;
  while (per_split > CA_PT && (int64_t)32 * cdiv(N, per_split) < 512) per_split /= 2;
  return per_split;
}

extern "C" int64_t pc3d_curve_attn_bwd_ws_floats(int B, int N, int C, int cn, int cl) {
  if (B <= 0 || N <= 0) return 0;
  const int64_t R = cn + cl;
  return (int64_t)B * N * (2 * R + C) + (int64_t)B * cdiv(N, curve_attn_per_split(B, N)) * 2 * C * R;
}

extern "C" int pc3d_curve_attn_bwd_f32(const float* gout, const float* out, const float* x, const float* Kp,
                                       const float* Vp, int B, int N, int C, int cn, int cl, float slope, float* gx,
                                       float* gKp, float* gVp, float* ws, void* stream) {
  PC3D_REQUIRE(curve_attn_ok(B, N, C, cn, cl),
               "pc3d_curve_attn_bwd_f32: bad sizes B=%d N=%d C=%d cn=%d cl=%d (C in {8,16,32,64}, cn + cl <= 128, 8 (cn + cl)(C + 4) bytes <= 64 KiB)", B, N, C, cn, cl);
  if (B == 0) return PC3D_OK;
  PC3D_REQUIRE(gout && out && x && Kp && Vp && gx && gKp && gVp && ws, "pc3d_curve_attn_bwd_f32: null pointer");
  const int R = cn + cl;
  CurveAttnArgs a{};
  a.x = x, a.Kp = Kp, a.Vp = Vp, a.N = N, a.cn = cn, a.R = R, a.slope = slope, a.out = const_cast<float*>(out);
  a.gout = gout, a.gx = gx;
  a.dS = ws, a.Wt = ws + (int64_t)B * N * R, a.G = ws + 2 * (int64_t)B * N * R;
  hipStream_t st = as_stream(stream);
  const size_t lds = (size_t)2 * R * (C + 4) * sizeof(float);
  PC3D_CA_DISPATCH(C, curve_attn_bwd_point_kernel, dim3(cdiv(N, CA_PPW), B), dim3(256), lds, st, a);
  PC3D_LAUNCH_CHECK("pc3d_curve_attn_bwd_f32/point");
  const int per_split = curve_attn_per_split(B, N), nsplit = cdiv(N, per_split);
  float* part = ws + (int64_t)B * N * (2 * R + C);
  const int Rp = 4 * ((R + 3) / 4);
  const size_t lds2 = (size_t)(2 * CA_PT * C + 2 * CA_PT * Rp) * sizeof(float);
  PC3D_CA_DISPATCH(C, curve_attn_bwd_cloud_kernel, dim3(nsplit, B), dim3(256), lds2, st, a, part, per_split);
  PC3D_LAUNCH_CHECK("pc3d_curve_attn_bwd_f32/cloud");
  hipLaunchKernelGGL(curve_attn_fold_kernel, dim3(cdiv(2 * C * R, 256), B), dim3(256), 0, st, part, nsplit, C * R, gKp, gVp);
  PC3D_LAUNCH_CHECK("pc3d_curve_attn_bwd_f32/fold");
  return PC3D_OK;
}
