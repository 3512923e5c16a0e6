// Small-batch dense layers of the classifier heads (model/pointnet.py:38-47 STN3d fc1-3, :144-148 PointNetCls
// fc1-3 + log_softmax) and the adversarial-loss gradient on the logits (attack/CW/CW_utils/adv_utils.py), forward
// and backward, as gfx950 kernels — the per-iteration replacement of ~14 rocBLAS launches + ~60 elementwise launches.
//
// pc3d_linear_f32: Y[b,o] = epilogue( sum_k X[b,k] W[o,k] + bias[o] ) for B <= a few hundred rows.
//   One workgroup = 32 rows x 32 outputs on v_mfma_f32_32x32x2_f32 (exact fp32), K split over the 8 waves, operands
//   straight from L2 into registers (each operand row is touched once per workgroup: no LDS staging), partial tiles
//   summed through LDS in fixed wave order (deterministic). Backward of a linear layer is the same kernel on the
//   transposed weight (the host keeps W^T next to W: weights are frozen).
#include "pc3d_common.h"

namespace pc3d {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int LN_T = 512;  // threads: 8 waves split K
constexpr int LN_W = LN_T / 64;

struct LinArgs {
  const float* X;     // [B, P, K] (P partial slabs summed on load; P = 1 for a plain matrix)
  int ldx;            // row stride of X in floats (>= P*K)
  int P;
  const float* W;     // [O, K] row-major
  const float* bias;  // [O] or null
  const float* gate;  // [B, O] or null: Y = gate > 0 ? Y : gslope * Y  (the (Leaky)ReLU mask of a saved forward activation)
  int ldg;
  float* Y;           // [B, O]
  int ldy;
  int B, K, O;
  int relu;           // 1: Y = Y > 0 ? Y : slope * Y  (slope 0 = ReLU)
  float slope, gslope;
};

__global__ __launch_bounds__(LN_T) void linear_kernel(LinArgs a) {
  __shared__ float red[LN_W][32][33];
  const int o0 = blockIdx.x * 32, b0 = blockIdx.y * 32;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  // operand rows (clamped: out-of-range rows are computed and discarded)
  const int xb = (b0 + r < a.B) ? b0 + r : a.B - 1;
  const int wo = (o0 + r < a.O) ? o0 + r : a.O - 1;
  const float* xrow = a.X + (int64_t)xb * a.ldx;
  const float* wrow = a.W + (int64_t)wo * a.K;
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  // K is walked in chunks of 8 (4 per lane half); wave w takes chunks w, w+8, ...
  const int nchunk = a.K / 8;
  if ((a.K & 7) == 0 && a.P == 1 && nchunk <= 16 * LN_W) {
    // all operands of this wave's K range are fetched up front (<= 16 + 16 float4 per lane): one L2 round trip
    // instead of one per 4 MFMAs — these launches are latency-, not throughput-bound
    float4 xv[16], wv[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int c = wave + i * LN_W;
      if (c < nchunk) {
        xv[i] = *reinterpret_cast<const float4*>(xrow + 8 * c + 4 * h);
        wv[i] = *reinterpret_cast<const float4*>(wrow + 8 * c + 4 * h);
      }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int c = wave + i * LN_W;
      if (c < nchunk) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xv[i].x, wv[i].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xv[i].y, wv[i].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xv[i].z, wv[i].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xv[i].w, wv[i].w, acc, 0, 0, 0);
      }
    }
  } else if ((a.K & 7) == 0) {
    // P partial slabs: sum_p (x_p . W) = (sum_p x_p) . W, so every (slab, chunk) pair is an independent work item
    // spread over the waves (the tower backward hands over one dL/dT partial per 32-point tile: P = N / 32)
    const int items = nchunk * a.P;
    for (int it0 = wave; it0 < items; it0 += 4 * LN_W) {
      float4 xv[4], wv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {          // four items' operands in flight
        const int it = it0 + u * LN_W;
        if (it < items) {
          const int p = it / nchunk, c = it - p * nchunk;
          const int k = 8 * c + 4 * h;
          xv[u] = *reinterpret_cast<const float4*>(xrow + (int64_t)p * a.K + k);
          wv[u] = *reinterpret_cast<const float4*>(wrow + k);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (it0 + u * LN_W < items) {
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xv[u].x, wv[u].x, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xv[u].y, wv[u].y, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xv[u].z, wv[u].z, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xv[u].w, wv[u].w, acc, 0, 0, 0);
        }
      }
    }
  } else {  // ragged K (e.g. the 9 entries of the STN transform): scalar loads with zero fill
    for (int c = wave; c * 8 < a.K; c += LN_W) {
      float xs[4], ws[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int k = 8 * c + 4 * h + e;
        float x = 0.f, w = 0.f;
        if (k < a.K) {
          x = xrow[k];
          for (int p = 1; p < a.P; ++p) x += xrow[(int64_t)p * a.K + k];
          w = wrow[k];
        }
        xs[e] = x, ws[e] = w;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xs[e], ws[e], acc, 0, 0, 0);
    }
  }
  // D[row = sample][col = output]: lane holds col r, rows (e&3) + 8*(e>>2) + 4*h
#pragma unroll
  for (int e = 0; e < 16; ++e) red[wave][(e & 3) + 8 * (e >> 2) + 4 * h][r] = acc[e];
  __syncthreads();
  for (int i = threadIdx.x; i < 32 * 32; i += LN_T) {
    const int row = i >> 5, col = i & 31;
    float s = red[0][row][col];
#pragma unroll
    for (int w = 1; w < LN_W; ++w) s += red[w][row][col];
    const int b = b0 + row, o = o0 + col;
    if (b < a.B && o < a.O) {
      if (a.bias) s += a.bias[o];
      if (a.relu) s = s > 0.f ? s : s * a.slope;
      if (a.gate && !(a.gate[(int64_t)b * a.ldg + o] > 0.f)) s *= a.gslope;
      a.Y[(int64_t)b * a.ldy + o] = s;
    }
  }
}

// 32 rows x 16 outputs per workgroup on v_mfma_f32_16x16x4_f32: twice the workgroups (CUs) of the 32 x 32 tiling for
// the same layer, half the MFMA time and half the weight bytes per workgroup — these launches are latency-bound and
// use at most O/32 of the 256 CUs. Operand mapping: lane (r = lane & 15, q = lane >> 4) holds the float4 at
// k = 16c + 4q of row r; MFMA call e consumes element e of every lane's float4, so its four k-slices are
// k = 16c + 4q + e — the same permutation on both operands, i.e. an exact fp32 sum in a permuted k order.
using f32x4 = __attribute__((ext_vector_type(4))) float;

__global__ __launch_bounds__(LN_T) void linear16_kernel(LinArgs a) {
  __shared__ float red[LN_W][32][17];
  const int o0 = blockIdx.x * 16, b0 = blockIdx.y * 32;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int xb0 = (b0 + r < a.B) ? b0 + r : a.B - 1;
  const int xb1 = (b0 + 16 + r < a.B) ? b0 + 16 + r : a.B - 1;
  const int wo = (o0 + r < a.O) ? o0 + r : a.O - 1;
  const float* x0 = a.X + (int64_t)xb0 * a.ldx + 4 * q;
  const float* x1 = a.X + (int64_t)xb1 * a.ldx + 4 * q;
  const float* wr = a.W + (int64_t)wo * a.K + 4 * q;
  const int nchunk = a.K / 16;          // chunks of 16 k; wave w takes chunks w, w+8, ... (<= 8 per wave)
  float4 xv0[8], xv1[8], wv[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = wave + i * LN_W;
    if (c < nchunk) {
      xv0[i] = *reinterpret_cast<const float4*>(x0 + 16 * c);
      xv1[i] = *reinterpret_cast<const float4*>(x1 + 16 * c);
      wv[i] = *reinterpret_cast<const float4*>(wr + 16 * c);
    }
  }
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = wave + i * LN_W;
    if (c < nchunk) {   // D[row = sample][col = output]; two independent accumulators cover the 40-cycle MFMA latency
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xv0[i].x, wv[i].x, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xv1[i].x, wv[i].x, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xv0[i].y, wv[i].y, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xv1[i].y, wv[i].y, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xv0[i].z, wv[i].z, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xv1[i].z, wv[i].z, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xv0[i].w, wv[i].w, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xv1[i].w, wv[i].w, acc1, 0, 0, 0);
    }
  }
  // 16x16x4 result layout: lane holds column r, rows 4*q + e
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    red[wave][4 * q + e][r] = acc0[e];
    red[wave][16 + 4 * q + e][r] = acc1[e];
  }
  __syncthreads();
  {
    const int row = threadIdx.x >> 4, col = threadIdx.x & 15;   // 512 threads = 32 x 16 outputs
    float s = red[0][row][col];
#pragma unroll
    for (int w = 1; w < LN_W; ++w) s += red[w][row][col];
    const int b = b0 + row, o = o0 + col;
    if (b < a.B && o < a.O) {
      if (a.bias) s += a.bias[o];
      if (a.relu) s = s > 0.f ? s : s * a.slope;
      if (a.gate && !(a.gate[(int64_t)b * a.ldg + o] > 0.f)) s *= a.gslope;
      a.Y[(int64_t)b * a.ldy + o] = s;
    }
  }
}

// The same 32 x 16 tiling with the X operand produced on the fly by a tiny PRE-layer:
//     X[b,k] = gate_pre[b,k] > 0 ? sum_{j<J} S[b,j] Wp[j,k] : 0,      S[b,j] = sum_p parts[b,p,j]
// — the backward of STN3d's fc3 (9 -> 256, model/pointnet.py:45) folded into the launch of fc2's backward
// (256 -> 512): `parts` are the per-tile dL/dT partials of the trunk's backward, Wp = fc3's weight [9,256], gate_pre
// = fc2's ReLU output. J <= 16, K <= 16 * 8 * 8 = 1024.
struct LinPreArgs {
  const float* parts;    // [B, P, Jp] (Jp >= J floats per slab, J used)
  int P, Jp, J;
  const float* Wp;       // [J, K] row-major
  const float* gate_pre; // [B, K]
  int ldgp;
  const float* W;        // [O, K]
  const float* gate;     // [B, O] or null
  int ldg;
  float* Y;              // [B, O]
  int ldy;
  int B, K, O;
};

__global__ __launch_bounds__(LN_T) void linear16_pre_kernel(LinPreArgs a) {
  __shared__ float red[LN_W][32][17];
  __shared__ float S[32][16];
  const int o0 = blockIdx.x * 16, b0 = blockIdx.y * 32;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, q = lane >> 4;
  {   // S: 32 rows x 16 columns = 512 threads, one element each, slabs summed in ascending order
    const int row = threadIdx.x >> 4, j = threadIdx.x & 15;
    const int bb = (b0 + row < a.B) ? b0 + row : a.B - 1;
    float sacc = 0.f;
    if (j < a.J) {
      const float* pp = a.parts + (int64_t)bb * a.P * a.Jp + j;
      int p = 0;
      for (; p + 8 <= a.P; p += 8) {       // eight slabs in flight; summed in ascending order
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = pp[(int64_t)(p + u) * a.Jp];
#pragma unroll
        for (int u = 0; u < 8; ++u) sacc += v[u];
      }
      for (; p < a.P; ++p) sacc += pp[(int64_t)p * a.Jp];
    }
    S[row][j] = sacc;
  }
  __syncthreads();
  const int xb0 = (b0 + r < a.B) ? b0 + r : a.B - 1;
  const int xb1 = (b0 + 16 + r < a.B) ? b0 + 16 + r : a.B - 1;
  const int wo = (o0 + r < a.O) ? o0 + r : a.O - 1;
  const float* wr = a.W + (int64_t)wo * a.K + 4 * q;
  const int nchunk = a.K / 16;
  using f32x4 = __attribute__((ext_vector_type(4))) float;
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  for (int c = wave; c < nchunk; c += LN_W) {
    const int k0 = 16 * c + 4 * q;
    float x0[4] = {0.f, 0.f, 0.f, 0.f}, x1[4] = {0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < a.J; ++j) {
      const float4 wp = *reinterpret_cast<const float4*>(a.Wp + (int64_t)j * a.K + k0);
      const float s0 = S[r][j], s1 = S[16 + r][j];
      x0[0] = __builtin_fmaf(s0, wp.x, x0[0]), x0[1] = __builtin_fmaf(s0, wp.y, x0[1]);
      x0[2] = __builtin_fmaf(s0, wp.z, x0[2]), x0[3] = __builtin_fmaf(s0, wp.w, x0[3]);
      x1[0] = __builtin_fmaf(s1, wp.x, x1[0]), x1[1] = __builtin_fmaf(s1, wp.y, x1[1]);
      x1[2] = __builtin_fmaf(s1, wp.z, x1[2]), x1[3] = __builtin_fmaf(s1, wp.w, x1[3]);
    }
    const float4 g0 = *reinterpret_cast<const float4*>(a.gate_pre + (int64_t)xb0 * a.ldgp + k0);
    const float4 g1 = *reinterpret_cast<const float4*>(a.gate_pre + (int64_t)xb1 * a.ldgp + k0);
    const float ge0[4] = {g0.x, g0.y, g0.z, g0.w}, ge1[4] = {g1.x, g1.y, g1.z, g1.w};
    const float4 wv = *reinterpret_cast<const float4*>(wr + 16 * c);
    const float we[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float v0 = ge0[e] > 0.f ? x0[e] : 0.f, v1 = ge1[e] > 0.f ? x1[e] : 0.f;
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(v0, we[e], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(v1, we[e], acc1, 0, 0, 0);
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    red[wave][4 * q + e][r] = acc0[e];
    red[wave][16 + 4 * q + e][r] = acc1[e];
  }
  __syncthreads();
  {
    const int row = threadIdx.x >> 4, col = threadIdx.x & 15;
    float sum = red[0][row][col];
#pragma unroll
    for (int w = 1; w < LN_W; ++w) sum += red[w][row][col];
    const int b = b0 + row, o = o0 + col;
    if (b < a.B && o < a.O) {
      if (a.gate && !(a.gate[(int64_t)b * a.ldg + o] > 0.f)) sum = 0.f;
      a.Y[(int64_t)b * a.ldy + o] = sum;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// log_softmax + prediction + adversarial loss and its gradient w.r.t. the LOGITS (pre-softmax), one wave per sample.
//   kind 0: UntargetedLogitsAdvLoss  mean_b max(real - other + kappa, 0)     (adv_utils.py:64-80)
//   kind 1: LogitsAdvLoss            mean_b max(other - real + kappa, 0)     (adv_utils.py:17-33)
//   kind 2: CrossEntropyAdvLoss      nll_loss(logp, target) (mean)           (adv_utils.py:42-51)
//   kind 3: minus kind 2             (attack/GeoA3/GeoA3_attack.py:125-127: the untargeted classification loss -CE)
// where real/other are taken on the model OUTPUT, i.e. on the log-probabilities, and
// other = max_j ((1-onehot) logp - onehot * 10000).  scale multiplies the gradient (1/B for the batch mean).
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void cls_loss_kernel(const float* logits, int ld, int ncls, const int64_t* target,
                                                      int kind, float kappa, float scale, float* logp,
                                                      int64_t* pred, float* loss, float* g_logits) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const float* z = logits + (int64_t)b * ld;
  const bool raw = (kind & 4) != 0;     // the loss is taken on z itself (what the reference's functors do with whatever the
  kind &= 3;                            // victim returns: log-probabilities or raw logits), not on log_softmax(z)
  float m = -__builtin_inff();
  int am = 0;
  for (int j = lane; j < ncls; j += 64) {
    const float v = z[j];
    if (v > m) m = v, am = j;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(m, o, 64);
    const int oi = __shfl_xor(am, o, 64);
    if (ov > m || (ov == m && oi < am)) m = ov, am = oi;
  }
  float se = 0.f;
  for (int j = lane; j < ncls; j += 64) se += expf(z[j] - m);
  se = wave_sum(se);
  const float lse = raw ? 0.f : m + logf(se);
  const int t = (int)target[b];
  float other = -__builtin_inff();
  int ao = 0;
  for (int j = lane; j < ncls; j += 64) {
    const float lp = z[j] - lse;
    if (logp) logp[(int64_t)b * ncls + j] = lp;
    const float cand = (j == t) ? -10000.f : lp;   // (1-onehot)*logp - onehot*10000
    if (cand > other) other = cand, ao = j;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(other, o, 64);
    const int oi = __shfl_xor(ao, o, 64);
    if (ov > other || (ov == other && oi < ao)) other = ov, ao = oi;
  }
  const float real = z[t] - lse;
  float lval = 0.f, gt = 0.f, go = 0.f;  // d loss / d logp at the target / at the runner-up
  if (kind == 0) {
    const float mgn = real - other + kappa;
    lval = fmaxf(mgn, 0.f);
    if (mgn > 0.f) gt = 1.f, go = -1.f;
  } else if (kind == 1) {
    const float mgn = other - real + kappa;
    lval = fmaxf(mgn, 0.f);
    if (mgn > 0.f) gt = -1.f, go = 1.f;
  } else if (kind == 2) {
    lval = 0.f - real;
    gt = -1.f;
  } else {                                 // kind 3: MINUS the cross-entropy (GeoA3's untargeted classification loss)
    lval = real;
    gt = 1.f;
  }
  if (lane == 0) {
    if (pred) pred[b] = am;
    if (loss) loss[b] = lval;
  }
  if (g_logits) {
    // through log_softmax: g_z = g_lp - softmax * sum(g_lp)
    const float gsum = gt + ((kind >= 2) ? 0.f : go);
    for (int j = lane; j < ncls; j += 64) {
      float g = (j == t ? gt : 0.f) + ((kind < 2 && j == ao) ? go : 0.f);
      if (!raw) g -= expf(z[j] - lse) * gsum;
      g_logits[(int64_t)b * ncls + j] = g * scale;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// Classifier tail in ONE launch (one workgroup per 32 samples): logits = c2 W3^T + b3, log_softmax, prediction,
// adversarial loss + its gradient on the logits, and g_c2 = (g_logits W3) * (c2 > 0) — i.e. fc3 forward, the loss
// kernel above and fc3 backward (+ ReLU mask of fc2's activation) without three launches. Also advances the Adam
// step word (it is the one single-workgroup launch of the iteration).  K2 = width of c2 (256), ncls <= 64.
// ---------------------------------------------------------------------------------------------------------
constexpr int CT_T = 256;      // 4 waves = 4 samples per workgroup
constexpr int CT_S = 4;
constexpr int CT_MAXCLS = 64;
constexpr int CT_K = 256;
constexpr int CT_LD = CT_K + 4;  // 16-byte aligned rows; lane*260 mod 32 = 4*lane -> b128 reads are conflict-free

struct ClsTailArgs {
  const float* c2;      // [B,K2] post-ReLU activation of fc2
  const float* W3;      // [ncls,K2]
  const float* b3;      // [ncls]
  const int64_t* target;
  int B, K2, ncls, kind;
  float kappa, scale;
  float* logp;          // [B,ncls] or null
  int64_t* pred;        // [B]
  float* loss;          // [B]
  float* g_c2;          // [B,K2]
  int32_t* step;        // incremented by block 0 (may be null)
};

__global__ __launch_bounds__(CT_T) void cls_tail_kernel(ClsTailArgs a) {
  __shared__ __attribute__((aligned(16))) float s_w[CT_MAXCLS * CT_LD];   // W3 (zero rows above ncls)
  __shared__ __attribute__((aligned(16))) float s_c[CT_S * CT_LD];        // this block's rows of c2
  __shared__ float s_g[CT_S][CT_MAXCLS];                                  // g_logits
  const int b0 = blockIdx.x * CT_S;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kq = a.K2 >> 2;   // float4 per row
  for (int i = tid; i < CT_MAXCLS * kq; i += CT_T) {
    const int r = i / kq, k4 = i - r * kq;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < a.ncls) v = *reinterpret_cast<const float4*>(a.W3 + (int64_t)r * a.K2 + 4 * k4);
    *reinterpret_cast<float4*>(&s_w[r * CT_LD + 4 * k4]) = v;
  }
  for (int i = tid; i < CT_S * kq; i += CT_T) {
    const int r = i / kq, k4 = i - r * kq;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (b0 + r < a.B) v = *reinterpret_cast<const float4*>(a.c2 + (int64_t)(b0 + r) * a.K2 + 4 * k4);
    *reinterpret_cast<float4*>(&s_c[r * CT_LD + 4 * k4]) = v;
  }
  const int b = b0 + wave;
  const float bias = (lane < a.ncls) ? a.b3[lane] : 0.f;
  const int t = (b < a.B) ? (int)a.target[b] : 0;
  __syncthreads();
  // logits of sample `wave`: lane = class, four interleaved fma chains over k (a re-association of the fp32 sum
  // pc3d_linear_f32 forms; the tests bound the difference)
  {
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const float* wr = &s_w[lane * CT_LD];
    const float* cr = &s_c[wave * CT_LD];
#pragma unroll 4
    for (int k = 0; k < a.K2; k += 4) {
      const float4 w = *reinterpret_cast<const float4*>(wr + k);
      const float4 c = *reinterpret_cast<const float4*>(cr + k);
      acc[0] = __builtin_fmaf(c.x, w.x, acc[0]);
      acc[1] = __builtin_fmaf(c.y, w.y, acc[1]);
      acc[2] = __builtin_fmaf(c.z, w.z, acc[2]);
      acc[3] = __builtin_fmaf(c.w, w.w, acc[3]);
    }
    const float zj = (lane < a.ncls) ? ((acc[0] + acc[1]) + (acc[2] + acc[3]) + bias) : -__builtin_inff();
    float m = zj;
    int am = lane;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(m, o, 64);
      const int oi = __shfl_xor(am, o, 64);
      if (ov > m || (ov == m && oi < am)) m = ov, am = oi;
    }
    const float ex = (lane < a.ncls) ? expf(zj - m) : 0.f;
    const float lse = m + logf(wave_sum(ex));
    const float lp = zj - lse;
    float other = (lane < a.ncls) ? ((lane == t) ? -10000.f : lp) : -__builtin_inff();
    int ao = lane;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(other, o, 64);
      const int oi = __shfl_xor(ao, o, 64);
      if (ov > other || (ov == other && oi < ao)) other = ov, ao = oi;
    }
    const float real = __shfl(lp, t, 64);
    float lval = 0.f, gt = 0.f, go = 0.f;
    if (a.kind == 0) {
      const float mg = real - other + a.kappa;
      lval = fmaxf(mg, 0.f);
      if (mg > 0.f) gt = 1.f, go = -1.f;
    } else if (a.kind == 1) {
      const float mg = other - real + a.kappa;
      lval = fmaxf(mg, 0.f);
      if (mg > 0.f) gt = -1.f, go = 1.f;
    } else {
      lval = 0.f - real;
      gt = -1.f;
    }
    const float gsum = gt + ((a.kind == 2) ? 0.f : go);
    float g = (lane == t ? gt : 0.f) + ((a.kind != 2 && lane == ao) ? go : 0.f);
    g -= expf(zj - lse) * gsum;
    s_g[wave][lane] = (lane < a.ncls) ? g * a.scale : 0.f;
    if (b < a.B) {
      if (a.logp && lane < a.ncls) a.logp[(int64_t)b * a.ncls + lane] = lp;
      if (lane == 0) {
        a.pred[b] = am;
        if (a.loss) a.loss[b] = lval;
      }
    }
  }
  __syncthreads();
  // g_c2[b,k] = (sum_c g_logits[b,c] W3[c,k]) * (c2[b,k] > 0): thread = k, all CT_S samples at once
  if (tid < a.K2) {
    float acc[CT_S] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
    for (int c = 0; c < a.ncls; ++c) {
      const float w = s_w[c * CT_LD + tid];
#pragma unroll
      for (int j = 0; j < CT_S; ++j) acc[j] = __builtin_fmaf(s_g[j][c], w, acc[j]);
    }
#pragma unroll
    for (int j = 0; j < CT_S; ++j)
      if (b0 + j < a.B) a.g_c2[(int64_t)(b0 + j) * a.K2 + tid] = (s_c[j * CT_LD + tid] > 0.f) ? acc[j] : 0.f;
  }
  if (a.step && blockIdx.x == 0 && tid == 0) a.step[0] += 1;
}

}  // namespace pc3d

using namespace pc3d;

extern "C" int pc3d_cls_tail_f32(const float* c2, int B, int K2, const float* W3, const float* b3, int ncls,
                                 const int64_t* target, int kind, float kappa, float scale, float* logp,
                                 int64_t* pred, float* loss, float* g_c2, int32_t* step, void* stream) {
  PC3D_REQUIRE(B >= 0 && K2 >= 4 && K2 <= CT_K && (K2 % 4) == 0 && ncls >= 2 && ncls <= CT_MAXCLS,
               "pc3d_cls_tail_f32: unsupported sizes B=%d K2=%d ncls=%d (K2 <= 256, K2 %% 4 == 0, ncls <= 64)", B, K2, ncls);
  PC3D_REQUIRE(kind >= 0 && kind <= 2, "pc3d_cls_tail_f32: kind=%d not in {0,1,2}", kind);
  if (B == 0) return PC3D_OK;
  PC3D_REQUIRE(c2 && W3 && b3 && target && pred && g_c2, "pc3d_cls_tail_f32: null pointer");
  ClsTailArgs a{c2, W3, b3, target, B, K2, ncls, kind, kappa, scale, logp, pred, loss, g_c2, step};
  hipLaunchKernelGGL(cls_tail_kernel, dim3(cdiv(B, CT_S)), dim3(CT_T), 0, as_stream(stream), a);
  PC3D_LAUNCH_CHECK("pc3d_cls_tail_f32");
  return PC3D_OK;
}

extern "C" int pc3d_linear_f32(const float* X, int ldx, int P, int B, int K, const float* W, const float* bias,
                               int O, int relu, float slope, const float* gate, int ldg, float gate_slope, float* Y, int ldy,
                               void* stream) {
  PC3D_REQUIRE(B >= 0 && K >= 1 && O >= 1 && P >= 1, "pc3d_linear_f32: bad sizes B=%d K=%d O=%d P=%d", B, K, O, P);
  PC3D_REQUIRE(ldx >= P * K && ldy >= O, "pc3d_linear_f32: leading dimensions too small (ldx=%d ldy=%d)", ldx, ldy);
  PC3D_REQUIRE((K % 8 != 0) || (ldx % 4 == 0), "pc3d_linear_f32: ldx=%d must be a multiple of 4 for 16-byte loads", ldx);
  if (B == 0) return PC3D_OK;
  PC3D_REQUIRE(X && W && Y, "pc3d_linear_f32: null pointer");
  PC3D_REQUIRE(gate == nullptr || ldg >= O, "pc3d_linear_f32: ldg=%d too small", ldg);
  LinArgs a{X, ldx, P, W, bias, gate, ldg, Y, ldy, B, K, O, relu, slope, gate_slope};
  if ((K & 15) == 0 && P == 1 && K / 16 <= 8 * LN_W && O >= 64)
    hipLaunchKernelGGL(linear16_kernel, dim3(cdiv(O, 16), cdiv(B, 32)), dim3(LN_T), 0, as_stream(stream), a);
  else
    hipLaunchKernelGGL(linear_kernel, dim3(cdiv(O, 32), cdiv(B, 32)), dim3(LN_T), 0, as_stream(stream), a);
  PC3D_LAUNCH_CHECK("pc3d_linear_f32");
  return PC3D_OK;
}

extern "C" int pc3d_cls_loss_f32(const float* logits, int ld, int B, int ncls, const int64_t* target, int kind,
                                 float kappa, float scale, float* logp, int64_t* pred, float* loss,
                                 float* g_logits, void* stream) {
  PC3D_REQUIRE(B >= 0 && ncls >= 2 && ld >= ncls, "pc3d_cls_loss_f32: bad sizes B=%d ncls=%d ld=%d", B, ncls, ld);
  PC3D_REQUIRE(kind >= 0 && kind <= 7, "pc3d_cls_loss_f32: kind=%d not in {0,1,2,3} (+4)", kind);
  if (B == 0) return PC3D_OK;
  PC3D_REQUIRE(logits && target, "pc3d_cls_loss_f32: null pointer");
  hipLaunchKernelGGL(cls_loss_kernel, dim3(B), dim3(64), 0, as_stream(stream), logits, ld, ncls, target, kind, kappa,
                     scale, logp, pred, loss, g_logits);
  PC3D_LAUNCH_CHECK("pc3d_cls_loss_f32");
  return PC3D_OK;
}

extern "C" int pc3d_linear_pre_f32(const float* parts, int P, int Jp, int J, const float* Wp, const float* gate_pre,
                                   int ldgp, int B, int K, const float* W, int O, const float* gate, int ldg, float* Y,
                                   int ldy, void* stream) {
  PC3D_REQUIRE(B >= 0 && P >= 1 && J >= 1 && J <= 16 && Jp >= J && K >= 16 && K % 16 == 0 && O >= 1,
               "pc3d_linear_pre_f32: bad sizes B=%d P=%d J=%d Jp=%d K=%d O=%d (J <= 16, K %% 16 == 0)", B, P, J, Jp, K, O);
  if (B == 0) return PC3D_OK;
  PC3D_REQUIRE(parts && Wp && gate_pre && W && Y && ldgp >= K && ldy >= O && (!gate || ldg >= O),
               "pc3d_linear_pre_f32: null pointer or row stride smaller than the row");
  LinPreArgs a{parts, P, Jp, J, Wp, gate_pre, ldgp, W, gate, ldg, Y, ldy, B, K, O};
  hipLaunchKernelGGL(linear16_pre_kernel, dim3(cdiv(O, 16), cdiv(B, 32)), dim3(LN_T), 0, as_stream(stream), a);
  PC3D_LAUNCH_CHECK("pc3d_linear_pre_f32");
  return PC3D_OK;
}
