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
  const float* gate;  // [B, O] or null: Y = gate > 0 ? Y : 0  (ReLU mask of a saved forward activation)
  int ldg;
  float* Y;           // [B, O]
  int ldy;
  int B, K, O;
  int relu;
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
  if ((a.K & 7) == 0) {
    for (int c = wave; c < nchunk; c += LN_W) {
      const int k = 8 * c + 4 * h;
      float4 xv = *reinterpret_cast<const float4*>(xrow + k);
      for (int p = 1; p < a.P; ++p) {
        const float4 t = *reinterpret_cast<const float4*>(xrow + (int64_t)p * a.K + k);
        xv.x += t.x, xv.y += t.y, xv.z += t.z, xv.w += t.w;
      }
      const float4 wv = *reinterpret_cast<const float4*>(wrow + k);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xv.x, wv.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xv.y, wv.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xv.z, wv.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xv.w, wv.w, acc, 0, 0, 0);
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
      if (a.relu) s = fmaxf(s, 0.f);
      if (a.gate && !(a.gate[(int64_t)b * a.ldg + o] > 0.f)) s = 0.f;
      a.Y[(int64_t)b * a.ldy + o] = s;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// log_softmax + prediction + adversarial loss and its gradient w.r.t. the LOGITS (pre-softmax), one wave per sample.
//   kind 0: UntargetedLogitsAdvLoss  mean_b max(real - other + kappa, 0)     (adv_utils.py:64-80)
//   kind 1: LogitsAdvLoss            mean_b max(other - real + kappa, 0)     (adv_utils.py:17-33)
//   kind 2: CrossEntropyAdvLoss      nll_loss(logp, target) (mean)           (adv_utils.py:42-51)
// where real/other are taken on the model OUTPUT, i.e. on the log-probabilities, and
// other = max_j ((1-onehot) logp - onehot * 10000).  scale multiplies the gradient (1/B for the batch mean).
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void cls_loss_kernel(const float* logits, int ld, int ncls, const int64_t* target,
                                                      int kind, float kappa, float scale, float* logp,
                                                      int64_t* pred, float* loss, float* g_logits) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const float* z = logits + (int64_t)b * ld;
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
  const float lse = m + logf(se);
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
  } else {
    lval = 0.f - real;
    gt = -1.f;
  }
  if (lane == 0) {
    if (pred) pred[b] = am;
    if (loss) loss[b] = lval;
  }
  if (g_logits) {
    // through log_softmax: g_z = g_lp - softmax * sum(g_lp)
    const float gsum = gt + ((kind == 2) ? 0.f : go);
    for (int j = lane; j < ncls; j += 64) {
      float g = (j == t ? gt : 0.f) + ((kind != 2 && j == ao) ? go : 0.f);
      g -= expf(z[j] - lse) * gsum;
      g_logits[(int64_t)b * ncls + j] = g * scale;
    }
  }
}

}  // namespace pc3d

using namespace pc3d;

extern "C" int pc3d_linear_f32(const float* X, int ldx, int P, int B, int K, const float* W, const float* bias,
                               int O, int relu, const float* gate, int ldg, float* Y, int ldy, void* stream) {
  PC3D_REQUIRE(B >= 0 && K >= 1 && O >= 1 && P >= 1, "pc3d_linear_f32: bad sizes B=%d K=%d O=%d P=%d", B, K, O, P);
  PC3D_REQUIRE(ldx >= P * K && ldy >= O, "pc3d_linear_f32: leading dimensions too small (ldx=%d ldy=%d)", ldx, ldy);
  PC3D_REQUIRE((K % 8 != 0) || (ldx % 4 == 0), "pc3d_linear_f32: ldx=%d must be a multiple of 4 for 16-byte loads", ldx);
  if (B == 0) return PC3D_OK;
  PC3D_REQUIRE(X && W && Y, "pc3d_linear_f32: null pointer");
  PC3D_REQUIRE(gate == nullptr || ldg >= O, "pc3d_linear_f32: ldg=%d too small", ldg);
  LinArgs a{X, ldx, P, W, bias, gate, ldg, Y, ldy, B, K, O, relu};
  hipLaunchKernelGGL(linear_kernel, dim3(cdiv(O, 32), cdiv(B, 32)), dim3(LN_T), 0, as_stream(stream), a);
  PC3D_LAUNCH_CHECK("pc3d_linear_f32");
  return PC3D_OK;
}

extern "C" int pc3d_cls_loss_f32(const float* logits, int ld, int B, int ncls, const int64_t* target, int kind,
                                 float kappa, float scale, float* logp, int64_t* pred, float* loss,
                                 float* g_logits, void* stream) {
  PC3D_REQUIRE(B >= 0 && ncls >= 2 && ld >= ncls, "pc3d_cls_loss_f32: bad sizes B=%d ncls=%d ld=%d", B, ncls, ld);
  PC3D_REQUIRE(kind >= 0 && kind <= 2, "pc3d_cls_loss_f32: kind=%d not in {0,1,2}", kind);
  if (B == 0) return PC3D_OK;
  PC3D_REQUIRE(logits && target, "pc3d_cls_loss_f32: null pointer");
  hipLaunchKernelGGL(cls_loss_kernel, dim3(B), dim3(64), 0, as_stream(stream), logits, ld, ncls, target, kind, kappa,
                     scale, logp, pred, loss, g_logits);
  PC3D_LAUNCH_CHECK("pc3d_cls_loss_f32");
  return PC3D_OK;
}
