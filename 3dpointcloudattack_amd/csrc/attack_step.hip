// K11 + K10/K9 fused for the CW-family loops: the per-iteration bookkeeping of the reference
// (attack/CW/CW_attack.py:121-153 — argmax already done by the loss kernel; per-sample L2 perturbation norm,
// best-distance / best-attack selection, the D2H copy of the whole cloud and the Python loop over samples) and the
// optimiser update (loss.backward() of the distance term + Adam + clip, :160-174) as TWO launches
// (cw_bookkeep_kernel + cw_step_kernel) or as ONE (cw_update_kernel, what the captured CW iteration uses).
#include "pc3d_common.h"

namespace pc3d {

struct BookArgs {
  PtsView adv, ori;          // [B,K] points
  int K;
  const int64_t* pred;       // [B]
  const int64_t* label;      // [B]
  int untarget;              // success = pred != label (1) or pred == label (0)
  float* bestdist;           // [B] per-binary-step best
  int64_t* bestscore;        // [B]
  float* o_bestdist;         // [B] overall best
  int64_t* o_bestscore;      // [B]
  PtsViewMut o_bestattack;   // [B,K] points: copy of adv where the overall best improved
  PtsViewMut input_val;      // [B,K] points: always the iterate this pass started from (may be null)
  float* dist_val;           // [B] out: ||adv-ori||_F (feeds the L2 distance gradient)
  int32_t* step;             // Adam step word, incremented once per launch (may be null)
};

// one workgroup per sample
__global__ __launch_bounds__(256) void cw_bookkeep_kernel(BookArgs a) {
  __shared__ float part[4];
  __shared__ int s_copy;
  const int b = blockIdx.x;
  float acc = 0.f;
  for (int k = threadIdx.x; k < a.K; k += 256) {
    const float* p = a.adv.p + (int64_t)b * a.adv.bs + (int64_t)k * a.adv.ps;
    const float* o = a.ori.p + (int64_t)b * a.ori.bs + (int64_t)k * a.ori.ps;
    const float dx = p[0] - o[0], dy = p[a.adv.cs] - o[a.ori.cs], dz = p[2 * a.adv.cs] - o[2 * a.ori.cs];
    acc += dx * dx + dy * dy + dz * dz;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float dist = __builtin_sqrtf(part[0] + part[1] + part[2] + part[3]);
    if (a.dist_val) a.dist_val[b] = dist;
    const int64_t pr = a.pred[b], lb = a.label[b];
    const bool succ = a.untarget ? (pr != lb) : (pr == lb);
    if (succ && dist < a.bestdist[b]) {
      a.bestdist[b] = dist;
      a.bestscore[b] = pr;
    }
    int copy = 0;
    if (succ && dist < a.o_bestdist[b]) {
      a.o_bestdist[b] = dist;
      a.o_bestscore[b] = pr;
      copy = 1;
    }
    s_copy = copy;
    if (b == 0 && a.step) a.step[0] += 1;
  }
  __syncthreads();
  const bool copy = s_copy != 0;
  if (!copy && a.input_val.p == nullptr) return;
  for (int k = threadIdx.x; k < a.K; k += 256) {
    const float* p = a.adv.p + (int64_t)b * a.adv.bs + (int64_t)k * a.adv.ps;
    const float x = p[0], y = p[a.adv.cs], z = p[2 * a.adv.cs];
    if (a.input_val.p) {
      float* q = a.input_val.p + (int64_t)b * a.input_val.bs + (int64_t)k * a.input_val.ps;
      q[0] = x, q[a.input_val.cs] = y, q[2 * a.input_val.cs] = z;
    }
    if (copy) {
      float* q = a.o_bestattack.p + (int64_t)b * a.o_bestattack.bs + (int64_t)k * a.o_bestattack.ps;
      q[0] = x, q[a.o_bestattack.cs] = y, q[2 * a.o_bestattack.cs] = z;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// CW update: g = g_model + d/dadv [ mean_b w_b * D(adv_b, ori_b) ], then Adam, then per-point clip, one launch.
//   dist_kind 0: no distance term (caller already added it to g_model)
//   dist_kind 1: L2Dist      D = ||adv-ori||_F            -> w_b/B * (adv-ori)/D      (dist_utils.py:30-33)
//   dist_kind 2: ChamferDist adv2ori, D = mean_i |adv_i - ori_nn(i)|^2 -> 2 w_b/(B K) (adv_i - ori_nn(i))
//                                                                       (dist_utils.py:60-63, distance.py:44-46)
// ---------------------------------------------------------------------------------------------------------
struct StepArgs {
  PtsViewMut p;
  PtsView g;
  PtsViewMut m, v;
  PtsView ori;
  int K, B;
  double lr, b1, b2;
  float eps, budget;
  const int* step_dev;
  int step_host;
  int dist_kind;
  const float* w;         // [B] distance weights (binary-search variable)
  const float* l2norm;    // [B] (kind 1)
  const int32_t* nn_idx;  // [B,K] (kind 2)
};

__global__ __launch_bounds__(256) void cw_step_kernel(StepArgs a) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  if (k >= a.K) return;
  const int t = a.step_dev ? a.step_dev[0] : a.step_host;
  const float omb1 = (float)(1.0 - a.b1), omb2 = (float)(1.0 - a.b2), fb2 = (float)a.b2;
  const float step_size = (float)(a.lr / (1.0 - pow(a.b1, (double)t)));
  const float bc2s = (float)sqrt(1.0 - pow(a.b2, (double)t));
  float* pp = a.p.p + (int64_t)b * a.p.bs + (int64_t)k * a.p.ps;
  const float* gp = a.g.p + (int64_t)b * a.g.bs + (int64_t)k * a.g.ps;
  float* mp = a.m.p + (int64_t)b * a.m.bs + (int64_t)k * a.m.ps;
  float* vp = a.v.p + (int64_t)b * a.v.bs + (int64_t)k * a.v.ps;
  const float* o = a.ori.p + (int64_t)b * a.ori.bs + (int64_t)k * a.ori.ps;
  const float ox = o[0], oy = o[a.ori.cs], oz = o[2 * a.ori.cs];
  const float px = pp[0], py = pp[a.p.cs], pz = pp[2 * a.p.cs];
  float g[3] = {gp[0], gp[a.g.cs], gp[2 * a.g.cs]};
  if (a.dist_kind == 1) {
    const float nrm = a.l2norm[b];
    // torch: d sqrt(s)/ds = 1/(2 sqrt(s)), ds/dp = 2 (p - o)  ->  (p-o)/norm ; weight/B from the batch mean
    const float c = a.w[b] / (float)a.B;
    g[0] += c * ((px - ox) / nrm);
    g[1] += c * ((py - oy) / nrm);
    g[2] += c * ((pz - oz) / nrm);
  } else if (a.dist_kind == 2) {
    const int j = a.nn_idx[(int64_t)b * a.K + k];
    const float* q = a.ori.p + (int64_t)b * a.ori.bs + (int64_t)j * a.ori.ps;
    const float c = 2.f * (a.w[b] / (float)a.B) / (float)a.K;
    g[0] += c * (px - q[0]);
    g[1] += c * (py - q[a.ori.cs]);
    g[2] += c * (pz - q[2 * a.ori.cs]);
  }
  float np_[3];
  const float pin[3] = {px, py, pz};
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float m = mp[c * a.m.cs], v = vp[c * a.v.cs];
    m = m + (g[c] - m) * omb1;
    v = v * fb2 + omb2 * g[c] * g[c];
    mp[c * a.m.cs] = m;
    vp[c * a.v.cs] = v;
    const float denom = __builtin_sqrtf(v) / bc2s + a.eps;
    np_[c] = pin[c] - step_size * (m / denom);
  }
  float dx = np_[0] - ox, dy = np_[1] - oy, dz = np_[2] - oz;
  if (a.budget > 0.f) {  // ClipPointsLinf (clip_utils.py:43-56)
    const float norm = __builtin_sqrtf(dx * dx + dy * dy + dz * dz);
    const float s = fminf(a.budget / (norm + 1e-9f), 1.f);
    dx *= s, dy *= s, dz *= s;
  }
  pp[0] = ox + dx;
  pp[a.p.cs] = oy + dy;
  pp[2 * a.p.cs] = oz + dz;
}

// ---------------------------------------------------------------------------------------------------------
// Bookkeeping + update merged: one workgroup per sample first reduces ||adv-ori||, updates the best-attack state
// and copies the CURRENT iterate where required (exactly cw_bookkeep_kernel), then applies cw_step_kernel's
// gradient-assembly + Adam + clip to its own points. The step word is only READ here (the classifier-tail launch
// advances it earlier in the iteration), so no workgroup can observe a half-updated counter.
// ---------------------------------------------------------------------------------------------------------
struct UpdateArgs {
  BookArgs bk;
  StepArgs st;
};

// PER points per thread, all operands fetched up front (one memory round trip), then reduce -> decide -> update.
template <int PER>
__global__ __launch_bounds__(1024) void cw_update_kernel(UpdateArgs u) {
  __shared__ float part[16];
  __shared__ int s_copy;
  __shared__ float s_dist, s_step_size, s_bc2s;
  const BookArgs& a = u.bk;
  const StepArgs& s = u.st;
  const int b = blockIdx.x;
  const int tid = threadIdx.x;
  float px[PER], py[PER], pz[PER], ox[PER], oy[PER], oz[PER], gx[PER], gy[PER], gz[PER];
  float m_[PER][3], v_[PER][3], qx[PER], qy[PER], qz[PER];
  int nj[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int k = tid + i * 1024;
    nj[i] = (s.dist_kind == 2 && k < a.K) ? s.nn_idx[(int64_t)b * s.K + k] : 0;
  }
  const int t = s.step_dev ? s.step_dev[0] : s.step_host;
  const int64_t pr = a.pred[b], lb = a.label[b];
  const float bd = a.bestdist[b], obd = a.o_bestdist[b];
  const float wb = s.w ? s.w[b] : 0.f;
  float acc = 0.f;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int k = tid + i * 1024;
    if (k < a.K) {
      const float* p = s.p.p + (int64_t)b * s.p.bs + (int64_t)k * s.p.ps;
      const float* o = s.ori.p + (int64_t)b * s.ori.bs + (int64_t)k * s.ori.ps;
      const float* gp = s.g.p + (int64_t)b * s.g.bs + (int64_t)k * s.g.ps;
      const float* mp = s.m.p + (int64_t)b * s.m.bs + (int64_t)k * s.m.ps;
      const float* vp = s.v.p + (int64_t)b * s.v.bs + (int64_t)k * s.v.ps;
      px[i] = p[0], py[i] = p[s.p.cs], pz[i] = p[2 * s.p.cs];
      ox[i] = o[0], oy[i] = o[s.ori.cs], oz[i] = o[2 * s.ori.cs];
      gx[i] = gp[0], gy[i] = gp[s.g.cs], gz[i] = gp[2 * s.g.cs];
#pragma unroll
      for (int c = 0; c < 3; ++c) m_[i][c] = mp[c * s.m.cs], v_[i][c] = vp[c * s.v.cs];
    } else {
      px[i] = py[i] = pz[i] = ox[i] = oy[i] = oz[i] = gx[i] = gy[i] = gz[i] = 0.f;
#pragma unroll
      for (int c = 0; c < 3; ++c) m_[i][c] = v_[i][c] = 0.f;
    }
  }
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const float* q = s.ori.p + (int64_t)b * s.ori.bs + (int64_t)nj[i] * s.ori.ps;
    if (s.dist_kind == 2) qx[i] = q[0], qy[i] = q[s.ori.cs], qz[i] = q[2 * s.ori.cs];
    else qx[i] = qy[i] = qz[i] = 0.f;
  }
  // the reduction tree differs from cw_bookkeep_kernel's (16 waves of 1024-strided partial sums instead of 4 waves of
  // 256-strided ones), so ||adv-ori|| may differ from the two-launch path in the last ulp
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const float dx = px[i] - ox[i], dy = py[i] - oy[i], dz = pz[i] - oz[i];
    acc += dx * dx + dy * dy + dz * dz;
  }
  acc = wave_sum(acc);
  if ((tid & 63) == 0) part[tid >> 6] = acc;
  __syncthreads();
  if (tid == 0) {
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) tot += part[w];
    const float dist = __builtin_sqrtf(tot);
    s_dist = dist;
    if (a.dist_val) a.dist_val[b] = dist;
    const bool succ = a.untarget ? (pr != lb) : (pr == lb);
    if (succ && dist < bd) {
      a.bestdist[b] = dist;
      a.bestscore[b] = pr;
    }
    int copy = 0;
    if (succ && dist < obd) {
      a.o_bestdist[b] = dist;
      a.o_bestscore[b] = pr;
      copy = 1;
    }
    s_copy = copy;
    // Adam bias corrections in double (as torch): evaluated once per workgroup, not by all 1024 threads
    s_step_size = (float)(s.lr / (1.0 - pow(s.b1, (double)t)));
    s_bc2s = (float)sqrt(1.0 - pow(s.b2, (double)t));
  }
  __syncthreads();
  const bool copy = s_copy != 0;
  const float l2n = s_dist;
  const float omb1 = (float)(1.0 - s.b1), omb2 = (float)(1.0 - s.b2), fb2 = (float)s.b2;
  const float step_size = s_step_size, bc2s = s_bc2s;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int k = tid + i * 1024;
    if (k >= a.K) continue;
    if (a.input_val.p) {
      float* q = a.input_val.p + (int64_t)b * a.input_val.bs + (int64_t)k * a.input_val.ps;
      q[0] = px[i], q[a.input_val.cs] = py[i], q[2 * a.input_val.cs] = pz[i];
    }
    if (copy) {
      float* q = a.o_bestattack.p + (int64_t)b * a.o_bestattack.bs + (int64_t)k * a.o_bestattack.ps;
      q[0] = px[i], q[a.o_bestattack.cs] = py[i], q[2 * a.o_bestattack.cs] = pz[i];
    }
    float g[3] = {gx[i], gy[i], gz[i]};
    if (s.dist_kind == 1) {
      const float c = wb / (float)s.B;
      g[0] += c * ((px[i] - ox[i]) / l2n);
      g[1] += c * ((py[i] - oy[i]) / l2n);
      g[2] += c * ((pz[i] - oz[i]) / l2n);
    } else if (s.dist_kind == 2) {
      const float c = 2.f * (wb / (float)s.B) / (float)s.K;
      g[0] += c * (px[i] - qx[i]);
      g[1] += c * (py[i] - qy[i]);
      g[2] += c * (pz[i] - qz[i]);
    }
    float* pp = s.p.p + (int64_t)b * s.p.bs + (int64_t)k * s.p.ps;
    float* mp = s.m.p + (int64_t)b * s.m.bs + (int64_t)k * s.m.ps;
    float* vp = s.v.p + (int64_t)b * s.v.bs + (int64_t)k * s.v.ps;
    float np_[3];
    const float pin[3] = {px[i], py[i], pz[i]};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float m = m_[i][c], v = v_[i][c];
      m = m + (g[c] - m) * omb1;
      v = v * fb2 + omb2 * g[c] * g[c];
      mp[c * s.m.cs] = m;
      vp[c * s.v.cs] = v;
      const float denom = __builtin_sqrtf(v) / bc2s + s.eps;
      np_[c] = pin[c] - step_size * (m / denom);
    }
    float dx = np_[0] - ox[i], dy = np_[1] - oy[i], dz = np_[2] - oz[i];
    if (s.budget > 0.f) {
      const float norm = __builtin_sqrtf(dx * dx + dy * dy + dz * dz);
      const float sc = fminf(s.budget / (norm + 1e-9f), 1.f);
      dx *= sc, dy *= sc, dz *= sc;
    }
    pp[0] = ox[i] + dx;
    pp[s.p.cs] = oy[i] + dy;
    pp[2 * s.p.cs] = oz[i] + dz;
  }
}

}  // namespace pc3d

using namespace pc3d;

extern "C" int pc3d_cw_update_f32(float* adv, int64_t a_bs, int64_t a_ps, int64_t a_cs,
                                  const float* ori, int64_t o_bs, int64_t o_ps, int64_t o_cs, int B, int K,
                                  const int64_t* pred, const int64_t* label, int untarget,
                                  float* bestdist, int64_t* bestscore, float* o_bestdist, int64_t* o_bestscore,
                                  float* o_bestattack, float* input_val, float* dist_val,
                                  const float* g, int64_t g_bs, int64_t g_ps, int64_t g_cs, float* m, float* v,
                                  double lr, double beta1, double beta2, double eps, float budget,
                                  const int32_t* step_dev, int step_host, int dist_kind, const float* w,
                                  const int32_t* nn_idx, void* stream) {
  PC3D_REQUIRE(B >= 0 && K >= 1 && K <= 8192, "pc3d_cw_update_f32: bad sizes B=%d K=%d (K <= 8192)", B, K);
  PC3D_REQUIRE(step_dev != nullptr || step_host >= 1, "pc3d_cw_update_f32: step_host must be >= 1 without a device counter");
  PC3D_REQUIRE(dist_kind >= 0 && dist_kind <= 2, "pc3d_cw_update_f32: dist_kind=%d not in {0,1,2}", dist_kind);
  if (B == 0) return PC3D_OK;
  PC3D_REQUIRE(adv && ori && pred && label && bestdist && bestscore && o_bestdist && o_bestscore && o_bestattack && g && m && v,
               "pc3d_cw_update_f32: null pointer");
  PC3D_REQUIRE(dist_kind == 0 || w != nullptr, "pc3d_cw_update_f32: distance term needs the weights w");
  PC3D_REQUIRE(dist_kind != 2 || nn_idx != nullptr, "pc3d_cw_update_f32: Chamfer term needs nn_idx");
  UpdateArgs u{};
  // o_bestattack / input_val / m / v share adv's layout
  u.bk = BookArgs{{adv, a_bs, a_ps, a_cs}, {ori, o_bs, o_ps, o_cs}, K, pred, label, untarget, bestdist, bestscore,
                  o_bestdist, o_bestscore, {o_bestattack, a_bs, a_ps, a_cs}, {input_val, a_bs, a_ps, a_cs}, dist_val,
                  nullptr};
  u.st = StepArgs{{adv, a_bs, a_ps, a_cs}, {g, g_bs, g_ps, g_cs}, {m, a_bs, a_ps, a_cs}, {v, a_bs, a_ps, a_cs},
                  {ori, o_bs, o_ps, o_cs}, K, B, lr, beta1, beta2, (float)eps, budget, step_dev, step_host, dist_kind, w,
                  nullptr, nn_idx};
  if (K <= 1024) hipLaunchKernelGGL(cw_update_kernel<1>, dim3(B), dim3(1024), 0, as_stream(stream), u);
  else if (K <= 2048) hipLaunchKernelGGL(cw_update_kernel<2>, dim3(B), dim3(1024), 0, as_stream(stream), u);
  else if (K <= 4096) hipLaunchKernelGGL(cw_update_kernel<4>, dim3(B), dim3(1024), 0, as_stream(stream), u);
  else hipLaunchKernelGGL(cw_update_kernel<8>, dim3(B), dim3(1024), 0, as_stream(stream), u);
  PC3D_LAUNCH_CHECK("pc3d_cw_update_f32");
  return PC3D_OK;
}

extern "C" int pc3d_cw_bookkeep_f32(const float* adv, int64_t a_bs, int64_t a_ps, int64_t a_cs,
                                    const float* ori, int64_t o_bs, int64_t o_ps, int64_t o_cs, int B, int K,
                                    const int64_t* pred, const int64_t* label, int untarget,
                                    float* bestdist, int64_t* bestscore, float* o_bestdist, int64_t* o_bestscore,
                                    float* o_bestattack, int64_t ba_bs, int64_t ba_ps, int64_t ba_cs,
                                    float* input_val, int64_t iv_bs, int64_t iv_ps, int64_t iv_cs,
                                    float* dist_val, int32_t* step, void* stream) {
  PC3D_REQUIRE(B >= 0 && K >= 1, "pc3d_cw_bookkeep_f32: bad sizes B=%d K=%d", B, K);
  if (B == 0) return PC3D_OK;
  PC3D_REQUIRE(adv && ori && pred && label && bestdist && bestscore && o_bestdist && o_bestscore && o_bestattack,
               "pc3d_cw_bookkeep_f32: null pointer");
  BookArgs a{{adv, a_bs, a_ps, a_cs}, {ori, o_bs, o_ps, o_cs}, K, pred, label, untarget, bestdist, bestscore,
             o_bestdist, o_bestscore, {o_bestattack, ba_bs, ba_ps, ba_cs}, {input_val, iv_bs, iv_ps, iv_cs},
             dist_val, step};
  hipLaunchKernelGGL(cw_bookkeep_kernel, dim3(B), dim3(256), 0, as_stream(stream), a);
  PC3D_LAUNCH_CHECK("pc3d_cw_bookkeep_f32");
  return PC3D_OK;
}

extern "C" int pc3d_cw_step_f32(float* p, int64_t p_bs, int64_t p_ps, int64_t p_cs,
                                const float* g, int64_t g_bs, int64_t g_ps, int64_t g_cs, float* m, float* v,
                                const float* ori, int64_t o_bs, int64_t o_ps, int64_t o_cs, int B, int K,
                                double lr, double beta1, double beta2, double eps, float budget,
                                const int32_t* step_dev, int step_host, int dist_kind, const float* w,
                                const float* l2norm, const int32_t* nn_idx, void* stream) {
  PC3D_REQUIRE(B >= 0 && K >= 1 && B <= 65535, "pc3d_cw_step_f32: bad sizes B=%d K=%d", B, K);
  PC3D_REQUIRE(step_dev != nullptr || step_host >= 1, "pc3d_cw_step_f32: step_host must be >= 1 without a device counter");
  PC3D_REQUIRE(dist_kind >= 0 && dist_kind <= 2, "pc3d_cw_step_f32: dist_kind=%d not in {0,1,2}", dist_kind);
  if (B == 0) return PC3D_OK;
  PC3D_REQUIRE(p && g && m && v && ori, "pc3d_cw_step_f32: null pointer");
  PC3D_REQUIRE(dist_kind == 0 || w != nullptr, "pc3d_cw_step_f32: distance term needs the weights w");
  PC3D_REQUIRE(dist_kind != 1 || l2norm != nullptr, "pc3d_cw_step_f32: L2 term needs l2norm");
  PC3D_REQUIRE(dist_kind != 2 || nn_idx != nullptr, "pc3d_cw_step_f32: Chamfer term needs nn_idx");
  StepArgs a{{p, p_bs, p_ps, p_cs}, {g, g_bs, g_ps, g_cs}, {m, p_bs, p_ps, p_cs}, {v, p_bs, p_ps, p_cs},
             {ori, o_bs, o_ps, o_cs}, K, B, lr, beta1, beta2, (float)eps, budget, step_dev, step_host, dist_kind, w,
             l2norm, nn_idx};
  hipLaunchKernelGGL(cw_step_kernel, dim3(cdiv(K, 256), B), dim3(256), 0, as_stream(stream), a);
  PC3D_LAUNCH_CHECK("pc3d_cw_step_f32");
  return PC3D_OK;
}
