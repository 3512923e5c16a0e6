// K9/K10 + a1/a7 — the HBM-bound elementwise pieces of the attack iteration, each one launch:
//   * clip / inner-point projection of the perturbation   (attack/CW/CW_utils/clip_utils.py:5-136)
//   * Adam step on the adversarial points, optionally fused with the clip (torch.optim.Adam as used at
//     attack/CW/CW_attack.py:100,169; attack/KNN/KNN_attack.py:87,129)
//   * dense pairwise (squared) distance matrix for the callers that really want [B,N,M]
//     (attack/CW/CW_utils/distance.py:15-32, utils/dis_utils_torch.py:8-11, utils/dis_utils_numpy.py:13-20)
// One thread owns one POINT (3 coordinates), so the per-point norm/cross products need no communication.
#include "pc3d_common.h"

namespace pc3d {

struct ClipArgs {
  PtsView pc, ori, normal;  // normal.p == null -> no projection
  PtsViewMut out;
  int K;
  float budget;  // per-point L2 budget ("Linf" in the reference's naming, SURVEY A-10); <= 0 -> no clip
};

// ProjectInnerPoints (clip_utils.py:67-108) then ClipPointsLinf (:43-56) on one point's perturbation.
__device__ __forceinline__ void project_clip(float& dx, float& dy, float& dz, bool has_normal, float nx, float ny,
                                             float nz, float budget) {
  if (has_normal) {
    const float inner = dx * nx + dy * ny + dz * nz;
    if (inner < 0.f) {
      // vng = n x d ; vref = vng x n
      const float gx = ny * dz - nz * dy, gy = nz * dx - nx * dz, gz = nx * dy - ny * dx;
      const float gnorm = __builtin_sqrtf(gx * gx + gy * gy + gz * gz);
      const float rx = gy * nz - gz * ny, ry = gz * nx - gx * nz, rz = gx * ny - gy * nx;
      const float rnorm = __builtin_sqrtf(rx * rx + ry * ry + rz * rz);
      // reference: diff_proj = diff * vref / (|vref| + 1e-9)   (elementwise product, kept as written)
      const float den = rnorm + 1e-9f;
      float px = dx * rx / den, py = dy * ry / den, pz = dz * rz / den;
      if (gnorm < 1e-6f) px = py = pz = 0.f;
      dx = px, dy = py, dz = pz;
    }
  }
  if (budget > 0.f) {
    const float norm = __builtin_sqrtf(dx * dx + dy * dy + dz * dz);
    float s = budget / (norm + 1e-9f);
    s = fminf(s, 1.f);
    dx *= s, dy *= s, dz *= s;
  }
}

__global__ __launch_bounds__(256) void clip_kernel(ClipArgs a) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  if (k >= a.K) return;
  const float* p = a.pc.p + (int64_t)b * a.pc.bs + (int64_t)k * a.pc.ps;
  const float* o = a.ori.p + (int64_t)b * a.ori.bs + (int64_t)k * a.ori.ps;
  const float ox = o[0], oy = o[a.ori.cs], oz = o[2 * a.ori.cs];
  float dx = p[0] - ox, dy = p[a.pc.cs] - oy, dz = p[2 * a.pc.cs] - oz;
  float nx = 0.f, ny = 0.f, nz = 0.f;
  const bool hn = a.normal.p != nullptr;
  if (hn) {
    const float* n = a.normal.p + (int64_t)b * a.normal.bs + (int64_t)k * a.normal.ps;
    nx = n[0], ny = n[a.normal.cs], nz = n[2 * a.normal.cs];
  }
  project_clip(dx, dy, dz, hn, nx, ny, nz, a.budget);
  float* q = a.out.p + (int64_t)b * a.out.bs + (int64_t)k * a.out.ps;
  q[0] = ox + dx;
  q[a.out.cs] = oy + dy;
  q[2 * a.out.cs] = oz + dz;
}

// global-L2 clip (clip_utils.py:16-29): one workgroup per sample computes |pc-ori|_F, then rescales.
__global__ __launch_bounds__(256) void clip_l2_kernel(ClipArgs a) {
  __shared__ float part[4];
  const int b = blockIdx.x;
  float acc = 0.f;
  for (int k = threadIdx.x; k < a.K; k += 256) {
    const float* p = a.pc.p + (int64_t)b * a.pc.bs + (int64_t)k * a.pc.ps;
    const float* o = a.ori.p + (int64_t)b * a.ori.bs + (int64_t)k * a.ori.ps;
    const float dx = p[0] - o[0], dy = p[a.pc.cs] - o[a.ori.cs], dz = p[2 * a.pc.cs] - o[2 * a.ori.cs];
    acc += dx * dx + dy * dy + dz * dz;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  const float norm = __builtin_sqrtf(part[0] + part[1] + part[2] + part[3]);
  const float s = fminf(a.budget / (norm + 1e-9f), 1.f);
  for (int k = threadIdx.x; k < a.K; k += 256) {
    const float* p = a.pc.p + (int64_t)b * a.pc.bs + (int64_t)k * a.pc.ps;
    const float* o = a.ori.p + (int64_t)b * a.ori.bs + (int64_t)k * a.ori.ps;
    float* q = a.out.p + (int64_t)b * a.out.bs + (int64_t)k * a.out.ps;
    const float ox = o[0], oy = o[a.ori.cs], oz = o[2 * a.ori.cs];
    q[0] = ox + (p[0] - ox) * s;
    q[a.out.cs] = oy + (p[a.pc.cs] - oy) * s;
    q[2 * a.out.cs] = oz + (p[2 * a.pc.cs] - oz) * s;
  }
}

// ---------------------------------------------------------------------------------------------------------
// Adam (+ optional clip) on the point tensor. torch.optim.Adam single-tensor semantics, weight_decay 0:
//   m += (g - m)(1-b1);  v = v*b2 + (1-b2) g*g;  denom = sqrt(v)/sqrt(1-b2^t) + eps;  p -= lr/(1-b1^t) * m/denom
// Scalars follow torch's python-double arithmetic (1-beta, 1-beta**t, lr/bc1 in double, rounded to fp32 once).
// The step number t is either passed by the host or read from a device counter (so the whole iteration can be
// replayed from a hipGraph); the counter is advanced by a separate launch (pc3d_i32_add / the bookkeeping
// kernel), never by this one.
// ---------------------------------------------------------------------------------------------------------
struct AdamArgs {
  PtsViewMut p;        // parameters (adv points), updated in place
  PtsView g;           // gradient
  PtsView g2;          // second gradient summed with g on load (the two branches of an attack's loss), p null -> none
  PtsViewMut m, v;     // exp_avg, exp_avg_sq
  PtsView ori, normal; // clip against ori (ori.p null -> no clip/projection); normal optional
  int K;
  double lr, b1, b2;
  float eps, budget;
  const int* step_dev;  // device step number t (>= 1); null -> use step_host
  int step_host;
};

__global__ __launch_bounds__(256) void adam_clip_kernel(AdamArgs a) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  if (k >= a.K) return;
  const int t = a.step_dev ? a.step_dev[0] : a.step_host;
  const float omb1 = (float)(1.0 - a.b1), omb2 = (float)(1.0 - a.b2), fb2 = (float)a.b2;
  const float step_size = (float)(a.lr / (1.0 - pow(a.b1, (double)t)));
  const float bc2s = (float)sqrt(1.0 - pow(a.b2, (double)t));
  float* pp = a.p.p + (int64_t)b * a.p.bs + (int64_t)k * a.p.ps;
  const float* gp = a.g.p + (int64_t)b * a.g.bs + (int64_t)k * a.g.ps;
  const float* gq = a.g2.p ? a.g2.p + (int64_t)b * a.g2.bs + (int64_t)k * a.g2.ps : nullptr;
  float* mp = a.m.p + (int64_t)b * a.m.bs + (int64_t)k * a.m.ps;
  float* vp = a.v.p + (int64_t)b * a.v.bs + (int64_t)k * a.v.ps;
  float np_[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float g = gp[c * a.g.cs];
    if (gq) g += gq[c * a.g2.cs];
    float m = mp[c * a.m.cs], v = vp[c * a.v.cs];
    m = m + (g - m) * omb1;
    v = v * fb2 + omb2 * g * g;
    mp[c * a.m.cs] = m;
    vp[c * a.v.cs] = v;
    const float denom = __builtin_sqrtf(v) / bc2s + a.eps;
    np_[c] = pp[c * a.p.cs] - step_size * (m / denom);
  }
  if (a.ori.p) {
    const float* o = a.ori.p + (int64_t)b * a.ori.bs + (int64_t)k * a.ori.ps;
    const float ox = o[0], oy = o[a.ori.cs], oz = o[2 * a.ori.cs];
    float dx = np_[0] - ox, dy = np_[1] - oy, dz = np_[2] - oz;
    float nx = 0.f, ny = 0.f, nz = 0.f;
    const bool hn = a.normal.p != nullptr;
    if (hn) {
      const float* n = a.normal.p + (int64_t)b * a.normal.bs + (int64_t)k * a.normal.ps;
      nx = n[0], ny = n[a.normal.cs], nz = n[2 * a.normal.cs];
    }
    project_clip(dx, dy, dz, hn, nx, ny, nz, a.budget);
    np_[0] = ox + dx, np_[1] = oy + dy, np_[2] = oz + dz;
  }
  pp[0] = np_[0];
  pp[a.p.cs] = np_[1];
  pp[2 * a.p.cs] = np_[2];
}

// ---------------------------------------------------------------------------------------------------------
// Dense pairwise distances out[b,i,j] = |x_i - y_j|^2 (mode 0) or its sqrt (mode 1). Write-bound: each thread
// produces 4 consecutive j for one i (16-B stores when M % 4 == 0).
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pairwise_kernel(PtsView x, PtsView y, int N, int M, int mode, float* out) {
  const int j0 = (blockIdx.x * 256 + threadIdx.x) * 4;
  const int i = blockIdx.y, b = blockIdx.z;
  if (j0 >= M) return;
  const float* xp = x.p + (int64_t)b * x.bs + (int64_t)i * x.ps;
  const float qx = xp[0], qy = xp[x.cs], qz = xp[2 * x.cs];
  float d[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int j = (j0 + e < M) ? j0 + e : M - 1;
    const float* yp = y.p + (int64_t)b * y.bs + (int64_t)j * y.ps;
    const float dx = qx - yp[0], dy = qy - yp[y.cs], dz = qz - yp[2 * y.cs];
    float v = dx * dx;
    v = __builtin_fmaf(dy, dy, v);
    v = __builtin_fmaf(dz, dz, v);
    d[e] = mode ? __builtin_sqrtf(v) : v;
  }
  float* o = out + ((int64_t)b * N + i) * M + j0;
  if ((M & 3) == 0) {
    *reinterpret_cast<float4*>(o) = make_float4(d[0], d[1], d[2], d[3]);
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (j0 + e < M) o[e] = d[e];
  }
}

}  // namespace pc3d

using namespace pc3d;

extern "C" int pc3d_clip_f32(const float* pc, int64_t pc_bs, int64_t pc_ps, int64_t pc_cs,
                             const float* ori, int64_t o_bs, int64_t o_ps, int64_t o_cs,
                             const float* normal, int64_t n_bs, int64_t n_ps, int64_t n_cs,
                             int B, int K, int mode, float budget,
                             float* out, int64_t out_bs, int64_t out_ps, int64_t out_cs, void* stream) {
  PC3D_REQUIRE(B >= 0 && K >= 1 && B <= 65535, "pc3d_clip_f32: bad sizes B=%d K=%d", B, K);
  PC3D_REQUIRE(mode == 0 || mode == 1, "pc3d_clip_f32: mode must be 0 (per-point) or 1 (global L2), got %d", mode);
  if (B == 0) return PC3D_OK;
  PC3D_REQUIRE(pc && ori && out, "pc3d_clip_f32: null pointer");
  ClipArgs a{{pc, pc_bs, pc_ps, pc_cs}, {ori, o_bs, o_ps, o_cs}, {normal, n_bs, n_ps, n_cs},
             {out, out_bs, out_ps, out_cs}, K, budget};
  if (mode == 0)
    hipLaunchKernelGGL(clip_kernel, dim3(cdiv(K, 256), B), dim3(256), 0, as_stream(stream), a);
  else
    hipLaunchKernelGGL(clip_l2_kernel, dim3(B), dim3(256), 0, as_stream(stream), a);
  PC3D_LAUNCH_CHECK("pc3d_clip_f32");
  return PC3D_OK;
}

extern "C" int pc3d_adam_clip_step_f32(float* p, int64_t p_bs, int64_t p_ps, int64_t p_cs,
                                       const float* g, int64_t g_bs, int64_t g_ps, int64_t g_cs,
                                       const float* g2, int64_t g2_bs, int64_t g2_ps, int64_t g2_cs,
                                       float* m, float* v, /* same strides as p */
                                       const float* ori, int64_t o_bs, int64_t o_ps, int64_t o_cs,
                                       const float* normal, int64_t n_bs, int64_t n_ps, int64_t n_cs,
                                       int B, int K, double lr, double beta1, double beta2, double eps, float budget,
                                       const int32_t* step_dev, int step_host, void* stream) {
  PC3D_REQUIRE(B >= 0 && K >= 1 && B <= 65535, "pc3d_adam_clip_step_f32: bad sizes B=%d K=%d", B, K);
  PC3D_REQUIRE(step_dev != nullptr || step_host >= 1, "pc3d_adam_clip_step_f32: step_host must be >= 1 without a device counter");
  if (B == 0) return PC3D_OK;
  PC3D_REQUIRE(p && g && m && v, "pc3d_adam_clip_step_f32: null pointer");
  AdamArgs a{{p, p_bs, p_ps, p_cs}, {g, g_bs, g_ps, g_cs}, {g2, g2_bs, g2_ps, g2_cs}, {m, p_bs, p_ps, p_cs}, {v, p_bs, p_ps, p_cs},
             {ori, o_bs, o_ps, o_cs}, {normal, n_bs, n_ps, n_cs}, K, lr, beta1, beta2, (float)eps, budget,
             step_dev, step_host};
  hipLaunchKernelGGL(adam_clip_kernel, dim3(cdiv(K, 256), B), dim3(256), 0, as_stream(stream), a);
  PC3D_LAUNCH_CHECK("pc3d_adam_clip_step_f32");
  return PC3D_OK;
}

__global__ void i32_add_kernel(int* p, int delta) { p[0] += delta; }

extern "C" int pc3d_i32_add(int32_t* ctr, int delta, void* stream) {
  PC3D_REQUIRE(ctr != nullptr, "pc3d_i32_add: null pointer");
  hipLaunchKernelGGL(i32_add_kernel, dim3(1), dim3(1), 0, as_stream(stream), ctr, delta);
  PC3D_LAUNCH_CHECK("pc3d_i32_add");
  return PC3D_OK;
}

extern "C" int pc3d_pairwise_f32(const float* x, int64_t x_bs, int64_t x_ps, int64_t x_cs,
                                 const float* y, int64_t y_bs, int64_t y_ps, int64_t y_cs,
                                 int B, int N, int M, int mode, float* out, void* stream) {
  PC3D_REQUIRE(B >= 0 && N >= 1 && M >= 1, "pc3d_pairwise_f32: bad sizes B=%d N=%d M=%d", B, N, M);
  PC3D_REQUIRE(N <= 65535 && B <= 65535, "pc3d_pairwise_f32: N=%d / B=%d exceed grid limits", N, B);
  PC3D_REQUIRE(mode == 0 || mode == 1, "pc3d_pairwise_f32: mode must be 0 (squared) or 1 (euclidean)");
  if (B == 0) return PC3D_OK;
  PC3D_REQUIRE(x && y && out, "pc3d_pairwise_f32: null pointer");
  hipLaunchKernelGGL(pairwise_kernel, dim3(cdiv(cdiv(M, 4), 256), N, B), dim3(256), 0, as_stream(stream),
                     PtsView{x, x_bs, x_ps, x_cs}, PtsView{y, y_bs, y_ps, y_cs}, N, M, mode, out);
  PC3D_LAUNCH_CHECK("pc3d_pairwise_f32");
  return PC3D_OK;
}
