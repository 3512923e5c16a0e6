// GeoA3's per-sample loss assembly (attack/GeoA3/GeoA3_attack.py:139-181 with loss_utils.py:36-58,92-105), gfx950:
//   dis   = mean_i d_ao[i] (+ mean_j d_oa[j])                 Chamfer (pseudo-Chamfer when d_oa is absent)
//   hd    = max_i d_ao[i]                                      one-sided Hausdorff
//   curv  = mean_i (kappa_adv[i] - kappa_ori[idx_ao[i]])^2     curvature consistency
//   constrain = w_dis dis + w_hd hd + w_curv curv ;  loss_n = cls + scale constrain
// from the nearest-neighbour distances / indices and curvature proxies the search kernels produced. The reference (and
// the round-1 mirror) assembles this from ~15 ATen reductions and scalings per step and ~20 more in the backward; here it
// is one launch each way: a workgroup per sample reduces over the points, the backward is element-wise.
#include "pc3d_common.h"

namespace pc3d {

struct GeoTermsArgs {
  const float* d_ao;      // [B,N]
  const float* d_oa;      // [B,M] or null
  const float* k_adv;     // [B,N] or null
  const float* k_ori;     // [B,M] (with idx_ao) or null
  const int64_t* idx_ao;  // [B,N]
  const float* cls;       // [B]
  const float* scale;     // [B]
  int N, M;
  float w_dis, w_hd, w_curv;
  float* out;             // [5,B]: dis, hd, curv, constrain, loss_n
  int32_t* hd_arg;        // [B]
  // backward
  const float* g_out;     // [5,B] upstream gradients on the five outputs
  float* g_d_ao;          // [B,N]
  float* g_d_oa;          // [B,M] or null
  float* g_k_adv;         // [B,N] or null
  float* g_cls;           // [B]
};

__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void geoa3_terms_fwd_kernel(GeoTermsArgs a, int B) {
  __shared__ float red[4];
  __shared__ float redv[4];
  __shared__ int redi[4];
  const int b = blockIdx.x;
  const float* dao = a.d_ao + (int64_t)b * a.N;
  float s_ao = 0.f, s_cv = 0.f, mx = -__builtin_inff();
  int mi = 0x7fffffff;
  for (int i = threadIdx.x; i < a.N; i += 256) {
    const float d = dao[i];
    s_ao += d;
    if (d > mx) mx = d, mi = i;                    // ascending i within a thread: strict > keeps the first maximum
    if (a.k_adv) {
      const float t = a.k_adv[(int64_t)b * a.N + i] - a.k_ori[(int64_t)b * a.M + a.idx_ao[(int64_t)b * a.N + i]];
      s_cv += t * t;
    }
  }
  float s_oa = 0.f;
  if (a.d_oa)
    for (int j = threadIdx.x; j < a.M; j += 256) s_oa += a.d_oa[(int64_t)b * a.M + j];
  s_ao = block_sum(s_ao, red);
  s_oa = block_sum(s_oa, red);
  s_cv = block_sum(s_cv, red);
  // arg-max over the workgroup: (value, lowest index)
  {
    const float wv = wave_max(mx);
    int cand = (mx == wv) ? mi : 0x7fffffff;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cand = min(cand, __shfl_xor(cand, o, 64));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) redv[wave] = wv, redi[wave] = cand;
    __syncthreads();
    mx = redv[0], mi = redi[0];
#pragma unroll
    for (int w = 1; w < 4; ++w)
      if (redv[w] > mx || (redv[w] == mx && redi[w] < mi)) mx = redv[w], mi = redi[w];
  }
  if (threadIdx.x == 0) {
    const float dis = s_ao / (float)a.N + (a.d_oa ? s_oa / (float)a.M : 0.f);
    const float hd = mx;
    const float curv = a.k_adv ? s_cv / (float)a.N : 0.f;
    const float con = (a.w_dis * dis + a.w_hd * hd) + a.w_curv * curv;
    a.out[b] = dis, a.out[B + b] = hd, a.out[2 * B + b] = curv, a.out[3 * B + b] = con;
    a.out[4 * B + b] = a.cls[b] + a.scale[b] * con;
    a.hd_arg[b] = mi == 0x7fffffff ? 0 : mi;
  }
}

__global__ __launch_bounds__(256) void geoa3_terms_bwd_kernel(GeoTermsArgs a, int B) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  const float g_ln = a.g_out[4 * B + b];
  const float g_con = a.g_out[3 * B + b] + g_ln * a.scale[b];
  const float g_dis = a.g_out[b] + a.w_dis * g_con;
  const float g_hd = a.g_out[B + b] + a.w_hd * g_con;
  const float g_cv = a.g_out[2 * B + b] + a.w_curv * g_con;
  if (i == 0) a.g_cls[b] = g_ln;
  if (i < a.N) {
    a.g_d_ao[(int64_t)b * a.N + i] = g_dis / (float)a.N + (i == a.hd_arg[b] ? g_hd : 0.f);
    if (a.g_k_adv) {
      const float t = a.k_adv[(int64_t)b * a.N + i] - a.k_ori[(int64_t)b * a.M + a.idx_ao[(int64_t)b * a.N + i]];
      a.g_k_adv[(int64_t)b * a.N + i] = g_cv * 2.f * t / (float)a.N;
    }
  }
  if (a.g_d_oa && i < a.M) a.g_d_oa[(int64_t)b * a.M + i] = g_dis / (float)a.M;
}

// Best-attack bookkeeping of one GeoA3 iteration (attack/GeoA3/GeoA3_attack.py:307-330): predicted label of the iterate
// (first arg-max of the logits, NaN counts as the maximum like torch.argmax), success test, and the conditional updates
// of the running bests — ~25 ATen launches (argmax, compares, full_like, where x6 incl. one over the whole cloud) as one.
struct GeoRecordArgs {
  const float* logits;    // [B,ncls], row stride ld
  int ld, ncls;
  const int64_t* target;  // [B] label compared against (the attack target when targeted, else the ground truth)
  int targeted;
  const float* metric;    // [B]
  const float* iterate;   // [B,3N] contiguous
  int n3;
  int64_t search_step, step;
  float* best_loss;       // [B]
  float* best_attack;     // [B,3N]
  int64_t* best_bs;       // [B]
  int64_t* best_step;     // [B]
  float* iter_best_loss;  // [B]
  int64_t* iter_best_score;  // [B]
  int64_t* label_out;     // [B]
};

__global__ __launch_bounds__(256) void geoa3_record_kernel(GeoRecordArgs a) {
  __shared__ int s_upd;
  const int b = blockIdx.x;
  if (threadIdx.x < 64) {
    const float* lg = a.logits + (int64_t)b * a.ld;
    float bv = -__builtin_inff();
    int bi = 0x7fffffff;
    bool bnan = false;
    for (int c = threadIdx.x; c < a.ncls; c += 64) {   // ascending c within a lane: strict > keeps the first maximum
      const float v = lg[c];
      const bool vnan = v != v;
      if (bi == 0x7fffffff || (!bnan && (vnan || v > bv))) bv = v, bi = c, bnan = vnan;
    }
    // wave arg-max: NaN beats everything, then the larger value, then the lower index
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      const bool on = ov != ov;
      const bool take = oi != 0x7fffffff && (bi == 0x7fffffff || (on && !bnan) || (on == bnan && (on ? oi < bi : (ov > bv || (ov == bv && oi < bi)))));
      if (take) bv = ov, bi = oi, bnan = on;
    }
    if (threadIdx.x == 0) {
      const int64_t lab = bi;
      a.label_out[b] = lab;
      const bool ok = a.targeted ? (lab == a.target[b]) : (lab != a.target[b]);
      const float m = a.metric[b];
      const bool upd = ok && m < a.best_loss[b];
      if (upd) a.best_loss[b] = m, a.best_bs[b] = a.search_step, a.best_step[b] = a.step;
      if (ok && m < a.iter_best_loss[b]) a.iter_best_loss[b] = m, a.iter_best_score[b] = lab;
      s_upd = upd;
    }
  }
  __syncthreads();
  if (!s_upd) return;
  const float* src = a.iterate + (int64_t)b * a.n3;
  float* dst = a.best_attack + (int64_t)b * a.n3;
  for (int t = threadIdx.x; t < a.n3; t += 256) dst[t] = src[t];
}

}  // namespace pc3d

using namespace pc3d;

extern "C" int pc3d_geoa3_record_f32(const float* logits, int ld, int B, int ncls, const int64_t* target, int targeted,
                                     const float* metric, const float* iterate, int n3, int64_t search_step, int64_t step,
                                     float* best_loss, float* best_attack, int64_t* best_bs, int64_t* best_step,
                                     float* iter_best_loss, int64_t* iter_best_score, int64_t* label_out, void* stream) {
  PC3D_REQUIRE(B >= 0 && ncls >= 1 && ld >= ncls && n3 >= 1, "pc3d_geoa3_record_f32: bad sizes B=%d ncls=%d ld=%d n3=%d", B, ncls, ld, n3);
  if (B == 0) return PC3D_OK;
  PC3D_REQUIRE(logits && target && metric && iterate && best_loss && best_attack && best_bs && best_step && iter_best_loss &&
                   iter_best_score && label_out, "pc3d_geoa3_record_f32: null pointer");
  GeoRecordArgs a{logits, ld, ncls, target, targeted, metric, iterate, n3, search_step, step, best_loss, best_attack,
                  best_bs, best_step, iter_best_loss, iter_best_score, label_out};
  hipLaunchKernelGGL(geoa3_record_kernel, dim3(B), dim3(256), 0, as_stream(stream), a);
  PC3D_LAUNCH_CHECK("pc3d_geoa3_record_f32");
  return PC3D_OK;
}

extern "C" int pc3d_geoa3_terms_f32(const float* d_ao, const float* d_oa, const float* k_adv, const float* k_ori,
                                    const int64_t* idx_ao, const float* cls, const float* scale, int B, int N, int M,
                                    float w_dis, float w_hd, float w_curv, float* out, int32_t* hd_arg, void* stream) {
  PC3D_REQUIRE(B >= 0 && N >= 1 && M >= 1, "pc3d_geoa3_terms_f32: bad sizes B=%d N=%d M=%d", B, N, M);
  if (B == 0) return PC3D_OK;
  PC3D_REQUIRE(d_ao && cls && scale && out && hd_arg && (!k_adv || (k_ori && idx_ao)), "pc3d_geoa3_terms_f32: null pointer");
  GeoTermsArgs a{};
  a.d_ao = d_ao, a.d_oa = d_oa, a.k_adv = k_adv, a.k_ori = k_ori, a.idx_ao = idx_ao, a.cls = cls, a.scale = scale;
  a.N = N, a.M = M, a.w_dis = w_dis, a.w_hd = w_hd, a.w_curv = w_curv, a.out = out, a.hd_arg = hd_arg;
  hipLaunchKernelGGL(geoa3_terms_fwd_kernel, dim3(B), dim3(256), 0, as_stream(stream), a, B);
  PC3D_LAUNCH_CHECK("pc3d_geoa3_terms_f32");
  return PC3D_OK;
}

extern "C" int pc3d_geoa3_terms_bwd_f32(const float* g_out, const float* k_adv, const float* k_ori, const int64_t* idx_ao,
                                        const float* scale, const int32_t* hd_arg, int B, int N, int M, float w_dis,
                                        float w_hd, float w_curv, float* g_d_ao, float* g_d_oa, float* g_k_adv,
                                        float* g_cls, void* stream) {
  PC3D_REQUIRE(B >= 0 && B <= 65535 && N >= 1 && M >= 1, "pc3d_geoa3_terms_bwd_f32: bad sizes B=%d N=%d M=%d", B, N, M);
  if (B == 0) return PC3D_OK;
  PC3D_REQUIRE(g_out && scale && hd_arg && g_d_ao && g_cls && (!g_k_adv || (k_adv && k_ori && idx_ao)),
               "pc3d_geoa3_terms_bwd_f32: null pointer");
  GeoTermsArgs a{};
  a.k_adv = k_adv, a.k_ori = k_ori, a.idx_ao = idx_ao, a.scale = scale, a.N = N, a.M = M;
  a.w_dis = w_dis, a.w_hd = w_hd, a.w_curv = w_curv, a.hd_arg = const_cast<int32_t*>(hd_arg);
  a.g_out = g_out, a.g_d_ao = g_d_ao, a.g_d_oa = g_d_oa, a.g_k_adv = g_k_adv, a.g_cls = g_cls;
  const int L = N > M ? N : M;
  hipLaunchKernelGGL(geoa3_terms_bwd_kernel, dim3(cdiv(L, 256), B), dim3(256), 0, as_stream(stream), a, B);
  PC3D_LAUNCH_CHECK("pc3d_geoa3_terms_bwd_f32");
  return PC3D_OK;
}
