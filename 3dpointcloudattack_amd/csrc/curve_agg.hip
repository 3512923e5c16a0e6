// K18: the per-cloud half of CurveNet's curve aggregation (model/curvenet_util.py:379-437, `CurveAggregation`) as one
// launch per direction. From the curves of one cloud [cn, cl, C] it produces the attention KEYS and VALUES that the
// per-point half then uses in two batched GEMMs:
//     att[n,l]  = w_att . curves[n,l,:]                                    (line_conv_att, :395)
//     ci[n,:]   = sum_l curves[n,l,:] softmax_l(att)[n,l]                  (inter-curve descriptor, :413-416)
//     cj[l,:]   = sum_n curves[n,l,:] softmax_n(att)[n,l]                  (intra-curve descriptor, :418-420)
//     inter = Wa ci, intra = Wb cj                                         (conva / convb, :422-423)
//     keys    Kp[:, r] = Wc^T (inter | intra)[r]        so that  x_i . Kp[:, r] = (convc x_i) . (inter | intra)[r]
//     values  Vp[r, :] = Wd_half (Wn inter | Wl intra)[r]  (+ the folded BatchNorm shift on the inter rows: each
//             softmax row sums to one, so the shift lands on every point exactly once)
// with r running over the cn curves, then the cl positions. Pre-multiplying by convc / convd is exact algebra (both
// are linear and sit directly against the attention products); it removes two GEMMs, a concat and two transposes per
// block from the per-point half. The reference runs ~20 launches on these [cn x cl] tensors, ~45 backwards.
// One workgroup per cloud; everything but the curves themselves lives in LDS. The backward recomputes att and its
// softmaxes instead of storing them.
#include "pc3d_common.h"

namespace pc3d {

struct CurveAggArgs {
  const float* curves;  // [B,cn,cl,C]
  const float* w_att;   // [C]
  const float* Wa;      // [mid,C]
  const float* Wb;      // [mid,C]
  const float* Wn;      // [mid,mid]
  const float* Wl;      // [mid,mid]
  const float* Wc;      // [mid,C]
  const float* Wd;      // [C,2*mid]  (BatchNorm scale folded)
  const float* bd;      // [C]        (BatchNorm shift)
  int cn, cl, C, mid;
  float* Kp;            // [B,C,R]   R = cn + cl
  float* Vp;            // [B,R,C]
  // backward
  const float* gKp;     // [B,C,R]
  const float* gVp;     // [B,R,C]
  float* gcurves;       // [B,cn,cl,C]
};

constexpr int CA_T = 256;

// att, softmax over l (sl) and softmax over n (sn), all [cn*cl] in LDS. Ends with a barrier.
__device__ __forceinline__ void curve_att_softmaxes(const CurveAggArgs& a, const float* __restrict__ cv, float* att,
                                                    float* sl, float* sn) {
  const int cn = a.cn, cl = a.cl, C = a.C;
  for (int e = threadIdx.x; e < cn * cl; e += CA_T) {
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += a.w_att[c] * cv[(int64_t)e * C + c];
    att[e] = s;
  }
  __syncthreads();
  for (int n = threadIdx.x; n < cn; n += CA_T) {
    float mx = -INFINITY, sum = 0.f;
    for (int l = 0; l < cl; ++l) mx = fmaxf(mx, att[n * cl + l]);
    for (int l = 0; l < cl; ++l) sum += expf(att[n * cl + l] - mx);
    for (int l = 0; l < cl; ++l) sl[n * cl + l] = expf(att[n * cl + l] - mx) / sum;
  }
  for (int l = threadIdx.x; l < cl; l += CA_T) {
    float mx = -INFINITY, sum = 0.f;
    for (int n = 0; n < cn; ++n) mx = fmaxf(mx, att[n * cl + l]);
    for (int n = 0; n < cn; ++n) sum += expf(att[n * cl + l] - mx);
    for (int n = 0; n < cn; ++n) sn[n * cl + l] = expf(att[n * cl + l] - mx) / sum;
  }
  __syncthreads();
}

static size_t curve_agg_fwd_lds(int cn, int cl, int C, int mid) {
  const size_t R = (size_t)cn + cl;
  return sizeof(float) * (3 * (size_t)cn * cl + R * C + 2 * R * mid);
}
static size_t curve_agg_bwd_lds(int cn, int cl, int C, int mid) {
  const size_t R = (size_t)cn + cl;
  return sizeof(float) * (6 * (size_t)cn * cl + R * C + 2 * R * mid + R);
}

__global__ __launch_bounds__(CA_T) void curve_agg_fwd_kernel(CurveAggArgs a) {
  extern __shared__ float ca_sm[];
  const int cn = a.cn, cl = a.cl, C = a.C, mid = a.mid, R = cn + cl, b = blockIdx.x;
  float* att = ca_sm;               // [cn*cl]
  float* sl = att + cn * cl;        // [cn*cl]
  float* sn = sl + cn * cl;         // [cn*cl]
  float* cd = sn + cn * cl;         // [R,C]   rows < cn: ci, then cj
  float* it = cd + R * C;           // [R,mid] inter | intra
  float* vv = it + R * mid;         // [R,mid] Wn inter | Wl intra
  const float* __restrict__ cv = a.curves + (int64_t)b * cn * cl * C;
  curve_att_softmaxes(a, cv, att, sl, sn);
  for (int e = threadIdx.x; e < R * C; e += CA_T) {
    const int r = e / C, c = e % C;
    float s = 0.f;
    if (r < cn) {
      for (int l = 0; l < cl; ++l) s += cv[((int64_t)r * cl + l) * C + c] * sl[r * cl + l];
    } else {
      const int l = r - cn;
      for (int n = 0; n < cn; ++n) s += cv[((int64_t)n * cl + l) * C + c] * sn[n * cl + l];
    }
    cd[e] = s;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < R * mid; e += CA_T) {
    const int r = e / mid, m = e % mid;
    const float* W = (r < cn ? a.Wa : a.Wb) + (int64_t)m * C;
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += W[c] * cd[r * C + c];
    it[e] = s;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < R * mid; e += CA_T) {
    const int r = e / mid, m2 = e % mid;
    const float* W = (r < cn ? a.Wn : a.Wl) + (int64_t)m2 * mid;
    float s = 0.f;
    for (int m = 0; m < mid; ++m) s += W[m] * it[r * mid + m];
    vv[e] = s;
  }
  for (int e = threadIdx.x; e < C * R; e += CA_T) {   // keys: Kp[c,r] = sum_m Wc[m,c] it[r,m]
    const int c = e / R, r = e % R;
    float s = 0.f;
    for (int m = 0; m < mid; ++m) s += a.Wc[(int64_t)m * C + c] * it[r * mid + m];
    a.Kp[(int64_t)b * C * R + e] = s;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < R * C; e += CA_T) {   // values: Vp[r,c] = sum_m Wd[c, off+m] vv[r,m] (+ bd on curves)
    const int r = e / C, c = e % C;
    const float* W = a.Wd + (int64_t)c * 2 * mid + (r < cn ? 0 : mid);
    float s = r < cn ? a.bd[c] : 0.f;
    for (int m = 0; m < mid; ++m) s += W[m] * vv[r * mid + m];
    a.Vp[(int64_t)b * R * C + e] = s;
  }
}

__global__ __launch_bounds__(CA_T) void curve_agg_bwd_kernel(CurveAggArgs a) {
  extern __shared__ float ca_sm[];
  const int cn = a.cn, cl = a.cl, C = a.C, mid = a.mid, R = cn + cl, b = blockIdx.x;
  float* att = ca_sm;                // [cn*cl]  (reused as g_att at the end)
  float* sl = att + cn * cl;
  float* sn = sl + cn * cl;
  float* gsl = sn + cn * cl;         // [cn*cl]
  float* gsn = gsl + cn * cl;        // [cn*cl]
  float* gat = gsn + cn * cl;        // [cn*cl]
  float* gc = gat + cn * cl;         // [R,C]    g_ci | g_cj
  float* gv = gc + R * C;            // [R,mid]
  float* gi = gv + R * mid;          // [R,mid]
  float* tt = gi + R * mid;          // [R]      softmax-backward row / column sums
  const float* __restrict__ cv = a.curves + (int64_t)b * cn * cl * C;
  const float* __restrict__ gK = a.gKp + (int64_t)b * C * R;
  const float* __restrict__ gV = a.gVp + (int64_t)b * R * C;
  curve_att_softmaxes(a, cv, att, sl, sn);
  for (int e = threadIdx.x; e < R * mid; e += CA_T) {   // through convd's half: g_vv[r,m] = sum_c Wd[c,off+m] gVp[r,c]
    const int r = e / mid, m = e % mid, off = r < cn ? 0 : mid;
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += a.Wd[(int64_t)c * 2 * mid + off + m] * gV[r * C + c];
    gv[e] = s;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < R * mid; e += CA_T) {   // g_it[r,m] = sum_c Wc[m,c] gKp[c,r] + sum_m2 W(n|l)[m2,m] g_vv[r,m2]
    const int r = e / mid, m = e % mid;
    const float* W = r < cn ? a.Wn : a.Wl;
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += a.Wc[(int64_t)m * C + c] * gK[(int64_t)c * R + r];
    for (int m2 = 0; m2 < mid; ++m2) s += W[(int64_t)m2 * mid + m] * gv[r * mid + m2];
    gi[e] = s;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < R * C; e += CA_T) {     // g_cd[r,c] = sum_m W(a|b)[m,c] g_it[r,m]
    const int r = e / C, c = e % C;
    const float* W = r < cn ? a.Wa : a.Wb;
    float s = 0.f;
    for (int m = 0; m < mid; ++m) s += W[(int64_t)m * C + c] * gi[r * mid + m];
    gc[e] = s;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < cn * cl; e += CA_T) {   // gradients of the two softmax outputs
    const int n = e / cl, l = e % cl;
    float s1 = 0.f, s2 = 0.f;
    for (int c = 0; c < C; ++c) {
      const float v = cv[(int64_t)e * C + c];
      s1 += gc[n * C + c] * v, s2 += gc[(cn + l) * C + c] * v;
    }
    gsl[e] = s1, gsn[e] = s2;
  }
  __syncthreads();
  for (int r = threadIdx.x; r < R; r += CA_T) {
    float s = 0.f;
    if (r < cn) {
      for (int l = 0; l < cl; ++l) s += sl[r * cl + l] * gsl[r * cl + l];
    } else {
      const int l = r - cn;
      for (int n = 0; n < cn; ++n) s += sn[n * cl + l] * gsn[n * cl + l];
    }
    tt[r] = s;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < cn * cl; e += CA_T) {
    const int n = e / cl, l = e % cl;
    gat[e] = sl[e] * (gsl[e] - tt[n]) + sn[e] * (gsn[e] - tt[cn + l]);
  }
  __syncthreads();
  float* __restrict__ go = a.gcurves + (int64_t)b * cn * cl * C;
  for (int e = threadIdx.x; e < cn * cl * C; e += CA_T) {
    const int c = e % C, nl = e / C, n = nl / cl, l = nl % cl;
    go[e] = gc[n * C + c] * sl[nl] + gc[(cn + l) * C + c] * sn[nl] + a.w_att[c] * gat[nl];
  }
}

static int curve_agg_check(const char* nm, int B, const CurveAggArgs& a, size_t lds) {
  PC3D_REQUIRE(B >= 0 && a.cn >= 1 && a.cl >= 1 && a.C >= 1 && a.mid >= 1, "%s: bad sizes B=%d cn=%d cl=%d C=%d mid=%d",
               nm, B, a.cn, a.cl, a.C, a.mid);
  PC3D_REQUIRE(lds <= 64 * 1024, "%s: cn=%d cl=%d C=%d mid=%d needs %zu bytes of LDS (limit 65536)", nm, a.cn, a.cl, a.C,
               a.mid, lds);
  PC3D_REQUIRE(a.curves && a.w_att && a.Wa && a.Wb && a.Wn && a.Wl && a.Wc && a.Wd && a.bd, "%s: null pointer", nm);
  return PC3D_OK;
}

}  // namespace pc3d

using namespace pc3d;

extern "C" int pc3d_curve_agg_kv_f32(const float* curves, const float* w_att, const float* Wa, const float* Wb,
                                     const float* Wn, const float* Wl, const float* Wc, const float* Wd,
                                     const float* bd, int B, int cn, int cl, int C, int mid, float* Kp, float* Vp,
                                     void* stream) {
  CurveAggArgs a{curves, w_att, Wa, Wb, Wn, Wl, Wc, Wd, bd, cn, cl, C, mid, Kp, Vp, nullptr, nullptr, nullptr};
  const size_t lds = curve_agg_fwd_lds(cn, cl, C, mid);
  if (int rc = curve_agg_check("pc3d_curve_agg_kv_f32", B, a, lds)) return rc;
  if (B == 0) return PC3D_OK;
  PC3D_REQUIRE(Kp && Vp, "pc3d_curve_agg_kv_f32: null output");
  hipLaunchKernelGGL(curve_agg_fwd_kernel, dim3(B), dim3(CA_T), lds, as_stream(stream), a);
  PC3D_LAUNCH_CHECK("pc3d_curve_agg_kv_f32");
  return PC3D_OK;
}

extern "C" int pc3d_curve_agg_kv_bwd_f32(const float* gKp, const float* gVp, const float* curves, const float* w_att,
                                         const float* Wa, const float* Wb, const float* Wn, const float* Wl,
                                         const float* Wc, const float* Wd, const float* bd, int B, int cn, int cl,
                                         int C, int mid, float* gcurves, void* stream) {
  CurveAggArgs a{curves, w_att, Wa, Wb, Wn, Wl, Wc, Wd, bd, cn, cl, C, mid, nullptr, nullptr, gKp, gVp, gcurves};
  const size_t lds = curve_agg_bwd_lds(cn, cl, C, mid);
  if (int rc = curve_agg_check("pc3d_curve_agg_kv_bwd_f32", B, a, lds)) return rc;
  if (B == 0) return PC3D_OK;
  PC3D_REQUIRE(gKp && gVp && gcurves, "pc3d_curve_agg_kv_bwd_f32: null gradient pointer");
  hipLaunchKernelGGL(curve_agg_bwd_kernel, dim3(B), dim3(CA_T), lds, as_stream(stream), a);
  PC3D_LAUNCH_CHECK("pc3d_curve_agg_kv_bwd_f32");
  return PC3D_OK;
}
