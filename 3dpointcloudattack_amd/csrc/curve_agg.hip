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
// One workgroup (16 waves) per cloud; everything but the curves themselves lives in LDS, the block's weights included
// (the phases are short dependent loops: what they cost is load latency, so nothing inside them reads global memory
// except the one coalesced pass over the curves). The backward recomputes att and its softmaxes instead of storing them.
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

constexpr int CA_T = 1024;             // 16 waves: the phases are short and latency-bound, one workgroup per cloud
constexpr int CA_W = CA_T / 64;
constexpr size_t kCurveAggLdsMax = 160 * 1024;

// LDS map (offsets in floats). Rows that different lanes walk with a stride are padded by one float (bank conflicts).
struct CaLds {
  int sl, sn, a0, a1, cd, x, gk, gvv, tt, wa, wb, wn, wl, wc, wd, total;
};
__host__ __device__ inline int ca_groups(int C, int cn) {      // groups of curves in the one pass over them
  const int g = CA_T >= C ? CA_T / C : 1;
  return g < cn ? g : cn;
}
__host__ __device__ inline CaLds ca_layout(int cn, int cl, int C, int mid, bool bwd) {
  const int NL = cn * cl, R = cn + cl, part = ca_groups(C, cn) * cl * C, itv = 2 * R * (mid + 1);
  CaLds o;
  int p = 0;
  o.sl = p, p += NL;                                  // softmax over the curve (l)
  o.sn = p, p += NL;                                  // softmax over the curves (n)
  o.a0 = p, p += NL;                                  // att; backward: -> sl * g_sl -> g_att
  o.a1 = p, p += bwd ? NL : 0;                        // backward: sn * g_sn
  o.cd = p, p += R * (C + 1);                         // descriptors ci | cj; backward: their gradients
  o.x = p, p += bwd ? itv : (part > itv ? part : itv);  // forward: per-group partial cj, then inter|intra and Wn|Wl of them
  o.gk = p, p += bwd ? C * R : 0;                     // backward: g_Kp of this cloud
  o.gvv = p, p += bwd ? R * (C + 1) : 0;              // backward: g_Vp of this cloud
  o.tt = p, p += bwd ? R : 0;                         // backward: softmax row / column sums
  o.wa = p, p += mid * (C + 1);
  o.wb = p, p += mid * (C + 1);
  o.wn = p, p += mid * (mid + 1);
  o.wl = p, p += mid * (mid + 1);
  o.wc = p, p += mid * (C + 1);
  o.wd = p, p += C * (2 * mid + 1);
  o.total = p;
  return o;
}

// dst[r*ldd + c] = src[r*cols + c]
__device__ __forceinline__ void ca_stage(float* dst, int ldd, const float* __restrict__ src, int rows, int cols) {
  for (int e = threadIdx.x; e < rows * cols; e += CA_T) dst[(e / cols) * ldd + e % cols] = src[e];
}
__device__ __forceinline__ void ca_stage_weights(const CurveAggArgs& a, float* sm, const CaLds& o) {
  const int C = a.C, mid = a.mid;
  ca_stage(sm + o.wa, C + 1, a.Wa, mid, C);
  ca_stage(sm + o.wb, C + 1, a.Wb, mid, C);
  ca_stage(sm + o.wn, mid + 1, a.Wn, mid, mid);
  ca_stage(sm + o.wl, mid + 1, a.Wl, mid, mid);
  ca_stage(sm + o.wc, C + 1, a.Wc, mid, C);
  ca_stage(sm + o.wd, 2 * mid + 1, a.Wd, C, 2 * mid);
}

// att, softmax over l (sl) and softmax over n (sn), all [cn*cl] in LDS. Ends with a barrier. V4: C % 4 == 0 and the
// curves are 16-byte aligned (rows are then read as float4).
template <bool V4>
__device__ __forceinline__ void curve_att_softmaxes(const CurveAggArgs& a, const float* __restrict__ cv, float* att,
                                                    float* sl, float* sn) {
  const int cn = a.cn, cl = a.cl, C = a.C;
  for (int e = threadIdx.x; e < cn * cl; e += CA_T) {
    float s = 0.f;
    if (V4) {
      const float4* __restrict__ row = reinterpret_cast<const float4*>(cv + (int64_t)e * C);
      const float4* __restrict__ w4 = reinterpret_cast<const float4*>(a.w_att);
      for (int c = 0; c < C / 4; ++c) {
        const float4 v = row[c], w = w4[c];
        s += w.x * v.x, s += w.y * v.y, s += w.z * v.z, s += w.w * v.w;
      }
    } else {
      for (int c = 0; c < C; ++c) s += a.w_att[c] * cv[(int64_t)e * C + c];
    }
    att[e] = s;
  }
  __syncthreads();
  for (int n = threadIdx.x; n < cn; n += CA_T) {
    float mx = -INFINITY, sum = 0.f;
    for (int l = 0; l < cl; ++l) mx = fmaxf(mx, att[n * cl + l]);
    for (int l = 0; l < cl; ++l) sum += expf(att[n * cl + l] - mx);
    for (int l = 0; l < cl; ++l) sl[n * cl + l] = expf(att[n * cl + l] - mx) / sum;
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int l = wave; l < cl; l += CA_W) {            // a wave per position: the curves across the lanes
    float mx = -INFINITY, sum = 0.f;
    for (int n = lane; n < cn; n += 64) mx = fmaxf(mx, att[n * cl + l]);
    mx = wave_max(mx);
    for (int n = lane; n < cn; n += 64) sum += expf(att[n * cl + l] - mx);
    sum = wave_sum(sum);
    for (int n = lane; n < cn; n += 64) sn[n * cl + l] = expf(att[n * cl + l] - mx) / sum;
  }
  __syncthreads();
}

template <bool V4>
__global__ __launch_bounds__(CA_T) void curve_agg_fwd_kernel(CurveAggArgs a) {
  extern __shared__ float ca_sm[];
  const int cn = a.cn, cl = a.cl, C = a.C, mid = a.mid, R = cn + cl, b = blockIdx.x;
  const int ldc = C + 1, ldm = mid + 1;
  const CaLds o = ca_layout(cn, cl, C, mid, false);
  float* sl = ca_sm + o.sl;
  float* sn = ca_sm + o.sn;
  float* cd = ca_sm + o.cd;           // [R,ldc]  rows < cn: ci, then cj
  float* part = ca_sm + o.x;          // [G,cl,C] partial cj of each group of curves
  float* it = ca_sm + o.x;            // [R,ldm]  inter | intra   (after part is consumed)
  float* vv = it + R * ldm;           // [R,ldm]  Wn inter | Wl intra
  const float* __restrict__ cv = a.curves + (int64_t)b * cn * cl * C;
  ca_stage_weights(a, ca_sm, o);
  curve_att_softmaxes<V4>(a, cv, ca_sm + o.a0, sl, sn);
  // One pass over the curves, channels across the lanes (coalesced): thread (c, g) walks curves g, g+G, ...; ci[n,c]
  // is complete within the thread, cj[l,c] is summed over the G groups afterwards.
  const int G = ca_groups(C, cn);
  for (int w = threadIdx.x; w < G * C; w += CA_T) {
    const int c = w % C, g = w / C;
    for (int l = 0; l < cl; ++l) part[(g * cl + l) * C + c] = 0.f;
    for (int n = g; n < cn; n += G) {
      float s = 0.f;
      for (int l = 0; l < cl; ++l) {
        const float v = cv[((int64_t)n * cl + l) * C + c];
        s += v * sl[n * cl + l];
        part[(g * cl + l) * C + c] += v * sn[n * cl + l];
      }
      cd[n * ldc + c] = s;
    }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < cl * C; e += CA_T) {
    const int l = e / C, c = e % C;
    float s = 0.f;
    for (int g = 0; g < G; ++g) s += part[(g * cl + l) * C + c];
    cd[(cn + l) * ldc + c] = s;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < R * mid; e += CA_T) {
    const int r = e / mid, m = e % mid;
    const float* W = ca_sm + (r < cn ? o.wa : o.wb) + m * ldc;
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += W[c] * cd[r * ldc + c];
    it[r * ldm + m] = s;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < R * mid; e += CA_T) {
    const int r = e / mid, m2 = e % mid;
    const float* W = ca_sm + (r < cn ? o.wn : o.wl) + m2 * ldm;
    float s = 0.f;
    for (int m = 0; m < mid; ++m) s += W[m] * it[r * ldm + m];
    vv[r * ldm + m2] = s;
  }
  for (int e = threadIdx.x; e < C * R; e += CA_T) {   // keys: Kp[c,r] = sum_m Wc[m,c] it[r,m]
    const int c = e / R, r = e % R;
    const float* W = ca_sm + o.wc + c;
    float s = 0.f;
    for (int m = 0; m < mid; ++m) s += W[m * ldc] * it[r * ldm + m];
    a.Kp[(int64_t)b * C * R + e] = s;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < R * C; e += CA_T) {   // values: Vp[r,c] = sum_m Wd[c, off+m] vv[r,m] (+ bd on curves)
    const int r = e / C, c = e % C;
    const float* W = ca_sm + o.wd + c * (2 * mid + 1) + (r < cn ? 0 : mid);
    float s = r < cn ? a.bd[c] : 0.f;
    for (int m = 0; m < mid; ++m) s += W[m] * vv[r * ldm + m];
    a.Vp[(int64_t)b * R * C + e] = s;
  }
}

template <bool V4>
__global__ __launch_bounds__(CA_T) void curve_agg_bwd_kernel(CurveAggArgs a) {
  extern __shared__ float ca_sm[];
  const int cn = a.cn, cl = a.cl, C = a.C, mid = a.mid, R = cn + cl, b = blockIdx.x;
  const int ldc = C + 1, ldm = mid + 1, ldd = 2 * mid + 1;
  const CaLds o = ca_layout(cn, cl, C, mid, true);
  float* sl = ca_sm + o.sl;
  float* sn = ca_sm + o.sn;
  float* psl = ca_sm + o.a0;         // att, then sl * g_sl, then g_att
  float* psn = ca_sm + o.a1;         // sn * g_sn
  float* gc = ca_sm + o.cd;          // [R,ldc]  g_ci | g_cj
  float* gv = ca_sm + o.x;           // [R,ldm]
  float* gi = gv + R * ldm;          // [R,ldm]
  float* gK = ca_sm + o.gk;          // [C,R]
  float* gV = ca_sm + o.gvv;         // [R,ldc]
  float* tt = ca_sm + o.tt;          // [R]
  const float* __restrict__ cv = a.curves + (int64_t)b * cn * cl * C;
  ca_stage_weights(a, ca_sm, o);
  ca_stage(gK, R, a.gKp + (int64_t)b * C * R, C, R);
  ca_stage(gV, ldc, a.gVp + (int64_t)b * R * C, R, C);
  curve_att_softmaxes<V4>(a, cv, psl, sl, sn);
  for (int e = threadIdx.x; e < R * mid; e += CA_T) {   // through convd's half: g_vv[r,m] = sum_c Wd[c,off+m] gVp[r,c]
    const int r = e / mid, m = e % mid;
    const float* W = ca_sm + o.wd + (r < cn ? 0 : mid) + m;
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += W[c * ldd] * gV[r * ldc + c];
    gv[r * ldm + m] = s;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < R * mid; e += CA_T) {   // g_it[r,m] = sum_c Wc[m,c] gKp[c,r] + sum_m2 W(n|l)[m2,m] g_vv[r,m2]
    const int r = e / mid, m = e % mid;
    const float* W = ca_sm + (r < cn ? o.wn : o.wl) + m;
    const float* Wc = ca_sm + o.wc + m * ldc;
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += Wc[c] * gK[c * R + r];
    for (int m2 = 0; m2 < mid; ++m2) s += W[m2 * ldm] * gv[r * ldm + m2];
    gi[r * ldm + m] = s;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < R * C; e += CA_T) {     // g_cd[r,c] = sum_m W(a|b)[m,c] g_it[r,m]
    const int r = e / C, c = e % C;
    const float* W = ca_sm + (r < cn ? o.wa : o.wb) + c;
    float s = 0.f;
    for (int m = 0; m < mid; ++m) s += W[m * ldc] * gi[r * ldm + m];
    gc[r * ldc + c] = s;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < cn * cl; e += CA_T) {   // gradients of the two softmax outputs, times the outputs
    const int n = e / cl, l = e % cl;
    const float* g1 = gc + n * ldc;
    const float* g2 = gc + (cn + l) * ldc;
    float s1 = 0.f, s2 = 0.f;
    if (V4) {
      const float4* __restrict__ row = reinterpret_cast<const float4*>(cv + (int64_t)e * C);
      for (int c = 0; c < C / 4; ++c) {
        const float4 v = row[c];
        s1 += g1[4 * c] * v.x, s1 += g1[4 * c + 1] * v.y, s1 += g1[4 * c + 2] * v.z, s1 += g1[4 * c + 3] * v.w;
        s2 += g2[4 * c] * v.x, s2 += g2[4 * c + 1] * v.y, s2 += g2[4 * c + 2] * v.z, s2 += g2[4 * c + 3] * v.w;
      }
    } else {
      for (int c = 0; c < C; ++c) {
        const float v = cv[(int64_t)e * C + c];
        s1 += g1[c] * v, s2 += g2[c] * v;
      }
    }
    psl[e] = sl[e] * s1, psn[e] = sn[e] * s2;
  }
  __syncthreads();
  for (int n = threadIdx.x; n < cn; n += CA_T) {
    float s = 0.f;
    for (int l = 0; l < cl; ++l) s += psl[n * cl + l];
    tt[n] = s;
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int l = wave; l < cl; l += CA_W) {
    float s = 0.f;
    for (int n = lane; n < cn; n += 64) s += psn[n * cl + l];
    s = wave_sum(s);
    if (lane == 0) tt[cn + l] = s;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < cn * cl; e += CA_T) {   // g_att, in place
    const int n = e / cl, l = e % cl;
    psl[e] = psl[e] - sl[e] * tt[n] + psn[e] - sn[e] * tt[cn + l];
  }
  __syncthreads();
  float* __restrict__ go = a.gcurves + (int64_t)b * cn * cl * C;
  for (int e = threadIdx.x; e < cn * cl * C; e += CA_T) {
    const int c = e % C, nl = e / C, n = nl / cl, l = nl % cl;
    go[e] = gc[n * ldc + c] * sl[nl] + gc[(cn + l) * ldc + c] * sn[nl] + a.w_att[c] * psl[nl];
  }
}

static int curve_agg_check(const char* nm, int B, const CurveAggArgs& a, size_t lds) {
  PC3D_REQUIRE(B >= 0 && a.cn >= 1 && a.cl >= 1 && a.C >= 1 && a.mid >= 1, "%s: bad sizes B=%d cn=%d cl=%d C=%d mid=%d",
               nm, B, a.cn, a.cl, a.C, a.mid);
  PC3D_REQUIRE(lds <= kCurveAggLdsMax, "%s: cn=%d cl=%d C=%d mid=%d needs %zu bytes of LDS (limit %zu)", nm, a.cn, a.cl,
               a.C, a.mid, lds, kCurveAggLdsMax);
  PC3D_REQUIRE(a.curves && a.w_att && a.Wa && a.Wb && a.Wn && a.Wl && a.Wc && a.Wd && a.bd, "%s: null pointer", nm);
  return PC3D_OK;
}

// The kernels take up to the CU's whole 160 KB of LDS; above the default 64 KB window that has to be allowed per kernel
// and device, once (outside any stream capture: the first launch of a victim is always an eager warm-up).
template <typename K>
static int curve_agg_allow_lds(const char* nm, K kernel, bool* done) {
  int dev = 0;
  PC3D_REQUIRE(hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64, "%s: no current device", nm);
  if (!done[dev]) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)kCurveAggLdsMax);
    PC3D_REQUIRE(e == hipSuccess, "%s: hipFuncSetAttribute(MaxDynamicSharedMemorySize): %s", nm, hipGetErrorString(e));
    done[dev] = true;
  }
  return PC3D_OK;
}
static bool ca_v4(const CurveAggArgs& a) {
  return a.C % 4 == 0 && (reinterpret_cast<uintptr_t>(a.curves) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.w_att) & 15) == 0;
}

}  // namespace pc3d

using namespace pc3d;

extern "C" int64_t pc3d_curve_agg_lds_bytes(int cn, int cl, int C, int mid, int backward) {
  if (cn < 1 || cl < 1 || C < 1 || mid < 1) return -1;
  const int64_t R = (int64_t)cn + cl, NL = (int64_t)cn * cl;        // guard the int arithmetic of ca_layout
  if (NL > (1 << 22) || R * (C + 1) > (1 << 22) || (int64_t)C * (2 * mid + 1) > (1 << 22) || (int64_t)mid * (C + 1) > (1 << 22) ||
      (int64_t)ca_groups(C, cn) * cl * C > (1 << 22) || R * (mid + 1) > (1 << 22))
    return INT64_MAX;
  return (int64_t)sizeof(float) * ca_layout(cn, cl, C, mid, backward != 0).total;
}

extern "C" int pc3d_curve_agg_kv_f32(const float* curves, const float* w_att, const float* Wa, const float* Wb,
                                     const float* Wn, const float* Wl, const float* Wc, const float* Wd,
                                     const float* bd, int B, int cn, int cl, int C, int mid, float* Kp, float* Vp,
                                     void* stream) {
  CurveAggArgs a{curves, w_att, Wa, Wb, Wn, Wl, Wc, Wd, bd, cn, cl, C, mid, Kp, Vp, nullptr, nullptr, nullptr};
  const char* nm = "pc3d_curve_agg_kv_f32";
  const size_t lds = (size_t)pc3d_curve_agg_lds_bytes(cn, cl, C, mid, 0);
  if (int rc = curve_agg_check(nm, B, a, lds)) return rc;
  if (B == 0) return PC3D_OK;
  PC3D_REQUIRE(Kp && Vp, "pc3d_curve_agg_kv_f32: null output");
  static bool allowed[2][64];
  if (ca_v4(a)) {
    if (int rc = curve_agg_allow_lds(nm, curve_agg_fwd_kernel<true>, allowed[1])) return rc;
    hipLaunchKernelGGL(curve_agg_fwd_kernel<true>, dim3(B), dim3(CA_T), lds, as_stream(stream), a);
  } else {
    if (int rc = curve_agg_allow_lds(nm, curve_agg_fwd_kernel<false>, allowed[0])) return rc;
    hipLaunchKernelGGL(curve_agg_fwd_kernel<false>, dim3(B), dim3(CA_T), lds, as_stream(stream), a);
  }
  PC3D_LAUNCH_CHECK("pc3d_curve_agg_kv_f32");
  return PC3D_OK;
}

extern "C" int pc3d_curve_agg_kv_bwd_f32(const float* gKp, const float* gVp, const float* curves, const float* w_att,
                                         const float* Wa, const float* Wb, const float* Wn, const float* Wl,
                                         const float* Wc, const float* Wd, const float* bd, int B, int cn, int cl,
                                         int C, int mid, float* gcurves, void* stream) {
  CurveAggArgs a{curves, w_att, Wa, Wb, Wn, Wl, Wc, Wd, bd, cn, cl, C, mid, nullptr, nullptr, gKp, gVp, gcurves};
  const char* nm = "pc3d_curve_agg_kv_bwd_f32";
  const size_t lds = (size_t)pc3d_curve_agg_lds_bytes(cn, cl, C, mid, 1);
  if (int rc = curve_agg_check(nm, B, a, lds)) return rc;
  if (B == 0) return PC3D_OK;
  PC3D_REQUIRE(gKp && gVp && gcurves, "pc3d_curve_agg_kv_bwd_f32: null gradient pointer");
  static bool allowed[2][64];
  if (ca_v4(a)) {
    if (int rc = curve_agg_allow_lds(nm, curve_agg_bwd_kernel<true>, allowed[1])) return rc;
    hipLaunchKernelGGL(curve_agg_bwd_kernel<true>, dim3(B), dim3(CA_T), lds, as_stream(stream), a);
  } else {
    if (int rc = curve_agg_allow_lds(nm, curve_agg_bwd_kernel<false>, allowed[0])) return rc;
    hipLaunchKernelGGL(curve_agg_bwd_kernel<false>, dim3(B), dim3(CA_T), lds, as_stream(stream), a);
  }
  PC3D_LAUNCH_CHECK("pc3d_curve_agg_kv_bwd_f32");
  return PC3D_OK;
}
