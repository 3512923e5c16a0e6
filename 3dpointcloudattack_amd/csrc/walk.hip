// K16: the guided curve walk of CurveNet (model/walk.py:74-153): one launch per walk step and direction.
//
// The reference (and the step-by-step torch formulation) spends ~30 launches per walk step on tensors of
// curve_num x k x C elements (100 x 20 x 16..32): 1.5 ms forward / 3.3 ms forward+backward per CIC block at B=32, four
// blocks per CurveNet forward. Here a wavefront takes one or two curves: lane j of a group holds candidate neighbour
// j's feature row (C registers), the curve state (descriptor `pre`, current feature `cur`) sits in LDS and is read as
// broadcasts, the softmax / arg-max over the k candidates are wave reductions. The forward stores, per step, the node
// whose neighbours were scored, the picked slot, `pre` and the curve's momentum softmax; the backward walks the steps
// in reverse, recomputing scores from those.
//
// Why a launch per step: the reference reshapes the momentum softmax [B,2,cn] with .view(B,1,cn,2) (walk.py:104-105),
// i.e. curve c blends with elements 2c and 2c+1 of the FLATTENED [2,cn] array — softmax values of other curves. That
// reinterpretation is part of the function being mirrored, so the curves of a cloud exchange their momentum values
// (and, backwards, the gradients with respect to them) once per step; everything else about a curve is independent of
// the others, and the exchange goes through the arrays the forward stores anyway.
//
// Gradient scatter: d(loss)/d(feats[row]) receives (a) G_cur on the picked row, (b) gscore_j * w_nbr on each of the k
// candidate rows — a rank-1 term with the SAME vector w_nbr for every candidate, so the kernel accumulates only the
// scalar coefficient per row (coef[b,row], one atomic instead of C) and the host adds coef (x) w_nbr in one pass —
// and (c) G_pre on the start row.
#include "pc3d_common.h"

namespace pc3d {

struct WalkArgs {
  const float* feats;  // [B,N,C]
  const int* adj;      // [B,N,k]
  const int* start;    // [B,cn]
  const float* aw;     // [2C] agent weights: [0,C) neighbour part, [C,2C) descriptor part (BatchNorm folded)
  const float* ab;     // [1]
  const float* mw;     // [2,2C] momentum weights: columns [0,C) current feature, [C,2C) descriptor
  const float* mb;     // [2]
  int B, N, k, cn, L;
  float* curves;       // [B,cn,L,C]
  int* nodes;          // [B,cn,L]
  int* pick;           // [B,cn,L]
  float* pre;          // [B,cn,L,C]
  float* mom;          // [B,cn,L,2]  the curve's OWN momentum softmax entering step s (unused at s = 0)
  // backward only
  const float* gcurves;  // [B,cn,L,C]
  float* gfeats;         // [B,N,C] (accumulated)
  float* coef;           // [B,N]   (accumulated)
  float* ws;             // pc3d_curve_walk_bwd_ws_floats(B, cn, C) floats of scratch
  // deterministic mode (rec_rt != null): instead of float atomics the steps RECORD their contributions —
  //   rows : rec_rt [B, cn (L+1)] target row, rec_rv [B, cn (L+1), C] value   (record (c, s) at c L + s; the start
  //          row of curve c at cn L + c)
  //   coefs: rec_ct [B, cn L k] target row, rec_cv [B, cn L k] scalar
  // — and the entry point sums them in record order with the ordered LDS scatter of det.hip.
  int* rec_rt;
  float* rec_rv;
  int* rec_ct;
  float* rec_cv;
};

__device__ __forceinline__ float readlane_f32(float v, int l) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}

template <int C>
__device__ __forceinline__ void load_row(const float* __restrict__ p, float (&r)[C]) {
#pragma unroll
  for (int c = 0; c < C; c += 4) {
    const float4 v = *reinterpret_cast<const float4*>(p + c);
    r[c] = v.x, r[c + 1] = v.y, r[c + 2] = v.z, r[c + 3] = v.w;
  }
}
template <int C>
__device__ __forceinline__ void store_row(float* __restrict__ p, const float (&r)[C]) {
#pragma unroll
  for (int c = 0; c < C; c += 4) *reinterpret_cast<float4*>(p + c) = make_float4(r[c], r[c + 1], r[c + 2], r[c + 3]);
}

// score of candidate row nb given descriptor pre (agent_mlp, walk.py:128-131) and, from the second step on, the
// crossover-suppression factor d (walk.py:55-72, :133-137). pre / cur are wave-uniform rows (LDS or global).
// Returns score * d; *d_out = d.
template <int C>
__device__ __forceinline__ float walk_score(const WalkArgs& a, const float (&nb)[C], const float* pre,
                                            const float* cur, bool first, float* d_out) {
  float sn = 0.f, sp = 0.f;
#pragma unroll
  for (int c = 0; c < C; ++c) sn += a.aw[c] * nb[c], sp += a.aw[C + c] * pre[c];
  float sc = sn + sp + a.ab[0];
  float d = 1.f;
  if (!first) {
    float dot = 0.f, nu = 0.f, nv = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const float cc = cur[c], u = cc - pre[c], v = nb[c] - cc;
      dot += u * v, nu += u * u, nv += v * v;
    }
    const float div = fmaxf(sqrtf(nu) * sqrtf(nv), 1e-8f);
    d = fminf(fmaxf(1.f + dot / div, 0.f), 1.f);
    sc *= d;
  }
  *d_out = d;
  return sc;
}

// A curve is walked by a GROUP of G lanes (one candidate neighbour per lane): G = 32 puts two curves in a wavefront
// (k <= 32 and C <= 32: CurveNet's k = 20, C = 16 / 32), G = 64 one. Reductions stay inside the group.
template <int G>
__device__ __forceinline__ float group_max(float v) {
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
template <int G>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// lowest set lane of this group's slice of a wave ballot, 0 if none
template <int G>
__device__ __forceinline__ int group_first(unsigned long long bal, int sub) {
  const unsigned long long gb = G == 64 ? bal : ((bal >> (sub * G)) & ((1ull << (G & 63)) - 1ull));
  return gb ? __ffsll((long long)gb) - 1 : 0;
}

// One launch per walk step. Within a step the curves are independent; between steps they exchange their momentum
// softmax values (see the header), and that exchange is the launch boundary: the state a step needs (the previous
// step's descriptor `pre`, feature `curves`, node and momentum values) is exactly what the forward stores for the
// backward anyway. A workgroup is four independent wavefronts of one or two curves each, so B x cn curves spread over
// the whole chip (the one-workgroup-per-cloud form this replaces kept a cloud's 100 curves on ONE CU, 4 trips x L steps
// of issue-bound work: 109 / 182 us forward / backward at C = 32 against ~L x 6 us here).
constexpr int kWalkWaves = 4;

// flat momentum entries 2c, 2c+1 of the [2,cn] softmax array of step s (walk.py:104-105): rows f / cn of curves f % cn
__device__ __forceinline__ void walk_momentum(const WalkArgs& a, int b, int cc, int s, float* m0, float* m1) {
  const int cn = a.cn, f0 = 2 * cc, f1 = 2 * cc + 1;
  *m0 = a.mom[(((long)b * cn + f0 % cn) * a.L + s) * 2 + f0 / cn];
  *m1 = a.mom[(((long)b * cn + f1 % cn) * a.L + s) * 2 + f1 / cn];
}

template <int C, int G>
__global__ __launch_bounds__(kWalkWaves * 64) void curve_walk_fwd_step_kernel(WalkArgs a, int s) {
  constexpr int PER = 64 / G;
  __shared__ float s_rows[kWalkWaves][PER][2][C];   // per group: descriptor `pre`, current feature `cur`
  const int cn = a.cn;
  const int lane = threadIdx.x & 63, sub = lane / G, gl = lane % G;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int rounds = (cn + PER - 1) / PER;
  const long r = (long)blockIdx.x * kWalkWaves + wave;
  if (r >= (long)a.B * rounds) return;   // whole wavefronts leave; nothing below synchronises across wavefronts
  const int b = (int)(r / rounds), p = (int)(r % rounds);
  const float* __restrict__ F = a.feats + (long)b * a.N * C;
  const int* __restrict__ adj = a.adj + (long)b * a.N * a.k;
  const bool act = gl < a.k;
  // A group whose curve index runs past cn repeats the last curve with every global side effect masked (`valid`), so
  // that all lanes of a wave execute the same shuffles.
  const int c = p * PER + sub, cc = c < cn ? c : cn - 1;
  const bool valid = c < cn;
  float* pre = &s_rows[wave][sub][0][0];
  float* cur = &s_rows[wave][sub][1][0];
  const long o = ((long)b * cn + cc) * a.L + s;
  int node;
  if (s == 0) {  // walk.py:96-99: the start point's feature is the first descriptor
    node = min(max(a.start[(long)b * cn + cc], 0), a.N - 1);  // indices are clamped: a bad graph must not become a
    if (gl < C) pre[gl] = F[(long)node * C + gl];             // wild read
  } else {       // dynamic momentum (walk.py:104-108)
    node = min(max(a.nodes[o], 0), a.N - 1);
    float m0, m1;
    walk_momentum(a, b, cc, s, &m0, &m1);
    if (gl < C) {
      const float cv = a.curves[(o - 1) * C + gl], pv = a.pre[(o - 1) * C + gl];
      cur[gl] = cv;
      pre[gl] = cv * m0 + pv * m1;
    }
  }
  wave_lds_sync();
  if (valid && gl < C) a.pre[o * C + gl] = pre[gl];
  const int idx = act ? min(max(adj[(long)node * a.k + gl], 0), a.N - 1) : node;
  float nb[C];
  load_row<C>(F + (long)idx * C, nb);
  float d;
  float sc = walk_score<C>(a, nb, pre, cur, s == 0, &d);
  sc = act ? sc : -INFINITY;
  const float mx = group_max<G>(sc);
  const float e = act ? expf(sc - mx) : 0.f;
  const float y = e / group_sum<G>(e);
  const float ymax = group_max<G>(y);
  const int jstar = group_first<G>(__ballot(act && y == ymax), sub);  // lowest slot on ties
  const int next = __shfl(idx, sub * G + jstar, 64);
  wave_lds_sync();  // every lane has read cur / pre before cur is replaced
  if (gl == jstar) {  // straight-through hard pick: the new current feature is the chosen neighbour's row
    if (valid) store_row<C>(a.curves + o * C, nb);
    store_row<C>(cur, nb);
  }
  if (valid && gl == 0) {
    a.pick[o] = jstar;
    if (s == 0) a.nodes[o] = node, a.mom[o * 2] = 0.f, a.mom[o * 2 + 1] = 0.f;
  }
  if (s + 1 < a.L) {  // this curve's momentum softmax and node for the next step (walk.py:102-105)
    wave_lds_sync();
    float p0 = 0.f, p1 = 0.f;
    if (gl < C) {
      const float cv = cur[gl], pv = pre[gl];
      p0 = a.mw[gl] * cv + a.mw[C + gl] * pv;
      p1 = a.mw[2 * C + gl] * cv + a.mw[3 * C + gl] * pv;
    }
    const float z0 = group_sum<G>(p0) + a.mb[0], z1 = group_sum<G>(p1) + a.mb[1];
    const float zm = fmaxf(z0, z1), e0 = expf(z0 - zm), e1 = expf(z1 - zm), es = e0 + e1;
    if (valid && gl == 0) {
      a.mom[(o + 1) * 2] = e0 / es, a.mom[(o + 1) * 2 + 1] = e1 / es;
      a.nodes[o + 1] = next;
    }
  }
}

// Backward of step s (launched for s = L-1 .. 0). The gradients with respect to a curve's current feature and
// descriptor travel between the launches in ws: Gc [B,cn,C] | Gp [B,cn,C] | gM [2 parities][B][2 cn] (the gradients
// with respect to the flattened momentum array of a step, written by the curves that READ an entry, consumed one launch
// later by the curve that PRODUCED it).
template <int C, int G>
__global__ __launch_bounds__(kWalkWaves * 64) void curve_walk_bwd_step_kernel(WalkArgs a, int s) {
  constexpr int PER = 64 / G;
  __shared__ float s_gc[kWalkWaves][PER][C];
  const int cn = a.cn;
  const int lane = threadIdx.x & 63, sub = lane / G, gl = lane % G;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int rounds = (cn + PER - 1) / PER;
  const long r = (long)blockIdx.x * kWalkWaves + wave;
  if (r >= (long)a.B * rounds) return;
  const int b = (int)(r / rounds), p = (int)(r % rounds);
  const float* __restrict__ F = a.feats + (long)b * a.N * C;
  const int* __restrict__ adj = a.adj + (long)b * a.N * a.k;
  float* __restrict__ gF = a.gfeats + (long)b * a.N * C;
  float* __restrict__ coef = a.coef + (long)b * a.N;
  const bool act = gl < a.k;
  const int c = p * PER + sub, cc = c < cn ? c : cn - 1;
  const bool valid = c < cn;
  const long o = ((long)b * cn + cc) * a.L + s;
  float* Gcg = a.ws + ((long)b * cn + cc) * C;
  float* Gpg = Gcg + (long)a.B * cn * C;
  float* gM = a.ws + 2L * a.B * cn * C;
  float* gMw = gM + ((long)(s & 1) * a.B + b) * 2 * cn;
  const float* gMr = gM + ((long)((s + 1) & 1) * a.B + b) * 2 * cn;
  float* Gc = &s_gc[wave][sub][0];
  const bool last = s == a.L - 1;
  float gc_l = 0.f, gp_l = 0.f;   // lane gl < C holds channel gl of the two running gradients
  if (!last) {
    if (gl < C) gc_l = Gcg[gl], gp_l = Gpg[gl];
    // through the momentum softmax this curve produced for step s+1: rows (0,c) and (1,c) of that step's [2,cn] array
    const float M0 = a.mom[(o + 1) * 2], M1 = a.mom[(o + 1) * 2 + 1];
    const float gM0 = gMr[cc], gM1 = gMr[cn + cc];
    const float tt = M0 * gM0 + M1 * gM1, gz0 = M0 * (gM0 - tt), gz1 = M1 * (gM1 - tt);
    if (gl < C) {
      gc_l += a.mw[gl] * gz0 + a.mw[2 * C + gl] * gz1;
      gp_l += a.mw[C + gl] * gz0 + a.mw[3 * C + gl] * gz1;
    }
  }
  if (gl < C) {
    gc_l += a.gcurves[o * C + gl];
    Gc[gl] = gc_l;
  }
  wave_lds_sync();
  const int node = min(max(a.nodes[o], 0), a.N - 1), jstar = a.pick[o] & (G - 1);
  const int idx = act ? min(max(adj[(long)node * a.k + gl], 0), a.N - 1) : node;
  float nb[C];
  load_row<C>(F + (long)idx * C, nb);
  const float* pre = a.pre + o * C;             // group-uniform rows
  const float* curp = a.curves + (o - 1) * C;   // only dereferenced when s > 0
  float d;
  float sc = walk_score<C>(a, nb, pre, curp, s == 0, &d);
  sc = act ? sc : -INFINITY;
  const float mx = group_max<G>(sc);
  const float e = act ? expf(sc - mx) : 0.f;
  const float y = e / group_sum<G>(e);
  // cur = sum_j nb_j * (hard_j + y_j - stopgrad(y_j)):  d/d nb_j = hard_j,  d/d y_j = nb_j
  float gy = 0.f;
#pragma unroll
  for (int ch = 0; ch < C; ++ch) gy += Gc[ch] * nb[ch];
  const float t = group_sum<G>(act ? y * gy : 0.f);
  const float gsc = act ? y * (gy - t) * d : 0.f;  // softmax backward, then through the (constant) factor d
  const int prow = __shfl(idx, sub * G + jstar, 64);
  const long rrec = (long)b * cn * (a.L + 1) + (long)cc * a.L + s;      // this (curve, step)'s row record
  if (a.rec_rt) {
    if (valid && act) {
      const long cr = (((long)b * cn + cc) * a.L + s) * a.k + gl;
      a.rec_ct[cr] = idx, a.rec_cv[cr] = gsc;
    }
    if (valid && gl == 0) a.rec_rt[rrec] = prow;
    if (valid && gl < C) a.rec_rv[rrec * C + gl] = gc_l;
  } else {
    if (valid && act) atomicAdd(coef + idx, gsc);
    if (valid && gl < C) atomicAdd(gF + (long)prow * C + gl, gc_l);
  }
  const float S = group_sum<G>(gsc);
  float gp = 0.f;
  if (gl < C) gp = gp_l + S * a.aw[C + gl];  // total gradient with respect to pre_s
  if (s > 0) {
    // pre_s = cur_{s-1} m0 + pre_{s-1} m1 with (m0, m1) = entries 2c, 2c+1 of the flattened [2,cn] softmax array
    const float* pp = a.pre + (o - 1) * C;
    const float g0 = group_sum<G>(gl < C ? gp * curp[gl] : 0.f), g1 = group_sum<G>(gl < C ? gp * pp[gl] : 0.f);
    float m0, m1;
    walk_momentum(a, b, cc, s, &m0, &m1);
    if (valid && gl == 0) gMw[2 * cc] = g0, gMw[2 * cc + 1] = g1;
    if (valid && gl < C) Gcg[gl] = gp * m0, Gpg[gl] = gp * m1;  // direct terms of d pre_s / d cur_{s-1}, d pre_{s-1}
  } else if (a.rec_rt) {                      // pre_0 is the start row
    const long srec = (long)b * cn * (a.L + 1) + (long)cn * a.L + cc;
    if (valid && gl == 0) a.rec_rt[srec] = node;
    if (valid && gl < C) a.rec_rv[srec * C + gl] = gp;
  } else if (valid && gl < C) {
    atomicAdd(gF + (long)node * C + gl, gp);
  }
}

template <int C>
static int launch_walk(bool bwd, const WalkArgs& a, hipStream_t st) {
  constexpr int G = C <= 32 ? 32 : 64;   // two curves per wavefront when a candidate row fits half a wave
  const bool half = G == 32 && a.k <= 32;
  const int per = half ? 2 : 1, rounds = (a.cn + per - 1) / per;
  const long waves = (long)a.B * rounds;
  const dim3 grid((unsigned)((waves + kWalkWaves - 1) / kWalkWaves)), block(kWalkWaves * kWave);
  for (int i = 0; i < a.L; ++i) {
    const int s = bwd ? a.L - 1 - i : i;
    if (bwd) {
      if (half) hipLaunchKernelGGL((curve_walk_bwd_step_kernel<C, G>), grid, block, 0, st, a, s);
      else hipLaunchKernelGGL((curve_walk_bwd_step_kernel<C, 64>), grid, block, 0, st, a, s);
    } else {
      if (half) hipLaunchKernelGGL((curve_walk_fwd_step_kernel<C, G>), grid, block, 0, st, a, s);
      else hipLaunchKernelGGL((curve_walk_fwd_step_kernel<C, 64>), grid, block, 0, st, a, s);
    }
    PC3D_LAUNCH_CHECK(bwd ? "pc3d_curve_walk_bwd_f32" : "pc3d_curve_walk_fwd_f32");
  }
  return PC3D_OK;
}

static int walk_dispatch(bool bwd, const WalkArgs& a, int C, void* stream) {
  const char* nm = bwd ? "pc3d_curve_walk_bwd_f32" : "pc3d_curve_walk_fwd_f32";
  PC3D_REQUIRE(a.B > 0 && a.N > 0 && a.cn > 0 && a.L > 0, "%s: empty problem", nm);
  PC3D_REQUIRE((long)a.B * a.cn < (1L << 31) / kWalkWaves, "%s: B * curve_num = %ld is too large", nm, (long)a.B * a.cn);
  PC3D_REQUIRE(a.k >= 1 && a.k <= 64, "%s: k=%d (supported: 1..64, one lane per candidate)", nm, a.k);
  PC3D_REQUIRE(a.feats && a.adj && a.aw && a.ab && a.mw && a.mb && a.curves && a.nodes && a.pick && a.pre && a.mom,
               "%s: null pointer", nm);
  switch (C) {
    case 8: return launch_walk<8>(bwd, a, as_stream(stream));
    case 16: return launch_walk<16>(bwd, a, as_stream(stream));
    case 32: return launch_walk<32>(bwd, a, as_stream(stream));
    case 64: return launch_walk<64>(bwd, a, as_stream(stream));
    default: PC3D_REQUIRE(false, "%s: C=%d (supported: 8, 16, 32, 64)", nm, C);
  }
}

}  // namespace pc3d

extern "C" int pc3d_curve_walk_fwd_f32(const float* feats, const int32_t* adj, const int32_t* start,
                                       const float* agent_w, const float* agent_b, const float* mom_w,
                                       const float* mom_b, int B, int N, int C, int k, int cn, int L, float* curves,
                                       int32_t* nodes, int32_t* pick, float* pre, float* mom, void* stream) {
  PC3D_REQUIRE(start != nullptr, "pc3d_curve_walk_fwd_f32: null start");
  pc3d::WalkArgs a{feats, adj, start, agent_w, agent_b, mom_w, mom_b, B, N, k, cn, L, curves, nodes, pick, pre, mom,
                   nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  return pc3d::walk_dispatch(false, a, C, stream);
}

extern "C" int pc3d_curve_walk_bwd_f32(const float* gcurves, const float* feats, const int32_t* adj,
                                       const float* agent_w, const float* agent_b, const float* mom_w,
                                       const float* mom_b, int B, int N, int C, int k, int cn, int L,
                                       const float* curves, const int32_t* nodes, const int32_t* pick,
                                       const float* pre, const float* mom, float* gfeats, float* coef, float* ws,
                                       int deterministic, void* stream) {
  PC3D_REQUIRE(gcurves && gfeats && coef && ws, "pc3d_curve_walk_bwd_f32: null gradient / workspace pointer");
  pc3d::WalkArgs a{feats, adj, nullptr, agent_w, agent_b, mom_w, mom_b, B, N, k, cn, L,
                   const_cast<float*>(curves), const_cast<int32_t*>(nodes), const_cast<int32_t*>(pick),
                   const_cast<float*>(pre), const_cast<float*>(mom), gcurves, gfeats, coef, ws,
                   nullptr, nullptr, nullptr, nullptr};
  if (!deterministic) return pc3d::walk_dispatch(true, a, C, stream);
  // deterministic: the record arrays live behind the running-gradient scratch (pc3d_curve_walk_bwd_ws_floats with
  // deterministic = 1 sizes them); gfeats and coef are OVERWRITTEN (no zero fill needed)
  PC3D_REQUIRE(B > 0 && cn > 0 && L > 0 && k >= 1 && C >= 1, "pc3d_curve_walk_bwd_f32: empty problem");
  const int64_t nrow = (int64_t)cn * (L + 1), ncoef = (int64_t)cn * L * k;
  PC3D_REQUIRE(nrow <= 0x7fffffffLL && ncoef <= 0x7fffffffLL, "pc3d_curve_walk_bwd_f32: too many curve records");
  float* p = ws + 2 * (int64_t)B * cn * C + 4 * (int64_t)B * cn;
  a.rec_rv = p, p += (int64_t)B * nrow * C;
  a.rec_cv = p, p += (int64_t)B * ncoef;
  a.rec_rt = reinterpret_cast<int*>(p), p += (int64_t)B * nrow;
  a.rec_ct = reinterpret_cast<int*>(p);
  if (int rc = pc3d::walk_dispatch(true, a, C, stream)) return rc;
  // the per-point score coefficients first; the feature rows then leave with the rank-1 score term added on the way out
  // (gfeats[b,n,c] = sum + coef[b,n] * agent_w[c]: every candidate row gets coef * w_nbr — an addcmul launch before)
  if (int rc = pc3d::scatter_rows_det("pc3d_curve_walk_bwd_f32", a.rec_ct, a.rec_cv, 1, nullptr, 0, 0.f, B, (int)ncoef, N, 1, coef, 1, 0,
                                      1, stream))
    return rc;
  return pc3d::scatter_rows_det("pc3d_curve_walk_bwd_f32", a.rec_rt, a.rec_rv, C, nullptr, 0, 0.f, B, (int)nrow, N, C, gfeats, C, 0, 1,
                                stream, nullptr, 0, 1, coef, agent_w);
}

extern "C" int64_t pc3d_curve_walk_bwd_ws_floats(int B, int cn, int C, int L, int k, int deterministic) {
  if (B < 0 || cn < 0 || C < 0 || L < 0 || k < 0) return -1;
  int64_t n = 2 * (int64_t)B * cn * C + 4 * (int64_t)B * cn;
  if (deterministic) n += (int64_t)B * cn * (L + 1) * (C + 1) + 2 * (int64_t)B * cn * L * k;
  return n;
}
