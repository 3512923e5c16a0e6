// K16: the guided curve walk of CurveNet (model/walk.py:74-153) as ONE forward and ONE backward launch.
//
// The reference (and the step-by-step torch formulation) spends ~30 launches per walk step on tensors of
// curve_num x k x C elements (100 x 20 x 16..32): 1.5 ms forward / 3.3 ms forward+backward per CIC block at B=32, four
// blocks per CurveNet forward. Here one workgroup walks all curves of one cloud; a wavefront takes one or two curves
// at a time: lane j of a group holds candidate neighbour j's feature row (C registers), the curve state (descriptor `pre`, current
// feature `cur`) lives in LDS and is read as broadcasts, the softmax / arg-max over the k candidates are wave
// reductions. The forward stores, per step, the node whose neighbours were scored, the picked slot, `pre` and the
// curve's momentum softmax; the backward walks the steps in reverse, recomputing scores from those.
//
// Why a workgroup per cloud and not a wave per curve: the reference reshapes the momentum softmax [B,2,cn] with
// .view(B,1,cn,2) (walk.py:104-105), i.e. curve c blends with elements 2c and 2c+1 of the FLATTENED [2,cn] array —
// softmax values of other curves. That reinterpretation is part of the function being mirrored, so the curves of a
// cloud exchange their momentum values (and, backwards, the gradients with respect to them) through LDS once per step.
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
};

__device__ __forceinline__ float readlane_f32(float v, int l) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
// Orders LDS traffic between the lanes of ONE wavefront (LDS operations of a wave complete in issue order; this only
// stops the compiler from moving them across).
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
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

// workgroup size: 16 waves (128 VGPRs each) hold a C <= 32 row per lane; C = 64 needs the 256-VGPR budget of 8 waves
template <int C>
constexpr int kWalkThreads = C <= 32 ? 1024 : 512;

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

// LDS (floats): pre [cn][C] | cur [cn][C] | M [2 parities][2][cn] | node [cn] (int)
static size_t walk_fwd_lds(int cn, int C) { return sizeof(float) * ((size_t)2 * cn * C + 5 * cn); }
// LDS (floats): Gc [cn][C] | Gp [cn][C] | gM [2 parities][2 cn]
static size_t walk_bwd_lds(int cn, int C) { return sizeof(float) * ((size_t)2 * cn * C + 4 * cn); }

template <int C, int G>
__global__ __launch_bounds__(kWalkThreads<C>) void curve_walk_fwd_kernel(WalkArgs a) {
  extern __shared__ float4 walk_sm4[];
  constexpr int PER = 64 / G;
  const int cn = a.cn;
  float* const s_pre = reinterpret_cast<float*>(walk_sm4);
  float* const s_cur = s_pre + cn * C;
  float* const s_M = s_cur + cn * C;
  int* const s_node = reinterpret_cast<int*>(s_M + 4 * cn);
  const int lane = threadIdx.x & 63, sub = lane / G, gl = lane % G;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
  const int rounds = (cn + PER - 1) / PER, b = blockIdx.x;
  const float* __restrict__ F = a.feats + (long)b * a.N * C;
  const int* __restrict__ adj = a.adj + (long)b * a.N * a.k;
  const bool act = gl < a.k;
  // A group whose curve index runs past cn repeats the last curve with every side effect masked (`valid`), so that
  // all lanes of a wave execute the same shuffles.
  for (int p = wave; p < rounds; p += nw) {  // walk.py:96-99: the start point's feature is the first descriptor
    const int c = p * PER + sub, cc = c < cn ? c : cn - 1;
    const bool valid = c < cn;
    const int node = min(max(a.start[(long)b * cn + cc], 0), a.N - 1);  // indices are clamped: a bad graph must not
    if (valid && gl < C) s_pre[cc * C + gl] = F[(long)node * C + gl];   // become a wild read
    if (valid && gl == 0) s_node[cc] = node;
  }
  __syncthreads();
  for (int s = 0; s < a.L; ++s) {
    const float* Mr = s_M + (s & 1) * 2 * cn;
    float* Mw = s_M + ((s + 1) & 1) * 2 * cn;
    for (int p = wave; p < rounds; p += nw) {
      const int c = p * PER + sub, cc = c < cn ? c : cn - 1;
      const bool valid = c < cn;
      float* pre = s_pre + cc * C;
      float* cur = s_cur + cc * C;
      const long o = ((long)b * cn + cc) * a.L + s;
      if (s > 0) {  // dynamic momentum (walk.py:104-108): entries 2c, 2c+1 of the flattened [2,cn] softmax array
        const float m0 = Mr[2 * cc], m1 = Mr[2 * cc + 1];
        if (valid && gl < C) pre[gl] = cur[gl] * m0 + pre[gl] * m1;
        wave_lds_sync();
      }
      if (valid && gl < C) a.pre[o * C + gl] = pre[gl];
      const int node = s_node[cc];
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
      if (valid && gl == jstar) {  // straight-through hard pick: the new current feature is the chosen neighbour's row
        store_row<C>(a.curves + o * C, nb);
        store_row<C>(cur, nb);
      }
      if (valid && gl == 0) {
        a.nodes[o] = node, a.pick[o] = jstar;
        if (s == 0) a.mom[o * 2] = 0.f, a.mom[o * 2 + 1] = 0.f;
      }
      if (s + 1 < a.L) {  // this curve's momentum softmax for the next step (walk.py:102-105)
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
          Mw[cc] = e0 / es, Mw[cn + cc] = e1 / es;
          a.mom[(o + 1) * 2] = e0 / es, a.mom[(o + 1) * 2 + 1] = e1 / es;
        }
      }
      // the node a curve stands on is only read again after the barrier below
      if (valid && gl == 0) s_node[cc] = next;
    }
    __syncthreads();
  }
}

template <int C, int G>
__global__ __launch_bounds__(kWalkThreads<C>) void curve_walk_bwd_kernel(WalkArgs a) {
  extern __shared__ float4 walk_sm4[];
  constexpr int PER = 64 / G;
  const int cn = a.cn;
  float* const s_Gc = reinterpret_cast<float*>(walk_sm4);
  float* const s_Gp = s_Gc + cn * C;
  float* const s_gM = s_Gp + cn * C;
  const int lane = threadIdx.x & 63, sub = lane / G, gl = lane % G;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
  const int rounds = (cn + PER - 1) / PER, b = blockIdx.x;
  const float* __restrict__ F = a.feats + (long)b * a.N * C;
  const int* __restrict__ adj = a.adj + (long)b * a.N * a.k;
  float* __restrict__ gF = a.gfeats + (long)b * a.N * C;
  float* __restrict__ coef = a.coef + (long)b * a.N;
  const bool act = gl < a.k;
  for (int i = threadIdx.x; i < 2 * cn * C; i += blockDim.x) s_Gc[i] = 0.f;  // Gc and Gp are adjacent
  __syncthreads();
  for (int s = a.L - 1; s >= 0; --s) {
    float* gMw = s_gM + (s & 1) * 2 * cn;
    for (int p = wave; p < rounds; p += nw) {
      const int c = p * PER + sub, cc = c < cn ? c : cn - 1;
      const bool valid = c < cn;
      float* Gc = s_Gc + cc * C;
      float* Gp = s_Gp + cc * C;
      const long o = ((long)b * cn + cc) * a.L + s;
      if (valid && gl < C) Gc[gl] += a.gcurves[o * C + gl];
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
      if (valid && act) atomicAdd(coef + idx, gsc);
      const int prow = __shfl(idx, sub * G + jstar, 64);
      if (valid && gl < C) atomicAdd(gF + (long)prow * C + gl, Gc[gl]);
      const float S = group_sum<G>(gsc);
      float gp = 0.f;
      if (gl < C) gp = Gp[gl] + S * a.aw[C + gl];  // total gradient with respect to pre_s
      if (s > 0) {
        // pre_s = cur_{s-1} m0 + pre_{s-1} m1 with (m0, m1) = entries 2c, 2c+1 of the flattened [2,cn] softmax array
        const float* pp = a.pre + (o - 1) * C;
        const float g0 = group_sum<G>(gl < C ? gp * curp[gl] : 0.f), g1 = group_sum<G>(gl < C ? gp * pp[gl] : 0.f);
        const int f0 = 2 * cc, f1 = 2 * cc + 1;
        const float m0 = a.mom[(((long)b * cn + f0 % cn) * a.L + s) * 2 + f0 / cn];
        const float m1 = a.mom[(((long)b * cn + f1 % cn) * a.L + s) * 2 + f1 / cn];
        if (valid && gl == 0) gMw[f0] = g0, gMw[f1] = g1;
        wave_lds_sync();  // Gc was read (scatter) before it is replaced
        if (valid && gl < C) Gc[gl] = gp * m0, Gp[gl] = gp * m1;  // direct terms of d pre_s / d cur_{s-1}, d pre_{s-1}
      } else if (valid && gl < C) {
        atomicAdd(gF + (long)node * C + gl, gp);  // pre_0 is the start row
      }
    }
    __syncthreads();
    if (s > 0) {  // through the momentum softmax of curve c: rows (0,c) and (1,c) of the [2,cn] array
      for (int p = wave; p < rounds; p += nw) {
        const int c = p * PER + sub, cc = c < cn ? c : cn - 1;
        const bool valid = c < cn;
        const long o = ((long)b * cn + cc) * a.L + s;
        const float M0 = a.mom[o * 2], M1 = a.mom[o * 2 + 1];
        const float gM0 = gMw[cc], gM1 = gMw[cn + cc];
        const float tt = M0 * gM0 + M1 * gM1, gz0 = M0 * (gM0 - tt), gz1 = M1 * (gM1 - tt);
        if (valid && gl < C) {
          s_Gc[cc * C + gl] += a.mw[gl] * gz0 + a.mw[2 * C + gl] * gz1;
          s_Gp[cc * C + gl] += a.mw[C + gl] * gz0 + a.mw[3 * C + gl] * gz1;
        }
      }
      // no barrier: the next step's first phase touches only this wave's own curves and the other gM parity
    }
  }
}

template <int C>
static int launch_walk(bool bwd, const WalkArgs& a, hipStream_t st) {
  constexpr int G = C <= 32 ? 32 : 64;   // two curves per wavefront when a candidate row fits half a wave
  const bool half = G == 32 && a.k <= 32;
  const int per = half ? 2 : 1, rounds = (a.cn + per - 1) / per;
  const int waves = rounds < kWalkThreads<C> / kWave ? rounds : kWalkThreads<C> / kWave;
  const dim3 grid((unsigned)a.B), block(waves * kWave);
  const size_t lds = bwd ? walk_bwd_lds(a.cn, C) : walk_fwd_lds(a.cn, C);
  PC3D_REQUIRE(lds <= 64 * 1024, "pc3d_curve_walk: curve_num * C = %d * %d does not fit the 64 KB LDS window", a.cn, C);
  if (bwd) {
    if (half) hipLaunchKernelGGL((curve_walk_bwd_kernel<C, G>), grid, block, lds, st, a);
    else hipLaunchKernelGGL((curve_walk_bwd_kernel<C, 64>), grid, block, lds, st, a);
  } else {
    if (half) hipLaunchKernelGGL((curve_walk_fwd_kernel<C, G>), grid, block, lds, st, a);
    else hipLaunchKernelGGL((curve_walk_fwd_kernel<C, 64>), grid, block, lds, st, a);
  }
  PC3D_LAUNCH_CHECK(bwd ? "pc3d_curve_walk_bwd_f32" : "pc3d_curve_walk_fwd_f32");
  return PC3D_OK;
}

static int walk_dispatch(bool bwd, const WalkArgs& a, int C, void* stream) {
  const char* nm = bwd ? "pc3d_curve_walk_bwd_f32" : "pc3d_curve_walk_fwd_f32";
  PC3D_REQUIRE(a.B > 0 && a.N > 0 && a.cn > 0 && a.L > 0, "%s: empty problem", nm);
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
                   nullptr, nullptr, nullptr};
  return pc3d::walk_dispatch(false, a, C, stream);
}

extern "C" int pc3d_curve_walk_bwd_f32(const float* gcurves, const float* feats, const int32_t* adj,
                                       const float* agent_w, const float* agent_b, const float* mom_w,
                                       const float* mom_b, int B, int N, int C, int k, int cn, int L,
                                       const float* curves, const int32_t* nodes, const int32_t* pick,
                                       const float* pre, const float* mom, float* gfeats, float* coef, void* stream) {
  PC3D_REQUIRE(gcurves && gfeats && coef, "pc3d_curve_walk_bwd_f32: null gradient pointer");
  pc3d::WalkArgs a{feats, adj, nullptr, agent_w, agent_b, mom_w, mom_b, B, N, k, cn, L,
                   const_cast<float*>(curves), const_cast<int32_t*>(nodes), const_cast<int32_t*>(pick),
                   const_cast<float*>(pre), const_cast<float*>(mom), gcurves, gfeats, coef};
  return pc3d::walk_dispatch(true, a, C, stream);
}
