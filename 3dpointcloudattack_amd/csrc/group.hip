// K5/K6/K7 — PointNet++ sampling and grouping ops (model/pointnet2_utils.py:60-155, model/curvenet_util.py:69-140):
//   * farthest-point sampling: the reference runs a Python loop of `npoint` steps x ~6 launches on [B,N] tensors;
//     here ONE workgroup per cloud keeps the cloud and the running min-distances on chip and does all `npoint`
//     dependent arg-max steps in one launch (registers + wave shuffles + one LDS exchange per step);
//   * ball query: the reference builds [B,S,N] distances, masks, SORTS the int64 index tensor along N and slices;
//     here one wave per centroid scans the cloud in index order and compacts the in-radius hits with a ballot /
//     prefix-popcount — "first nsample in ascending index order" without any sort or [B,S,N] tensor;
//   * group gather: xyz[idx]-centroid (+ features) written channels-last for the following 1x1-conv MLP, and its
//     backward (scatter-add into points and features — the attack's gradient path through the grouping).
#include "pc3d_common.h"

namespace pc3d {

// ---------------------------------------------------------------------------------------------------------
// FPS.  distance[i] = min(distance[i], |p_i - p_far|^2) ; far = argmax distance (lowest index on ties).
// Arithmetic = the reference's: ((dx*dx + dy*dy) + dz*dz) in fp32 without contraction, distance init 1e10.
// ---------------------------------------------------------------------------------------------------------
typedef float fps_f2 __attribute__((ext_vector_type(2)));

constexpr int FPS_T = 256;
constexpr int FPS_MAXPER = 32;  // points per thread -> N <= 8192

struct FpsArgs {
  PtsView x;
  int N, S;
  const int32_t* start;  // [B] or null (=0: model/curvenet_util.py:81)
  int32_t* out;          // [B,S]
#ifdef FPS_DIAG
  long long* diag;       // [B, waves, 4] clocks per phase (diagnostic build only: tools/exp/fps_full_diag.py)
#endif
};
#ifdef FPS_DIAG
static long long* g_fps_diag = nullptr;
#define FPS_STAMP(k)                                       \
  do {                                                     \
    const long long now_ = __builtin_amdgcn_s_memtime();   \
    dg[k] += now_ - t_;                                    \
    t_ = now_;                                             \
  } while (0)
#else
#define FPS_STAMP(k)
#endif

template <int PER, int T = FPS_T>
__global__ __launch_bounds__(T) void fps_kernel(FpsArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* sx = lds;                 // [N] staged coordinates (winner lookup)
  float* sy = lds + a.N;
  float* sz = lds + 2 * a.N;
  // one 8-byte key per wave and step parity: (value bits as a signed word : 0x7fffffff - index) — a signed 64-bit maximum
  // is the larger value, the lower index on a tie (values are running distances >= 0 or the -1 / -2 of padding, never NaN)
  __shared__ __attribute__((aligned(16))) long long red_k[2][T / 64 > 1 ? T / 64 : 1];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // A chain of dependent steps on one wavefront per SIMD, usually beside a chip-filling kernel of another stream (the
  // previous layer's MLP, the attack's searches): its instructions go first in the SIMD's issue arbitration.
#ifndef PC3D_FPS_NOPRIO
  __builtin_amdgcn_s_setprio(3);
#endif
  const float* xb = a.x.p + (int64_t)b * a.x.bs;
  float px[PER], py[PER], pz[PER], dist[PER];
#pragma unroll
  for (int e = 0; e < PER; ++e) {
    const int i = e * T + tid;  // strided ownership: coalesced loads, ascending index within a thread
    // Slots past the cloud hold COPIES of its last point: a copy has that point's running distance at every step and a
    // higher index, so the lowest-index tie rule never picks it (inside a lane, between lanes, between waves) — and every
    // running distance is then >= +0, which is what lets the arg-max below compare float bits as integers.
    const float* p = xb + (int64_t)(i < a.N ? i : a.N - 1) * a.x.ps;
    const float x = p[0], y = p[a.x.cs], z = p[2 * a.x.cs];
    if (i < a.N) sx[i] = x, sy[i] = y, sz[i] = z;
    px[e] = x, py[e] = y, pz[e] = z;
    dist[e] = 1e10f;
  }
  int far = a.start ? a.start[b] : 0;
  far = far < 0 ? 0 : (far >= a.N ? a.N - 1 : far);  // a start index outside the cloud cannot fault the launch
  __syncthreads();
  int keep = 0;
#ifdef FPS_DIAG
  long long dg[4] = {0, 0, 0, 0};
  long long t_ = __builtin_amdgcn_s_memtime();
#endif
  for (int s = 0; s < a.S; ++s) {
    // the picks leave 64 at a time: lane (s % 64) of wave 0 keeps step s's pick, a coalesced store every 64 steps
    keep = (lane == (s & 63)) ? far : keep;
    if ((s & 63) == 63 && wave == 0) a.out[(int64_t)b * a.S + (s - 63) + lane] = keep;
    const float cx = sx[far], cy = sy[far], cz = sz[far];
    // Per-lane arg-max on the BITS of the running distances (all >= +0, never NaN, so they order like the values; best
    // starts at -1): the float compare + two selects of round 3 (v_cmp_gt_f32_e64 into an SGPR pair, v_cndmask twice, each
    // two wait states behind it) were 30 of the 62 s_nop in the compiled step; an integer maximum and ONE select on VCC
    // remain. 707 -> 640 us for 4096 -> 1024. Ascending e inside the lane and the strict > keep the lowest index on ties.
    int bvi = -1;
    int bi = 0x7fffffff;
    // ROW PAIRS on packed fp32 (v_pk_add_f32 / v_pk_mul_f32: the same IEEE operations per component, in the reference's
    // order): a single wavefront issues an instruction every four clocks whatever its width, so a packed instruction is
    // two rows for the price of one — 8 per pair instead of the 12 the compiler's own SLP leaves (it packs two
    // coordinates of ONE row). 640 -> see DESIGN.md §3.9.
    constexpr int PP = PER / 2;
    const fps_f2 cX = {cx, cx}, cY = {cy, cy}, cZ = {cz, cz};
    // (hipcc turns the idiom below back into v_cmp_lt_i32 + v_cndmask on VCC, and that is the faster form: forced through
    // inline asm as sub / ashr / max / bfi the select becomes one more dependent link per row, 737 us against 640)
#define FPS_TRACK(eu, ndv)                                                                          \
  {                                                                                                 \
    const int ndi_ = __builtin_bit_cast(int, ndv);                                                  \
    const int gt_ = (bvi - ndi_) >> 31; /* -1: nd > best so far */                                  \
    bvi = bvi > ndi_ ? bvi : ndi_;                                                                  \
    bi = (gt_ & ((eu) * T + tid)) | (~gt_ & bi);                                                    \
  }
    // Two pairs per trip, written interleaved: 244 -> 230 us at 8 rows per lane (2048 -> 512), 105 -> 104 at 4, but 585 -> 595
    // at 16 (and 606 with the order forced through inline asm) — so only up to 8 rows per lane. (Stable to +-0.5 us.)
    constexpr int PQ = PER <= 8 ? (PP & ~1) : 0;            // pairs taken two at a time
#pragma unroll
    for (int q = 0; q < PQ; q += 2) {
      const int e = 2 * q, f = e + 2;
      const fps_f2 XA = {px[e], px[e + 1]}, YA = {py[e], py[e + 1]}, ZA = {pz[e], pz[e + 1]};
      const fps_f2 XB = {px[f], px[f + 1]}, YB = {py[f], py[f + 1]}, ZB = {pz[f], pz[f + 1]};
      const fps_f2 dxA = XA - cX, dxB = XB - cX, dyA = YA - cY, dyB = YB - cY, dzA = ZA - cZ, dzB = ZB - cZ;
      const fps_f2 sA = dxA * dxA, sB = dxB * dxB, tA = dyA * dyA, tB = dyB * dyB, uA = dzA * dzA, uB = dzB * dzB;
      const fps_f2 rA = sA + tA, rB = sB + tB;
      const fps_f2 dA = rA + uA, dB = rB + uB;
      float n0, n1, n2, n3;
      asm("v_min_f32_e32 %0, %1, %2" : "=v"(n0) : "v"(dA[0]), "v"(dist[e]));
      asm("v_min_f32_e32 %0, %1, %2" : "=v"(n1) : "v"(dA[1]), "v"(dist[e + 1]));
      asm("v_min_f32_e32 %0, %1, %2" : "=v"(n2) : "v"(dB[0]), "v"(dist[f]));
      asm("v_min_f32_e32 %0, %1, %2" : "=v"(n3) : "v"(dB[1]), "v"(dist[f + 1]));
      dist[e] = n0, dist[e + 1] = n1, dist[f] = n2, dist[f + 1] = n3;
      FPS_TRACK(e, n0) FPS_TRACK(e + 1, n1) FPS_TRACK(f, n2) FPS_TRACK(f + 1, n3)
    }
#pragma unroll
    for (int q = PQ; q < PP; ++q) {
      const int e = 2 * q;
      const fps_f2 X = {px[e], px[e + 1]}, Y = {py[e], py[e + 1]}, Z = {pz[e], pz[e + 1]};
      const fps_f2 dx = X - cX, dy = Y - cY, dz = Z - cZ;
      const fps_f2 d = (dx * dx + dy * dy) + dz * dz;
      float n0, n1;       // v_min_f32 returns its other operand for a quiet NaN: exactly (d < dist) ? d : dist, dist never NaN
      asm("v_min_f32_e32 %0, %1, %2" : "=v"(n0) : "v"(d[0]), "v"(dist[e]));
      asm("v_min_f32_e32 %0, %1, %2" : "=v"(n1) : "v"(d[1]), "v"(dist[e + 1]));
      dist[e] = n0, dist[e + 1] = n1;
      FPS_TRACK(e, n0) FPS_TRACK(e + 1, n1)
    }
#pragma unroll
    for (int e = 2 * PP; e < PER; ++e) {                  // (PER = 1)
      const float dx = px[e] - cx, dy = py[e] - cy, dz = pz[e] - cz;
      const float d = (dx * dx + dy * dy) + dz * dz;
      float nd;
      asm("v_min_f32_e32 %0, %1, %2" : "=v"(nd) : "v"(d), "v"(dist[e]));
      dist[e] = nd;
      FPS_TRACK(e, nd)
    }
#undef FPS_TRACK
    float bv = __builtin_bit_cast(float, bvi);
    FPS_STAMP(0);      // pick bookkeeping + centre lookup + row updates
    // wave arg-max on DPP (the shuffle version cost ~800 cycles of LDS-crossbar latency per step): the maximum
    // first, then the lowest index among the lanes that hold it (= the reference's first-index tie rule)
    {
      // the lanes that hold the maximum by ballot: ONE lane unless two points are exactly equidistant — then its index comes
      // by readlane, and only a tie takes the second DPP reduction
      const float wv = wave_max_chain(bv);     // (bv is a running distance or -2: never NaN)
      const unsigned long long tie = __builtin_amdgcn_ballot_w64(bv == wv);
      if (__builtin_popcountll(tie) == 1) bi = __builtin_amdgcn_readlane(bi, __builtin_ctzll(tie));
      else bi = wave_min_dpp_i32(bv == wv ? bi : 0x7fffffff);
      bv = wv;
    }
    FPS_STAMP(1);      // wave arg-max
    int fi = bi;
    if (T > 64) {           // (one wavefront per cloud: the wave's arg-max is the cloud's — no exchange, no barrier)
      const int buf = s & 1;  // double-buffered exchange: one barrier per step
      if (lane == 0)
        red_k[buf][wave] = ((long long)__builtin_bit_cast(int, bv) << 32) | (long long)(unsigned)(0x7fffffff - bi);
      __syncthreads();
      FPS_STAMP(2);    // key write + barrier
      long long best = red_k[buf][0];
#pragma unroll
      for (int w = 1; w < T / 64; ++w) {
        const long long o = red_k[buf][w];
        best = o > best ? o : best;
      }
      fi = 0x7fffffff - (int)(unsigned)(best & 0xffffffffll);
    }
    if (fi != 0x7fffffff) far = fi;  // no finite candidate (NaN cloud): stay put instead of indexing LDS at 2^31
    FPS_STAMP(3);      // key reads + selection
  }
#ifdef FPS_DIAG
  if (lane == 0 && a.diag)
    for (int k = 0; k < 4; ++k) a.diag[((int64_t)b * (T / 64) + wave) * 4 + k] = dg[k];
#endif
  if ((a.S & 63) != 0 && wave == 0 && lane < (a.S & 63)) a.out[(int64_t)b * a.S + (a.S & ~63) + lane] = keep;
}

// ---------------------------------------------------------------------------------------------------------
// Ball query: out[b,s,:] = first `nsample` indices i (ascending) with |xyz_i - c_s|^2 <= r2, padded with the first hit
// (model/pointnet2_utils.py:84-104). No hit at all -> N (what the reference's sort leaves there).
// ---------------------------------------------------------------------------------------------------------
struct BallArgs {
  PtsView x, c;
  int N, S, ns;
  float r2;
  int32_t* out;  // [B,S,ns]
};

__global__ __launch_bounds__(256) void ball_query_kernel(BallArgs a) {
  int bx_, by_;
  xcd_swizzle(bx_, by_);      // a cloud's workgroups on one XCD (pc3d_common.h)
  const int b = by_;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int s = bx_ * 4 + wave;
  if (s >= a.S) return;
  const float* cp = a.c.p + (int64_t)b * a.c.bs + (int64_t)s * a.c.ps;
  const float cx = cp[0], cy = cp[a.c.cs], cz = cp[2 * a.c.cs];
  const float* xb = a.x.p + (int64_t)b * a.x.bs;
  int32_t* o = a.out + ((int64_t)b * a.S + s) * a.ns;
  int found = 0, first = a.N;
  for (int i0 = 0; i0 < a.N && found < a.ns; i0 += 64) {
    const int i = i0 + lane;
    bool hit = false;
    if (i < a.N) {
      const float* p = xb + (int64_t)i * a.x.ps;
      const float dx = p[0] - cx, dy = p[a.x.cs] - cy, dz = p[2 * a.x.cs] - cz;
      hit = ((dx * dx + dy * dy) + dz * dz) <= a.r2;
    }
    const unsigned long long m = __builtin_amdgcn_ballot_w64(hit);
    if (m) {
      if (first == a.N) first = i0 + __builtin_ctzll(m);
      const int rank = __builtin_popcountll(m & ((1ull << lane) - 1ull));
      if (hit && found + rank < a.ns) o[found + rank] = i;
      found += __builtin_popcountll(m);
    }
  }
  if (found > a.ns) found = a.ns;
  for (int j = found + lane; j < a.ns; j += 64) o[j] = first;
}

// The same query with the roles turned round: a lane is a CENTRE (64 per workgroup), the four wavefronts of the
// workgroup each scan a quarter of the cloud in index order, 64 points at a time through a wave-private LDS buffer
// (the next 64 are in flight meanwhile). The point a wavefront looks at is the same for all its lanes — one broadcast
// LDS read — so a pair costs nine vector instructions with no cross-lane step, and "the first nsample in index order"
// is the scan order itself. Each (quarter, centre) keeps its hits in LDS (16-bit indices); the quarters are
// concatenated, cut at nsample and padded afterwards. Against a wavefront per centre (the kernel above: three strided
// loads, a ballot and a prefix count per 64 pairs, one trip per load) this is what CurveNet's first pooling level
// (B=32, N=4096, S=1024, r=0.05) needs: it sits right behind the 4096 -> 1024 sampling chain on the forward's
// critical path.
constexpr int BQ_SEG = 4;
static size_t ball_tile_lds(int ns) {
  return (size_t)BQ_SEG * 2 * 64 * sizeof(float4) + BQ_SEG * 64 * sizeof(int) + (size_t)BQ_SEG * 64 * ns * sizeof(uint16_t);
}

__global__ __launch_bounds__(BQ_SEG * 64) void ball_query_tile_kernel(BallArgs a) {
  extern __shared__ float4 bq_sm4[];
  float4* const stage = bq_sm4;                                              // [BQ_SEG][2][64] points
  int* const cnts = reinterpret_cast<int*>(bq_sm4 + BQ_SEG * 2 * 64);        // [BQ_SEG][64]
  uint16_t* const lists = reinterpret_cast<uint16_t*>(cnts + BQ_SEG * 64);   // [BQ_SEG][64][ns]
  const int b = blockIdx.y, lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int s = blockIdx.x * 64 + lane;
  const bool live = s < a.S;
  const float* cp = a.c.p + (int64_t)b * a.c.bs + (int64_t)(live ? s : a.S - 1) * a.c.ps;
  const float cx = cp[0], cy = cp[a.c.cs], cz = cp[2 * a.c.cs];
  const float* __restrict__ xb = a.x.p + (int64_t)b * a.x.bs;
  const int per = (a.N + BQ_SEG - 1) / BQ_SEG, i0 = wave * per, i1 = min(i0 + per, a.N);
  uint16_t* list = lists + (size_t)(wave * 64 + lane) * a.ns;
  int cnt = live ? 0 : a.ns;      // a lane without a centre counts as full: it must not hold up the early exit
  auto fetch = [&](int i) {
    float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < i1) {
      const float* p = xb + (int64_t)i * a.x.ps;
      q.x = p[0], q.y = p[a.x.cs], q.z = p[2 * a.x.cs];
    }
    return q;
  };
  auto test = [&](const float4 q, int i) {
    const float dx = q.x - cx, dy = q.y - cy, dz = q.z - cz;
    if (((dx * dx + dy * dy) + dz * dz) <= a.r2 && cnt < a.ns) list[cnt++] = (uint16_t)i;
  };
  float4 nxt = fetch(i0 + lane);
  for (int ib = i0, k = 0; ib < i1; ib += 64, ++k) {
    float4* buf = stage + (wave * 2 + (k & 1)) * 64;
    buf[lane] = nxt;
    nxt = fetch(ib + 64 + lane);
    wave_lds_sync();
    if (__builtin_amdgcn_ballot_w64(cnt < a.ns) == 0) break;   // every centre of this tile has its nsample hits
    if (ib + 64 <= i1) {
#pragma unroll 8
      for (int t = 0; t < 64; ++t) test(buf[t], ib + t);
    } else {
      for (int t = 0; t < i1 - ib; ++t) test(buf[t], ib + t);
    }
  }
  cnts[wave * 64 + lane] = live ? cnt : 0;
  __syncthreads();
  for (int cc = 0; cc < 64 / BQ_SEG; ++cc) {   // each wavefront writes the rows of 16 centres, slots across the lanes
    const int c = wave * (64 / BQ_SEG) + cc, s2 = blockIdx.x * 64 + c;
    if (s2 >= a.S) break;
    int n[BQ_SEG], first = a.N;
#pragma unroll
    for (int q = BQ_SEG - 1; q >= 0; --q) {
      n[q] = cnts[q * 64 + c];
      if (n[q] > 0) first = lists[(size_t)(q * 64 + c) * a.ns];
    }
    int32_t* o = a.out + ((int64_t)b * a.S + s2) * a.ns;
    for (int j = lane; j < a.ns; j += 64) {
      int v = first, r = j;
#pragma unroll
      for (int q = 0; q < BQ_SEG; ++q) {
        if (r >= 0 && r < n[q]) v = lists[(size_t)(q * 64 + c) * a.ns + r];
        r = r < n[q] ? -1 : r - n[q];
      }
      o[j] = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// Group gather, channels-last: out[b,s,j,0:3] = xyz[b,idx]-center[b,s] (center optional), out[b,s,j,3:3+D] = feat[b,idx]
// (model/pointnet2_utils.py:41-57,121-131). One thread per output element group of 4 channels.
// ---------------------------------------------------------------------------------------------------------
struct GatherArgs {
  PtsView x;            // may be null (features only)
  const float* feat;    // [B,N,D] row-major or null
  const int32_t* idx;   // [B,S,ns]
  PtsView center;       // [B,S] points or null
  int N, S, ns, D, C;   // C = (x ? 3 : 0) + D
  float* out;           // [B,S,ns,C]
};

__global__ __launch_bounds__(256) void group_gather_kernel(GatherArgs a) {
  int bx_, by_;
  xcd_swizzle(bx_, by_);      // a cloud's workgroups on one XCD (pc3d_common.h)
  const int64_t row = (int64_t)bx_ * 4 + (threadIdx.x >> 6);  // one wave per (b,s,j) row
  const int lane = threadIdx.x & 63;
  const int64_t rows = (int64_t)gridDim.y * a.S * a.ns;
  (void)rows;
  const int b = by_;
  if (row >= (int64_t)a.S * a.ns) return;
  const int s = (int)(row / a.ns);
  const int i = a.idx[((int64_t)b * a.S) * a.ns + row];
  // an index outside [0,N) (ball query's "no point in the ball" marker N, garbage from a NaN cloud) reads as a zero
  // row instead of faulting the GPU; the reference fails with a device assert there
  const bool ok = (unsigned)i < (unsigned)a.N;
  float* o = a.out + (((int64_t)b * a.S) * a.ns + row) * a.C;
  int c0 = 0;
  if (a.x.p) {
    if (lane < 3) {
      float v = ok ? a.x.p[(int64_t)b * a.x.bs + (int64_t)i * a.x.ps + lane * a.x.cs] : 0.f;
      if (ok && a.center.p) v -= a.center.p[(int64_t)b * a.center.bs + (int64_t)s * a.center.ps + lane * a.center.cs];
      o[lane] = v;
    }
    c0 = 3;
  }
  if (a.feat) {
    const float* f = a.feat + ((int64_t)b * a.N + (ok ? i : 0)) * a.D;
    for (int d = lane; d < a.D; d += 64) o[c0 + d] = ok ? f[d] : 0.f;
  }
}

// backward: grad_x[b, idx] += g[...,0:3]; grad_x[b, cidx[s]] -= sum_j g[...,0:3]; grad_feat[b, idx] += g[...,3:]
struct GatherBwdArgs {
  const float* g;          // [B,S,ns,C]
  const int32_t* idx;      // [B,S,ns]
  const int32_t* cidx;     // [B,S] index of the centre point in x, or null (no centre term)
  int N, S, ns, D, C, has_x;
  float* gx;               // [B,N,3] contiguous (zero-filled by the caller of the kernel) or null
  float* gf;               // [B,N,D] contiguous or null
};

// Workgroup = one group (b, s), its rows spread over the 4 waves. Ball query pads a group by repeating its FIRST index
// (56-65 % of all entries at SSG's sizes), and those repeats all land on one point: their rows are summed in registers
// (+ one LDS combine across the waves) and sent as ONE atomic per channel instead of one per row; the centre term
// (-sum over the whole group on xyz) is folded the same way.
constexpr int GGB_MAXC = 1024;   // channels whose padded-tail sum fits the LDS combine buffer

__global__ __launch_bounds__(256) void group_gather_bwd_kernel(GatherBwdArgs a) {
  __shared__ float s_tail[4][GGB_MAXC];
  __shared__ float s_ctr[4][4];
  int bx_, by_;
  xcd_swizzle(bx_, by_);      // a cloud's workgroups on one XCD (pc3d_common.h)
  const int s = bx_, b = by_;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int32_t* idx = a.idx + ((int64_t)b * a.S + s) * a.ns;
  const float* gbase = a.g + (((int64_t)b * a.S + s) * a.ns) * a.C;
  const int i0 = idx[0];
  const int c0 = a.has_x ? 3 : 0;
  const bool merge = a.C <= GGB_MAXC;
  // per-lane partial sums over this wave's padded rows, channel chunks of 64
  float tail[GGB_MAXC / 64];
#pragma unroll
  for (int q = 0; q < GGB_MAXC / 64; ++q) tail[q] = 0.f;
  float csum = 0.f;                                       // lane < 3: sum of g[..., lane] over this wave's rows
  const bool i0_ok = (unsigned)i0 < (unsigned)a.N;
  for (int j = wave; j < a.ns; j += 4) {
    const int i = idx[j];
    if ((unsigned)i >= (unsigned)a.N) continue;           // zero row in the forward: nothing flows back
    const float* g = gbase + (int64_t)j * a.C;
    if (a.has_x && a.gx && lane < 3) csum += g[lane];
    if (merge && j > 0 && i == i0) {                      // a repeat of the first index: accumulate, no atomics
#pragma unroll
      for (int q = 0; q < GGB_MAXC / 64; ++q)
        if (64 * q + lane < a.C) tail[q] += g[64 * q + lane];
      continue;
    }
    if (a.has_x && a.gx && lane < 3) atomicAdd(a.gx + ((int64_t)b * a.N + i) * 3 + lane, g[lane]);
    if (a.gf) {
      float* f = a.gf + ((int64_t)b * a.N + i) * a.D;
      for (int d = lane; d < a.D; d += 64) atomicAdd(f + d, g[c0 + d]);
    }
  }
  if (merge) {
#pragma unroll
    for (int q = 0; q < GGB_MAXC / 64; ++q)
      if (64 * q + lane < a.C) s_tail[wave][64 * q + lane] = tail[q];
  }
  if (lane < 3) s_ctr[wave][lane] = csum;
  __syncthreads();
  if (merge && i0_ok) {
    for (int c = threadIdx.x; c < a.C; c += 256) {
      const float v = (s_tail[0][c] + s_tail[1][c]) + (s_tail[2][c] + s_tail[3][c]);
      if (v == 0.f) continue;
      if (c < c0) {
        if (a.gx) atomicAdd(a.gx + ((int64_t)b * a.N + i0) * 3 + c, v);
      } else if (a.gf) {
        atomicAdd(a.gf + ((int64_t)b * a.N + i0) * a.D + (c - c0), v);
      }
    }
  }
  if (a.has_x && a.gx && a.cidx && threadIdx.x < 3 &&
      (unsigned)a.cidx[(int64_t)b * a.S + s] < (unsigned)a.N) {
    const float v = (s_ctr[0][threadIdx.x] + s_ctr[1][threadIdx.x]) + (s_ctr[2][threadIdx.x] + s_ctr[3][threadIdx.x]);
    atomicAdd(a.gx + ((int64_t)b * a.N + a.cidx[(int64_t)b * a.S + s]) * 3 + threadIdx.x, -v);
  }
}

// ---------------------------------------------------------------------------------------------------------
// First layer of a set-abstraction MLP WITHOUT the grouped input tensor. The layer is linear in [x_j - c_s ; f_j], so
//   W1 [x_j - c_s ; f_j] + b1 = (Wx x_j + Wf f_j) - Wx c_s + b1 = P[idx[s,j]] + Bc[s]
// with ONE product per POINT (P = [x | f] W1^T: B*N rows) instead of one per grouped row (B*S*ns rows: 16x more at
// SSG's second layer, and K = 131 / 259 there is not even a multiple of 4), and
//   group_act : H[b,s,j,:] = act(P[b,idx[b,s,j],:] + Bc[b,s,:])
// is the gather the layer needed anyway, now emitting the layer-1 OUTPUT (model/pointnet2_utils.py:118-135,190-197).
// Backward: dP[idx] += g', dBc[s] = sum_j g' with g' = act'(H) g; like group_gather_bwd, the rows that repeat the
// group's first index (ball query's padding, 56-65 % of all entries) are summed on chip and sent as one atomic per
// channel. An index outside [0, NA) reads as a zero row of P and receives no gradient.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void group_act_fwd_kernel(const float4* __restrict__ P, const float4* __restrict__ Bc,
                                                            const int* __restrict__ idx, int NA, int S, int K, int C4,
                                                            float slope, float4* __restrict__ H, int64_t total) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const int c4 = (int)(t % C4);
  const int64_t e = t / C4;            // (b, s, j)
  const int64_t bs = e / K;            // (b, s)
  const int64_t b = bs / S;
  const int i = idx[e];
  const float4 c = Bc[bs * C4 + c4];
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  if ((unsigned)i < (unsigned)NA) a = P[(b * NA + i) * C4 + c4];
  float4 v = make_float4(a.x + c.x, a.y + c.y, a.z + c.z, a.w + c.w);
  v.x = v.x > 0.f ? v.x : v.x * slope, v.y = v.y > 0.f ? v.y : v.y * slope;
  v.z = v.z > 0.f ? v.z : v.z * slope, v.w = v.w > 0.f ? v.w : v.w * slope;
  H[t] = v;
}

constexpr int GAB_MAXC = 512;   // channels (<= 8 per lane)

// H == null: the activation's sign comes from the bit mask pc3d_gemm_nt_gather_f32 wrote while it generated the layer-1
// output on load (that output was never stored): C/4 bytes per row instead of a second [B,S,K,C] stream. (Recomputing
// the sign from P[idx] + Bc was measured first: 270 us against 179 us with H at SSG's SA1 — the row gather from the
// 33 MB P misses the L2 where the H stream did not.)
template <bool SCATTER>   // false: only gBc (the deterministic path scatters gP in a second, ordered launch: det.hip)
__global__ __launch_bounds__(256) void group_act_bwd_kernel(const float* __restrict__ gH, const float* __restrict__ H,
                                                            const int* __restrict__ idx, int NA, int S, int K, int C,
                                                            float slope, float* __restrict__ gP,
                                                            float* __restrict__ gBc, const uint8_t* __restrict__ mask) {
  extern __shared__ float gab_lds[];            // [2][4][C]: per-wave group sums and padded-tail sums
  int bx_, by_;
  xcd_swizzle(bx_, by_);      // a cloud's workgroups on one XCD (pc3d_common.h)
  const int s = bx_, b = by_;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t g0 = ((int64_t)b * S + s) * K;
  const int* id = idx + g0;
  const int i0 = id[0];
  float tot[GAB_MAXC / 64], tail[GAB_MAXC / 64];
#pragma unroll
  for (int q = 0; q < GAB_MAXC / 64; ++q) tot[q] = 0.f, tail[q] = 0.f;
  for (int j = wave; j < K; j += 4) {
    const int i = id[j];
    const float* g = gH + (g0 + j) * C;
    const bool ok = (unsigned)i < (unsigned)NA;
    const float* h = H + (g0 + j) * C;
    const uint8_t* mrow = mask + (g0 + j) * (C >> 2);
    const bool rep = j > 0 && i == i0;          // wave-uniform
    float* dst = gP + ((int64_t)b * NA + (ok ? i : 0)) * C;
#pragma unroll
    for (int q = 0; q < GAB_MAXC / 64; ++q) {
      const int c = 64 * q + lane;
      if (c < C) {
        const bool pos = H ? h[c] > 0.f : ((mrow[c >> 2] >> (c & 3)) & 1) != 0;
        const float v = pos ? g[c] : g[c] * slope;
        tot[q] += v;
        if (!SCATTER) continue;
        if (rep) tail[q] += v;
        else if (ok) atomicAdd(dst + c, v);
      }
    }
  }
  float* s_tot = gab_lds;
  float* s_tail = gab_lds + 4 * C;
#pragma unroll
  for (int q = 0; q < GAB_MAXC / 64; ++q) {
    const int c = 64 * q + lane;
    if (c < C) s_tot[wave * C + c] = tot[q], s_tail[wave * C + c] = tail[q];
  }
  __syncthreads();
  const bool i0_ok = (unsigned)i0 < (unsigned)NA;
  for (int c = threadIdx.x; c < C; c += 256) {
    gBc[((int64_t)b * S + s) * C + c] = (s_tot[c] + s_tot[C + c]) + (s_tot[2 * C + c] + s_tot[3 * C + c]);
    const float v = (s_tail[c] + s_tail[C + c]) + (s_tail[2 * C + c] + s_tail[3 * C + c]);
    if (SCATTER && v != 0.f && i0_ok) atomicAdd(gP + ((int64_t)b * NA + i0) * C + c, v);
  }
}

// ---------------------------------------------------------------------------------------------------------
// Backward of  out[g,c] = max_r relu(x[g,r,:] . W[c,:] + b[c])  (the last 1x1 conv + ReLU + max over the group of a
// PointNet++ set-abstraction layer, model/pointnet2_utils.py:190-197) to x. `max` hands each channel's gradient to ONE
// row of its group, so dL/dx[g,r,:] = sum_{c: arg[g,c]==r, out[g,c]>0} gout[g,c] W[c,:] is a sparse row accumulation
// (C3 rows of W per group) instead of the dense [ns x C3] x [C3 x C2] product autograd runs on a tensor that is zero
// except for one entry per (group, channel). Workgroup = (group, block of 128 input channels).
// ---------------------------------------------------------------------------------------------------------
constexpr int GMB_T = 128;       // threads = input-channel columns handled by a workgroup
constexpr int GMB_MAXNS = 128;   // rows per group

struct GroupMaxBwdArgs {
  const float* gout;    // [G,C3]
  const float* out;     // [G,C3] forward result (post-ReLU max): gradient flows where it is > 0
  const int64_t* arg;   // [G,C3] winning row inside the group (torch.max indices)
  const float* W;       // [C3,C2]
  int ns, C2, C3;
  float* gx;            // [G,ns,C2]
  const float* xin;     // [G,ns,C2] or null: the operator's input when it is itself a ReLU output — gx is then
                        // zeroed where xin <= 0, i.e. the previous layer's ReLU backward is applied on the way out
  const uint32_t* xmask = nullptr;   // [G*ns, C2/32] or null: the same signs as bits (bit k % 32 of word k / 32 of a row),
                                     // as pc3d_gemm_nt_gather_f32 writes them — 1/32 of the bytes of xin
  uint32_t* amask = nullptr;         // [G, ceil(ns/32)] or null: bit j = "row j of the group won at least one channel". With
                                     // it, ONLY those rows of gx are written (the others are all zero and nobody reads them:
                                     // 58-65 % of the rows at SSG's levels — the ball query pads a group with copies of its
                                     // first point, and a copy never wins the max)
};

// Thread = one input channel k of one group; its NS row accumulators live in REGISTERS and are addressed with the
// winning row of each channel, which is the same for every thread (a scalar load of arg[g,c]) — so the compiler indexes
// the register file through M0 (v_movrel), and a channel costs one coalesced weight load + one FMA: no LDS, no
// barrier, no atomics; channels are walked in ascending order => deterministic.
using gmb_f32x32 = __attribute__((ext_vector_type(32))) float;
using gmb_f32x16 = __attribute__((ext_vector_type(16))) float;

// KS > 1 (few groups: a classifier's group-all layer has ONE group per cloud and 1024 channels to walk): the channels are
// split over KS thread groups of the workgroup (threadIdx.y), each with its own accumulators; the partial sums are added
// in chunk order through LDS, one 32-row register vector at a time. The order of a row's sum is then "chunk by chunk" —
// fixed for a given KS, which the HOST chooses from the per-cloud shape (ops.gmb_ksplit), never from the batch.
template <int NS, int KS = 1>
__global__ __launch_bounds__(GMB_T * KS) void group_max_linear_bwd_kernel(GroupMaxBwdArgs a) {
  constexpr int NV = (NS + 31) / 32;          // accumulators as 32-wide register vectors: a uniform dynamic index into
  const int g = blockIdx.x;                    // one of those lowers to M0-relative register addressing (v_movrel)
  const int q = KS > 1 ? threadIdx.y : 0;      // channel chunk of this thread group
  const int tlin = threadIdx.y * blockDim.x + threadIdx.x, nthr = blockDim.x * blockDim.y;
  const int k = blockIdx.y * blockDim.x + threadIdx.x;
  const bool live = k < a.C2;
  const float* wcol = a.W + (live ? k : 0);
  const int64_t base = (int64_t)g * a.C3;
  gmb_f32x32 acc[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v)
#pragma unroll
    for (int e = 0; e < 32; ++e) acc[v][e] = 0.f;
  // the group's (row, masked gradient) pairs: one coalesced round trip into LDS, then read 8 channels ahead so the LDS
  // latency of the register index is paid once per 8 channels, not per channel
  extern __shared__ float gmb_lds[];
  float* s_g = gmb_lds;                                 // [C3]
  int* s_r = reinterpret_cast<int*>(gmb_lds + a.C3);    // [C3]
  __shared__ uint32_t s_act[4];
  if (tlin < 4) s_act[tlin] = 0u;
  __syncthreads();
  for (int c = tlin; c < a.C3; c += nthr) {
    s_g[c] = (a.out[base + c] > 0.f) ? a.gout[base + c] : 0.f;
    const int rw = (int)a.arg[base + c];
    s_r[c] = rw;
    if (a.amask) atomicOr(&s_act[(rw >> 5) & 3], 1u << (rw & 31));
  }
  __syncthreads();
  uint32_t act[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) act[v] = a.amask ? s_act[v] : 0xffffffffu;
  if (a.amask && blockIdx.y == 0 && tlin < (a.ns + 31) / 32) a.amask[(int64_t)g * ((a.ns + 31) / 32) + tlin] = s_act[tlin];
  const int cper = a.C3 / KS, cend = (q + 1) * cper;   // (the entry point checks C3 % (8 KS) == 0 for KS > 1)
  int c = q * cper;
  for (; c + 8 <= cend; c += 8) {
    int rr[8];
    float gg[8], ww[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      rr[e] = s_r[c + e];
      gg[e] = s_g[c + e];
      ww[e] = wcol[(int64_t)(c + e) * a.C2];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int r = __builtin_amdgcn_readfirstlane(rr[e]);
      const int hi = r >> 5, lo = r & 31;
#pragma unroll
      for (int v = 0; v < NV; ++v)
        if (hi == v) acc[v][lo] = __builtin_fmaf(gg[e], ww[e], acc[v][lo]);
    }
  }
  for (; c < cend; ++c) {
    const int r = __builtin_amdgcn_readfirstlane(s_r[c]);
    const float gv = s_g[c], w = wcol[(int64_t)c * a.C2];
    const int hi = r >> 5, lo = r & 31;
#pragma unroll
    for (int v = 0; v < NV; ++v)
      if (hi == v) acc[v][lo] = __builtin_fmaf(gv, w, acc[v][lo]);
  }
  if (KS > 1) {
    float* comb = gmb_lds + 2 * a.C3;                    // [KS - 1][32][blockDim.x]
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      if (q > 0)
#pragma unroll
        for (int e = 0; e < 32; ++e) comb[((q - 1) * 32 + e) * blockDim.x + threadIdx.x] = acc[v][e];
      __syncthreads();
      if (q == 0)
#pragma unroll
        for (int e = 0; e < 32; ++e)
#pragma unroll
          for (int p = 0; p < KS - 1; ++p) acc[v][e] += comb[(p * 32 + e) * blockDim.x + threadIdx.x];
      __syncthreads();
    }
    if (q > 0) return;
  }
  if (a.xmask) {
    // sign bits instead of the stored activation: the wave's 32 rows x 2 words of a register vector are ONE load (lane l
    // takes row l & 31, word column l >> 5 of the wave's two), and row e's word reaches every lane through v_readlane —
    // a thread-per-word load per row (32 broadcast loads per thread) made this kernel 30 us slower than reading xin
    const int lane = threadIdx.x & 63;
    const int kw = (blockIdx.y * blockDim.x + (threadIdx.x & ~63)) >> 5;     // first word column of this wave
    const int wpr = a.C2 >> 5;                                               // words per row
    float* o = a.gx + (int64_t)g * a.ns * a.C2 + k;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int wrow = 32 * v + (lane & 31), wcol = kw + (lane >> 5);
      const uint32_t word = (wrow < a.ns && wcol < wpr) ? a.xmask[((int64_t)g * a.ns + wrow) * wpr + wcol] : 0u;
#pragma unroll
      for (int e = 0; e < 32; ++e) {
        const uint32_t lo = __builtin_amdgcn_readlane(word, e), hi = __builtin_amdgcn_readlane(word, 32 + e);
        const uint32_t w = (lane & 32) ? hi : lo;
        if (live && 32 * v + e < a.ns && ((act[v] >> e) & 1u)) o[(int64_t)(32 * v + e) * a.C2] = ((w >> (k & 31)) & 1u) ? acc[v][e] : 0.f;
      }
    }
    return;
  }
  if (live) {
    float* o = a.gx + (int64_t)g * a.ns * a.C2 + k;
    const float* xi = a.xin ? a.xin + (int64_t)g * a.ns * a.C2 + k : nullptr;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      float m[32];
      if (xi) {   // all 32 mask loads of this register vector in flight before the first store
#pragma unroll
        for (int e = 0; e < 32; ++e) m[e] = (32 * v + e < a.ns) ? xi[(int64_t)(32 * v + e) * a.C2] : 0.f;
      }
#pragma unroll
      for (int e = 0; e < 32; ++e)
        if (32 * v + e < a.ns) o[(int64_t)(32 * v + e) * a.C2] = (xi && !(m[e] > 0.f)) ? 0.f : acc[v][e];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// The same backward with the group's channels SORTED BY WINNING ROW first (a counting sort in LDS by the wavefront that
// owns the group): a row's gradient is then a plain register sum over a contiguous run of (channel, gradient) pairs and
// leaves in one store — no register array addressed through M0 (the form above spends ~15 instructions per channel, ten of
// them on the indexed read-modify-write), channels whose masked gradient is zero drop out (a third to a half of them: the
// max was not positive), and only rows that won something are visited. Order: the slots of the sort are handed out by
// ds_add_rtn of ONE wavefront in channel order (chunks of 64, inside a chunk the hardware's fixed lane order), so a row's
// channels are summed in one order for given inputs.
// Workgroup = one wavefront = (group, 64 input channels); needs the sign-bit mask form (xmask) and writes amask.
// ---------------------------------------------------------------------------------------------------------
template <int NS>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(6, 8))) void group_max_bwd_csr_kernel(GroupMaxBwdArgs a) {
  extern __shared__ float gmc_lds[];
  float2* s_e = reinterpret_cast<float2*>(gmc_lds);              // [C3] (channel | row << 16, gradient), sorted by row
  int* s_off = reinterpret_cast<int*>(s_e + 2 * a.C3);           // [NS + 1] row offsets
  int* s_cur = s_off + NS + 1;                                    // [NS] fill cursors
  const int g = blockIdx.x, lane = threadIdx.x;
  const int k = blockIdx.y * 64 + lane;
  const bool live = k < a.C2;
  const int64_t base = (int64_t)g * a.C3;
  for (int j = lane; j <= NS; j += 64) s_off[j] = 0;
  wave_lds_sync();
  // pass 1: the group's kept (row, gradient) pairs, parked unsorted in LDS; row histogram
  float2* s_t = s_e + a.C3;                                       // [C3] parking area (behind the sorted list)
  for (int c0 = 0; c0 < a.C3; c0 += 64) {
    const int c = c0 + lane;
    float x = 0.f;
    int rw = -1;
    if (c < a.C3) {
      x = (a.out[base + c] > 0.f) ? a.gout[base + c] : 0.f;
      if (x != 0.f) {
        rw = (int)a.arg[base + c] & (NS - 1);
        atomicAdd(&s_off[rw + 1], 1);
      }
      s_t[c] = make_float2(__builtin_bit_cast(float, rw), x);
    }
  }
  wave_lds_sync();
  {   // inclusive scan of s_off[1 .. NS] (NS <= 128: two bins per lane)
    const int e0 = 2 * lane + 1, e1 = 2 * lane + 2;
    const int v0 = e0 <= NS ? s_off[e0] : 0, v1 = e1 <= NS ? s_off[e1] : 0;
    int sc = v0 + v1;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int o = __shfl_up(sc, d, 64);
      if (lane >= d) sc += o;
    }
    const int excl = sc - (v0 + v1);
    wave_lds_sync();
    if (e0 <= NS) s_off[e0] = excl + v0, s_cur[e0 - 1] = excl;
    if (e1 <= NS) s_off[e1] = excl + v0 + v1, s_cur[e1 - 1] = excl + v0;
  }
  wave_lds_sync();
  // pass 2: slots in channel order
  for (int c0 = 0; c0 < a.C3; c0 += 64) {
    const int c = c0 + lane;
    if (c < a.C3) {
      const float2 pr = s_t[c];
      const int rw = __builtin_bit_cast(int, pr.x);
      if (rw >= 0) {
        const int slot = atomicAdd(&s_cur[rw], 1);
        s_e[slot] = make_float2(__builtin_bit_cast(float, c | (rw << 16)), pr.y);
      }
    }
  }
  wave_lds_sync();
  const int nent = s_off[NS];
  // rows that won something; the first wavefront column of the group publishes them
  constexpr int NW = NS / 32;
  uint32_t act[NW];
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    const int j = 32 * w + (lane & 31);
    const unsigned long long bal = __builtin_amdgcn_ballot_w64(lane < 32 && s_off[j + 1] > s_off[j]);
    act[w] = (uint32_t)bal;
  }
  if (a.amask && blockIdx.y == 0 && lane < NW) a.amask[(int64_t)g * NW + lane] = lane == 0 ? act[0] : (lane == 1 ? act[NW > 1 ? 1 : 0] : (lane == 2 ? act[NW > 2 ? 2 : 0] : act[NW > 3 ? 3 : 0]));
  // the rows' layer-2 sign words for this wavefront's 64 columns: lane l holds the word of row (l & 31) + 32 w, column half l >> 5
  const int wpr = a.C2 >> 5, kw = (blockIdx.y * 64) >> 5;
  uint32_t mword[NW];
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    const int wrow = 32 * w + (lane & 31), wcol = kw + (lane >> 5);
    mword[w] = (wrow < a.ns && wcol < wpr) ? a.xmask[((int64_t)g * a.ns + wrow) * wpr + wcol] : 0u;
  }
  const float* wcol = a.W + (live ? k : 0);
  float* o = a.gx + (int64_t)g * a.ns * a.C2 + k;
  auto flush = [&](int row, float acc) {          // row: uniform
    uint32_t lo = 0u, hi = 0u;
#pragma unroll
    for (int w = 0; w < NW; ++w)
      if ((row >> 5) == w) lo = __builtin_amdgcn_readlane(mword[w], row & 31), hi = __builtin_amdgcn_readlane(mword[w], 32 + (row & 31));
    const uint32_t wd = (lane & 32) ? hi : lo;
    if (live && row < a.ns) o[(int64_t)row * a.C2] = ((wd >> (k & 31)) & 1u) ? acc : 0.f;
  };
  float acc = 0.f;
  int cur_row = -1;
  for (int t0 = 0; t0 < nent; t0 += 8) {
    float2 e2[8];
    float ww[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int t = t0 + u < nent ? t0 + u : nent - 1;
      e2[u] = s_e[t];
      const int code = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, e2[u].x));
      ww[u] = wcol[(int64_t)(code & 0xffff) * a.C2];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (t0 + u < nent) {
        const int row = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, e2[u].x)) >> 16;
        if (row != cur_row) {
          if (cur_row >= 0) flush(cur_row, acc);
          acc = 0.f, cur_row = row;
        }
        acc = __builtin_fmaf(e2[u].y, ww[u], acc);
      }
    }
  }
  if (cur_row >= 0) flush(cur_row, acc);
}

// ---------------------------------------------------------------------------------------------------------
// Forward of the same operator: out[g,c] = max_r relu(x[g,r,:] . W[c,:] + b[c]), arg[g,c] = winning row — the last
// 1x1 conv + ReLU + max over the group of a set-abstraction layer (model/pointnet2_utils.py:190-197) WITHOUT writing
// the [G*ns, C3] activation (537 MB per layer at SSG's B=64, N=2048) and reading it back for the max.
// Workgroup = group g; wave w owns rows 32w..32w+31 of it: its A operands (32 rows x C2) are loaded once into
// registers and every 32-column block of W streams past them on v_mfma_f32_32x32x2_f32 (rows on the MFMA row index, so
// the max over the wave's rows is an in-register reduction + one cross-half shuffle, like the PointNet tower);
// several waves (ns > 32) combine through LDS in ascending row order (lowest row wins ties, as torch.max).
// ---------------------------------------------------------------------------------------------------------
using glm_f32x16 = __attribute__((ext_vector_type(16))) float;

struct GroupLinMaxArgs {
  const float* x;      // [G,ns,C2]
  const float* W;      // [C3,C2]
  const float* b;      // [C3]
  int ns, C2, C3;
  float* out;          // [G,C3]
  int64_t* arg;        // [G,C3]
};

template <int NT>     // float4 per lane along K: C2 <= 8 * NT
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void group_linear_max_kernel(GroupLinMaxArgs a) {
  extern __shared__ float glm_lds[];           // [nw][C3] values + [nw][C3] rows (only when more than one wave)
  const int g = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int nt = a.C2 / 8;
  const int r0 = 32 * wave;
  const int row = (r0 + r < a.ns) ? r0 + r : a.ns - 1;              // clamped: masked out of the max below
  const float* xr = a.x + ((int64_t)g * a.ns + row) * a.C2 + 4 * h;
  float4 av[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) av[t] = (t < nt) ? *reinterpret_cast<const float4*>(xr + 8 * t) : make_float4(0.f, 0.f, 0.f, 0.f);
  // W streams from L2 one float4 per lane per 4 MFMAs; nothing of it is kept: the small register footprint (about 4
  // waves per SIMD) is what hides the load latency here (double-buffering whole blocks in registers measured slower)
  const int ncb = a.C3 / 32;
  for (int cb = 0; cb < ncb; ++cb) {
    const float* wr = a.W + (int64_t)(cb * 32 + r) * a.C2 + 4 * h;
    glm_f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (t < nt) {   // D[row = group row][col = output channel]
        const float4 bw = *reinterpret_cast<const float4*>(wr + 8 * t);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t].x, bw.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t].y, bw.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t].z, bw.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t].w, bw.w, acc, 0, 0, 0);
      }
    }
    float best = -__builtin_inff();
    int bi = r0;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int rr = r0 + (e & 3) + 8 * (e >> 2) + 4 * h;          // ascending in e for fixed h
      if (rr < a.ns && acc[e] > best) best = acc[e], bi = rr;
    }
    argmax_xor32(best, bi);
    const int ch = cb * 32 + r;
    if (nw == 1) {
      if (h == 0) {
        a.out[(int64_t)g * a.C3 + ch] = fmaxf(best + a.b[ch], 0.f);
        a.arg[(int64_t)g * a.C3 + ch] = bi;
      }
    } else if (h == 0) {
      glm_lds[wave * a.C3 + ch] = best;
      reinterpret_cast<int*>(glm_lds + nw * a.C3)[wave * a.C3 + ch] = bi;
    }
  }
  if (nw > 1) {
    __syncthreads();
    const int* pi = reinterpret_cast<const int*>(glm_lds + nw * a.C3);
    for (int ch = threadIdx.x; ch < a.C3; ch += blockDim.x) {
      float best = glm_lds[ch];
      int bi = pi[ch];
      for (int w = 1; w < nw; ++w) {
        const float v = glm_lds[w * a.C3 + ch];
        if (v > best) best = v, bi = pi[w * a.C3 + ch];           // ascending wave = ascending rows: strict >
      }
      a.out[(int64_t)g * a.C3 + ch] = fmaxf(best + a.b[ch], 0.f);
      a.arg[(int64_t)g * a.C3 + ch] = bi;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// group_act backward WITHOUT float atomics: gather over a reverse index of the grouping.
// The scatter above costs 270 us per set-abstraction level at SSG's sizes (B=64: 1.05 M rows of 64 / 0.5 M rows of 128
// channels, one row-wide atomic per non-padded row): 12 % of the KNN-attack iteration. The grouping indices are known as
// soon as the ball query has run, long before the backward — so the forward's geometry chain also builds, per cloud, the
// list of rows that reference each point (a counting sort: two passes of integer atomics + a scan, ~30 us on the side
// stream), and the backward becomes
//   groups pass : per group, dBc = sum of the masked rows and `tail` = sum of the rows that repeat the group's first
//                 index (the ball query's padding, 56-65 % of the entries) — sequential reads, no atomics;
//   points pass : per point, the sum of the masked rows in its list plus the tails of the groups it leads — every dP
//                 row is written exactly once (no zero fill).
// List entry codes: s * K + j for row (s, j) of the cloud, S * K + s for the tail of group s.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void group_rev_count_kernel(const int* __restrict__ idx, int NA, int S, int K,
                                                              int* __restrict__ cnt, int64_t total) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int j = (int)(e % K);
  const int64_t bs = e / K;
  const int b = (int)(bs / S);
  const int i = idx[e], i0 = idx[e - j];
  if ((unsigned)i >= (unsigned)NA) return;
  if (j == 0) {
    atomicAdd(cnt + (int64_t)b * NA + i, K > 1 ? 2 : 1);      // the row itself + the group's tail entry
  } else if (i != i0) {
    atomicAdd(cnt + (int64_t)b * NA + i, 1);
  }
}

// exclusive scan of one cloud's counters -> off [NA + 1]; the counters are zeroed (they become the fill cursors)
__global__ __launch_bounds__(256) void group_rev_scan_kernel(int* __restrict__ cnt, int NA, int* __restrict__ off) {
  __shared__ int part[256];
  const int b = blockIdx.x, t = threadIdx.x;
  int* c = cnt + (int64_t)b * NA;
  int* o = off + (int64_t)b * (NA + 1);
  const int per = (NA + 255) / 256, lo = t * per, hi = min(lo + per, NA);
  int s = 0;
  for (int i = lo; i < hi; ++i) s += c[i];
  part[t] = s;
  __syncthreads();
  for (int d = 1; d < 256; d <<= 1) {          // Hillis-Steele inclusive scan of the 256 partial sums
    const int v = t >= d ? part[t - d] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  int run = part[t] - s;                        // exclusive prefix of this thread's range
  for (int i = lo; i < hi; ++i) {
    const int v = c[i];
    o[i] = run;
    run += v;
    c[i] = 0;
  }
  if (t == 255) o[NA] = part[255];
}

__global__ __launch_bounds__(256) void group_rev_fill_kernel(const int* __restrict__ idx, int NA, int S, int K,
                                                             int* __restrict__ cur, const int* __restrict__ off,
                                                             int* __restrict__ lst, int64_t L, int64_t total) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int j = (int)(e % K);
  const int64_t bs = e / K;
  const int b = (int)(bs / S), s = (int)(bs - (int64_t)b * S);
  const int i = idx[e], i0 = idx[e - j];
  if ((unsigned)i >= (unsigned)NA) return;
  const int* o = off + (int64_t)b * (NA + 1);
  int* l = lst + (int64_t)b * L;
  if (j == 0) {
    const int slot = atomicAdd(cur + (int64_t)b * NA + i, K > 1 ? 2 : 1);
    l[o[i] + slot] = s * K;
    if (K > 1) l[o[i] + slot + 1] = S * K + s;
  } else if (i != i0) {
    const int slot = atomicAdd(cur + (int64_t)b * NA + i, 1);
    l[o[i] + slot] = s * K + j;
  }
}

// groups pass: dBc and the padded-tail sums (the scatter kernel above minus its atomics). Q = channels per lane
// (C <= 64 Q). A WAVE owns a group — no LDS, no barrier — and loads its rows EIGHT AT A TIME before any is used: with one
// row per loop trip (as in the scatter kernel) every trip waits out a full memory latency, and a workgroup per group of 32
// rows is 8 KB of work per workgroup; in that form the pass took as long as the scatter with its atomics (228 us), with
// four rows in flight per wave 159 us.
template <int Q>
__global__ __launch_bounds__(256) void group_act_bwd_groups_kernel(const float* __restrict__ gH, const float* __restrict__ H,
                                                                   const uint8_t* __restrict__ mask,
                                                                   const int* __restrict__ idx, int S, int K, int C,
                                                                   float slope, float* __restrict__ gBc,
                                                                   float* __restrict__ tail_out) {
  int bx_, by_;
  xcd_swizzle(bx_, by_);      // a cloud's workgroups on one XCD (pc3d_common.h)
  const int b = by_, lane = threadIdx.x & 63;
  const int s = bx_ * 4 + (threadIdx.x >> 6);
  if (s >= S) return;
  const int64_t g0 = ((int64_t)b * S + s) * K;
  const int* id = idx + g0;
  const int i0 = id[0];
  float tot[Q], tail[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) tot[q] = 0.f, tail[q] = 0.f;
  constexpr int U = Q == 1 ? 8 : 4;             // rows in flight (Q = 2 with 8: 140 us against 124 at SSG SA2)
  for (int j0 = 0; j0 < K; j0 += U) {
    float gv[U][Q];
    bool pos[U][Q], rep[U], live[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int j = j0 + u;
      live[u] = j < K;
      const int jj = live[u] ? j : j0;
      rep[u] = jj > 0 && id[jj] == i0;          // wave-uniform
      const float* g = gH + (g0 + jj) * C;
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        const int c = 64 * q + lane;
        gv[u][q] = c < C ? g[c] : 0.f;
        if (H) pos[u][q] = c < C && H[(g0 + jj) * C + c] > 0.f;
        else pos[u][q] = c < C && ((mask[(g0 + jj) * (C >> 2) + (c >> 2)] >> (c & 3)) & 1) != 0;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (!live[u]) continue;
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        const float v = pos[u][q] ? gv[u][q] : gv[u][q] * slope;
        tot[q] += v;
        if (rep[u]) tail[q] += v;
      }
    }
  }
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const int c = 64 * q + lane;
    if (c < C) {
      const int64_t o = ((int64_t)b * S + s) * C + c;
      gBc[o] = tot[q];
      tail_out[o] = tail[q];
    }
  }
}

// points pass: a wave per point walks the point's list, four entries in flight
template <int Q>
__global__ __launch_bounds__(256) void group_act_bwd_points_kernel(const float* __restrict__ gH, const float* __restrict__ H,
                                                                   const uint8_t* __restrict__ mask,
                                                                   const float* __restrict__ tail, const int* __restrict__ off,
                                                                   const int* __restrict__ lst, int64_t L, int NA, int S,
                                                                   int K, int C, float slope, float* __restrict__ gP,
                                                                   const uint32_t* __restrict__ amask = nullptr) {
  // amask [B*S, ceil(K/32)] or null: bit j of group s = "row (s, j) of gH was written"; the other rows are all zero and
  // are skipped (their memory is not even defined)
  int bx_, by_;
  xcd_swizzle(bx_, by_);      // a cloud's workgroups on one XCD (pc3d_common.h)
  const int b = by_, lane = threadIdx.x & 63;
  const int p = bx_ * 4 + (threadIdx.x >> 6);
  if (p >= NA) return;
  const int* o = off + (int64_t)b * (NA + 1);
  const int* l = lst + (int64_t)b * L;
  const int t0 = __builtin_amdgcn_readfirstlane(o[p]), t1 = __builtin_amdgcn_readfirstlane(o[p + 1]);
  const int SK = S * K;
  float acc[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) acc[q] = 0.f;
  constexpr int U = 4;
  for (int t = t0; t < t1; t += U) {
    float gv[U][Q];
    bool pos[U][Q];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      bool live = t + u < t1;
      const int code = __builtin_amdgcn_readfirstlane(l[live ? t + u : t]);
      const bool is_tail = code >= SK;
      if (amask && live && !is_tail) {                       // (uniform) a row the sparse producer did not write
        const int sg = code / K, jj = code - sg * K;
        live = (amask[((int64_t)b * S + sg) * ((K + 31) >> 5) + (jj >> 5)] >> (jj & 31)) & 1u;
      }
      const int64_t row = (int64_t)b * SK + (is_tail ? 0 : code);
      const float* src = is_tail ? tail + ((int64_t)b * S + (code - SK)) * C : gH + row * C;
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        const int c = 64 * q + lane;
        gv[u][q] = (live && c < C) ? src[c] : 0.f;
        if (is_tail) pos[u][q] = true;
        else if (H) pos[u][q] = c < C && H[row * C + c] > 0.f;
        else pos[u][q] = c < C && ((mask[row * (C >> 2) + (c >> 2)] >> (c & 3)) & 1) != 0;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int q = 0; q < Q; ++q) acc[q] += pos[u][q] ? gv[u][q] : gv[u][q] * slope;
  }
  float* dst = gP + ((int64_t)b * NA + p) * C;
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const int c = 64 * q + lane;
    if (c < C) dst[c] = acc[q];
  }
}

}  // namespace pc3d


using namespace pc3d;

static int fps_launch(const char* nm, int threads, const float* xyz, int64_t x_bs, int64_t x_ps, int64_t x_cs, int B, int N, int S,
                      const int32_t* start, int32_t* out, void* stream) {
  PC3D_REQUIRE(B >= 0 && N >= 1 && S >= 1, "%s: bad sizes B=%d N=%d S=%d", nm, B, N, S);
  PC3D_REQUIRE(threads == 64 || threads == 128 || threads == 256 || threads == 512 || threads == 1024,
               "%s: threads=%d (64, 128, 256, 512 or 1024)", nm, threads);
  PC3D_REQUIRE(N <= threads * FPS_MAXPER && N <= FPS_T * FPS_MAXPER, "%s: N=%d exceeds %d (threads * %d)", nm, N,
               threads * FPS_MAXPER < FPS_T * FPS_MAXPER ? threads * FPS_MAXPER : FPS_T * FPS_MAXPER, FPS_MAXPER);
  if (B == 0) return PC3D_OK;
  PC3D_REQUIRE(xyz && out, "%s: null pointer", nm);
#ifdef FPS_DIAG
  FpsArgs a{{xyz, x_bs, x_ps, x_cs}, N, S, start, out, g_fps_diag};
#else
  FpsArgs a{{xyz, x_bs, x_ps, x_cs}, N, S, start, out};
#endif
  const size_t lds = (size_t)3 * N * sizeof(float);
  hipStream_t st = as_stream(stream);
  const int per = cdiv(N, threads);
#define PC3D_FPS(TV)                                                                                 \
  do {                                                                                               \
    if (per <= 1) hipLaunchKernelGGL((fps_kernel<1, TV>), dim3(B), dim3(TV), lds, st, a);            \
    else if (per <= 2) hipLaunchKernelGGL((fps_kernel<2, TV>), dim3(B), dim3(TV), lds, st, a);       \
    else if (per <= 4) hipLaunchKernelGGL((fps_kernel<4, TV>), dim3(B), dim3(TV), lds, st, a);       \
    else if (per <= 8) hipLaunchKernelGGL((fps_kernel<8, TV>), dim3(B), dim3(TV), lds, st, a);       \
    else if (per <= 16) hipLaunchKernelGGL((fps_kernel<16, TV>), dim3(B), dim3(TV), lds, st, a);     \
    else hipLaunchKernelGGL((fps_kernel<32, TV>), dim3(B), dim3(TV), lds, st, a);                    \
  } while (0)
  if (threads == 64) PC3D_FPS(64);
  else if (threads == 128) PC3D_FPS(128);
  else if (threads == 256) PC3D_FPS(256);
  else if (threads == 512) PC3D_FPS(512);
  else PC3D_FPS(1024);
#undef PC3D_FPS
  PC3D_LAUNCH_CHECK(nm);
  return PC3D_OK;
}

#ifdef FPS_DIAG
extern "C" void fps_set_diag(long long* p) { g_fps_diag = p; }
#endif

extern "C" int pc3d_fps_f32(const float* xyz, int64_t x_bs, int64_t x_ps, int64_t x_cs, int B, int N, int S,
                            const int32_t* start, int32_t* out, void* stream) {
  // threads per cloud by N (tools/bench_fps.py, ns per step at 64 / 128 / 256 / 512 threads, round 4 with the one-instruction-
  // per-level DPP maximum): N = 512: 325 / 429 / 508 / 786; 1024: 468 / 493 / 556 / 816; 2048: 908 / 617 / 610 / 875; 4096:
  // - / 1068 / 750 / 1003 — a step is a latency chain (update, wave arg-max, exchange + barrier, winner's coordinates): fewer
  // wavefronts shorten the exchange until the per-lane work takes over. (Taking 64 threads at N = 1024 first made the
  // hipGraph-replayed CurveNet loop lose run == run in one run of twelve: the single high-priority wavefront per CU skews the
  // wavefronts of whatever shares the CU, and topk_desc_kernel's bitonic sort was missing a workgroup barrier — fixed in
  // curvenet_cl.hip, tools/exp/curvenet_graph_race.py; 0 of 600 replays differ since.) The pruned form (fps_pruned.hip,
  // pc3d_fps_pruned_f32) gives the same picks at 788 / 699 / 672 / 636 ns per step for N = 4096 / 2048 / 1024 / 512: not
  // faster, so not chosen here
  const int threads = N <= 512 ? 64 : FPS_T;
  return fps_launch("pc3d_fps_f32", threads, xyz, x_bs, x_ps, x_cs, B, N, S, start, out, stream);
}

// the same sampling with the workgroup size named (64 / 128 / 256 / 512 / 1024 threads per cloud): for tests and measurements
extern "C" int pc3d_fps_threads_f32(int threads, const float* xyz, int64_t x_bs, int64_t x_ps, int64_t x_cs, int B, int N, int S,
                                    const int32_t* start, int32_t* out, void* stream) {
  return fps_launch("pc3d_fps_threads_f32", threads, xyz, x_bs, x_ps, x_cs, B, N, S, start, out, stream);
}

// kernel: 0 = choose, 1 = a wavefront per centre, 2 = a centre per lane (error when its limits are exceeded)
static int ball_query_launch(const char* nm, int kernel, const float* xyz, int64_t x_bs, int64_t x_ps, int64_t x_cs,
                             const float* centers, int64_t c_bs, int64_t c_ps, int64_t c_cs, int B, int N, int S,
                             float radius, int nsample, int32_t* out, void* stream) {
  PC3D_REQUIRE(B >= 0 && N >= 1 && S >= 1 && nsample >= 1, "%s: bad sizes B=%d N=%d S=%d ns=%d", nm, B, N, S, nsample);
  PC3D_REQUIRE(B <= 65535, "%s: B=%d exceeds grid.y limit", nm, B);
  const bool tile_ok = N <= 65535 && nsample <= 96;   // 16-bit hit lists of four quarters per centre: <= 57 KB of LDS
  PC3D_REQUIRE(kernel != 2 || tile_ok, "%s: the centre-per-lane kernel needs N <= 65535 and nsample <= 96 (N=%d ns=%d)", nm,
               N, nsample);
  if (B == 0) return PC3D_OK;
  PC3D_REQUIRE(xyz && centers && out, "%s: null pointer", nm);
  BallArgs a{{xyz, x_bs, x_ps, x_cs}, {centers, c_bs, c_ps, c_cs}, N, S, nsample, radius * radius, out};
  // the centre-per-lane kernel runs S/64 x B workgroups: it wins once those fill the chip (measured at B=32/64:
  // 181 -> 93 us for N=4096, S=1024; 99 -> 56 us for N=2048, S=512; 20 -> 32 us for N=1024, S=256)
  const bool tile = kernel == 2 || (kernel == 0 && tile_ok && (int64_t)cdiv(S, 64) * B >= 256);
  if (tile)
    hipLaunchKernelGGL(ball_query_tile_kernel, dim3(cdiv(S, 64), B), dim3(BQ_SEG * 64), ball_tile_lds(nsample),
                       as_stream(stream), a);
  else
    hipLaunchKernelGGL(ball_query_kernel, dim3(cdiv(S, 4), B), dim3(256), 0, as_stream(stream), a);
  PC3D_LAUNCH_CHECK(nm);
  return PC3D_OK;
}

extern "C" int pc3d_ball_query_f32(const float* xyz, int64_t x_bs, int64_t x_ps, int64_t x_cs,
                                   const float* centers, int64_t c_bs, int64_t c_ps, int64_t c_cs,
                                   int B, int N, int S, float radius, int nsample, int32_t* out, void* stream) {
  return ball_query_launch("pc3d_ball_query_f32", 0, xyz, x_bs, x_ps, x_cs, centers, c_bs, c_ps, c_cs, B, N, S, radius,
                           nsample, out, stream);
}

extern "C" int pc3d_ball_query_kernel_f32(int kernel, const float* xyz, int64_t x_bs, int64_t x_ps, int64_t x_cs,
                                          const float* centers, int64_t c_bs, int64_t c_ps, int64_t c_cs, int B, int N,
                                          int S, float radius, int nsample, int32_t* out, void* stream) {
  PC3D_REQUIRE(kernel == 1 || kernel == 2, "pc3d_ball_query_kernel_f32: kernel=%d (1: wavefront per centre, 2: centre per lane)",
               kernel);
  return ball_query_launch("pc3d_ball_query_kernel_f32", kernel, xyz, x_bs, x_ps, x_cs, centers, c_bs, c_ps, c_cs, B, N, S,
                           radius, nsample, out, stream);
}

extern "C" int pc3d_group_gather_f32(const float* xyz, int64_t x_bs, int64_t x_ps, int64_t x_cs, const float* feat,
                                     int D, const int32_t* idx, const float* centers, int64_t c_bs, int64_t c_ps,
                                     int64_t c_cs, int B, int N, int S, int ns, float* out, void* stream) {
  PC3D_REQUIRE(B >= 0 && N >= 1 && S >= 1 && ns >= 1 && D >= 0, "pc3d_group_gather_f32: bad sizes");
  PC3D_REQUIRE(xyz != nullptr || (feat != nullptr && D > 0), "pc3d_group_gather_f32: nothing to gather");
  PC3D_REQUIRE(B <= 65535, "pc3d_group_gather_f32: B=%d exceeds grid.y limit", B);
  if (B == 0) return PC3D_OK;
  PC3D_REQUIRE(idx && out, "pc3d_group_gather_f32: null pointer");
  if (feat == nullptr) D = 0;
  GatherArgs a{{xyz, x_bs, x_ps, x_cs}, feat, idx, {centers, c_bs, c_ps, c_cs}, N, S, ns, D, (xyz ? 3 : 0) + D, out};
  hipLaunchKernelGGL(group_gather_kernel, dim3(cdiv(S * ns, 4), B), dim3(256), 0, as_stream(stream), a);
  PC3D_LAUNCH_CHECK("pc3d_group_gather_f32");
  return PC3D_OK;
}

// deterministic centre term: csum[b,s,c] = - sum_j g[b,s,j,c] over the rows whose index is valid (c < 3), j ascending
__global__ __launch_bounds__(256) void group_center_sum_kernel(const float* __restrict__ g, const int* __restrict__ idx, int N,
                                                               int ns, int C, float* __restrict__ csum, int64_t total) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;     // over B * S * 3
  if (t >= total) return;
  const int c = (int)(t % 3);
  const int64_t bs = t / 3;
  float acc = 0.f;
  for (int j = 0; j < ns; ++j)
    if ((unsigned)idx[bs * ns + j] < (unsigned)N) acc += g[(bs * ns + j) * C + c];
  csum[t] = -acc;
}

extern "C" int pc3d_group_gather_bwd_f32(const float* g_out, const int32_t* idx, const int32_t* center_idx, int B,
                                         int N, int S, int ns, int D, int has_xyz, float* grad_xyz, float* grad_feat,
                                         float* det_ws, void* stream) {
  PC3D_REQUIRE(B >= 0 && N >= 1 && S >= 1 && ns >= 1 && D >= 0, "pc3d_group_gather_bwd_f32: bad sizes");
  PC3D_REQUIRE(B <= 65535, "pc3d_group_gather_bwd_f32: B=%d exceeds grid.y limit", B);
  if (B == 0) return PC3D_OK;
  PC3D_REQUIRE(g_out && idx, "pc3d_group_gather_bwd_f32: null pointer");
  hipStream_t st = as_stream(stream);
  if (det_ws) {
    // deterministic (det_ws: B * S * 3 floats of scratch): every output is OVERWRITTEN by an ordered LDS scatter over
    // the S * ns rows of a cloud (det.hip); the centre term is summed per group first and scattered on top
    const char* nm = "pc3d_group_gather_bwd_f32";
    PC3D_REQUIRE((int64_t)S * ns <= 0x7fffffffLL, "%s: S * ns too large", nm);
    const int C = (has_xyz ? 3 : 0) + D;
    if (grad_xyz && has_xyz) {
      if (int rc = scatter_rows_det(nm, idx, g_out, C, nullptr, 0, 0.f, B, S * ns, N, 3, grad_xyz, 3, 0, 0, stream)) return rc;
      if (center_idx) {
        const int64_t total = (int64_t)B * S * 3;
        hipLaunchKernelGGL(group_center_sum_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, g_out, idx, N, ns, C,
                           det_ws, total);
        PC3D_LAUNCH_CHECK(nm);
        if (int rc = scatter_rows_det(nm, center_idx, det_ws, 3, nullptr, 0, 0.f, B, S, N, 3, grad_xyz, 3, 1, 0, stream)) return rc;
      }
    }
    if (grad_feat && D > 0)
      return scatter_rows_det(nm, idx, g_out + (has_xyz ? 3 : 0), C, nullptr, 0, 0.f, B, S * ns, N, D, grad_feat, D, 0, 0, stream);
    return PC3D_OK;
  }
  hipError_t e = hipSuccess;
  if (grad_xyz) e = zero_async(grad_xyz, (size_t)B * N * 3, st);
  if (e == hipSuccess && grad_feat) e = zero_async(grad_feat, (size_t)B * N * D, st);
  if (e != hipSuccess) {
    set_error("pc3d_group_gather_bwd_f32: zero fill failed: %s", hipGetErrorString(e));
    return (int)e;
  }
  GatherBwdArgs a{g_out, idx, center_idx, N, S, ns, D, (has_xyz ? 3 : 0) + D, has_xyz, grad_xyz, grad_feat};
  hipLaunchKernelGGL(group_gather_bwd_kernel, dim3(S, B), dim3(256), 0, st, a);
  PC3D_LAUNCH_CHECK("pc3d_group_gather_bwd_f32");
  return PC3D_OK;
}

static int group_max_linear_bwd_launch(const float* gout, const float* out, const int64_t* arg, const float* W, int G, int ns,
                                       int C2, int C3, const float* xin, const uint32_t* xmask, float* gx, void* stream,
                                       uint32_t* amask = nullptr, int ksplit = 1) {
  PC3D_REQUIRE(G >= 0 && ns >= 1 && ns <= GMB_MAXNS && C2 >= 1 && C3 >= 1 && C3 <= 4096,
               "pc3d_group_max_linear_bwd_f32: bad sizes G=%d ns=%d C2=%d C3=%d (ns <= 128, C3 <= 4096)", G, ns, C2, C3);
  PC3D_REQUIRE(!xmask || C2 % 32 == 0, "pc3d_group_max_linear_bwd_mask_f32: C2=%d must be a multiple of 32", C2);
  if (G == 0) return PC3D_OK;
  PC3D_REQUIRE(gout && out && arg && W && gx, "pc3d_group_max_linear_bwd_f32: null pointer");
  GroupMaxBwdArgs a{gout, out, arg, W, ns, C2, C3, gx, xin, xmask, amask};
  const int bt = C2 <= 64 ? 64 : GMB_T;          // one wave per group when the layer is narrow
  const dim3 grid(G, cdiv(C2, bt)), block(bt);
  hipStream_t st = as_stream(stream);
  const size_t lds = 2 * (size_t)C3 * sizeof(float);
  if (amask && xmask && (ns == 32 || ns == 64 || ns == 128) && C3 <= 1024 && C3 <= 65535) {
    // the sparse form: channels sorted by winning row (group_max_bwd_csr_kernel), one wavefront per (group, 64 columns)
    const dim3 gridc(G, cdiv(C2, 64));
    const size_t ldsc = (size_t)2 * C3 * sizeof(float2) + (size_t)(2 * ns + 1) * sizeof(int);
    if (ns == 32) hipLaunchKernelGGL(group_max_bwd_csr_kernel<32>, gridc, dim3(64), ldsc, st, a);
    else if (ns == 64) hipLaunchKernelGGL(group_max_bwd_csr_kernel<64>, gridc, dim3(64), ldsc, st, a);
    else hipLaunchKernelGGL(group_max_bwd_csr_kernel<128>, gridc, dim3(64), ldsc, st, a);
    PC3D_LAUNCH_CHECK("pc3d_group_max_linear_bwd_sparse_f32");
    return PC3D_OK;
  }
  if (ksplit > 1) {
    PC3D_REQUIRE(ksplit == 4 && C3 % 32 == 0, "pc3d_group_max_linear_bwd_ks_f32: ksplit=%d (1 or 4; C3 %% 32 == 0)", ksplit);
    const size_t lds4 = lds + (size_t)3 * 32 * bt * sizeof(float);
    const dim3 block4(bt, 4);
    if (ns <= 32) hipLaunchKernelGGL((group_max_linear_bwd_kernel<32, 4>), grid, block4, lds4, st, a);
    else if (ns <= 64) hipLaunchKernelGGL((group_max_linear_bwd_kernel<64, 4>), grid, block4, lds4, st, a);
    else hipLaunchKernelGGL((group_max_linear_bwd_kernel<128, 4>), grid, block4, lds4, st, a);
    PC3D_LAUNCH_CHECK("pc3d_group_max_linear_bwd_ks_f32");
    return PC3D_OK;
  }
  if (ns <= 32) hipLaunchKernelGGL(group_max_linear_bwd_kernel<32>, grid, block, lds, st, a);
  else if (ns <= 64) hipLaunchKernelGGL(group_max_linear_bwd_kernel<64>, grid, block, lds, st, a);
  else hipLaunchKernelGGL(group_max_linear_bwd_kernel<128>, grid, block, lds, st, a);
  PC3D_LAUNCH_CHECK("pc3d_group_max_linear_bwd_f32");
  return PC3D_OK;
}

extern "C" int pc3d_group_max_linear_bwd_f32(const float* gout, const float* out, const int64_t* arg, const float* W,
                                             int G, int ns, int C2, int C3, const float* xin, float* gx, void* stream) {
  return group_max_linear_bwd_launch(gout, out, arg, W, G, ns, C2, C3, xin, nullptr, gx, stream);
}

extern "C" int pc3d_group_max_linear_bwd_ks_f32(const float* gout, const float* out, const int64_t* arg, const float* W,
                                                int G, int ns, int C2, int C3, const float* xin, float* gx, int ksplit,
                                                void* stream) {
  return group_max_linear_bwd_launch(gout, out, arg, W, G, ns, C2, C3, xin, nullptr, gx, stream, nullptr, ksplit);
}

extern "C" int pc3d_group_max_linear_bwd_mask_f32(const float* gout, const float* out, const int64_t* arg, const float* W,
                                                  int G, int ns, int C2, int C3, const uint32_t* xmask, float* gx,
                                                  void* stream) {
  PC3D_REQUIRE(xmask != nullptr, "pc3d_group_max_linear_bwd_mask_f32: null mask");
  return group_max_linear_bwd_launch(gout, out, arg, W, G, ns, C2, C3, nullptr, xmask, gx, stream);
}

extern "C" int pc3d_group_max_linear_bwd_sparse_f32(const float* gout, const float* out, const int64_t* arg, const float* W,
                                                    int G, int ns, int C2, int C3, const uint32_t* xmask, float* gx,
                                                    uint32_t* amask, void* stream) {
  PC3D_REQUIRE(xmask != nullptr && amask != nullptr, "pc3d_group_max_linear_bwd_sparse_f32: null mask");
  return group_max_linear_bwd_launch(gout, out, arg, W, G, ns, C2, C3, nullptr, xmask, gx, stream, amask);
}

extern "C" int pc3d_group_act_f32(const float* P, const float* Bc, const int32_t* idx, int B, int NA, int S, int K, int C,
                                  float slope, float* H, void* stream) {
  PC3D_REQUIRE(B >= 0 && NA >= 1 && S >= 1 && K >= 1 && C >= 4 && C % 4 == 0, "pc3d_group_act_f32: bad sizes B=%d NA=%d S=%d K=%d C=%d (C %% 4 == 0)", B, NA, S, K, C);
  if (B == 0) return PC3D_OK;
  PC3D_REQUIRE(P && Bc && idx && H, "pc3d_group_act_f32: null pointer");
  const int64_t total = (int64_t)B * S * K * (C / 4);
  const int64_t nb = (total + 255) / 256;
  PC3D_REQUIRE(nb <= 0x7fffffffLL, "pc3d_group_act_f32: problem too large for one launch");
  hipLaunchKernelGGL(group_act_fwd_kernel, dim3((unsigned)nb), dim3(256), 0, as_stream(stream), (const float4*)P,
                     (const float4*)Bc, idx, NA, S, K, C / 4, slope, (float4*)H, total);
  PC3D_LAUNCH_CHECK("pc3d_group_act_f32");
  return PC3D_OK;
}

extern "C" int pc3d_group_act_bwd_f32(const float* gH, const float* H, const int32_t* idx, int B, int NA, int S, int K,
                                      int C, float slope, float* gP, float* gBc, int deterministic, void* stream) {
  PC3D_REQUIRE(B >= 0 && NA >= 1 && S >= 1 && K >= 1 && C >= 1 && C <= GAB_MAXC, "pc3d_group_act_bwd_f32: bad sizes B=%d NA=%d S=%d K=%d C=%d (C <= %d)", B, NA, S, K, C, GAB_MAXC);
  PC3D_REQUIRE(B <= 65535, "pc3d_group_act_bwd_f32: B=%d exceeds grid.y limit", B);
  if (B == 0) return PC3D_OK;
  PC3D_REQUIRE(gH && H && idx && gP && gBc, "pc3d_group_act_bwd_f32: null pointer");
  hipStream_t st = as_stream(stream);
  if (deterministic) {
    hipLaunchKernelGGL(group_act_bwd_kernel<false>, dim3(S, B), dim3(256), (size_t)8 * C * sizeof(float), st, gH, H, idx, NA, S,
                       K, C, slope, gP, gBc, (const uint8_t*)nullptr);
    PC3D_LAUNCH_CHECK("pc3d_group_act_bwd_f32");
    return scatter_rows_det("pc3d_group_act_bwd_f32", idx, gH, C, H, C, slope, B, S * K, NA, C, gP, C, 0, 0, stream);
  }
  if (hipError_t e = zero_async(gP, (size_t)B * NA * C, st); e != hipSuccess) {
    set_error("pc3d_group_act_bwd_f32: zero fill failed: %s", hipGetErrorString(e));
    return (int)e;
  }
  hipLaunchKernelGGL(group_act_bwd_kernel<true>, dim3(S, B), dim3(256), (size_t)8 * C * sizeof(float), st, gH, H, idx, NA, S,
                     K, C, slope, gP, gBc, (const uint8_t*)nullptr);
  PC3D_LAUNCH_CHECK("pc3d_group_act_bwd_f32");
  return PC3D_OK;
}

extern "C" int pc3d_group_act_bwd_mask_f32(const float* gH, const uint8_t* mask, const int32_t* idx, int B, int NA, int S,
                                           int K, int C, float slope, float* gP, float* gBc, int deterministic, void* stream) {
  PC3D_REQUIRE(B >= 0 && NA >= 1 && S >= 1 && K >= 1 && C >= 4 && C % 4 == 0 && C <= GAB_MAXC, "pc3d_group_act_bwd_mask_f32: bad sizes B=%d NA=%d S=%d K=%d C=%d (C %% 4 == 0, C <= %d)", B, NA, S, K, C, GAB_MAXC);
  PC3D_REQUIRE(B <= 65535, "pc3d_group_act_bwd_mask_f32: B=%d exceeds grid.y limit", B);
  if (B == 0) return PC3D_OK;
  PC3D_REQUIRE(gH && mask && idx && gP && gBc, "pc3d_group_act_bwd_mask_f32: null pointer");
  hipStream_t st = as_stream(stream);
  if (deterministic) {
    hipLaunchKernelGGL(group_act_bwd_kernel<false>, dim3(S, B), dim3(256), (size_t)8 * C * sizeof(float), st, gH,
                       (const float*)nullptr, idx, NA, S, K, C, slope, gP, gBc, mask);
    PC3D_LAUNCH_CHECK("pc3d_group_act_bwd_mask_f32");
    return scatter_rows_det("pc3d_group_act_bwd_mask_f32", idx, gH, C, nullptr, 0, slope, B, S * K, NA, C, gP, C, 0, 0, stream, mask);
  }
  if (hipError_t e = zero_async(gP, (size_t)B * NA * C, st); e != hipSuccess) {
    set_error("pc3d_group_act_bwd_mask_f32: zero fill failed: %s", hipGetErrorString(e));
    return (int)e;
  }
  hipLaunchKernelGGL(group_act_bwd_kernel<true>, dim3(S, B), dim3(256), (size_t)8 * C * sizeof(float), st, gH,
                     (const float*)nullptr, idx, NA, S, K, C, slope, gP, gBc, mask);
  PC3D_LAUNCH_CHECK("pc3d_group_act_bwd_mask_f32");
  return PC3D_OK;
}

extern "C" int64_t pc3d_group_reverse_list_len(int S, int K) { return (int64_t)S * K + S; }

extern "C" int pc3d_group_reverse_i32(const int32_t* idx, int B, int NA, int S, int K, int32_t* cnt, int32_t* off,
                                      int32_t* lst, void* stream) {
  PC3D_REQUIRE(B >= 0 && NA >= 1 && S >= 1 && K >= 1 && (int64_t)S * K + S <= 0x7fffffffLL,
               "pc3d_group_reverse_i32: bad sizes B=%d NA=%d S=%d K=%d", B, NA, S, K);
  if (B == 0) return PC3D_OK;
  PC3D_REQUIRE(idx && cnt && off && lst, "pc3d_group_reverse_i32: null pointer");
  hipStream_t st = as_stream(stream);
  const int64_t total = (int64_t)B * S * K, L = (int64_t)S * K + S;
  const int64_t nb = (total + 255) / 256;
  PC3D_REQUIRE(nb <= 0x7fffffffLL, "pc3d_group_reverse_i32: problem too large for one launch");
  if (hipError_t e = zero_async(reinterpret_cast<float*>(cnt), (size_t)B * NA, st); e != hipSuccess) {   // all-zero bits
    set_error("pc3d_group_reverse_i32: zero fill failed: %s", hipGetErrorString(e));
    return (int)e;
  }
  hipLaunchKernelGGL(group_rev_count_kernel, dim3((unsigned)nb), dim3(256), 0, st, idx, NA, S, K, cnt, total);
  hipLaunchKernelGGL(group_rev_scan_kernel, dim3(B), dim3(256), 0, st, cnt, NA, off);
  hipLaunchKernelGGL(group_rev_fill_kernel, dim3((unsigned)nb), dim3(256), 0, st, idx, NA, S, K, cnt, off, lst, L, total);
  PC3D_LAUNCH_CHECK("pc3d_group_reverse_i32");
  // the fill hands out slots through integer atomics: a point's entries arrive in a different order every run. Sorted,
  // the points pass of the backward sums them in ONE order (deterministic gradients; det.hip)
  return sort_segments("pc3d_group_reverse_i32", off, lst, B, NA, L, stream);
}

extern "C" int pc3d_group_act_bwd_rev_f32(const float* gH, const float* H, const uint8_t* mask, const int32_t* idx,
                                          const int32_t* off, const int32_t* lst, int B, int NA, int S, int K, int C,
                                          float slope, float* gP, float* gBc, float* tail, void* stream) {
  PC3D_REQUIRE(B >= 0 && NA >= 1 && S >= 1 && K >= 1 && C >= 4 && C % 4 == 0 && C <= GAB_MAXC,
               "pc3d_group_act_bwd_rev_f32: bad sizes B=%d NA=%d S=%d K=%d C=%d (C %% 4 == 0, C <= %d)", B, NA, S, K, C, GAB_MAXC);
  PC3D_REQUIRE(B <= 65535, "pc3d_group_act_bwd_rev_f32: B=%d exceeds grid.y limit", B);
  if (B == 0) return PC3D_OK;
  PC3D_REQUIRE(gH && (H || mask) && idx && off && lst && gP && gBc && tail, "pc3d_group_act_bwd_rev_f32: null pointer");
  hipStream_t st = as_stream(stream);
  const size_t lds = 0;
  const int64_t L = (int64_t)S * K + S;
  const dim3 gg(cdiv(S, 4), B), gp(cdiv(NA, 4), B), blk(256);
#define PC3D_GAB_LAUNCH(Q)                                                                                                \
  hipLaunchKernelGGL(group_act_bwd_groups_kernel<Q>, gg, blk, lds, st, gH, H, mask, idx, S, K, C, slope, gBc, tail);      \
  hipLaunchKernelGGL(group_act_bwd_points_kernel<Q>, gp, blk, 0, st, gH, H, mask, tail, off, lst, L, NA, S, K, C, slope, gP)
  if (C <= 64) { PC3D_GAB_LAUNCH(1); }
  else if (C <= 128) { PC3D_GAB_LAUNCH(2); }
  else if (C <= 256) { PC3D_GAB_LAUNCH(4); }
  else { PC3D_GAB_LAUNCH(8); }
#undef PC3D_GAB_LAUNCH
  PC3D_LAUNCH_CHECK("pc3d_group_act_bwd_rev_f32");
  return PC3D_OK;
}

// the points pass alone, for a caller that already has the groups pass's outputs (pc3d_sa_chain_bwd_f32 writes them)
extern "C" int pc3d_group_act_bwd_points_f32(const float* gH, const uint8_t* mask, const float* tail, const int32_t* off,
                                             const int32_t* lst, const uint32_t* amask, int B, int NA, int S, int K, int C,
                                             float slope, float* gP, void* stream) {
  const char* nm = "pc3d_group_act_bwd_points_f32";
  PC3D_REQUIRE(B >= 0 && NA >= 1 && S >= 1 && K >= 1 && C >= 4 && C % 4 == 0 && C <= GAB_MAXC,
               "%s: bad sizes B=%d NA=%d S=%d K=%d C=%d (C %% 4 == 0, C <= %d)", nm, B, NA, S, K, C, GAB_MAXC);
  PC3D_REQUIRE(B <= 65535, "%s: B=%d exceeds grid.y limit", nm, B);
  if (B == 0) return PC3D_OK;
  PC3D_REQUIRE(gH && mask && tail && off && lst && gP, "%s: null pointer", nm);
  hipStream_t st = as_stream(stream);
  const int64_t L = (int64_t)S * K + S;
  const dim3 gp(cdiv(NA, 4), B), blk(256);
  const float* H = nullptr;
#define PC3D_GAP_LAUNCH(Q) \
  hipLaunchKernelGGL(group_act_bwd_points_kernel<Q>, gp, blk, 0, st, gH, H, mask, tail, off, lst, L, NA, S, K, C, slope, gP, amask)
  if (C <= 64) { PC3D_GAP_LAUNCH(1); }
  else if (C <= 128) { PC3D_GAP_LAUNCH(2); }
  else if (C <= 256) { PC3D_GAP_LAUNCH(4); }
  else { PC3D_GAP_LAUNCH(8); }
#undef PC3D_GAP_LAUNCH
  PC3D_LAUNCH_CHECK(nm);
  return PC3D_OK;
}

// kernel: 0 = choose, 1 = a workgroup per group, 2 = the tiled GEMM main loop with the group-max epilogue (ns = 32 / 64 / 128)
static int group_linear_max_launch(int kernel, const float* x, const float* W, const float* b, int G, int ns, int C2, int C3,
                                   float* out, int64_t* arg, void* stream) {
  // groups of 32 / 64 / 128 rows: the tiled GEMM main loop (operands through LDS, 128 rows share a weight tile) with a
  // group-max epilogue — measured against the one-workgroup-per-group kernel below in tools/bench_glm.py. It takes any
  // input width; the per-group kernel keeps a group's rows in registers (C2 <= 128).
  const bool gemm_ok = (ns == 32 || ns == 64 || ns == 128) && (int64_t)G * ns <= 0x7fffffff;
  PC3D_REQUIRE(G >= 0 && ns >= 1 && ns <= 128 && C2 >= 8 && C2 % 8 == 0 && (C2 <= 128 || (gemm_ok && kernel != 1)) && C3 >= 32 &&
                   C3 % 32 == 0 && C3 <= 4096,
               "pc3d_group_linear_max_f32: unsupported sizes ns=%d C2=%d C3=%d (ns <= 128, C2 %% 8 == 0, C2 <= 128 unless ns in "
               "{32, 64, 128}, C3 %% 32 == 0)", ns, C2, C3);
  if (G == 0) return PC3D_OK;
  PC3D_REQUIRE(x && W && b && out && arg, "pc3d_group_linear_max_f32: null pointer");
  PC3D_REQUIRE(kernel != 2 || gemm_ok, "pc3d_group_linear_max_kernel_f32: the GEMM form needs ns in {32, 64, 128} (ns=%d)", ns);
  if (kernel != 1 && gemm_ok) return gemm_nt_groupmax(x, W, b, G, ns, C2, C3, out, arg, stream);
  GroupLinMaxArgs a{x, W, b, ns, C2, C3, out, arg};
  const int nw = cdiv(ns, 32);
  const size_t lds = nw > 1 ? (size_t)2 * nw * C3 * sizeof(float) : 0;
  hipStream_t st = as_stream(stream);
  if (C2 <= 64) hipLaunchKernelGGL(group_linear_max_kernel<8>, dim3(G), dim3(64 * nw), lds, st, a);
  else hipLaunchKernelGGL(group_linear_max_kernel<16>, dim3(G), dim3(64 * nw), lds, st, a);
  PC3D_LAUNCH_CHECK("pc3d_group_linear_max_f32");
  return PC3D_OK;
}

extern "C" int pc3d_group_linear_max_f32(const float* x, const float* W, const float* b, int G, int ns, int C2, int C3,
                                         float* out, int64_t* arg, void* stream) {
  return group_linear_max_launch(0, x, W, b, G, ns, C2, C3, out, arg, stream);
}

extern "C" int pc3d_group_linear_max_kernel_f32(int kernel, const float* x, const float* W, const float* b, int G, int ns,
                                                int C2, int C3, float* out, int64_t* arg, void* stream) {
  PC3D_REQUIRE(kernel >= 0 && kernel <= 2, "pc3d_group_linear_max_kernel_f32: kernel=%d (0 choose, 1 per group, 2 GEMM)", kernel);
  return group_linear_max_launch(kernel, x, W, b, G, ns, C2, C3, out, arg, stream);
}
