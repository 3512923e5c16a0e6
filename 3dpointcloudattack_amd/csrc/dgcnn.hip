// K3 — DGCNN dynamic-graph ops (model/dgcnn.py:194-227 knn + get_graph_feature, :299-313 EdgeConv), gfx950.
//
// (1) knn_feat: k nearest neighbours in C-dimensional FEATURE space (C in {64,64,128} after the first layer), self
//     included, for channels-last features x [B,N,C]. The reference materialises -|xi-xj|^2 as a [B,N,N] matrix
//     (GEMM + 2 broadcasts) and calls topk. Here a workgroup owns 32 queries and walks the references in blocks of
//     128: each wave forms one 32 x 32 tile with v_mfma_f32_32x32x2_f32 (this IS a dense contraction over C) into a
//     double-buffered LDS block (never HBM), then scans the block against the K-lists of its 8 queries, which live
//     across the lanes (knn_list.h: ballot/popcount + one DPP shift per insertion). Any N; K <= 64.
// (2) gather_max: out[b,i,c] = max_{j in nbr(i)} P[b,idx[b,i,j],c] (and the arg-max for the backward), the
//     neighbour reduction of an EdgeConv rewritten as two point-wise GEMMs:
//         W [x_j - x_i ; x_i] = Wa x_j + (Wb - Wa) x_i = P_j + Q_i ,   W = [Wa | Wb]
//     so  max_j leaky(bn(P_j + Q_i)) = leaky(bn(max_j P_j + Q_i))  for a positive BN scale (min_j for a negative one):
//     the [B,2C,N,k] edge tensor (up to 671 MB at B=32) and the k-fold conv FLOPs disappear.
#include "pc3d_common.h"
#include "knn_list.h"

namespace pc3d {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int KF_T = 256;      // 4 waves
constexpr int KF_Q = 32;       // queries per workgroup (the MFMA row block)
constexpr int KF_BLK = 128;    // reference points per block: one 32-column MFMA tile per wave
constexpr int KF_LD = KF_BLK + 4;
constexpr int KF_QW = KF_Q / 4;   // queries whose K-lists a wave maintains

struct KnnFeatArgs {
  const float* x;  // [B,N,C] channels-last
  int N, C, K;
  int32_t* idx;    // [B,N,K]
  int B;
};

// (Round 2, measured: a barrier-free variant — every wave owning 16 queries on v_mfma_f32_16x16x4_f32, distances
// transposed through a wave-private LDS strip — took 228 us against this kernel's 162 at B=32, N=1024, C=64, K=20,
// and 136 against 97 at K=1: the cost that does not scale with C or K (scan, seed sort, LDS round trips) dominates
// both forms, not the barriers. Cheaper list insertions (7 VALU, knn_list.h) moved this kernel from 171 to 162 us.)
// Register diet on purpose (~80 VGPRs -> 4+ waves per SIMD): the insertions are chains of dependent VALU/SALU
// hops, and only other resident waves hide them. So the A operand (the 32 query rows) sits in LDS, the B operand is
// streamed from global memory a few float4 ahead, and nothing but the accumulator tile and the lists stays live.
// NT = C / 8 when it is known at compile time (8: C = 64, 16: C = 128), 0 = any C. With a run-time trip count the compiler drains the
// load queue (s_waitcnt vmcnt(0)) inside the product loop, so the "4 float4 ahead" are in flight for one MFMA group at
// most. With NT static the loop is straight-line code: all 8 float4 of a row slice in flight with graded waits, and the
// 8 of the NEXT block are requested before this block's strip write and scan (133 -> 129 us at B=32, N=1024, C=64, K=20;
// products + strip alone 61 -> 57 us against 31 us of matrix-pipe work: the rest is the strip write's VALU work and the
// two barriers per block, which four lock-stepped workgroups per CU do not hide). C = 128 (NT = 16) is static too, with 4
// float4 in flight and WITHOUT the cross-block request (the operand would stay live through the scan: 136 or, 8 ahead,
// 152 VGPRs = 3 waves per SIMD, 195 us): 154 -> 150 us.
constexpr int KF_PF = 4;              // float4 of the B operand in flight per lane (run-time C)
constexpr int KF_PFS = 8;             // ... with a static C

template <int NT, bool KGE2>
__global__ __launch_bounds__(KF_T) void knn_feat_kernel(KnnFeatArgs a) {
  extern __shared__ __attribute__((aligned(16))) float kf_lds[];
  const int ldq = a.C + 4;                                  // row stride of the query block (16-byte aligned rows)
  int (*strip)[KF_LD] = reinterpret_cast<int (*)[KF_LD]>(kf_lds);              // [32][132] block of distance KEYS (knn_key)
  float* qs = kf_lds + KF_Q * KF_LD;                                             // [32][C+4] query rows
  int bx, b;                                                // all query blocks of a cloud on one XCD: they stream the same rows
  if (!xcd_block((a.N + KF_Q - 1) / KF_Q, a.B, bx, b)) return;
  const int q0 = bx * KF_Q;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 31, h = lane >> 5;
  const float* xb = a.x + (int64_t)b * a.N * a.C;
  const int nt = NT ? NT : a.C / 8;
  const int K = a.K;
  // stage the query rows (clamped at the cloud's end)
  {
    const int cq = a.C / 4;   // float4 per row
    for (int i = threadIdx.x; i < KF_Q * cq; i += KF_T) {
      const int row = i / cq, c4 = i - row * cq;
      const int qrow = (q0 + row < a.N) ? q0 + row : a.N - 1;
      *reinterpret_cast<float4*>(qs + row * ldq + 4 * c4) = *reinterpret_cast<const float4*>(xb + (int64_t)qrow * a.C + 4 * c4);
    }
  }
  // K-lists of this wave's queries (8*wave .. +7), across the lanes
  int lk[KF_QW], thr[KF_QW];
  int li[KF_QW];
#pragma unroll
  for (int u = 0; u < KF_QW; ++u) lk[u] = kKnnInfKey, li[u] = 0x7fffffff, thr[u] = kKnnInfKey;

  const int nblk = (a.N + KF_BLK - 1) / KF_BLK;
  constexpr int PF = NT == 8 ? KF_PFS : KF_PF;
  float4 bv[PF];            // B operand in flight, carried across blocks
  const float* rp_next;
  auto first_rows = [&](int blk) {
    const int rrow0 = blk * KF_BLK + 32 * wave + r;
    rp_next = xb + (int64_t)(rrow0 < a.N ? rrow0 : a.N - 1) * a.C + 4 * h;
#pragma unroll
    for (int t = 0; t < PF; ++t) bv[t] = (t < nt) ? *reinterpret_cast<const float4*>(rp_next + 8 * t) : make_float4(0.f, 0.f, 0.f, 0.f);
  };
  first_rows(0);
  __syncthreads();   // query rows visible
  for (int blk = 0; blk < nblk; ++blk) {
    int (*st)[KF_LD] = strip;
    // ---- phase 1: this wave's 32 x 32 tile of the block on MFMA (this IS a dense contraction over C)
    {
      f32x16 acc;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.f;
      float rn = 0.f;
      if (NT == 16 && blk > 0) first_rows(blk);
      const float* rp = rp_next;               // this block's row (its first PF float4 are already in flight)
      auto step = [&](const float4 v, int t) {
        const float4 q = *reinterpret_cast<const float4*>(qs + r * ldq + 8 * t + 4 * h);
        rn += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;   // squares summed, then added: the reference's xx = sum(x ** 2)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(q.x, v.x, acc, 0, 0, 0);   // D[row = query][col = reference]
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(q.y, v.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(q.z, v.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(q.w, v.w, acc, 0, 0, 0);
      };
      if constexpr (NT != 0) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const float4 v = bv[t % PF];
          if (t + PF < NT) bv[t % PF] = *reinterpret_cast<const float4*>(rp + 8 * (t + PF));
          step(v, t);
        }
      } else {
        for (int t0 = 0; t0 < nt; t0 += PF) {
#pragma unroll
          for (int tt = 0; tt < PF; ++tt) {
            const int t = t0 + tt;
            const float4 v = bv[tt];
            if (t + PF < nt) bv[tt] = *reinterpret_cast<const float4*>(rp + 8 * (t + PF));
            if (t < nt) step(v, t);
          }
        }
      }
      if (NT != 16 && blk + 1 < nblk) first_rows(blk + 1);   // (C = 128: the operand would stay live through the scan: 136 VGPRs)
      // |r_j|^2 for column j = r (the two lane halves hold the two k halves). A/B in round 4 (tools/bench_knn_feat.py, B=32,
      // N=1024, K=20 / K=1): this ds_swizzle-backed shuffle 111.4 / 65.1 us, sum_xor32 (v_permlane32_swap, round 3) 137.4 / 74.1 —
      // the swap's result is needed by the very next VALU instruction of every lane, and its wait states stall the wave where the
      // LDS round trip of the shuffle is overlapped with the next block's loads
      rn += __shfl_xor(rn, 32, 64);
      // model/dgcnn.py:195-197 ranks the references of a query by 2 q.r - |q|^2 - |r|^2, largest first. |q|^2 is the
      // same for every candidate of a query, so the key is that of |r|^2 - 2 q.r, smallest first (never -0: x - x is +0;
      // NaN -> +inf by v_min_f32, which returns its other operand for a quiet NaN): 6 VALU per element instead of 10 and
      // no |q|^2 read — this write is ~2/3 of the phase's VALU work.
      auto key_of = [&](float accv) {
        float d = rn - (accv + accv);
        asm("v_min_f32_e32 %0, 0x7f800000, %0" : "+v"(d));
        return knn_ord(d);
      };
      if (__builtin_expect((blk + 1) * KF_BLK > a.N, 0)) {          // ragged last block: columns beyond the cloud never rank
        const bool valid = (blk * KF_BLK + 32 * wave + r) < a.N;
#pragma unroll
        for (int e = 0; e < 16; ++e) st[(e & 3) + 8 * (e >> 2) + 4 * h][32 * wave + r] = valid ? key_of(acc[e]) : kKnnInfKey;
      } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) st[(e & 3) + 8 * (e >> 2) + 4 * h][32 * wave + r] = key_of(acc[e]);
      }
    }
    __syncthreads();
    // ---- phase 2: two steps of 64 candidates against this wave's 8 lists
#pragma unroll
    for (int s = 0; s < KF_BLK / 64; ++s) {
      const int jbase = blk * KF_BLK + 64 * s;
      int d[KF_QW];
#pragma unroll
      for (int u = 0; u < KF_QW; ++u) d[u] = st[KF_QW * wave + u][64 * s + lane];
      if (blk == 0 && s == 0) {
#pragma unroll
        for (int u = 0; u < KF_QW; ++u) {      // seed: sort the first 64 candidates
          int sk = d[u], si = lane;
          wave_sort_pairs_dpp(sk, si, lane);
          lk[u] = sk, li[u] = si;
          thr[u] = __builtin_amdgcn_readlane(sk, K - 1);
        }
        continue;
      }
#pragma unroll
      for (int u = 0; u < KF_QW; ++u) knn_scan_insert<KGE2>(lk[u], li[u], thr[u], d[u], jbase, K);
    }
    __syncthreads();   // the block is consumed before the next one overwrites it (single buffer: LDS buys residency)
  }
#pragma unroll
  for (int u = 0; u < KF_QW; ++u) {
    const int qi = q0 + KF_QW * wave + u;
    if (qi < a.N && lane < K) a.idx[((int64_t)b * a.N + qi) * K + lane] = li[u];
  }
}

// ---------------------------------------------------------------------------------------------------------
// gather-max / gather-min over neighbour lists, channels-last. sign[c] >= 0 -> max, < 0 -> min (BN scale sign).
// ---------------------------------------------------------------------------------------------------------
struct GMaxArgs {
  const float* P;       // [B,N,C]
  const int32_t* idx;   // [B,S,K] (entries clamped to [0,N-1])
  const float* sign;    // [C] or null (all max)
  int N, C, K;
  float* out;           // [B,S,C]
  int32_t* arg;         // [B,S,C] winning neighbour (absolute point index) or null
  int S;                // output rows per cloud (S == N for a neighbour graph on the points themselves)
  int B;
};

__global__ __launch_bounds__(256) void gather_max_kernel(GMaxArgs a) {
  int bx, b;
  if (!xcd_block((a.S + 3) / 4, a.B, bx, b)) return;
  const int i = bx * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (i >= a.S) return;
  const int32_t* nb = a.idx + ((int64_t)b * a.S + i) * a.K;
  const float* Pb = a.P + (int64_t)b * a.N * a.C;
  for (int c = lane; c < a.C; c += 64) {
    const bool mx = (a.sign == nullptr) || (a.sign[c] >= 0.f);
    float best = mx ? -__builtin_inff() : __builtin_inff();
    int bj = min(max(nb[0], 0), a.N - 1);
    for (int k = 0; k < a.K; ++k) {
      const int j = min(max(nb[k], 0), a.N - 1);
      const float v = Pb[(int64_t)j * a.C + c];
      if (mx ? (v > best) : (v < best)) best = v, bj = j;
    }
    a.out[((int64_t)b * a.S + i) * a.C + c] = best;
    if (a.arg) a.arg[((int64_t)b * a.S + i) * a.C + c] = bj;
  }
}

// C % 4 == 0: a lane owns 4 consecutive channels (one 16-byte load per neighbour), C/4 lanes per point and
// 256 / (C/4) points per workgroup; the K neighbour rows are independent loads, so they pipeline.
__global__ __launch_bounds__(256) void gather_max4_kernel(GMaxArgs a, int lpp) {   // lpp = C / 4 lanes per point
  const int ppw = 256 / lpp;                       // points per workgroup
  int bx, b;                                       // a cloud's workgroups on one XCD (see edge_max_kernel)
  if (!xcd_block((a.S + ppw - 1) / ppw, a.B, bx, b)) return;
  const int i = bx * ppw + threadIdx.x / lpp;
  const int l = threadIdx.x % lpp;
  if (threadIdx.x >= ppw * lpp || i >= a.S) return;
  const int32_t* nb = a.idx + ((int64_t)b * a.S + i) * a.K;
  const float* Pb = a.P + (int64_t)b * a.N * a.C + 4 * l;
  bool mx[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) mx[e] = (a.sign == nullptr) || (a.sign[4 * l + e] >= 0.f);
  float best[4];
  int bj[4];
  const int j0 = min(max(nb[0], 0), a.N - 1);
#pragma unroll
  for (int e = 0; e < 4; ++e) best[e] = mx[e] ? -__builtin_inff() : __builtin_inff(), bj[e] = j0;
  constexpr int U = 4;      // neighbours in flight (indices, then rows), see edge_max_kernel
  for (int k0 = 0; k0 < a.K; k0 += U) {
    int jj[U];
#pragma unroll
    for (int u = 0; u < U; ++u) jj[u] = min(max(nb[k0 + u < a.K ? k0 + u : k0], 0), a.N - 1);
    float4 vv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) vv[u] = *reinterpret_cast<const float4*>(Pb + (int64_t)jj[u] * a.C);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (k0 + u >= a.K) break;
      const float v[4] = {vv[u].x, vv[u].y, vv[u].z, vv[u].w};
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (mx[e] ? (v[e] > best[e]) : (v[e] < best[e])) best[e] = v[e], bj[e] = jj[u];
    }
  }
  const int64_t o = ((int64_t)b * a.S + i) * a.C + 4 * l;
  *reinterpret_cast<float4*>(a.out + o) = make_float4(best[0], best[1], best[2], best[3]);
  if (a.arg) *reinterpret_cast<int4*>(a.arg + o) = make_int4(bj[0], bj[1], bj[2], bj[3]);
}

// backward: gP[b, arg[b,i,c], c] += g[b,i,c]   (g, arg [B,S,C]; gP [B,N,C] zero-filled first)
__global__ __launch_bounds__(256) void gather_max_bwd_kernel(const float* g, const int32_t* arg, int N, int S, int C,
                                                             float* gP) {
  const int b = blockIdx.y;
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (int64_t)S * C) return;
  const int c = (int)(e % C);
  const int j = arg[(int64_t)b * S * C + e];
  atomicAdd(gP + ((int64_t)b * N + j) * C + c, g[(int64_t)b * S * C + e]);
}

// ---------------------------------------------------------------------------------------------------------
// EdgeConv epilogue in one launch: PQ[b,i,:] = [P_i | Q_i] (one GEMM against [U;V] with bias [0;t]),
//   out[b,i,c] = leaky(max_{j in nbr(i)} P[b,j,c] + Q[b,i,c])        (model/dgcnn.py:299-313 per layer)
// and its backward: g_pre = g * (out > 0 ? 1 : slope); dQ[b,i,c] = g_pre; dP[b,arg,c] += g_pre.
// Same lane layout as gather_max4_kernel (a lane owns 4 consecutive channels); C % 4 == 0.
// ---------------------------------------------------------------------------------------------------------
struct EdgeMaxArgs {
  const float* PQ;      // [B,N,2C]
  const int32_t* idx;   // [B,N,K]
  int N, C, K;
  float slope;
  float* out;           // [B,N,C]
  int32_t* arg;         // [B,N,C]
  int B;
  float* out2;          // null, or a second copy of out with row stride ld2 (a column slice of a wider [B,N,ld2] buffer)
  int64_t ld2;
};

__global__ __launch_bounds__(256) void edge_max_kernel(EdgeMaxArgs a, int lpp) {
  const int ppw = 256 / lpp;
  int bx, b;                                      // a cloud's workgroups on one XCD: its P rows (0.25-1 MB) stay in that L2
  if (!xcd_block((a.N + ppw - 1) / ppw, a.B, bx, b)) return;
  const int i = bx * ppw + threadIdx.x / lpp;
  const int l = threadIdx.x % lpp;
  if (threadIdx.x >= ppw * lpp || i >= a.N) return;
  const int32_t* nb = a.idx + ((int64_t)b * a.N + i) * a.K;
  const float* Pb = a.PQ + (int64_t)b * a.N * 2 * a.C + 4 * l;
  const float4 q = *reinterpret_cast<const float4*>(Pb + (int64_t)i * 2 * a.C + a.C);
  float best[4] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
  const int j0 = nb[0];
  int bj[4] = {j0, j0, j0, j0};
  // neighbours four at a time: their indices, then their rows, are in flight together (one neighbour per trip is an index
  // load -> dependent row load -> compare chain). Measured: no change at DGCNN's sizes (39.7 us per call either way) —
  // with 32 k points x 16-64 lanes resident the chip hides the chain, and the kernel sits on the L2 -> CU bandwidth of
  // its K row gathers (168-671 MB per call); kept because small batches do not have that occupancy.
  constexpr int U = 4;
  for (int k0 = 0; k0 < a.K; k0 += U) {
    int jj[U];
#pragma unroll
    for (int u = 0; u < U; ++u) jj[u] = nb[k0 + u < a.K ? k0 + u : k0];
    float4 vv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) vv[u] = *reinterpret_cast<const float4*>(Pb + (int64_t)jj[u] * 2 * a.C);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (k0 + u >= a.K) break;
      const float v[4] = {vv[u].x, vv[u].y, vv[u].z, vv[u].w};
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (v[e] > best[e]) best[e] = v[e], bj[e] = jj[u];
    }
  }
  const float qq[4] = {q.x, q.y, q.z, q.w};
  float o[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float z = best[e] + qq[e];
    o[e] = z > 0.f ? z : z * a.slope;
  }
  const int64_t off = ((int64_t)b * a.N + i) * a.C + 4 * l;
  *reinterpret_cast<float4*>(a.out + off) = make_float4(o[0], o[1], o[2], o[3]);
  *reinterpret_cast<int4*>(a.arg + off) = make_int4(bj[0], bj[1], bj[2], bj[3]);
  if (a.out2) *reinterpret_cast<float4*>(a.out2 + ((int64_t)b * a.N + i) * a.ld2 + 4 * l) = make_float4(o[0], o[1], o[2], o[3]);
}

// backward: one thread per (b, i, c): dQ written, dP scattered with float atomics into the zero-filled P half
__global__ __launch_bounds__(256) void edge_max_bwd_kernel(const float* g, int64_t ldg, const float* out, const int32_t* arg,
                                                           int N, int C, float slope, float* gPQ) {
  const int b = blockIdx.y;
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (int64_t)N * C) return;
  const int i = (int)(e / C), c = (int)(e - (int64_t)i * C);
  const int64_t o = (int64_t)b * N * C + e;
  const float gp = g[((int64_t)b * N + i) * ldg + c] * (out[o] > 0.f ? 1.f : slope);
  float* base = gPQ + (int64_t)b * N * 2 * C;
  base[(int64_t)i * 2 * C + C + c] = gp;
  atomicAdd(base + (int64_t)arg[o] * 2 * C + c, gp);
}

// The same backward without global atomics: a workgroup owns CH channels of one cloud and accumulates the scattered
// dP rows in LDS ([N][CH+1] floats, the +1 spreads the rows over the banks), then writes its column block of dP once.
// Each lane's global atomic above lands in a different row (the arg-max neighbour differs per channel), i.e. 64
// separate L2 transactions per wave instruction: 130 us per call on average in DGCNN's four layers at B=32, N=1024.
template <int CH>
__global__ __launch_bounds__(256) void edge_max_bwd_lds_kernel(const float* __restrict__ g, int64_t ldg,
                                                               const float* __restrict__ out,
                                                               const int32_t* __restrict__ arg, int N, int C, float slope,
                                                               float* __restrict__ gPQ) {
  extern __shared__ float emb_acc[];   // [N][CH+1]
  constexpr int ST = CH + 1;
  const int b = blockIdx.y, c0 = blockIdx.x * CH;
  for (int e = threadIdx.x; e < N * ST; e += 256) emb_acc[e] = 0.f;
  __syncthreads();
  float* base = gPQ + (int64_t)b * N * 2 * C;
  for (int i = threadIdx.x; i < N; i += 256) {
    const int64_t o = ((int64_t)b * N + i) * C + c0;
    float gv[CH], ov[CH];
    int av[CH];
#pragma unroll
    for (int q = 0; q < CH; q += 4) {
      const float4 a = *reinterpret_cast<const float4*>(g + ((int64_t)b * N + i) * ldg + c0 + q);
      const float4 w = *reinterpret_cast<const float4*>(out + o + q);
      const int4 r = *reinterpret_cast<const int4*>(arg + o + q);
      gv[q] = a.x, gv[q + 1] = a.y, gv[q + 2] = a.z, gv[q + 3] = a.w;
      ov[q] = w.x, ov[q + 1] = w.y, ov[q + 2] = w.z, ov[q + 3] = w.w;
      av[q] = r.x, av[q + 1] = r.y, av[q + 2] = r.z, av[q + 3] = r.w;
    }
#pragma unroll
    for (int q = 0; q < CH; ++q) {
      gv[q] *= ov[q] > 0.f ? 1.f : slope;
      atomicAdd(&emb_acc[min(max(av[q], 0), N - 1) * ST + q], gv[q]);
    }
    float* qrow = base + (int64_t)i * 2 * C + C + c0;
#pragma unroll
    for (int q = 0; q < CH; q += 4) *reinterpret_cast<float4*>(qrow + q) = make_float4(gv[q], gv[q + 1], gv[q + 2], gv[q + 3]);
  }
  __syncthreads();
  for (int r = threadIdx.x; r < N; r += 256) {
    float* prow = base + (int64_t)r * 2 * C + c0;
#pragma unroll
    for (int q = 0; q < CH; q += 4)
      *reinterpret_cast<float4*>(prow + q) =
          make_float4(emb_acc[r * ST + q], emb_acc[r * ST + q + 1], emb_acc[r * ST + q + 2], emb_acc[r * ST + q + 3]);
  }
}

// ---------------------------------------------------------------------------------------------------------
// Global pooling head of DGCNN / CurveNet (model/dgcnn.py:317-320, model/curvenet.py:64-67):
//   z = leaky_slope(Y) (slope 0 = ReLU); out[b, c] = max_i z[b,i,c]; out[b, C + c] = mean_i z[b,i,c]
// in ONE pass over Y [B,N,C] (the reference: activation pass + two reductions + cat), and the backward
//   gY[b,i,c] = leaky'(Y[b,i,c]) * (gmax[b,c] * [i == arg[b,c]] + gmean[b,c] / N)
// in one pass as well (autograd: scatter into zeros + expand + add + activation backward).
// Workgroup = (64 channels, cloud b); 16 row groups x 64 channels, fixed-order LDS combine => deterministic.
// ---------------------------------------------------------------------------------------------------------
constexpr int AP_RG = 16;   // row groups (waves) per workgroup

__global__ __launch_bounds__(64 * AP_RG) void act_pool_fwd_kernel(const float* Y, int N, int C, float slope, float* out,
                                                                  int32_t* arg) {
  __shared__ float s_mx[AP_RG][64], s_sm[AP_RG][64];
  __shared__ int s_ai[AP_RG][64];
  const int b = blockIdx.y, c = blockIdx.x * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6;
  const bool live = c < C;
  const int rows = (N + AP_RG - 1) / AP_RG, r0 = rg * rows, r1 = (r0 + rows < N) ? r0 + rows : N;
  float mx = -__builtin_inff(), sm = 0.f;
  int ai = r0 < N ? r0 : 0;
  if (live) {
    const float* col = Y + (int64_t)b * N * C + c;
    int i = r0;
    for (; i + 8 <= r1; i += 8) {       // 8 independent row loads in flight
      float y[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) y[e] = col[(int64_t)(i + e) * C];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float z = y[e] > 0.f ? y[e] : y[e] * slope;
        sm += z;
        if (z > mx) mx = z, ai = i + e;   // strict: lowest row wins ties (torch.max)
      }
    }
    for (; i < r1; ++i) {
      const float y = col[(int64_t)i * C];
      const float z = y > 0.f ? y : y * slope;
      sm += z;
      if (z > mx) mx = z, ai = i;
    }
  }
  s_mx[rg][threadIdx.x & 63] = mx, s_sm[rg][threadIdx.x & 63] = sm, s_ai[rg][threadIdx.x & 63] = ai;
  __syncthreads();
  if (rg == 0 && live) {
    const int l = threadIdx.x & 63;
    float m = s_mx[0][l], t = s_sm[0][l];
    int a = s_ai[0][l];
#pragma unroll
    for (int g = 1; g < AP_RG; ++g) {
      t += s_sm[g][l];
      if (s_mx[g][l] > m) m = s_mx[g][l], a = s_ai[g][l];
    }
    out[(int64_t)b * 2 * C + c] = m;
    out[(int64_t)b * 2 * C + C + c] = t / (float)N;
    arg[(int64_t)b * C + c] = a;
  }
}

__global__ __launch_bounds__(256) void act_pool_bwd_kernel(const float* Y, const float* gout, const int32_t* arg, int N,
                                                           int C, float slope, float* gY) {
  const int b = blockIdx.y;
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;     // over N * C / 4 (float4 of channels)
  const int c4 = C / 4;
  if (e >= (int64_t)N * c4) return;
  const int i = (int)(e / c4), c = 4 * (int)(e - (int64_t)i * c4);
  const float4 y = *reinterpret_cast<const float4*>(Y + ((int64_t)b * N + i) * C + c);
  const float4 gm = *reinterpret_cast<const float4*>(gout + (int64_t)b * 2 * C + c);
  const float4 ga = *reinterpret_cast<const float4*>(gout + (int64_t)b * 2 * C + C + c);
  const int4 a = *reinterpret_cast<const int4*>(arg + (int64_t)b * C + c);
  const float inv = 1.f / (float)N;
  float4 g;
  g.x = (y.x > 0.f ? 1.f : slope) * ((a.x == i ? gm.x : 0.f) + ga.x * inv);
  g.y = (y.y > 0.f ? 1.f : slope) * ((a.y == i ? gm.y : 0.f) + ga.y * inv);
  g.z = (y.z > 0.f ? 1.f : slope) * ((a.z == i ? gm.z : 0.f) + ga.z * inv);
  g.w = (y.w > 0.f ? 1.f : slope) * ((a.w == i ? gm.w : 0.f) + ga.w * inv);
  *reinterpret_cast<float4*>(gY + ((int64_t)b * N + i) * C + c) = g;
}

}  // namespace pc3d

using namespace pc3d;

extern "C" int pc3d_knn_feat_f32(const float* x, int B, int N, int C, int K, int32_t* idx, void* stream) {
  PC3D_REQUIRE(B >= 0 && N >= 1 && K >= 1 && K <= N && K <= 64, "pc3d_knn_feat_f32: bad sizes B=%d N=%d K=%d (K <= 64)", B, N, K);
  PC3D_REQUIRE(C >= 8 && C <= 128 && C % 8 == 0, "pc3d_knn_feat_f32: C=%d must be a multiple of 8 in [8,128]", C);
  PC3D_REQUIRE(B <= 65535, "pc3d_knn_feat_f32: B=%d exceeds grid.y limit", B);
  if (B == 0) return PC3D_OK;
  PC3D_REQUIRE(x && idx, "pc3d_knn_feat_f32: null pointer");
  KnnFeatArgs a{x, N, C, K, idx, B};
  const size_t lds = (size_t)(KF_Q * KF_LD + KF_Q * (C + 4)) * sizeof(float);   // 25.6 KiB (C=64) / 33.8 KiB (C=128)
  auto* kern = C == 64 ? (K >= 2 ? knn_feat_kernel<8, true> : knn_feat_kernel<8, false>)
             : C == 128 ? (K >= 2 ? knn_feat_kernel<16, true> : knn_feat_kernel<16, false>)
                        : (K >= 2 ? knn_feat_kernel<0, true> : knn_feat_kernel<0, false>);
  hipLaunchKernelGGL(kern, dim3(xcd_grid(cdiv(N, KF_Q) * B)), dim3(KF_T), lds, as_stream(stream), a);
  PC3D_LAUNCH_CHECK("pc3d_knn_feat_f32");
  return PC3D_OK;
}

static int gather_max_launch(const char* name, const float* P, const int32_t* idx, const float* sign, int B, int N, int S,
                             int C, int K, float* out, int32_t* arg, void* stream) {
  PC3D_REQUIRE(B >= 0 && N >= 1 && S >= 1 && C >= 1 && K >= 1, "%s: bad sizes", name);
  PC3D_REQUIRE(B <= 65535, "%s: B=%d exceeds grid.y limit", name, B);
  if (B == 0) return PC3D_OK;
  PC3D_REQUIRE(P && idx && out, "%s: null pointer", name);
  GMaxArgs a{P, idx, sign, N, C, K, out, arg, S, B};
  if (C % 4 == 0 && C <= 1024) {
    const int lpp = C / 4, ppw = 256 / lpp;
    hipLaunchKernelGGL(gather_max4_kernel, dim3(xcd_grid(cdiv(S, ppw) * B)), dim3(256), 0, as_stream(stream), a, lpp);
  } else {
    hipLaunchKernelGGL(gather_max_kernel, dim3(xcd_grid(cdiv(S, 4) * B)), dim3(256), 0, as_stream(stream), a);
  }
  PC3D_LAUNCH_CHECK(name);
  return PC3D_OK;
}

static int gather_max_bwd_launch(const char* name, const float* g, const int32_t* arg, int B, int N, int S, int C,
                                 float* gP, int deterministic, void* stream) {
  PC3D_REQUIRE(B >= 0 && N >= 1 && S >= 1 && C >= 1, "%s: bad sizes", name);
  PC3D_REQUIRE(B <= 65535, "%s: B=%d exceeds grid.y limit", name, B);
  if (B == 0) return PC3D_OK;
  PC3D_REQUIRE(g && arg && gP, "%s: null pointer", name);
  // deterministic: one wavefront per (cloud, channel slice) accumulates in LDS in source order (det.hip), no zero fill
  if (deterministic) return arg_scatter_det(name, g, C, nullptr, arg, B, S, N, C, 0.f, gP, 0, stream);
  hipStream_t st = as_stream(stream);
  hipError_t e = zero_async(gP, (size_t)B * N * C, st);
  if (e != hipSuccess) {
    set_error("%s: zero fill failed: %s", name, hipGetErrorString(e));
    return (int)e;
  }
  hipLaunchKernelGGL(gather_max_bwd_kernel, dim3((unsigned)(((int64_t)S * C + 255) / 256), B), dim3(256), 0, st, g, arg,
                     N, S, C, gP);
  PC3D_LAUNCH_CHECK(name);
  return PC3D_OK;
}

extern "C" int pc3d_gather_max_f32(const float* P, const int32_t* idx, const float* sign, int B, int N, int C, int K,
                                   float* out, int32_t* arg, void* stream) {
  return gather_max_launch("pc3d_gather_max_f32", P, idx, sign, B, N, N, C, K, out, arg, stream);
}

extern "C" int pc3d_gather_max_bwd_f32(const float* g, const int32_t* arg, int B, int N, int C, float* gP,
                                       int deterministic, void* stream) {
  return gather_max_bwd_launch("pc3d_gather_max_bwd_f32", g, arg, B, N, N, C, gP, deterministic, stream);
}

extern "C" int pc3d_gather_max_rows_f32(const float* P, const int32_t* idx, int B, int N, int S, int C, int K, float* out,
                                        int32_t* arg, void* stream) {
  return gather_max_launch("pc3d_gather_max_rows_f32", P, idx, nullptr, B, N, S, C, K, out, arg, stream);
}

extern "C" int pc3d_gather_max_rows_bwd_f32(const float* g, const int32_t* arg, int B, int N, int S, int C, float* gP,
                                            int deterministic, void* stream) {
  return gather_max_bwd_launch("pc3d_gather_max_rows_bwd_f32", g, arg, B, N, S, C, gP, deterministic, stream);
}

static int edge_max_launch(const char* nm, const float* PQ, const int32_t* idx, int B, int N, int C, int K, float slope,
                           float* out, int32_t* arg, float* out2, int64_t ld2, void* stream) {
  PC3D_REQUIRE(B >= 0 && N >= 1 && C >= 4 && C % 4 == 0 && C <= 1024 && K >= 1, "%s: bad sizes (C %% 4 == 0, C <= 1024)", nm);
  if (B == 0) return PC3D_OK;
  PC3D_REQUIRE(PQ && idx && out && arg, "%s: null pointer", nm);
  PC3D_REQUIRE(!out2 || (ld2 >= C && ld2 % 4 == 0 && (reinterpret_cast<uintptr_t>(out2) & 15) == 0),
               "%s: the second output needs a 16-byte aligned start and a row stride ld2 >= C, ld2 %% 4 == 0", nm);
  EdgeMaxArgs a{PQ, idx, N, C, K, slope, out, arg, B, out2, ld2};
  const int lpp = C / 4, ppw = 256 / lpp;
  hipLaunchKernelGGL(edge_max_kernel, dim3(xcd_grid(cdiv(N, ppw) * B)), dim3(256), 0, as_stream(stream), a, lpp);
  PC3D_LAUNCH_CHECK(nm);
  return PC3D_OK;
}

extern "C" int pc3d_edge_max_f32(const float* PQ, const int32_t* idx, int B, int N, int C, int K, float slope,
                                 float* out, int32_t* arg, void* stream) {
  return edge_max_launch("pc3d_edge_max_f32", PQ, idx, B, N, C, K, slope, out, arg, nullptr, 0, stream);
}

extern "C" int pc3d_edge_max_cat_f32(const float* PQ, const int32_t* idx, int B, int N, int C, int K, float slope,
                                     float* out, int32_t* arg, float* out2, int64_t ld2, void* stream) {
  return edge_max_launch("pc3d_edge_max_cat_f32", PQ, idx, B, N, C, K, slope, out, arg, out2, ld2, stream);
}

extern "C" int pc3d_edge_max_bwd_f32(const float* g, int64_t ldg, const float* out, const int32_t* arg, int B, int N, int C,
                                     float slope, float* gPQ, int deterministic, void* stream) {
  PC3D_REQUIRE(B >= 0 && N >= 1 && C >= 1 && ldg >= C, "pc3d_edge_max_bwd_f32: bad sizes (row stride of g smaller than C)");
  PC3D_REQUIRE(B <= 65535, "pc3d_edge_max_bwd_f32: B=%d exceeds grid.y limit", B);
  if (B == 0) return PC3D_OK;
  PC3D_REQUIRE(g && out && arg && gPQ, "pc3d_edge_max_bwd_f32: null pointer");
  // deterministic: the LDS accumulator tile belongs to ONE wavefront, which walks the points in order (det.hip); the
  // kernels below share a tile between four wavefronts, whose ds_add_f32 interleave differently from run to run
  if (deterministic) return arg_scatter_det("pc3d_edge_max_bwd_f32", g, ldg, out, arg, B, N, N, C, slope, gPQ, 1, stream);
  hipStream_t st = as_stream(stream);
  const bool al = ((reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(arg) |
                    reinterpret_cast<uintptr_t>(gPQ)) & 15) == 0 && ldg % 4 == 0;
  if (al && C % 8 == 0 && (size_t)N * 9 * sizeof(float) <= 64 * 1024) {
    hipLaunchKernelGGL(edge_max_bwd_lds_kernel<8>, dim3(C / 8, B), dim3(256), (size_t)N * 9 * sizeof(float), st, g, ldg, out,
                       arg, N, C, slope, gPQ);
    PC3D_LAUNCH_CHECK("pc3d_edge_max_bwd_f32");
    return PC3D_OK;
  }
  if (al && C % 4 == 0 && (size_t)N * 5 * sizeof(float) <= 64 * 1024) {
    hipLaunchKernelGGL(edge_max_bwd_lds_kernel<4>, dim3(C / 4, B), dim3(256), (size_t)N * 5 * sizeof(float), st, g, ldg, out,
                       arg, N, C, slope, gPQ);
    PC3D_LAUNCH_CHECK("pc3d_edge_max_bwd_f32");
    return PC3D_OK;
  }
  hipError_t e = zero_async(gPQ, (size_t)B * N * 2 * C, st);
  if (e != hipSuccess) {
    set_error("pc3d_edge_max_bwd_f32: zero fill failed: %s", hipGetErrorString(e));
    return (int)e;
  }
  hipLaunchKernelGGL(edge_max_bwd_kernel, dim3((unsigned)(((int64_t)N * C + 255) / 256), B), dim3(256), 0, st, g, ldg, out,
                     arg, N, C, slope, gPQ);
  PC3D_LAUNCH_CHECK("pc3d_edge_max_bwd_f32");
  return PC3D_OK;
}

extern "C" int pc3d_edge_max_bwd_sum_f32(const float* g, int64_t ldg, const float* g2, int64_t ldg2, const float* out,
                                         const int32_t* arg, int B, int N, int C, float slope, float* gPQ, void* stream) {
  const char* nm = "pc3d_edge_max_bwd_sum_f32";
  PC3D_REQUIRE(B >= 0 && N >= 1 && C >= 1 && ldg >= C && ldg2 >= C, "%s: bad sizes (a row stride smaller than C)", nm);
  if (B == 0) return PC3D_OK;
  PC3D_REQUIRE(g && g2 && out && arg && gPQ, "%s: null pointer", nm);
  return arg_scatter_det(nm, g, ldg, out, arg, B, N, N, C, slope, gPQ, 1, stream, 0, g2, ldg2);
}

extern "C" int pc3d_edge_max_bwd_slice_f32(const float* g, int64_t ldg, const float* out, const int32_t* arg, int B, int N, int C,
                                           float slope, float* gPQ, int slice, void* stream) {
  PC3D_REQUIRE(B >= 0 && N >= 1 && C >= 1 && ldg >= C, "pc3d_edge_max_bwd_slice_f32: bad sizes (row stride of g smaller than C)");
  PC3D_REQUIRE(B <= 65535, "pc3d_edge_max_bwd_slice_f32: B=%d exceeds grid.y limit", B);
  if (B == 0) return PC3D_OK;
  PC3D_REQUIRE(g && out && arg && gPQ, "pc3d_edge_max_bwd_slice_f32: null pointer");
  return arg_scatter_det("pc3d_edge_max_bwd_slice_f32", g, ldg, out, arg, B, N, N, C, slope, gPQ, 1, stream, slice);
}

extern "C" int pc3d_act_pool_f32(const float* Y, int B, int N, int C, float slope, float* out, int32_t* arg,
                                 void* stream) {
  PC3D_REQUIRE(B >= 0 && N >= 1 && C >= 4 && C % 4 == 0, "pc3d_act_pool_f32: bad sizes B=%d N=%d C=%d (C %% 4 == 0)", B, N, C);
  PC3D_REQUIRE(B <= 65535, "pc3d_act_pool_f32: B=%d exceeds grid.y limit", B);
  if (B == 0) return PC3D_OK;
  PC3D_REQUIRE(Y && out && arg, "pc3d_act_pool_f32: null pointer");
  hipLaunchKernelGGL(act_pool_fwd_kernel, dim3(cdiv(C, 64), B), dim3(64 * AP_RG), 0, as_stream(stream), Y, N, C, slope, out, arg);
  PC3D_LAUNCH_CHECK("pc3d_act_pool_f32");
  return PC3D_OK;
}

extern "C" int pc3d_act_pool_bwd_f32(const float* Y, const float* gout, const int32_t* arg, int B, int N, int C,
                                     float slope, float* gY, void* stream) {
  PC3D_REQUIRE(B >= 0 && N >= 1 && C >= 4 && C % 4 == 0, "pc3d_act_pool_bwd_f32: bad sizes");
  PC3D_REQUIRE(B <= 65535, "pc3d_act_pool_bwd_f32: B=%d exceeds grid.y limit", B);
  if (B == 0) return PC3D_OK;
  PC3D_REQUIRE(Y && gout && arg && gY, "pc3d_act_pool_bwd_f32: null pointer");
  hipLaunchKernelGGL(act_pool_bwd_kernel, dim3((unsigned)(((int64_t)N * (C / 4) + 255) / 256), B), dim3(256), 0,
                     as_stream(stream), Y, gout, arg, N, C, slope, gY);
  PC3D_LAUNCH_CHECK("pc3d_act_pool_bwd_f32");
  return PC3D_OK;
}
