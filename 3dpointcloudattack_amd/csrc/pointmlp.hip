// K8 — PointNet per-point MLP 3 -> C1(64) -> C2(128) -> C3 (1x1 conv + folded eval-BN + ReLU) fused with the
// max-pool over points, forward and backward-to-input.  gfx950, fp32-input MFMA (exact fp32 FMA chains).
//
// Replaces: model/pointnet.py:34-37 (STN3d tower) and :110-123 (PointNetfeat trunk): three Conv1d+BN(+ReLU) and
// torch.max over N, which materialise [B,64,N], [B,128,N] and [B,1024,N] activations (134 MB at B=32,N=1024).
//
// Forward: workgroup = (batch b, tile of 128 points), 8 waves. Layers 1-2 are computed once per tile into LDS
// (h2 tile [128 pts][128 ch], 66 KiB) and their ReLU decisions leave as per-point bit masks; layer 3 runs on v_mfma_f32_32x32x2_f32 with POINTS on the MFMA row index
// and CHANNELS on the column (lane) index, so the max over a tile's points is an in-register reduction
// (16 accumulator values per lane and channel block) + one cross-half shuffle; W3 rows stream from L2 as float4 per lane with a
// permuted-k order shared by both operands. Only (max, argmax) per (b, tile, channel) leaves the CU; a tiny second
// kernel folds the tiles. The [B,C3,N] activation is never written.
//
// Backward: max-pool routes each channel's gradient to ONE point, so dgrad of layer 3 is sparse:
// workgroup = (b, tile of 32 points) gathers the channels whose argmax falls in its tile in ascending channel order
// (deterministic, no atomics), applies the ReLU decisions of layers 1-2 that the FORWARD launch recorded as per-point
// bit masks (24 B per point — nothing is recomputed) and chains W2^T on MFMA and W1^T on the VALU.
#include <stdlib.h>
#include "pc3d_common.h"

namespace pc3d {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int PM_C1 = 64;
constexpr int PM_C2 = 128;
constexpr int PM_TP = 128;           // forward: points per workgroup
constexpr int PM_LD1 = PM_C1 + 4;    // LDS row strides (floats): +4 keeps ds_read_b128 conflict-free
constexpr int PM_LD2 = PM_C2 + 4;
constexpr int PM_BTP = 32;           // backward: points per workgroup
constexpr int PM_MAXC3 = 1024;       // backward: widest pooled layer held in LDS

struct PMFwdArgs {
  PtsView x;
  int N, C3, ntiles;
  const float* T;  // [B,3,3] or null: x'[n,:] = x[n,:] @ T  (model/pointnet.py:106-109)
  const float *W1, *b1, *W2, *b2, *W3, *b3;
  float* part_val;    // [B, ntiles, C3]
  int32_t* part_idx;  // [B, ntiles, C3]
  uint64_t* mask1;    // [B,N]    bit c  = (layer-1 output c of the point > 0), or null
  uint32_t* mask2;    // [B,N,4]  word j bit r = (layer-2 output 32j+r of the point > 0), or null
  // optional "transform head" (T == null): T[b] = th_W [9,th_K] . th_in[b] + th_b — STN3d's fc3 (+ identity folded
  // into th_b, model/pointnet.py:45-47) evaluated in this kernel's prologue instead of a launch of its own; tile 0 of
  // every cloud also writes it to th_out [B,9] for the backward
  const float *th_in, *th_W, *th_b;
  int th_K;
  float* th_out;
};

// T: the 3 x 3 transform of THIS cloud (or null)
__device__ __forceinline__ void load_point(const PtsView& x, const float* T, int b, int n, int N, float& px,
                                           float& py, float& pz) {
  px = py = pz = 0.f;
  if (n < N) {
    const float* p = x.p + (int64_t)b * x.bs + (int64_t)n * x.ps;
    const float x0 = p[0], x1 = p[x.cs], x2 = p[2 * x.cs];
    if (T) {
      const float* t = T;
      px = __builtin_fmaf(x2, t[6], __builtin_fmaf(x1, t[3], x0 * t[0]));
      py = __builtin_fmaf(x2, t[7], __builtin_fmaf(x1, t[4], x0 * t[1]));
      pz = __builtin_fmaf(x2, t[8], __builtin_fmaf(x1, t[5], x0 * t[2]));
    } else {
      px = x0, py = x1, pz = x2;
    }
  }
}

// h1[p][c] = relu(W1[c,:].x_p + b1[c]) for `npts` points whose coordinates sit in xs[3][npts]; lanes run over c.
// m1 (may be null) receives, for point p, the 64-bit mask of its positive channels: the wave's lanes ARE the 64
// channels, so the mask is one ballot; lane i keeps point i's mask and the wave stores `per` consecutive words.
template <int NPTS, int NTHREADS>
__device__ __forceinline__ void layer1_to_lds(const float* xs, float* h1, const float* W1, const float* b1,
                                              uint64_t* m1 = nullptr, int nvalid = NPTS) {
  const int c = threadIdx.x & (PM_C1 - 1);
  const int grp = threadIdx.x >> 6;
  constexpr int per = NPTS / (NTHREADS / 64);
  static_assert(per <= 64, "one mask word per lane");
  const float w0 = W1[c * 3 + 0], w1 = W1[c * 3 + 1], w2 = W1[c * 3 + 2], bb = b1[c];
  unsigned long long mine = 0ull;
#pragma unroll 4
  for (int i = 0; i < per; ++i) {
    const int p = grp * per + i;
    float v = __builtin_fmaf(w2, xs[2 * NPTS + p], __builtin_fmaf(w1, xs[NPTS + p], __builtin_fmaf(w0, xs[p], bb)));
    h1[p * PM_LD1 + c] = fmaxf(v, 0.f);
    const unsigned long long bal = __builtin_amdgcn_ballot_w64(v > 0.f);
    if (c == i) mine = bal;
  }
  if (m1 != nullptr && c < per && grp * per + c < nvalid) m1[grp * per + c] = mine;
}

constexpr int PM_MAXC3F = 1024;      // forward: widest layer 3 (cross-wave max scratch aliases the h2 tile)
constexpr int PM_FT = 512;           // forward: threads per workgroup (8 waves = 2 per SIMD: one wave's epilogue /
                                     // operand waits overlap the other's MFMAs)

__global__ __launch_bounds__(PM_FT) __attribute__((amdgpu_waves_per_eu(2, 2))) void pointmlp3_max_fwd_kernel(PMFwdArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[PM_TP * PM_LD2 + 3 * PM_TP];  // 69,120 B static
  float* h1 = lds;                       // [128][68]   (dead after layer 2)
  float* h2 = lds;                       // [128][132]  (overwrites h1 behind a barrier)
  float* xs = lds + PM_TP * PM_LD2;      // [3][128]
  const int tile = blockIdx.x, b = blockIdx.y;
  const int n0 = tile * PM_TP;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;

  const float* Tb = a.T ? a.T + (int64_t)b * 9 : nullptr;
  // raw coordinates first (their load latency overlaps the transform head's), the transform is applied afterwards
  float px = 0.f, py = 0.f, pz = 0.f;
  if (threadIdx.x < PM_TP) load_point(a.x, nullptr, b, n0 + threadIdx.x, a.N, px, py, pz);
  if (a.th_in) {   // 9 outputs x th_K: 32 lanes per output, strided partial sums + a half-wave reduction
    __shared__ float Ts[9];
    if (threadIdx.x < 9 * 32) {
      const int j = threadIdx.x >> 5, l = threadIdx.x & 31;
      const float* in = a.th_in + (int64_t)b * a.th_K;
      const float* w = a.th_W + (int64_t)j * a.th_K;
      float sacc = 0.f;
      if (a.th_K == 256) {      // STN3d: all eight operand pairs in flight at once
        float iv[8], wv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) iv[u] = in[l + 32 * u], wv[u] = w[l + 32 * u];
#pragma unroll
        for (int u = 0; u < 8; ++u) sacc = __builtin_fmaf(iv[u], wv[u], sacc);
      } else {
        for (int k = l; k < a.th_K; k += 32) sacc = __builtin_fmaf(in[k], w[k], sacc);
      }
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) sacc += __shfl_xor(sacc, o, 32);
      if (l == 0) {
        const float t = sacc + a.th_b[j];
        Ts[j] = t;
        if (tile == 0) a.th_out[(int64_t)b * 9 + j] = t;
      }
    }
    __syncthreads();
    Tb = Ts;
  }
  if (threadIdx.x < PM_TP) {
    if (Tb) {   // x' = x @ T (model/pointnet.py:106-109)
      const float x0 = px, x1 = py, x2 = pz;
      px = __builtin_fmaf(x2, Tb[6], __builtin_fmaf(x1, Tb[3], x0 * Tb[0]));
      py = __builtin_fmaf(x2, Tb[7], __builtin_fmaf(x1, Tb[4], x0 * Tb[1]));
      pz = __builtin_fmaf(x2, Tb[8], __builtin_fmaf(x1, Tb[5], x0 * Tb[2]));
    }
    xs[threadIdx.x] = px;
    xs[PM_TP + threadIdx.x] = py;
    xs[2 * PM_TP + threadIdx.x] = pz;
  }
  __syncthreads();
  layer1_to_lds<PM_TP, PM_FT>(xs, h1, a.W1, a.b1, a.mask1 ? a.mask1 + (int64_t)b * a.N + n0 : nullptr, a.N - n0);
  __syncthreads();

  // ---- layer 2 on MFMA: D[pt][c2] = sum_k h1[pt][k] W2[c2][k]; wave owns c2 block (wave&3) and 2 of the 4 point tiles
  {
    const int c2b = wave & 3, tl0 = (wave >> 2) * 2;
    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    const float* wrow = a.W2 + (32 * c2b + r) * PM_C1 + 4 * h;
#pragma unroll
    for (int t = 0; t < PM_C1 / 8; ++t) {
      const float4 bw = *reinterpret_cast<const float4*>(wrow + 8 * t);
      float4 av[2];
#pragma unroll
      for (int tl = 0; tl < 2; ++tl)
        av[tl] = *reinterpret_cast<const float4*>(h1 + ((tl0 + tl) * 32 + r) * PM_LD1 + 8 * t + 4 * h);
#pragma unroll
      for (int tl = 0; tl < 2; ++tl) {
        acc[tl] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[tl].x, bw.x, acc[tl], 0, 0, 0);
        acc[tl] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[tl].y, bw.y, acc[tl], 0, 0, 0);
        acc[tl] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[tl].z, bw.z, acc[tl], 0, 0, 0);
        acc[tl] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[tl].w, bw.w, acc[tl], 0, 0, 0);
      }
    }
    __syncthreads();  // every wave is done reading h1
    const float bias = a.b2[32 * c2b + r];
    unsigned long long mine = 0ull;   // lane 16*tl + e keeps the ballot of (tl, e): points pt(e,0) [low word], pt(e,1) [high]
#pragma unroll
    for (int tl = 0; tl < 2; ++tl)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int pt = (tl0 + tl) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        const float v = acc[tl][e] + bias;
        h2[pt * PM_LD2 + 32 * c2b + r] = fmaxf(v, 0.f);
        const unsigned long long bal = __builtin_amdgcn_ballot_w64(v > 0.f);
        if (lane == 16 * tl + e) mine = bal;
      }
    if (a.mask2 != nullptr && lane < 32) {   // the backward's layer-2 ReLU mask: exactly the decisions taken here
      const int tl = lane >> 4, e = lane & 15;
      const int pt0 = n0 + (tl0 + tl) * 32 + (e & 3) + 8 * (e >> 2);
      uint32_t* m2 = a.mask2 + ((int64_t)b * a.N) * 4 + c2b;
      if (pt0 < a.N) m2[(int64_t)pt0 * 4] = (uint32_t)mine;
      if (pt0 + 4 < a.N) m2[(int64_t)(pt0 + 4) * 4] = (uint32_t)(mine >> 32);
    }
  }
  __syncthreads();

  // ---- layer 3 + max over the tile's points.
  // Wave w owns point tile (w & 3) = 32 points and half of the channel blocks ((w >> 2) selects blocks
  // [16*(w>>2), +16) of 32 channels when C3 = 1024). Its A operands (h2 of its 32 points, all 128 k) are read from LDS
  // ONCE into 64 VGPRs; only W3 streams (float4 per lane per 8 k, L2-resident, 4 waves share each row block).
  // One 16-register accumulator per channel block: a single dependent MFMA chain issues back-to-back on gfx950
  // (32x32x2 f32: issue interval = dependent latency = 64 cycles), and the SIMD's second wave fills every gap.
  const int ptile = wave & 3, cgrp = wave >> 2;
  float4 areg[PM_C2 / 8];
#pragma unroll
  for (int t = 0; t < PM_C2 / 8; ++t)
    areg[t] = *reinterpret_cast<const float4*>(h2 + (ptile * 32 + r) * PM_LD2 + 8 * t + 4 * h);
  __syncthreads();  // h2 fully consumed into registers: the LDS region is reused for the cross-wave max below
  float* pv = lds;                                        // [4 point tiles][C3]
  int* pi = reinterpret_cast<int*>(lds + 4 * PM_MAXC3F);  // [4 point tiles][C3]
  const int nblk = a.C3 / 32;
  const int blk_per_grp = (nblk + 1) / 2;
  const int cb_end = (cgrp + 1) * blk_per_grp < nblk ? (cgrp + 1) * blk_per_grp : nblk;
  // W3 rows are double-buffered in registers ACROSS channel blocks: while block cb runs its 64 MFMAs, the 16 float4
  // of block cb+1 are fetched (one load per 4 MFMAs, pinned with sched_group_barrier so hipcc cannot sink them to
  // just-in-time), i.e. every load has a full block (~4k cycles) to land.
  auto load_row = [&](float4 (&dst)[PM_C2 / 8], int cb) {
    const int cbc = cb < cb_end ? cb : cb_end - 1;  // past the end: harmless re-load of the last block
    const float* wrow = a.W3 + (int64_t)(cbc * 32 + r) * PM_C2 + 4 * h;
#pragma unroll
    for (int t = 0; t < PM_C2 / 8; ++t) dst[t] = *reinterpret_cast<const float4*>(wrow + 8 * t);
  };
  auto run_block = [&](const float4 (&cur)[PM_C2 / 8], float4 (&nxt)[PM_C2 / 8], int cb) {
    const int cbn = (cb + 1 < cb_end) ? cb + 1 : cb_end - 1;
    const float* nrow = a.W3 + (int64_t)(cbn * 32 + r) * PM_C2 + 4 * h;
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
    for (int t = 0; t < PM_C2 / 8; ++t) {
      nxt[t] = *reinterpret_cast<const float4*>(nrow + 8 * t);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[t].x, cur[t].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[t].y, cur[t].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[t].z, cur[t].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[t].w, cur[t].w, acc, 0, 0, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // 1 VMEM read
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);  // 4 MFMA
    }
    const int ch = cb * 32 + r;
    float best = -__builtin_inff();
    // the winning accumulator row as an inline constant per select; the point index is formed once after the loop (and the
    // bounds test only runs in a ragged last tile): the epilogue's VALU instructions do not overlap the other wave's MFMAs
    int be = 0;
    const int base = n0 + ptile * 32;
    if (base + 32 <= a.N) {             // (uniform)
#pragma unroll
      for (int e = 0; e < 16; ++e)      // ascending point index in e for fixed h: strict > keeps the lowest
        if (acc[e] > best) best = acc[e], be = (e & 3) + 8 * (e >> 2);
    } else {
#pragma unroll
      for (int e = 0; e < 16; ++e)
        if (base + (e & 3) + 8 * (e >> 2) + 4 * h < a.N && acc[e] > best) best = acc[e], be = (e & 3) + 8 * (e >> 2);
    }
    int bi = best == -__builtin_inff() ? base : base + be + 4 * h;
    argmax_xor32(best, bi);
    if (h == 0) {
      pv[ptile * a.C3 + ch] = best;
      pi[ptile * a.C3 + ch] = bi;
    }
  };
  float4 bwA[PM_C2 / 8], bwB[PM_C2 / 8];
  const int cb_begin = cgrp * blk_per_grp;
  if (cb_begin < cb_end) {
    load_row(bwA, cb_begin);
    for (int cb = cb_begin; cb < cb_end; cb += 2) {
      run_block(bwA, bwB, cb);
      if (cb + 1 < cb_end) run_block(bwB, bwA, cb + 1);
    }
  }
  __syncthreads();
  for (int ch = threadIdx.x; ch < a.C3; ch += PM_FT) {
    float best = pv[ch];
    int bi = pi[ch];
#pragma unroll
    for (int t = 1; t < 4; ++t) {  // ascending point tile: strict > keeps the lowest point index on ties
      const float v = pv[t * a.C3 + ch];
      if (v > best) {
        best = v;
        bi = pi[t * a.C3 + ch];
      }
    }
    const int64_t o = ((int64_t)b * a.ntiles + tile) * a.C3 + ch;
    a.part_val[o] = best + a.b3[ch];
    a.part_idx[o] = bi;
  }
}

// fold tiles: pooled[b,c] = max_t part[b,t,c] (first tile wins ties => lowest point index), optional ReLU
__global__ __launch_bounds__(256) void pointmlp3_fold_kernel(const float* part_val, const int32_t* part_idx,
                                                             int ntiles, int C3, int relu_last, float* pooled,
                                                             int32_t* argidx) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  if (c >= C3) return;
  const int64_t base = (int64_t)b * ntiles * C3 + c;
  float best = part_val[base];
  int bi = part_idx[base];
  for (int t = 1; t < ntiles; ++t) {
    const float v = part_val[base + (int64_t)t * C3];
    if (v > best) {
      best = v;
      bi = part_idx[base + (int64_t)t * C3];
    }
  }
  if (relu_last) best = fmaxf(best, 0.f);
  pooled[(int64_t)b * C3 + c] = best;
  argidx[(int64_t)b * C3 + c] = bi;
}

// ---------------------------------------------------------------------------------------------------------
struct PMBwdArgs {
  PtsView x;
  int N, C3;
  const float* T;
  const float *W1, *b1, *W2, *b2, *W3, *W2T;  // W2T = W2 transposed, [64][128] row-major
  const int32_t* argidx;  // [B,C3]
  const uint64_t* mask1;  // [B,N]   layer-1 ReLU decisions of the forward launch (bit c)
  const uint32_t* mask2;  // [B,N,4] layer-2 ReLU decisions (word j, bit r = channel 32j+r)
  const float* g;         // [B,C3] upstream gradient on pooled (already masked for relu_last by the caller)
  PtsViewMut gx;          // T == null: gradient wrt the tower input; T given: gradient wrt the RAW points x
  float* part_gT;         // [B, ntiles, 16] per-tile partial of d/dT (9 used) or null
  int accumulate;         // gx += instead of gx =
};

// Workgroup = (batch b, 32 points), 4 waves; <= 36 KiB LDS so four workgroups share a CU (the kernel is a chain of
// dependent latencies, residency is what hides them).
//  A. channels whose arg-max lies in the tile are compacted IN CHANNEL ORDER (block prefix sum) into two lists
//     (points 0-15 / 16-31) and their rows g[c]*W3[c,:] are accumulated into g2s[pt][128] — ordered => deterministic;
//     each thread then clears the entries of its own column whose layer-2 ReLU was off in the FORWARD launch (the
//     forward kernel hands over its decisions as bit masks: nothing is recomputed, and no decision can differ);
//  C. g1 = (g2 masked) . W2 on MFMA with K = 128 split over wave pairs, masked by the forward's layer-1 bits;
//  D. g' = g1 . W1 on the VALU, then the x' = x @ T chain (dL/dx, per-tile partial of dL/dT).
__global__ __launch_bounds__(256) void pointmlp3_max_bwd_kernel(PMBwdArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[PM_BTP * PM_LD2 + PM_BTP * PM_LD1 + 4 * PM_BTP + PM_MAXC3 + (3 * PM_MAXC3) / 2];  // 36.4 KB
  float* g2s = lds;                                   // [32][132]
  float* h1s = g2s + PM_BTP * PM_LD2;                 // [32][68]   g1
  float* xs = h1s + PM_BTP * PM_LD1;                  // [3][32]    scratch of phase D
  int* s_scan = reinterpret_cast<int*>(xs + 3 * PM_BTP);   // [32] wave totals
  float* s_g = xs + 4 * PM_BTP;                       // [C3]
  short* s_n = reinterpret_cast<short*>(s_g + PM_MAXC3);   // [C3] local point index or -1
  short* list0 = s_n + PM_MAXC3;                      // [C3] channels hitting points 0..15, ascending
  short* list1 = list0 + PM_MAXC3;                    // [C3] channels hitting points 16..31
  __shared__ uint32_t s_m2[PM_BTP][4];
  __shared__ uint64_t s_m1[PM_BTP];
  const int tile = blockIdx.x, b = blockIdx.y;
  const int n0 = tile * PM_BTP;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;

  // ---- A1. classify 4 consecutive channels per thread, block-wide ordered compaction.
  // Issue the (tiny) arg-max / gradient / mask loads FIRST: vector-memory results return in issue order, so anything
  // issued behind the 32 KiB weight prefetch below would wait for all of it.
  int ld_n[PM_MAXC3 / 256];
  float ld_g[PM_MAXC3 / 256];
#pragma unroll
  for (int e = 0; e < PM_MAXC3 / 256; ++e) {
    const int c = tid * (PM_MAXC3 / 256) + e;
    ld_n[e] = (c < a.C3) ? a.argidx[(int64_t)b * a.C3 + c] : -1;
    ld_g[e] = (c < a.C3) ? a.g[(int64_t)b * a.C3 + c] : 0.f;
  }
  uint32_t ld_m2 = 0u;
  uint64_t ld_m1 = 0ull;
  if (tid < PM_BTP * 4 && n0 + (tid >> 2) < a.N) ld_m2 = a.mask2[((int64_t)b * a.N + n0) * 4 + tid];
  if (tid < PM_BTP && n0 + tid < a.N) ld_m1 = a.mask1[(int64_t)b * a.N + n0 + tid];
  // MFMA B operand of phase C depends on nothing: fetch it now so its L2 latency hides under phase A
  float4 w2tr[PM_C2 / 16];
  {
    // phase C: wave = (j block wave&1, K half wave>>1): k in [64*(wave>>1), +64)
    const float* wtrow = a.W2T + (32 * (wave & 1) + r) * PM_C2 + 64 * (wave >> 1) + 4 * h;
#pragma unroll
    for (int t = 0; t < PM_C2 / 16; ++t) w2tr[t] = *reinterpret_cast<const float4*>(wtrow + 8 * t);
  }
  int cnt0 = 0, cnt1 = 0;
  int myn[PM_MAXC3 / 256];
#pragma unroll
  for (int e = 0; e < PM_MAXC3 / 256; ++e) {
    const int c = tid * (PM_MAXC3 / 256) + e;
    int n = -1;
    if (c < a.C3) {
      n = ld_n[e] - n0;
      const float gv = ld_g[e];
      if (n < 0 || n >= PM_BTP || gv == 0.f) n = -1;
      s_g[c] = gv;
      s_n[c] = (short)n;
    }
    myn[e] = n;
    cnt0 += (n >= 0 && n < 16) ? 1 : 0;
    cnt1 += (n >= 16) ? 1 : 0;
  }
  if (tid < PM_BTP * 4) s_m2[tid >> 2][tid & 3] = ld_m2;
  if (tid < PM_BTP) s_m1[tid] = ld_m1;
  for (int i = tid; i < PM_BTP * PM_LD2; i += 256) g2s[i] = 0.f;
  int packed = cnt0 | (cnt1 << 16);
  int incl = packed;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int v = __shfl_up(incl, o, 64);
    if (lane >= o) incl += v;
  }
  if (lane == 63) s_scan[wave] = incl;
  __syncthreads();
  int base = 0, total = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const int v = s_scan[w];
    if (w < wave) base += v;
    total += v;
  }
  const int len0 = total & 0xffff, len1 = total >> 16;
  if (len0 + len1 == 0) {  // no critical point in this tile: gradient is exactly zero
    if (tid < 3 * PM_BTP && !a.accumulate) {
      const int p = tid & (PM_BTP - 1), c = tid >> 5;
      if (n0 + p < a.N) a.gx.p[(int64_t)b * a.gx.bs + (int64_t)(n0 + p) * a.gx.ps + c * a.gx.cs] = 0.f;
    }
    if (a.part_gT && tid < 16) a.part_gT[((int64_t)b * gridDim.x + tile) * 16 + tid] = 0.f;
    return;
  }
  {
    const int excl = base + incl - packed;
    int o0 = excl & 0xffff, o1 = excl >> 16;
#pragma unroll
    for (int e = 0; e < PM_MAXC3 / 256; ++e) {
      const int n = myn[e];
      const int c = tid * (PM_MAXC3 / 256) + e;
      if (n >= 0 && n < 16) list0[o0++] = (short)c;
      if (n >= 16) list1[o1++] = (short)c;
    }
  }
  __syncthreads();

  // ---- A2. ordered accumulation: thread (k, half) walks its half's list; loads are independent -> pipelined
  {
    const int k = tid & (PM_C2 - 1), ph = tid >> 7;
    const short* list = ph ? list1 : list0;
    const int len = ph ? len1 : len0;
    int i = 0;
    for (; i + 8 <= len; i += 8) {  // 8 independent W3 loads in flight per round trip
      int c[8];
      float w[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        c[e] = list[i + e];
        w[e] = a.W3[(int64_t)c[e] * PM_C2 + k];
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float* dst = g2s + (int)s_n[c[e]] * PM_LD2 + k;
        *dst = __builtin_fmaf(s_g[c[e]], w[e], *dst);
      }
    }
    for (; i + 2 <= len; i += 2) {
      const int c0 = list[i], c1 = list[i + 1];
      const float w0 = a.W3[(int64_t)c0 * PM_C2 + k], w1 = a.W3[(int64_t)c1 * PM_C2 + k];
      float* d0 = g2s + (int)s_n[c0] * PM_LD2 + k;
      *d0 = __builtin_fmaf(s_g[c0], w0, *d0);
      float* d1 = g2s + (int)s_n[c1] * PM_LD2 + k;
      *d1 = __builtin_fmaf(s_g[c1], w1, *d1);
    }
    for (; i < len; ++i) {
      const int c = list[i];
      float* dst = g2s + (int)s_n[c] * PM_LD2 + k;
      *dst = __builtin_fmaf(s_g[c], a.W3[(int64_t)c * PM_C2 + k], *dst);
    }
    // layer-2 ReLU of the forward pass: this thread is the only writer of column k for its half's 16 points, so it
    // applies the mask to them without a barrier (bit k&31 of word k>>5 of the point's mask2)
#pragma unroll
    for (int p = 0; p < 16; ++p) {
      const int pt = 16 * ph + p;
      if (!((s_m2[pt][k >> 5] >> (k & 31)) & 1u)) g2s[pt * PM_LD2 + k] = 0.f;
    }
  }
  __syncthreads();

  // ---- C. g1[pt][j] = sum_k2 g2[pt][k2] W2[k2][j] on MFMA: wave = (j block wave&1, K half wave>>1)
  {
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    const int jb = wave & 1, kh = wave >> 1;
#pragma unroll
    for (int t = 0; t < PM_C2 / 16; ++t) {
      const float4 bw = w2tr[t];
      const float4 av = *reinterpret_cast<const float4*>(g2s + r * PM_LD2 + 64 * kh + 8 * t + 4 * h);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bw.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bw.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bw.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bw.w, acc, 0, 0, 0);
    }
    // the upper K half hands its partial tile to the lower one through LDS (fixed order: deterministic); g2s is dead
    // once every wave has read its A operands, so it doubles as the exchange buffer (one [32][33] tile per j block)
    __syncthreads();
    float* cr = g2s + jb * (32 * 33);
    if (kh == 1) {
#pragma unroll
      for (int e = 0; e < 16; ++e) cr[((e & 3) + 8 * (e >> 2) + 4 * h) * 33 + r] = acc[e];
    }
    __syncthreads();
    if (kh == 0) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int pt = (e & 3) + 8 * (e >> 2) + 4 * h;
        const bool on = (s_m1[pt] >> (32 * jb + r)) & 1ull;          // layer-1 ReLU of the forward pass
        h1s[pt * PM_LD1 + 32 * jb + r] = on ? (acc[e] + cr[pt * 33 + r]) : 0.f;
      }
    }
  }
  __syncthreads();

  // ---- D. g'[p][c] = sum_j W1[j][c] g1[p][j]  (gradient wrt the tower input x' = x @ T)
  float* gp = xs;  // [3][32] scratch
  if (wave < 3 && lane < PM_BTP) {   // wave = coordinate c (uniform -> W1 comes through scalar loads), lane = point
    const int p = lane, c = wave;
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int j = 0; j < PM_C1; j += 4) {
      const float4 hv = *reinterpret_cast<const float4*>(h1s + p * PM_LD1 + j);
      s0 = __builtin_fmaf(a.W1[j * 3 + c], hv.x, s0);
      s1 = __builtin_fmaf(a.W1[(j + 1) * 3 + c], hv.y, s1);
      s0 = __builtin_fmaf(a.W1[(j + 2) * 3 + c], hv.z, s0);
      s1 = __builtin_fmaf(a.W1[(j + 3) * 3 + c], hv.w, s1);
    }
    gp[c * PM_BTP + p] = s0 + s1;
  }
  __syncthreads();
  if (wave == 0) {  // lanes 0..31 = the tile's points; lanes 32..63 contribute zeros to the reductions
    const int p = lane & (PM_BTP - 1);
    const bool live = lane < PM_BTP;
    const float g0 = live ? gp[p] : 0.f, g1v = live ? gp[PM_BTP + p] : 0.f, g2v = live ? gp[2 * PM_BTP + p] : 0.f;
    float o0 = g0, o1 = g1v, o2 = g2v;
    if (a.T) {
      // x' = x @ T  =>  dL/dx[c] = sum_c' g'[c'] T[c][c'] ;  dL/dT[c][c'] = sum_p x[p][c] g'[p][c']
      const float* t = a.T + (int64_t)b * 9;
      o0 = __builtin_fmaf(g2v, t[2], __builtin_fmaf(g1v, t[1], g0 * t[0]));
      o1 = __builtin_fmaf(g2v, t[5], __builtin_fmaf(g1v, t[4], g0 * t[3]));
      o2 = __builtin_fmaf(g2v, t[8], __builtin_fmaf(g1v, t[7], g0 * t[6]));
      if (a.part_gT) {
        float xr[3] = {0.f, 0.f, 0.f};
        if (live && n0 + p < a.N) {
          const float* xp = a.x.p + (int64_t)b * a.x.bs + (int64_t)(n0 + p) * a.x.ps;
          xr[0] = xp[0], xr[1] = xp[a.x.cs], xr[2] = xp[2 * a.x.cs];
        }
        const float gv[3] = {g0, g1v, g2v};
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
          for (int d = 0; d < 3; ++d) {
            const float sum = wave_sum(xr[c] * gv[d]);
            if (lane == 0) a.part_gT[((int64_t)b * gridDim.x + tile) * 16 + c * 3 + d] = sum;
          }
        if (lane >= 9 && lane < 16) a.part_gT[((int64_t)b * gridDim.x + tile) * 16 + lane] = 0.f;
      }
    }
    if (live && n0 + p < a.N) {
      float* q = a.gx.p + (int64_t)b * a.gx.bs + (int64_t)(n0 + p) * a.gx.ps;
      if (a.accumulate) {
        q[0] += o0, q[a.gx.cs] += o1, q[2 * a.gx.cs] += o2;
      } else {
        q[0] = o0, q[a.gx.cs] = o1, q[2 * a.gx.cs] = o2;
      }
    }
  }
}

}  // namespace pc3d

using namespace pc3d;

extern "C" int pc3d_pointmlp3_tile_points(void) { return PM_TP; }
extern "C" int pc3d_pointmlp3_bwd_tile_points(void) { return PM_BTP; }

static int pm_fwd_launch(const float* x, int64_t x_bs, int64_t x_ps, int64_t x_cs, int B, int N, const float* T,
                         const float* th_in, const float* th_W, const float* th_b, int th_K, float* th_out,
                         const float* W1, const float* b1, const float* W2, const float* b2, const float* W3,
                         const float* b3, int C1, int C2, int C3, int relu_last, float* part_val, int32_t* part_idx,
                         float* pooled, int32_t* argidx, uint64_t* mask1, uint32_t* mask2, void* stream) {
  PC3D_REQUIRE(B >= 0 && N >= 1, "pc3d_pointmlp3_max_fwd_f32: bad sizes B=%d N=%d", B, N);
  PC3D_REQUIRE(C1 == PM_C1 && C2 == PM_C2 && C3 >= 32 && C3 % 32 == 0 && C3 <= PM_MAXC3F,
               "pc3d_pointmlp3_max_fwd_f32: unsupported widths %d/%d/%d (need 64/128/multiple of 32 <= 1024)", C1, C2, C3);
  PC3D_REQUIRE(B <= 65535, "pc3d_pointmlp3_max_fwd_f32: B=%d exceeds grid.y limit", B);
  if (B == 0) return PC3D_OK;
  PC3D_REQUIRE(x && W1 && b1 && W2 && b2 && W3 && b3 && part_val && part_idx,
               "pc3d_pointmlp3_max_fwd_f32: null pointer");
  PC3D_REQUIRE((pooled == nullptr) == (argidx == nullptr),
               "pc3d_pointmlp3_max_fwd_f32: pooled and argidx must both be given or both be NULL");
  PC3D_REQUIRE((mask1 == nullptr) == (mask2 == nullptr),
               "pc3d_pointmlp3_max_fwd_f32: mask1 and mask2 must both be given or both be NULL");
  const int ntiles = cdiv(N, PM_TP);
  PMFwdArgs a{{x, x_bs, x_ps, x_cs}, N, C3, ntiles, T, W1, b1, W2, b2, W3, b3, part_val, part_idx, mask1, mask2,
              th_in, th_W, th_b, th_K, th_out};
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(pointmlp3_max_fwd_kernel, dim3(ntiles, B), dim3(PM_FT), 0, st, a);
  PC3D_LAUNCH_CHECK("pc3d_pointmlp3_max_fwd_f32");
  if (pooled) {  // NULL: leave the per-tile partials unfolded (a fused consumer, or kernel-only timing)
    hipLaunchKernelGGL(pointmlp3_fold_kernel, dim3(cdiv(C3, 256), B), dim3(256), 0, st, part_val, part_idx, ntiles,
                       C3, relu_last, pooled, argidx);
    PC3D_LAUNCH_CHECK("pc3d_pointmlp3_max_fwd_f32/fold");
  }
  return PC3D_OK;
}

extern "C" int pc3d_pointmlp3_max_fwd_f32(const float* x, int64_t x_bs, int64_t x_ps, int64_t x_cs, int B, int N,
                                          const float* T, const float* W1, const float* b1, const float* W2,
                                          const float* b2, const float* W3, const float* b3, int C1, int C2,
                                          int C3, int relu_last, float* part_val, int32_t* part_idx,
                                          float* pooled, int32_t* argidx, uint64_t* mask1, uint32_t* mask2,
                                          void* stream) {
  return pm_fwd_launch(x, x_bs, x_ps, x_cs, B, N, T, nullptr, nullptr, nullptr, 0, nullptr, W1, b1, W2, b2, W3, b3, C1, C2,
                       C3, relu_last, part_val, part_idx, pooled, argidx, mask1, mask2, stream);
}

extern "C" int pc3d_pointmlp3_max_fwd_th_f32(const float* x, int64_t x_bs, int64_t x_ps, int64_t x_cs, int B, int N,
                                             const float* th_in, const float* th_W, const float* th_b, int th_K,
                                             float* T_out, const float* W1, const float* b1, const float* W2,
                                             const float* b2, const float* W3, const float* b3, int C1, int C2, int C3,
                                             int relu_last, float* part_val, int32_t* part_idx, float* pooled,
                                             int32_t* argidx, uint64_t* mask1, uint32_t* mask2, void* stream) {
  PC3D_REQUIRE(th_in && th_W && th_b && T_out && th_K >= 1,
               "pc3d_pointmlp3_max_fwd_th_f32: the transform head needs its input, weights [9,K], bias [9] and T_out");
  return pm_fwd_launch(x, x_bs, x_ps, x_cs, B, N, nullptr, th_in, th_W, th_b, th_K, T_out, W1, b1, W2, b2, W3, b3, C1, C2,
                       C3, relu_last, part_val, part_idx, pooled, argidx, mask1, mask2, stream);
}

extern "C" int pc3d_pointmlp3_max_bwd_f32(const float* x, int64_t x_bs, int64_t x_ps, int64_t x_cs, int B, int N,
                                          const float* T, const float* W1, const float* b1, const float* W2,
                                          const float* b2, const float* W3, const float* W2T, int C1, int C2,
                                          int C3, const int32_t* argidx, const uint64_t* mask1,
                                          const uint32_t* mask2, const float* g_pooled, float* grad_x,
                                          int64_t gx_bs, int64_t gx_ps, int64_t gx_cs, float* part_gT,
                                          int accumulate, void* stream) {
  PC3D_REQUIRE(B >= 0 && N >= 1, "pc3d_pointmlp3_max_bwd_f32: bad sizes B=%d N=%d", B, N);
  PC3D_REQUIRE(C1 == PM_C1 && C2 == PM_C2 && C3 >= 32 && C3 % 32 == 0 && C3 <= PM_MAXC3,
               "pc3d_pointmlp3_max_bwd_f32: unsupported widths %d/%d/%d", C1, C2, C3);
  PC3D_REQUIRE(B <= 65535, "pc3d_pointmlp3_max_bwd_f32: B=%d exceeds grid.y limit", B);
  if (B == 0) return PC3D_OK;
  PC3D_REQUIRE(x && W1 && b1 && W2 && b2 && W3 && W2T && argidx && mask1 && mask2 && g_pooled && grad_x,
               "pc3d_pointmlp3_max_bwd_f32: null pointer");
  PMBwdArgs a{{x, x_bs, x_ps, x_cs}, N, C3, T, W1, b1, W2, b2, W3, W2T, argidx, mask1, mask2, g_pooled, {grad_x, gx_bs, gx_ps, gx_cs},
              part_gT, accumulate};
  hipLaunchKernelGGL(pointmlp3_max_bwd_kernel, dim3(cdiv(N, PM_BTP), B), dim3(256), 0, as_stream(stream), a);
  PC3D_LAUNCH_CHECK("pc3d_pointmlp3_max_bwd_f32");
  return PC3D_OK;
}
