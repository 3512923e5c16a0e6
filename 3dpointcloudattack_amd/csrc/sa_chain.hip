// A set-abstraction MLP after its (per-point) first layer as ONE launch: gather -> layer 1 (generated) -> layer 2 ->
// layer 3 -> max over the group, without the [B*S*ns, C2] layer-2 output ever leaving the chip.
//
//   out[g, c] = relu( max_{j < ns} ( W3 . relu( W2 . relu(P[b, idx[g, j], :] + Bc[g, :]) + b2 ) )[c] + b3[c] )
//
// = model/pointnet2_utils.py:173-199 (PointNetSetAbstraction.forward: index_points + centring + three Conv2d 1x1 + BN +
// ReLU + max over nsample) with the first layer in its per-point form (group.hip). Until round 3 this was two launches —
// pc3d_gemm_nt_gather_f32 (layers 1 + 2) and the group-max GEMM (layer 3 + max) — with the [1 048 576, 64] layer-2 output
// of SSG's SA1 (268 MB at B=64, N=2048; 268 MB again at SA2) written by the first and read back by the second: both
// launches sat on that traffic (3.4 TB/s; 0.40 / 0.56 of the fp32-MFMA peak end to end).
//
// Workgroup = 128 rows (128 / ns groups), 8 waves in 4 x 2. One LDS buffer AH [128][max(C1, C2) + 4] holds first the
// generated layer-1 rows (the A operand of layer 2), then — once every wave has finished reading it — the layer-2
// output (the A operand of layer 3); the weights stream through a double-buffered [128][32 + 4] slice per K step, as in
// gemm_nt_kernel (same operand k-order, same fp32 MFMA sequence: results are bit-identical to the two-launch form).
// The signs the backward needs leave as bits: one byte per 4 generated layer-1 elements, one word per 32 layer-2 outputs.
#include "pc3d_common.h"

namespace pc3d {

using sc_f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int SC_T = 512, SC_BM = 128, SC_BK = 32, SC_LD = SC_BK + 4;

struct SaChainArgs {
  const float* P;       // [B*NA, C1], row stride ldp
  const float* Bc;      // [B*S, C1]
  const int32_t* idx;   // [B*S*ns]
  const float* W2;      // [C2, C1]
  const float* b2;      // [C2]
  const float* W3;      // [C3, C2]
  const float* b3;      // [C3]
  int64_t ldp;
  int M, NA, S, ns, C1, C2, C3;
  int ns_shift, c4_shift;   // log2(ns), log2(C1 / 4): both powers of two (the entry point checks) — the row / group / channel
                            // of a generated element come from shifts; an integer division costs ~25 VALU instructions, and
                            // fp32 VALU work does NOT overlap fp32 MFMA work of other waves on the same SIMD (they share the
                            // multipliers): with three divisions per gathered float4 the non-MFMA skeleton of a tile took
                            // as long as its MFMAs (measured: 375 us full, 190 us with the MFMAs removed)
  uint8_t* mask1;       // [M, C1/4]
  uint32_t* mask2;      // [M, C2/32]
  float* out;           // [M/ns, C3]
  int64_t* arg;         // [M/ns, C3]
  const int32_t* tb = nullptr;      // [tiles][4] block table of the streaming kernel (see there) or null
  const int32_t* ntiles = nullptr;  // [1] tiles in the table
};

// one K step of a 32 x (32 TN) wave tile: A from the resident buffer (row stride lda, k offset k0), B from the staged slice
// (the operand fragments of group t + 1 are requested before the MFMAs of group t are issued: with two waves per SIMD
// nothing else hides the ds_read latency of a "read, wait, 4 MFMAs" loop)
template <int TN, bool FIRST = false>
__device__ __forceinline__ void sc_step(sc_f32x16 (&acc)[TN], const float* __restrict__ Arow, const float* __restrict__ Bs, int wn,
                                        int r, int h, int ldb = SC_LD) {
  const float* Ap = Arow + 4 * h;
  const float* Bp = Bs + (wn + r) * ldb + 4 * h;
  float4 av[2], bv[2][TN];
  av[0] = *reinterpret_cast<const float4*>(Ap);
#pragma unroll
  for (int j = 0; j < TN; ++j) bv[0][j] = *reinterpret_cast<const float4*>(Bp + j * 32 * ldb);
#pragma unroll
  for (int t = 0; t < SC_BK / 8; ++t) {
    const int c = t & 1, n = c ^ 1;
    if (t + 1 < SC_BK / 8) {
      av[n] = *reinterpret_cast<const float4*>(Ap + 8 * (t + 1));
#pragma unroll
      for (int j = 0; j < TN; ++j) bv[n][j] = *reinterpret_cast<const float4*>(Bp + j * 32 * ldb + 8 * (t + 1));
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c].x, bv[c][j].x, acc[j], 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c].y, bv[c][j].y, acc[j], 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c].z, bv[c][j].z, acc[j], 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c].w, bv[c][j].w, acc[j], 0, 0, 0);
    }
  }
}

// TN2: 32-column MFMA tiles per wave in layer 2 (its output has <= 64 TN2 columns)
template <int TN2>
__global__ __launch_bounds__(SC_T, 2) void sa_chain_kernel(SaChainArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sc_lds[];
  const int LDH = (a.C1 > a.C2 ? a.C1 : a.C2) + 4;
  float* AH = sc_lds;                                   // [128][LDH]
  float* Ws = sc_lds + SC_BM * LDH;                      // [2][128][SC_LD]
  float* pv = Ws + 2 * 128 * SC_LD;                      // [4][128] partial maxima of a column tile (ns > 32)
  int* pi = reinterpret_cast<int*>(pv + 4 * 128);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int wm = (wave >> 1) * 32;                       // this wave's 32 rows
  const int m0 = blockIdx.x * SC_BM;
  const int lrow = tid >> 3, lk = (tid & 7) * 4;         // weight staging: 8 threads cover a row's 32 k

  // ---- the tile's four 32-row blocks in ORIGINAL row space: block k = rows 32 k .. 32 k + 31. Without a table the tile
  // is blocks 4 t .. 4 t + 3; with one (a.tb, pc3d_sa_blocks_i32) it is up to four blocks of consecutive whole groups
  // that hold at least one listed point — blocks of nothing but the ball query's padding copies are left out and the
  // others move up, so a launch has fewer tiles, not idle waves (skipping them in place changed nothing: a tile waits
  // for its active blocks).
  int* s_blk = pi + 4 * 128;                             // [4] block ids (-1: empty slot), [4] their clouds
  if (a.tb) {
    if ((int)blockIdx.x >= a.ntiles[0]) return;          // (uniform, before any barrier)
    if (tid < 4) s_blk[tid] = a.tb[blockIdx.x * 4 + tid];
  } else if (tid < 4) {
    const int k = (m0 >> 5) + tid;
    s_blk[tid] = ((int64_t)k * 32 < a.M) ? k : -1;
  }
  __syncthreads();
  if (tid < 4) s_blk[4 + tid] = s_blk[tid] >= 0 ? ((s_blk[tid] * 32) >> a.ns_shift) / a.S : 0;
  __syncthreads();
  const int blk[4] = {s_blk[0], s_blk[1], s_blk[2], s_blk[3]};
  const int bcl[4] = {s_blk[4], s_blk[5], s_blk[6], s_blk[7]};
  auto pick4 = [](const int (&v)[4], int i) { return i == 0 ? v[0] : (i == 1 ? v[1] : (i == 2 ? v[2] : v[3])); };
  const bool act = pick4(blk, wm >> 5) >= 0;              // this wave's block (uniform per wave)

  // ---- gather + layer 1: AH[row][:] = relu(P[src(row)] + Bc[group(row)]), one sign bit per element to mask1
  {
    const int c4n = a.C1 >> 2;                           // float4 per row
    for (int f0 = tid; f0 < SC_BM * c4n; f0 += 4 * SC_T) {
      float4 v[4], c[4];
      int gmv[4], rowv[4], c4v[4];
      bool live[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int f = f0 + u * SC_T;
        live[u] = f < SC_BM * c4n;
        const int row = live[u] ? f >> a.c4_shift : 0, c4 = live[u] ? f & (c4n - 1) : 0;
        const int kb = pick4(blk, row >> 5);
        if (kb < 0) live[u] = false;                       // an empty slot: nobody will read these rows
        const int gm = kb * 32 + (row & 31);
        gmv[u] = gm, rowv[u] = row, c4v[u] = c4;
        v[u] = make_float4(0.f, 0.f, 0.f, 0.f), c[u] = v[u];
        if (live[u]) {
          const int g = gm >> a.ns_shift, p = a.idx[gm];
          const int bb = pick4(bcl, row >> 5);
          c[u] = *reinterpret_cast<const float4*>(a.Bc + (int64_t)g * a.C1 + 4 * c4);
          if ((unsigned)p < (unsigned)a.NA)
            v[u] = *reinterpret_cast<const float4*>(a.P + ((int64_t)bb * a.NA + p) * a.ldp + 4 * c4);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (!live[u]) continue;
        float4 x = make_float4(v[u].x + c[u].x, v[u].y + c[u].y, v[u].z + c[u].z, v[u].w + c[u].w);
        a.mask1[(int64_t)gmv[u] * c4n + c4v[u]] =
            (uint8_t)((x.x > 0.f ? 1 : 0) | (x.y > 0.f ? 2 : 0) | (x.z > 0.f ? 4 : 0) | (x.w > 0.f ? 8 : 0));
        x.x = x.x > 0.f ? x.x : 0.f, x.y = x.y > 0.f ? x.y : 0.f, x.z = x.z > 0.f ? x.z : 0.f, x.w = x.w > 0.f ? x.w : 0.f;
        *reinterpret_cast<float4*>(AH + rowv[u] * LDH + 4 * c4v[u]) = x;
      }
    }
  }

  // weight slice of K step k0 of W [ncols, K] into registers (rows past ncols read as zero)
  float4 wb[2];
  auto fetch_w = [&](const float* W, int ncols, int K, int n0, int k0) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int gn = n0 + q * 64 + lrow;
      wb[q] = (gn < ncols) ? *reinterpret_cast<const float4*>(W + (int64_t)gn * K + k0 + lk) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto stash_w = [&](float* dst) {
#pragma unroll
    for (int q = 0; q < 2; ++q) *reinterpret_cast<float4*>(dst + (q * 64 + lrow) * SC_LD + lk) = wb[q];
  };

  // ---- layer 2: H = relu(A . W2^T + b2), 128 x (64 TN2) per workgroup, wave (wm, wn2)
  const int wn2 = (wave & 1) * (32 * TN2);
  sc_f32x16 acc2[TN2];
#pragma unroll
  for (int j = 0; j < TN2; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc2[j][e] = 0.f;
  fetch_w(a.W2, a.C2, a.C1, 0, 0);
  stash_w(Ws);
  __syncthreads();                                       // AH (generated rows) and the first slice are in place
  int cur = 0;
  for (int k0 = 0; k0 < a.C1; k0 += SC_BK) {
    const bool more = k0 + SC_BK < a.C1;
    if (more) fetch_w(a.W2, a.C2, a.C1, 0, k0 + SC_BK);
    if (act) sc_step<TN2>(acc2, AH + (wm + r) * LDH + k0, Ws + cur * 128 * SC_LD, wn2, r, h);
    if (more) stash_w(Ws + (cur ^ 1) * 128 * SC_LD);
    __syncthreads();                                     // (after the last step: every wave has finished reading AH)
    cur ^= 1;
  }
  // first slice of layer 3's weights on its way while the epilogue below runs
  fetch_w(a.W3, a.C3, a.C2, 0, 0);
  // epilogue of layer 2 into AH: D[row][col]: lane holds column r of each 32-column tile, rows (e & 3) + 8 (e >> 2) + 4 h
#pragma unroll
  for (int j = 0; j < TN2; ++j) {
    if (!act) break;
    const int col = wn2 + j * 32 + r;
    const bool col_ok = col < a.C2;
    const float bj = col_ok ? a.b2[col] : 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = wm + (e & 3) + 8 * (e >> 2) + 4 * h;
      const float v = acc2[j][e] + bj;
      const unsigned long long bal = __builtin_amdgcn_ballot_w64(v > 0.f && col_ok);
      const int gm = pick4(blk, wm >> 5) * 32 + (row & 31);
      if (r == 0 && wn2 + j * 32 < a.C2)                  // (C2 % 32 == 0: a 32-column tile is all in or all out)
        a.mask2[(int64_t)gm * (a.C2 >> 5) + ((wn2 + j * 32) >> 5)] = (uint32_t)(h ? (bal >> 32) : bal);
      if (col_ok) AH[row * LDH + col] = v > 0.f ? v : 0.f;
    }
  }
  stash_w(Ws + cur * 128 * SC_LD);
  __syncthreads();                                       // AH now holds the layer-2 output; W3's first slice is staged

  // ---- layer 3 + group max, one 128-column tile of C3 at a time; wave (wm, wn3 = 64 (wave & 1))
  const int wn3 = (wave & 1) * 64;
  const int tpg = a.ns >> 5;                             // 32-row MFMA tiles per group
  for (int n0 = 0; n0 < a.C3; n0 += 128) {
    sc_f32x16 acc3[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc3[j][e] = 0.f;
    for (int k0 = 0; k0 < a.C2; k0 += SC_BK) {
      const bool more_k = k0 + SC_BK < a.C2, more_n = n0 + 128 < a.C3;
      if (more_k) fetch_w(a.W3, a.C3, a.C2, n0, k0 + SC_BK);
      else if (more_n) fetch_w(a.W3, a.C3, a.C2, n0 + 128, 0);
      if (act) sc_step<2>(acc3, AH + (wm + r) * LDH + k0, Ws + cur * 128 * SC_LD, wn3, r, h);
      if (more_k || more_n) stash_w(Ws + (cur ^ 1) * 128 * SC_LD);
      __syncthreads();
      cur ^= 1;
    }
    // group-max epilogue (bias + ReLU after the max: both monotone). Per 32-row tile and column: in-lane over the 16
    // accumulator rows (ascending in e: strict > keeps the lowest row), across the lane halves (compare (value, row)).
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float best = -__builtin_inff();
      int bi = wm;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int rl = wm + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (act && acc3[j][e] > best) best = acc3[j][e], bi = rl;     // (an empty slot never wins)
      }
      argmax_xor32(best, bi);
      const int col = n0 + wn3 + j * 32 + r;
      if (tpg == 1) {                                    // ns == 32: the block IS the group
        const int64_t grp = pick4(blk, wm >> 5);
        if (h == 0 && act && col < a.C3) {
          a.out[grp * a.C3 + col] = fmaxf(best + a.b3[col], 0.f);
          a.arg[grp * a.C3 + col] = bi - wm;
        }
      } else if (h == 0) {
        pv[(wm >> 5) * 128 + wn3 + j * 32 + r] = best;
        pi[(wm >> 5) * 128 + wn3 + j * 32 + r] = bi;
      }
    }
    if (tpg > 1) {
      // a group's blocks sit in consecutive slots of ONE tile, its block 0 (always kept) first: slot sb heads a group when
      // its block id is a multiple of tpg; the group's other kept blocks follow in ascending row order (strict >: lowest row)
      __syncthreads();
      for (int t = tid; t < 4 * 128; t += SC_T) {
        const int sb = t >> 7, cl = t & 127;
        const int kb = pick4(blk, sb);
        if (kb < 0 || (kb & (tpg - 1)) != 0) continue;
        float best = pv[sb * 128 + cl];
        int bi = pi[sb * 128 + cl] - 32 * sb;              // row inside the group: 32 * (block of the group) + row of the block
        for (int s2 = sb + 1; s2 < 4; ++s2) {
          const int k2 = pick4(blk, s2);
          if (k2 < 0 || (k2 >> (a.ns_shift - 5)) != (kb >> (a.ns_shift - 5))) break;
          const float v = pv[s2 * 128 + cl];
          if (v > best) best = v, bi = pi[s2 * 128 + cl] - 32 * s2 + 32 * (k2 - kb);
        }
        const int64_t grp = kb >> (a.ns_shift - 5);
        const int col = n0 + cl;
        if (col < a.C3) {
          a.out[grp * a.C3 + col] = fmaxf(best + a.b3[col], 0.f);
          a.arg[grp * a.C3 + col] = bi;
        }
      }
      __syncthreads();                                   // pv / pi are rewritten by the next column tile
    }
  }
}


// ---------------------------------------------------------------------------------------------------------
// The same chain for SMALL weights (SSG's first level: 64 -> 64 -> 128, 48 KB): W2 and W3 stay in LDS for the lifetime
// of a PERSISTENT workgroup that loops over 64-row tiles. In the streaming kernel above a K step of this level is 16-32
// MFMAs per wave (~0.5-1 us) — shorter than the L2 latency of the next weight slice it prefetches, and six barriers per
// tile: 340 us for 25.8 GFLOP. Here a tile costs four barriers and no weight traffic, the next tile's rows are gathered
// into registers while the current tile's layer 3 runs, and two 4-wave workgroups per CU are in different phases.
// Same operand order and MFMA sequence: bit-identical results. ns in {32, 64}; C3 <= 128 per resident column tile.
// ---------------------------------------------------------------------------------------------------------
constexpr int SR_T = 256, SR_BM = 64;

// A 32 x (32 TN) wave tile over K = 32 KS from two LDS-resident operands, fully unrolled, with the operand fragments of
// group g + 1 (8 k each) requested BEFORE the MFMAs of group g are issued: with two waves per SIMD nothing else hides the
// ds_read latency of a "read, wait, 4 MFMAs" loop. The chain's first MFMA takes a literal zero accumulator.
template <int TN, int KS, int LDA, int LDB>
__device__ __forceinline__ void sc_mm(sc_f32x16 (&acc)[TN], const float* __restrict__ Ap, const float* __restrict__ Bp) {
  // Ap = A + row * LDA + 4 h;  Bp = B + (wn + r) * LDB + 4 h   (this lane's operand rows, its half of every 8-k group)
  constexpr int NG = 4 * KS;
  float4 av[2], bv[2][TN];
  av[0] = *reinterpret_cast<const float4*>(Ap);
#pragma unroll
  for (int j = 0; j < TN; ++j) bv[0][j] = *reinterpret_cast<const float4*>(Bp + j * 32 * LDB);
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int c = g & 1, n = c ^ 1;
    if (g + 1 < NG) {
      av[n] = *reinterpret_cast<const float4*>(Ap + 8 * (g + 1));
#pragma unroll
      for (int j = 0; j < TN; ++j) bv[n][j] = *reinterpret_cast<const float4*>(Bp + j * 32 * LDB + 8 * (g + 1));
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      if (g == 0) {
        const sc_f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c].x, bv[c][j].x, zero, 0, 0, 0);
      } else {
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c].x, bv[c][j].x, acc[j], 0, 0, 0);
      }
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c].y, bv[c][j].y, acc[j], 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c].z, bv[c][j].z, acc[j], 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c].w, bv[c][j].w, acc[j], 0, 0, 0);
    }
  }
}

// C1, C2 in {32, 64}: every LDS stride and trip count is a compile-time constant (operand addresses become immediate
// offsets: the address arithmetic of the generic form was a sixth of a tile's VALU instructions)
template <int C1, int C2>
__global__ __launch_bounds__(SR_T, 2) void sa_chain_res_kernel(SaChainArgs a, int tiles) {
  extern __shared__ __attribute__((aligned(16))) float sc_lds[];
  constexpr int LDH = (C1 > C2 ? C1 : C2) + 4, LD2 = C1 + 4, LD3 = C2 + 4;
  constexpr int c4n = C1 / 4, G4 = C1 / 16;              // float4 per generated row / per thread (64 rows x c4n = 256 x G4)
  float* W2s = sc_lds;                                   // [64][LD2] (rows past C2 are zero)
  float* W3s = W2s + 64 * LD2;                            // [round128(C3)][LD3] (rows past C3 are zero)
  const int C3p = (a.C3 + 127) & ~127;
  float* AH = W3s + C3p * LD3;                            // [64][LDH]
  float* pv = AH + SR_BM * LDH;                           // [2][128]
  int* pi = reinterpret_cast<int*>(pv + 2 * 128);
  float* b2s = reinterpret_cast<float*>(pi + 2 * 128);     // [64] + [C3p]: the biases, read by every tile's epilogues
  float* b3s = b2s + 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int wm = (wave >> 1) * 32;

  for (int f = tid; f < 64 * c4n; f += SR_T) {            // weights + biases: once per workgroup
    const int row = f / c4n, c4 = f % c4n;
    *reinterpret_cast<float4*>(W2s + row * LD2 + 4 * c4) =
        row < C2 ? *reinterpret_cast<const float4*>(a.W2 + (int64_t)row * C1 + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (int f = tid; f < C3p * (C2 / 4); f += SR_T) {
    const int row = f / (C2 / 4), c4 = f % (C2 / 4);
    *reinterpret_cast<float4*>(W3s + row * LD3 + 4 * c4) =
        row < a.C3 ? *reinterpret_cast<const float4*>(a.W3 + (int64_t)row * C2 + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (int c = tid; c < 64; c += SR_T) b2s[c] = c < C2 ? a.b2[c] : 0.f;
  for (int c = tid; c < C3p; c += SR_T) b3s[c] = c < a.C3 ? a.b3[c] : 0.f;

  // generated rows of a tile, G4 float4 per thread: element f = u * 256 + tid -> (row = f / c4n, c4 = f % c4n).
  // Two-stage prefetch: the INDICES of tile i+2 and the ROWS of tile i+1 are in flight while tile i computes — a row's
  // address depends on its index, and a load -> address -> load chain inside one stage would stall the wave for a memory
  // latency per tile right before its MFMAs (measured with the loads removed: 53 us of the 375 us first version).
  float4 gv[G4], gc[G4];
  int nidx[G4];
  auto load_idx = [&](int m0) {
#pragma unroll
    for (int u = 0; u < G4; ++u) {
      const int gm = m0 + (u * SR_T + tid) / c4n;
      nidx[u] = gm < a.M ? a.idx[gm] : -1;
    }
  };
  auto load_rows = [&](int m0) {
    const int g_first = m0 >> a.ns_shift, b_first = g_first / a.S;       // uniform: one division per tile
#pragma unroll
    for (int u = 0; u < G4; ++u) {
      const int f = u * SR_T + tid, row = f / c4n, c4 = f % c4n;
      const int gm = m0 + row;
      gv[u] = make_float4(0.f, 0.f, 0.f, 0.f), gc[u] = gv[u];
      if (gm < a.M) {
        const int g = gm >> a.ns_shift, p = nidx[u];
        const int bb = b_first + (g >= (b_first + 1) * a.S ? 1 : 0);       // a 64-row tile spans <= 2 groups
        gc[u] = *reinterpret_cast<const float4*>(a.Bc + (int64_t)g * C1 + 4 * c4);
        if ((unsigned)p < (unsigned)a.NA)
          gv[u] = *reinterpret_cast<const float4*>(a.P + ((int64_t)bb * a.NA + p) * a.ldp + 4 * c4);
      }
    }
  };
  int tile = blockIdx.x;
  if (tile < tiles) {
    load_idx(tile * SR_BM);
    load_rows(tile * SR_BM);
    if (tile + (int)gridDim.x < tiles) load_idx((tile + gridDim.x) * SR_BM);
  }
  const int wn2 = (wave & 1) * 32, wn3 = (wave & 1) * 64;
  const int tpg = a.ns >> 5;
  const float* Ap = AH + (wm + r) * LDH + 4 * h;           // this lane's A-operand row (both layers)
  for (; tile < tiles; tile += gridDim.x) {
    const int m0 = tile * SR_BM;
    __syncthreads();                                      // the previous tile's layer 3 has finished reading AH (first trip: weights are in)
#pragma unroll
    for (int u = 0; u < G4; ++u) {
      const int f = u * SR_T + tid, row = f / c4n, c4 = f % c4n;
      float4 x = make_float4(gv[u].x + gc[u].x, gv[u].y + gc[u].y, gv[u].z + gc[u].z, gv[u].w + gc[u].w);
      const int gm = m0 + row;
      if (gm < a.M)
        a.mask1[(int64_t)gm * c4n + c4] = (uint8_t)((x.x > 0.f ? 1 : 0) | (x.y > 0.f ? 2 : 0) | (x.z > 0.f ? 4 : 0) | (x.w > 0.f ? 8 : 0));
      x.x = x.x > 0.f ? x.x : 0.f, x.y = x.y > 0.f ? x.y : 0.f, x.z = x.z > 0.f ? x.z : 0.f, x.w = x.w > 0.f ? x.w : 0.f;
      *reinterpret_cast<float4*>(AH + row * LDH + 4 * c4) = x;
    }
    __syncthreads();
    if (tile + (int)gridDim.x < tiles) {                   // in flight under this tile's MFMAs
      load_rows((tile + gridDim.x) * SR_BM);              // (their indices arrived during the previous tile)
      if (tile + 2 * (int)gridDim.x < tiles) load_idx((tile + 2 * gridDim.x) * SR_BM);
    }
    // ---- layer 2: wave tile 32 x 32 (columns wn2 ..)
    sc_f32x16 acc2[1];
    sc_mm<1, C1 / 32, LDH, LD2>(acc2, Ap, W2s + (wn2 + r) * LD2 + 4 * h);
    __syncthreads();                                      // every wave has finished reading the generated rows
    {
      const int col = wn2 + r;
      const float bj = b2s[col];
      float* hp = AH + (wm + 4 * h) * LDH + col;           // rows (e & 3) + 8 (e >> 2) of this lane's column
      if (col < C2) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const float v = acc2[0][e] + bj;
          hp[((e & 3) + 8 * (e >> 2)) * LDH] = v > 0.f ? v : 0.f;
        }
      }
    }
    __syncthreads();
    // the layer-2 signs, one word per row and 32 columns: relu(v) > 0 iff v > 0, so they are read back from the tile just
    // written (a packed pass: 8 ds_read_b128 + 32 compares per word) instead of a ballot, an address and a two-lane store
    // per accumulator register in the epilogue above
    for (int t = tid; t < SR_BM * (C2 / 32); t += SR_T) {
      const int row = t & (SR_BM - 1), w = t >> 6;        // consecutive lanes: consecutive rows (conflict-free reads)
      const float* hp = AH + row * LDH + 32 * w;
      uint32_t bits = 0;
#pragma unroll
      for (int c = 0; c < 32; c += 4) {
        const float4 x = *reinterpret_cast<const float4*>(hp + c);
        bits |= (x.x > 0.f ? 1u : 0u) << c | (x.y > 0.f ? 2u : 0u) << c | (x.z > 0.f ? 4u : 0u) << c | (x.w > 0.f ? 8u : 0u) << c;
      }
      if (m0 + row < a.M) a.mask2[(int64_t)(m0 + row) * (C2 / 32) + w] = bits;
    }
    // ---- layer 3 + group max: wave tile 32 x 64 of every 128-column tile
    const bool full = m0 + SR_BM <= a.M;                   // (uniform) every row of the tile exists: no row guards below
    for (int n0 = 0; n0 < a.C3; n0 += 128) {
      sc_f32x16 acc3[2];
      sc_mm<2, C2 / 32, LDH, LD3>(acc3, Ap, W3s + (n0 + wn3 + r) * LD3 + 4 * h);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        float best = -__builtin_inff();
        int bi = wm;
        if (full) {
#pragma unroll
          for (int e = 0; e < 16; ++e)
            if (acc3[j][e] > best) best = acc3[j][e], bi = wm + (e & 3) + 8 * (e >> 2) + 4 * h;
        } else {
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int rl = wm + (e & 3) + 8 * (e >> 2) + 4 * h;
            if (m0 + rl < a.M && acc3[j][e] > best) best = acc3[j][e], bi = rl;
          }
        }
        argmax_xor32(best, bi);
        const int col = n0 + wn3 + j * 32 + r;
        if (tpg == 1) {
          const int64_t grp = (int64_t)((m0 + wm) >> a.ns_shift);
          if (h == 0 && m0 + wm < a.M && col < a.C3) {
            a.out[grp * a.C3 + col] = fmaxf(best + b3s[col], 0.f);
            a.arg[grp * a.C3 + col] = bi - wm;
          }
        } else if (h == 0) {
          pv[(wm >> 5) * 128 + wn3 + j * 32 + r] = best;
          pi[(wm >> 5) * 128 + wn3 + j * 32 + r] = bi;
        }
      }
      if (tpg > 1) {                                      // ns == 64: the two 32-row tiles of this workgroup are ONE group
        __syncthreads();
        for (int cl = tid; cl < 128; cl += SR_T) {
          float best = pv[cl];
          int bi = pi[cl];
          const float v = pv[128 + cl];
          if (v > best) best = v, bi = pi[128 + cl];       // ascending tile = ascending rows: strict >
          const int64_t grp = (int64_t)(m0 >> a.ns_shift);
          const int col = n0 + cl;
          if (m0 < a.M && col < a.C3) {
            a.out[grp * a.C3 + col] = fmaxf(best + b3s[col], 0.f);
            a.arg[grp * a.C3 + col] = bi;
          }
        }
        __syncthreads();
      }
    }
  }
}

// The resident kernel over a table of 16-row UNITS (pc3d_sa_blocks_i32 with unit = 16): at SSG's first level a group
// lists 13.5 points of 32 on average, so the second half of most groups is nothing but padding copies; those units are
// left out and the rest packed four to a 64-row tile, whole groups per tile. A wave's 32-row MFMA block is then two units
// of possibly different groups: the group max is taken per unit (registers 0-7 / 8-15 of an accumulator tile) and the
// units of a group are combined through LDS in ascending row order. Same results as the kernel above.
template <int C1, int C2>
__global__ __launch_bounds__(SR_T, 2) void sa_chain_res_tb_kernel(SaChainArgs a, int tiles_max) {
  extern __shared__ __attribute__((aligned(16))) float sc_lds[];
  constexpr int LDH = (C1 > C2 ? C1 : C2) + 4, LD2 = C1 + 4, LD3 = C2 + 4;
  constexpr int c4n = C1 / 4, G4 = C1 / 16;              // float4 per generated row / per thread (64 rows x c4n = 256 x G4)
  float* W2s = sc_lds;                                   // [64][LD2] (rows past C2 are zero)
  float* W3s = W2s + 64 * LD2;                            // [round128(C3)][LD3] (rows past C3 are zero)
  const int C3p = (a.C3 + 127) & ~127;
  float* AH = W3s + C3p * LD3;                            // [64][LDH]
  float* pv = AH + SR_BM * LDH;                           // [4][128]: a unit's (16 rows) maximum per column of the tile
  int* pi = reinterpret_cast<int*>(pv + 4 * 128);
  float* b2s = reinterpret_cast<float*>(pi + 4 * 128);     // [64] + [C3p]: the biases, read by every tile's epilogues
  float* b3s = b2s + 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int wm = (wave >> 1) * 32;

  for (int f = tid; f < 64 * c4n; f += SR_T) {            // weights + biases: once per workgroup
    const int row = f / c4n, c4 = f % c4n;
    *reinterpret_cast<float4*>(W2s + row * LD2 + 4 * c4) =
        row < C2 ? *reinterpret_cast<const float4*>(a.W2 + (int64_t)row * C1 + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (int f = tid; f < C3p * (C2 / 4); f += SR_T) {
    const int row = f / (C2 / 4), c4 = f % (C2 / 4);
    *reinterpret_cast<float4*>(W3s + row * LD3 + 4 * c4) =
        row < a.C3 ? *reinterpret_cast<const float4*>(a.W3 + (int64_t)row * C2 + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (int c = tid; c < 64; c += SR_T) b2s[c] = c < C2 ? a.b2[c] : 0.f;
  for (int c = tid; c < C3p; c += SR_T) b3s[c] = c < a.C3 ? a.b3[c] : 0.f;

  // The tile's four 16-row UNITS in original row space (unit k = rows 16 k .. 16 k + 15), from the table of
  // pc3d_sa_blocks_i32: units of nothing but padding copies are left out, whole groups per tile. Three-stage prefetch: the
  // units of tile i+3, the INDICES of tile i+2 and the ROWS of tile i+1 are in flight while tile i computes.
  const int tiles = min(tiles_max, a.ntiles[0]);
  auto load_units = [&](int tile, int (&U)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) U[q] = tile < tiles ? a.tb[tile * 4 + q] : -1;
  };
  auto unit_of = [](const int (&U)[4], int q) { return q == 0 ? U[0] : (q == 1 ? U[1] : (q == 2 ? U[2] : U[3])); };
  float4 gv[G4], gc[G4];
  int nidx[G4];
  auto load_idx = [&](const int (&U)[4]) {
#pragma unroll
    for (int u = 0; u < G4; ++u) {
      const int row = (u * SR_T + tid) / c4n;
      const int ub = unit_of(U, row >> 4);
      nidx[u] = ub >= 0 ? a.idx[ub * 16 + (row & 15)] : -1;
    }
  };
  auto load_rows = [&](const int (&U)[4]) {
    const int b_first = U[0] >= 0 ? ((U[0] * 16) >> a.ns_shift) / a.S : 0;     // uniform: one division per tile
#pragma unroll
    for (int u = 0; u < G4; ++u) {
      const int f = u * SR_T + tid, row = f / c4n, c4 = f % c4n;
      const int ub = unit_of(U, row >> 4);
      gv[u] = make_float4(0.f, 0.f, 0.f, 0.f), gc[u] = gv[u];
      if (ub >= 0) {
        const int g = (ub * 16) >> a.ns_shift, p = nidx[u];
        const int bb = b_first + (g >= (b_first + 1) * a.S ? 1 : 0);       // a tile's <= 4 consecutive groups: one boundary (S >= 4)
        gc[u] = *reinterpret_cast<const float4*>(a.Bc + (int64_t)g * C1 + 4 * c4);
        if ((unsigned)p < (unsigned)a.NA)
          gv[u] = *reinterpret_cast<const float4*>(a.P + ((int64_t)bb * a.NA + p) * a.ldp + 4 * c4);
      }
    }
  };
  int tile = blockIdx.x;
  const int step = gridDim.x;
  int Ua[4], Ub[4], Uc[4];
  load_units(tile, Ua);
  load_units(tile + step, Ub);
  load_units(tile + 2 * step, Uc);
  if (tile < tiles) {
    load_idx(Ua);
    load_rows(Ua);
    load_idx(Ub);
  }
  const int wn2 = (wave & 1) * 32, wn3 = (wave & 1) * 64;
  const float* Ap = AH + (wm + r) * LDH + 4 * h;           // this lane's A-operand row (both layers)
  const int upg = a.ns >> 4;                               // units per group
  for (; tile < tiles; tile += step) {
    __syncthreads();                                      // the previous tile's layer 3 has finished reading AH (first trip: weights are in)
#pragma unroll
    for (int u = 0; u < G4; ++u) {
      const int f = u * SR_T + tid, row = f / c4n, c4 = f % c4n;
      float4 x = make_float4(gv[u].x + gc[u].x, gv[u].y + gc[u].y, gv[u].z + gc[u].z, gv[u].w + gc[u].w);
      const int ub = unit_of(Ua, row >> 4);
      if (ub >= 0)
        a.mask1[(int64_t)(ub * 16 + (row & 15)) * c4n + c4] = (uint8_t)((x.x > 0.f ? 1 : 0) | (x.y > 0.f ? 2 : 0) | (x.z > 0.f ? 4 : 0) | (x.w > 0.f ? 8 : 0));
      x.x = x.x > 0.f ? x.x : 0.f, x.y = x.y > 0.f ? x.y : 0.f, x.z = x.z > 0.f ? x.z : 0.f, x.w = x.w > 0.f ? x.w : 0.f;
      *reinterpret_cast<float4*>(AH + row * LDH + 4 * c4) = x;
    }
    __syncthreads();
    int Ud[4];
    load_units(tile + 3 * step, Ud);
    if (tile + step < tiles) {                             // in flight under this tile's MFMAs
      load_rows(Ub);                                      // (their indices arrived during the previous tile)
      if (tile + 2 * step < tiles) load_idx(Uc);
    }
    // ---- layer 2: wave tile 32 x 32 (columns wn2 ..)
    sc_f32x16 acc2[1];
    sc_mm<1, C1 / 32, LDH, LD2>(acc2, Ap, W2s + (wn2 + r) * LD2 + 4 * h);
    __syncthreads();                                      // every wave has finished reading the generated rows
    {
      const int col = wn2 + r;
      const float bj = b2s[col];
      float* hp = AH + (wm + 4 * h) * LDH + col;           // rows (e & 3) + 8 (e >> 2) of this lane's column
      if (col < C2) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const float v = acc2[0][e] + bj;
          hp[((e & 3) + 8 * (e >> 2)) * LDH] = v > 0.f ? v : 0.f;
        }
      }
    }
    __syncthreads();
    // the layer-2 signs, one word per row and 32 columns: relu(v) > 0 iff v > 0, so they are read back from the tile just
    // written (a packed pass: 8 ds_read_b128 + 32 compares per word) instead of a ballot, an address and a two-lane store
    // per accumulator register in the epilogue above
    for (int t = tid; t < SR_BM * (C2 / 32); t += SR_T) {
      const int row = t & (SR_BM - 1), w = t >> 6;        // consecutive lanes: consecutive rows (conflict-free reads)
      const float* hp = AH + row * LDH + 32 * w;
      uint32_t bits = 0;
#pragma unroll
      for (int c = 0; c < 32; c += 4) {
        const float4 x = *reinterpret_cast<const float4*>(hp + c);
        bits |= (x.x > 0.f ? 1u : 0u) << c | (x.y > 0.f ? 2u : 0u) << c | (x.z > 0.f ? 4u : 0u) << c | (x.w > 0.f ? 8u : 0u) << c;
      }
      const int ub = unit_of(Ua, row >> 4);
      if (ub >= 0) a.mask2[(int64_t)(ub * 16 + (row & 15)) * (C2 / 32) + w] = bits;
    }
    // ---- layer 3 + group max: wave tile 32 x 64 of every 128-column tile; the wave's 32 rows are TWO units
    const int us0 = unit_of(Ua, wm >> 4), us1 = unit_of(Ua, (wm >> 4) + 1);
    for (int n0 = 0; n0 < a.C3; n0 += 128) {
      sc_f32x16 acc3[2];
      sc_mm<2, C2 / 32, LDH, LD3>(acc3, Ap, W3s + (n0 + wn3 + r) * LD3 + 4 * h);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {             // rows 16 half .. 16 half + 15 of the wave's block: registers 8 half ..
          float best = -__builtin_inff();
          int bi = 0;
#pragma unroll
          for (int e = 8 * half; e < 8 * half + 8; ++e)
            if (acc3[j][e] > best) best = acc3[j][e], bi = (e & 3) + 8 * ((e >> 2) & 1) + 4 * h;     // row inside the unit
          argmax_xor32(best, bi);
          if (h == 0) {
            const int slot = (wm >> 4) + half;
            pv[slot * 128 + wn3 + j * 32 + r] = (half == 0 ? us0 : us1) >= 0 ? best : -__builtin_inff();   // (an empty slot never wins)
            pi[slot * 128 + wn3 + j * 32 + r] = bi;
          }
        }
      }
      // a group's units sit in consecutive slots of ONE tile, its unit 0 (always kept) first: slot s heads a group when its
      // unit id is a multiple of upg; the group's other kept units follow in ascending row order (strict >: lowest row)
      __syncthreads();
      for (int t = tid; t < 4 * 128; t += SR_T) {
        const int sb = t >> 7, cl = t & 127;
        const int ub = unit_of(Ua, sb);
        if (ub < 0 || (ub & (upg - 1)) != 0) continue;
        float best = pv[sb * 128 + cl];
        int bi = pi[sb * 128 + cl];
        for (int s2 = sb + 1; s2 < 4; ++s2) {
          const int u2 = unit_of(Ua, s2);
          if (u2 < 0 || (u2 >> (a.ns_shift - 4)) != (ub >> (a.ns_shift - 4))) break;
          const float v = pv[s2 * 128 + cl];
          if (v > best) best = v, bi = pi[s2 * 128 + cl] + 16 * (u2 - ub);
        }
        const int64_t grp = ub >> (a.ns_shift - 4);
        const int col = n0 + cl;
        if (col < a.C3) {
          a.out[grp * a.C3 + col] = fmaxf(best + b3s[col], 0.f);
          a.arg[grp * a.C3 + col] = bi;
        }
      }
      __syncthreads();                                    // pv / pi are rewritten by the next column tile / tile
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) Ua[q] = Ub[q], Ub[q] = Uc[q], Uc[q] = Ud[q];
  }
}

// ---------------------------------------------------------------------------------------------------------
// The streaming kernel over a table of 8-ROW units. A 32 x 32 MFMA accumulator tile holds rows 8 q + (e & 3) + 4 h of its
// block in registers e = 4 q .. 4 q + 3: a QUARTER of the block is a register group as well, so the group max can be taken
// per 8 rows at the cost it had per 32. At SSG's second level a group lists 23 of 64 points (the rest are the ball query's
// copies of the first): in 32-row blocks that is 35.1 rows per group (37.6 executed after packing), in 8-row units 26.5
// (31.9) — tools/exp/sa_listed_rows.py; 379 -> 320 us. The table (pc3d_sa_blocks_i32 with unit = 8) has 64-row tiles of
// eight slots, whole groups per tile; a (four-wave) workgroup takes one of them.
// (The RESIDENT kernel was measured with 8-row units as well — 18.0 instead of 21.6 rows per group at SSG's first level —
// and dropped: eight units per tile double its epilogue's shuffles, LDS stores and combine trips, 238 -> 252 us for the
// launch and 2.19 -> 2.31 ms for the iteration. It keeps 16-row units.)
// ---------------------------------------------------------------------------------------------------------
// BM rows per workgroup: 128 (eight waves, two table tiles) or 64 (four waves, one table tile: 80 KB of LDS, so that TWO
// workgroups share a CU and one's barriers and epilogues run under the other's MFMAs)
template <int TN2, int BM>
__global__ __launch_bounds__(4 * BM, BM == 64 ? 2 : 1) void sa_chain_u8_kernel(SaChainArgs a) {
  constexpr int NT = 4 * BM, SLOTS = BM / 8, TT = BM / 64;   // threads, unit slots, table tiles per workgroup
  extern __shared__ __attribute__((aligned(16))) float sc_lds[];
  const int LDH = (a.C1 > a.C2 ? a.C1 : a.C2) + 4;
  float* AH = sc_lds;                                   // [128][LDH]
  float* Ws = sc_lds + BM * LDH;                         // [2][128][SC_LD]
  float* pv = Ws + 2 * 128 * SC_LD;                      // [SLOTS][128] a unit's maximum per column of the column tile
  int* pi = reinterpret_cast<int*>(pv + SLOTS * 128);
  int* s_unit = pi + SLOTS * 128;                        // [16] unit ids (-1: empty slot), [16] their clouds, [16] run lengths
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int wm = (wave >> 1) * 32;                       // this wave's 32 rows
  const int lrow = tid >> 3, lk = (tid & 7) * 4;         // weight staging: 8 threads cover a row's 32 k
  const int nt = a.ntiles[0];
  if ((int)blockIdx.x * TT >= nt) return;                // (uniform, before any barrier)
  if (tid < SLOTS) {
    const int tt = blockIdx.x * TT + (tid >> 3);
    s_unit[tid] = tt < nt ? a.tb[tt * 8 + (tid & 7)] : -1;
  }
  __syncthreads();
  if (tid < SLOTS) {
    const int ub = s_unit[tid];
    s_unit[16 + tid] = ub >= 0 ? ((ub * 8) >> a.ns_shift) / a.S : 0;
    int len = 0;                                         // units of the group this slot heads (0: not a head)
    if (ub >= 0 && (ub & ((a.ns >> 3) - 1)) == 0) {
      len = 1;
      for (int s2 = tid + 1; s2 < SLOTS; ++s2) {
        const int u2 = s_unit[s2];
        if (u2 < 0 || (u2 >> (a.ns_shift - 3)) != (ub >> (a.ns_shift - 3))) break;
        ++len;
      }
    }
    s_unit[32 + tid] = len;
  }
  __syncthreads();
  int us[4];                                             // the wave's four units (slots fill in order: the first decides)
#pragma unroll
  for (int q = 0; q < 4; ++q) us[q] = s_unit[(wm >> 3) + q];
  const bool act = us[0] >= 0;

  // ---- gather + layer 1: AH[row][:] = relu(P[src(row)] + Bc[group(row)]), one sign bit per element to mask1
  {
    const int c4n = a.C1 >> 2;                           // float4 per row
    for (int f0 = tid; f0 < BM * c4n; f0 += 4 * NT) {
      float4 v[4], c[4];
      int gmv[4], rowv[4], c4v[4];
      bool live[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int f = f0 + u * NT;
        live[u] = f < BM * c4n;
        const int row = live[u] ? f >> a.c4_shift : 0, c4 = live[u] ? f & (c4n - 1) : 0;
        const int kb = s_unit[row >> 3];
        if (kb < 0) live[u] = false;                       // an empty slot: nobody will read these rows
        const int gm = kb * 8 + (row & 7);
        gmv[u] = gm, rowv[u] = row, c4v[u] = c4;
        v[u] = make_float4(0.f, 0.f, 0.f, 0.f), c[u] = v[u];
        if (live[u]) {
          const int g = gm >> a.ns_shift, p = a.idx[gm];
          const int bb = s_unit[16 + (row >> 3)];
          c[u] = *reinterpret_cast<const float4*>(a.Bc + (int64_t)g * a.C1 + 4 * c4);
          if ((unsigned)p < (unsigned)a.NA)
            v[u] = *reinterpret_cast<const float4*>(a.P + ((int64_t)bb * a.NA + p) * a.ldp + 4 * c4);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (!live[u]) continue;
        float4 x = make_float4(v[u].x + c[u].x, v[u].y + c[u].y, v[u].z + c[u].z, v[u].w + c[u].w);
        a.mask1[(int64_t)gmv[u] * c4n + c4v[u]] =
            (uint8_t)((x.x > 0.f ? 1 : 0) | (x.y > 0.f ? 2 : 0) | (x.z > 0.f ? 4 : 0) | (x.w > 0.f ? 8 : 0));
        x.x = x.x > 0.f ? x.x : 0.f, x.y = x.y > 0.f ? x.y : 0.f, x.z = x.z > 0.f ? x.z : 0.f, x.w = x.w > 0.f ? x.w : 0.f;
        *reinterpret_cast<float4*>(AH + rowv[u] * LDH + 4 * c4v[u]) = x;
      }
    }
  }

  constexpr int NQ = 1024 / NT, QR = NT / 8;             // float4 per thread of a [128][32] weight slice, rows per pass
  float4 wb[NQ];
  auto fetch_w = [&](const float* W, int ncols, int K, int n0, int k0) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int gn = n0 + q * QR + lrow;
      wb[q] = (gn < ncols) ? *reinterpret_cast<const float4*>(W + (int64_t)gn * K + k0 + lk) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto stash_w = [&](float* dst) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) *reinterpret_cast<float4*>(dst + (q * QR + lrow) * SC_LD + lk) = wb[q];
  };

  // ---- layer 2: H = relu(A . W2^T + b2), 128 x (64 TN2) per workgroup, wave (wm, wn2)
  const int wn2 = (wave & 1) * (32 * TN2);
  sc_f32x16 acc2[TN2];
#pragma unroll
  for (int j = 0; j < TN2; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc2[j][e] = 0.f;
  fetch_w(a.W2, a.C2, a.C1, 0, 0);
  stash_w(Ws);
  __syncthreads();                                       // AH (generated rows) and the first slice are in place
  int cur = 0;
  for (int k0 = 0; k0 < a.C1; k0 += SC_BK) {
    const bool more = k0 + SC_BK < a.C1;
    if (more) fetch_w(a.W2, a.C2, a.C1, 0, k0 + SC_BK);
    if (act) sc_step<TN2>(acc2, AH + (wm + r) * LDH + k0, Ws + cur * 128 * SC_LD, wn2, r, h);
    if (more) stash_w(Ws + (cur ^ 1) * 128 * SC_LD);
    __syncthreads();                                     // (after the last step: every wave has finished reading AH)
    cur ^= 1;
  }
  fetch_w(a.W3, a.C3, a.C2, 0, 0);
#pragma unroll
  for (int j = 0; j < TN2; ++j) {
    if (!act) break;
    const int col = wn2 + j * 32 + r;
    const bool col_ok = col < a.C2;
    const float bj = col_ok ? a.b2[col] : 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = wm + (e & 3) + 8 * (e >> 2) + 4 * h;
      const float v = acc2[j][e] + bj;
      const unsigned long long bal = __builtin_amdgcn_ballot_w64(v > 0.f && col_ok);
      const int ub = us[e >> 2];
      if (r == 0 && ub >= 0 && wn2 + j * 32 < a.C2)        // (C2 % 32 == 0: a 32-column tile is all in or all out)
        a.mask2[(int64_t)(ub * 8 + (e & 3) + 4 * h) * (a.C2 >> 5) + ((wn2 + j * 32) >> 5)] = (uint32_t)(h ? (bal >> 32) : bal);
      if (col_ok) AH[row * LDH + col] = v > 0.f ? v : 0.f;
    }
  }
  stash_w(Ws + cur * 128 * SC_LD);
  __syncthreads();                                       // AH now holds the layer-2 output; W3's first slice is staged

  // ---- layer 3 + group max, one 128-column tile of C3 at a time; wave (wm, wn3 = 64 (wave & 1))
  const int wn3 = (wave & 1) * 64;
  for (int n0 = 0; n0 < a.C3; n0 += 128) {
    sc_f32x16 acc3[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc3[j][e] = 0.f;
    for (int k0 = 0; k0 < a.C2; k0 += SC_BK) {
      const bool more_k = k0 + SC_BK < a.C2, more_n = n0 + 128 < a.C3;
      if (more_k) fetch_w(a.W3, a.C3, a.C2, n0, k0 + SC_BK);
      else if (more_n) fetch_w(a.W3, a.C3, a.C2, n0 + 128, 0);
      if (act) sc_step<2>(acc3, AH + (wm + r) * LDH + k0, Ws + cur * 128 * SC_LD, wn3, r, h);
      if (more_k || more_n) stash_w(Ws + (cur ^ 1) * 128 * SC_LD);
      __syncthreads();
      cur ^= 1;
    }
    // per 8-row unit and column: in-lane over its four accumulator rows (ascending: strict > keeps the lowest row), then
    // across the lane halves (compare (value, row))
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float best = -__builtin_inff();
        int bi = 0;
#pragma unroll
        for (int e = 4 * q; e < 4 * q + 4; ++e)
          if (acc3[j][e] > best) best = acc3[j][e], bi = (e & 3) + 4 * h;
        argmax_xor32(best, bi);
        if (h == 0) {
          const int slot = (wm >> 3) + q;
          pv[slot * 128 + wn3 + j * 32 + r] = us[q] >= 0 ? best : -__builtin_inff();     // (an empty slot never wins)
          pi[slot * 128 + wn3 + j * 32 + r] = bi;
        }
      }
    }
    __syncthreads();
    for (int t = tid; t < SLOTS * 128; t += NT) {
      const int sb = t >> 7, cl = t & 127;
      const int len = s_unit[32 + sb];
      if (len == 0) continue;
      const int ub = s_unit[sb];
      float best = pv[sb * 128 + cl];
      int bi = pi[sb * 128 + cl];
      for (int k = 1; k < len; ++k) {
        const float v = pv[(sb + k) * 128 + cl];
        if (v > best) best = v, bi = pi[(sb + k) * 128 + cl] + 8 * (s_unit[sb + k] - ub);
      }
      const int64_t grp = ub >> (a.ns_shift - 3);
      const int col = n0 + cl;
      if (col < a.C3) {
        a.out[grp * a.C3 + col] = fmaxf(best + a.b3[col], 0.f);
        a.arg[grp * a.C3 + col] = bi;
      }
    }
    __syncthreads();                                     // pv / pi are rewritten by the next column tile
  }
}

// ---------------------------------------------------------------------------------------------------------
// Block table of the streaming kernel: which 32-row blocks of the grouped rows hold at least one LISTED point, packed
// four to a tile without splitting a group over two tiles.
//   sa_block_flags_kernel : a wavefront per group; bit b of flags[g] = block b of the group has a row that is not a copy
//                           of row 0 (block 0 always); the table is filled with -1 on the way.
//   sa_block_pack_kernel  : ONE workgroup packs the groups in order (a group with `a` kept blocks opens a new tile when
//                           the current one has fewer than `a` free slots). The packing is sequential by nature; it is
//                           run as a scan over per-thread chunks: each thread first maps every possible fill level at its
//                           chunk's start to (tiles opened, fill level at its end), thread 0 chains the 256 maps, then
//                           every thread replays its chunk from its true start and writes its slots.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sa_block_flags_kernel(const int32_t* __restrict__ idx, int G, int ns, int unit,
                                                             uint8_t* __restrict__ flags, int32_t* __restrict__ tb, int tb_len) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < tb_len; i += gridDim.x * 256) tb[i] = -1;
  const int lane = threadIdx.x & 63;
  const int g = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (g >= G) return;
  const int32_t* id = idx + (int64_t)g * ns;
  const int i0 = id[0];
  unsigned fl = 0;
  for (int j0 = 0; j0 < ns; j0 += 64) {
    const int j = j0 + lane;
    const bool own = j < ns && (j == 0 || id[j] != i0);
    const unsigned long long bal = __builtin_amdgcn_ballot_w64(own);
    for (int q = 0; q < 64 / unit; ++q) {                  // unit = 8, 16 or 32 rows
      const unsigned long long m = unit == 32 ? 0xffffffffull : (unit == 16 ? 0xffffull : 0xffull);
      if ((bal >> (q * unit)) & m) fl |= 1u << (j0 / unit + q);
    }
  }
  if (lane == 0) flags[g] = (uint8_t)fl;
}

constexpr int SP_T = 1024;

// a chunk of groups as a map: fill level at its start (0 .. 4) -> (tiles opened << 3 | fill level at its end)
struct SpMap {
  int m[5];
};
__device__ __forceinline__ int sp_pick(const SpMap& a, int f) {
  return f == 0 ? a.m[0] : (f == 1 ? a.m[1] : (f == 2 ? a.m[2] : (f == 3 ? a.m[3] : a.m[4])));
}
__device__ __forceinline__ SpMap sp_then(const SpMap& a, const SpMap& b) {     // first a, then b
  SpMap c;
#pragma unroll
  for (int f = 0; f < 5; ++f) {
    const int x = a.m[f], y = sp_pick(b, x & 7);
    c.m[f] = (((x >> 3) + (y >> 3)) << 3) | (y & 7);
  }
  return c;
}

__global__ __launch_bounds__(SP_T) void sa_block_pack_kernel(const uint8_t* __restrict__ flags, int G, int bpg, int32_t* __restrict__ tb,
                                                             int32_t* __restrict__ ntiles) {
  extern __shared__ uint8_t s_fl[];                       // [G] the groups' unit flags (staged once: the loops below are serial)
  __shared__ int s_wave[SP_T / 64][5];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  for (int g = t; g < G; g += SP_T) s_fl[g] = flags[g];
  __syncthreads();
  const int per = (G + SP_T - 1) / SP_T, lo = min(t * per, G), hi = min(lo + per, G);
  SpMap mine;                                             // the five possible fill levels at the chunk's start, side by side
#pragma unroll
  for (int f = 0; f < 5; ++f) mine.m[f] = f;
  for (int g = lo; g < hi; ++g) {
    const int a = __builtin_popcount((unsigned)s_fl[g]);
#pragma unroll
    for (int f = 0; f < 5; ++f) {
      const int fill = mine.m[f] & 7, til = mine.m[f] >> 3;
      const bool open = fill + a > 4;
      mine.m[f] = ((til + (open ? 1 : 0)) << 3) | ((open ? 0 : fill) + a);
    }
  }
  // inclusive scan of the maps over the lanes of a wave, then over the waves
  SpMap inc = mine;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    SpMap o;
#pragma unroll
    for (int f = 0; f < 5; ++f) o.m[f] = __shfl_up(inc.m[f], d, 64);
    if (lane >= d) inc = sp_then(o, inc);
  }
  if (lane == 63)
#pragma unroll
    for (int f = 0; f < 5; ++f) s_wave[wave][f] = inc.m[f];
  __syncthreads();
  int tile = 0, fill = 0;                                  // state at this wave's start
  for (int w = 0; w < wave; ++w) {
    const int y = s_wave[w][fill];
    tile += y >> 3, fill = y & 7;
  }
  {                                                        // ... at this thread's start: the lanes before it
    SpMap ex;
#pragma unroll
    for (int f = 0; f < 5; ++f) ex.m[f] = __shfl_up(inc.m[f], 1, 64);
    if (lane > 0) {
      const int y = sp_pick(ex, fill);
      tile += y >> 3, fill = y & 7;
    }
  }
  for (int g = lo; g < hi; ++g) {
    const unsigned fb = s_fl[g];
    const int a = __builtin_popcount(fb);
    if (fill + a > 4) ++tile, fill = 0;
    for (int b = 0; b < bpg; ++b)
      if ((fb >> b) & 1u) tb[tile * 4 + fill++] = g * bpg + b;
  }
  if (t == SP_T - 1) ntiles[0] = tile + (fill > 0 ? 1 : 0);
}


// The table of 8-row units: eight slots per tile, whole groups per tile. Here the packing is NOT chained across the whole
// launch: every chunk of >= 32 groups opens a tile of its own, so chunks are packed independently and the tiles are
// numbered by one prefix sum over the chunks' tile counts — which EVERY workgroup computes for itself (a thread per
// chunk: 32 serial steps over flags that sit in L2), so one launch of many workgroups does it, a wavefront per chunk
// writing its piece of the table through LDS. (One workgroup doing all the writes took 44 us at SSG's first level — 131 072
// uncoalesced stores from one CU — and the five-state maps above cost more with eight slots.) Half a tile per chunk is
// lost: 4 % of the rows.
constexpr int SU_CAP = 8;                                  // slots per tile at most
__global__ __launch_bounds__(SP_T) void sa_unit_pack_kernel(const uint8_t* __restrict__ flags, int G, int upg, int per, int cap,
                                                            int32_t* __restrict__ tb, int32_t* __restrict__ ntiles) {
  __shared__ int s_base[SP_T + 1];                         // tiles before chunk c
  __shared__ int s_wave[SP_T / 64];
  __shared__ int s_piece[SP_T / 64][64 * SU_CAP];          // a chunk's tiles (<= one per group, per <= 64)
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int nchunks = (G + per - 1) / per;                 // <= 1024
  int tiles = 0;
  if (t < nchunks) {
    const int lo = t * per, hi = min(lo + per, G);
    int fill = cap;
    if (per == 32 && hi - lo == 32 && (reinterpret_cast<uintptr_t>(flags) & 15) == 0) {      // the chunk's 32 flag bytes in two loads
      const uint4 v0 = *reinterpret_cast<const uint4*>(flags + lo), v1 = *reinterpret_cast<const uint4*>(flags + lo + 16);
      const uint32_t wd[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
      for (int i = 0; i < 32; ++i) {
        const int a = __builtin_popcount((wd[i >> 2] >> (8 * (i & 3))) & 0xffu);
        if (fill + a > cap) ++tiles, fill = 0;
        fill += a;
      }
    } else {
      for (int g = lo; g < hi; ++g) {
        const int a = __builtin_popcount((unsigned)flags[g]);
        if (fill + a > cap) ++tiles, fill = 0;
        fill += a;
      }
    }
  }
  int incl = tiles;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int up = __shfl_up(incl, d, 64);
    if (lane >= d) incl += up;
  }
  if (lane == 63) s_wave[wave] = incl;
  __syncthreads();
  int base = incl - tiles;
  for (int w = 0; w < wave; ++w) base += s_wave[w];
  s_base[t] = base;
  if (t == SP_T - 1) s_base[SP_T] = base + tiles;
  __syncthreads();
  if (blockIdx.x == 0 && t == 0) ntiles[0] = s_base[SP_T];
  // a wavefront per chunk: lane l holds group lo + l; the greedy walk over the chunk runs on the scalar unit
  for (int c = blockIdx.x * (SP_T / 64) + wave; c < nchunks; c += gridDim.x * (SP_T / 64)) {
    const int lo = c * per, n = min(per, G - lo);
    const unsigned fb = lane < n ? flags[lo + lane] : 0u;
    const int a = __builtin_popcount(fb);
    int tile = -1, fill = cap, my_tile = 0, my_fill = 0;
    for (int i = 0; i < n; ++i) {
      const int ai = __builtin_amdgcn_readlane(a, i);
      if (fill + ai > cap) ++tile, fill = 0;
      if (lane == i) my_tile = tile, my_fill = fill;
      fill += ai;
    }
    const int nt = tile + 1;                               // = s_base[c + 1] - s_base[c]
    int* piece = s_piece[wave];
    for (int i = lane; i < nt * cap; i += 64) piece[i] = -1;
    if (lane < n) {
      int k = my_tile * cap + my_fill;
      for (int b = 0; b < upg; ++b)
        if ((fb >> b) & 1u) piece[k++] = (lo + lane) * upg + b;
    }
    int32_t* dst = tb + (int64_t)s_base[c] * cap;
    for (int i = lane; i < nt * cap; i += 64) dst[i] = piece[i];
  }
}

}  // namespace pc3d

using namespace pc3d;

extern "C" int pc3d_sa_blocks_i32(const int32_t* idx, int B, int S, int ns, int unit, uint8_t* flags, int32_t* tb, int32_t* ntiles,
                                  void* stream) {
  const char* nm = "pc3d_sa_blocks_i32";
  PC3D_REQUIRE(B >= 0 && S >= 1 && (ns == 32 || ns == 64 || ns == 128) && (int64_t)B * S * ns <= 0x7fffffffLL,
               "%s: bad sizes B=%d S=%d ns=%d (ns in {32,64,128})", nm, B, S, ns);
  PC3D_REQUIRE(((unit == 16 || unit == 32) && ns / unit >= 1 && ns / unit <= 4) || (unit == 8 && ns <= 64),
               "%s: unit=%d rows: 16 or 32 with at most four units per group, or 8 with at most eight (ns=%d)", nm, unit, ns);
  if (B == 0) return PC3D_OK;
  PC3D_REQUIRE(idx && flags && tb && ntiles, "%s: null pointer", nm);
  const int G = B * S, M = G * ns;
  // unit 8: 64-row tiles of eight slots; unit 16 / 32: four slots per tile. Units 8 and 16 (at most one tile per group) are
  // packed chunk by chunk by many workgroups, which write whole tiles: nothing to pre-fill, tiles past ntiles are never read
  const bool chunked = unit != 32;
  const int tb_len = chunked ? 0 : cdiv(M, 4 * unit) * 4;
  hipStream_t st = as_stream(stream);
  // (checked before any launch by ops.sa_blocks, which then builds no table: the chain launch takes every row)
  PC3D_REQUIRE(G <= (chunked ? 64 : 48) * 1024, "%s: B * S = %d groups exceed the packing kernel's limit (%d K)", nm, G, chunked ? 64 : 48);
  hipLaunchKernelGGL(sa_block_flags_kernel, dim3(cdiv(G, 4)), dim3(256), 0, st, idx, G, ns, unit, flags, tb, tb_len);
  if (chunked) {
    const int per = cdiv(G, SP_T) > 32 ? cdiv(G, SP_T) : 32;                  // groups per chunk (<= 64: a lane per group)
    const int nchunks = cdiv(G, per);
    hipLaunchKernelGGL(sa_unit_pack_kernel, dim3(cdiv(nchunks, SP_T / 64)), dim3(SP_T), 0, st, flags, G, ns / unit, per,
                       unit == 8 ? 8 : 4, tb, ntiles);
  } else {
    hipLaunchKernelGGL(sa_block_pack_kernel, dim3(1), dim3(SP_T), (size_t)G, st, flags, G, ns / unit, tb, ntiles);
  }
  PC3D_LAUNCH_CHECK(nm);
  return PC3D_OK;
}

static int sa_chain_launch(const char* nm, const float* P, int64_t ldp, const float* Bc, const int32_t* idx, int B, int NA, int S, int ns,
                           const float* W2, const float* b2, int C1, int C2, const float* W3, const float* b3, int C3,
                           uint8_t* mask1, uint32_t* mask2, float* out, int64_t* arg, const int32_t* tb, const int32_t* ntiles,
                           int unit, void* stream) {
  PC3D_REQUIRE(B >= 0 && NA >= 1 && S >= 1 && (ns == 32 || ns == 64 || ns == 128), "%s: bad sizes B=%d NA=%d S=%d ns=%d (ns in {32,64,128})",
               nm, B, NA, S, ns);
  PC3D_REQUIRE((C1 == 32 || C1 == 64 || C1 == 128) && C2 >= 32 && C2 <= 128 && C2 % 32 == 0 && C3 >= 32 && C3 % 32 == 0,
               "%s: widths C1=%d C2=%d C3=%d (C1 in {32,64,128}, C2 a multiple of 32 up to 128, C3 a multiple of 32)", nm, C1, C2, C3);
  PC3D_REQUIRE((int64_t)B * S * ns <= 0x7fffffffLL && (int64_t)B * NA <= 0x7fffffffLL && ldp >= C1 && ldp % 4 == 0,
               "%s: problem too large or row stride of P not a multiple of 4", nm);
  if (B == 0) return PC3D_OK;
  PC3D_REQUIRE(P && Bc && idx && W2 && b2 && W3 && b3 && mask1 && mask2 && out && arg, "%s: null pointer", nm);
  PC3D_REQUIRE(((reinterpret_cast<uintptr_t>(P) | reinterpret_cast<uintptr_t>(Bc) | reinterpret_cast<uintptr_t>(W2) |
                 reinterpret_cast<uintptr_t>(W3)) & 15) == 0, "%s: operands must be 16-byte aligned", nm);
  SaChainArgs a{P, Bc, idx, W2, b2, W3, b3, ldp, B * S * ns, NA, S, ns, C1, C2, C3, ns == 32 ? 5 : (ns == 64 ? 6 : 7),
                C1 == 32 ? 3 : (C1 == 64 ? 4 : 5), mask1, mask2, out, arg, tb, ntiles};
  const int ldh = (C1 > C2 ? C1 : C2) + 4;
  hipStream_t st = as_stream(stream);
  if ((C1 == 32 || C1 == 64) && (C2 == 32 || C2 == 64) && ns <= 64) {
    // small weights: the persistent resident-weight kernel, when its LDS (weights padded to 64 / 128 rows, a 64-row
    // tile) leaves room for two workgroups per CU
    const int c3p = (C3 + 127) & ~127;
    const size_t lds_r = ((size_t)64 * (C1 + 4) + (size_t)c3p * (C2 + 4) + (size_t)SR_BM * ldh + (tb ? 4 : 2) * 2 * 128 + 64 + c3p) * sizeof(float);
    if (lds_r <= 80 * 1024) {
      PC3D_REQUIRE(!tb || S >= 4, "%s: the unit table needs at least four groups per cloud (S=%d)", nm, S);
      PC3D_REQUIRE(!tb || unit == 16, "%s: this shape runs on the resident kernel, whose table has 16-row units (unit=%d)", nm, unit);
      const int tiles = tb ? B * S : cdiv(a.M, SR_BM);     // (the table's tiles: at most one per group)
      const int grid_r = tiles < 512 ? tiles : 512;
      auto launch = [&](auto kern) -> int {
        if (lds_r > 64 * 1024)
          if (hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_r);
              e != hipSuccess) { set_error("%s: LDS opt-in failed: %s", nm, hipGetErrorString(e)); return (int)e; }
        hipLaunchKernelGGL(kern, dim3(grid_r), dim3(SR_T), lds_r, st, a, tiles);
        return PC3D_OK;
      };
      int rc;
      if (tb)
        rc = C1 == 32 ? (C2 == 32 ? launch(sa_chain_res_tb_kernel<32, 32>) : launch(sa_chain_res_tb_kernel<32, 64>))
                      : (C2 == 32 ? launch(sa_chain_res_tb_kernel<64, 32>) : launch(sa_chain_res_tb_kernel<64, 64>));
      else
        rc = C1 == 32 ? (C2 == 32 ? launch(sa_chain_res_kernel<32, 32>) : launch(sa_chain_res_kernel<32, 64>))
                      : (C2 == 32 ? launch(sa_chain_res_kernel<64, 32>) : launch(sa_chain_res_kernel<64, 64>));
      if (rc) return rc;
      PC3D_LAUNCH_CHECK(nm);
      return PC3D_OK;
    }
  }
  PC3D_REQUIRE(!tb || unit == 32 || (unit == 8 && ns <= 64),
               "%s: this shape runs on the streaming kernel, whose table has 32-row blocks or (ns <= 64) 8-row units (unit=%d)", nm, unit);
  if (tb && unit == 8) {
    // 64 rows per workgroup: 79 KB of LDS at the widest shape (128 -> 128), two workgroups per CU (128-row workgroups, one
    // per CU, were 319 instead of 311 us at SSG's second level)
    const size_t lds8 = ((size_t)64 * ldh + 2 * 128 * SC_LD + 2 * 8 * 128 + 48) * sizeof(float);
    const dim3 grid8(B * S), block8(256);                  // (table tiles: <= one per group)
    auto launch8 = [&](auto kern) -> int {
      if (lds8 > 64 * 1024)
        if (hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds8);
            e != hipSuccess) { set_error("%s: LDS opt-in failed: %s", nm, hipGetErrorString(e)); return (int)e; }
      hipLaunchKernelGGL(kern, grid8, block8, lds8, st, a);
      return PC3D_OK;
    };
    const int rc = C2 <= 64 ? launch8(sa_chain_u8_kernel<1, 64>) : launch8(sa_chain_u8_kernel<2, 64>);
    if (rc) return rc;
    PC3D_LAUNCH_CHECK(nm);
    return PC3D_OK;
  }
  const size_t lds = ((size_t)SC_BM * ldh + 2 * 128 * SC_LD + 2 * 4 * 128 + 8) * sizeof(float);
  const dim3 grid(cdiv(a.M, SC_BM)), block(SC_T);
  if (C2 <= 64) {
    if (lds > 64 * 1024)
      if (hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(sa_chain_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
          e != hipSuccess) { set_error("%s: LDS opt-in failed: %s", nm, hipGetErrorString(e)); return (int)e; }
    hipLaunchKernelGGL(sa_chain_kernel<1>, grid, block, lds, st, a);
  } else {
    if (lds > 64 * 1024)
      if (hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(sa_chain_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
          e != hipSuccess) { set_error("%s: LDS opt-in failed: %s", nm, hipGetErrorString(e)); return (int)e; }
    hipLaunchKernelGGL(sa_chain_kernel<2>, grid, block, lds, st, a);
  }
  PC3D_LAUNCH_CHECK(nm);
  return PC3D_OK;
}

// rows per unit of the table pc3d_sa_chain_tb_f32 expects for a shape (the dispatch rule of sa_chain_launch): 16 = the
// resident kernel, 8 = the streaming kernel with groups of <= 64 rows, 32 = with groups of 128 rows, 0 = no table
extern "C" int pc3d_sa_chain_table_unit(int S, int ns, int C1, int C2, int C3) {
  if (!(ns == 32 || ns == 64 || ns == 128) || !(C1 == 32 || C1 == 64 || C1 == 128) || C2 < 32 || C2 > 128 || C2 % 32 || C3 < 32 || C3 % 32)
    return 0;
  if ((C1 == 32 || C1 == 64) && (C2 == 32 || C2 == 64) && ns <= 64) {          // the resident kernel, if its LDS fits
    const int c3p = (C3 + 127) & ~127, ldh = (C1 > C2 ? C1 : C2) + 4;
    const size_t fixed = (size_t)64 * (C1 + 4) + (size_t)c3p * (C2 + 4) + (size_t)SR_BM * ldh + 64 + c3p;
    if (S >= 4 && (fixed + 4 * 2 * 128) * sizeof(float) <= 80 * 1024) return 16;
    if ((fixed + 2 * 2 * 128) * sizeof(float) <= 80 * 1024) return 0;          // resident without a table
  }
  return ns <= 64 ? 8 : 32;                                                    // the streaming kernel
}

extern "C" int pc3d_sa_chain_f32(const float* P, int64_t ldp, const float* Bc, const int32_t* idx, int B, int NA, int S, int ns,
                                 const float* W2, const float* b2, int C1, int C2, const float* W3, const float* b3, int C3,
                                 uint8_t* mask1, uint32_t* mask2, float* out, int64_t* arg, void* stream) {
  return sa_chain_launch("pc3d_sa_chain_f32", P, ldp, Bc, idx, B, NA, S, ns, W2, b2, C1, C2, W3, b3, C3, mask1, mask2, out, arg,
                         nullptr, nullptr, 0, stream);
}

extern "C" int pc3d_sa_chain_tb_f32(const float* P, int64_t ldp, const float* Bc, const int32_t* idx, int B, int NA, int S, int ns,
                                    const float* W2, const float* b2, int C1, int C2, const float* W3, const float* b3, int C3,
                                    uint8_t* mask1, uint32_t* mask2, float* out, int64_t* arg, const int32_t* tb,
                                    const int32_t* ntiles, int unit, void* stream) {
  PC3D_REQUIRE((tb == nullptr) == (ntiles == nullptr), "pc3d_sa_chain_tb_f32: tb and ntiles go together");
  return sa_chain_launch("pc3d_sa_chain_tb_f32", P, ldp, Bc, idx, B, NA, S, ns, W2, b2, C1, C2, W3, b3, C3, mask1, mask2, out, arg,
                         tb, ntiles, unit, stream);
}
