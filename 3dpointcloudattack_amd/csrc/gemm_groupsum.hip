// gh1 = gz . W2 (the backward GEMM of a set-abstraction chain's second layer, pc3d_gemm_nt_f32 on W2^T) and the GROUPS
// pass of the first layer's backward (group_act_bwd_groups_kernel, group.hip) in ONE launch:
//
//   gh1[row, c1]  = sum_c2 gz[row, c2] * W2[c2, c1]
//   gBc [g, c1]   = sum_{j}                           mask1[row(g, j), c1] * gh1[row(g, j), c1]
//   tail[g, c1]   = sum_{j > 0, idx[g,j] == idx[g,0]}  mask1[...] * gh1[...]       (the ball query's padding, see group.hip)
//
// (model/pointnet2_utils.py:190-197 differentiated: the second Conv2d 1x1, then the ReLU of the first and the sum over
// nsample that its per-centre bias takes.) As two launches the [B*S*ns, C1] tensor gh1 (268 MB at each of SSG's levels,
// B=64, N=2048) was written by the GEMM and read back by the groups pass — 114 / 104 us of pure traffic; here the sums are
// taken from the tile while it is on its way out.
//
// Workgroup = 128 rows (128 / ns whole groups), four wavefronts, wave w owns rows 32 w .. 32 w + 31 and all C1 columns.
// The A operand arrives in 64-column chunks (global -> registers -> LDS, the next chunk under this chunk's MFMAs), W2^T in
// K slices of 32 as in gemm_nt_kernel (same k order, same MFMA sequence as the separate launch); the accumulator tiles go
// out through LDS 64 columns at a time: coalesced stores of gh1, and a thread per (group, column) sums the group's rows in
// ascending order — the groups pass's order. Results are bit-identical to the two launches (tests/test_sa_chain_gpu.py).
//
// Measured and dropped on the way here: the max-backward's sparse row generation in the SAME launch (the 268 MB of gz
// would not be written either). With the group's rows in registers addressed through M0 (group_max_linear_bwd_kernel's
// form) the launch took 317 / 560 us at SSG's levels against 370 / 443 for the three launches; as ordered ds_add_f32 row
// adds into the LDS tile 1400 us (an LDS float atomic retires ~one lane per clock); as a dense one-hot MFMA product it is
// twice the flops of the GEMM itself. The generation stays a launch of its own.
#include "pc3d_common.h"

namespace pc3d {

using gs_f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int GS_T = 256, GS_BM = 128, GS_KC = 64, GS_LDH = GS_KC + 4, GS_BK = 32, GS_LDW = GS_BK + 4;

struct GroupSumArgs {
  const float* X;         // [M, K] row stride ldx  (gz)
  int64_t ldx;
  const float* Wt;        // [C1, K]  (W2 transposed: the NT operand)
  const uint8_t* mask1;   // [M, C1/4]
  const int32_t* idx;     // [M]
  const uint32_t* amask;  // [G, NS/32] or null: bit j of group g = "row (g, j) of X is not all zero (and was written)". Only
                          // those rows are read, multiplied and written; the sums skip the others (exact zeros)
  int M, G, K;
  float* Y;               // [M, C1]
  float* gBc;             // [G, C1]
  float* tail;            // [G, C1]
};

// one K slice (32 k) of a 32 x (32 TN) wave tile: A rows from the chunk (row stride GS_LDH), B from the staged slice
template <int TN>
__device__ __forceinline__ void gs_step(gs_f32x16 (&acc)[TN], const float* __restrict__ Ap, const float* __restrict__ Bp) {
  // Ap = AH + row * GS_LDH + k0 + 4 h;  Bp = Ws + r * GS_LDW + 4 h  (+ j * 32 * GS_LDW per column tile)
  float4 av[2], bv[2][TN];
  av[0] = *reinterpret_cast<const float4*>(Ap);
#pragma unroll
  for (int j = 0; j < TN; ++j) bv[0][j] = *reinterpret_cast<const float4*>(Bp + j * 32 * GS_LDW);
#pragma unroll
  for (int t = 0; t < GS_BK / 8; ++t) {
    const int c = t & 1, n = c ^ 1;
    if (t + 1 < GS_BK / 8) {
      av[n] = *reinterpret_cast<const float4*>(Ap + 8 * (t + 1));
#pragma unroll
      for (int j = 0; j < TN; ++j) bv[n][j] = *reinterpret_cast<const float4*>(Bp + j * 32 * GS_LDW + 8 * (t + 1));
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

// NS = rows per group (32 / 64 / 128), TN = C1 / 32 column tiles per wave
template <int NS, int TN>
__global__ __launch_bounds__(GS_T, (TN > 2 ? 2 : 3)) void gemm_groupsum_kernel(GroupSumArgs a) {
  constexpr int GPT = GS_BM / NS;                          // groups per tile
  constexpr int C1 = 32 * TN;
  // weight-slice buffers: two for the widest layer; ONE for C1 <= 64, whose 48 KB of LDS then let three workgroups share a
  // CU (the kernel waits on memory latency between its short phases: residency pays more than the saved barrier)
  constexpr int NBUF = TN > 2 ? 2 : 1;
  extern __shared__ __attribute__((aligned(16))) float gs_lds[];
  float* AH = gs_lds;                                      // [128][GS_LDH]: a chunk of X, later a 64-column slab of Y
  float* Ws = AH + GS_BM * GS_LDH;                          // [NBUF][C1][GS_LDW]
  uint8_t* m1s = reinterpret_cast<uint8_t*>(Ws + NBUF * C1 * GS_LDW);    // [128][C1 / 4] layer-1 sign bits of the tile
  uint32_t* s_rep = reinterpret_cast<uint32_t*>(m1s + GS_BM * (C1 / 4));  // [GPT][4] bit j: row j repeats the group's first index
  int* rowmap = reinterpret_cast<int*>(s_rep + GPT * 4);                   // [128] compact row -> row of the tile
  int* zlist = rowmap + GS_BM;                                             // [128] rows to be written as zeros (see below)
  __shared__ int s_nzero;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int m0 = blockIdx.x * GS_BM;
  const int g_first = m0 / NS;

  // ---- the tile's ACTIVE rows, compacted in ascending order: word w of the tile covers its rows 32 w .. 32 w + 31
  uint32_t aw[4];
  int abase[5];
  abase[0] = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const int g = g_first + (32 * w) / NS;
    uint32_t word = 0u;
    if (g < a.G) word = a.amask ? a.amask[(int64_t)g * (NS / 32) + ((32 * w) % NS) / 32] : 0xffffffffu;
    aw[w] = word;
    abase[w + 1] = abase[w] + __builtin_popcount(word);
  }
  const int n_act = abase[4];
  if (tid < GS_BM) {
    const int w = tid >> 5, bit = tid & 31;
    const uint32_t word = w == 0 ? aw[0] : (w == 1 ? aw[1] : (w == 2 ? aw[2] : aw[3]));
    const int bw = w == 0 ? abase[0] : (w == 1 ? abase[1] : (w == 2 ? abase[2] : abase[3]));
    if ((word >> bit) & 1u) rowmap[bw + __builtin_popcount(word & ((1u << bit) - 1u))] = tid;
  }
  __syncthreads();

  // the A chunk kc of the tile's active rows: <= 128 rows x 16 float4, eight per thread (the rest reads as zero)
  float4 xa[8];
  auto fetch_x = [&](int kc) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int f = u * GS_T + tid, crow = f >> 4, c4 = f & 15;
      const int k = kc + 4 * c4;
      xa[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (crow < n_act && k < a.K) xa[u] = *reinterpret_cast<const float4*>(a.X + (int64_t)(m0 + rowmap[crow]) * a.ldx + k);
    }
  };
  auto stash_x = [&]() {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int f = u * GS_T + tid;
      *reinterpret_cast<float4*>(AH + (f >> 4) * GS_LDH + 4 * (f & 15)) = xa[u];
    }
  };
  const int lrow = tid >> 3, lk = (tid & 7) * 4;            // weight staging: 8 threads cover a row's 32 k
  float4 wb[TN];
  auto fetch_w = [&](int k0) {
#pragma unroll
    for (int q = 0; q < TN; ++q)
      wb[q] = (k0 + lk < a.K) ? *reinterpret_cast<const float4*>(a.Wt + (int64_t)(q * 32 + lrow) * a.K + k0 + lk)
                              : make_float4(0.f, 0.f, 0.f, 0.f);
  };
  auto stash_w = [&](float* dst) {
#pragma unroll
    for (int q = 0; q < TN; ++q) *reinterpret_cast<float4*>(dst + (q * 32 + lrow) * GS_LDW + lk) = wb[q];
  };
  fetch_x(0);
  fetch_w(0);

  // ---- tile constants into LDS: the layer-1 sign bytes, the padding flags
  {
    constexpr int BYTES = GS_BM * (C1 / 4);                // 1 / 2 / 4 KB: 16-byte pieces
    const int64_t total = (int64_t)a.M * (C1 / 4);
    for (int e = tid; e < BYTES / 16; e += GS_T) {
      const int64_t byte0 = (int64_t)m0 * (C1 / 4) + 16 * e;
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (byte0 + 16 <= total) v = *reinterpret_cast<const uint4*>(a.mask1 + byte0);
      else
        for (int q = 0; q < 16; ++q)
          if (byte0 + q < total) reinterpret_cast<uint8_t*>(&v)[q] = a.mask1[byte0 + q];
      *reinterpret_cast<uint4*>(m1s + 16 * e) = v;
    }
  }
  if (wave < GPT) {                                        // one wavefront per group: 64 rows per ballot
    const int g = g_first + wave;
#pragma unroll
    for (int j0 = 0; j0 < NS; j0 += 64) {
      const int j = j0 + lane;
      bool rp = false;
      if (g < a.G && j > 0 && j < NS) rp = a.idx[(int64_t)g * NS + j] == a.idx[(int64_t)g * NS];
      const unsigned long long bal = __builtin_amdgcn_ballot_w64(rp);
      if (lane == 0) s_rep[wave * 4 + (j0 >> 5)] = (uint32_t)bal, s_rep[wave * 4 + (j0 >> 5) + 1] = (uint32_t)(bal >> 32);
    }
  }

  gs_f32x16 acc[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
  if (tid == 0) s_nzero = 0;

  // ---- Y tile = X tile . Wt^T, K in chunks of 64 (A) and slices of 32 (B)
  int cur = 0;
  for (int kc = 0; kc < a.K; kc += GS_KC) {
    if (kc > 0) __syncthreads();                            // the previous chunk's last slice has read AH
    stash_x();
    stash_w(Ws + cur * C1 * GS_LDW);
    __syncthreads();
    if (kc + GS_KC < a.K) fetch_x(kc + GS_KC);              // in flight under this chunk's MFMAs
    const int nks = (a.K - kc < GS_KC ? a.K - kc : GS_KC) / GS_BK;
    for (int ks = 0; ks < nks; ++ks) {
      const int knext = kc + (ks + 1) * GS_BK;
      const bool more = knext < a.K;
      if (more) fetch_w(knext);
      if (wave * 32 < n_act)                                // (uniform) a wave whose 32 compact rows are all past the active ones idles
        gs_step<TN>(acc, AH + (wave * 32 + r) * GS_LDH + ks * GS_BK + 4 * h, Ws + cur * C1 * GS_LDW + r * GS_LDW + 4 * h);
      if (NBUF == 2) {
        if (ks + 1 < nks) {                                 // the chunk's next slice into the other buffer
          stash_w(Ws + (cur ^ 1) * C1 * GS_LDW);
          __syncthreads();
        }
        cur ^= 1;                                           // (after the last slice: the next chunk's first goes to the other buffer)
      } else if (ks + 1 < nks) {
        __syncthreads();                                    // every wave has read the slice in the only buffer
        stash_w(Ws);
        __syncthreads();
      }
    }
  }
  __syncthreads();                                         // every wave has read the last chunk
  if (a.amask && tid < GS_BM && m0 + tid < a.M) {           // (order in the list does not matter: rows of zeros)
    const int t = tid, gl = t / NS, j = t - gl * NS;
    const uint32_t word = (t >> 5) == 0 ? aw[0] : ((t >> 5) == 1 ? aw[1] : ((t >> 5) == 2 ? aw[2] : aw[3]));
    if (!((word >> (t & 31)) & 1u) && !((s_rep[gl * 4 + (j >> 5)] >> (j & 31)) & 1u)) zlist[atomicAdd(&s_nzero, 1)] = t;
  }
  __syncthreads();
  const int nzero = s_nzero;

  // ---- Y out through LDS, 64 columns at a time; group sums in ascending row order
  for (int hh = 0; hh < C1; hh += 64) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      if (j * 32 < hh || j * 32 >= hh + 64) continue;
      float* hp = AH + (wave * 32 + 4 * h) * GS_LDH + (j * 32 - hh) + r;
#pragma unroll
      for (int e = 0; e < 16; ++e) hp[((e & 3) + 8 * (e >> 2)) * GS_LDH] = acc[j][e];
    }
    __syncthreads();
    const int wcols = C1 - hh < 64 ? C1 - hh : 64;          // 32 or 64 columns in this slab
    for (int f = tid; f < GS_BM * 16; f += GS_T) {
      const int crow = f >> 4, c4 = f & 15;
      if (4 * c4 < wcols && crow < n_act)
        *reinterpret_cast<float4*>(a.Y + (int64_t)(m0 + rowmap[crow]) * C1 + hh + 4 * c4) =
            *reinterpret_cast<const float4*>(AH + crow * GS_LDH + 4 * c4);
    }
    // an inactive row that is NOT one of the group's padding copies is still in some point's reverse list: it reads as zero
    // there (few rows: the copies, which the points pass reaches through `tail` only, are the bulk of the inactive)
    for (int f = tid; f < nzero * 16; f += GS_T) {
      const int t = zlist[f >> 4], c4 = f & 15;
      if (4 * c4 < wcols) *reinterpret_cast<float4*>(a.Y + (int64_t)(m0 + t) * C1 + hh + 4 * c4) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int t = tid; t < GPT * 64; t += GS_T) {
      const int gl = t >> 6, cl = t & 63;
      const int g = g_first + gl, col = hh + cl;
      if (cl < wcols && g < a.G) {
        float tot = 0.f, tl = 0.f;
        // the group's active rows, ascending (the rows left out are exact zeros of the full sum)
        const int w0 = gl * (NS / 32);
        const int c_lo = w0 == 0 ? abase[0] : (w0 == 1 ? abase[1] : (w0 == 2 ? abase[2] : abase[3]));
        const int w1 = w0 + NS / 32;
        const int c_hi = w1 == 1 ? abase[1] : (w1 == 2 ? abase[2] : (w1 == 3 ? abase[3] : abase[4]));
        const uint8_t* mp = m1s + (col >> 2);
        for (int ci = c_lo; ci < c_hi; ++ci) {
          const int row = rowmap[ci], j = row - gl * NS;
          const float gv = AH[ci * GS_LDH + cl];
          const bool pos = (mp[row * (C1 / 4)] >> (col & 3)) & 1;
          const float v = pos ? gv : gv * 0.f;
          tot += v;
          if ((s_rep[gl * 4 + (j >> 5)] >> (j & 31)) & 1u) tl += v;
        }
        a.gBc[(int64_t)g * C1 + col] = tot;
        a.tail[(int64_t)g * C1 + col] = tl;
      }
    }
    __syncthreads();
  }
}


// ---------------------------------------------------------------------------------------------------------------------------
// The same launch over PACKED tiles. Above, a tile is 128 consecutive rows of which 35-42 % are active at SSG's levels:
// its MFMA blocks are compacted, but the tile's fixed costs (weight slices, barriers, the epilogue) are paid per 45-54
// rows. Here a tile is a run of WHOLE groups whose active rows fill it: gs_tiles_kernel (one workgroup) gives group g the
// weight w = max(active rows, 128 / GT), takes the exclusive prefix P and puts the group into tile P / Q with
// Q = 129 - max w — a tile's weights then sum to <= Q - 1 + max w = 128, consecutive groups differ by <= 1 tile (w <= Q
// because w <= 64), and a tile holds <= GT = 512 / ns groups. Everything else is the kernel above with "row of the tile"
// replaced by "row among the tile's groups' ns-row blocks" (<= 512 of them: sixteen words of flags).
constexpr int GS_TILES_LDS = 48 * 1024;                     // weights of up to 48K groups stay in LDS (else in the scratch)
__global__ __launch_bounds__(1024) void gs_tiles_kernel(const uint32_t* __restrict__ amask, int G, int nw, int wmin,
                                                        int* __restrict__ tiles, uint8_t* __restrict__ w8g) {
  __shared__ uint8_t s_w[GS_TILES_LDS];
  __shared__ int s_red[16];
  __shared__ int s_wmax;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  uint8_t* w8 = G <= GS_TILES_LDS ? s_w : w8g;
  int mx = wmin;
  if (nw == 1) {
#pragma unroll 8
    for (int g = tid; g < G; g += 1024) {
      const int cnt = __builtin_popcount(amask[g]);
      const int wt = cnt > wmin ? cnt : wmin;
      w8[g] = (uint8_t)wt;
      mx = mx > wt ? mx : wt;
    }
  } else {
#pragma unroll 8
    for (int g = tid; g < G; g += 1024) {
      const uint2 v = *reinterpret_cast<const uint2*>(amask + 2 * (int64_t)g);
      const int cnt = __builtin_popcount(v.x) + __builtin_popcount(v.y);
      const int wt = cnt > wmin ? cnt : wmin;
      w8[g] = (uint8_t)wt;
      mx = mx > wt ? mx : wt;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const int other = __shfl_xor(mx, o);
    mx = mx > other ? mx : other;
  }
  if (lane == 0) s_red[wave] = mx;
  __syncthreads();
  if (tid == 0) {
    int m = s_red[0];
    for (int i = 1; i < 16; ++i) m = m > s_red[i] ? m : s_red[i];
    s_wmax = m;
  }
  __syncthreads();
  const int Q = 129 - s_wmax;
  const int per = (G + 1023) / 1024;
  const int lo = tid * per < G ? tid * per : G, hi = lo + per < G ? lo + per : G;
  int sum = 0;
  for (int g = lo; g < hi; ++g) sum += w8[g];
  int incl = sum;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int up = __shfl_up(incl, o);
    if (lane >= o) incl += up;
  }
  __syncthreads();                                          // (s_red is read above)
  if (lane == 63) s_red[wave] = incl;
  __syncthreads();
  int P = incl - sum;
  for (int i = 0; i < wave; ++i) P += s_red[i];
  if (lo < hi) {
    // tile of a group = P / Q; one division per thread, then the boundary moves by at most one tile per group (w <= Q)
    int prev = lo == 0 ? -1 : (P - (int)w8[lo - 1]) / Q;
    int t = P / Q, nextb = (t + 1) * Q;
    for (int g = lo; g < hi; ++g) {
      if (P >= nextb) ++t, nextb += Q;
      if (t != prev) tiles[2 + t] = g, prev = t;
      P += w8[g];
    }
    if (hi == G) tiles[0] = prev + 1, tiles[1] = Q, tiles[2 + prev + 1] = G;
  }
}

template <int NS, int TN>
__global__ __launch_bounds__(GS_T, (TN > 2 ? 2 : 3)) void gemm_groupsum_packed_kernel(GroupSumArgs a, const int* __restrict__ tiles) {
  constexpr int GT = 512 / NS, NW = NS / 32;                // groups per tile at most, flag words per group
  constexpr int C1 = 32 * TN;
  constexpr int NBUF = TN > 2 ? 2 : 1;
  if ((int)blockIdx.x >= tiles[0]) return;
  extern __shared__ __attribute__((aligned(16))) float gs_lds[];
  float* AH = gs_lds;                                      // [128][GS_LDH]
  float* Ws = AH + GS_BM * GS_LDH;                          // [NBUF][C1][GS_LDW]
  uint8_t* m1s = reinterpret_cast<uint8_t*>(Ws + NBUF * C1 * GS_LDW);    // [128][C1 / 4] layer-1 sign bits of the COMPACT rows
  uint32_t* s_rep = reinterpret_cast<uint32_t*>(m1s + GS_BM * (C1 / 4));  // [16] bit f: flat row f repeats its group's first index
  uint32_t* s_aw = s_rep + 16;                                             // [16] bit f: flat row f is active
  int* s_base = reinterpret_cast<int*>(s_aw + 16);                         // [17] active rows before word w
  int* rowmap = s_base + 20;                                               // [128] compact row -> flat row
  int* zlist = rowmap + GS_BM;                                             // [512] flat rows to be written as zeros
  __shared__ int s_nzero;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int g0 = tiles[2 + blockIdx.x], ng = tiles[3 + blockIdx.x] - g0;
  const int64_t row0 = (int64_t)g0 * NS;                    // flat row f of the tile = row row0 + f of the tensors

  if (wave == 0) {
    const uint32_t word = lane < ng * NW ? a.amask[(int64_t)g0 * NW + lane] : 0u;
    int incl = __builtin_popcount(word);
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
      const int up = __shfl_up(incl, o);
      if (lane >= o) incl += up;
    }
    if (lane < 16) s_aw[lane] = word, s_base[lane + 1] = incl;
    if (lane == 0) s_base[0] = 0, s_nzero = 0;
  }
  for (int c = wave; c < 8; c += 4) {                        // padding flags: 64 flat rows per ballot
    const int f = c * 64 + lane, gl = f / NS, j = f - gl * NS;
    bool rp = false;
    if (gl < ng && j > 0) rp = a.idx[row0 + f] == a.idx[row0 + gl * NS];
    const unsigned long long bal = __builtin_amdgcn_ballot_w64(rp);
    if (lane == 0) s_rep[2 * c] = (uint32_t)bal, s_rep[2 * c + 1] = (uint32_t)(bal >> 32);
  }
  __syncthreads();
  const int n_act = s_base[16];
  for (int f = tid; f < 512; f += GS_T) {
    const uint32_t word = s_aw[f >> 5];
    const int bit = f & 31;
    if ((word >> bit) & 1u) rowmap[s_base[f >> 5] + __builtin_popcount(word & ((1u << bit) - 1u))] = f;
    else if (f < ng * NS && !((s_rep[f >> 5] >> bit) & 1u)) zlist[atomicAdd(&s_nzero, 1)] = f;
  }
  __syncthreads();
  const int nzero = s_nzero;

  float4 xa[8];
  auto fetch_x = [&](int kc) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int f = u * GS_T + tid, crow = f >> 4, c4 = f & 15;
      const int k = kc + 4 * c4;
      xa[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (crow < n_act && k < a.K) xa[u] = *reinterpret_cast<const float4*>(a.X + (row0 + rowmap[crow]) * a.ldx + k);
    }
  };
  auto stash_x = [&]() {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int f = u * GS_T + tid;
      *reinterpret_cast<float4*>(AH + (f >> 4) * GS_LDH + 4 * (f & 15)) = xa[u];
    }
  };
  const int lrow = tid >> 3, lk = (tid & 7) * 4;
  float4 wb[TN];
  auto fetch_w = [&](int k0) {
#pragma unroll
    for (int q = 0; q < TN; ++q)
      wb[q] = (k0 + lk < a.K) ? *reinterpret_cast<const float4*>(a.Wt + (int64_t)(q * 32 + lrow) * a.K + k0 + lk)
                              : make_float4(0.f, 0.f, 0.f, 0.f);
  };
  auto stash_w = [&](float* dst) {
#pragma unroll
    for (int q = 0; q < TN; ++q) *reinterpret_cast<float4*>(dst + (q * 32 + lrow) * GS_LDW + lk) = wb[q];
  };
  fetch_x(0);
  fetch_w(0);
  for (int e = tid; e < n_act * TN; e += GS_T) {             // the compact rows' sign bytes, eight at a time
    const int crow = e / TN, p = e - crow * TN;
    *reinterpret_cast<uint2*>(m1s + crow * (C1 / 4) + 8 * p) =
        *reinterpret_cast<const uint2*>(a.mask1 + (row0 + rowmap[crow]) * (C1 / 4) + 8 * p);
  }

  gs_f32x16 acc[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

  int cur = 0;
  for (int kc = 0; kc < a.K; kc += GS_KC) {
    if (kc > 0) __syncthreads();
    stash_x();
    stash_w(Ws + cur * C1 * GS_LDW);
    __syncthreads();
    if (kc + GS_KC < a.K) fetch_x(kc + GS_KC);
    const int nks = (a.K - kc < GS_KC ? a.K - kc : GS_KC) / GS_BK;
    for (int ks = 0; ks < nks; ++ks) {
      const int knext = kc + (ks + 1) * GS_BK;
      if (knext < a.K) fetch_w(knext);
      if (wave * 32 < n_act)
        gs_step<TN>(acc, AH + (wave * 32 + r) * GS_LDH + ks * GS_BK + 4 * h, Ws + cur * C1 * GS_LDW + r * GS_LDW + 4 * h);
      if (NBUF == 2) {
        if (ks + 1 < nks) {
          stash_w(Ws + (cur ^ 1) * C1 * GS_LDW);
          __syncthreads();
        }
        cur ^= 1;
      } else if (ks + 1 < nks) {
        __syncthreads();
        stash_w(Ws);
        __syncthreads();
      }
    }
  }
  __syncthreads();

  for (int hh = 0; hh < C1; hh += 64) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      if (j * 32 < hh || j * 32 >= hh + 64) continue;
      float* hp = AH + (wave * 32 + 4 * h) * GS_LDH + (j * 32 - hh) + r;
#pragma unroll
      for (int e = 0; e < 16; ++e) hp[((e & 3) + 8 * (e >> 2)) * GS_LDH] = acc[j][e];
    }
    __syncthreads();
    const int wcols = C1 - hh < 64 ? C1 - hh : 64;
    for (int f = tid; f < n_act * 16; f += GS_T) {
      const int crow = f >> 4, c4 = f & 15;
      if (4 * c4 < wcols)
        *reinterpret_cast<float4*>(a.Y + (row0 + rowmap[crow]) * C1 + hh + 4 * c4) =
            *reinterpret_cast<const float4*>(AH + crow * GS_LDH + 4 * c4);
    }
    for (int f = tid; f < nzero * 16; f += GS_T) {
      const int t = zlist[f >> 4], c4 = f & 15;
      if (4 * c4 < wcols) *reinterpret_cast<float4*>(a.Y + (row0 + t) * C1 + hh + 4 * c4) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int t = tid; t < GT * 64; t += GS_T) {
      const int gl = t >> 6, cl = t & 63;
      if (gl >= ng) break;                                  // (gl grows with t)
      const int col = hh + cl;
      if (cl < wcols) {
        float tot = 0.f, tl = 0.f;
        const int c_lo = s_base[gl * NW], c_hi = s_base[(gl + 1) * NW];
        const uint8_t* mp = m1s + (col >> 2);
        for (int ci = c_lo; ci < c_hi; ++ci) {
          const int f = rowmap[ci];
          const float gv = AH[ci * GS_LDH + cl];
          const bool pos = (mp[ci * (C1 / 4)] >> (col & 3)) & 1;
          const float v = pos ? gv : gv * 0.f;
          tot += v;
          if ((s_rep[f >> 5] >> (f & 31)) & 1u) tl += v;
        }
        a.gBc[(int64_t)(g0 + gl) * C1 + col] = tot;
        a.tail[(int64_t)(g0 + gl) * C1 + col] = tl;
      }
    }
    __syncthreads();
  }
}

}  // namespace pc3d

using namespace pc3d;

extern "C" int pc3d_gemm_nt_groupsum_f32(const float* X, int64_t ldx, const float* Wt, const uint8_t* mask1, const int32_t* idx,
                                         const uint32_t* amask, int B, int S, int ns, int C1, int K, float* Y, float* gBc,
                                         float* tail, void* stream) {
  const char* nm = "pc3d_gemm_nt_groupsum_f32";
  PC3D_REQUIRE(B >= 0 && S >= 1 && (ns == 32 || ns == 64 || ns == 128), "%s: bad sizes B=%d S=%d ns=%d (ns in {32,64,128})", nm, B, S, ns);
  PC3D_REQUIRE((C1 == 32 || C1 == 64 || C1 == 128) && K >= 32 && K % 32 == 0 && ldx >= K && ldx % 4 == 0,
               "%s: widths C1=%d K=%d ldx=%lld (C1 in {32,64,128}, K a multiple of 32)", nm, C1, K, (long long)ldx);
  PC3D_REQUIRE((int64_t)B * S * ns <= 0x7fffffffLL, "%s: problem too large", nm);
  if (B == 0) return PC3D_OK;
  PC3D_REQUIRE(X && Wt && mask1 && idx && Y && gBc && tail, "%s: null pointer", nm);
  PC3D_REQUIRE(((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(Wt) | reinterpret_cast<uintptr_t>(Y) |
                 reinterpret_cast<uintptr_t>(mask1)) & 15) == 0, "%s: X / Wt / Y / mask1 must be 16-byte aligned", nm);
  const int G = B * S, M = G * ns;
  GroupSumArgs a{X, ldx, Wt, mask1, idx, amask, M, G, K, Y, gBc, tail};
  const int gpt = GS_BM / ns;
  const size_t lds = ((size_t)GS_BM * GS_LDH + (size_t)(C1 > 64 ? 2 : 1) * C1 * GS_LDW) * sizeof(float) + (size_t)GS_BM * (C1 / 4) + (size_t)gpt * 16 +
                     2 * GS_BM * sizeof(int);
  const dim3 grid(cdiv(M, GS_BM)), block(GS_T);
  hipStream_t st = as_stream(stream);
#define PC3D_GS(NSV, TNV)                                                                                              \
  do {                                                                                                                 \
    if (lds > 64 * 1024)                                                                                               \
      if (hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_groupsum_kernel<NSV, TNV>),            \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                    \
          e != hipSuccess) { set_error("%s: LDS opt-in failed: %s", nm, hipGetErrorString(e)); return (int)e; }         \
    hipLaunchKernelGGL((gemm_groupsum_kernel<NSV, TNV>), grid, block, lds, st, a);                                      \
  } while (0)
#define PC3D_GS_NS(NSV)              \
  do {                               \
    if (C1 == 32) PC3D_GS(NSV, 1);   \
    else if (C1 == 64) PC3D_GS(NSV, 2); \
    else PC3D_GS(NSV, 4);            \
  } while (0)
  if (ns == 32) PC3D_GS_NS(32);
  else if (ns == 64) PC3D_GS_NS(64);
  else PC3D_GS_NS(128);
#undef PC3D_GS_NS
#undef PC3D_GS
  PC3D_LAUNCH_CHECK(nm);
  return PC3D_OK;
}

extern "C" int pc3d_gemm_nt_groupsum_packed_f32(const float* X, int64_t ldx, const float* Wt, const uint8_t* mask1,
                                                const int32_t* idx, const uint32_t* amask, int B, int S, int ns, int C1, int K,
                                                float* Y, float* gBc, float* tail, int32_t* scratch, void* stream) {
  const char* nm = "pc3d_gemm_nt_groupsum_packed_f32";
  PC3D_REQUIRE(B >= 0 && S >= 1 && (ns == 32 || ns == 64), "%s: bad sizes B=%d S=%d ns=%d (ns in {32,64})", nm, B, S, ns);
  PC3D_REQUIRE((C1 == 32 || C1 == 64 || C1 == 128) && K >= 32 && K % 32 == 0 && ldx >= K && ldx % 4 == 0,
               "%s: widths C1=%d K=%d ldx=%lld (C1 in {32,64,128}, K a multiple of 32)", nm, C1, K, (long long)ldx);
  PC3D_REQUIRE((int64_t)B * S * ns <= 0x7fffffffLL, "%s: problem too large", nm);
  if (B == 0) return PC3D_OK;
  PC3D_REQUIRE(X && Wt && mask1 && idx && amask && Y && gBc && tail && scratch, "%s: null pointer", nm);
  PC3D_REQUIRE(((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(Wt) | reinterpret_cast<uintptr_t>(Y) |
                 reinterpret_cast<uintptr_t>(mask1)) & 15) == 0, "%s: X / Wt / Y / mask1 must be 16-byte aligned", nm);
  PC3D_REQUIRE((reinterpret_cast<uintptr_t>(amask) & 7) == 0, "%s: amask must be 8-byte aligned", nm);
  const int G = B * S, M = G * ns;
  GroupSumArgs a{X, ldx, Wt, mask1, idx, amask, M, G, K, Y, gBc, tail};
  hipStream_t st = as_stream(stream);
  // scratch: [0] tiles, [1] Q, [2 .. 2 + G] first group of each tile, then G bytes of weights
  uint8_t* w8 = reinterpret_cast<uint8_t*>(scratch + G + 4);
  hipLaunchKernelGGL(gs_tiles_kernel, dim3(1), dim3(1024), 0, st, amask, G, ns / 32, GS_BM / (512 / ns), scratch, w8);  // (nw is 1 or 2)
  const size_t lds = ((size_t)GS_BM * GS_LDH + (size_t)(C1 > 64 ? 2 : 1) * C1 * GS_LDW) * sizeof(float) + (size_t)GS_BM * (C1 / 4) +
                     (16 + 16 + 20 + GS_BM + 512) * sizeof(int);
  const dim3 grid(cdiv(M, 129 - ns) + 1), block(GS_T);      // an upper bound of the tile count; the rest return at once
#define PC3D_GSP(NSV, TNV)                                                                                             \
  do {                                                                                                                 \
    if (lds > 64 * 1024)                                                                                               \
      if (hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_groupsum_packed_kernel<NSV, TNV>),     \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                    \
          e != hipSuccess) { set_error("%s: LDS opt-in failed: %s", nm, hipGetErrorString(e)); return (int)e; }         \
    hipLaunchKernelGGL((gemm_groupsum_packed_kernel<NSV, TNV>), grid, block, lds, st, a, scratch);                      \
  } while (0)
#define PC3D_GSP_NS(NSV)                 \
  do {                                   \
    if (C1 == 32) PC3D_GSP(NSV, 1);      \
    else if (C1 == 64) PC3D_GSP(NSV, 2); \
    else PC3D_GSP(NSV, 4);               \
  } while (0)
  if (ns == 32) PC3D_GSP_NS(32);
  else PC3D_GSP_NS(64);
#undef PC3D_GSP_NS
#undef PC3D_GSP
  PC3D_LAUNCH_CHECK(nm);
  return PC3D_OK;
}
