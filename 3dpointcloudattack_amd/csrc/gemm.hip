// Dense point-wise layers of the victims as ONE fp32-MFMA kernel:   Y[M,N] = act( gate(X)[M,K] . W[N,K]^T + bias[N] + R[M,N] )
//
// Replaces the library GEMMs (hipBLASLt through F.linear / torch._addmm_activation) + separate activation passes of
//   * the shared MLPs of PointNet++ set abstraction, layers 1-2 (model/pointnet2_utils.py:190-197,243-257: Conv2d 1x1 +
//     BatchNorm2d + ReLU over [B,C,ns,S] — here channels-last rows [B*S*ns, C] with BN folded into W, bias),
//   * DGCNN's EdgeConv point-wise products [P | Q] = x [U ; V]^T and conv5 (model/dgcnn.py:297-320),
//   * CurveNet's 1x1 convolutions (model/curvenet_util.py:189-193,321-331),
// and, with W^T in place of W and `gate` = the layer's own output, their backward to the input
//   dX = (dY * act'(Y)) . W   (weights are frozen: no weight gradient exists on the attack path).
//
// Exact fp32: v_mfma_f32_32x32x2_f32 (products and accumulation in fp32, no tf32-style truncation), so logits keep
// their label parity with the reference.
//
// Tiling (gfx950): workgroup = 128 rows x 128 columns, 4 waves in 2 x 2, each wave 64 x 64 = 2 x 2 MFMA tiles (64
// accumulator VGPRs). K advances in steps of 32 through LDS: As[128][32+4], Bs[128][32+4] (the +4 floats per row keep
// ds_read_b128 conflict-free); the next step's global loads (8 x float4 per thread) are issued before the current
// step's MFMAs and parked in registers. Operand k-order inside a step: lane half h of MFMA e of group t holds
// k = 8 t + 4 h + e for BOTH operands, so one ds_read_b128 per 32-row block feeds four MFMAs.
// 64 MFMAs (4096 cycles on the matrix pipe) per 16 ds_read_b128 and 8 global float4 loads per wave and step.
#include "pc3d_common.h"

namespace pc3d {

using gm_f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int GM_T = 256;
constexpr int GM_BK = 32, GM_LD = GM_BK + 4;

struct GemmArgs {
  const float* X;      // [M, K], row stride ldx
  const float* W;      // [N, K] row-major (row stride K)
  const float* bias;   // [N] or null
  const float* gate;   // [M, K] (row stride ldg) or null: X is read as  gate > 0 ? X : gslope * X
  const float* res;    // [M, N] (row stride ldr) or null: added before the activation (residual / shortcut branch)
  float* Y;            // [M, N], row stride ldy
  int64_t ldx, ldg, ldy, ldr;
  int M, N, K;
  int act;             // 0 none, 1 ReLU, 2 LeakyReLU(slope)
  float slope, gslope;
  // group-max epilogue (gm_ns > 0): rows are groups of gm_ns consecutive rows (32, 64 or 128; M a multiple of it);
  // instead of Y the kernel writes gm_out[g][n] = relu(max_rows(acc) + bias[n]) and the winning row inside the group
  int gm_ns;
  float* gm_out;       // [M / gm_ns, N]
  int64_t* gm_arg;     // [M / gm_ns, N]
  // gathered A operand (ga_idx != null): row m = (group g = m / ga_ns, member j) is GENERATED on load as
  //   act_in(X[(g / ga_S) * ga_NA + ga_idx[m], :] + ga_Bc[g, :])     act_in = LeakyReLU(ga_slope), 0 = ReLU
  // i.e. the output of a set-abstraction MLP's first layer in its per-point form (group.hip, pc3d_group_act_f32) without
  // that [M, K] tensor ever existing; an index outside [0, ga_NA) reads as a zero row of X.
  const int32_t* ga_idx;
  const float* ga_Bc;  // [M / ga_ns, K]
  int ga_ns, ga_S, ga_NA;
  float ga_slope;
  uint8_t* ga_mask;    // [M, K / 4] or null: bit c % 4 of byte c / 4 of row m = "the generated element (m, c) came from a
                       // positive pre-activation" — what the backward of act_in needs, 1/32 of the tensor it replaces
  // pooled-gradient A operand (pb_g != null): X holds the PRE-activation Y [M, K] of a layer that was followed by
  // LeakyReLU(pb_slope) and [max | mean] pooling over the pb_N rows of every cloud; row m = (cloud b, point n) is
  // generated as  dY[m, c] = (Y > 0 ? 1 : pb_slope) * ((pb_arg[b,c] == n ? g[b, c] : 0) + g[b, K + c] / pb_N)
  // — the backward of that pooling (act_pool_bwd_kernel, dgcnn.hip) without the [M, K] gradient tensor.
  uint32_t* ymask;     // [M, N / 32] or null (N % 32 == 0): bit n % 32 of word n / 32 of row m = "Y[m, n] > 0 before the
                       // activation" — the activation's backward mask, written by the standard epilogue (16 ballots per tile)
  const float* pb_g;       // [M / pb_N, 2K]
  const int32_t* pb_arg;   // [M / pb_N, K]
  int pb_N;
  float pb_slope;
};

__device__ __forceinline__ float4 gm_load4(const float* p, int k, int K, bool row_ok) {
  // 4 consecutive k of one row with zero fill past K / past the matrix edge; float4 when aligned and fully inside
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (!row_ok) return v;
  if (k + 3 < K && ((reinterpret_cast<uintptr_t>(p + k) & 15) == 0)) return *reinterpret_cast<const float4*>(p + k);
  if (k < K) v.x = p[k];
  if (k + 1 < K) v.y = p[k + 1];
  if (k + 2 < K) v.z = p[k + 2];
  if (k + 3 < K) v.w = p[k + 3];
  return v;
}

// WM x WN waves of 64 x 64 outputs each: <2,2> = 128 x 128 tile (wide layers), <4,1> = 256 x 64 (layers with <= 64
// outputs: no half-empty MFMA tiles, and the kernel is then bound by reading / writing the [M, 64] activations).
// KS > 1: K split INSIDE the workgroup — KS groups of WM x WN waves work on the same output tile, group g on K steps
// g, g + KS, ...; a step stages KS slices of 32 k and the groups' accumulators are summed through LDS before the
// epilogue. For launches with few tiles and a long K (CurveNet's deep levels): more waves per tile, not more tiles.
template <int WM, int WN, int TM, int TN, bool DB, int OCC, int GA = 0, int KS = 1>   // GA: generated A operand — 1: GemmArgs::ga_idx, 2: ::pb_g
__global__ __launch_bounds__(WM * WN * KS * 64, OCC) void gemm_nt_kernel(GemmArgs a) {
  constexpr int NT = WM * WN * KS * 64;          // threads
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr int RPS = NT / 8;                    // staged rows (tile row x K slice) covered by one pass of the staging threads
  constexpr int QA = BM * KS / RPS, QB = BN * KS / RPS;    // float4 per thread and step of the X / W tile
  static_assert((BM * KS) % RPS == 0 && (BN * KS) % RPS == 0, "tile rows must be a multiple of the staging pass");
  static_assert(KS == 1 || (GA == 0 && !DB), "the K split is written for the plain single-buffered path");
  extern __shared__ __attribute__((aligned(16))) float gm_lds[];   // 2 x (As[BM][GM_LD] + Bs[BN][GM_LD])
  // XCD-aware tile order: consecutive workgroup ids land on different XCDs (round-robin dispatch), so give each XCD a
  // contiguous band of row tiles — its L2 then holds one band of X and all of W instead of a slice of everything.
  const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + BN - 1) / BN;
  const int ntile = tiles_m * tiles_n;
  int wg = blockIdx.x;
  {
    const int nx = 8, per = (ntile + nx - 1) / nx;
    const int xcd = wg % nx, slot = wg / nx;
    const int t = xcd * per + slot;
    if (slot >= per || t >= ntile) return;   // (grid is rounded up to a multiple of 8 bands)
    wg = t;
  }
  const int tm = wg / tiles_n, tn = wg - tm * tiles_n;   // column tiles of one row band are neighbours: X stays in L2
  const int m0 = tm * BM, n0 = tn * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kg = wave / (WM * WN), wv = wave - kg * (WM * WN);   // K group of this wave, its place in the tile
  const int wm = (wv / WN) * (TM * 32), wn = (wv % WN) * (TN * 32);
  const int r = lane & 31, h = lane >> 5;

  gm_f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // global -> register staging: tile element (row = q*32 + tid/8, k = (tid%8)*4): 8 threads cover one row's 32 k
  // => 128 B contiguous per row
  const int lrow = tid >> 3, lk = (tid & 7) * 4;
  float4 xa[QA], wb[QB];
  // gathered A operand: source row and group of this thread's QA tile rows, resolved once
  int ga_row[GA ? QA : 1], ga_grp[GA ? QA : 1];
  if constexpr (GA == 2) {     // (cloud, point) of this thread's tile rows
#pragma unroll
    for (int q = 0; q < QA; ++q) {
      const int gm = m0 + q * RPS + lrow;
      ga_grp[q] = gm / a.pb_N;
      ga_row[q] = gm - ga_grp[q] * a.pb_N;
    }
  }
  if constexpr (GA == 1) {
#pragma unroll
    for (int q = 0; q < QA; ++q) {
      const int gm = m0 + q * RPS + lrow;
      ga_row[q] = -1, ga_grp[q] = 0;
      if (gm < a.M) {
        const int g = gm / a.ga_ns, p = a.ga_idx[gm];
        ga_grp[q] = g;
        if ((unsigned)p < (unsigned)a.ga_NA) ga_row[q] = (g / a.ga_S) * a.ga_NA + p;   // < B * NA <= 2^31 (entry point)
      }
    }
  }
  float4 pb_gx = make_float4(0.f, 0.f, 0.f, 0.f), pb_ge = pb_gx;
  int4 pb_ar = make_int4(0, 0, 0, 0);
  auto fetch = [&](int k0) {
#pragma unroll
    for (int q = 0; q < QA; ++q) {
      const int gm = m0 + q * RPS + lrow;
      if constexpr (GA == 2) {
        // a row tile lies inside ONE cloud when pb_N is a multiple of the tile height (the entry point checks it), so the
        // per-(cloud, channel) operands are those of tile row 0 for every q: loaded once per K step (q == 0)
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gm < a.M && k0 + lk < a.K) {       // K % 4 == 0 (entry point): whole float4s
          const int k = k0 + lk;
          const float4 y = *reinterpret_cast<const float4*>(a.X + (int64_t)gm * a.ldx + k);
          if (q == 0) {
            const float* gb = a.pb_g + (int64_t)ga_grp[0] * 2 * a.K + k;
            pb_gx = *reinterpret_cast<const float4*>(gb);
            pb_ge = *reinterpret_cast<const float4*>(gb + a.K);
            pb_ar = *reinterpret_cast<const int4*>(a.pb_arg + (int64_t)ga_grp[0] * a.K + k);
            const float inv = 1.f / (float)a.pb_N;
            pb_ge.x *= inv, pb_ge.y *= inv, pb_ge.z *= inv, pb_ge.w *= inv;
          }
          const int n = ga_row[q];
          v.x = (y.x > 0.f ? 1.f : a.pb_slope) * ((pb_ar.x == n ? pb_gx.x : 0.f) + pb_ge.x);
          v.y = (y.y > 0.f ? 1.f : a.pb_slope) * ((pb_ar.y == n ? pb_gx.y : 0.f) + pb_ge.y);
          v.z = (y.z > 0.f ? 1.f : a.pb_slope) * ((pb_ar.z == n ? pb_gx.z : 0.f) + pb_ge.z);
          v.w = (y.w > 0.f ? 1.f : a.pb_slope) * ((pb_ar.w == n ? pb_gx.w : 0.f) + pb_ge.w);
        }
        xa[q] = v;
        continue;
      }
      if constexpr (GA == 1) {
        float4 v = gm_load4(a.X + (int64_t)ga_row[q] * a.ldx, k0 + lk, a.K, ga_row[q] >= 0);
        const float4 c = gm_load4(a.ga_Bc + (int64_t)ga_grp[q] * a.K, k0 + lk, a.K, gm < a.M);
        v.x += c.x, v.y += c.y, v.z += c.z, v.w += c.w;
        if (a.ga_mask && tn == 0 && gm < a.M && k0 + lk < a.K)
          a.ga_mask[(int64_t)gm * (a.K >> 2) + ((k0 + lk) >> 2)] =
              (uint8_t)((v.x > 0.f ? 1 : 0) | (v.y > 0.f ? 2 : 0) | (v.z > 0.f ? 4 : 0) | (v.w > 0.f ? 8 : 0));
        v.x = v.x > 0.f ? v.x : v.x * a.ga_slope, v.y = v.y > 0.f ? v.y : v.y * a.ga_slope;
        v.z = v.z > 0.f ? v.z : v.z * a.ga_slope, v.w = v.w > 0.f ? v.w : v.w * a.ga_slope;
        xa[q] = v;
        continue;
      }
      if constexpr (KS > 1) {
        const int sr = q * RPS + lrow, ks = sr / BM, row = m0 + (sr - ks * BM), k = k0 + ks * GM_BK + lk;
        float4 v = gm_load4(a.X + (int64_t)row * a.ldx, k, a.K, row < a.M);
        if (a.gate) {
          const float4 g = gm_load4(a.gate + (int64_t)row * a.ldg, k, a.K, row < a.M);
          v.x = g.x > 0.f ? v.x : a.gslope * v.x;
          v.y = g.y > 0.f ? v.y : a.gslope * v.y;
          v.z = g.z > 0.f ? v.z : a.gslope * v.z;
          v.w = g.w > 0.f ? v.w : a.gslope * v.w;
        }
        xa[q] = v;
        continue;
      }
      float4 v = gm_load4(a.X + (int64_t)gm * a.ldx, k0 + lk, a.K, gm < a.M);
      if (a.gate) {
        const float4 g = gm_load4(a.gate + (int64_t)gm * a.ldg, k0 + lk, a.K, gm < a.M);
        v.x = g.x > 0.f ? v.x : a.gslope * v.x;
        v.y = g.y > 0.f ? v.y : a.gslope * v.y;
        v.z = g.z > 0.f ? v.z : a.gslope * v.z;
        v.w = g.w > 0.f ? v.w : a.gslope * v.w;
      }
      xa[q] = v;
    }
#pragma unroll
    for (int q = 0; q < QB; ++q) {
      if constexpr (KS > 1) {
        const int sr = q * RPS + lrow, ks = sr / BN, gn = n0 + (sr - ks * BN);
        wb[q] = gm_load4(a.W + (int64_t)gn * a.K, k0 + ks * GM_BK + lk, a.K, gn < a.N);
        continue;
      }
      const int gn = n0 + q * RPS + lrow;
      wb[q] = gm_load4(a.W + (int64_t)gn * a.K, k0 + lk, a.K, gn < a.N);
    }
  };
  auto stash = [&](float* As, float* Bs) {
#pragma unroll
    for (int q = 0; q < QA; ++q) *reinterpret_cast<float4*>(As + (q * RPS + lrow) * GM_LD + lk) = xa[q];
#pragma unroll
    for (int q = 0; q < QB; ++q) *reinterpret_cast<float4*>(Bs + (q * RPS + lrow) * GM_LD + lk) = wb[q];
  };

  constexpr int BUF = (BM + BN) * KS * GM_LD;   // staged rows are (K slice, tile row): slice s of the X tile at s * BM
  constexpr int KSTEP = GM_BK * KS;
  fetch(0);
  stash(gm_lds, gm_lds + BM * KS * GM_LD);
  __syncthreads();
  int cur = 0;
  for (int k0 = 0; k0 < a.K; k0 += KSTEP) {
    const bool more = k0 + KSTEP < a.K;
    if (more) fetch(k0 + KSTEP);   // global loads in flight during this step's MFMAs
    const float* As = gm_lds + cur * BUF + kg * BM * GM_LD;
    const float* Bs = gm_lds + cur * BUF + BM * KS * GM_LD + kg * BN * GM_LD;
    const int kleft = a.K - k0 - kg * GM_BK;
    const int groups = kleft >= GM_BK ? GM_BK / 8 : (kleft > 0 ? (kleft + 7) / 8 : 0);   // narrow layers (K = 3) run one group, not four
    if (groups == GM_BK / 8) {
      // a full K step, unrolled, with the operand fragments of group t + 1 requested before the MFMAs of group t are
      // issued (the rolled loop below waits for each group's ds_read_b128 right before its MFMAs)
      const float* Ap = As + (wm + r) * GM_LD + 4 * h;
      const float* Bp = Bs + (wn + r) * GM_LD + 4 * h;
      float4 av[2][TM], bv[2][TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) av[0][i] = *reinterpret_cast<const float4*>(Ap + i * 32 * GM_LD);
#pragma unroll
      for (int j = 0; j < TN; ++j) bv[0][j] = *reinterpret_cast<const float4*>(Bp + j * 32 * GM_LD);
#pragma unroll
      for (int t = 0; t < GM_BK / 8; ++t) {
        const int c = t & 1, n = c ^ 1;
        if (t + 1 < GM_BK / 8) {
#pragma unroll
          for (int i = 0; i < TM; ++i) av[n][i] = *reinterpret_cast<const float4*>(Ap + i * 32 * GM_LD + 8 * (t + 1));
#pragma unroll
          for (int j = 0; j < TN; ++j) bv[n][j] = *reinterpret_cast<const float4*>(Bp + j * 32 * GM_LD + 8 * (t + 1));
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c][i].x, bv[c][j].x, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c][i].y, bv[c][j].y, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c][i].z, bv[c][j].z, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c][i].w, bv[c][j].w, acc[i][j], 0, 0, 0);
          }
      }
    } else
    for (int t = 0; t < groups; ++t) {
      float4 av[TM], bv[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) av[i] = *reinterpret_cast<const float4*>(As + (wm + i * 32 + r) * GM_LD + 8 * t + 4 * h);
#pragma unroll
      for (int j = 0; j < TN; ++j) bv[j] = *reinterpret_cast<const float4*>(Bs + (wn + j * 32 + r) * GM_LD + 8 * t + 4 * h);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].x, bv[j].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].y, bv[j].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].z, bv[j].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].w, bv[j].w, acc[i][j], 0, 0, 0);
        }
    }
    if (DB) {
      if (more) {
        float* An = gm_lds + (cur ^ 1) * BUF;    // the other buffer: nobody reads it during this step
        stash(An, An + BM * GM_LD);
      }
      __syncthreads();                           // one barrier per step (double-buffered tiles)
      cur ^= 1;
    } else if (more) {
      __syncthreads();                           // everyone is done reading the tile
      stash(gm_lds, gm_lds + BM * KS * GM_LD);
      __syncthreads();
    }
  }

  if constexpr (KS > 1) {   // sum the K groups' accumulators into group 0 (the operand tiles are dead: reuse their LDS)
    static_assert((KS - 1) * WM * WN * TM * TN * 16 * 64 <= (BM + BN) * KS * GM_LD, "K-split partial sums must fit the operand tiles");
    __syncthreads();
    float* red = gm_lds + ((kg > 0 ? kg - 1 : 0) * WM * WN + wv) * (TM * TN * 16 * 64);
    if (kg > 0) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) red[((i * TN + j) * 16 + e) * 64 + lane] = acc[i][j][e];
    }
    __syncthreads();
    if (kg > 0) return;
#pragma unroll
    for (int g = 1; g < KS; ++g) {
      const float* rg = gm_lds + ((g - 1) * WM * WN + wv) * (TM * TN * 16 * 64);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[i][j][e] += rg[((i * TN + j) * 16 + e) * 64 + lane];
    }
  }

  if (a.gm_ns) {
    // ---- group-max epilogue (the last layer of a set-abstraction MLP, model/pointnet2_utils.py:190-197: conv + BN + ReLU
    // + max over the group, without the [M,N] activation). ReLU and the bias are monotone: max first, then bias + ReLU.
    // Per 32-row MFMA tile and column: in-lane over the 16 accumulator rows (ascending in e: strict > keeps the lowest
    // row), across the two lane halves (rows interleave: compare (value, row)), then the tiles of one group through LDS.
    __syncthreads();                                   // the operand tiles are dead: reuse their LDS
    float* pv = gm_lds;                                // [BM / 32][BN] partial maxima
    int* pi = reinterpret_cast<int*>(gm_lds + (BM / 32) * BN);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        float best = -__builtin_inff();
        int bi = wm + i * 32;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int rl = wm + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
          if (m0 + rl < a.M && acc[i][j][e] > best) best = acc[i][j][e], bi = rl;
        }
        argmax_xor32(best, bi);
        if (h == 0) {
          pv[((wm >> 5) + i) * BN + wn + j * 32 + r] = best;
          pi[((wm >> 5) + i) * BN + wn + j * 32 + r] = bi;
        }
      }
    __syncthreads();
    const int tpg = a.gm_ns / 32;                      // 32-row tiles per group
    for (int t = tid; t < (BM / a.gm_ns) * BN; t += NT) {
      const int g = t / BN, col = t - g * BN;
      float best = pv[g * tpg * BN + col];
      int bi = pi[g * tpg * BN + col];
      for (int w = 1; w < tpg; ++w) {                  // ascending tile = ascending rows: strict >
        const float v = pv[(g * tpg + w) * BN + col];
        if (v > best) best = v, bi = pi[(g * tpg + w) * BN + col];
      }
      const int64_t grp = (int64_t)(m0 / a.gm_ns) + g;
      if ((int64_t)m0 + (int64_t)g * a.gm_ns < a.M && n0 + col < a.N) {
        const float bj = a.bias ? a.bias[n0 + col] : 0.f;
        a.gm_out[grp * a.N + n0 + col] = fmaxf(best + bj, 0.f);
        a.gm_arg[grp * a.N + n0 + col] = bi - g * a.gm_ns;
      }
    }
    return;
  }

  // epilogue: D[row = sample][col = output]: lane holds column r, rows (e & 3) + 8 (e >> 2) + 4 h of each 32 x 32 tile
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = n0 + wn + j * 32 + r;
    const bool col_ok = col < a.N;
    if (!col_ok && !a.ymask) continue;
    const float bj = (a.bias && col_ok) ? a.bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        float v = acc[i][j][e] + bj;
        if (a.res && row < a.M && col_ok) v += a.res[(int64_t)row * a.ldr + col];
        if (a.ymask) {     // the 32 columns of this tile as one word per row: lanes 0-31 hold row `row`, lanes 32-63 row + 4
          const unsigned long long bal = __builtin_amdgcn_ballot_w64(v > 0.f && col_ok);
          if (r == 0 && row < a.M && n0 + wn + j * 32 < a.N)      // (N % 32 == 0: a 32-column tile is all in or all out)
            a.ymask[(int64_t)row * (a.N >> 5) + ((n0 + wn + j * 32) >> 5)] = (uint32_t)(h ? (bal >> 32) : bal);
        }
        if (row < a.M && col_ok) {
          if (a.act == 1) v = v > 0.f ? v : 0.f;
          else if (a.act == 2) v = v > 0.f ? v : a.slope * v;
          a.Y[(int64_t)row * a.ldy + col] = v;
        }
      }
  }
}

}  // namespace pc3d

using namespace pc3d;

// variant < 0: the library chooses the tiling from (N, K) — never from M, so the fp32 summation order of an output
// element does not depend on how many rows (clouds) share the launch; variants 11 / 12 (K split inside the workgroup:
// a different summation order) are only ever taken on explicit request (pc3d_gemm_nt_tiled_f32).
static int gemm_nt_launch(int variant, const float* X, int64_t ldx, const float* W, const float* bias, const float* gate, int64_t ldg,
                          float gate_slope, const float* R, int64_t ldr, int M, int N, int K, int act, float slope,
                          float* Y, int64_t ldy, void* stream, int gm_ns = 0, float* gm_out = nullptr,
                          int64_t* gm_arg = nullptr, const int32_t* ga_idx = nullptr, const float* ga_Bc = nullptr,
                          int ga_ns = 0, int ga_S = 0, int ga_NA = 0, float ga_slope = 0.f, uint8_t* ga_mask = nullptr,
                          const float* pb_g = nullptr, const int32_t* pb_arg = nullptr, int pb_N = 0, float pb_slope = 0.f,
                          uint32_t* ymask = nullptr) {
  PC3D_REQUIRE(M >= 0 && N >= 1 && K >= 1, "pc3d_gemm_nt_f32: bad sizes M=%d N=%d K=%d", M, N, K);
  PC3D_REQUIRE(act >= 0 && act <= 2, "pc3d_gemm_nt_f32: bad activation %d", act);
  if (M == 0) return PC3D_OK;
  PC3D_REQUIRE(X && W && (Y || gm_ns), "pc3d_gemm_nt_f32: null pointer");
  PC3D_REQUIRE(ldx >= K && ldy >= N && (!gate || ldg >= K) && (!R || ldr >= N),
               "pc3d_gemm_nt_f32: row stride smaller than the row");
  GemmArgs a{};
  a.X = X, a.W = W, a.bias = bias, a.gate = gate, a.res = R, a.Y = Y, a.ldx = ldx, a.ldg = ldg, a.ldy = ldy, a.ldr = ldr;
  a.M = M, a.N = N, a.K = K, a.act = act, a.slope = slope, a.gslope = gate_slope;
  a.gm_ns = gm_ns, a.gm_out = gm_out, a.gm_arg = gm_arg;
  a.ga_idx = ga_idx, a.ga_Bc = ga_Bc, a.ga_ns = ga_ns, a.ga_S = ga_S, a.ga_NA = ga_NA, a.ga_slope = ga_slope, a.ga_mask = ga_mask;
  a.pb_g = pb_g, a.pb_arg = pb_arg, a.pb_N = pb_N, a.pb_slope = pb_slope, a.ymask = ymask;
  PC3D_REQUIRE(!ymask || N % 32 == 0, "pc3d_gemm_nt_f32: the output sign mask needs N %% 32 == 0 (N=%d)", N);
  // Tile shapes, measured on MI355X (tools/bench_gemm.py, us; hipBLASLt beside them):
  //   layer [M,N,K]                 0: 128x128 DB   2: 128x64   4: 64x128   5: 128x128/8 waves   hipBLASLt
  //   DGCNN conv5 [32768,1024,512]       387           361         350            320               275
  //   SSG SA2 l2  [524288,128,128]       264           225         226            212               174
  //   SSG SA1 l2  [1048576,64,64]        339           160         242            222               146
  //   SSG SA1 l1  [1048576,64,3]         213            86         120            104                80
  // Wave tiles of 32 x 64 (two accumulator tiles, ~100 VGPRs) beat 64 x 64 everywhere: these layers are short in K
  // (2-16 steps), so what counts is how many workgroups a CU holds to cover the load -> MFMA -> store latency of each.
  //   0 / 1: 128x128, 4 waves of 64x64, double / single buffered     2: 128x64, 4 waves of 32x64 (N <= 64)
  //   3: 256x64, 4 waves of 64x64, double buffered                    4: 64x128, 4 waves of 32x64
  //   5 / 6: 128x128, 8 waves of 32x64, single / double buffered       8: as 5 with four workgroups per CU (<= 64 VGPRs)
  // Round 3, measured and not kept: 256x128 workgroup tiles with eight waves of 64x64 (a third less L2 -> LDS traffic per
  // flop, half the ds_read_b128 per MFMA) — conv5 335-338 us against 320 for variant 5, every narrower layer 10-60 %
  // slower; operand fragments of group t + 1 requested before the MFMAs of group t (kept: same 320 us — with four waves
  // per SIMD the ds_read latency was already hidden).
  int v = variant;
  PC3D_REQUIRE(v < 0 || v <= 6 || v == 8 || v == 11 || v == 12, "pc3d_gemm_nt_f32: unknown tile variant %d", v);
  PC3D_REQUIRE(v < 11 || (!ga_idx && !pb_g && !gm_ns && !ymask), "pc3d_gemm_nt_f32: the K-split variants take plain operands only");
  if (v < 0 || ga_idx || pb_g) v = (N <= 64) ? 2 : 5;
  // the group-max epilogue is written for 128-row tiles of 32-row wave tiles; with K <= 64 (two K steps per tile) four
  // workgroups per CU instead of two hide the tile prologue better: 221 -> 203 us on SSG's SA1 (no change at K = 128)
  if (gm_ns) v = (K <= 64) ? 8 : 5;
  // Few tiles and a long K (CurveNet's deep levels: M = B x 64 .. 256 rows, K up to 512: 16 - 64 tiles of 128 x 128 on
  // 256 CUs, 31 - 41 us a launch): 64 x 64 tiles with the K steps split over two groups of waves (variant 11), or
  // 32 x 64 tiles with four K groups of one wave (variant 12). The CALLER asks for them (the host picks from the rows
  // per cloud, see ops.gemm_nt): the K split changes the order in which an element's products are summed, and that
  // order must not depend on the batch size.
  int ks = 1;
  int bm, bn, db;
  switch (v) {
    case 11: bm = 64, bn = 64, db = 0, ks = 2; break;
    case 12: bm = 32, bn = 64, db = 0, ks = 4; break;
    case 1: bm = 128, bn = 128, db = 0; break;
    case 2: bm = 128, bn = 64, db = 0; break;
    case 3: bm = 256, bn = 64, db = 1; break;
    case 4: bm = 64, bn = 128, db = 0; break;
    case 5: bm = 128, bn = 128, db = 0; break;
    case 6: bm = 128, bn = 128, db = 1; break;
    case 8: bm = 128, bn = 128, db = 0; break;
    default: bm = 128, bn = 128, db = 1; break;
  }
  const long tiles = (long)cdiv(M, bm) * cdiv(N, bn);
  PC3D_REQUIRE(tiles <= 0x7fffff00L, "pc3d_gemm_nt_f32: too many tiles (%ld)", tiles);
  const int per = (int)((tiles + 7) / 8);
  const size_t lds = (size_t)(db ? 2 : 1) * (bm + bn) * ks * GM_LD * sizeof(float);
  const dim3 grid(per * 8), block(v == 5 || v == 6 || v == 8 ? 512 : GM_T);
  hipStream_t st = as_stream(stream);
  if (ga_idx) {
    if (v == 2) hipLaunchKernelGGL((gemm_nt_kernel<4, 1, 1, 2, false, 4, 1>), grid, block, lds, st, a);
    else hipLaunchKernelGGL((gemm_nt_kernel<4, 2, 1, 2, false, 2, 1>), grid, block, lds, st, a);
    PC3D_LAUNCH_CHECK("pc3d_gemm_nt_gather_f32");
    return PC3D_OK;
  }
  if (pb_g) {
    // the generated operand costs ~50 VALU instructions per K step between load arrival and the LDS store; with
    // double-buffered tiles that work overlaps the other waves' MFMAs instead of sitting between two barriers
    if (v == 2) hipLaunchKernelGGL((gemm_nt_kernel<4, 1, 1, 2, false, 4, 2>), grid, block, lds, st, a);
    else hipLaunchKernelGGL((gemm_nt_kernel<4, 2, 1, 2, true, 2, 2>), grid, block, 2 * lds, st, a);
    PC3D_LAUNCH_CHECK("pc3d_gemm_nt_poolbwd_f32");
    return PC3D_OK;
  }
  switch (v) {
    case 1: hipLaunchKernelGGL((gemm_nt_kernel<2, 2, 2, 2, false, 3>), grid, block, lds, st, a); break;
    case 2: hipLaunchKernelGGL((gemm_nt_kernel<4, 1, 1, 2, false, 4>), grid, block, lds, st, a); break;
    case 3: hipLaunchKernelGGL((gemm_nt_kernel<4, 1, 2, 2, true, 2>), grid, block, lds, st, a); break;
    case 4: hipLaunchKernelGGL((gemm_nt_kernel<2, 2, 1, 2, false, 4>), grid, block, lds, st, a); break;
    case 5: hipLaunchKernelGGL((gemm_nt_kernel<4, 2, 1, 2, false, 2>), grid, block, lds, st, a); break;
    case 6: hipLaunchKernelGGL((gemm_nt_kernel<4, 2, 1, 2, true, 2>), grid, block, lds, st, a); break;
    case 8: hipLaunchKernelGGL((gemm_nt_kernel<4, 2, 1, 2, false, 4>), grid, block, lds, st, a); break;
    case 11: hipLaunchKernelGGL((gemm_nt_kernel<2, 1, 1, 2, false, 2, 0, 2>), grid, block, lds, st, a); break;
    case 12: hipLaunchKernelGGL((gemm_nt_kernel<1, 1, 1, 2, false, 2, 0, 4>), grid, block, lds, st, a); break;
    default: hipLaunchKernelGGL((gemm_nt_kernel<2, 2, 2, 2, true, 2>), grid, block, lds, st, a); break;
  }
  PC3D_LAUNCH_CHECK("pc3d_gemm_nt_f32");
  return PC3D_OK;
}

// internal (not part of the ABI): pc3d_group_linear_max_f32 on the GEMM main loop for groups of 32 / 64 / 128 rows
namespace pc3d {
int gemm_nt_groupmax(const float* X, const float* W, const float* bias, int G, int ns, int K, int N, float* out, int64_t* arg,
                     void* stream) {
  return gemm_nt_launch(-1, X, K, W, bias, nullptr, 0, 0.f, nullptr, 0, G * ns, N, K, 0, 0.f, nullptr, N, stream, ns, out, arg);
}
}  // namespace pc3d

extern "C" int pc3d_gemm_nt_f32(const float* X, int64_t ldx, const float* W, const float* bias, const float* gate,
                                int64_t ldg, float gate_slope, int M, int N, int K, int act, float slope, float* Y,
                                int64_t ldy, void* stream) {
  return gemm_nt_launch(-1, X, ldx, W, bias, gate, ldg, gate_slope, nullptr, 0, M, N, K, act, slope, Y, ldy, stream);
}

extern "C" int pc3d_gemm_nt_tiled_f32(const float* X, int64_t ldx, const float* W, const float* bias, const float* gate,
                                      int64_t ldg, float gate_slope, int M, int N, int K, int act, float slope, float* Y,
                                      int64_t ldy, int variant, void* stream) {
  PC3D_REQUIRE(variant >= 0, "pc3d_gemm_nt_tiled_f32: variant %d (0-6, 8, 11, 12)", variant);
  return gemm_nt_launch(variant, X, ldx, W, bias, gate, ldg, gate_slope, nullptr, 0, M, N, K, act, slope, Y, ldy, stream);
}

extern "C" int pc3d_gemm_nt_res_f32(const float* X, int64_t ldx, const float* W, const float* bias, const float* R,
                                    int64_t ldr, int M, int N, int K, int act, float slope, float* Y, int64_t ldy,
                                    int variant, void* stream) {
  PC3D_REQUIRE(R != nullptr, "pc3d_gemm_nt_res_f32: null residual");
  return gemm_nt_launch(variant < 0 ? -1 : variant, X, ldx, W, bias, nullptr, 0, 0.f, R, ldr, M, N, K, act, slope, Y, ldy, stream);
}

extern "C" int pc3d_gemm_nt_gather_f32(const float* P, int64_t ldp, const float* Bc, const int32_t* idx, int B, int NA, int S,
                                       int ns, float slope_in, const float* W, const float* bias, int N, int K, int act,
                                       float slope, float* Y, int64_t ldy, uint8_t* mask, uint32_t* ymask, void* stream) {
  PC3D_REQUIRE(B >= 0 && NA >= 1 && S >= 1 && ns >= 1 && (int64_t)B * S * ns <= 0x7fffffffLL && (int64_t)B * NA <= 0x7fffffffLL,
               "pc3d_gemm_nt_gather_f32: bad sizes B=%d NA=%d S=%d ns=%d", B, NA, S, ns);
  if (B == 0) return PC3D_OK;
  PC3D_REQUIRE(P && Bc && idx, "pc3d_gemm_nt_gather_f32: null pointer");
  PC3D_REQUIRE(!mask || K % 4 == 0, "pc3d_gemm_nt_gather_f32: the sign mask needs K %% 4 == 0 (K=%d)", K);
  return gemm_nt_launch(-1, P, ldp, W, bias, nullptr, 0, 0.f, nullptr, 0, B * S * ns, N, K, act, slope, Y, ldy, stream, 0, nullptr,
                        nullptr, idx, Bc, ns, S, NA, slope_in, mask, nullptr, nullptr, 0, 0.f, ymask);
}

extern "C" int pc3d_gemm_nt_poolbwd_f32(const float* Y, int64_t ldy_in, const float* g, const int32_t* arg, int B, int Npts,
                                        float slope_pool, const float* W, int N, int K, float* dX, int64_t ldx_out,
                                        void* stream) {
  PC3D_REQUIRE(B >= 0 && Npts >= 1 && K >= 4 && K % 4 == 0 && (int64_t)B * Npts <= 0x7fffffffLL && ldy_in % 4 == 0,
               "pc3d_gemm_nt_poolbwd_f32: bad sizes B=%d Npts=%d K=%d ldy=%lld (K, ldy multiples of 4)", B, Npts, K,
               (long long)ldy_in);
  if (B == 0) return PC3D_OK;
  PC3D_REQUIRE(Y && g && arg, "pc3d_gemm_nt_poolbwd_f32: null pointer");
  PC3D_REQUIRE(Npts % 128 == 0, "pc3d_gemm_nt_poolbwd_f32: Npts=%d must be a multiple of the 128-row tile", Npts);
  return gemm_nt_launch(-1, Y, ldy_in, W, nullptr, nullptr, 0, 0.f, nullptr, 0, B * Npts, N, K, 0, 0.f, dX, ldx_out, stream, 0,
                        nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0.f, nullptr, g, arg, Npts, slope_pool);
}
