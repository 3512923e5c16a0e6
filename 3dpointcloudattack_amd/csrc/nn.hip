// K1 — fused nearest-neighbour (min + argmin of squared L2) between two point sets, never materialising
// the [B,N,M] distance matrix.  gfx950 / wave64.
//
// Replaces: attack/CW/CW_utils/distance.py:15-32,40-50,58-70; utils/dis_utils_torch.py:8-28;
//           utils/dis_utils_numpy.py:13-38; attack/GeoA3/knn_utils.py:10-20 (K=1).
//
// Structure: one 256-thread workgroup = (direction, batch b, tile of 64*Q query points).
//   * the reference cloud of batch b is staged through LDS as SoA x[],y[],z[] in tiles of <= 4096 points;
//   * all four waves hold the SAME 64*Q queries in registers (Q per lane) and each scans one quarter of the
//     LDS tile with broadcast ds_read_b128 (4 reference points per read) -> split-M across waves gives
//     4x more waves per query tile, which is what fills 256 CUs at N=1024;
//   * the running arg-min is kept per CHUNK of 8 reference points (6 VALU ops per pair for the distance + 0.5 for
//     v_min3 + 3/8 for the compare/select = 6.9 instead of 9 with a per-pair compare/select); the position inside
//     the winning chunk is recovered after the scan by recomputing its 8 distances;
//   * distance is the direct-difference form (dx*dx + dy*dy + dz*dz with FMA) — the |a|^2+|b|^2-2ab expansion
//     loses 1e-5 relative accuracy on near-coincident clouds;
//   * per-wave (min,argmin) are merged through LDS in ascending reference order so ties resolve to the
//     LOWEST index (torch.min / numpy argmin behaviour).
#include "pc3d_common.h"

namespace pc3d {

struct NNDir {
  PtsView q, r;
  int N, M;
  float* d;
  int32_t* i;
  int64_t* i64 = nullptr;   // the same indices as 64-bit integers (the pytorch3d-style API hands out int64), or null
};
struct NNArgs {
  NNDir dir[2];
};

constexpr int kNNThreads = 256;
constexpr int kNNWaves = kNNThreads / kWave;
constexpr int kNNMaxTile = 4096;           // reference points per LDS tile (48 KiB SoA)
constexpr float kFar = 1.0e18f;            // sentinel coordinate: (1e18)^2*3 < FLT_MAX, never the minimum

constexpr int kNNChunk = 8;                // reference points per arg-min bookkeeping step

// the one distance formula of this file (scan and index resolution must agree bit for bit)
__device__ __forceinline__ float nn_dist(float rx, float ry, float rz, float qx, float qy, float qz) {
  const float dx = rx - qx, dy = ry - qy, dz = rz - qz;
  float d = dx * dx;
  d = __builtin_fmaf(dy, dy, d);
  return __builtin_fmaf(dz, dz, d);
}

template <int Q, int WAVES>   // WAVES waves share the queries and split the staged reference tile (8 for small, launch-bound problems)
__global__ __launch_bounds__(WAVES * 64) void nn_kernel(NNArgs args, int mt_cap) {
  constexpr int kNNWaves = WAVES, kNNThreads = WAVES * 64;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const NNDir& D = args.dir[blockIdx.z];
  const int N = D.N, M = D.M;
  int bx_, b;
  xcd_swizzle(bx_, b);      // a cloud's workgroups on one XCD: its reference points are fetched into one L2, not eight
  const int q0 = bx_ * (kWave * Q);
  if (q0 >= N) return;  // grid.x is sized for the larger direction
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;

  // tile geometry (uniform): mt = points staged per pass, slice = points scanned per wave, multiple of the chunk
  const int mt = M < mt_cap ? M : mt_cap;
  const int slice = ((mt + kNNWaves * kNNChunk - 1) / (kNNWaves * kNNChunk)) * kNNChunk;
  const int mt_pad = slice * kNNWaves;
  float* sx = lds;
  float* sy = lds + mt_pad;
  float* sz = lds + 2 * mt_pad;

  float qx[Q], qy[Q], qz[Q], best[Q];
  int bidx[Q];
  const float* qb = D.q.p + (int64_t)b * D.q.bs;
#pragma unroll
  for (int k = 0; k < Q; ++k) {
    int qi = q0 + k * kWave + lane;
    if (qi >= N) qi = N - 1;  // clamp: duplicates a valid query, result discarded at the store
    const float* qp = qb + (int64_t)qi * D.q.ps;
    qx[k] = qp[0];
    qy[k] = qp[D.q.cs];
    qz[k] = qp[2 * D.q.cs];
    best[k] = __builtin_inff();
    bidx[k] = 0;
  }

  const float* rb = D.r.p + (int64_t)b * D.r.bs;
  for (int m0 = 0; m0 < M; m0 += mt) {
    __syncthreads();  // previous tile fully consumed
    for (int j = threadIdx.x; j < mt_pad; j += kNNThreads) {
      const int m = m0 + j;
      float x = kFar, y = kFar, z = kFar;
      if (j < mt && m < M) {
        const float* rp = rb + (int64_t)m * D.r.ps;
        x = rp[0];
        y = rp[D.r.cs];
        z = rp[2 * D.r.cs];
      }
      sx[j] = x;
      sy[j] = y;
      sz[j] = z;
    }
    __syncthreads();

    const int s0 = wave * slice;
    for (int j = s0; j < s0 + slice; j += kNNChunk) {
      const float4 rx0 = *reinterpret_cast<const float4*>(sx + j), rx1 = *reinterpret_cast<const float4*>(sx + j + 4);
      const float4 ry0 = *reinterpret_cast<const float4*>(sy + j), ry1 = *reinterpret_cast<const float4*>(sy + j + 4);
      const float4 rz0 = *reinterpret_cast<const float4*>(sz + j), rz1 = *reinterpret_cast<const float4*>(sz + j + 4);
      const float rxa[kNNChunk] = {rx0.x, rx0.y, rx0.z, rx0.w, rx1.x, rx1.y, rx1.z, rx1.w};
      const float rya[kNNChunk] = {ry0.x, ry0.y, ry0.z, ry0.w, ry1.x, ry1.y, ry1.z, ry1.w};
      const float rza[kNNChunk] = {rz0.x, rz0.y, rz0.z, rz0.w, rz1.x, rz1.y, rz1.z, rz1.w};
#pragma unroll
      for (int k = 0; k < Q; ++k) {
        float d[kNNChunk];
#pragma unroll
        for (int e = 0; e < kNNChunk; ++e) d[e] = nn_dist(rxa[e], rya[e], rza[e], qx[k], qy[k], qz[k]);
        // chunk minimum with v_min3 (0.5 op / pair); WHICH of the 8 it was is resolved once, after the scan
        float m = __builtin_fminf(__builtin_fminf(d[0], d[1]), d[2]);
        m = __builtin_fminf(__builtin_fminf(m, d[3]), d[4]);
        m = __builtin_fminf(__builtin_fminf(m, d[5]), d[6]);
        m = __builtin_fminf(m, d[7]);
        if (m < best[k]) {     // strict: the EARLIEST chunk holding the minimum wins
          best[k] = m;
          bidx[k] = m0 + j;
        }
      }
    }
  }

  // merge the four waves' candidates (ascending wave = ascending reference index inside a tile; across
  // tiles compare indices explicitly so the lowest index wins ties)
  // (the merge area sits BEHIND the staged tile: when the whole reference cloud fitted one tile it is still there,
  // and the arg-min below is resolved from LDS instead of eight dependent global loads per query)
  __syncthreads();
  float* cd = lds + 3 * mt_pad;                           // [kNNWaves][64*Q]
  int* ci = reinterpret_cast<int*>(cd + kNNWaves * kWave * Q);
  const bool resident = M <= mt;
#pragma unroll
  for (int k = 0; k < Q; ++k) {
    cd[wave * (kWave * Q) + k * kWave + lane] = best[k];
    ci[wave * (kWave * Q) + k * kWave + lane] = bidx[k];
  }
  __syncthreads();
  for (int t = threadIdx.x; t < kWave * Q; t += kNNThreads) {
    float bd = cd[t];
    int bi = ci[t];
#pragma unroll
    for (int w = 1; w < kNNWaves; ++w) {
      const float d = cd[w * (kWave * Q) + t];
      const int i = ci[w * (kWave * Q) + t];
      if (d < bd || (d == bd && i < bi)) {
        bd = d;
        bi = i;
      }
    }
    const int qi = q0 + t;
    if (qi < N) {
      if (D.d) D.d[(int64_t)b * N + qi] = bd;
      if (D.i || D.i64) {
        // bi is the first index of the winning chunk: the arg-min is the first of its 8 points whose distance,
        // recomputed with the same instructions, equals the minimum (ties -> lowest index, as torch.min)
        const float* qp = qb + (int64_t)qi * D.q.ps;
        const float x = qp[0], y = qp[D.q.cs], z = qp[2 * D.q.cs];
        int arg = bi;
        if (resident) {        // bi is a multiple of 8 inside the staged (padded with far sentinels) tile
#pragma unroll
          for (int e = kNNChunk - 1; e >= 0; --e)
            if (nn_dist(sx[bi + e], sy[bi + e], sz[bi + e], x, y, z) == bd) arg = bi + e;
        } else {
#pragma unroll
          for (int e = kNNChunk - 1; e >= 0; --e) {
            const int m = bi + e;
            if (m < M) {
              const float* rp = rb + (int64_t)m * D.r.ps;
              if (nn_dist(rp[0], rp[D.r.cs], rp[2 * D.r.cs], x, y, z) == bd) arg = m;
            }
          }
        }
        if (D.i) D.i[(int64_t)b * N + qi] = arg;
        if (D.i64) D.i64[(int64_t)b * N + qi] = arg;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// Bidirectional search with ONE distance evaluation per (a_i, b_j) pair feeding both directions.
//
// The two-scan form above evaluates every distance twice (once per direction): 2 x 6.9 VALU ops per pair. Here a
// workgroup holds 64*Q points of A in registers (Q per lane, the same in all four waves) and the waves share out a
// range of B staged in LDS, as before — but every distance d(a_i, b_j) now also feeds the COLUMN minimum of b_j:
//   * rows (a_i -> nearest b): per-chunk v_min3 + chunk bookkeeping exactly as in nn_kernel;
//   * columns (b_j -> nearest a): per lane the Q distances of b_j fold with v_min3 (0.5 op / pair); after 32 points of
//     B the 32 per-lane column vectors are reduced ACROSS the 64 lanes by a halving butterfly — at every level two
//     vectors exchange halves and merge into one (v_permlane32_swap / v_permlane16_swap for lane distances 32 / 16,
//     bank-masked DPP row_ror:8 and row_shl/shr:4 for 8 / 4, quad_perm for 2 / 1): 70 VALU ops per 32 points instead
//     of 6 x 32 for one full DPP reduction each, i.e. 2.2 / Q ops per pair. Lane L ends up with the minimum of point
//     j0 + (L >> 1) over this workgroup's 64*Q points of A.
// Per pair: 6 (distance) + 0.875 (row) + 0.5 + 2.2/Q (column) = 7.9 VALU ops at Q = 4, against 13.8.
// What one workgroup cannot finish alone is written as partials and folded by nn_shared_finalize_kernel:
//   * colpart[b][tileA][j]  — column minimum over one A tile (values only; the arg-min is recovered by re-evaluating
//     the winning tile's 64*Q distances with the same instructions, first exact match = lowest index);
//   * rowpart[b][split][i]  — (minimum, chunk base) when B is split over several workgroups to fill the chip.
// Distances are bit-identical to nn_kernel's: (b - a)^2 == (a - b)^2 in fp32 and the FMA chain has the same order.
// ---------------------------------------------------------------------------------------------------------
constexpr int kNNTree = 32;   // points of B per cross-lane reduction

struct NNSharedArgs {
  PtsView a, b;
  int N, M;
  int tilesA;        // ceil(N / (64*Q))
  int nsplit;        // workgroups sharing one A tile, each scanning per_split points of B
  int per_split;     // multiple of kNNWaves * kNNTree
  float* dA;         // [B,N]  final (nsplit == 1) — may be null
  int32_t* iA;       // [B,N]  final (nsplit == 1) — may be null
  float* colpart;    // [B,tilesA,M]
  int32_t* collane;  // [B,tilesA,M,2] 64-bit mask of the lanes that hold the column minimum — CI only
  float2* rowpart;   // [B,nsplit,N] (value, chunk base as int bits) when nsplit > 1
};

// ctrl / bank are immediates of the instruction: compile-time constants at every use
#define PC3D_DPP_MOV(old, src, ctrl, bank)                                                                     \
  __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, (old)), __builtin_bit_cast(int, (src)), \
                                                        (ctrl), 0xf, (bank), false))

typedef unsigned int nn_u32x2 __attribute__((ext_vector_type(2)));

// v_writelane_b32: hipcc 7.2 has no __builtin_amdgcn_writelane, but the LLVM intrinsic is reachable by its name. (An
// inline-asm v_writelane is NOT equivalent: the instruction needs a wait state after the VALU compare that wrote its
// SGPR operand, and the hazard recogniser does not look into asm — that version returned wrong masks.)
extern "C" __device__ int nn_writelane(int value, int lane, int old) __asm("llvm.amdgcn.writelane.i32");

// Column butterfly: c[0..31] hold, per lane, a partial minimum for 32 different points; returns for lane L the minimum
// over all 64 lanes of c[L >> 1]. (Lane algebra checked by emulation: tools/emul_nn_tree.py.)
__device__ __forceinline__ float nn_tree_min32(float (&c)[kNNTree], int lane) {
  // NB: __builtin_bit_cast(float, r.y) on the swizzle expression itself reads element 0 (hipcc 7.2: the v_min after
  // every swap vanished and columns were reduced over 16 lanes only) — copy the elements to scalars first.
#pragma unroll
  for (int p = 0; p < 16; ++p) {   // lane distance 32: lanes 32..63 of c[p] <-> lanes 0..31 of c[p+16]
    const nn_u32x2 r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, c[p]),
                                                        __builtin_bit_cast(unsigned, c[p + 16]), false, false);
    const unsigned ux = r[0], uy = r[1];
    c[p] = __builtin_fminf(__builtin_bit_cast(float, ux), __builtin_bit_cast(float, uy));
  }
#pragma unroll
  for (int p = 0; p < 8; ++p) {    // lane distance 16: odd rows of c[p] <-> even rows of c[p+8]
    const nn_u32x2 r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, c[p]),
                                                        __builtin_bit_cast(unsigned, c[p + 8]), false, false);
    const unsigned ux = r[0], uy = r[1];
    c[p] = __builtin_fminf(__builtin_bit_cast(float, ux), __builtin_bit_cast(float, uy));
  }
#pragma unroll
  for (int p = 0; p < 4; ++p) {    // lane distance 8: banks 0-1 keep c[p], banks 2-3 keep c[p+4]
    const float x = PC3D_DPP_MOV(c[p + 4], c[p], 0x128, 0x3);   // row_ror:8
    const float y = PC3D_DPP_MOV(c[p], c[p + 4], 0x128, 0xc);
    c[p] = __builtin_fminf(x, y);
  }
#pragma unroll
  for (int p = 0; p < 2; ++p) {    // lane distance 4: banks 0,2 keep c[p] (partner L+4), banks 1,3 keep c[p+2] (L-4)
    const float x = PC3D_DPP_MOV(c[p + 2], c[p], 0x104, 0x5);   // row_shl:4
    const float y = PC3D_DPP_MOV(c[p], c[p + 2], 0x114, 0xa);   // row_shr:4
    c[p] = __builtin_fminf(x, y);
  }
  const bool hi = (lane & 2) != 0; // lane distance 2
  const float keep = hi ? c[1] : c[0], give = hi ? c[0] : c[1];
  float r = __builtin_fminf(keep, PC3D_DPP_MOV(give, give, 0x4e, 0xf));     // quad_perm:[2,3,0,1]
  r = __builtin_fminf(r, PC3D_DPP_MOV(r, r, 0xb1, 0xf));                   // quad_perm:[1,0,3,2]
  return r;
}

template <int Q, bool CI>   // CI: also record WHERE each column minimum sits (needed for the b -> a indices)
__global__ __launch_bounds__(kNNThreads, 4) void nn_shared_kernel(NNSharedArgs A, int mt_cap) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int N = A.N, M = A.M;
  const int tile = blockIdx.x, b = blockIdx.y, split = blockIdx.z;
  const int a0 = tile * (kWave * Q);
  const int m_lo = split * A.per_split;
  if (m_lo >= M) return;                                   // uniform
  const int m_hi = (m_lo + A.per_split < M) ? m_lo + A.per_split : M;
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;

  float qx[Q], qy[Q], qz[Q], best[Q];
  int bidx[Q];
  const float* ab = A.a.p + (int64_t)b * A.a.bs;
#pragma unroll
  for (int k = 0; k < Q; ++k) {
    int qi = a0 + k * kWave + lane;
    if (qi >= N) qi = N - 1;  // clamp: a duplicate of a valid point changes neither direction's minimum
    const float* qp = ab + (int64_t)qi * A.a.ps;
    qx[k] = qp[0];
    qy[k] = qp[A.a.cs];
    qz[k] = qp[2 * A.a.cs];
    best[k] = __builtin_inff();
    bidx[k] = 0;
  }

  const float* bb = A.b.p + (int64_t)b * A.b.bs;
  float* cp = A.colpart + ((int64_t)b * A.tilesA + tile) * M;
  int32_t* cl = CI ? A.collane + ((int64_t)b * A.tilesA + tile) * M * 2 : nullptr;
  for (int m0 = m_lo; m0 < m_hi; m0 += mt_cap) {
    const int cnt = (m_hi - m0 < mt_cap) ? m_hi - m0 : mt_cap;
    const int slice = ((cnt + kNNWaves * kNNTree - 1) / (kNNWaves * kNNTree)) * kNNTree;
    const int mt_pad = slice * kNNWaves;
    float* sx = lds;
    float* sy = lds + mt_pad;
    float* sz = lds + 2 * mt_pad;
    __syncthreads();  // previous tile fully consumed
    for (int j = threadIdx.x; j < mt_pad; j += kNNThreads) {
      float x = kFar, y = kFar, z = kFar;
      if (j < cnt) {
        const float* rp = bb + (int64_t)(m0 + j) * A.b.ps;
        x = rp[0];
        y = rp[A.b.cs];
        z = rp[2 * A.b.cs];
      }
      sx[j] = x;
      sy[j] = y;
      sz[j] = z;
    }
    __syncthreads();

    const int s0 = wave * slice;
    for (int jc = s0; jc < s0 + slice; jc += kNNTree) {
      float c[kNNTree];
#pragma unroll
      for (int sub = 0; sub < kNNTree / kNNChunk; ++sub) {
        const int j = jc + sub * kNNChunk;
        const float4 rx0 = *reinterpret_cast<const float4*>(sx + j), rx1 = *reinterpret_cast<const float4*>(sx + j + 4);
        const float4 ry0 = *reinterpret_cast<const float4*>(sy + j), ry1 = *reinterpret_cast<const float4*>(sy + j + 4);
        const float4 rz0 = *reinterpret_cast<const float4*>(sz + j), rz1 = *reinterpret_cast<const float4*>(sz + j + 4);
        const float rxa[kNNChunk] = {rx0.x, rx0.y, rx0.z, rx0.w, rx1.x, rx1.y, rx1.z, rx1.w};
        const float rya[kNNChunk] = {ry0.x, ry0.y, ry0.z, ry0.w, ry1.x, ry1.y, ry1.z, ry1.w};
        const float rza[kNNChunk] = {rz0.x, rz0.y, rz0.z, rz0.w, rz1.x, rz1.y, rz1.z, rz1.w};
#pragma unroll
        for (int k = 0; k < Q; k += 2) {
          float d0[kNNChunk], d1[kNNChunk];
#pragma unroll
          for (int e = 0; e < kNNChunk; ++e) {
            d0[e] = nn_dist(rxa[e], rya[e], rza[e], qx[k], qy[k], qz[k]);
            if (k + 1 < Q) d1[e] = nn_dist(rxa[e], rya[e], rza[e], qx[k + 1], qy[k + 1], qz[k + 1]);
          }
          // columns: fold this lane's Q distances of every point (v_min3 over query pairs)
#pragma unroll
          for (int e = 0; e < kNNChunk; ++e) {
            float& ce = c[sub * kNNChunk + e];
            if (k == 0) ce = (Q > 1) ? __builtin_fminf(d0[e], d1[e]) : d0[e];
            else ce = (k + 1 < Q) ? __builtin_fminf(__builtin_fminf(ce, d0[e]), d1[e]) : __builtin_fminf(ce, d0[e]);
          }
          // rows: chunk minimum with v_min3; WHICH of the 8 it was is resolved once, after the scan
#pragma unroll
          for (int kk = 0; kk < 2; ++kk) {
            if (k + kk >= Q) break;
            const float* d = kk ? d1 : d0;
            float m = __builtin_fminf(__builtin_fminf(d[0], d[1]), d[2]);
            m = __builtin_fminf(__builtin_fminf(m, d[3]), d[4]);
            m = __builtin_fminf(__builtin_fminf(m, d[5]), d[6]);
            m = __builtin_fminf(m, d[7]);
            if (m < best[k + kk]) {   // strict: the EARLIEST chunk holding the minimum wins
              best[k + kk] = m;
              bidx[k + kk] = m0 + j;
            }
          }
          // program-order anchor: without it hipcc sinks ALL row work of the 32-point block behind the column work and
          // keeps its 128 distances alive (204 VGPRs; with it 106 and four waves per SIMD)
          asm volatile("" : "+v"(best[k]), "+v"(bidx[k]), "+v"(best[k + 1 < Q ? k + 1 : k]), "+v"(bidx[k + 1 < Q ? k + 1 : k]));
        }
      }
      float keep[kNNTree / 2];               // the butterfly overwrites c[0..15]; c[16..31] survive it
      if (CI) {
#pragma unroll
        for (int e = 0; e < kNNTree / 2; ++e) keep[e] = c[e];
      }
      const float r = nn_tree_min32(c, lane);
      const int m = m0 + jc + (lane >> 1);
      if ((lane & 1) == 0 && m < m_hi) cp[m] = r;
      if (CI) {
        // Which lanes hold each minimum: broadcast r_j (v_readlane), compare it with every lane's own folded value
        // (v_cmp -> a 64-bit mask in an SGPR pair) and park the mask in lanes 2j, 2j+1 of `win` (2 x v_writelane):
        // 4 VALU per point and NO scalar ALU work — the scalar unit is shared by the whole CU, and a version that
        // reduced the mask to (lowest lane, count) with s_ff1 / s_bcnt1 / shifts spent 6 SALU per point and cost
        // +27 % kernel time. The finalize kernel walks the mask (almost always one bit) and re-evaluates only those
        // lanes' Q points. Three passes so that no instruction waits on its neighbour's SGPR result.
        int win = 0;
        float rj[kNNTree];
#pragma unroll
        for (int e = 0; e < kNNTree; ++e)
          rj[e] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, r), 2 * e));
#pragma unroll
        for (int e = 0; e < kNNTree; ++e) {
          const unsigned long long msk = __builtin_amdgcn_ballot_w64((e < kNNTree / 2 ? keep[e] : c[e]) == rj[e]);
          win = nn_writelane((int)(unsigned)msk, 2 * e, win);
          win = nn_writelane((int)(unsigned)(msk >> 32), 2 * e + 1, win);
        }
        if (m < m_hi) cl[2 * (int64_t)(m0 + jc) + lane] = win;   // = cl[2 m + (lane & 1)]: mask word lo / hi of point m
      }
    }
  }

  // rows: merge the four waves' candidates (ascending wave = ascending index inside a tile; across tiles compare
  // indices explicitly so the lowest index wins ties)
  __syncthreads();
  float* cd = lds;                                        // [kNNWaves][64*Q]
  int* ci = reinterpret_cast<int*>(lds + kNNWaves * kWave * Q);
#pragma unroll
  for (int k = 0; k < Q; ++k) {
    cd[wave * (kWave * Q) + k * kWave + lane] = best[k];
    ci[wave * (kWave * Q) + k * kWave + lane] = bidx[k];
  }
  __syncthreads();
  for (int t = threadIdx.x; t < kWave * Q; t += kNNThreads) {
    float bd = cd[t];
    int bi = ci[t];
#pragma unroll
    for (int w = 1; w < kNNWaves; ++w) {
      const float d = cd[w * (kWave * Q) + t];
      const int i = ci[w * (kWave * Q) + t];
      if (d < bd || (d == bd && i < bi)) {
        bd = d;
        bi = i;
      }
    }
    const int qi = a0 + t;
    if (qi >= N) continue;
    if (A.nsplit > 1) {
      A.rowpart[((int64_t)b * A.nsplit + split) * N + qi] = make_float2(bd, __builtin_bit_cast(float, bi));
      continue;
    }
    if (A.dA) A.dA[(int64_t)b * N + qi] = bd;
    if (A.iA) {
      const float* qp = ab + (int64_t)qi * A.a.ps;
      const float x = qp[0], y = qp[A.a.cs], z = qp[2 * A.a.cs];
      int arg = bi;
#pragma unroll
      for (int e = kNNChunk - 1; e >= 0; --e) {
        const int m = bi + e;
        if (m < M) {
          const float* rp = bb + (int64_t)m * A.b.ps;
          if (nn_dist(rp[0], rp[A.b.cs], rp[2 * A.b.cs], x, y, z) == bd) arg = m;
        }
      }
      A.iA[(int64_t)b * N + qi] = arg;
    }
  }
}

// Folds the partials of nn_shared_kernel. blockIdx.z = 0: rows of A (only when B was split), 1: columns (points of B).
struct NNFinalArgs {
  PtsView a, b;
  int N, M, tilesA, tileA, nsplit;
  const float* colpart;
  const int32_t* collane;
  const float2* rowpart;
  float* dA;
  int32_t* iA;
  float* dB;
  int32_t* iB;
};

__global__ __launch_bounds__(256) void nn_shared_finalize_kernel(NNFinalArgs F) {
  const int b = blockIdx.y;
  const int t = blockIdx.x * 256 + threadIdx.x;
  const float* ab = F.a.p + (int64_t)b * F.a.bs;
  const float* bb = F.b.p + (int64_t)b * F.b.bs;
  if (blockIdx.z == 0) {
    if (F.nsplit <= 1 || t >= F.N) return;
    float bd = __builtin_inff();
    int bi = 0;
    for (int s0 = 0; s0 < F.nsplit; s0 += 8) {             // eight loads in flight, then the ordered compare
      float2 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        v[u] = (s0 + u < F.nsplit) ? F.rowpart[((int64_t)b * F.nsplit + s0 + u) * F.N + t] : make_float2(__builtin_inff(), 0.f);
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (v[u].x < bd) {                                   // ascending split = ascending index: strict < keeps the lowest
          bd = v[u].x;
          bi = __builtin_bit_cast(int, v[u].y);
        }
    }
    if (F.dA) F.dA[(int64_t)b * F.N + t] = bd;
    if (F.iA) {
      const float* qp = ab + (int64_t)t * F.a.ps;
      const float x = qp[0], y = qp[F.a.cs], z = qp[2 * F.a.cs];
      int arg = bi;
#pragma unroll
      for (int e = kNNChunk - 1; e >= 0; --e) {
        const int m = bi + e;
        if (m < F.M) {
          const float* rp = bb + (int64_t)m * F.b.ps;
          if (nn_dist(rp[0], rp[F.b.cs], rp[2 * F.b.cs], x, y, z) == bd) arg = m;
        }
      }
      F.iA[(int64_t)b * F.N + t] = arg;
    }
    return;
  }
  if (t >= F.M || (F.dB == nullptr && F.iB == nullptr)) return;
  float bd = __builtin_inff();
  int bt = 0;
  const float* cp = F.colpart + (int64_t)b * F.tilesA * F.M + t;
  for (int s0 = 0; s0 < F.tilesA; s0 += 8) {               // coalesced across t, eight loads in flight
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = (s0 + u < F.tilesA) ? cp[(int64_t)(s0 + u) * F.M] : __builtin_inff();
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (v[u] < bd) {                                     // ascending tile = ascending index: strict < keeps the lowest
        bd = v[u];
        bt = s0 + u;
      }
  }
  if (F.dB) F.dB[(int64_t)b * F.M + t] = bd;
  if (F.iB == nullptr) return;
  const int2 w2 = reinterpret_cast<const int2*>(F.collane)[((int64_t)b * F.tilesA + bt) * F.M + t];
  unsigned long long msk = ((unsigned long long)(unsigned)w2.y << 32) | (unsigned)w2.x;
  const float* rp = bb + (int64_t)t * F.b.ps;
  const float x = rp[0], y = rp[F.b.cs], z = rp[2 * F.b.cs];
  const int i0 = bt * F.tileA, rows = F.tileA / kWave;
  // The lanes in the mask hold the minimum among their Q points i0 + 64k + lane; the lowest matching index over all of
  // them is the answer (more than one bit only for duplicated points). Points past the end of the cloud were clamped
  // to the last one in the scan — that point is a candidate of this tile under its own index, so clamping here finds it.
  int arg = 0x7fffffff;
  while (msk) {
    const int l = __builtin_ctzll(msk);
    msk &= msk - 1;
    float d[4];
    int ii[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      int i = i0 + k * kWave + l;
      i = i < F.N ? i : F.N - 1;
      ii[k] = i;
      const float* qp = ab + (int64_t)i * F.a.ps;
      d[k] = nn_dist(x, y, z, qp[0], qp[F.a.cs], qp[2 * F.a.cs]);
    }
#pragma unroll
    for (int k = 3; k >= 0; --k)
      if (k < rows && d[k] == bd && ii[k] < arg) arg = ii[k];
  }
  if (arg == 0x7fffffff) arg = i0;                           // (NaN input: nothing compares equal)
  F.iB[(int64_t)b * F.M + t] = arg;
}

static size_t nn_lds_bytes(int M, int Q, int waves, int* mt_cap_out) {
  const int cap = kNNMaxTile;   // 1024 / 2048 / 4096 measured equal at B=32, N=4096
  const int mt = M < cap ? M : cap;
  const int slice = ((mt + waves * kNNChunk - 1) / (waves * kNNChunk)) * kNNChunk;
  const size_t tile = (size_t)3 * slice * waves * sizeof(float);
  const size_t merge = (size_t)waves * kWave * Q * 8;
  *mt_cap_out = cap;
  return tile + merge;
}

static int nn_launch(const NNArgs& a, int ndir, int B, hipStream_t st) {
  int maxN = a.dir[0].N, maxM = a.dir[0].M;
  if (ndir == 2) {
    if (a.dir[1].N > maxN) maxN = a.dir[1].N;
    if (a.dir[1].M > maxM) maxM = a.dir[1].M;
  }
  // Queries per lane: enough ILP to cover the LDS broadcast reads, but keep >= ~4 waves per SIMD's worth of
  // workgroups on 256 CUs (grid = tiles x B x ndir).
  const long q_total = (long)maxN * B * ndir;
  int Q = 4;
  if (q_total / (kWave * 4) < 1024) Q = 2;
  if (q_total / (kWave * 2) < 1024) Q = 1;
  // Small problems are bound by each workgroup's latency chain (stage the tile, scan it, merge), not by issue rate:
  // eight waves split the staged tile instead of four, halving the scan each wave walks.
  // (sixteen waves per workgroup measured no better: 16.4 against 14.4 us for both directions at B=32, N=1024)
  const int waves = (Q == 1 && maxM >= 512) ? 8 : 4;
  int mt_cap;
  const size_t lds = nn_lds_bytes(maxM, Q, waves, &mt_cap);
  dim3 grid(cdiv(maxN, kWave * Q), B, ndir), block(waves * 64);
  if (waves == 8) hipLaunchKernelGGL((nn_kernel<1, 8>), grid, block, lds, st, a, mt_cap);
  else switch (Q) {
    case 1: hipLaunchKernelGGL((nn_kernel<1, 4>), grid, block, lds, st, a, mt_cap); break;
    case 2: hipLaunchKernelGGL((nn_kernel<2, 4>), grid, block, lds, st, a, mt_cap); break;
    default: hipLaunchKernelGGL((nn_kernel<4, 4>), grid, block, lds, st, a, mt_cap); break;
  }
  PC3D_LAUNCH_CHECK("pc3d_nn");
  return PC3D_OK;
}

// Work split of the shared-evaluation search for a problem size (used by the launch and by the workspace query).
struct NNSharedPlan {
  int Q, tileA, tilesA, nsplit, per_split;
  size_t col_floats, lane_off_bytes, row_off_bytes, ws_bytes;
};

static NNSharedPlan nn_shared_plan(int B, int N, int M) {
  NNSharedPlan p;
  p.Q = N > 2 * kWave ? 4 : (N > kWave ? 2 : 1);
  p.tileA = kWave * p.Q;
  p.tilesA = cdiv(N, p.tileA);
  // Sixteen workgroups per CU's worth of grid, in slices of at least two butterflies of B per wave — one butterfly
  // where the grid would otherwise stay under four workgroups per CU. Measured inside replayed graphs at B=32
  // (tools/bench_nn_small.py with the rule varied; values only / with indices, us): target 2048 workgroups N=1024 12.3 / 14.7,
  // N=2048 27.4 / 31.9, N=4096 90.4 / 110.4; this rule 10.9 / 13.1, 26.5 / 31.0, 87.1 / 100.3. (Q = 2 instead of 4:
  // 12-20 % slower at every size.)
  const long wgs = (long)p.tilesA * (B > 0 ? B : 1);
  int ns = (int)((4096 + wgs - 1) / wgs);
  const int gran = kNNWaves * kNNTree;
  int max_ns = cdiv(M, 2 * gran);
  if (wgs * (ns < max_ns ? ns : max_ns) < 1024) max_ns = cdiv(M, gran);
  if (ns > max_ns) ns = max_ns;
  if (ns < 1) ns = 1;
  p.per_split = cdiv(cdiv(M, ns), gran) * gran;
  p.nsplit = cdiv(M, p.per_split);
  p.col_floats = (size_t)B * p.tilesA * M;
  p.lane_off_bytes = (p.col_floats * sizeof(float) + 15) & ~(size_t)15;
  p.row_off_bytes = p.lane_off_bytes + ((p.col_floats * 2 * sizeof(int32_t) + 15) & ~(size_t)15);
  p.ws_bytes = p.row_off_bytes + (p.nsplit > 1 ? (size_t)B * p.nsplit * N * sizeof(float2) : 0);
  return p;
}

static int nn_shared_launch(const PtsView& a, const PtsView& b, int B, int N, int M, float* dA, int32_t* iA, float* dB,
                            int32_t* iB, void* ws, hipStream_t st) {
  const NNSharedPlan p = nn_shared_plan(B, N, M);
  NNSharedArgs A{};
  A.a = a, A.b = b, A.N = N, A.M = M, A.tilesA = p.tilesA, A.nsplit = p.nsplit, A.per_split = p.per_split;
  A.dA = dA, A.iA = iA;
  A.colpart = reinterpret_cast<float*>(ws);
  A.collane = reinterpret_cast<int32_t*>(reinterpret_cast<char*>(ws) + p.lane_off_bytes);
  A.rowpart = reinterpret_cast<float2*>(reinterpret_cast<char*>(ws) + p.row_off_bytes);
  const int cap = kNNMaxTile / 2;                           // 24 KiB of LDS per workgroup: four of them fit a CU
  const int mt = p.per_split < cap ? p.per_split : cap;     // both multiples of kNNWaves * kNNTree
  const size_t tile = (size_t)3 * mt * sizeof(float);
  const size_t merge = (size_t)kNNWaves * kWave * p.Q * 8;
  const size_t lds = tile > merge ? tile : merge;
  dim3 grid(p.tilesA, B, p.nsplit), block(kNNThreads);
  const bool ci = iB != nullptr;
  switch (p.Q * 2 + (ci ? 1 : 0)) {
    case 2: hipLaunchKernelGGL((nn_shared_kernel<1, false>), grid, block, lds, st, A, cap); break;
    case 3: hipLaunchKernelGGL((nn_shared_kernel<1, true>), grid, block, lds, st, A, cap); break;
    case 4: hipLaunchKernelGGL((nn_shared_kernel<2, false>), grid, block, lds, st, A, cap); break;
    case 5: hipLaunchKernelGGL((nn_shared_kernel<2, true>), grid, block, lds, st, A, cap); break;
    case 8: hipLaunchKernelGGL((nn_shared_kernel<4, false>), grid, block, lds, st, A, cap); break;
    default: hipLaunchKernelGGL((nn_shared_kernel<4, true>), grid, block, lds, st, A, cap); break;
  }
  PC3D_LAUNCH_CHECK("pc3d_nn_bidir_shared_f32/scan");
  NNFinalArgs F{};
  F.a = a, F.b = b, F.N = N, F.M = M, F.tilesA = p.tilesA, F.tileA = p.tileA, F.nsplit = p.nsplit;
  F.colpart = A.colpart, F.collane = A.collane, F.rowpart = A.rowpart;
  F.dA = dA, F.iA = iA, F.dB = dB, F.iB = iB;
  // one launch folds the partials: z = 0 the rows of A (only when B was split), z = 1 the columns
  const bool rows = p.nsplit > 1 && (dA || iA);
  const bool cols = dB || iB;
  if (rows || cols) {
    const int nmax = (rows ? N : 0) > (cols ? M : 0) ? (rows ? N : 0) : (cols ? M : 0);
    hipLaunchKernelGGL(nn_shared_finalize_kernel, dim3(cdiv(nmax, 256), B, 2), dim3(256), 0, st, F);
  }
  PC3D_LAUNCH_CHECK("pc3d_nn_bidir_shared_f32/finalize");
  return PC3D_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Row reduction [B,N] -> [B]: one workgroup per row, fixed-order tree => bitwise reproducible.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rowreduce_kernel(const float* x, int N, int op, int pre, float* out) {
  __shared__ float part[4];
  const float* row = x + (int64_t)blockIdx.x * N;
  float acc = (op == 1) ? -__builtin_inff() : 0.f;
  for (int i = threadIdx.x; i < N; i += 256) {
    float v = row[i];
    if (pre == 1) v = __builtin_sqrtf(fmaxf(v, 0.f));
    acc = (op == 1) ? fmaxf(acc, v) : acc + v;
  }
  acc = (op == 1) ? wave_max(acc) : wave_sum(acc);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float r = part[0];
    for (int w = 1; w < 4; ++w) r = (op == 1) ? fmaxf(r, part[w]) : r + part[w];
    if (op == 0) r /= (float)N;
    out[blockIdx.x] = r;
  }
}

// ---------------------------------------------------------------------------------------------------------
// Backward through gathered NN pairs.
//   dA[b,i] = |a_i - b_{iA[i]}|^2 with upstream weight wA[b,i]   (scaled by sA)
//   dB[b,j] = |b_j - a_{iB[j]}|^2 with upstream weight wB[b,j]   (scaled by sB)
//   grad_a[i] = 2 sA wA[i] (a_i - b_{iA[i]})  +  sum_{j: iB[j]==i} 2 sB wB[j] (a_i - b_j)
//   grad_b[j] = 2 sB wB[j] (b_j - a_{iB[j]})  +  sum_{i: iA[i]==j} 2 sA wA[i] (b_j - a_i)
// Pass 1 writes the "own" term densely (overwrite); pass 2 adds the scattered term with float atomics, or —
// deterministic mode — every destination point scans the opposite index list in ascending order.
// ---------------------------------------------------------------------------------------------------------
struct BwdSide {
  PtsView self, other;   // self = the set whose points own the NN index list `idx` (own term)
  int n_self, n_other;
  const int32_t* idx;    // [B,n_self] -> index into other
  const float* w;        // upstream weight, element strides (w_bs, w_ps); may be NULL (=> zero)
  int64_t w_bs, w_ps;
  float scale;
  PtsViewMut g_self;     // gradient wrt self (may be null)
  PtsViewMut g_other;    // gradient wrt other (may be null)
};
struct BwdArgs {
  BwdSide s[2];
};

// own term: grid (ceil(n/256), B, 2)
__global__ __launch_bounds__(256) void nn_bwd_own_kernel(BwdArgs args) {
  const BwdSide& S = args.s[blockIdx.z];
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  if (i >= S.n_self || S.g_self.p == nullptr) return;
  float gx = 0.f, gy = 0.f, gz = 0.f;
  if (S.w != nullptr && S.idx != nullptr) {
    const float w = 2.f * S.scale * S.w[(int64_t)b * S.w_bs + (int64_t)i * S.w_ps];
    const int j = S.idx[(int64_t)b * S.n_self + i];
    const float* p = S.self.p + (int64_t)b * S.self.bs + (int64_t)i * S.self.ps;
    const float* o = S.other.p + (int64_t)b * S.other.bs + (int64_t)j * S.other.ps;
    gx = w * (p[0] - o[0]);
    gy = w * (p[S.self.cs] - o[S.other.cs]);
    gz = w * (p[2 * S.self.cs] - o[2 * S.other.cs]);
  }
  float* g = S.g_self.p + (int64_t)b * S.g_self.bs + (int64_t)i * S.g_self.ps;
  g[0] = gx;
  g[S.g_self.cs] = gy;
  g[2 * S.g_self.cs] = gz;
}

// scattered term, atomic flavour: thread per (b, i in self) adds into g_other[idx]
__global__ __launch_bounds__(256) void nn_bwd_scatter_atomic_kernel(BwdArgs args) {
  const BwdSide& S = args.s[blockIdx.z];
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  if (i >= S.n_self || S.g_other.p == nullptr || S.w == nullptr || S.idx == nullptr) return;
  const float w = 2.f * S.scale * S.w[(int64_t)b * S.w_bs + (int64_t)i * S.w_ps];
  if (w == 0.f) return;
  const int j = S.idx[(int64_t)b * S.n_self + i];
  const float* p = S.self.p + (int64_t)b * S.self.bs + (int64_t)i * S.self.ps;
  const float* o = S.other.p + (int64_t)b * S.other.bs + (int64_t)j * S.other.ps;
  float* g = S.g_other.p + (int64_t)b * S.g_other.bs + (int64_t)j * S.g_other.ps;
  atomicAdd(g, w * (o[0] - p[0]));
  atomicAdd(g + S.g_other.cs, w * (o[S.other.cs] - p[S.self.cs]));
  atomicAdd(g + 2 * S.g_other.cs, w * (o[2 * S.other.cs] - p[2 * S.self.cs]));
}

// scattered term, deterministic flavour: one workgroup per (b, 256 destinations); the index list of the
// opposite side is streamed through LDS and every destination accumulates its matches in ascending order.
__global__ __launch_bounds__(256) void nn_bwd_scatter_det_kernel(BwdArgs args) {
  __shared__ __attribute__((aligned(16))) int s_idx[1024];
  __shared__ float s_w[1024];
  const BwdSide& S = args.s[blockIdx.z];
  const int b = blockIdx.y;
  const int j = blockIdx.x * 256 + threadIdx.x;  // destination in `other`
  if (blockIdx.x * 256 >= S.n_other) return;
  if (S.g_other.p == nullptr || S.w == nullptr || S.idx == nullptr) return;
  const bool live = j < S.n_other;
  float ox = 0.f, oy = 0.f, oz = 0.f;
  if (live) {
    const float* o = S.other.p + (int64_t)b * S.other.bs + (int64_t)j * S.other.ps;
    ox = o[0];
    oy = o[S.other.cs];
    oz = o[2 * S.other.cs];
  }
  float gx = 0.f, gy = 0.f, gz = 0.f;
  for (int i0 = 0; i0 < S.n_self; i0 += 1024) {
    __syncthreads();
    for (int t = threadIdx.x; t < 1024; t += 256) {
      const int i = i0 + t;
      s_idx[t] = (i < S.n_self) ? S.idx[(int64_t)b * S.n_self + i] : -1;
      s_w[t] = (i < S.n_self) ? 2.f * S.scale * S.w[(int64_t)b * S.w_bs + (int64_t)i * S.w_ps] : 0.f;
    }
    __syncthreads();
    const int lim = (S.n_self - i0) < 1024 ? (S.n_self - i0) : 1024;
    // eight list entries per trip (two broadcast ds_read_b128): a matching entry is rare — one in n_other — so the
    // loop is bound by how fast the list streams past, and one entry per trip paid a full LDS latency each
    for (int t0 = 0; t0 < lim; t0 += 8) {          // (entries past lim hold -1: never equal to a live j)
      const int4 ia = *reinterpret_cast<const int4*>(s_idx + t0), ib = *reinterpret_cast<const int4*>(s_idx + t0 + 4);
      const int ii[8] = {ia.x, ia.y, ia.z, ia.w, ib.x, ib.y, ib.z, ib.w};
      if (!live || (ia.x != j && ia.y != j && ia.z != j && ia.w != j && ib.x != j && ib.y != j && ib.z != j && ib.w != j)) continue;
#pragma unroll
      for (int e = 0; e < 8; ++e) {                 // ascending list order: one fixed summation order
        if (ii[e] != j) continue;
        const float w = s_w[t0 + e];
        const float* p = S.self.p + (int64_t)b * S.self.bs + (int64_t)(i0 + t0 + e) * S.self.ps;
        gx += w * (ox - p[0]);
        gy += w * (oy - p[S.self.cs]);
        gz += w * (oz - p[2 * S.self.cs]);
      }
    }
  }
  if (live) {
    float* g = S.g_other.p + (int64_t)b * S.g_other.bs + (int64_t)j * S.g_other.ps;
    g[0] += gx;
    g[S.g_other.cs] += gy;
    g[2 * S.g_other.cs] += gz;
  }
}

}  // namespace pc3d

using namespace pc3d;

extern "C" int pc3d_nn_f32(const float* q, int64_t q_bs, int64_t q_ps, int64_t q_cs,
                           const float* r, int64_t r_bs, int64_t r_ps, int64_t r_cs,
                           int B, int N, int M, float* min_d2, int32_t* idx, void* stream) {
  PC3D_REQUIRE(B >= 0 && N >= 0 && M >= 1, "pc3d_nn_f32: bad sizes B=%d N=%d M=%d (M must be >= 1)", B, N, M);
  PC3D_REQUIRE(B <= 65535, "pc3d_nn_f32: B=%d exceeds grid.y limit 65535", B);
  if (B == 0 || N == 0) return PC3D_OK;
  PC3D_REQUIRE(q && r, "pc3d_nn_f32: null input pointer");
  NNArgs a{};
  a.dir[0] = NNDir{{q, q_bs, q_ps, q_cs}, {r, r_bs, r_ps, r_cs}, N, M, min_d2, idx};
  return nn_launch(a, 1, B, as_stream(stream));
}

extern "C" int pc3d_nn_i64_f32(const float* q, int64_t q_bs, int64_t q_ps, int64_t q_cs,
                               const float* r, int64_t r_bs, int64_t r_ps, int64_t r_cs,
                               int B, int N, int M, float* min_d2, int32_t* idx, int64_t* idx64, void* stream) {
  PC3D_REQUIRE(B >= 0 && N >= 0 && M >= 1, "pc3d_nn_i64_f32: bad sizes B=%d N=%d M=%d (M must be >= 1)", B, N, M);
  PC3D_REQUIRE(B <= 65535, "pc3d_nn_i64_f32: B=%d exceeds grid.y limit 65535", B);
  if (B == 0 || N == 0) return PC3D_OK;
  PC3D_REQUIRE(q && r, "pc3d_nn_i64_f32: null input pointer");
  NNArgs a{};
  a.dir[0] = NNDir{{q, q_bs, q_ps, q_cs}, {r, r_bs, r_ps, r_cs}, N, M, min_d2, idx, idx64};
  return nn_launch(a, 1, B, as_stream(stream));
}

extern "C" int pc3d_nn_bidir_f32(const float* a_, int64_t a_bs, int64_t a_ps, int64_t a_cs,
                                 const float* b_, int64_t b_bs, int64_t b_ps, int64_t b_cs,
                                 int B, int N, int M,
                                 float* dA, int32_t* iA, float* dB, int32_t* iB, void* stream) {
  PC3D_REQUIRE(B >= 0 && N >= 1 && M >= 1, "pc3d_nn_bidir_f32: bad sizes B=%d N=%d M=%d", B, N, M);
  PC3D_REQUIRE(B <= 65535, "pc3d_nn_bidir_f32: B=%d exceeds grid.y limit 65535", B);
  if (B == 0) return PC3D_OK;
  PC3D_REQUIRE(a_ && b_, "pc3d_nn_bidir_f32: null input pointer");
  NNArgs a{};
  a.dir[0] = NNDir{{a_, a_bs, a_ps, a_cs}, {b_, b_bs, b_ps, b_cs}, N, M, dA, iA};
  a.dir[1] = NNDir{{b_, b_bs, b_ps, b_cs}, {a_, a_bs, a_ps, a_cs}, M, N, dB, iB};
  return nn_launch(a, 2, B, as_stream(stream));
}

extern "C" int64_t pc3d_nn_bidir_shared_ws_bytes(int B, int N, int M) {
  if (B < 0 || N < 1 || M < 1) return -1;
  return (int64_t)nn_shared_plan(B, N, M).ws_bytes;
}

extern "C" int pc3d_nn_bidir_shared_f32(const float* a_, int64_t a_bs, int64_t a_ps, int64_t a_cs,
                                        const float* b_, int64_t b_bs, int64_t b_ps, int64_t b_cs,
                                        int B, int N, int M,
                                        float* dA, int32_t* iA, float* dB, int32_t* iB,
                                        void* ws, int64_t ws_bytes, void* stream) {
  PC3D_REQUIRE(B >= 0 && N >= 1 && M >= 1, "pc3d_nn_bidir_shared_f32: bad sizes B=%d N=%d M=%d", B, N, M);
  PC3D_REQUIRE(B <= 65535, "pc3d_nn_bidir_shared_f32: B=%d exceeds grid.y limit 65535", B);
  if (B == 0) return PC3D_OK;
  PC3D_REQUIRE(a_ && b_, "pc3d_nn_bidir_shared_f32: null input pointer");
  const NNSharedPlan p = nn_shared_plan(B, N, M);
  PC3D_REQUIRE(p.nsplit <= 65535, "pc3d_nn_bidir_shared_f32: M=%d needs %d splits (> grid.z limit)", M, p.nsplit);
  PC3D_REQUIRE(ws != nullptr && ws_bytes >= (int64_t)p.ws_bytes && (reinterpret_cast<uintptr_t>(ws) & 15) == 0,
               "pc3d_nn_bidir_shared_f32: workspace of %lld bytes (16-byte aligned) needed, got %lld",
               (long long)p.ws_bytes, (long long)ws_bytes);
  return nn_shared_launch(PtsView{a_, a_bs, a_ps, a_cs}, PtsView{b_, b_bs, b_ps, b_cs}, B, N, M, dA, iA, dB, iB, ws,
                          as_stream(stream));
}

extern "C" int pc3d_rowreduce_f32(const float* x, int B, int N, int op, int pre, float* out, void* stream) {
  PC3D_REQUIRE(B >= 0 && N >= 1, "pc3d_rowreduce_f32: bad sizes B=%d N=%d", B, N);
  PC3D_REQUIRE(op >= 0 && op <= 2 && (pre == 0 || pre == 1), "pc3d_rowreduce_f32: bad op=%d pre=%d", op, pre);
  if (B == 0) return PC3D_OK;
  PC3D_REQUIRE(x && out, "pc3d_rowreduce_f32: null pointer");
  hipLaunchKernelGGL(rowreduce_kernel, dim3(B), dim3(256), 0, as_stream(stream), x, N, op, pre, out);
  PC3D_LAUNCH_CHECK("pc3d_rowreduce_f32");
  return PC3D_OK;
}

extern "C" int pc3d_nn_bwd_f32(const float* a_, int64_t a_bs, int64_t a_ps, int64_t a_cs,
                               const float* b_, int64_t b_bs, int64_t b_ps, int64_t b_cs,
                               int B, int N, int M,
                               const int32_t* iA, const float* wA, int64_t wA_bs, int64_t wA_ps, float sA,
                               const int32_t* iB, const float* wB, int64_t wB_bs, int64_t wB_ps, float sB,
                               float* grad_a, int64_t ga_bs, int64_t ga_ps, int64_t ga_cs,
                               float* grad_b, int64_t gb_bs, int64_t gb_ps, int64_t gb_cs,
                               int deterministic, void* stream) {
  PC3D_REQUIRE(B >= 0 && N >= 1 && M >= 1, "pc3d_nn_bwd_f32: bad sizes B=%d N=%d M=%d", B, N, M);
  PC3D_REQUIRE(B <= 65535, "pc3d_nn_bwd_f32: B=%d exceeds grid.y limit 65535", B);
  if (B == 0) return PC3D_OK;
  PC3D_REQUIRE(a_ && b_, "pc3d_nn_bwd_f32: null input pointer");
  PC3D_REQUIRE((wA == nullptr) || (iA != nullptr), "pc3d_nn_bwd_f32: wA given without iA");
  PC3D_REQUIRE((wB == nullptr) || (iB != nullptr), "pc3d_nn_bwd_f32: wB given without iB");
  BwdArgs g{};
  PtsView A{a_, a_bs, a_ps, a_cs}, Bv{b_, b_bs, b_ps, b_cs};
  PtsViewMut GA{grad_a, ga_bs, ga_ps, ga_cs}, GB{grad_b, gb_bs, gb_ps, gb_cs};
  g.s[0] = BwdSide{A, Bv, N, M, iA, wA, wA_bs, wA_ps, sA, GA, GB};
  g.s[1] = BwdSide{Bv, A, M, N, iB, wB, wB_bs, wB_ps, sB, GB, GA};
  hipStream_t st = as_stream(stream);
  const int nmax = N > M ? N : M;
  hipLaunchKernelGGL(nn_bwd_own_kernel, dim3(cdiv(nmax, 256), B, 2), dim3(256), 0, st, g);
  PC3D_LAUNCH_CHECK("pc3d_nn_bwd_f32/own");
  const bool scatter_a = grad_a && wB, scatter_b = grad_b && wA;
  if (scatter_a || scatter_b) {
    if (deterministic)
      hipLaunchKernelGGL(nn_bwd_scatter_det_kernel, dim3(cdiv(nmax, 256), B, 2), dim3(256), 0, st, g);
    else
      hipLaunchKernelGGL(nn_bwd_scatter_atomic_kernel, dim3(cdiv(nmax, 256), B, 2), dim3(256), 0, st, g);
    PC3D_LAUNCH_CHECK("pc3d_nn_bwd_f32/scatter");
  }
  return PC3D_OK;
}
