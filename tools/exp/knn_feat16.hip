// EXPERIMENT, not part of libpc3d_hip.so (round 2): the accumulator-resident form of the feature-space kNN. Measured on
// MI355X at B=32, N=1024, C=64, K=20: 174 us against 141 us for knn_feat_kernel (dgcnn.hip) — products + staging 57 us,
// seed sorts 12, ballots 14, the ~59 insertions per query 96 (the same whether an insertion is 22 instructions, 16, or
// 5 VALU in inline assembly: with two waves per SIMD the chain latency of each insertion is not hidden). Kept as the
// record of that experiment; DESIGN.md section 3 has the numbers. Build: see the entry points at the bottom.

// K3b — feature-space kNN of DGCNN (model/dgcnn.py:194-200), second form: every wave OWNS 16 queries and scans all
// references, the distance tile never leaves the MFMA accumulator.
//
// knn_feat_kernel (dgcnn.hip) forms a 32 x 128 block of distances per workgroup in LDS and lets each wave scan 8 rows
// of it: two barriers, a strip write and a strip read per block, and (SQ counters, round 2) the MFMA pipe 26 % busy
// while the waves wait on each other. Here, for C in {16,32,64,128}:
//   * a workgroup = 4 waves = 64 queries of one cloud; references come through LDS in tiles (512 * 16 / C rows,
//     double-buffered, ONE barrier per tile), shared by the four waves;
//   * a wave multiplies its 16 query rows (registers) with 16 reference rows at a time on v_mfma_f32_16x16x4_f32:
//     D[row = query][col = reference], so lane (j, g) holds the distances of reference j to queries 4g..4g+3 —
//     one v_cmp against a per-lane threshold register + ballot gives, per accumulator register e, a 64-bit mask of
//     4 queries x 16 references; its 16-bit fields are walked with STATIC query numbers (4g+e), so the 16 K-lists
//     (across the lanes, knn_list.h) stay in registers;
//   * distances are ranked as order-preserving integers (one v_xor per value), so the threshold re-check and the
//     list position are scalar-ALU / integer compares;
//   * the products of the next 16 references are issued before the survivors of the current ones are inserted: the
//     matrix pipe works under the insertion chains of the same wave.
// Same arithmetic as knn_feat_kernel: d = -(2 q.r - |q|^2 - |r|^2) with the norms summed in the same partition (two
// interleaved float4 streams), so both kernels rank identically up to the fp32 order of the dot product itself.
#include "pc3d_common.h"
#include "knn_list.h"

namespace pc3d {

using kf_f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int KG_T = 256;
constexpr int KG_PF = 8;          // float4 per thread and tile in flight (tile rows = 512 * 16 / C)

struct KnnFeat16Args {
  const float* x;     // [B,N,C]
  const float* nrm;   // [B,N] squared norms (row_sqnorm_kernel)
  int N, K, tile;
  int mode;           // timing experiments only (pc3d_knn_feat_tune): 1 = no insertions, 2 = no ballots either, 4 = no seed sort
  int32_t* idx;       // [B,N,K]
};

// |row|^2 in the summation order of knn_feat_kernel: s_h = sum_t sq4(float4 at 8t + 4h), h = 0, 1; norm = s_0 + s_1
__global__ __launch_bounds__(256) void row_sqnorm_kernel(const float* __restrict__ x, int64_t M, int C,
                                                         float* __restrict__ out) {
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= M) return;
  const float* p = x + r * C;
  float s0 = 0.f, s1 = 0.f;
  for (int t = 0; t < C / 8; ++t) {
    const float4 a = *reinterpret_cast<const float4*>(p + 8 * t);
    const float4 b = *reinterpret_cast<const float4*>(p + 8 * t + 4);
    s0 += a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w;
    s1 += b.x * b.x + b.y * b.y + b.z * b.z + b.w * b.w;
  }
  out[r] = s0 + s1;
}

// insert (dc, ic) — wave-uniform — after every entry <= dc of the ascending list held across the 64 lanes; the last
// entry falls off. VALU only: the lane right after the last "keep" lane takes the candidate, the lanes above it their
// left neighbour — no popcount -> M0 -> v_writelane chain through the scalar unit (measured: the insertions, a chain of
// VALU -> SALU -> VALU hops of ~270 cycles each, were 60 % of the kernel). A candidate that no longer beats entry K-1
// lands behind it, where nothing is read: the caller needs no re-check against a freshly read threshold.
__device__ __forceinline__ void knn_list_insert_v(int& ld, int& li, int dc, int ic) {
  const bool k = ld <= dc;
  // k of the left neighbour (lane 0: "keep", so an all-smaller candidate lands in lane 0)
  const int kprev = __builtin_amdgcn_update_dpp(1, k ? 1 : 0, 0x138, 0xf, 0xf, false);   // wave_shr:1
  const int sd = __builtin_amdgcn_update_dpp(0, ld, 0x138, 0xf, 0xf, false);
  const int si = __builtin_amdgcn_update_dpp(0, li, 0x138, 0xf, 0xf, false);
  ld = k ? ld : (kprev ? dc : sd);
  li = k ? li : (kprev ? ic : si);
}

// The same insertion in 5 VALU instructions (the C++ form above compiles to 17, and the loop is VALU-issue bound): the
// lanes above the candidate are made the ONLY active lanes, so two in-place DPP moves shift the list's tail, and the
// candidate is written into the first of them with v_writelane (which ignores EXEC). Inline assembly because EXEC
// cannot be steered from HIP C++; the compiler does not see inside it, so the wait states the hardware does not
// interlock are written out (EXEC write -> DPP: 5; SALU write of M0 -> lane select: 1).
__device__ __forceinline__ void knn_list_insert_x(int& ld, int& li, int dc, int ic) {
  unsigned long long saved;
  asm volatile(
      "v_cmp_lt_i32 vcc, %3, %0\n\t"                     // entries above the candidate
      "s_and_saveexec_b64 %2, vcc\n\t"
      "s_cbranch_execz 1f\n\t"
      "s_ff1_i32_b64 m0, vcc\n\t"                        // the first of them takes the candidate
      "s_nop 4\n\t"
      "v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      "v_mov_b32_dpp %1, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_writelane_b32 %0, %3, m0\n\t"
      "v_writelane_b32 %1, %4, m0\n\t"
      "1:\n\t"
      "s_mov_b64 exec, %2\n\t"
      : "+v"(ld), "+v"(li), "=&s"(saved)
      : "s"(dc), "s"(ic)
      : "vcc", "scc", "m0");
}

template <int NU, int PAIR>   // C = 16 * NU; PAIR 16-reference tiles per step (two independent MFMA chains)
__global__ __launch_bounds__(KG_T) void knn_feat16_kernel(KnnFeat16Args a) {
  constexpr int C = 16 * NU, LDR = C + 4;
  extern __shared__ __attribute__((aligned(16))) float kg_lds[];
  const int tile = a.tile;                               // reference rows per tile (multiple of 64)
  float* rbuf = kg_lds;                                  // [2][tile][LDR]
  float* nbuf = kg_lds + 2 * tile * LDR;                 // [2][tile]
  const int b = blockIdx.y, q0 = blockIdx.x * 64;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, g = lane >> 4;
  const int N = a.N, K = a.K;
  const float* xb = a.x + (int64_t)b * N * C;
  const float* nb = a.nrm + (int64_t)b * N;

  // this wave's 16 queries as the MFMA A operand: lane (i = j, kq = g) holds q[i][16u + 4g .. +3]
  float4 qv[NU];
  {
    const int qrow = min(q0 + 16 * wave + j, N - 1);
#pragma unroll
    for (int u = 0; u < NU; ++u) qv[u] = *reinterpret_cast<const float4*>(xb + (int64_t)qrow * C + 16 * u + 4 * g);
  }
  float qn[4];   // |q|^2 of query 4g + e (the row of accumulator register e in this lane)
#pragma unroll
  for (int e = 0; e < 4; ++e) qn[e] = nb[min(q0 + 16 * wave + 4 * g + e, N - 1)];

  // global -> register -> LDS staging of one tile: float4 f = tid + 256 p  ->  row f / (4 NU), float4 column f % (4 NU)
  float4 pf[KG_PF];
  float pn[2];
  auto fetch = [&](int t) {
    const int base = t * tile;
#pragma unroll
    for (int p = 0; p < KG_PF; ++p) {
      const int f = tid + 256 * p;
      const int row = base + f / (4 * NU);
      pf[p] = row < N ? *reinterpret_cast<const float4*>(xb + (int64_t)row * C + 4 * (f % (4 * NU)))
                      : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int r = tid + 256 * p;
      pn[p] = (r < tile && base + r < N) ? nb[base + r] : 0.f;
    }
  };
  auto stash = [&](int which) {
    float* rb = rbuf + which * tile * LDR;
#pragma unroll
    for (int p = 0; p < KG_PF; ++p) {
      const int f = tid + 256 * p;
      *reinterpret_cast<float4*>(rb + (f / (4 * NU)) * LDR + 4 * (f % (4 * NU))) = pf[p];
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int r = tid + 256 * p;
      if (r < tile) nbuf[which * tile + r] = pn[p];
    }
  };

  // B operand of PAIR 16-reference tiles starting at 16-row block `sub` of LDS buffer rb: lane (col = j, kq = g)
  auto load_refs = [&](const float* rb, int sub, float4 (&rv)[PAIR][NU]) {
#pragma unroll
    for (int h = 0; h < PAIR; ++h) {
      const float* rows = rb + (16 * (sub + h) + j) * LDR + 4 * g;
#pragma unroll
      for (int u = 0; u < NU; ++u) rv[h][u] = *reinterpret_cast<const float4*>(rows + 16 * u);
    }
  };
  // D[row = query][col = reference] for the PAIR tiles: independent accumulator chains, interleaved
  auto products = [&](const float4 (&rv)[PAIR][NU], kf_f32x4 (&acc)[PAIR]) {
#pragma unroll
    for (int h = 0; h < PAIR; ++h) acc[h] = kf_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < NU; ++u) {
#pragma unroll
      for (int h = 0; h < PAIR; ++h) acc[h] = __builtin_amdgcn_mfma_f32_16x16x4f32(qv[u].x, rv[h][u].x, acc[h], 0, 0, 0);
#pragma unroll
      for (int h = 0; h < PAIR; ++h) acc[h] = __builtin_amdgcn_mfma_f32_16x16x4f32(qv[u].y, rv[h][u].y, acc[h], 0, 0, 0);
#pragma unroll
      for (int h = 0; h < PAIR; ++h) acc[h] = __builtin_amdgcn_mfma_f32_16x16x4f32(qv[u].z, rv[h][u].z, acc[h], 0, 0, 0);
#pragma unroll
      for (int h = 0; h < PAIR; ++h) acc[h] = __builtin_amdgcn_mfma_f32_16x16x4f32(qv[u].w, rv[h][u].w, acc[h], 0, 0, 0);
    }
  };
  // distances as order-preserving integers
  auto rank = [&](const kf_f32x4& acc, const float* nrow, int sub, int ref0, int (&di)[4]) {
    const float rn = nrow[16 * sub + j];
    const bool valid = ref0 + j < N;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      // model/dgcnn.py:195-197 ranks by 2 q.r - |q|^2 - |r|^2 (largest first); its exact negation, smallest first
      float d = -(2.f * acc[e] - qn[e] - rn);
      d = (valid && d == d) ? d : __builtin_inff();
      di[e] = knn_ord(d);
    }
  };

  int ld[16], li[16], thrv[4];
  int n_ins = 0, n_field = 0;   // mode 8: counted and added to the two ints behind the norms
#pragma unroll
  for (int q = 0; q < 16; ++q) ld[q] = 0x7fffffff, li[q] = 0x7fffffff;
#pragma unroll
  for (int e = 0; e < 4; ++e) thrv[e] = 0x7fffffff;

  const int ntile = (N + tile - 1) / tile;
  const int nsub = tile / 16;
  fetch(0);
  stash(0);
  __syncthreads();
  for (int t = 0; t < ntile; ++t) {
    const bool more = t + 1 < ntile;
    if (more) fetch(t + 1);
    const float* rb = rbuf + (t & 1) * tile * LDR;
    const float* nrow = nbuf + (t & 1) * tile;
    int s0 = 0;
    if (t == 0) {
      // ---- seed: the first 64 references, sorted. Query 4g'+e owns lanes 16g'..16g'+15 of register e of each of
      // the four 16-reference tiles; one ds_bpermute per tile brings them to lanes 16s + (lane & 15).
      int dsv[4][4];
#pragma unroll
      for (int s = 0; s < 4; s += PAIR) {
        float4 rv[PAIR][NU];
        kf_f32x4 acc[PAIR];
        load_refs(rb, s, rv);
        products(rv, acc);
#pragma unroll
        for (int h = 0; h < PAIR; ++h) rank(acc[h], nrow, s + h, 16 * (s + h), dsv[s + h]);
      }
#pragma unroll
      for (int gq = 0; gq < 4; ++gq)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          int v = 0x7fffffff;
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const int piece = __builtin_amdgcn_ds_bpermute(4 * (16 * gq + j), dsv[s][e]);
            v = (g == s) ? piece : v;
          }
          int si = lane;
          if (!(a.mode & 4)) wave_sort_pairs_dpp(v, si, lane);
          ld[4 * gq + e] = v, li[4 * gq + e] = si;
          const int t0 = __builtin_amdgcn_readlane(v, K - 1);
          thrv[e] = (g == gq) ? t0 : thrv[e];
        }
      s0 = 4;
    }
    if (s0 < nsub) {
      float4 rv[PAIR][NU];
      kf_f32x4 acc[PAIR];
      load_refs(rb, s0, rv);
      products(rv, acc);
      if (s0 + PAIR < nsub) load_refs(rb, s0 + PAIR, rv);     // operands one step ahead of their products
      for (int s = s0; s < nsub; s += PAIR) {
        int di[PAIR][4];
#pragma unroll
        for (int h = 0; h < PAIR; ++h) rank(acc[h], nrow, s + h, t * tile + 16 * (s + h), di[h]);
        if (s + PAIR < nsub) {
          products(rv, acc);                                   // on the matrix pipe during the insertions below
          if (s + 2 * PAIR < nsub) load_refs(rb, s + 2 * PAIR, rv);
        }
        if (a.mode & 2) {
          thrv[0] += di[0][0] & di[0][1] & di[0][2] & di[0][3] & di[PAIR - 1][0] & 1;   // keep the products alive
          continue;
        }
#pragma unroll
        for (int h = 0; h < PAIR; ++h) {
          const int ref0 = t * tile + 16 * (s + h);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const unsigned long long mask = __builtin_amdgcn_ballot_w64(di[h][e] < thrv[e]);
            if (mask == 0 || (a.mode & 1)) {
              thrv[e] += (int)(mask >> 63);   // (mode 1: keep the ballot alive; mask bit 63 is almost never set)
              continue;
            }
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
              unsigned m16 = (unsigned)(mask >> (16 * gq)) & 0xffffu;
              if (m16 == 0) continue;
              const int q = 4 * gq + e;
              ++n_field;
              do {
                ++n_ins;
                const int c = __builtin_ctz(m16);
                m16 &= m16 - 1;
                knn_list_insert_x(ld[q], li[q], __builtin_amdgcn_readlane(di[h][e], 16 * gq + c), ref0 + c);
              } while (m16);
              const int t1 = __builtin_amdgcn_readlane(ld[q], K - 1);
              thrv[e] = (g == gq) ? t1 : thrv[e];
            }
          }
        }
      }
    }
    if (more) stash((t + 1) & 1);
    __syncthreads();   // tile t+1 is complete, and everyone is done with tile t before it is overwritten at t+2
  }
  if ((a.mode & 8) && lane == 0) {
    int* ctr = reinterpret_cast<int*>(const_cast<float*>(a.nrm) + (int64_t)gridDim.y * N - (int64_t)b * N);
    atomicAdd(ctr, n_ins), atomicAdd(ctr + 1, n_field);
  }
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int qi = q0 + 16 * wave + q;
    if (qi < N && lane < K) a.idx[((int64_t)b * N + qi) * K + lane] = li[q];
  }
}

}  // namespace pc3d

using namespace pc3d;

static int g_knn_feat_mode = 0;
extern "C" int pc3d_knn_feat_tune(int mode) {   // timing experiments (tools/bench_knn_feat.py), not part of the ABI
  g_knn_feat_mode = mode;
  return 0;
}

extern "C" int64_t pc3d_knn_feat_ws_floats(int B, int N, int C) {
  return (B > 0 && N > 0 && (C == 16 || C == 32 || C == 64 || C == 128)) ? (int64_t)B * N + 2 : 0;
}

extern "C" int pc3d_knn_feat_ws_f32(const float* x, int B, int N, int C, int K, int32_t* idx, float* ws, void* stream) {
  if (!(C == 16 || C == 32 || C == 64 || C == 128)) return pc3d_knn_feat_f32(x, B, N, C, K, idx, stream);
  PC3D_REQUIRE(B >= 0 && N >= 1 && K >= 1 && K <= N && K <= 64, "pc3d_knn_feat_ws_f32: bad sizes B=%d N=%d K=%d (K <= 64)", B, N, K);
  PC3D_REQUIRE(B <= 65535, "pc3d_knn_feat_ws_f32: B=%d exceeds grid.y limit", B);
  if (B == 0) return PC3D_OK;
  PC3D_REQUIRE(x && idx && ws, "pc3d_knn_feat_ws_f32: null pointer");
  hipStream_t st = as_stream(stream);
  const int64_t M = (int64_t)B * N;
  hipLaunchKernelGGL(row_sqnorm_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, st, x, M, C, ws);
  PC3D_LAUNCH_CHECK("pc3d_knn_feat_ws_f32/norms");
  const int NU = C / 16;
  KnnFeat16Args a{x, ws, N, K, 512 / NU, g_knn_feat_mode, idx};
  const size_t lds = (size_t)(2 * a.tile * (C + 4) + 2 * a.tile) * sizeof(float);
  const dim3 grid(cdiv(N, 64), B), block(KG_T);
  // 67-82 KiB of dynamic LDS per workgroup (gfx950 has 160 KiB per CU): above the 64 KiB default limit of a launch
#define PC3D_KG_LAUNCH(NU_, PAIR_)                                                                                          \
  do {                                                                                                               \
    static bool raised = false;                                                                                      \
    if (!raised) {                                                                                                   \
      hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(&knn_feat16_kernel<NU_, PAIR_>),                    \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);                    \
      if (e_ != hipSuccess) {                                                                                        \
        set_error("pc3d_knn_feat_ws_f32: cannot raise the dynamic LDS limit: %s", hipGetErrorString(e_));            \
        return (int)e_;                                                                                              \
      }                                                                                                              \
      raised = true;                                                                                                 \
    }                                                                                                                \
    hipLaunchKernelGGL((knn_feat16_kernel<NU_, PAIR_>), grid, block, lds, st, a);                                             \
  } while (0)
  switch (NU) {
    case 1: PC3D_KG_LAUNCH(1, 2); break;
    case 2: PC3D_KG_LAUNCH(2, 2); break;
    case 4: PC3D_KG_LAUNCH(4, 2); break;
    default: PC3D_KG_LAUNCH(8, 1); break;
  }
#undef PC3D_KG_LAUNCH
  PC3D_LAUNCH_CHECK("pc3d_knn_feat_ws_f32");
  return PC3D_OK;
}
