// K12 — graph Laplacian of a point cloud for the AOF spectral front-end (attack/AOF/TAOF_attack.py:31-52):
// A_ij = exp(-|p_i - p_j|^2) on the symmetrised kNN graph, L = D - A, dense [B,N,N] (the eigensolver wants it dense).
// The reference materialises the [B,N,N,3] difference tensor (403 MB at B=32, N=1024), a dense A and a dense mask; here
// only the O(N k) graph edges are evaluated and scattered into a zero-filled L, then one pass fixes the diagonal.
#include "pc3d_common.h"

namespace pc3d {

struct LapArgs {
  PtsView x;
  const int32_t* idx;  // [B,N,K] (self may be included; self-loops cancel in D - A)
  int N, K;
  float* L;            // [B,N,N], zero-filled before edges_kernel
};

__global__ __launch_bounds__(256) void lap_edges_kernel(LapArgs a) {
  const int b = blockIdx.y;
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= a.N * a.K) return;
  const int i = e / a.K;
  const int j = a.idx[(int64_t)b * a.N * a.K + e];
  if (j == i) return;
  const float* pi = a.x.p + (int64_t)b * a.x.bs + (int64_t)i * a.x.ps;
  const float* pj = a.x.p + (int64_t)b * a.x.bs + (int64_t)j * a.x.ps;
  const float dx = pi[0] - pj[0], dy = pi[a.x.cs] - pj[a.x.cs], dz = pi[2 * a.x.cs] - pj[2 * a.x.cs];
  const float w = -expf(-((dx * dx + dy * dy) + dz * dz));
  float* Lb = a.L + (int64_t)b * a.N * a.N;
  // symmetrised mask: (i,j) and (j,i) both get the same value; concurrent writers write identical bits
  Lb[(int64_t)i * a.N + j] = w;
  Lb[(int64_t)j * a.N + i] = w;
}

// L_ii = -sum_{j != i} L_ij : one wave per row
__global__ __launch_bounds__(256) void lap_diag_kernel(float* L, int N) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (i >= N) return;
  float* row = L + ((int64_t)b * N + i) * N;
  float s = 0.f;
  for (int j = lane; j < N; j += 64)
    if (j != i) s += row[j];
  s = wave_sum(s);
  if (lane == 0) row[i] = -s;
}

}  // namespace pc3d

using namespace pc3d;

extern "C" int pc3d_graph_laplacian_f32(const float* xyz, int64_t x_bs, int64_t x_ps, int64_t x_cs,
                                        const int32_t* idx, int B, int N, int K, float* L, void* stream) {
  PC3D_REQUIRE(B >= 0 && N >= 1 && K >= 1, "pc3d_graph_laplacian_f32: bad sizes B=%d N=%d K=%d", B, N, K);
  PC3D_REQUIRE(B <= 65535, "pc3d_graph_laplacian_f32: B=%d exceeds grid.y limit", B);
  if (B == 0) return PC3D_OK;
  PC3D_REQUIRE(xyz && idx && L, "pc3d_graph_laplacian_f32: null pointer");
  hipStream_t st = as_stream(stream);
  hipError_t e = zero_async(L, (size_t)B * N * N, st);
  if (e != hipSuccess) {
    set_error("pc3d_graph_laplacian_f32: zero fill failed: %s", hipGetErrorString(e));
    return (int)e;
  }
  LapArgs a{{xyz, x_bs, x_ps, x_cs}, idx, N, K, L};
  hipLaunchKernelGGL(lap_edges_kernel, dim3(cdiv(N * K, 256), B), dim3(256), 0, st, a);
  PC3D_LAUNCH_CHECK("pc3d_graph_laplacian_f32/edges");
  hipLaunchKernelGGL(lap_diag_kernel, dim3(cdiv(N, 4), B), dim3(256), 0, st, L, N);
  PC3D_LAUNCH_CHECK("pc3d_graph_laplacian_f32/diag");
  return PC3D_OK;
}

// ---------------------------------------------------------------------------------------------------------
// K12b — AOF's per-iteration spectral re-projection (attack/AOF/TAOF_attack.py:114-126,164-170):
//   coeff = adv · V          [B,3,N] x [B,N,N]
//   lfc   = coeff[..., :lp] · V[..., :lp]^T ,  hfc = coeff[..., lp:] · V[..., lp:]^T
// three batched products with M = 3 rows: no matrix-pipe work to speak of (6 N^2 flop per product against 4 N^2 bytes
// of V) — the products are bound by streaming V from HBM. Both are written as ROW DOTS over a contiguous matrix:
//   out[c, r] = sum_k vec[c, k] * Mat[r, k]
// coeff from Mat = V^T (kept beside V: the basis is constant for a whole binary step), lfc / hfc TOGETHER from Mat = V
// (row n of V meets coeff[:, :lp] and coeff[:, lp:] in the same pass), so V and V^T are each read exactly once per
// iteration, in whole 16-byte lanes of whole rows. A wavefront owns a row at a time: lane l holds the elements
// k = 4 (l + 64 i) .. +3 of the three (six) vectors in registers for the whole launch (N <= 1024) and meets every row
// with one float4 load per 256 columns; the 64 partial sums are added by a fixed DPP / permlane tree — the result is a
// pure function of the inputs (no atomics, no split over workgroups).
// ---------------------------------------------------------------------------------------------------------
namespace pc3d {

struct RowdotArgs {
  const float* mat;   // [B, R, K] rows contiguous
  const float* vec;   // [B, 3, K]
  int R, K, split;    // columns k < split go to out_lo, the others to out_hi (out_hi null: one output, all columns)
  float* out_lo;      // [B, 3, R]
  float* out_hi;      // [B, 3, R] or null
};

// sum over the 64 lanes, the same value (and the same summation tree) in every lane; VALU only
__device__ __forceinline__ float wave_sum_dpp(float v) {
#define PC3D_ADD_DPP(ctrl)                                                                                                   \
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xf, 0xf, false))
  PC3D_ADD_DPP(0xb1);    // quad_perm [1,0,3,2]: l ^ 1
  PC3D_ADD_DPP(0x4e);    // quad_perm [2,3,0,1]: l ^ 2
  PC3D_ADD_DPP(0x141);   // row_half_mirror: the other quad of the 8 (quads are uniform by now)
  PC3D_ADD_DPP(0x140);   // row_mirror: the other half of the 16
#undef PC3D_ADD_DPP
  {
    const unsigned u = __builtin_bit_cast(unsigned, v);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);     // rows 0|1 and 2|3 meet
    v = __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
  }
  return sum_xor32(v);
}

constexpr int kRdRowsPerWave = 8;
constexpr int kRdWaves = 4;

// KI float4 per lane, row and column chunk of 256 KI; SPLIT: two outputs. K <= 256 KI: one chunk, a row's 64 partial sums
// are reduced as soon as the row is done. Larger K (CHUNKS): the wave walks the chunks with its rows' per-lane partial sums
// in registers and re-loads its slice of the vectors per chunk (L2 hits: 12 KB against 32 KB of matrix).
template <int KI, bool SPLIT, bool CHUNKS>
__global__ __launch_bounds__(64 * kRdWaves, 2) void rowdot3_kernel(RowdotArgs a) {
  constexpr int NV = SPLIT ? 6 : 3;
  const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r0 = (blockIdx.x * kRdWaves + wave) * kRdRowsPerWave;
  if (r0 >= a.R) return;
  const int K = a.K;
  const float* vec = a.vec + (int64_t)b * 3 * K;
  const float* mat = a.mat + ((int64_t)b * a.R + r0) * K;
  const int nrow = min(kRdRowsPerWave, a.R - r0);
  float keep[NV];          // lane j < 8 keeps row j's sums
#pragma unroll
  for (int q = 0; q < NV; ++q) keep[q] = 0.f;
  float part[CHUNKS ? kRdRowsPerWave : 1][NV];
#pragma unroll
  for (int j = 0; j < (CHUNKS ? kRdRowsPerWave : 1); ++j)
#pragma unroll
    for (int q = 0; q < NV; ++q) part[j][q] = 0.f;
  for (int k0 = 0; k0 < (CHUNKS ? K : 1); k0 += 256 * KI) {
    float4 v[NV][KI];
#pragma unroll
    for (int i = 0; i < KI; ++i) {
      const int k = k0 + 4 * (lane + 64 * i);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < K) x = *reinterpret_cast<const float4*>(vec + (int64_t)c * K + k);
        if (SPLIT) {
          const int s = a.split;
          v[c][i] = make_float4(k < s ? x.x : 0.f, k + 1 < s ? x.y : 0.f, k + 2 < s ? x.z : 0.f, k + 3 < s ? x.w : 0.f);
          v[3 + c][i] = make_float4(k < s ? 0.f : x.x, k + 1 < s ? 0.f : x.y, k + 2 < s ? 0.f : x.z, k + 3 < s ? 0.f : x.w);
        } else {
          v[c][i] = x;
        }
      }
    }
    float4 m[2][KI];
    auto load_row = [&](int j, float4* dst) {
#pragma unroll
      for (int i = 0; i < KI; ++i) {
        const int k = k0 + 4 * (lane + 64 * i);
        dst[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (j < nrow && k < K) {
          typedef float rd_f4 __attribute__((ext_vector_type(4)));
          const rd_f4 t = __builtin_nontemporal_load(reinterpret_cast<const rd_f4*>(mat + (int64_t)j * K + k));   // read once
          dst[i] = make_float4(t.x, t.y, t.z, t.w);
        }
      }
    };
    load_row(0, m[0]);
#pragma unroll
    for (int j = 0; j < kRdRowsPerWave; ++j) {
      if (j + 1 < kRdRowsPerWave) load_row(j + 1, m[(j + 1) & 1]);
#pragma unroll
      for (int q = 0; q < NV; ++q) {
        float t = CHUNKS ? part[j][q] : 0.f;
#pragma unroll
        for (int i = 0; i < KI; ++i) {
          const float4 x = m[j & 1][i], w = v[q][i];
          t = fmaf(x.x, w.x, t);
          t = fmaf(x.y, w.y, t);
          t = fmaf(x.z, w.z, t);
          t = fmaf(x.w, w.w, t);
        }
        if (CHUNKS) {
          part[j][q] = t;
        } else {
          const float sum = wave_sum_dpp(t);
          keep[q] = lane == j ? sum : keep[q];
        }
      }
    }
  }
  if (CHUNKS) {
#pragma unroll
    for (int j = 0; j < kRdRowsPerWave; ++j)
#pragma unroll
      for (int q = 0; q < NV; ++q) {
        const float sum = wave_sum_dpp(part[j][q]);
        keep[q] = lane == j ? sum : keep[q];
      }
  }
  if (lane < nrow) {
    const int64_t o = (int64_t)b * 3 * a.R + r0 + lane;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      a.out_lo[o + (int64_t)c * a.R] = keep[c];
      if (SPLIT) a.out_hi[o + (int64_t)c * a.R] = keep[3 + c];
    }
  }
}

// K % 4 != 0: the vectors stay in L2 (one wave per row, scalar loads)
__global__ __launch_bounds__(256) void rowdot3_generic_kernel(RowdotArgs a) {
  const int b = blockIdx.y, lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= a.R) return;
  const int K = a.K;
  const float* vec = a.vec + (int64_t)b * 3 * K;
  const float* row = a.mat + ((int64_t)b * a.R + r) * K;
  const int split = a.out_hi ? a.split : K;
  float lo[3] = {0.f, 0.f, 0.f}, hi[3] = {0.f, 0.f, 0.f};
  for (int k = lane; k < K; k += 64) {
    const float x = row[k];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float w = vec[(int64_t)c * K + k];
      if (k < split) lo[c] = fmaf(x, w, lo[c]);
      else hi[c] = fmaf(x, w, hi[c]);
    }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    lo[c] = wave_sum_dpp(lo[c]);
    hi[c] = wave_sum_dpp(hi[c]);
  }
  if (lane == 0) {
    const int64_t o = (int64_t)b * 3 * a.R + r;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      a.out_lo[o + (int64_t)c * a.R] = lo[c];
      if (a.out_hi) a.out_hi[o + (int64_t)c * a.R] = hi[c];
    }
  }
}

static int rowdot3(const char* nm, const float* mat, const float* vec, int B, int R, int K, int split, float* out_lo, float* out_hi,
                   void* stream) {
  PC3D_REQUIRE(B >= 0 && R >= 1 && K >= 1 && split >= 0 && split <= K, "%s: bad sizes B=%d R=%d K=%d split=%d", nm, B, R, K, split);
  PC3D_REQUIRE(B <= 65535, "%s: B=%d exceeds grid.y limit", nm, B);
  if (B == 0) return PC3D_OK;
  PC3D_REQUIRE(mat && vec && out_lo, "%s: null pointer", nm);
  RowdotArgs a{mat, vec, R, K, split, out_lo, out_hi};
  hipStream_t st = as_stream(stream);
  const bool fast = K % 4 == 0 && ((reinterpret_cast<uintptr_t>(mat) | reinterpret_cast<uintptr_t>(vec)) & 15) == 0;
  if (fast) {
    const dim3 grid(cdiv(R, kRdWaves * kRdRowsPerWave), B), block(64 * kRdWaves);
#define PC3D_RD(KI, CH)                                                                          \
  do {                                                                                           \
    if (out_hi) hipLaunchKernelGGL((rowdot3_kernel<KI, true, CH>), grid, block, 0, st, a);       \
    else hipLaunchKernelGGL((rowdot3_kernel<KI, false, CH>), grid, block, 0, st, a);             \
  } while (0)
    if (K <= 256) PC3D_RD(1, false);
    else if (K <= 512) PC3D_RD(2, false);
    else if (K <= 1024) PC3D_RD(4, false);
    else PC3D_RD(2, true);
#undef PC3D_RD
  } else {
    hipLaunchKernelGGL(rowdot3_generic_kernel, dim3(cdiv(R, 4), B), dim3(256), 0, st, a);
  }
  PC3D_LAUNCH_CHECK(nm);
  return PC3D_OK;
}

}  // namespace pc3d

extern "C" int pc3d_rowdot3_f32(const float* mat, const float* vec, int B, int R, int K, int split, float* out_lo, float* out_hi,
                                void* stream) {
  return rowdot3("pc3d_rowdot3_f32", mat, vec, B, R, K, split, out_lo, out_hi, stream);
}

extern "C" int pc3d_spectral_reproject_f32(const float* adv, const float* V, const float* Vt, int B, int N, int lp, float* coeff,
                                           float* lfc, float* hfc, void* stream) {
  PC3D_REQUIRE(lp >= 0 && lp <= N, "pc3d_spectral_reproject_f32: low_pass=%d outside [0, N=%d]", lp, N);
  PC3D_REQUIRE(coeff && lfc && hfc, "pc3d_spectral_reproject_f32: null output");
  int rc = rowdot3("pc3d_spectral_reproject_f32/coeff", Vt, adv, B, N, N, N, coeff, nullptr, stream);
  if (rc != PC3D_OK) return rc;
  return rowdot3("pc3d_spectral_reproject_f32/bands", V, coeff, B, N, N, lp, lfc, hfc, stream);
}
