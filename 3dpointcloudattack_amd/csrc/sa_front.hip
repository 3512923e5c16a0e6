// The coordinate part of a set-abstraction layer's FIRST 1x1 convolution, and the plain max over a group's rows.
//
// model/pointnet2_utils.py:118-135,190-197 concatenates [xyz_j - centre_s ; feat_j] per grouped row and runs the first
// Conv2d on that [B, 3+D, ns, S] tensor. The convolution is linear, so  W1 [x_j - c_s ; f_j] + b1 = P[j] + Bc[s]  with
//   P  = Wx x + Wf f   per POINT    ([B,N,C1]; the Wf part is a GEMM with P's coordinate part as its residual operand)
//   Bc = b1 - Wx c     per CENTRE   ([B,S,C1])
// The three-column products are these kernels (a GEMM tile with K = 3 would run empty); with them a layer's front needs
// neither a contiguous copy of the strided [B,N,3] view, nor a gather of P's rows at the centres, nor the subtraction and
// the negation / accumulation launches autograd adds behind them.
//   affine3_fwd : out[b,n,c] = bias[c] + sign * (W[c,0] x + W[c,1] y + W[c,2] z)          (x read through any strides)
//   affine3_bwd : out[b,n,:] = add[b,n,:] + sign * sum_c g[b,n,c] W[c,:]                  (written through any strides)
//   rows_max    : out[g,c] = max_r Y[g,r,c], arg = the first row that holds it              (:198, torch.max(new_points, 2)[0])
// All sums run in a fixed order: bit-reproducible, and a cloud's values do not depend on the batch.
#include "pc3d_common.h"

namespace pc3d {

__global__ __launch_bounds__(256) void affine3_fwd_kernel(PtsView x, int64_t rows, int N, const float* __restrict__ W,
                                                          const float* __restrict__ bias, float sign, int C4,
                                                          float* __restrict__ out, int64_t ldo) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * C4) return;
  const int64_t row = i / C4;
  const int cg = (int)(i - row * C4);
  const int64_t b = row / N, n = row - b * N;
  const float* xp = x.p + b * x.bs + n * x.ps;
  const float px = xp[0], py = xp[x.cs], pz = xp[2 * x.cs];
  const float4* w4 = reinterpret_cast<const float4*>(W + (int64_t)cg * 12);      // rows 4cg .. 4cg+3 of W [C,3]
  const float4 a = w4[0], bq = w4[1], c = w4[2];
  const float w[12] = {a.x, a.y, a.z, a.w, bq.x, bq.y, bq.z, bq.w, c.x, c.y, c.z, c.w};
  float r[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float t = w[3 * q] * px;
    t = __builtin_fmaf(w[3 * q + 1], py, t);
    t = __builtin_fmaf(w[3 * q + 2], pz, t);
    r[q] = sign * t;
  }
  if (bias) {
    const float4 bb = *reinterpret_cast<const float4*>(bias + 4 * cg);
    r[0] = bb.x + r[0], r[1] = bb.y + r[1], r[2] = bb.z + r[2], r[3] = bb.w + r[3];
  }
  *reinterpret_cast<float4*>(out + row * ldo + 4 * cg) = make_float4(r[0], r[1], r[2], r[3]);
}

// L lanes per row (a power of two <= 64): lane l sums the channel groups l, l + L, ... in ascending order, then a
// butterfly over the L lanes — one fixed order per (C, L).
template <int L>
__global__ __launch_bounds__(256) void affine3_bwd_kernel(const float* __restrict__ g, int64_t ldg, int64_t rows, int N, int C4,
                                                          const float* __restrict__ W, float sign, PtsView add, PtsViewMut out) {
  const int64_t row = ((int64_t)blockIdx.x * 256 + threadIdx.x) / L;
  const int l = threadIdx.x & (L - 1);
  const bool live = row < rows;
  float ax = 0.f, ay = 0.f, az = 0.f;
  if (live) {
    const float* gr = g + row * ldg;
    for (int cg = l; cg < C4; cg += L) {
      const float4 v = *reinterpret_cast<const float4*>(gr + 4 * cg);
      const float4* w4 = reinterpret_cast<const float4*>(W + (int64_t)cg * 12);
      const float4 a = w4[0], bq = w4[1], c = w4[2];
      ax = __builtin_fmaf(v.x, a.x, ax), ay = __builtin_fmaf(v.x, a.y, ay), az = __builtin_fmaf(v.x, a.z, az);
      ax = __builtin_fmaf(v.y, a.w, ax), ay = __builtin_fmaf(v.y, bq.x, ay), az = __builtin_fmaf(v.y, bq.y, az);
      ax = __builtin_fmaf(v.z, bq.z, ax), ay = __builtin_fmaf(v.z, bq.w, ay), az = __builtin_fmaf(v.z, c.x, az);
      ax = __builtin_fmaf(v.w, c.y, ax), ay = __builtin_fmaf(v.w, c.z, ay), az = __builtin_fmaf(v.w, c.w, az);
    }
  }
#pragma unroll
  for (int d = L >> 1; d >= 1; d >>= 1) {
    ax += __shfl_xor(ax, d, 64);
    ay += __shfl_xor(ay, d, 64);
    az += __shfl_xor(az, d, 64);
  }
  if (live && l == 0) {
    const int64_t b = row / N, n = row - b * N;
    float rx = sign * ax, ry = sign * ay, rz = sign * az;
    if (add.p) {
      const float* ap = add.p + b * add.bs + n * add.ps;
      rx = ap[0] + rx, ry = ap[add.cs] + ry, rz = ap[2 * add.cs] + rz;
    }
    float* op = out.p + b * out.bs + n * out.ps;
    op[0] = rx, op[out.cs] = ry, op[2 * out.cs] = rz;
  }
}

__global__ __launch_bounds__(256) void rows_max_kernel(const float* __restrict__ Y, int64_t G, int ns, int C,
                                                       float* __restrict__ out, int64_t* __restrict__ arg) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= G * C) return;
  const int64_t gidx = i / C;
  const int c = (int)(i - gidx * C);
  const float* y = Y + gidx * ns * C + c;
  float best = y[0];
  int bi = 0;
  for (int r = 1; r < ns; ++r) {
    const float v = y[(int64_t)r * C];
    if (v > best || (v != v && best == best)) best = v, bi = r;     // first maximum; a NaN wins and stays (torch.max)
  }
  out[i] = best;
  arg[i] = bi;
}

}  // namespace pc3d

using namespace pc3d;

extern "C" int pc3d_affine3_f32(const float* x, int64_t x_bs, int64_t x_ps, int64_t x_cs, int B, int N, const float* W,
                                const float* bias, float sign, int C, float* out, int64_t ldo, void* stream) {
  const char* nm = "pc3d_affine3_f32";
  PC3D_REQUIRE(B >= 0 && N >= 1 && C >= 4 && C % 4 == 0 && ldo >= C && ldo % 4 == 0, "%s: bad sizes B=%d N=%d C=%d (C %% 4 == 0)", nm, B, N, C);
  if (B == 0) return PC3D_OK;
  PC3D_REQUIRE(x && W && out, "%s: null pointer", nm);
  PC3D_REQUIRE(!(reinterpret_cast<uintptr_t>(W) & 15) && !(reinterpret_cast<uintptr_t>(out) & 15) &&
               !(reinterpret_cast<uintptr_t>(bias) & 15), "%s: W / bias / out must be 16-byte aligned", nm);
  const int64_t rows = (int64_t)B * N, total = rows * (C / 4);
  PC3D_REQUIRE((total + 255) / 256 <= 0x7fffffff, "%s: too many rows", nm);
  hipLaunchKernelGGL(affine3_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream),
                     PtsView{x, x_bs, x_ps, x_cs}, rows, N, W, bias, sign, C / 4, out, ldo);
  PC3D_LAUNCH_CHECK(nm);
  return PC3D_OK;
}

extern "C" int pc3d_affine3_bwd_f32(const float* g, int64_t ldg, int B, int N, int C, const float* W, float sign,
                                    const float* add, int64_t a_bs, int64_t a_ps, int64_t a_cs,
                                    float* out, int64_t o_bs, int64_t o_ps, int64_t o_cs, void* stream) {
  const char* nm = "pc3d_affine3_bwd_f32";
  PC3D_REQUIRE(B >= 0 && N >= 1 && C >= 4 && C % 4 == 0 && ldg >= C && ldg % 4 == 0, "%s: bad sizes B=%d N=%d C=%d (C %% 4 == 0)", nm, B, N, C);
  if (B == 0) return PC3D_OK;
  PC3D_REQUIRE(g && W && out, "%s: null pointer", nm);
  PC3D_REQUIRE(!(reinterpret_cast<uintptr_t>(W) & 15) && !(reinterpret_cast<uintptr_t>(g) & 15), "%s: g / W must be 16-byte aligned", nm);
  const int64_t rows = (int64_t)B * N;
  const int C4 = C / 4;
  int L = 1;
  while (L < 64 && 2 * L <= C4) L *= 2;
  const int64_t blocks = (rows * L + 255) / 256;
  PC3D_REQUIRE(blocks <= 0x7fffffff, "%s: too many rows", nm);
  const PtsView av{add, a_bs, a_ps, a_cs};
  const PtsViewMut ov{out, o_bs, o_ps, o_cs};
  hipStream_t st = as_stream(stream);
#define PC3D_A3B(LV) hipLaunchKernelGGL(affine3_bwd_kernel<LV>, dim3((unsigned)blocks), dim3(256), 0, st, g, ldg, rows, N, C4, W, sign, av, ov)
  switch (L) {
    case 64: PC3D_A3B(64); break;
    case 32: PC3D_A3B(32); break;
    case 16: PC3D_A3B(16); break;
    case 8: PC3D_A3B(8); break;
    case 4: PC3D_A3B(4); break;
    case 2: PC3D_A3B(2); break;
    default: PC3D_A3B(1); break;
  }
#undef PC3D_A3B
  PC3D_LAUNCH_CHECK(nm);
  return PC3D_OK;
}

extern "C" int pc3d_scatter_points_det_f32(const int32_t* idx, const float* val, int B, int S, int N, float* out, int64_t o_bs,
                                           int64_t o_ps, int64_t o_cs, void* stream) {
  // out[b, idx[b,s], :] += val[b,s,:] in ascending s (repeated centres included), out a [B,N,3] view with any strides
  return scatter_rows_det("pc3d_scatter_points_det_f32", idx, val, 3, nullptr, 0, 0.f, B, S, N, 3, out, o_ps, 1, 0, stream, nullptr,
                          o_bs, o_cs);
}

extern "C" int pc3d_rows_max_f32(const float* Y, int64_t G, int ns, int C, float* out, int64_t* arg, void* stream) {
  const char* nm = "pc3d_rows_max_f32";
  PC3D_REQUIRE(G >= 0 && ns >= 1 && C >= 1, "%s: bad sizes G=%lld ns=%d C=%d", nm, (long long)G, ns, C);
  if (G == 0) return PC3D_OK;
  PC3D_REQUIRE(Y && out && arg, "%s: null pointer", nm);
  const int64_t blocks = (G * C + 255) / 256;
  PC3D_REQUIRE(blocks <= 0x7fffffff, "%s: too many columns", nm);
  hipLaunchKernelGGL(rows_max_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), Y, G, ns, C, out, arg);
  PC3D_LAUNCH_CHECK(nm);
  return PC3D_OK;
}
