// Point normals from the k-NN covariance (attack/GeoA3/utility.py:43-92 estimate_normal): per point, the k neighbours
// (self excluded) are centred, their 3 x 3 covariance is formed and the eigenvector of its smallest eigenvalue is
// the normal; its sign is fixed against the summed neighbour directions (:73-75).
//
// The reference (and round 1 of this library) ran a batched 3 x 3 `symeig` / `torch.linalg.eigh`: on ROCm that is
// rocsolver's general tridiagonalisation path, 5.0 ms per call at B=32, N=1024 (larf_left_kernel_small ...). A
// symmetric 3 x 3 problem has a closed form: eigenvalues from the trigonometric solution of the characteristic cubic,
// the eigenvector as the largest cross product of two rows of (A - lambda I). One thread per point, arithmetic in
// double (the covariance of 3 neighbours has rank 2: lambda_min / lambda_max ~ 1e-7 in fp32), microseconds per call.
#include "pc3d_common.h"

namespace pc3d {

struct NormalArgs {
  PtsView x;            // [B,N] points
  const int32_t* idx;   // [B,N,K1] neighbour lists, self first (K1 = k + 1)
  int N, K1;
  PtsViewMut out;       // [B,N] normals
};

__global__ __launch_bounds__(256) void estimate_normal_kernel(NormalArgs a) {
  const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  if (i >= a.N) return;
  const float* xb = a.x.p + (int64_t)b * a.x.bs;
  const int32_t* nb = a.idx + ((int64_t)b * a.N + i) * a.K1 + 1;   // skip self
  const int k = a.K1 - 1;
  // fp32 centring exactly as the reference does it (mean, subtract, sum): the sign test below is taken against the
  // SUM of the centred neighbours, which is rounding noise of this arithmetic (utility.py:73-75)
  float mx = 0.f, my = 0.f, mz = 0.f;
  for (int j = 0; j < k; ++j) {
    const float* p = xb + (int64_t)nb[j] * a.x.ps;
    mx += p[0], my += p[a.x.cs], mz += p[2 * a.x.cs];
  }
  mx /= (float)k, my /= (float)k, mz /= (float)k;
  double c00 = 0, c01 = 0, c02 = 0, c11 = 0, c12 = 0, c22 = 0;
  float sx = 0.f, sy = 0.f, sz = 0.f;
  for (int j = 0; j < k; ++j) {
    const float* p = xb + (int64_t)nb[j] * a.x.ps;
    const float dx = p[0] - mx, dy = p[a.x.cs] - my, dz = p[2 * a.x.cs] - mz;
    sx += dx, sy += dy, sz += dz;
    c00 += (double)dx * dx, c01 += (double)dx * dy, c02 += (double)dx * dz;
    c11 += (double)dy * dy, c12 += (double)dy * dz, c22 += (double)dz * dz;
  }
  const double f = 1.0 / (double)(k > 1 ? k - 1 : 1);
  c00 *= f, c01 *= f, c02 *= f, c11 *= f, c12 *= f, c22 *= f;
  // smallest eigenvalue (Smith 1961)
  const double q = (c00 + c11 + c22) / 3.0;
  const double p1 = c01 * c01 + c02 * c02 + c12 * c12;
  const double d0 = c00 - q, d1 = c11 - q, d2 = c22 - q;
  const double p2 = d0 * d0 + d1 * d1 + d2 * d2 + 2.0 * p1;
  double nx = 0.0, ny = 0.0, nz = 1.0;
  if (p2 > 0.0) {
    const double p = sqrt(p2 / 6.0), ip = 1.0 / p;
    const double b00 = d0 * ip, b11 = d1 * ip, b22 = d2 * ip, b01 = c01 * ip, b02 = c02 * ip, b12 = c12 * ip;
    double r = 0.5 * (b00 * (b11 * b22 - b12 * b12) - b01 * (b01 * b22 - b12 * b02) + b02 * (b01 * b12 - b11 * b02));
    r = r < -1.0 ? -1.0 : (r > 1.0 ? 1.0 : r);
    const double phi = acos(r) / 3.0;
    const double lmin = q + 2.0 * p * cos(phi + 2.0943951023931953);   // + 2 pi / 3
    // eigenvector: the largest cross product of two rows of A - lmin I
    const double r0x = c00 - lmin, r0y = c01, r0z = c02;
    const double r1x = c01, r1y = c11 - lmin, r1z = c12;
    const double r2x = c02, r2y = c12, r2z = c22 - lmin;
    const double ax = r0y * r1z - r0z * r1y, ay = r0z * r1x - r0x * r1z, az = r0x * r1y - r0y * r1x;
    const double bx = r0y * r2z - r0z * r2y, by = r0z * r2x - r0x * r2z, bz = r0x * r2y - r0y * r2x;
    const double cx = r1y * r2z - r1z * r2y, cy = r1z * r2x - r1x * r2z, cz = r1x * r2y - r1y * r2x;
    const double na = ax * ax + ay * ay + az * az, nbn = bx * bx + by * by + bz * bz, nc = cx * cx + cy * cy + cz * cz;
    double vx = ax, vy = ay, vz = az, nn = na;
    if (nbn > nn) vx = bx, vy = by, vz = bz, nn = nbn;
    if (nc > nn) vx = cx, vy = cy, vz = cz, nn = nc;
    if (nn > 0.0) {
      const double inv = 1.0 / sqrt(nn);
      nx = vx * inv, ny = vy * inv, nz = vz * inv;
    }
  }
  // sign = -sign(<n, sum of the centred neighbours>) (sign(0) = 0, as torch.sign)
  const double dotp = nx * (double)sx + ny * (double)sy + nz * (double)sz;
  const double sg = dotp > 0.0 ? -1.0 : (dotp < 0.0 ? 1.0 : 0.0);
  float* o = a.out.p + (int64_t)b * a.out.bs + (int64_t)i * a.out.ps;
  o[0] = (float)(sg * nx), o[a.out.cs] = (float)(sg * ny), o[2 * a.out.cs] = (float)(sg * nz);
}

// ---------------------------------------------------------------------------------------------------------
// Curvature proxy of GeoA3 (attack/GeoA3/loss_utils.py:60-90, _get_kappa_ori / _get_kappa_adv):
//   kappa_i = mean over the k neighbours j of | <(p_j - p_i) / max(|p_j - p_i|, 1e-12), n_i> |
// for neighbour lists idx [B,N,K1] whose FIRST entry is the point itself (dropped, :66 / :86), and its backward to the
// points (the normals are gathered constants). The reference builds [b,3,n,k] tensors for this (gather, permute,
// slice, subtract, norm, clamp, divide, multiply, sum, abs, mean: eleven launches forward, ~twenty backward, one of
// them the scatter of the gather); here one launch each way, a thread per point.
// ---------------------------------------------------------------------------------------------------------
struct KappaArgs {
  PtsView x, nrm;
  const int32_t* idx;   // [B,N,K1]
  int N, K1;
  float* out;           // [B,N]
  const float* gout;    // [B,N]   (backward)
  float* gx;            // [B,N,3] (backward; zero-filled by the entry point, float atomics)
  const int64_t* nidx = nullptr;   // [B,N] or null: point i takes normal nidx[b,i] of the M source normals (clamped)
  float* nout = nullptr;           // [B,3,N] the normals used (written when nidx is given)
  int M = 0;
};

__global__ __launch_bounds__(256) void kappa_fwd_kernel(KappaArgs a) {
  const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  if (i >= a.N) return;
  const float* xb = a.x.p + (int64_t)b * a.x.bs;
  const float* pi = xb + (int64_t)i * a.x.ps;
  int64_t ni_ = i;
  if (a.nidx) {
    ni_ = a.nidx[(int64_t)b * a.N + i];
    ni_ = ni_ < 0 ? 0 : (ni_ >= a.M ? a.M - 1 : ni_);
  }
  const float* ni = a.nrm.p + (int64_t)b * a.nrm.bs + ni_ * a.nrm.ps;
  const float px = pi[0], py = pi[a.x.cs], pz = pi[2 * a.x.cs];
  const float nx = ni[0], ny = ni[a.nrm.cs], nz = ni[2 * a.nrm.cs];
  if (a.nout) {
    float* no = a.nout + (int64_t)b * 3 * a.N + i;
    no[0] = nx, no[a.N] = ny, no[2 * (int64_t)a.N] = nz;
  }
  const int32_t* nb = a.idx + ((int64_t)b * a.N + i) * a.K1;
  float s = 0.f;
  for (int k = 1; k < a.K1; ++k) {
    const int j = min(max(nb[k], 0), a.N - 1);
    const float* pj = xb + (int64_t)j * a.x.ps;
    const float dx = pj[0] - px, dy = pj[a.x.cs] - py, dz = pj[2 * a.x.cs] - pz;
    const float len = fmaxf(sqrtf(dx * dx + dy * dy + dz * dz), 1e-12f);
    s += fabsf((dx / len) * nx + (dy / len) * ny + (dz / len) * nz);
  }
  a.out[(int64_t)b * a.N + i] = s / (float)(a.K1 - 1);
}

// d kappa_i / d d_ij = sign(<v,n>) (n - v <v,n>) / |d| / k  for |d| above the clamp (below it: sign n / 1e-12 / k, the
// norm's own gradient being cut by the clamp); +g to p_j, -g to p_i.
__global__ __launch_bounds__(256) void kappa_bwd_kernel(KappaArgs a) {
  const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  if (i >= a.N) return;
  const float* xb = a.x.p + (int64_t)b * a.x.bs;
  const float* pi = xb + (int64_t)i * a.x.ps;
  const float* ni = a.nrm.p + (int64_t)b * a.nrm.bs + (int64_t)i * a.nrm.ps;
  const float px = pi[0], py = pi[a.x.cs], pz = pi[2 * a.x.cs];
  const float nx = ni[0], ny = ni[a.nrm.cs], nz = ni[2 * a.nrm.cs];
  const int32_t* nb = a.idx + ((int64_t)b * a.N + i) * a.K1;
  const float g = a.gout[(int64_t)b * a.N + i] / (float)(a.K1 - 1);
  float* gb = a.gx + (int64_t)b * a.N * 3;
  float sx = 0.f, sy = 0.f, sz = 0.f;
  for (int k = 1; k < a.K1; ++k) {
    const int j = min(max(nb[k], 0), a.N - 1);
    const float* pj = xb + (int64_t)j * a.x.ps;
    const float dx = pj[0] - px, dy = pj[a.x.cs] - py, dz = pj[2 * a.x.cs] - pz;
    const float nrm = sqrtf(dx * dx + dy * dy + dz * dz);
    const bool clamped = !(nrm > 1e-12f);
    const float len = clamped ? 1e-12f : nrm;
    const float vx = dx / len, vy = dy / len, vz = dz / len;
    const float dot = vx * nx + vy * ny + vz * nz;
    const float sg = dot > 0.f ? g : (dot < 0.f ? -g : 0.f);
    const float c = clamped ? 0.f : dot;                    // the -v <v,n> term comes from d|d|/dd, which the clamp cuts
    const float gxk = sg * (nx - vx * c) / len, gyk = sg * (ny - vy * c) / len, gzk = sg * (nz - vz * c) / len;
    sx += gxk, sy += gyk, sz += gzk;
    atomicAdd(gb + 3 * j, gxk), atomicAdd(gb + 3 * j + 1, gyk), atomicAdd(gb + 3 * j + 2, gzk);
  }
  atomicAdd(gb + 3 * i, -sx), atomicAdd(gb + 3 * i + 1, -sy), atomicAdd(gb + 3 * i + 2, -sz);
}

// The same backward without global atomics: workgroup (coordinate d, cloud b) keeps the cloud's points and ONE component
// of its gradient in LDS, a lane walks the neighbours of its point, adds component d of every edge's gradient to the
// neighbour's slot (ds_add_f32) and keeps its own point's sum in a register. The three workgroups of a cloud repeat the
// edge arithmetic (cheap) so that each issues a third of the LDS atomics, which are what bounds this form (two earlier
// one-workgroup-per-cloud variants with all three components took 70 and 150 us against 72 for the kernel above at B=32,
// N=1024, k=16: LDS float atomics retire about one lane per 1.5-3 clocks). N <= kKappaSlabMaxN (16 N bytes of LDS).
constexpr int kKappaSlabMaxN = 4096;
__global__ __launch_bounds__(1024) void kappa_bwd_slab_kernel(KappaArgs a) {
  extern __shared__ float ks_lds[];
  float* xs = ks_lds;                 // [3][N]
  float* acc = ks_lds + 3 * a.N;      // [N]
  const int d = blockIdx.x, b = blockIdx.y;
  const float* xb = a.x.p + (int64_t)b * a.x.bs;
  for (int t = threadIdx.x; t < 3 * a.N; t += 1024) {
    const int c = t / a.N, i = t - c * a.N;
    xs[t] = xb[(int64_t)i * a.x.ps + (int64_t)c * a.x.cs];
  }
  for (int i = threadIdx.x; i < a.N; i += 1024) acc[i] = 0.f;
  __syncthreads();
  const float inv_k = 1.f / (float)(a.K1 - 1);
  for (int i = threadIdx.x; i < a.N; i += 1024) {
    const float* ni = a.nrm.p + (int64_t)b * a.nrm.bs + (int64_t)i * a.nrm.ps;
    const float nx = ni[0], ny = ni[a.nrm.cs], nz = ni[2 * a.nrm.cs];
    const float nd = d == 0 ? nx : (d == 1 ? ny : nz);
    const float px = xs[i], py = xs[a.N + i], pz = xs[2 * a.N + i];
    const int32_t* nb = a.idx + ((int64_t)b * a.N + i) * a.K1;
    const float g = a.gout[(int64_t)b * a.N + i] * inv_k;
    float own = 0.f;
    for (int k = 1; k < a.K1; ++k) {
      const int j = min(max(nb[k], 0), a.N - 1);
      const float dx = xs[j] - px, dy = xs[a.N + j] - py, dz = xs[2 * a.N + j] - pz;
      const float nrm = sqrtf(dx * dx + dy * dy + dz * dz);
      const bool clamped = !(nrm > 1e-12f);
      const float len = clamped ? 1e-12f : nrm;
      const float vx = dx / len, vy = dy / len, vz = dz / len;
      const float dot = vx * nx + vy * ny + vz * nz;
      const float sg = dot > 0.f ? g : (dot < 0.f ? -g : 0.f);
      const float c = clamped ? 0.f : dot;
      const float vd = d == 0 ? vx : (d == 1 ? vy : vz);
      const float gk = sg * (nd - vd * c) / len;
      own += gk;
      atomicAdd(acc + j, gk);
    }
    atomicAdd(acc + i, -own);
  }
  __syncthreads();
  float* gb = a.gx + (int64_t)b * a.N * 3 + d;
  for (int i = threadIdx.x; i < a.N; i += 1024) gb[3 * i] = acc[i];
}

// The slab form made DETERMINISTIC: every wavefront of the workgroup accumulates into its OWN slab (ds_add_f32 of one
// wave execute in issue order, same-address lanes of one instruction in a fixed lane order), wave w walks the contiguous
// range of points [w N / W, (w+1) N / W), and the W slabs are combined in ascending w. LDS: (3 + W) N floats.
template <int W>
__global__ __launch_bounds__(64 * W) void kappa_bwd_det_kernel(KappaArgs a) {
  extern __shared__ float kd_lds[];
  float* xs = kd_lds;                 // [3][N]
  float* slabs = kd_lds + 3 * a.N;    // [W][N]
  const int d = blockIdx.x, b = blockIdx.y;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const float* xb = a.x.p + (int64_t)b * a.x.bs;
  for (int t = threadIdx.x; t < 3 * a.N; t += 64 * W) {
    const int c = t / a.N, i = t - c * a.N;
    xs[t] = xb[(int64_t)i * a.x.ps + (int64_t)c * a.x.cs];
  }
  for (int t = threadIdx.x; t < W * a.N; t += 64 * W) slabs[t] = 0.f;
  __syncthreads();
  float* acc = slabs + wave * a.N;
  const int per = (a.N + W - 1) / W, lo = wave * per, hi = min(lo + per, a.N);
  const float inv_k = 1.f / (float)(a.K1 - 1);
  for (int i = lo + lane; i < hi; i += 64) {
    const float* ni = a.nrm.p + (int64_t)b * a.nrm.bs + (int64_t)i * a.nrm.ps;
    const float nx = ni[0], ny = ni[a.nrm.cs], nz = ni[2 * a.nrm.cs];
    const float nd = d == 0 ? nx : (d == 1 ? ny : nz);
    const float px = xs[i], py = xs[a.N + i], pz = xs[2 * a.N + i];
    const int32_t* nb = a.idx + ((int64_t)b * a.N + i) * a.K1;
    const float g = a.gout[(int64_t)b * a.N + i] * inv_k;
    float own = 0.f;
    for (int k = 1; k < a.K1; ++k) {
      const int j = min(max(nb[k], 0), a.N - 1);
      const float dx = xs[j] - px, dy = xs[a.N + j] - py, dz = xs[2 * a.N + j] - pz;
      const float nrm = sqrtf(dx * dx + dy * dy + dz * dz);
      const bool clamped = !(nrm > 1e-12f);
      const float len = clamped ? 1e-12f : nrm;
      const float vx = dx / len, vy = dy / len, vz = dz / len;
      const float dot = vx * nx + vy * ny + vz * nz;
      const float sg = dot > 0.f ? g : (dot < 0.f ? -g : 0.f);
      const float c = clamped ? 0.f : dot;
      const float vd = d == 0 ? vx : (d == 1 ? vy : vz);
      const float gk = sg * (nd - vd * c) / len;
      own += gk;
      atomicAdd(acc + j, gk);          // wave-private slab
    }
    atomicAdd(acc + i, -own);
  }
  __syncthreads();
  float* gb = a.gx + (int64_t)b * a.N * 3 + d;
  for (int i = threadIdx.x; i < a.N; i += 64 * W) {
    float s = slabs[i];
#pragma unroll
    for (int w = 1; w < W; ++w) s += slabs[w * a.N + i];
    gb[3 * i] = s;
  }
}

}  // namespace pc3d

using namespace pc3d;

extern "C" int pc3d_estimate_normal_f32(const float* x, int64_t x_bs, int64_t x_ps, int64_t x_cs, const int32_t* idx,
                                        int B, int N, int K1, float* out, int64_t o_bs, int64_t o_ps, int64_t o_cs,
                                        void* stream) {
  PC3D_REQUIRE(B >= 0 && N >= 1 && K1 >= 2, "pc3d_estimate_normal_f32: bad sizes B=%d N=%d K1=%d (self + >= 1 neighbour)", B, N, K1);
  PC3D_REQUIRE(B <= 65535, "pc3d_estimate_normal_f32: B=%d exceeds grid.y limit", B);
  if (B == 0) return PC3D_OK;
  PC3D_REQUIRE(x && idx && out, "pc3d_estimate_normal_f32: null pointer");
  NormalArgs a{{x, x_bs, x_ps, x_cs}, idx, N, K1, {out, o_bs, o_ps, o_cs}};
  hipLaunchKernelGGL(estimate_normal_kernel, dim3(cdiv(N, 256), B), dim3(256), 0, as_stream(stream), a);
  PC3D_LAUNCH_CHECK("pc3d_estimate_normal_f32");
  return PC3D_OK;
}

extern "C" int pc3d_kappa_f32(const float* x, int64_t x_bs, int64_t x_ps, int64_t x_cs, const float* nrm, int64_t n_bs,
                              int64_t n_ps, int64_t n_cs, const int32_t* idx, int B, int N, int K1, float* out,
                              void* stream) {
  PC3D_REQUIRE(B >= 0 && N >= 1 && K1 >= 2, "pc3d_kappa_f32: bad sizes B=%d N=%d K1=%d (self + >= 1 neighbour)", B, N, K1);
  PC3D_REQUIRE(B <= 65535, "pc3d_kappa_f32: B=%d exceeds grid.y limit", B);
  if (B == 0) return PC3D_OK;
  PC3D_REQUIRE(x && nrm && idx && out, "pc3d_kappa_f32: null pointer");
  KappaArgs a{{x, x_bs, x_ps, x_cs}, {nrm, n_bs, n_ps, n_cs}, idx, N, K1, out, nullptr, nullptr};
  hipLaunchKernelGGL(kappa_fwd_kernel, dim3(cdiv(N, 256), B), dim3(256), 0, as_stream(stream), a);
  PC3D_LAUNCH_CHECK("pc3d_kappa_f32");
  return PC3D_OK;
}

extern "C" int pc3d_kappa_gather_f32(const float* x, int64_t x_bs, int64_t x_ps, int64_t x_cs, const float* nrm,
                                     int64_t n_bs, int64_t n_ps, int64_t n_cs, int M, const int64_t* nidx,
                                     const int32_t* idx, int B, int N, int K1, float* out, float* nout, void* stream) {
  PC3D_REQUIRE(B >= 0 && N >= 1 && M >= 1 && K1 >= 2, "pc3d_kappa_gather_f32: bad sizes B=%d N=%d M=%d K1=%d", B, N, M, K1);
  PC3D_REQUIRE(B <= 65535, "pc3d_kappa_gather_f32: B=%d exceeds grid.y limit", B);
  if (B == 0) return PC3D_OK;
  PC3D_REQUIRE(x && nrm && nidx && idx && out && nout, "pc3d_kappa_gather_f32: null pointer");
  KappaArgs a{{x, x_bs, x_ps, x_cs}, {nrm, n_bs, n_ps, n_cs}, idx, N, K1, out, nullptr, nullptr, nidx, nout, M};
  hipLaunchKernelGGL(kappa_fwd_kernel, dim3(cdiv(N, 256), B), dim3(256), 0, as_stream(stream), a);
  PC3D_LAUNCH_CHECK("pc3d_kappa_gather_f32");
  return PC3D_OK;
}

extern "C" int pc3d_kappa_bwd_f32(const float* x, int64_t x_bs, int64_t x_ps, int64_t x_cs, const float* nrm,
                                  int64_t n_bs, int64_t n_ps, int64_t n_cs, const int32_t* idx, const float* gout, int B,
                                  int N, int K1, float* gx, int deterministic, void* stream) {
  PC3D_REQUIRE(B >= 0 && N >= 1 && K1 >= 2, "pc3d_kappa_bwd_f32: bad sizes B=%d N=%d K1=%d", B, N, K1);
  PC3D_REQUIRE(B <= 65535, "pc3d_kappa_bwd_f32: B=%d exceeds grid.y limit", B);
  if (B == 0) return PC3D_OK;
  PC3D_REQUIRE(x && nrm && idx && gout && gx, "pc3d_kappa_bwd_f32: null pointer");
  hipStream_t st = as_stream(stream);
  KappaArgs a{{x, x_bs, x_ps, x_cs}, {nrm, n_bs, n_ps, n_cs}, idx, N, K1, nullptr, gout, gx};
  if (deterministic) {
    // per-wave slabs: 4 waves while (3 + 4) N floats fit the default 64 KB window, else 2, else 1 with the window raised
    const int W = (size_t)7 * N * sizeof(float) <= 64 * 1024 ? 4 : ((size_t)5 * N * sizeof(float) <= 160 * 1024 ? 2 : 1);
    const size_t lds = (size_t)(3 + W) * N * sizeof(float);
    PC3D_REQUIRE(lds <= 160 * 1024, "pc3d_kappa_bwd_f32: N=%d does not fit a CU's LDS (deterministic mode)", N);
    if (W == 4) {
      hipLaunchKernelGGL(kappa_bwd_det_kernel<4>, dim3(3, B), dim3(256), lds, st, a);
    } else {
      auto* kern = W == 2 ? kappa_bwd_det_kernel<2> : kappa_bwd_det_kernel<1>;
      if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
          set_error("pc3d_kappa_bwd_f32: LDS opt-in failed: %s", hipGetErrorString(e));
          return (int)e;
        }
      }
      hipLaunchKernelGGL(kern, dim3(3, B), dim3(64 * W), lds, st, a);
    }
    PC3D_LAUNCH_CHECK("pc3d_kappa_bwd_f32");
    return PC3D_OK;
  }
  if (N <= kKappaSlabMaxN) {
    hipLaunchKernelGGL(kappa_bwd_slab_kernel, dim3(3, B), dim3(1024), (size_t)N * 4 * sizeof(float), st, a);
    PC3D_LAUNCH_CHECK("pc3d_kappa_bwd_f32");
    return PC3D_OK;
  }
  hipError_t e = zero_async(gx, (size_t)B * N * 3, st);
  if (e != hipSuccess) {
    set_error("pc3d_kappa_bwd_f32: zero fill failed: %s", hipGetErrorString(e));
    return (int)e;
  }
  hipLaunchKernelGGL(kappa_bwd_kernel, dim3(cdiv(N, 256), B), dim3(256), 0, st, a);
  PC3D_LAUNCH_CHECK("pc3d_kappa_bwd_f32");
  return PC3D_OK;
}
