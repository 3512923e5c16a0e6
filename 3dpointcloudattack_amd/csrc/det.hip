// Deterministic scatter-adds: the backward of every gather on the attack path without float atomics whose order the
// hardware picks.
//
// The reference's autograd of index_points / get_graph_feature / knn_gather (model/pointnet2_utils.py:41-57,
// model/dgcnn.py:203-227, attack/GeoA3/knn_utils.py:58-86) is deterministic on its CPU path: the same seed gives the
// same adversarial cloud, bit for bit. A scatter with global float atomics is not — the order in which the memory
// side retires them changes from run to run, a 1e-7 difference flips a near-tie in a later kNN graph / arg-max, and
// Adam amplifies the re-wiring (round 2: twelve runs of one GeoA3-on-DGCNN case followed two different branches).
//
// Pattern used here ("owner wave"): ONE wavefront owns an accumulator tile acc[N][CH] in LDS — all destination rows
// of a cloud, a slice of CH channels — and walks the source records in a fixed order, adding with ds_add_f32. LDS
// operations of one wave execute in issue order, and lanes of one instruction that hit the same address are
// serialised by the LDS in a fixed lane order, so the sum is a pure function of the inputs: run == run, graph replay
// == eager launch, and (one workgroup never sees another cloud) a cloud in a batch == the cloud alone.
// Parallelism comes from clouds x channel slices (B x C / CH single-wave workgroups), not from splitting one sum.
#include "pc3d_common.h"

namespace pc3d {

// ---------------------------------------------------------------------------------------------------------
// Generic form: out[b, tgt[b,r], c] = sum over records r (ascending) of val[b,r,c]   (optionally through the mask of an
// activation: act[b,r,c] > 0 ? val : slope * val). Records whose target is outside [0, N) are skipped.
// ---------------------------------------------------------------------------------------------------------
struct ScatterArgs {
  const int32_t* tgt;   // [B, R]
  const float* val;     // [B, R, ldv]
  const float* act;     // [B, R, lda] or null
  int64_t ldv, lda;
  float slope;
  int R, N, C;
  float* out;           // [B, N, ldo] (written, or added to when `accumulate`)
  int64_t ldo;
  int accumulate;
  int clamp;            // 1: targets are clamped into [0, N) (what the forward's gather did) instead of skipped
  const uint8_t* mbits; // [B, R, C/4] or null: the activation's sign as bits (bit c % 4 of byte c / 4), instead of `act`
  int64_t out_bs, out_cs;   // batch / channel stride of out in elements (a [B,3,N] gradient: ldo = 1, out_cs = N)
  const float* row_bias;    // [B, N] or null  } a rank-1 term added on the way out: out[b,n,c] = sum + row_bias[b,n] * col_w[c]
  const float* col_w;       // [C]             } (the guided walk's score term; a separate product, then a separate add)
  int NC;                   // destination rows per LDS tile: N when they all fit a CU's LDS (grid.z = 1); else grid.z chunks of NC
                            // rows, every chunk walking ALL records and keeping those that land in it (same order, more passes)
};

// Lane = RECORD: a lane loads the CH contiguous channels of its record (one or two 16-byte loads when the slice is
// aligned) and issues CH ds_add_f32, one per channel, so an instruction carries 64 records. W wavefronts share a
// workgroup, each with a PRIVATE tile and a contiguous range of the records; the W tiles are combined in ascending wave
// order at the end — still one fixed summation order, W times the loads in flight. (The first version of this kernel
// gave a record to CH lanes: 8-byte pieces of 128-byte rows per lane, 121 us for 20480 records x 32 channels.)
// ACC: the tiles' element type — double wherever the tiles fit (see arg_scatter_own_kernel: ds_add_f64 retires ~5x faster than
// ds_add_f32 on gfx950, and the sum is rounded once).
template <int CH, int W, typename ACC>
__global__ __launch_bounds__(64 * W) void scatter_rows_own_kernel(ScatterArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char so_raw[];
  ACC* so_acc = reinterpret_cast<ACC*>(so_raw);      // [W][N][ST]
  constexpr int ST = CH > 1 ? CH + 1 : 1;            // odd row stride: the rows of one instruction spread over the banks
  constexpr int U = CH >= 4 ? 4 : 8;                 // records per lane in flight
  int bx_, by_;
  xcd_swizzle(bx_, by_);      // a cloud's workgroups on one XCD (pc3d_common.h)
  const int b = by_, c0 = bx_ * CH;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nch = min(CH, a.C - c0);
  const int n0 = blockIdx.z * a.NC, nr = min(a.NC, a.N - n0);    // this workgroup's destination rows [n0, n0 + nr)
  for (int e = threadIdx.x; e < W * a.NC * ST; e += 64 * W) so_acc[e] = (ACC)0;
  __syncthreads();
  ACC* acc = so_acc + wave * a.NC * ST;
  const int32_t* tg = a.tgt + (int64_t)b * a.R;
  const float* vb = a.val + (int64_t)b * a.R * a.ldv + c0;
  const float* ab = a.act ? a.act + (int64_t)b * a.R * a.lda + c0 : nullptr;
  const bool vec = CH >= 4 && nch == CH && (a.ldv & 3) == 0 && (c0 & 3) == 0 && (reinterpret_cast<uintptr_t>(a.val) & 15) == 0 &&
                   (!a.act || ((a.lda & 3) == 0 && (reinterpret_cast<uintptr_t>(a.act) & 15) == 0));
  const int per = ((a.R + W - 1) / W + 63) / 64 * 64, lo = wave * per, hi = min(lo + per, a.R);
  for (int r0 = lo; r0 < hi; r0 += 64 * U) {
    int t[U];
    float v[U][CH];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int r = r0 + u * 64 + lane;
      const bool ok = r < hi;
      t[u] = ok ? (a.clamp ? min(max(tg[r], 0), a.N - 1) : tg[r]) : -1;
      float m[CH];
#pragma unroll
      for (int q = 0; q < CH; ++q) v[u][q] = 0.f, m[q] = 1.f;
      if (ok) {
        if (vec) {
#pragma unroll
          for (int q = 0; q < CH; q += 4) {
            const float4 x = *reinterpret_cast<const float4*>(vb + (int64_t)r * a.ldv + q);
            v[u][q] = x.x, v[u][q + 1] = x.y, v[u][q + 2] = x.z, v[u][q + 3] = x.w;
            if (ab) {
              const float4 y = *reinterpret_cast<const float4*>(ab + (int64_t)r * a.lda + q);
              m[q] = y.x, m[q + 1] = y.y, m[q + 2] = y.z, m[q + 3] = y.w;
            }
          }
        } else {
#pragma unroll
          for (int q = 0; q < CH; ++q)
            if (q < nch) {
              v[u][q] = vb[(int64_t)r * a.ldv + q];
              if (ab) m[q] = ab[(int64_t)r * a.lda + q];
            }
        }
        if (a.mbits) {
#pragma unroll
          for (int q = 0; q < CH; ++q)
            if (q < nch)
              m[q] = ((a.mbits[((int64_t)b * a.R + r) * (a.C >> 2) + ((c0 + q) >> 2)] >> ((c0 + q) & 3)) & 1) ? 1.f : -1.f;
        }
#pragma unroll
        for (int q = 0; q < CH; ++q) v[u][q] = m[q] > 0.f ? v[u][q] : v[u][q] * a.slope;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int tt = t[u] - n0;                                     // (t = -1 for lanes past the records: never inside)
      if ((unsigned)t[u] < (unsigned)a.N && (unsigned)tt < (unsigned)nr) {
#pragma unroll
        for (int q = 0; q < CH; ++q)
          if (q < nch) atomicAdd(&acc[tt * ST + q], (ACC)v[u][q]);      // ds_add_f64 / _f32 into the wave-private tile
      }
    }
  }
  __syncthreads();
  float* ob = a.out + (int64_t)b * a.out_bs + (int64_t)c0 * a.out_cs;
  for (int e = threadIdx.x; e < nr * nch; e += 64 * W) {
    const int nl = e / nch, q = e - nl * nch, n = n0 + nl;
    ACC sum_acc = so_acc[nl * ST + q];
#pragma unroll
    for (int w = 1; w < W; ++w) sum_acc += so_acc[(w * a.NC + nl) * ST + q];
    float sum = (float)sum_acc;
    float* o = ob + (int64_t)n * a.ldo + (int64_t)q * a.out_cs;
    if (a.row_bias) {
      const float t = a.row_bias[(int64_t)b * a.N + n] * a.col_w[c0 + q];
      sum = sum + t;
    }
    *o = a.accumulate ? *o + sum : sum;
  }
}

// ---------------------------------------------------------------------------------------------------------
// Per-CHANNEL targets (the backward of a max over gathered rows): dst[b, arg[b,i,c], c] += w[b,i,c].
//   mode 0 (gather_max, MaskedMaxPool):  w = g                              g, arg [B,S,C] -> gP [B,N,C]
//   mode 1 (EdgeConv epilogue):          w = g * (out > 0 ? 1 : slope); the centre half dQ[b,i,c] = w is written too
//                                        g [B,N,ldg], out / arg [B,N,C] -> gPQ [B,N,2C] = [dP | dQ]
// Lane = source point (its CH channels as 16-byte loads), W = 1: the points are walked in order by one wavefront.
// ---------------------------------------------------------------------------------------------------------
struct ArgScatterArgs {
  const float* g;
  int64_t ldg;
  const float* out;     // mode 1
  const int32_t* arg;
  int S, N, C;
  float slope;
  float* dst;           // mode 0: gP [B,N,C]; mode 1: gPQ [B,N,2C]
  int mode;
  int NC;               // destination rows per LDS tile (N when they fit; else grid.z chunks, as in ScatterArgs)
  int B;
  int blind;            // measurements only: workgroup ids in launch order (XCD-blind), the order before round 4
  const float* g2;      // null, or a second upstream gradient [B,S,ldg2 >= C] ADDED to g on load (mode 1: an EdgeConv output
  int64_t ldg2;         // feeds conv5 and the next layer — the sum of their two gradients without a launch of its own)
};

// One trip of a wavefront: 64 * U source points, CH channels each. The load half only ISSUES the loads (nothing computed
// on what they return, so the wave does not wait for them); the add half applies the activation's mask, writes the centre
// half and issues the ds_add_f32 — one trip later.
template <int CH, int U>
struct ArgTrip {
  int t[U][CH];
  float v[U][CH], o[U][CH], v2[U][CH];
  int i[U];          // source point, -1: none
};

template <int CH, int U, bool VEC, int MODE>
__device__ __forceinline__ void arg_trip_load(ArgTrip<CH, U>& T, const ArgScatterArgs& a, const float* gb, const float* ob,
                                              const int32_t* rb, int i0, int hi, int lane, int nch, const float* g2b = nullptr) {
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int i = i0 + u * 64 + lane;
    T.i[u] = i < hi ? i : -1;
    if (VEC) {      // no branch around the loads: lanes past the end re-read the last point and are dropped in the add half
      const int ic = i < hi ? i : hi - 1;      // (branches would make the wait counters unknown at their joins: vmcnt(0) everywhere)
#pragma unroll
      for (int q = 0; q < CH; q += 4) {
        const float4 x = *reinterpret_cast<const float4*>(gb + (int64_t)ic * a.ldg + q);
        const int4 r = *reinterpret_cast<const int4*>(rb + (int64_t)ic * a.C + q);
        T.v[u][q] = x.x, T.v[u][q + 1] = x.y, T.v[u][q + 2] = x.z, T.v[u][q + 3] = x.w;
        T.t[u][q] = r.x, T.t[u][q + 1] = r.y, T.t[u][q + 2] = r.z, T.t[u][q + 3] = r.w;
        if (MODE) {
          const float4 o = *reinterpret_cast<const float4*>(ob + (int64_t)ic * a.C + q);
          T.o[u][q] = o.x, T.o[u][q + 1] = o.y, T.o[u][q + 2] = o.z, T.o[u][q + 3] = o.w;
        }
        if (g2b) {      // (uniform for the launch)
          const float4 y = *reinterpret_cast<const float4*>(g2b + (int64_t)ic * a.ldg2 + q);
          T.v2[u][q] = y.x, T.v2[u][q + 1] = y.y, T.v2[u][q + 2] = y.z, T.v2[u][q + 3] = y.w;
        }
      }
    } else {
#pragma unroll
      for (int q = 0; q < CH; ++q) T.t[u][q] = 0, T.v[u][q] = 0.f, T.o[u][q] = 1.f;
      if (i >= hi) continue;
#pragma unroll
      for (int q = 0; q < CH; ++q)
        if (q < nch) {
          T.v[u][q] = gb[(int64_t)i * a.ldg + q];
          T.t[u][q] = rb[(int64_t)i * a.C + q];
          if (MODE) T.o[u][q] = ob[(int64_t)i * a.C + q];
          if (g2b) T.v2[u][q] = g2b[(int64_t)i * a.ldg2 + q];
        }
    }
  }
}

template <int CH, int U, bool VEC, int MODE, typename ACC>
__device__ __forceinline__ void arg_trip_add(const ArgTrip<CH, U>& T, const ArgScatterArgs& a, ACC* acc, float* db, int64_t ldd,
                                             int n0, int nr, int nch, bool first, bool two = false) {
  constexpr int ST = CH > 1 ? CH + 1 : 1;
  constexpr bool vec = VEC;
#pragma unroll
  for (int u = 0; u < U; ++u)
    if (T.i[u] >= 0) {
      float v[CH];
#pragma unroll
      for (int q = 0; q < CH; ++q) {
        const float gsum = two ? T.v[u][q] + T.v2[u][q] : T.v[u][q];       // (one fp32 add, then the mask: what autograd's add + this launch did)
        v[q] = MODE ? gsum * (T.o[u][q] > 0.f ? 1.f : a.slope) : gsum;
      }
      if (MODE && first) {                                                      // dQ
        float* dq = db + (int64_t)T.i[u] * ldd + a.C;
        if (vec) {
#pragma unroll
          for (int q = 0; q < CH; q += 4) *reinterpret_cast<float4*>(dq + q) = make_float4(v[q], v[q + 1], v[q + 2], v[q + 3]);
        } else {
#pragma unroll
          for (int q = 0; q < CH; ++q)
            if (q < nch) dq[q] = v[q];
        }
      }
#pragma unroll
      for (int q = 0; q < CH; ++q) {  // (clamped like the forward's gather)
        const int tt = min(max(T.t[u][q], 0), a.N - 1) - n0;
        if (q < nch && (unsigned)tt < (unsigned)nr) atomicAdd(&acc[tt * ST + q], (ACC)v[q]);
      }
    }
}

// W wavefronts per (cloud, channel slice), each with a PRIVATE tile and a contiguous piece of the source points (walked in
// order, the next trip's loads issued before the current trip's ds_add_f32); the W tiles are added in ascending wave order
// on the way out. The pieces depend on S alone (args_pieces), so a sum is the same function of its inputs whatever the
// slice width and the batch size.
// VEC: whole, 16-byte aligned slices (the launcher checks) — straight-line 16-byte loads and stores.
// ACC: the tiles' element type. double by default — gfx950's LDS retires a ds_add_f64 in ~40 clocks per wavefront and a
// ds_add_f32 in ~190 whatever the addresses (tools/exp/lds_atomic_rate.hip: the float form is serialised per lane, the
// double and integer forms are not), and a sum of fp32 terms carried in fp64 is rounded once, on the way out.
template <int CH, int W, bool VEC, int MODE, typename ACC>
__global__ __launch_bounds__(64 * W) void arg_scatter_own_kernel(ArgScatterArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char as_raw[];
  ACC* as_acc = reinterpret_cast<ACC*>(as_raw);                                  // [W][NC][ST]
  constexpr int ST = CH > 1 ? CH + 1 : 1;
  constexpr int U = CH >= 16 ? 2 : (CH >= 8 ? 4 : 8);      // two trips of 64 U points in flight: ~190 VGPRs (one wave per SIMD anyway)
  // grid.x = (cloud, slice) in XCD bands: the slices of a cloud read the same lines of g / arg / out (a slice uses 16-64 bytes
  // of each 128-byte line) — spread over the XCDs every line was pulled by four L2s (FETCH_SIZE 101 MB against 25 MB at B=32, N=1024, C=64)
  const int nsl = (a.C + CH - 1) / CH;
  const int t = a.blind ? ((int)blockIdx.x < a.B * nsl ? (int)blockIdx.x : -1) : xcd_band_id(blockIdx.x, a.B * nsl);
  if (t < 0) return;
  const int b = t / nsl, c0 = (t - b * nsl) * CH, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nch = VEC ? CH : min(CH, a.C - c0);
  const int n0 = blockIdx.z * a.NC, nr = min(a.NC, a.N - n0);
  const bool first = blockIdx.z == 0;               // the chunk that also writes the centre half (mode 1)
  const int64_t ldd = MODE ? 2 * (int64_t)a.C : (int64_t)a.C;
  const float* gb = a.g + (int64_t)b * a.S * a.ldg + c0;
  const float* ob = a.out ? a.out + (int64_t)b * a.S * a.C + c0 : nullptr;
  const int32_t* rb = a.arg + (int64_t)b * a.S * a.C + c0;
  float* db = a.dst + (int64_t)b * a.N * ldd + c0;
  const float* g2b = a.g2 ? a.g2 + (int64_t)b * a.S * a.ldg2 + c0 : nullptr;
  const bool two = a.g2 != nullptr;
  constexpr bool vec = VEC;
  const int per = ((a.S + W - 1) / W + 63) / 64 * 64, lo = wave * per, hi = min(lo + per, a.S);
  ArgTrip<CH, U> A, Bt;
  const bool any = lo < hi;                                                              // (uniform per wavefront)
  if (any) arg_trip_load<CH, U, VEC, MODE>(A, a, gb, ob, rb, lo, hi, lane, nch, g2b);        // in flight while the tiles are zeroed
  for (int e = threadIdx.x; e < W * a.NC * ST; e += 64 * W) as_acc[e] = (ACC)0;
  if (W > 1) __syncthreads(); else wave_lds_sync();
  ACC* acc = as_acc + wave * a.NC * ST;
  if (any)
    for (int i0 = lo; i0 < hi; i0 += 2 * 64 * U) {
      arg_trip_load<CH, U, VEC, MODE>(Bt, a, gb, ob, rb, i0 + 64 * U, hi, lane, nch, g2b);
      arg_trip_add<CH, U, VEC, MODE, ACC>(A, a, acc, db, ldd, n0, nr, nch, first, two);
      arg_trip_load<CH, U, VEC, MODE>(A, a, gb, ob, rb, i0 + 2 * 64 * U, hi, lane, nch, g2b);
      arg_trip_add<CH, U, VEC, MODE, ACC>(Bt, a, acc, db, ldd, n0, nr, nch, first, two);
    }
  if (W > 1) __syncthreads(); else wave_lds_sync();
  for (int nl = threadIdx.x; nl < nr; nl += 64 * W) {
    const int n = n0 + nl;
    float r[CH];
#pragma unroll
    for (int q = 0; q < CH; ++q) {
      ACC sum = as_acc[nl * ST + q];
#pragma unroll
      for (int w = 1; w < W; ++w) sum += as_acc[(w * a.NC + nl) * ST + q];
      r[q] = (float)sum;
    }
    if (vec) {
#pragma unroll
      for (int q = 0; q < CH; q += 4) *reinterpret_cast<float4*>(db + (int64_t)n * ldd + q) = make_float4(r[q], r[q + 1], r[q + 2], r[q + 3]);
    } else {
#pragma unroll
      for (int q = 0; q < CH; ++q)
        if (q < nch) db[(int64_t)n * ldd + q] = r[q];
    }
  }
}

// largest channel slice whose tile fits `budget` bytes of LDS; 0: not even one channel does
static int own_slice(int N, size_t budget) {
  for (int ch = 8; ch >= 2; ch >>= 1)
    if ((size_t)N * (ch + 1) * sizeof(float) <= budget) return ch;
  return (size_t)N * sizeof(float) <= budget ? 1 : 0;
}
constexpr size_t kOwnLds = 64 * 1024;      // default dynamic-LDS window
constexpr size_t kOwnLdsMax = 160 * 1024;  // a CU's LDS (gfx950): one workgroup per CU beyond the default window
static size_t own_bytes(int N, int ch, int w = 1) { return (size_t)w * N * (ch > 1 ? ch + 1 : 1) * sizeof(float); }

template <typename K>
static hipError_t own_lds_optin(K kernel, size_t bytes) {
  if (bytes <= kOwnLds) return hipSuccess;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

#define PC3D_OWN_LAUNCH(KERNEL, threads, grid, lds, st, args)                    \
  do {                                                                           \
    hipError_t oe_ = own_lds_optin(KERNEL, lds);                                 \
    if (oe_ != hipSuccess) {                                                     \
      set_error("%s: LDS opt-in failed: %s", nm, hipGetErrorString(oe_));        \
      return (int)oe_;                                                           \
    }                                                                            \
    hipLaunchKernelGGL(KERNEL, grid, dim3(threads), lds, st, args);              \
  } while (0)

int scatter_rows_det(const char* nm, const int32_t* tgt, const float* val, int64_t ldv, const float* act, int64_t lda, float slope,
                     int B, int R, int N, int C, float* out, int64_t ldo, int accumulate, int clamp, void* stream,
                     const uint8_t* mbits, int64_t out_bs, int64_t out_cs, const float* row_bias, const float* col_w) {
  if (out_bs == 0) out_bs = (int64_t)N * ldo;
  PC3D_REQUIRE(B >= 0 && R >= 0 && N >= 1 && C >= 1 && ldv >= C && (ldo >= C || out_cs != 1) && (!act || lda >= C), "%s: bad sizes", nm);
  PC3D_REQUIRE(B <= 65535, "%s: B=%d exceeds grid.y limit", nm, B);
  if (B == 0) return PC3D_OK;
  PC3D_REQUIRE((tgt && val) || R == 0, "%s: null pointer", nm);
  PC3D_REQUIRE(out != nullptr, "%s: null output", nm);
  PC3D_REQUIRE(!mbits || C % 4 == 0, "%s: the bit mask needs C %% 4 == 0", nm);
  // channel slice: 4 (one 16-byte load per record) when the rows have >= 4 channels, else 1; narrower if even that
  // tile does not fit a CU's LDS. Waves per workgroup: as many private tiles as fit half a CU's LDS (<= 8),
  // but no more than leave every wave >= 256 records.
  int ch = C >= 4 ? 4 : 1;
  while (ch > 1 && own_bytes(N, ch) > kOwnLds) ch >>= 1;
  // More rows than a CU's LDS holds (N > ~40 000): tiles of NC rows, every tile walking all records (rounds 1-2
  // took such sizes with atomics; refusing them made the deterministic default a regression). Same summation order per row.
  int NC = N;
  if (own_bytes(N, ch) > kOwnLdsMax) {
    NC = (int)(kOwnLds / (sizeof(float) * (ch > 1 ? ch + 1 : 1)));
    PC3D_REQUIRE(cdiv(N, NC) <= 65535, "%s: N=%d needs more than 65535 row tiles", nm, N);
  }
  // tiles in fp64 when one of them fits the default window (a function of N and C alone, like everything that shapes a sum here)
  const bool f64 = NC == N && own_bytes(N, ch) * 2 <= kOwnLds;
  const size_t eb = f64 ? 2 : 1;
  int w = 1;
  while (NC == N && w < 8 && own_bytes(N, ch, 2 * w) * eb <= kOwnLdsMax / 2 && R >= 2 * w * 256) w *= 2;      // (80 KB: two workgroups per CU)
  ScatterArgs a{tgt, val, act, ldv, lda, slope, R, N, C, out, ldo, accumulate, clamp, mbits, out_bs, out_cs, row_bias, col_w, NC};
  const size_t lds = own_bytes(NC, ch, w) * eb;
  const dim3 grid(cdiv(C, ch), B, cdiv(N, NC));
  hipStream_t st = as_stream(stream);
#define PC3D_SR1(CHV, WV)                                                                                        \
  if (f64) PC3D_OWN_LAUNCH((scatter_rows_own_kernel<CHV, WV, double>), 64 * WV, grid, lds, st, a);                \
  else PC3D_OWN_LAUNCH((scatter_rows_own_kernel<CHV, WV, float>), 64 * WV, grid, lds, st, a);
#define PC3D_SR(CHV)                        \
  switch (w) {                              \
    case 8: PC3D_SR1(CHV, 8) break;         \
    case 4: PC3D_SR1(CHV, 4) break;         \
    case 2: PC3D_SR1(CHV, 2) break;         \
    default: PC3D_SR1(CHV, 1) break;        \
  }
  if (ch == 4) { PC3D_SR(4) } else if (ch == 2) { PC3D_SR(2) } else { PC3D_SR(1) }
#undef PC3D_SR1
#undef PC3D_SR
  PC3D_LAUNCH_CHECK(nm);
  return PC3D_OK;
}

// slice: bits 0..7 the channel-slice width to run (1, 2, 4, 8, 16; 0: chosen here; PC3D_EINVAL when its tile does not fit),
// bit 8: fp32 tiles where the library would take fp64 ones, bit 9: workgroups in launch order instead of XCD bands
// (measurements: tools/bench_det.py).
int arg_scatter_det(const char* nm, const float* g, int64_t ldg, const float* outv, const int32_t* arg, int B, int S, int N, int C,
                    float slope, float* dst, int mode, void* stream, int slice, const float* g2, int64_t ldg2) {
  PC3D_REQUIRE(!g2 || ldg2 >= C, "%s: row stride of the second gradient smaller than C", nm);
  // Tiles in fp64 whenever a 4-channel slice of them fits a CU's LDS (N <= 4096) — a function of N alone, so that a cloud's
  // sums do not depend on the batch it is in; the slice width does not change a sum (one wave, points in order).
  const int blind = (slice >> 9) & 1;
  slice &= 511;
  const bool f64 = !(slice & 256) && own_bytes(N, 4) * 2 <= kOwnLdsMax;
  const size_t eb = f64 ? 2 : 1;
  int ch = slice & 0xff;
  slice = ch;
  if (ch == 0) {
    // measured inside replayed graphs (tools/bench_det.py), us at B=32, N=1024, slice 4 / 8 / 16:
    // C=64 16.6 / 17.9 / 26.2, C=128 28.8 / 28.6 / 30.4, C=256 56.9 / 47.4 / 55.3 (float atomics: 22 / 44 / 92);
    // B=1, C=256: 9.4 / 15.1 / 25.2; N=4096, C=32: 36.3 (slice 4; atomics 209)
    ch = C >= 4 ? 4 : 1;
    if (C % 8 == 0 && (long)B * (C / 8) >= 128 && own_bytes(N, 8) * eb <= kOwnLdsMax) ch = 8;
    while (ch > 1 && own_bytes(N, ch) * eb > kOwnLdsMax) ch >>= 1;
  }
  PC3D_REQUIRE(ch == 1 || ch == 2 || ch == 4 || ch == 8 || ch == 16, "%s: slice width %d (1, 2, 4, 8, 16)", nm, ch);
  int NC = N;
  if (own_bytes(N, ch) * eb > kOwnLdsMax) {          // as scatter_rows_det: row tiles, all records per tile
    PC3D_REQUIRE(slice == 0, "%s: N=%d destination rows do not fit a CU's LDS at slice width %d", nm, N, ch);
    NC = (int)(kOwnLds / (sizeof(float) * eb * (ch > 1 ? ch + 1 : 1)));
    PC3D_REQUIRE(cdiv(N, NC) <= 65535, "%s: N=%d needs more than 65535 row tiles", nm, N);
  }
  ArgScatterArgs a{g, ldg, outv, arg, S, N, C, slope, dst, mode, NC, B, blind, g2, ldg2};
  const size_t lds = own_bytes(NC, ch) * eb;
  const dim3 grid(xcd_grid(cdiv(C, ch) * B), 1, cdiv(N, NC));
  hipStream_t st = as_stream(stream);
  const bool vec = ch >= 4 && C % ch == 0 && (C & 3) == 0 && (ldg & 3) == 0 && (ldg2 & 3) == 0 &&
                   ((reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(arg) | reinterpret_cast<uintptr_t>(dst) |
                     reinterpret_cast<uintptr_t>(outv) | reinterpret_cast<uintptr_t>(g2)) & 15) == 0;
#define PC3D_AS2(CHV, VECV, MODEV)                                                                              \
  if (f64) PC3D_OWN_LAUNCH((arg_scatter_own_kernel<CHV, 1, VECV, MODEV, double>), 64, grid, lds, st, a);         \
  else PC3D_OWN_LAUNCH((arg_scatter_own_kernel<CHV, 1, VECV, MODEV, float>), 64, grid, lds, st, a);
#define PC3D_AS(CHV)                                       \
  if (vec && CHV >= 4 && mode) { PC3D_AS2(CHV, (CHV >= 4), 1) } \
  else if (vec && CHV >= 4) { PC3D_AS2(CHV, (CHV >= 4), 0) }    \
  else if (mode) { PC3D_AS2(CHV, false, 1) }               \
  else { PC3D_AS2(CHV, false, 0) }
  switch (ch) {
    case 16: PC3D_AS(16) break;
    case 8: PC3D_AS(8) break;
    case 4: PC3D_AS(4) break;
    case 2: PC3D_AS(2) break;
    default: PC3D_AS(1) break;
  }
#undef PC3D_AS2
#undef PC3D_AS
  PC3D_LAUNCH_CHECK(nm);
  return PC3D_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Sorted reverse index of a gather (CSR): for idx [B, E] with values in [0, NA), lst [b, off[b,t] .. off[b,t+1]) = the
// entries e with idx[b,e] == t in ASCENDING e. Counting sort on the device (integer atomics: order-free), then every
// segment is sorted (sort_segments_kernel). With it the backward of the gather is itself a gather — rev_gather_sum: every
// destination row sums its segment front to back, all rows in parallel, coalesced row reads, no LDS — which is what the
// backward of wide, many-edge gathers wants (LPFA: 20480 edges x 16..128 channels per cloud; the owner-wave scatter
// above walks those with one wave per channel slice).
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rev_count_kernel(const int* __restrict__ idx, int E, int NA, int clamp, int* __restrict__ cnt) {
  const int b = blockIdx.y, e = blockIdx.x * 256 + threadIdx.x;
  if (e >= E) return;
  int t = idx[(int64_t)b * E + e];
  if (clamp) t = min(max(t, 0), NA - 1);
  if ((unsigned)t < (unsigned)NA) atomicAdd(cnt + (int64_t)b * NA + t, 1);
}

// exclusive scan of one cloud's counters -> off [NA + 1]; the counters are zeroed (they become the fill cursors)
__global__ __launch_bounds__(256) void rev_scan_kernel(int* __restrict__ cnt, int NA, int* __restrict__ off) {
  __shared__ int part[256];
  const int b = blockIdx.x, t = threadIdx.x;
  int* c = cnt + (int64_t)b * NA;
  int* o = off + (int64_t)b * (NA + 1);
  const int per = (NA + 255) / 256, lo = t * per, hi = min(lo + per, NA);
  int s = 0;
  for (int i = lo; i < hi; ++i) s += c[i];
  part[t] = s;
  __syncthreads();
  for (int d = 1; d < 256; d <<= 1) {
    const int v = t >= d ? part[t - d] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  int run = part[t] - s;
  for (int i = lo; i < hi; ++i) {
    const int v = c[i];
    o[i] = run;
    run += v;
    c[i] = 0;
  }
  if (t == 255) o[NA] = part[255];
}

__global__ __launch_bounds__(256) void rev_fill_kernel(const int* __restrict__ idx, int E, int NA, int clamp, int* __restrict__ cur,
                                                       const int* __restrict__ off, int* __restrict__ lst) {
  const int b = blockIdx.y, e = blockIdx.x * 256 + threadIdx.x;
  if (e >= E) return;
  int t = idx[(int64_t)b * E + e];
  if (clamp) t = min(max(t, 0), NA - 1);
  if ((unsigned)t >= (unsigned)NA) return;
  const int slot = atomicAdd(cur + (int64_t)b * NA + t, 1);
  lst[(int64_t)b * E + off[(int64_t)b * (NA + 1) + t] + slot] = e;
}

struct RevGatherArgs {
  const float* val;     // [B, E, ldv]
  const float* act;     // [B, E, lda] or null
  int64_t ldv, lda;
  float slope;
  const int* off;       // [B, NA + 1]
  const int* lst;       // [B, E]
  int E, NA, C;
  float* out;           // [B, NA, ldo]
  int64_t ldo;
};

// VEC: a lane owns 4 consecutive channels of one destination row (C / 4 lanes per row), else one channel.
template <bool VEC>
__global__ __launch_bounds__(256) void rev_gather_sum_kernel(RevGatherArgs a) {
  int bx_, by_;
  xcd_swizzle(bx_, by_);      // a cloud's workgroups on one XCD (pc3d_common.h)
  const int b = by_;
  const int lpt = VEC ? a.C >> 2 : a.C;                 // lanes per destination row
  const int64_t g = (int64_t)bx_ * 256 + threadIdx.x;
  const int t = (int)(g / lpt), l = (int)(g - (int64_t)t * lpt);
  if (t >= a.NA) return;
  const int* o = a.off + (int64_t)b * (a.NA + 1);
  const int* ls = a.lst + (int64_t)b * a.E;
  const int t0 = o[t], t1 = o[t + 1];
  const int c = VEC ? 4 * l : l;
  const float* vb = a.val + (int64_t)b * a.E * a.ldv + c;
  const float* ab = a.act ? a.act + (int64_t)b * a.E * a.lda + c : nullptr;
  constexpr int U = 4;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int p = t0; p < t1; p += U) {
    float v[U][4], m[U][4];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool ok = p + u < t1;
      const int e = ls[ok ? p + u : p];
#pragma unroll
      for (int q = 0; q < 4; ++q) v[u][q] = 0.f, m[u][q] = 1.f;
      if (!ok) continue;
      if (VEC) {
        const float4 x = *reinterpret_cast<const float4*>(vb + (int64_t)e * a.ldv);
        v[u][0] = x.x, v[u][1] = x.y, v[u][2] = x.z, v[u][3] = x.w;
        if (ab) {
          const float4 y = *reinterpret_cast<const float4*>(ab + (int64_t)e * a.lda);
          m[u][0] = y.x, m[u][1] = y.y, m[u][2] = y.z, m[u][3] = y.w;
        }
      } else {
        v[u][0] = vb[(int64_t)e * a.ldv];
        if (ab) m[u][0] = ab[(int64_t)e * a.lda];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)            // ascending list position: one fixed summation order
#pragma unroll
      for (int q = 0; q < (VEC ? 4 : 1); ++q) acc[q] += m[u][q] > 0.f ? v[u][q] : v[u][q] * a.slope;
  }
  float* ob = a.out + ((int64_t)b * a.NA + t) * a.ldo + c;
  if (VEC) *reinterpret_cast<float4*>(ob) = make_float4(acc[0], acc[1], acc[2], acc[3]);
  else *ob = acc[0];
}

int rev_gather_sum(const char* nm, const float* val, int64_t ldv, const float* act, int64_t lda, float slope, const int32_t* off,
                   const int32_t* lst, int B, int E, int NA, int C, float* out, int64_t ldo, void* stream) {
  PC3D_REQUIRE(B >= 0 && E >= 1 && NA >= 1 && C >= 1 && ldv >= C && ldo >= C && (!act || lda >= C), "%s: bad sizes", nm);
  PC3D_REQUIRE(B <= 65535, "%s: B=%d exceeds grid.y limit", nm, B);
  if (B == 0) return PC3D_OK;
  PC3D_REQUIRE(val && off && lst && out, "%s: null pointer", nm);
  const bool vec = C % 4 == 0 && ldv % 4 == 0 && ldo % 4 == 0 && (!act || lda % 4 == 0) &&
                   ((reinterpret_cast<uintptr_t>(val) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(act)) & 15) == 0;
  RevGatherArgs a{val, act, ldv, lda, slope, off, lst, E, NA, C, out, ldo};
  const int64_t threads = (int64_t)NA * (vec ? C / 4 : C);
  PC3D_REQUIRE((threads + 255) / 256 <= 0x7fffffffLL, "%s: problem too large", nm);
  const dim3 grid((unsigned)((threads + 255) / 256), B);
  if (vec) hipLaunchKernelGGL(rev_gather_sum_kernel<true>, grid, dim3(256), 0, as_stream(stream), a);
  else hipLaunchKernelGGL(rev_gather_sum_kernel<false>, grid, dim3(256), 0, as_stream(stream), a);
  PC3D_LAUNCH_CHECK(nm);
  return PC3D_OK;
}

bool own_fits(int N) { return own_slice(N, kOwnLdsMax) > 0; }

// ---------------------------------------------------------------------------------------------------------
// In-place ascending sort of the segments of a CSR list (off [B, NA + 1], lst [B, L], distinct entries per segment):
// a reverse index filled through integer atomics has its segments in arrival order, which differs from run to run;
// sorted, a gather that sums a segment front to back is deterministic. A wavefront per segment: every lane keeps up to
// 8 entries in registers and ranks them against all entries (v_readlane broadcasts); longer segments (degenerate
// inputs: hundreds of groups through one point) are insertion-sorted by one lane.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sort_segments_kernel(const int* __restrict__ off, int* __restrict__ lst, int NA, int64_t L) {
  const int b = blockIdx.y, lane = threadIdx.x & 63;
  const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= NA) return;
  const int* o = off + (int64_t)b * (NA + 1);
  int* l = lst + (int64_t)b * L;
  const int t0 = __builtin_amdgcn_readfirstlane(o[p]), n = __builtin_amdgcn_readfirstlane(o[p + 1]) - t0;
  if (n <= 1) return;
  constexpr int E = 8;
  if (n <= 64 * E) {
    int v[E], rk[E];
#pragma unroll
    for (int e = 0; e < E; ++e) v[e] = (e * 64 + lane < n) ? l[t0 + e * 64 + lane] : 0x7fffffff, rk[e] = 0;
#pragma unroll
    for (int e2 = 0; e2 < E; ++e2) {
      if (e2 * 64 >= n) break;                        // wave-uniform
      const int lim = min(64, n - e2 * 64);
      for (int s = 0; s < lim; ++s) {
        const int x = __builtin_amdgcn_readlane(v[e2], s);
#pragma unroll
        for (int e = 0; e < E; ++e) rk[e] += x < v[e] ? 1 : 0;
      }
    }
    wave_lds_sync();                                  // (global memory, one wave: all loads above are complete — the
    __builtin_amdgcn_s_waitcnt(0);                    //  ranks depend on them)
#pragma unroll
    for (int e = 0; e < E; ++e)
      if (e * 64 + lane < n) l[t0 + rk[e]] = v[e];
    return;
  }
  if (lane == 0) {
    for (int i = 1; i < n; ++i) {
      const int x = l[t0 + i];
      int j = i - 1;
      while (j >= 0 && l[t0 + j] > x) l[t0 + j + 1] = l[t0 + j], --j;
      l[t0 + j + 1] = x;
    }
  }
}

int sort_segments(const char* nm, const int32_t* off, int32_t* lst, int B, int NA, int64_t L, void* stream) {
  PC3D_REQUIRE(B <= 65535, "%s: B=%d exceeds grid.y limit", nm, B);
  hipLaunchKernelGGL(sort_segments_kernel, dim3(cdiv(NA, 4), B), dim3(256), 0, as_stream(stream), off, lst, NA, L);
  PC3D_LAUNCH_CHECK(nm);
  return PC3D_OK;
}

}  // namespace pc3d

using namespace pc3d;

extern "C" int pc3d_rev_index_i32(const int32_t* idx, int B, int E, int NA, int clamp, int32_t* cnt, int32_t* off, int32_t* lst,
                                  void* stream) {
  const char* nm = "pc3d_rev_index_i32";
  PC3D_REQUIRE(B >= 0 && E >= 1 && NA >= 1, "%s: bad sizes B=%d E=%d NA=%d", nm, B, E, NA);
  PC3D_REQUIRE(B <= 65535, "%s: B=%d exceeds grid.y limit", nm, B);
  if (B == 0) return PC3D_OK;
  PC3D_REQUIRE(idx && cnt && off && lst, "%s: null pointer", nm);
  hipStream_t st = as_stream(stream);
  if (hipError_t e = zero_async(reinterpret_cast<float*>(cnt), (size_t)B * NA, st); e != hipSuccess) {   // all-zero bits
    set_error("%s: zero fill failed: %s", nm, hipGetErrorString(e));
    return (int)e;
  }
  const dim3 grid(cdiv(E, 256), B);
  hipLaunchKernelGGL(rev_count_kernel, grid, dim3(256), 0, st, idx, E, NA, clamp, cnt);
  hipLaunchKernelGGL(rev_scan_kernel, dim3(B), dim3(256), 0, st, cnt, NA, off);
  hipLaunchKernelGGL(rev_fill_kernel, grid, dim3(256), 0, st, idx, E, NA, clamp, cnt, off, lst);
  PC3D_LAUNCH_CHECK(nm);
  return sort_segments(nm, off, lst, B, NA, E, stream);
}

extern "C" int pc3d_rev_gather_sum_f32(const float* val, int64_t ldv, const float* act, int64_t lda, float slope,
                                       const int32_t* off, const int32_t* lst, int B, int E, int NA, int C, float* out,
                                       int64_t ldo, void* stream) {
  return rev_gather_sum("pc3d_rev_gather_sum_f32", val, ldv, act, lda, slope, off, lst, B, E, NA, C, out, ldo, stream);
}

extern "C" int pc3d_scatter_rows_det_f32(const int32_t* tgt, const float* val, int64_t ldv, const float* act, int64_t lda,
                                         float slope, int B, int R, int N, int C, float* out, int64_t ldo, int accumulate,
                                         int clamp, void* stream) {
  return scatter_rows_det("pc3d_scatter_rows_det_f32", tgt, val, ldv, act, lda, slope, B, R, N, C, out, ldo, accumulate, clamp,
                          stream, nullptr, 0, 1);
}
