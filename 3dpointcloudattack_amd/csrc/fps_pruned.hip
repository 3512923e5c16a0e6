// K5b — farthest-point sampling with EXACT pruning (model/pointnet2_utils.py:60-81, model/curvenet_util.py:69-90).
//
// The sampling is a chain of S dependent steps; in fps_kernel (group.hip) every step updates all N running distances
// (317 instructions per wavefront and step at N = 4096 — a single wavefront issues one instruction per four clocks, so
// instructions are the currency) before four wavefronts meet at a barrier. Here a step touches only the points that can change:
//   * setup: the cloud is split into ROWS of 64 spatially close points by a binary k-d partition (per level a histogram of
//     one coordinate per segment, the median bin from a prefix scan, a two-sided scatter with wave-aggregated cursors; the
//     split dimensions are chosen once, from the cloud's extents); every row gets its bounding box. Rows are dealt to the four
//     wavefronts round-robin (neighbouring rows are spatial neighbours, so a step's work spreads over the waves) and live in
//     REGISTERS, as in fps_kernel;
//   * a step with new centre c: lane j of a wave holds the box of the wave's j-th row and its largest running distance. A
//     row whose box is at least sqrt(maximum) away from c cannot change — every point p of it has |p - c|^2 >= bound >= max
//     >= dist[p], and the bound is computed with the operations of the distance itself, each monotone under rounding, so the
//     comparison is exact in floating point, not only in real arithmetic. The other rows (4-5 of 64 on average for a uniform
//     ball, fewer for surfaces) are updated; each leaves its new maximum and arg-max lane in lane j;
//   * the wave's best row comes from a 16-lane DPP reduction over those maxima; the four waves exchange (value, position)
//     through LDS records stamped with the step number and POLL for each other's stamp instead of meeting at a barrier.
// The layout inside the rows depends on the arrival order of LDS atomics; the picks do not: distances are computed per point
// by the reference's expression, ties go to the lowest ORIGINAL index (inside a row, between a wave's rows and between
// waves: the slow paths below), so the index sequence is bit-identical to fps_kernel's — tests/test_fps_pruned_gpu.py.
#include "pc3d_common.h"

namespace pc3d {

typedef __attribute__((address_space(3))) volatile unsigned long long fpsp_lds_u64;

constexpr int FPSP_T = 256;
constexpr int FPSP_MAXN = 4096;          // 64 rows: a lane per row
constexpr int FPSP_MAXJ = FPSP_MAXN / FPSP_T;

struct FpsPrunedArgs {
  PtsView x;
  int N, S;
  const int32_t* start;  // [B] or null
  int32_t* out;          // [B,S]
  int Npad;              // 64 * 2^L >= N
  int L;                 // k-d levels = log2(rows)
#ifdef FPSP_DIAG
  long long* diag;       // [B,4 waves,8]: clocks spent per phase (diagnostic build only: tools/exp/fps_diag.hip)
#endif
};

#ifdef FPSP_DIAG
#define FPSP_STAMP(k)                                        \
  do {                                                       \
    const long long now_ = __builtin_amdgcn_s_memtime();     \
    dg[k] += now_ - t_;                                      \
    t_ = now_;                                               \
  } while (0)
#else
#define FPSP_STAMP(k)
#endif

__device__ __forceinline__ float wave_min_f_dpp(float v) { return -wave_max_dpp(-v); }

// v_writelane_b32 (hipcc 7.2 has no builtin; the LLVM intrinsic is reachable by name and, unlike inline asm, seen by the
// hazard recogniser)
extern "C" __device__ int fpsp_writelane(int value, int lane, int old) __asm("llvm.amdgcn.writelane.i32");

// two rows at once: the chains alternate, so a row's next level is two instructions behind its previous one (the other row's
// instruction + s_nop 0 = the two wait states a DPP read needs after the VALU write of its source)
__device__ __forceinline__ void wave_max_chain2(float a, float b, float& ra, float& rb) {
  asm volatile(
      "s_nop 1\n\t"
      "v_max_f32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_max_f32_dpp %2, %2, %2 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %3, %3, %3 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_max_f32_dpp %2, %2, %2 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %3, %3, %3 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_max_f32_dpp %2, %2, %2 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %3, %3, %3 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_max_f32_dpp %2, %2, %2 row_bcast:15 row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %3, %3, %3 row_bcast:15 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_max_f32_dpp %2, %2, %2 row_bcast:31 row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %3, %3, %3 row_bcast:31 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_readlane_b32 %0, %2, 63\n\t"
      "v_readlane_b32 %1, %3, 63\n\t"
      "s_nop 1"
      : "=s"(ra), "=s"(rb), "+v"(a), "+v"(b));
}

// the same over lanes 0..15 (DPP row 0): four levels, the result in lane 15
__device__ __forceinline__ float wave_max16_chain(float v) {
  float r;
  asm volatile(
      "s_nop 1\n\t"
      "v_max_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_f32_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_f32_dpp %1, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_f32_dpp %1, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_readlane_b32 %0, %1, 15\n\t"
      "s_nop 1"
      : "=s"(r), "+v"(v));
  return r;
}

// exclusive prefix sum over the 256 threads of the block (red: 4 ints of LDS); total via *tot
__device__ __forceinline__ int block_excl_scan(int v, int* red, int tid, int* tot) {
  const int lane = tid & 63, wave = tid >> 6;
  int inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  if (lane == 63) red[wave] = inc;
  __syncthreads();
  int base = 0, all = 0;
#pragma unroll
  for (int w = 0; w < FPSP_T / 64; ++w) {
    const int t = red[w];
    if (w < wave) base += t;
    all += t;
  }
  *tot = all;
  __syncthreads();
  return base + inc - v;
}

template <int PER>
__global__ __launch_bounds__(FPSP_T) void fps_pruned_kernel(FpsPrunedArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int Npad = a.Npad, N = a.N, L = a.L, R = Npad >> 6;
  float* sx = lds;                      // setup: coordinates by ORIGINAL index; chain: by sorted position
  float* sy = sx + Npad;
  float* sz = sy + Npad;
  float* sd = sz + Npad;                // chain: running distances by sorted position; setup: the histogram
  int* hist = reinterpret_cast<int*>(sd);
  uint16_t* permA = reinterpret_cast<uint16_t*>(sd + Npad);   // chain: original index of a sorted position
  uint16_t* permB = permA + Npad;
  __shared__ float s_bb[4][6];
  __shared__ int s_red[4];
  __shared__ int s_med[64], s_eql[64], s_eqc[64], s_lcur[64], s_rcur[64];
  __shared__ float s_rowbb[6][64];
  __shared__ float s_rowmax[64];
  __shared__ int s_rowarg[64];
  __shared__ int s_pf0;
  __shared__ unsigned long long s_slot[2][4];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* xb = a.x.p + (int64_t)b * a.x.bs;
  int start = a.start ? a.start[b] : 0;
  start = start < 0 ? 0 : (start >= N ? N - 1 : start);

  // ---- load: padding entries are copies of the last point (they live where it lives and never win: dist = -1) ----
  float lx = INFINITY, ly = INFINITY, lz = INFINITY, hx = -INFINITY, hy = -INFINITY, hz = -INFINITY;
  for (int i = tid; i < Npad; i += FPSP_T) {
    const float* p = xb + (int64_t)(i < N ? i : N - 1) * a.x.ps;
    const float x = p[0], y = p[a.x.cs], z = p[2 * a.x.cs];
    sx[i] = x, sy[i] = y, sz[i] = z;
    permA[i] = (uint16_t)i;
    lx = fminf(lx, x), ly = fminf(ly, y), lz = fminf(lz, z);      // (fminf / fmaxf skip NaN operands)
    hx = fmaxf(hx, x), hy = fmaxf(hy, y), hz = fmaxf(hz, z);
  }
  lx = wave_min_f_dpp(lx), ly = wave_min_f_dpp(ly), lz = wave_min_f_dpp(lz);
  hx = wave_max_dpp(hx), hy = wave_max_dpp(hy), hz = wave_max_dpp(hz);
  if (lane == 0) s_bb[wave][0] = lx, s_bb[wave][1] = ly, s_bb[wave][2] = lz, s_bb[wave][3] = hx, s_bb[wave][4] = hy, s_bb[wave][5] = hz;
  __syncthreads();
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    lx = fminf(lx, s_bb[w][0]), ly = fminf(ly, s_bb[w][1]), lz = fminf(lz, s_bb[w][2]);
    hx = fmaxf(hx, s_bb[w][3]), hy = fmaxf(hy, s_bb[w][4]), hz = fmaxf(hz, s_bb[w][5]);
  }
  // split dimension per level: always the longest side of the (uniformly halved) cell
  float ex = hx - lx, ey = hy - ly, ez = hz - lz;
  const int lgN = 6 + L;

  uint16_t* cur = permA;
  uint16_t* nxt = permB;
  for (int l = 0; l < L; ++l) {
    int d = 0;
    if (ey > ex && ey >= ez) d = 1;
    else if (ez > ex && ez > ey) d = 2;
    if (d == 0) ex *= 0.5f;
    else if (d == 1) ey *= 0.5f;
    else ez *= 0.5f;
    const float* sc = d == 0 ? sx : (d == 1 ? sy : sz);
    const float lo = d == 0 ? lx : (d == 1 ? ly : lz), hi = d == 0 ? hx : (d == 1 ? hy : hz);
    const int sh = lgN - l;                     // segment = position >> sh ; G = 1 << sh points per segment
    const int G = 1 << sh, nseg = 1 << l;
    const int bins = G < 1024 ? G : 1024;       // nseg * bins <= Npad entries: the histogram fits the distance array
    const float scale = (hi > lo) ? (float)bins / (hi - lo) : 0.f;
    for (int i = tid; i < nseg * bins; i += FPSP_T) hist[i] = 0;
    if (tid < nseg) s_eqc[tid] = 0, s_lcur[tid] = 0, s_rcur[tid] = 0, s_med[tid] = 0, s_eql[tid] = 0;
    __syncthreads();
    int binr[FPSP_MAXJ];
#pragma unroll
    for (int j = 0; j < FPSP_MAXJ; ++j) {
      const int i = tid + j * FPSP_T;
      binr[j] = 0;
      if (i < Npad) {
        const float c = sc[cur[i]];
        int bq = (int)((c - lo) * scale);       // NaN -> 0, out of range saturates
        bq = bq < 0 ? 0 : (bq > bins - 1 ? bins - 1 : bq);
        binr[j] = bq;
        atomicAdd(&hist[(i >> sh) * bins + bq], 1);
      }
    }
    __syncthreads();
    {  // the bin that holds the G/2-th point of its segment, and how many of its points still go left
      const int entries = nseg * bins;
      const int per = (entries + FPSP_T - 1) / FPSP_T;     // <= 16 consecutive entries per thread
      const int e0 = tid * per;
      int sum = 0;
      for (int q = 0; q < per; ++q)
        if (e0 + q < entries) sum += hist[e0 + q];
      int tot;
      int pre = block_excl_scan(sum, s_red, tid, &tot);
      for (int q = 0; q < per; ++q) {
        const int e = e0 + q;
        if (e < entries) {
          const int cnt = hist[e];
          const int seg = e / bins;
          const int inseg = pre - (seg << sh);
          if (cnt > 0 && inseg < (G >> 1) && (G >> 1) <= inseg + cnt) s_med[seg] = e - seg * bins, s_eql[seg] = (G >> 1) - inseg;
          pre += cnt;
        }
      }
    }
    __syncthreads();
    // two-sided scatter; a wavefront's 64 positions lie in ONE segment, so the cursors move once per wave and side
    int tick[FPSP_MAXJ];
#pragma unroll
    for (int j = 0; j < FPSP_MAXJ; ++j) {
      const int i = tid + j * FPSP_T;
      tick[j] = 0;
      if (i < Npad) {
        const int seg = i >> sh;
        const bool eq = binr[j] == s_med[seg];
        const unsigned long long m = __builtin_amdgcn_ballot_w64(eq);
        int base = 0;
        if (lane == 0 && m) base = atomicAdd(&s_eqc[seg], __builtin_popcountll(m));
        base = __builtin_amdgcn_readfirstlane(base);
        tick[j] = base + __builtin_popcountll(m & ((1ull << lane) - 1ull));
      }
    }
#pragma unroll
    for (int j = 0; j < FPSP_MAXJ; ++j) {
      const int i = tid + j * FPSP_T;
      if (i < Npad) {
        const int seg = i >> sh;
        const int med = s_med[seg];
        const bool left = binr[j] < med || (binr[j] == med && tick[j] < s_eql[seg]);
        const unsigned long long ml = __builtin_amdgcn_ballot_w64(left);
        int bl = 0, br = 0;
        if (lane == 0) {
          const int nl = __builtin_popcountll(ml);
          if (nl) bl = atomicAdd(&s_lcur[seg], nl);
          if (nl < 64) br = atomicAdd(&s_rcur[seg], 64 - nl);
        }
        bl = __builtin_amdgcn_readfirstlane(bl);
        br = __builtin_amdgcn_readfirstlane(br);
        const unsigned long long below = (1ull << lane) - 1ull;
        const int pos = left ? (seg << sh) + bl + __builtin_popcountll(ml & below)
                             : (seg << sh) + (G >> 1) + br + __builtin_popcountll(~ml & below);
        nxt[pos] = cur[i];
      }
    }
    __syncthreads();
    uint16_t* t = cur;
    cur = nxt;
    nxt = t;
  }

  // ---- coordinates into sorted order (through registers) as ROW RECORDS: row r = 256 floats x[64] y[64] z[64] dist[64],
  // so a row is one base address + immediate offsets; running distances; original indices ----
  float* rec = lds;
  {
    float X[FPSP_MAXJ], Y[FPSP_MAXJ], Z[FPSP_MAXJ];
    int O[FPSP_MAXJ];
#pragma unroll
    for (int j = 0; j < FPSP_MAXJ; ++j) {
      const int i = tid + j * FPSP_T;
      O[j] = 0, X[j] = Y[j] = Z[j] = 0.f;
      if (i < Npad) {
        O[j] = cur[i];
        X[j] = sx[O[j]], Y[j] = sy[O[j]], Z[j] = sz[O[j]];
      }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < FPSP_MAXJ; ++j) {
      const int i = tid + j * FPSP_T;
      if (i < Npad) {
        float* r = rec + (i >> 6) * 256 + (i & 63);
        r[0] = X[j], r[64] = Y[j], r[128] = Z[j];
        r[192] = O[j] < N ? 1e10f : -1.f;
        permA[i] = (uint16_t)O[j];
        if (O[j] == start) s_pf0 = i;
      }
    }
    __syncthreads();
  }
  // ---- per row: bounding box, initial maximum (1e10 / -1 for a row of padding) and the lane of its lowest original index ----
  for (int r = wave; r < R; r += FPSP_T / 64) {
    const float* rr = rec + r * 256 + lane;
    const float x = rr[0], y = rr[64], z = rr[128];
    const float v0 = wave_min_f_dpp(x), v1 = wave_min_f_dpp(y), v2 = wave_min_f_dpp(z);
    const float v3 = wave_max_dpp(x), v4 = wave_max_dpp(y), v5 = wave_max_dpp(z);
    const int o = permA[r * 64 + lane];
    const int mo = wave_min_dpp_i32(o < N ? o : 0x7fffffff);
    const unsigned long long who = __builtin_amdgcn_ballot_w64(o == mo);
    if (lane == 0) {
      s_rowbb[0][r] = v0, s_rowbb[1][r] = v1, s_rowbb[2][r] = v2, s_rowbb[3][r] = v3, s_rowbb[4][r] = v4, s_rowbb[5][r] = v5;
      s_rowmax[r] = mo != 0x7fffffff ? 1e10f : -1.f;
      s_rowarg[r] = who ? __builtin_ctzll(who) : 0;
    }
  }
  if (tid < 8) s_slot[tid >> 2][tid & 3] = 0ull;
  __syncthreads();

  // ---- the chain: wave w owns rows w, w + 4, ... in registers; lane j = its j-th row ----
  __builtin_amdgcn_s_setprio(3);
  const uint16_t* orig = permA;
  uint16_t* pfs = permB;                  // the picks as sorted positions; turned into original indices at the end
  float px[PER], py[PER], pz[PER], dd[PER];
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const float* rr = rec + (wave + 4 * j) * 256 + lane;
    px[j] = rr[0], py[j] = rr[64], pz[j] = rr[128], dd[j] = rr[192];
  }
  const bool live = lane < PER;
  const int myrow = wave + 4 * (live ? lane : 0);
  const float blx = s_rowbb[0][myrow], bly = s_rowbb[1][myrow], blz = s_rowbb[2][myrow];
  const float bhx = s_rowbb[3][myrow], bhy = s_rowbb[4][myrow], bhz = s_rowbb[5][myrow];
  int rmax_i = __builtin_bit_cast(int, live ? s_rowmax[myrow] : -2.f);     // lane j: row j's largest running distance ...
  int rarg = live ? s_rowarg[myrow] : 0;                                    // ... and the lane that holds it
  int pf = s_pf0;
#ifdef FPSP_DIAG
  long long dg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long t_ = __builtin_amdgcn_s_memtime();
  const long long t0_ = t_;
#endif
  for (int s = 0; s < a.S; ++s) {
    const float* pc = rec + (pf >> 6) * 256 + (pf & 63);
    const float cx = pc[0], cy = pc[64], cz = pc[128];
    if (tid == 0) pfs[s] = (uint16_t)pf;
    // lower bound of |p - c|^2 over the row's box, in the distance's own operations
    const float ddx = fmaxf(fmaxf(blx - cx, cx - bhx), 0.f), ddy = fmaxf(fmaxf(bly - cy, cy - bhy), 0.f),
                ddz = fmaxf(fmaxf(blz - cz, cz - bhz), 0.f);
    const float bound = (ddx * ddx + ddy * ddy) + ddz * ddz;
    const unsigned act = (unsigned)__builtin_amdgcn_ballot_w64(bound < __builtin_bit_cast(float, rmax_i));   // lanes >= PER: -2
    FPSP_STAMP(0);
#ifdef FPSP_DIAG
    dg[5] += __builtin_popcount(act);
#endif
    int tiec = 0;          // the largest number of lanes that tied for a row's maximum in this step (1: no tie anywhere)
#define FPSP_UPD(j, n)                                                                                                       \
  float n;                                                                                                                   \
  {                                                                                                                          \
    const float dx = px[j] - cx, dy = py[j] - cy, dz = pz[j] - cz;                                                           \
    const float d = (dx * dx + dy * dy) + dz * dz;                                                                           \
    /* v_min_f32 returns the other operand for a NaN: exactly (d < dist) ? d : dist, as fps_kernel */                        \
    asm("v_min_f32_e32 %0, %1, %2" : "=v"(n) : "v"(d), "v"(dd[j]));                                                          \
    dd[j] = n;                                                                                                               \
  }
#define FPSP_REC(j, n, m)                                                                                                    \
  {                                                                                                                          \
    const unsigned long long t = __builtin_amdgcn_ballot_w64(n == m);                                                        \
    const int c = __builtin_popcountll(t);                                                                                   \
    tiec = tiec > c ? tiec : c;                                                                                              \
    rmax_i = fpsp_writelane(__builtin_bit_cast(int, m), (j), rmax_i);                                                        \
    rarg = fpsp_writelane(__builtin_ctzll(t), (j), rarg);                                                                    \
  }
#define FPSP_ROW(j)                                                                                                          \
  {                                                                                                                          \
    FPSP_UPD(j, n_)                                                                                                          \
    const float m_ = wave_max_chain(n_);                                                                                     \
    FPSP_REC(j, n_, m_)                                                                                                      \
  }
  // two rows with their reductions interleaved (the DPP wait states of one are the other's instructions)
#define FPSP_ROW2(j, k)                                                                                                      \
  {                                                                                                                          \
    FPSP_UPD(j, na_)                                                                                                         \
    FPSP_UPD(k, nb_)                                                                                                         \
    float ma_, mb_;                                                                                                          \
    wave_max_chain2(na_, nb_, ma_, mb_);                                                                                     \
    FPSP_REC(j, na_, ma_)                                                                                                    \
    FPSP_REC(k, nb_, mb_)                                                                                                    \
  }
#define FPSP_PAIR(j)                                                                                                         \
  if ((j) + 1 < PER) {                                                                                                       \
    const unsigned sw_ = (act >> (j)) & 3u;                                                                                  \
    if (sw_ == 3u) FPSP_ROW2(j, (j) + 1 < PER ? (j) + 1 : (j))                                                               \
    else if (sw_ == 1u) FPSP_ROW(j)                                                                                          \
    else if (sw_ == 2u) FPSP_ROW((j) + 1 < PER ? (j) + 1 : (j))                                                              \
  } else if (act & (1u << (j))) FPSP_ROW(j)
#pragma unroll
    for (int j4 = 0; j4 < PER; j4 += 4) {
      if (PER <= 4 || (act & (0xfu << j4))) {
        FPSP_PAIR(j4)
        if (j4 + 2 < PER) { FPSP_PAIR(j4 + 2) }
      }
    }
    if (__builtin_expect(tiec > 1, 0)) {
      // equidistant points inside one of the rows just updated: that row's arg-max is the lowest ORIGINAL index among the
      // tied lanes (the row's maximum is in lane j of the table)
#pragma unroll
      for (int j = 0; j < PER; ++j)
        if (act & (1u << j)) {
          const float m = __builtin_bit_cast(float, __builtin_amdgcn_readlane(rmax_i, j));
          const int o = orig[(wave + 4 * j) * 64 + lane];
          const int mo = wave_min_dpp_i32(dd[j] == m ? o : 0x7fffffff);
          rarg = fpsp_writelane(__builtin_ctzll(__builtin_amdgcn_ballot_w64(dd[j] == m && o == mo)), j, rarg);
        }
    }
#undef FPSP_PAIR
#undef FPSP_ROW2
#undef FPSP_ROW
#undef FPSP_REC
#undef FPSP_UPD
    FPSP_STAMP(1);
    // the wave's best row
    const float rmax = __builtin_bit_cast(float, rmax_i);
    const float g = wave_max16_chain(rmax);
    const unsigned tg = (unsigned)__builtin_amdgcn_ballot_w64(rmax == g) & 0xffffu;
    int jw = __builtin_ctz(tg | 0x10000u);
    if (__builtin_expect((tg & (tg - 1)) != 0, 0)) {         // rows tie: the lowest original index among their arg-max points
      const int o = (live && rmax == g) ? (int)orig[(wave + 4 * lane) * 64 + rarg] : 0x7fffffff;
      const int mo = wave_min_dpp_i32(o);
      jw = __builtin_ctzll(__builtin_amdgcn_ballot_w64(o == mo));
    }
    const int mypos = ((wave + 4 * jw) << 6) + __builtin_amdgcn_readlane(rarg, jw);
    // exchange: (value, position | stamp) per wave, double-buffered by step parity; a record is one 8-byte LDS write, so a
    // reader that sees the stamp sees the value. Nobody can be a whole step ahead: posting step s + 1 needs everyone's s.
    FPSP_STAMP(2);
    const unsigned stamp = (unsigned)(s + 1) << 12;
    // (an LDS-address-space pointer: through a generic volatile pointer these become flat_load / flat_store with a full
    // vmcnt(0) drain each — 4 x ~250 cycles per step)
    fpsp_lds_u64* slot = (fpsp_lds_u64*)s_slot[s & 1];
    if (lane == 0) slot[wave] = (unsigned long long)__builtin_bit_cast(unsigned, g) | ((unsigned long long)((unsigned)mypos | stamp) << 32);
    unsigned long long r0, r1, r2, r3;
    unsigned y0, y1, y2, y3;
    for (;;) {
#ifdef FPSP_DIAG
      dg[6] += 1;
#endif
      r0 = slot[0], r1 = slot[1], r2 = slot[2], r3 = slot[3];
      y0 = (unsigned)(r0 >> 32), y1 = (unsigned)(r1 >> 32), y2 = (unsigned)(r2 >> 32), y3 = (unsigned)(r3 >> 32);
      if ((((y0 ^ stamp) | (y1 ^ stamp)) | ((y2 ^ stamp) | (y3 ^ stamp))) < 4096u) break;
    }
    FPSP_STAMP(3);
    const float v0 = __builtin_bit_cast(float, (unsigned)r0), v1 = __builtin_bit_cast(float, (unsigned)r1),
                v2 = __builtin_bit_cast(float, (unsigned)r2), v3 = __builtin_bit_cast(float, (unsigned)r3);
    const float gm = fmaxf(fmaxf(v0, v1), fmaxf(v2, v3));
    const int p0 = y0 & 4095, p1 = y1 & 4095, p2 = y2 & 4095, p3 = y3 & 4095;
    const int nt = (v0 == gm) + (v1 == gm) + (v2 == gm) + (v3 == gm);
    int np = v0 == gm ? p0 : (v1 == gm ? p1 : (v2 == gm ? p2 : p3));
    if (__builtin_expect(nt > 1, 0)) {                       // waves tie: the lowest original index
      const int o0 = v0 == gm ? (int)orig[p0] : 0x7fffffff, o1 = v1 == gm ? (int)orig[p1] : 0x7fffffff;
      const int o2 = v2 == gm ? (int)orig[p2] : 0x7fffffff, o3 = v3 == gm ? (int)orig[p3] : 0x7fffffff;
      int bo = o0;
      np = p0;
      if (o1 < bo) bo = o1, np = p1;
      if (o2 < bo) bo = o2, np = p2;
      if (o3 < bo) bo = o3, np = p3;
    }
    pf = __builtin_amdgcn_readfirstlane(np);
    FPSP_STAMP(4);
  }
#ifdef FPSP_DIAG
  dg[7] = __builtin_amdgcn_s_memtime() - t0_;
  if (lane == 0)
    for (int k = 0; k < 8; ++k) a.diag[((int64_t)b * 4 + wave) * 8 + k] = dg[k];
#endif
  __syncthreads();
  int32_t* out = a.out + (int64_t)b * a.S;
  for (int s = tid; s < a.S; s += FPSP_T) out[s] = orig[pfs[s]];
}

#ifdef FPSP_DIAG
static long long* g_fpsp_diag = nullptr;
#endif

size_t fps_pruned_lds_bytes(int Npad) { return (size_t)Npad * (4 * sizeof(float) + 2 * sizeof(uint16_t)); }

int fps_pruned_launch(const char* nm, const float* xyz, int64_t x_bs, int64_t x_ps, int64_t x_cs, int B, int N, int S,
                      const int32_t* start, int32_t* out, void* stream) {
  PC3D_REQUIRE(B >= 0 && N >= 1 && S >= 1, "%s: bad sizes B=%d N=%d S=%d", nm, B, N, S);
  PC3D_REQUIRE(N <= FPSP_MAXN, "%s: N=%d exceeds %d (16 rows of 64 points per wavefront)", nm, N, FPSP_MAXN);
  PC3D_REQUIRE(S <= 4095, "%s: S=%d exceeds 4095 (the exchange's stamp)", nm, S);
  if (B == 0) return PC3D_OK;
  PC3D_REQUIRE(xyz && out, "%s: null pointer", nm);
  int L = 2;                      // at least four rows: one per wavefront
  while ((64 << L) < N) ++L;
#ifdef FPSP_DIAG
  FpsPrunedArgs a{{xyz, x_bs, x_ps, x_cs}, N, S, start, out, 64 << L, L, g_fpsp_diag};
#else
  FpsPrunedArgs a{{xyz, x_bs, x_ps, x_cs}, N, S, start, out, 64 << L, L};
#endif
  size_t lds = fps_pruned_lds_bytes(a.Npad);
  if (lds < (size_t)S * 2 + (size_t)a.Npad * 18) lds = (size_t)S * 2 + (size_t)a.Npad * 18;     // the picks (S > Npad: repeats)
  hipStream_t st = as_stream(stream);
#define PC3D_FPSP(PERV)                                                                                                   \
  do {                                                                                                                    \
    if (lds > 48 * 1024) {                                                                                                \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fps_pruned_kernel<PERV>),                          \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                           \
      if (e != hipSuccess) {                                                                                              \
        set_error("%s: LDS opt-in failed: %s", nm, hipGetErrorString(e));                                                 \
        return (int)e;                                                                                                    \
      }                                                                                                                   \
    }                                                                                                                     \
    hipLaunchKernelGGL(fps_pruned_kernel<PERV>, dim3(B), dim3(FPSP_T), lds, st, a);                                       \
  } while (0)
  switch (L) {
    case 2: PC3D_FPSP(1); break;
    case 3: PC3D_FPSP(2); break;
    case 4: PC3D_FPSP(4); break;
    case 5: PC3D_FPSP(8); break;
    default: PC3D_FPSP(16); break;
  }
#undef PC3D_FPSP
  PC3D_LAUNCH_CHECK(nm);
  return PC3D_OK;
}

}  // namespace pc3d

#ifdef FPSP_DIAG
extern "C" void fpsp_set_diag(long long* p) { pc3d::g_fpsp_diag = p; }
#endif

extern "C" int pc3d_fps_pruned_f32(const float* xyz, int64_t x_bs, int64_t x_ps, int64_t x_cs, int B, int N, int S,
                                   const int32_t* start, int32_t* out, void* stream) {
  return pc3d::fps_pruned_launch("pc3d_fps_pruned_f32", xyz, x_bs, x_ps, x_cs, B, N, S, start, out, stream);
}
