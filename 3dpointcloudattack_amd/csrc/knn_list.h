// Sorted K-list kept ACROSS the lanes of a wave (lane j = j-th nearest so far) — shared by the xyz kNN (knn.hip) and
// the feature-space kNN of DGCNN (dgcnn.hip). gfx950 / wave64.
#pragma once
#include "pc3d_common.h"

namespace pc3d {

__device__ __forceinline__ float readlane_f(float v, int l) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}

// lexicographic (d, idx) bitonic sort of one value per lane, ascending over lanes
__device__ __forceinline__ void wave_sort_pairs(float& d, int& i, int lane) {
#pragma unroll
  for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
      const float od = __shfl_xor(d, j, 64);
      const int oi = __shfl_xor(i, j, 64);
      const bool keep_min = (((lane & j) == 0) == ((lane & k) == 0));
      const bool other_less = (od < d) || (od == d && oi < i);
      const bool other_more = (od > d) || (od == d && oi > i);
      const bool take = keep_min ? other_less : other_more;
      d = take ? od : d;
      i = take ? oi : i;
    }
  }
}

// ---- the same sort without a single LDS round trip (ds_bpermute + s_waitcnt per exchange made the seed sorts of the
// kNN kernels cost as much as their MFMA work): partners through DPP (quad_perm / row_half_mirror / row_mirror /
// row_ror / bank-masked row shifts) and v_permlane{16,32}_swap, keys compared as ONE signed 64-bit integer
// (order-preserving distance bits : index), and the bitonic network in its uniform-direction form (first exchange of
// every merge against lane ^ (k-1), the rest against lane ^ j: the LOWER lane always keeps the minimum), so the
// "who takes" predicate is the compare mask XNOR a constant: v_cmp_lt_i64, two s_xor_b32, two v_cndmask per exchange.
__device__ __forceinline__ int knn_ord(float f) {   // monotone float <-> signed int, an involution
  const int b = __builtin_bit_cast(int, f);
  return b ^ ((b >> 31) & 0x7fffffff);
}
__device__ __forceinline__ float knn_unord(int i) { return __builtin_bit_cast(float, i ^ ((i >> 31) & 0x7fffffff)); }

using knn_u32x2 = __attribute__((ext_vector_type(2))) unsigned;

template <int M>
__device__ __forceinline__ int lane_xor(int x, int lane) {   // value of lane ^ M
  if constexpr (M == 1) return __builtin_amdgcn_update_dpp(x, x, 0xB1, 0xf, 0xf, false);        // quad_perm [1,0,3,2]
  else if constexpr (M == 2) return __builtin_amdgcn_update_dpp(x, x, 0x4E, 0xf, 0xf, false);   // quad_perm [2,3,0,1]
  else if constexpr (M == 3) return __builtin_amdgcn_update_dpp(x, x, 0x1B, 0xf, 0xf, false);   // quad_perm [3,2,1,0]
  else if constexpr (M == 4) {
    const int t = __builtin_amdgcn_update_dpp(x, x, 0x104, 0xf, 0x5, false);                    // row_shl:4, banks 0, 2
    return __builtin_amdgcn_update_dpp(t, x, 0x114, 0xf, 0xa, false);                           // row_shr:4, banks 1, 3
  } else if constexpr (M == 7) return __builtin_amdgcn_update_dpp(x, x, 0x141, 0xf, 0xf, false);   // row_half_mirror
  else if constexpr (M == 8) return __builtin_amdgcn_update_dpp(x, x, 0x128, 0xf, 0xf, false);     // row_ror:8
  else if constexpr (M == 15) return __builtin_amdgcn_update_dpp(x, x, 0x140, 0xf, 0xf, false);    // row_mirror
  else if constexpr (M == 16) {
    const knn_u32x2 r = __builtin_amdgcn_permlane16_swap((unsigned)x, (unsigned)x, false, false);
    const unsigned r0 = r[0], r1 = r[1];     // (copied to scalars first: a swizzle fed to a cast reads element 0)
    return (int)((lane & 16) ? r0 : r1);
  } else if constexpr (M == 32) {
    const knn_u32x2 r = __builtin_amdgcn_permlane32_swap((unsigned)x, (unsigned)x, false, false);
    const unsigned r0 = r[0], r1 = r[1];
    return (int)((lane & 32) ? r0 : r1);
  } else if constexpr (M == 31) return lane_xor<16>(lane_xor<15>(x, lane), lane);
  else {
    static_assert(M == 63, "lane_xor: unsupported mask");
    return lane_xor<32>(lane_xor<16>(lane_xor<15>(x, lane), lane), lane);
  }
}

template <int M, unsigned long long LOWER>
__device__ __forceinline__ void knn_cmpx(int& key, int& idx, int lane) {
  const int ok = lane_xor<M>(key, lane), oi = lane_xor<M>(idx, lane);
  const long long self = ((long long)key << 32) | (unsigned)idx, other = ((long long)ok << 32) | (unsigned)oi;
  const unsigned long long less = __builtin_amdgcn_ballot_w64(other < self);
  const bool take = __builtin_amdgcn_inverse_ballot_w64(~(less ^ LOWER));   // lower lane: other < self; upper: other > self
  key = take ? ok : key;
  idx = take ? oi : idx;
}

// ascending (key, idx) over the lanes; idx >= 0 and all (key, idx) pairs distinct
__device__ __forceinline__ void wave_sort_pairs_dpp(int& key, int& idx, int lane) {
  constexpr unsigned long long L1 = 0x5555555555555555ull, L2 = 0x3333333333333333ull, L4 = 0x0f0f0f0f0f0f0f0full,
                               L8 = 0x00ff00ff00ff00ffull, L16 = 0x0000ffff0000ffffull, L32 = 0x00000000ffffffffull;
  knn_cmpx<1, L1>(key, idx, lane);
  knn_cmpx<3, L2>(key, idx, lane), knn_cmpx<1, L1>(key, idx, lane);
  knn_cmpx<7, L4>(key, idx, lane), knn_cmpx<2, L2>(key, idx, lane), knn_cmpx<1, L1>(key, idx, lane);
  knn_cmpx<15, L8>(key, idx, lane), knn_cmpx<4, L4>(key, idx, lane), knn_cmpx<2, L2>(key, idx, lane), knn_cmpx<1, L1>(key, idx, lane);
  knn_cmpx<31, L16>(key, idx, lane), knn_cmpx<8, L8>(key, idx, lane), knn_cmpx<4, L4>(key, idx, lane);
  knn_cmpx<2, L2>(key, idx, lane), knn_cmpx<1, L1>(key, idx, lane);
  knn_cmpx<63, L32>(key, idx, lane), knn_cmpx<16, L16>(key, idx, lane), knn_cmpx<8, L8>(key, idx, lane);
  knn_cmpx<4, L4>(key, idx, lane), knn_cmpx<2, L2>(key, idx, lane), knn_cmpx<1, L1>(key, idx, lane);
}

// wave-wide minimum of one float per lane (DPP row shifts + row broadcasts; min is idempotent, so overlapping
// contributions are harmless); the result is returned as a wave-uniform value
__device__ __forceinline__ float wave_min_dpp(float v) {
#define PC3D_DPP_MIN(ctrl)                                                                              \
  v = fminf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v),        \
                                                                     __builtin_bit_cast(int, v), ctrl, 0xf, 0xf, false)))
  PC3D_DPP_MIN(0x111);  // row_shr:1
  PC3D_DPP_MIN(0x112);  // row_shr:2
  PC3D_DPP_MIN(0x114);  // row_shr:4
  PC3D_DPP_MIN(0x118);  // row_shr:8  -> lane 15 of every row holds the row minimum
  PC3D_DPP_MIN(0x142);  // row_bcast:15
  PC3D_DPP_MIN(0x143);  // row_bcast:31 -> lane 63 holds the wave minimum
#undef PC3D_DPP_MIN
  return readlane_f(v, 63);
}

// wave-wide maximum of one int per lane, the same way (integer: distance KEYS may be denormal bit patterns)
__device__ __forceinline__ int wave_max_i32(int v) {
#define PC3D_DPP_MAXI(ctrl) v = max(v, __builtin_amdgcn_update_dpp(v, v, ctrl, 0xf, 0xf, false))
  PC3D_DPP_MAXI(0x111);
  PC3D_DPP_MAXI(0x112);
  PC3D_DPP_MAXI(0x114);
  PC3D_DPP_MAXI(0x118);
  PC3D_DPP_MAXI(0x142);
  PC3D_DPP_MAXI(0x143);
#undef PC3D_DPP_MAXI
  return __builtin_amdgcn_readlane(v, 63);
}

// v_writelane_b32 (hipcc 7.2 has no builtin; the LLVM intrinsic is reachable by name, and unlike inline asm it is seen
// by the hazard recogniser)
extern "C" __device__ int knn_writelane(int value, int lane, int old) __asm("llvm.amdgcn.writelane.i32");

// insert (dc, ic) — wave-uniform — AFTER every entry <= dc (candidates arrive in ascending index order, so equal
// distances keep the lower index first); the entry in lane 63 falls off. The list is sorted, so the lanes with
// ld <= dc are exactly the lanes below the insertion point: the compare's 64-bit mask IS the "keep" predicate of the
// shift (one v_cmp, two DPP moves, two v_cndmask on that mask, two v_writelane for the new entry — 7 VALU).
// (Round 2, measured: a 5-VALU form — EXEC narrowed to the lanes above the candidate, in-place DPP shifts, v_writelane —
// in inline assembly changed neither kNN kernel's time: the insertions are bound by the latency of their
// VALU -> SALU -> VALU hops and taken branches, not by VALU issue.)
__device__ __forceinline__ void knn_list_insert(float& ld, int& li, float dc, int ic, int lane) {
  const bool k = ld <= dc;
  const int pos = __builtin_popcountll(__builtin_amdgcn_ballot_w64(k));
  // lane l <- lane l-1 (v_mov_b32_dpp wave_shr:1)
  const float sd = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, ld), 0x138, 0xf, 0xf, false));
  const int si = __builtin_amdgcn_update_dpp(0, li, 0x138, 0xf, 0xf, false);
  ld = k ? ld : sd;
  li = k ? li : si;
  if (pos < 64) {   // uniform (always true when dc beats the list's last entry, which callers check)
    ld = __builtin_bit_cast(float, knn_writelane(__builtin_bit_cast(int, dc), pos, __builtin_bit_cast(int, ld)));
    li = knn_writelane(ic, pos, li);
  }
}

// ---- the same list on INTEGER keys (knn_ord of the distance). With float distances the "still below the threshold?"
// re-check of a candidate is a VALU compare of two wave-uniform values -> VCC -> branch, and the threshold lives in a
// VGPR; gfx950 has no scalar float compare. On keys it is s_cmp_lt_i32 on two SGPRs, the threshold is the SGPR that
// v_readlane returned, and the "position < 64" guard goes (a candidate below entry K-1 lands at a position < K).
constexpr int kKnnInfKey = 0x7f800000;   // knn_ord(+inf)
// total order on distances: NaN -> +inf (never a neighbour), -0 -> +0
__device__ __forceinline__ int knn_key(float d) { return knn_ord(fminf(d, __builtin_inff()) + 0.f); }

// One candidate of a 64-candidate step: lane c's key is read into an SGPR, the list lanes with lk <= kc keep their entry,
// the others take their lower neighbour's (select and wave_shr:1 shift are ONE v_cndmask_b32_dpp per array: lane 0 has no
// source lane and keeps its own) and the new entry goes to lane popcount(keep) (< K, since kc is below the key of lane
// K-1). Nine instructions, hand-written: the compiler's form of the same step took 13 (v_mov_b32_dpp + v_cndmask_b32
// pairs, three SALU instructions for mask &= mask - 1), and at four waves per SIMD these loops are bound by instruction
// issue — 27 -> 22 -> 18 -> 15 instructions per insertion moved knn_wave_kernel 82 -> 74 -> 66 -> 6x us (B=32, N=1024,
// K=20) and knn_feat_kernel 141 -> 133 -> 121 -> 1xx us.
// Hazards (inline code is not seen by the compiler's hazard recogniser), gfx950:
//   * VALU-written SGPR (kc, from v_readlane) read by a VALU: 2 wait states — the two SALU instructions in between;
//   * VALU-written VGPR read through DPP: 2 wait states — lk / li were last written by the v_writelane of an earlier
//     step or by the seed sort, and the step's first three instructions separate them in any case;
//   * M0 is written by SALU (no hazard before v_writelane) and is CLOBBERED: v_writelane takes an SGPR value only with
//     M0 as the lane select (one-SGPR constant-bus rule). It cannot be named in the clobber list (reserved); the
//     compiler never keeps a value in M0 across other code.
__device__ __forceinline__ void knn_list_step(int& lk, int& li, unsigned long long& mask, int key, int c, int jbase) {
  int kc, ic;
  asm volatile(
      "v_readlane_b32 %[kc], %[key], %[c]\n\t"
      "s_add_i32 %[ic], %[jb], %[c]\n\t"
      "s_bitset0_b64 %[mask], %[c]\n\t"
      "v_cmp_ge_i32_e32 vcc, %[kc], %[lk]\n\t"
      "v_cndmask_b32_dpp %[lk], %[lk], %[lk], vcc wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      "v_cndmask_b32_dpp %[li], %[li], %[li], vcc wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      "s_bcnt1_i32_b64 m0, vcc\n\t"
      "v_writelane_b32 %[lk], %[kc], m0\n\t"
      "v_writelane_b32 %[li], %[ic], m0"
      : [lk] "+v"(lk), [li] "+v"(li), [mask] "+s"(mask), [kc] "=&s"(kc), [ic] "=&s"(ic)
      : [key] "v"(key), [c] "s"(c), [jb] "s"(jbase)
      : "vcc", "scc");
}

// The same step for K >= 2 with the re-filter folded in and no wait states: the list is sorted, so the threshold AFTER
// the insertion is max(kc, key of entry K-2 before it) — two v_readlane up front, one s_max, and the compare that
// re-filters the remaining candidates issues under the insertion instead of behind a v_writelane -> v_readlane ->
// (2 wait states) -> v_cmp chain. 14 instructions; with s_ff1 and the loop branch 17 issue slots per insertion
// against 20 (18 + 2 s_nop) for knn_list_step + v_readlane + v_cmp + s_and.
// Hazards as above; additionally: SALU reads of VALU-written SGPRs (s_max on kc / t2, s_bcnt1 / s_and on VCC) are
// interlocked in hardware; v_readlane of lk follows the previous step's v_writelane by >= 4 instructions.
// cap: an upper bound (exclusive) on the keys worth looking at that does not come from the list — the hinted searches start
// from an EMPTY list and the bound of last call's neighbours (knn.hip); 0x7fffffff: none.
__device__ __forceinline__ void knn_list_step_refilter(int& lk, int& li, unsigned long long& mask, int& thr, int key, int c,
                                                       int jbase, int km2, int cap) {
  int kc, ic, t2;
  asm volatile(
      "v_readlane_b32 %[kc], %[key], %[c]\n\t"
      "v_readlane_b32 %[t2], %[lk], %[km2]\n\t"
      "s_bitset0_b64 %[mask], %[c]\n\t"
      "s_add_i32 %[ic], %[jb], %[c]\n\t"
      "s_max_i32 %[thr], %[kc], %[t2]\n\t"
      "s_min_i32 %[thr], %[thr], %[cap]\n\t"
      "v_cmp_ge_i32_e32 vcc, %[kc], %[lk]\n\t"
      "v_cndmask_b32_dpp %[lk], %[lk], %[lk], vcc wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      "v_cndmask_b32_dpp %[li], %[li], %[li], vcc wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      "s_bcnt1_i32_b64 m0, vcc\n\t"
      "v_cmp_gt_i32_e32 vcc, %[thr], %[key]\n\t"
      "s_and_b64 %[mask], vcc, %[mask]\n\t"
      "v_writelane_b32 %[lk], %[kc], m0\n\t"
      "v_writelane_b32 %[li], %[ic], m0"
      : [lk] "+v"(lk), [li] "+v"(li), [mask] "+s"(mask), [thr] "=&s"(thr), [kc] "=&s"(kc), [ic] "=&s"(ic), [t2] "=&s"(t2)
      : [key] "v"(key), [c] "s"(c), [jb] "s"(jbase), [km2] "s"(km2), [cap] "s"(cap)
      : "vcc", "scc");
}

// 64 candidates (one key per lane, index jbase + lane) against one list; thr = key of entry K-1, kept in an SGPR.
// Candidates are taken in ascending lane order; after every insertion the remaining ones are re-filtered against the
// tightened threshold, so no iteration is spent on a candidate that no longer qualifies.
template <bool KGE2>   // K >= 2 (decided per kernel instantiation: a run-time branch here doubles every unrolled call site)
__device__ __forceinline__ void knn_scan_insert(int& lk, int& li, int& thr, int key, int jbase, int K, int cap = 0x7fffffff) {
  unsigned long long mask = __builtin_amdgcn_ballot_w64(key < thr);
  if constexpr (KGE2) {
    const int km2 = K - 2;
    while (mask) knn_list_step_refilter(lk, li, mask, thr, key, __builtin_ctzll(mask), jbase, km2, cap);
  } else {
    while (mask) {
      knn_list_step(lk, li, mask, key, __builtin_ctzll(mask), jbase);
      thr = min(__builtin_amdgcn_readlane(lk, K - 1), cap);
      mask &= __builtin_amdgcn_ballot_w64(key < thr);
    }
  }
}

}  // namespace pc3d
