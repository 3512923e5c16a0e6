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

// v_writelane_b32 (hipcc 7.2 has no builtin; the LLVM intrinsic is reachable by name, and unlike inline asm it is seen
// by the hazard recogniser)
extern "C" __device__ int knn_writelane(int value, int lane, int old) __asm("llvm.amdgcn.writelane.i32");

// insert (dc, ic) — wave-uniform — AFTER every entry <= dc (candidates arrive in ascending index order, so equal
// distances keep the lower index first); the entry in lane 63 falls off. The list is sorted, so the lanes with
// ld <= dc are exactly the lanes below the insertion point: the compare's 64-bit mask IS the "keep" predicate of the
// shift (one v_cmp, two DPP moves, two v_cndmask on that mask, two v_writelane for the new entry — 7 VALU).
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

}  // namespace pc3d
