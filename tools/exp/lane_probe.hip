// Prints what the cross-lane primitives used by nn_tree_min32 actually do on gfx950 (lane-id in, lane-id out).
// build: hipcc -O2 --offload-arch=gfx950 tools/exp/lane_probe.hip -o tools/exp/lane_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
#define DPP(old, src, ctrl, bank) __builtin_amdgcn_update_dpp((old), (src), (ctrl), 0xf, (bank), false)
__global__ void probe(int* out) {
  const int l = threadIdx.x;
  const int a = l, b = 100 + l;
  u32x2 s32 = __builtin_amdgcn_permlane32_swap((unsigned)a, (unsigned)b, false, false);
  u32x2 s16 = __builtin_amdgcn_permlane16_swap((unsigned)a, (unsigned)b, false, false);
  out[0 * 64 + l] = s32.x; out[1 * 64 + l] = s32.y;
  out[2 * 64 + l] = s16.x; out[3 * 64 + l] = s16.y;
  out[4 * 64 + l] = DPP(b, a, 0x128, 0x3);
  out[5 * 64 + l] = DPP(a, b, 0x128, 0xc);
  out[6 * 64 + l] = DPP(b, a, 0x104, 0x5);
  out[7 * 64 + l] = DPP(a, b, 0x114, 0xa);
  out[8 * 64 + l] = DPP(a, a, 0x4e, 0xf);
  out[9 * 64 + l] = DPP(a, a, 0xb1, 0xf);
}
int main() {
  int* d; hipMalloc(&d, 10 * 64 * 4);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
  int h[640]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* nm[10] = {"swap32.vdst", "swap32.src0", "swap16.vdst", "swap16.src0", "ror8 bank3 old=b src=a", "ror8 bankC old=a src=b",
                        "shl4 bank5 old=b src=a", "shr4 bankA old=a src=b", "quad[2,3,0,1]", "quad[1,0,3,2]"};
  for (int r = 0; r < 10; ++r) { printf("%-26s", nm[r]); for (int l = 0; l < 64; ++l) printf(" %d", h[r * 64 + l]); printf("\n"); }
  return 0;
}
