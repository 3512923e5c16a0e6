// GPU box: does a second wavefront on a SIMD run beside the first one? One workgroup of 256 / 512 / 1024 threads (1 / 2 / 4 waves per
// SIMD), every wave runs the same loop; clocks per loop trip by s_memtime. (a) a dependent chain of v_add_f32, (b) independent
// v_add_f32, (c) a dependent chain of DPP maxima with their s_nop, (d) packed v_pk_add_f32 (dependent).
//   hipcc -O3 --offload-arch=gfx950 tools/exp/simd_issue.hip -o /tmp/simd_issue && /tmp/simd_issue
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP8(X) X X X X X X X X
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ void k(long long* out, float* sink) {
  float a = threadIdx.x, b = 1.5f, c = 2.5f, d = 3.5f, e = 4.5f;
  f2 p = {a, b}, q = {c, d};
  long long t0, t1;
  t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < 256; ++i) { REP8(asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(b));) }
  t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) out[(threadIdx.x >> 6) * 4 + 0] = t1 - t0;
  t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < 256; ++i) {
    asm volatile("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n"
                 "v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4"
                 : "+v"(a), "+v"(c), "+v"(d), "+v"(e) : "v"(b));
  }
  t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) out[(threadIdx.x >> 6) * 4 + 1] = t1 - t0;
  t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < 256; ++i) { REP8(asm volatile("s_nop 1\n v_max_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a));) }
  t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) out[(threadIdx.x >> 6) * 4 + 2] = t1 - t0;
  t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < 256; ++i) { REP8(asm volatile("v_pk_add_f32 %0, %0, %1\n s_nop 0" : "+v"(p) : "v"(q));) }
  t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) out[(threadIdx.x >> 6) * 4 + 3] = t1 - t0;
  sink[threadIdx.x] = a + c + d + e + p.x + p.y;
}
int main() {
  long long* out; float* sink;
  hipMalloc(&out, 16 * 4 * sizeof(long long)); hipMalloc(&sink, 1024 * 4);
  for (int thr : {64, 256, 512, 1024}) {
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k, dim3(1), dim3(thr), 0, 0, out, sink);
    hipDeviceSynchronize();
    long long h[64]; hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    printf("%4d threads (%d wave(s) per SIMD): clocks per instruction, wave 0: dependent v_add %.2f | 4 independent chains %.2f | nop1+dpp %.2f | pk_add+nop0 %.2f\n",
           thr, thr <= 256 ? 1 : thr / 256, h[0] / 2048.0, h[1] / 2048.0, h[2] / 2048.0, h[3] / 2048.0);
  }
  return 0;
}
