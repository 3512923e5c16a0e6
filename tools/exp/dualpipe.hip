// Feasibility probe: do an MFMA-bound wave and a VALU-bound (scalar-operand FMA) wave on the same SIMD both run at
// their own rate?  hipcc --offload-arch=gfx950 -O3 -o dualpipe dualpipe.hip && ./dualpipe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using f32x16 = __attribute__((ext_vector_type(16))) float;

// mode bit0: waves 0-3 run MFMAs; bit1: waves 4-7 run scalar-operand FMAs
__global__ __launch_bounds__(512) void probe(const float* __restrict__ wT, float* out, int mode, int nm, int nk) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (wave < 4) {
    if (!(mode & 1)) return;
    f32x16 acc[2];
    for (int t = 0; t < 2; ++t) for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    float a = lane * 0.001f, b = lane * 0.002f;
    for (int i = 0; i < nm; ++i) {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc[1], 0, 0, 0);
    }
    float s = 0.f;
    for (int t = 0; t < 2; ++t) for (int e = 0; e < 16; ++e) s += acc[t][e];
    out[blockIdx.x * 512 + threadIdx.x] = s;
  } else {
    if (!(mode & 2)) return;
    // lane = point: h[k] per lane in registers (32 of the 128 k here), 32 channel accumulators, weights through SGPRs
    float h[32], acc[32];
    for (int k = 0; k < 32; ++k) h[k] = out[(k * 64 + lane) & 1023];   // opaque values
    for (int c = 0; c < 32; ++c) acc[c] = 0.f;
    const float* w = wT + (size_t)(wave - 4) * 32 * 1024;   // uniform address -> s_load
    for (int it = 0; it < nk; ++it) {
#pragma unroll
      for (int k = 0; k < 32; ++k) {
        const float* wr = w + (size_t)((it * 32 + k) & 1023) * 32;
#pragma unroll
        for (int c = 0; c < 32; ++c) acc[c] = __builtin_fmaf(wr[c], h[k], acc[c]);
      }
    }
    float s = 0.f;
    for (int c = 0; c < 32; ++c) s += acc[c];
    out[blockIdx.x * 512 + threadIdx.x] = s;
  }
}

int main() {
  float *w, *out;
  hipMalloc(&w, 4 * 32 * 1024 * 4 * 8);
  hipMalloc(&out, 256 * 512 * 4);
  hipMemset(w, 0, 4 * 32 * 1024 * 4 * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int nm = 1024, nk = 64;   // 2048 MFMAs per wave; 64*1024 = 65536 v_fma per wave
  for (int mode : {1, 2, 3, 1, 2, 3}) {
    hipLaunchKernelGGL(probe, dim3(256), dim3(512), 0, 0, w, out, mode, nm, nk);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(probe, dim3(256), dim3(512), 0, 0, w, out, mode, nm, nk);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double us = ms * 100;
    double mf = 256.0 * 4 * (2.0 * nm) * 32 * 32 * 2 * 2;          // flops of the MFMA waves
    double vf = 256.0 * 4 * 64 * (double)nk * 1024 * 2;             // flops of the VALU waves
    printf("mode %d: %.1f us  mfma %.1f TF  valu %.1f TF\n", mode, us, (mode & 1) ? mf / us / 1e6 : 0.0, (mode & 2) ? vf / us / 1e6 : 0.0);
  }
  return 0;
}
