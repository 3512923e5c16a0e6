// GPU box: do the two fp32 MFMA shapes sustain the same FLOP/s on random data? (MI355X lowers its clock under matrix load; for
// bf16 the 16x16 shape is known to hold a higher clock.) Bare loops, operands in registers, 2 waves per SIMD, every CU busy.
//   hipcc -O3 --offload-arch=gfx950 tools/exp/mfma_f32_shapes.hip -o /tmp/mfma_shapes && /tmp/mfma_shapes
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
using f16v = __attribute__((ext_vector_type(16))) float;
using f4v = __attribute__((ext_vector_type(4))) float;

template <int SHAPE>
__global__ __launch_bounds__(512, 2) void k(const float* in, float* out, int iters) {
  const int tid = threadIdx.x + blockIdx.x * 512;
  float a[8], b[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = in[(tid * 16 + i) & 0xffff], b[i] = in[(tid * 16 + 8 + i) & 0xffff];
  if (SHAPE == 32) {
    f16v acc[2] = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[i], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[i], a[i], acc[1], 0, 0, 0);
      }
    }
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) s += acc[0][e] + acc[1][e];
    out[tid] = s;
  } else {
    f4v acc[8] = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[(2 * i + (j >> 1)) & 7] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[(i + j) & 7], acc[(2 * i + (j >> 1)) & 7], 0, 0, 0);
      }
    }
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) s += acc[e][0] + acc[e][1] + acc[e][2] + acc[e][3];
    out[tid] = s;
  }
}

int main() {
  float *in, *out;
  hipMalloc(&in, 65536 * 4);
  hipMalloc(&out, 512 * 512 * 4 * 4);
  float* h = (float*)malloc(65536 * 4);
  for (int zero = 0; zero < 2; ++zero) {
    for (int i = 0; i < 65536; ++i) h[i] = zero ? 0.f : (float)rand() / RAND_MAX - 0.5f;
    hipMemcpy(in, h, 65536 * 4, hipMemcpyHostToDevice);
    for (int shape : {32, 16, 32, 16}) {
      const int iters = 20000, blocks = 512;
      hipEvent_t e0, e1;
      hipEventCreate(&e0), hipEventCreate(&e1);
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (shape == 32) hipLaunchKernelGGL(k<32>, dim3(blocks), dim3(512), 0, 0, in, out, iters);
        else hipLaunchKernelGGL(k<16>, dim3(blocks), dim3(512), 0, 0, in, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
      }
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      // per iteration and wave: SHAPE 32: 16 MFMAs x 4096 flop; SHAPE 16: 32 MFMAs x 2048 flop
      const double flop = (double)blocks * 8 * iters * 16 * 4096.0;
      printf("%s data, %dx%d: %.2f ms, %.1f TFLOP/s\n", zero ? "zero" : "random", shape, shape, ms, flop / ms / 1e9);
    }
  }
  return 0;
}
