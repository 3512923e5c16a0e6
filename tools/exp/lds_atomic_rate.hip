// GPU box: what one wavefront pays per LDS update instruction on gfx950 (cycles by s_memtime, one wave per CU): the owner-wave
// scatters of csrc/det.hip issue one ds_add_f32 per (record, channel) and measured ~300 cycles per instruction in the kernel.
//   hipcc -O3 --offload-arch=gfx950 tools/exp/lds_atomic_rate.hip -o /tmp/lds_atomic_rate && /tmp/lds_atomic_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

constexpr int ROWS = 1024, ST = 9, IT = 256;

// mode 0: ds_add_f32 random rows (stride 9); 1: ds_add_f32 lane-private rows (no collisions, conflict-free banks);
// 2: ds_add_u32 random rows; 3: ds_write_b32 random rows; 4: read + add + write random rows (NOT a correct scatter: rate only);
// 5: ds_add_rtn_f32 random rows; 6: ds_add_f32 random rows, stride 8 (bank conflicts by construction);
// 7: ds_add_f32 with all 64 lanes on ONE address; 8: ds_pk_add_f16? (skipped); 9: ds_add_f64 random rows
template <int mode>
__global__ __launch_bounds__(64) void k(const int* rows, long long* out, float* sink) {
  extern __shared__ float lds[];
  for (int e = threadIdx.x; e < ROWS * 16; e += 64) lds[e] = 0.f;
  __syncthreads();
  int r[8];
  for (int q = 0; q < 8; ++q) r[q] = rows[(blockIdx.x * 8 + q) * 64 + threadIdx.x];
  float v = threadIdx.x * 0.001f + 1.f, acc = 0.f;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < IT; ++it) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      int row = (r[q] + it * 37) & (ROWS - 1);
      if (mode == 1) row = threadIdx.x + ((it & 15) << 6);
      if (mode == 7) row = 5;
      const int st = mode == 6 ? 8 : ST;
      float* p = lds + row * st + q;
      switch (mode) {
        case 2: atomicAdd(reinterpret_cast<unsigned*>(p), 3u); break;
        case 3: *reinterpret_cast<volatile float*>(p) = v; break;
        case 4: { float x = *reinterpret_cast<volatile float*>(p); *reinterpret_cast<volatile float*>(p) = x + v; } break;
        case 5: acc += atomicAdd(p, v); break;
        case 9: atomicAdd(reinterpret_cast<double*>(lds) + row * 5 + (q & 3), (double)v); break;
        default: atomicAdd(p, v); break;
      }
    }
  }
  __builtin_amdgcn_s_waitcnt(0);
  long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  if (acc == 12345.f) sink[0] = acc + lds[threadIdx.x];
  sink[1 + (threadIdx.x & 1)] = lds[threadIdx.x * 3];
}

int main() {
  const int WG = 256;
  int* h = (int*)malloc(WG * 8 * 64 * sizeof(int));
  srand(1);
  for (int i = 0; i < WG * 8 * 64; ++i) h[i] = rand() % ROWS;
  int* d;
  long long* out;
  float* sink;
  (void)hipMalloc(&d, WG * 8 * 64 * sizeof(int));
  (void)hipMalloc(&out, WG * sizeof(long long));
  (void)hipMalloc(&sink, 64);
  (void)hipMemcpy(d, h, WG * 8 * 64 * sizeof(int), hipMemcpyHostToDevice);
  const char* names[] = {"ds_add_f32 random rows, stride 9", "ds_add_f32 lane-private rows", "ds_add_u32 random rows", "ds_write_b32 random rows",
                         "read+add+write random rows", "ds_add_rtn_f32 random rows", "ds_add_f32 random rows, stride 8", "ds_add_f32 one address",
                         "", "ds_add_f64 random rows"};
  for (int wgs : {1, 256}) {
    for (int mode : {0, 1, 2, 3, 4, 5, 6, 7, 9}) {
      long long res[512];
      for (int rep = 0; rep < 2; ++rep) {
#define L(M) case M: (void)hipFuncSetAttribute((const void*)k<M>, hipFuncAttributeMaxDynamicSharedMemorySize, ROWS * 16 * 4); \
                hipLaunchKernelGGL(k<M>, dim3(wgs), dim3(64), ROWS * 16 * 4, 0, d, out, sink); break;
        switch (mode) { L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7) L(9) }
        (void)hipDeviceSynchronize();
      }
      (void)hipMemcpy(res, out, wgs * sizeof(long long), hipMemcpyDeviceToHost);
      double s = 0;
      for (int i = 0; i < wgs; ++i) s += res[i];
      // s_memtime counts at 100 MHz on gfx9; report both raw ticks and ns per instruction
      printf("wgs %3d  %-36s %8.1f ticks per instruction (x10 ns at 100 MHz)\n", wgs, names[mode], s / wgs / (IT * 8.0));
    }
  }
  return 0;
}
