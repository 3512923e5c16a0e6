// GPU box: what single-wavefront instruction sequences cost on gfx950 (cycles by s_memtime, one wave per SIMD, nothing else on
// the CU) — the price list the pruned FPS chain (csrc/fps_pruned.hip) is written against.
//   hipcc -O3 --offload-arch=gfx950 tools/exp/wave_costs.hip -o /tmp/wave_costs && /tmp/wave_costs
#include <hip/hip_runtime.h>
#include <stdio.h>

#define REP16(X) X X X X X X X X X X X X X X X X

#define TIMED(idx, BODY)                                     \
  {                                                          \
    long long t0 = __builtin_amdgcn_s_memtime();             \
    for (int it = 0; it < 64; ++it) { REP16(BODY) }          \
    long long t1 = __builtin_amdgcn_s_memtime();             \
    if (threadIdx.x == 0) out[idx] = (t1 - t0);              \
  }

__global__ void k(long long* out, float* sink, int zero) {
  __shared__ float lds[1024];
  float v = threadIdx.x * 1.5f, w = threadIdx.x * 0.25f + 3.f;
  int iv = threadIdx.x;
  lds[threadIdx.x] = v;
  lds[threadIdx.x + 64] = w;
  __syncthreads();
  float r = 0.f;
  // 0: empty loop overhead
  TIMED(0, asm volatile("" ::: "memory");)
  // 1: 16 independent v_add
  TIMED(1, asm volatile("v_add_f32 %0, %0, %1" : "+v"(v) : "v"(w));)
  // 2: s_nop 0
  TIMED(2, asm volatile("s_nop 0");)
  // 3: s_nop 1
  TIMED(3, asm volatile("s_nop 1");)
  // 4: s_nop 3
  TIMED(4, asm volatile("s_nop 3");)
  // 5: dpp max + s_nop 1 (dependent)
  TIMED(5, asm volatile("s_nop 1\n v_max_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(v));)
  // 6: dpp max row_bcast:15 + s_nop 1
  TIMED(6, asm volatile("s_nop 1\n v_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xf bank_mask:0xf" : "+v"(v));)
  // 7: full 6-level chain + readlane + s_nop 3
  TIMED(7, asm volatile("s_nop 1\n v_max_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n"
                        "v_max_f32_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n s_nop 1\n"
                        "v_max_f32_dpp %1, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xf\n s_nop 1\n"
                        "v_max_f32_dpp %1, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xf\n s_nop 1\n"
                        "v_max_f32_dpp %1, %1, %1 row_bcast:15 row_mask:0xf bank_mask:0xf\n s_nop 1\n"
                        "v_max_f32_dpp %1, %1, %1 row_bcast:31 row_mask:0xf bank_mask:0xf\n s_nop 1\n"
                        "v_readlane_b32 %0, %1, 63\n s_nop 3" : "=s"(r), "+v"(v));)
  // 8: v_readlane + s_nop 3 + v_cmp using it + s_ff1
  {
    int sres = 0;
    TIMED(8, asm volatile("v_readlane_b32 %0, %1, 63\n s_nop 3\n v_cmp_eq_f32 vcc, %0, %1\n s_ff1_i32_b64 %0, vcc" : "=s"(sres) : "v"(v) : "vcc");)
    iv += sres;
  }
  // 9: v_writelane pair with m0
  TIMED(9, asm volatile("s_mov_b32 m0, 5\n v_writelane_b32 %0, 7, m0\n v_writelane_b32 %1, 9, m0" : "+v"(v), "+v"(w) : : "m0");)
  // 10: dependent ds_read chain (address from the value read)
  {
    int addr = (threadIdx.x & 63) * 4;
    TIMED(10, asm volatile("ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)\n v_and_b32 %0, 0xfc, %0" : "+v"(addr) : : "memory");)
    iv += addr;
  }
  // 11: taken scalar branch
  TIMED(11, asm volatile("s_cmp_eq_u32 %0, 0\n s_cbranch_scc1 1f\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n1:" : : "s"(zero) : "scc");)
  // 12: not-taken scalar branch
  TIMED(12, asm volatile("s_cmp_lg_u32 %0, 0\n s_cbranch_scc1 1f\n1:" : : "s"(zero) : "scc");)
  // 13: v_cmp -> vcc -> s_cbranch_vccz (not taken)
  TIMED(13, asm volatile("v_cmp_lt_f32 vcc, %0, %1\n s_cbranch_vccz 1f\n1:" : : "v"(v), "v"(w) : "vcc");)
  // 14: ds_write + ds_read of another wave's slot pattern: write b64, read 4 x b64, wait
  {
    float2 a2 = make_float2(v, w);
    unsigned addr = (unsigned)(threadIdx.x >> 6) * 8;
    TIMED(14, asm volatile("ds_write_b64 %1, %0\n ds_read_b64 %0, %2\n s_waitcnt lgkmcnt(0)" : "+v"(a2) : "v"(addr), "v"(addr) : "memory");)
    v += a2.x;
  }
  // 15: 16 dependent v_add (same register chain)
  TIMED(15, asm volatile("v_add_f32 %0, %0, %0" : "+v"(v));)
  // 16: v_min + sub/mul/add distance block (9 VALU, dependent mix)
  TIMED(16, asm volatile("v_sub_f32 %0, %0, %1\n v_mul_f32 %0, %0, %0\n v_add_f32 %0, %0, %1\n v_min_f32 %0, %0, %1" : "+v"(v) : "v"(w));)
  sink[threadIdx.x] = v + w + r + iv;
}

int main() {
  long long* out;
  float* sink;
  hipMalloc(&out, 64 * sizeof(long long));
  hipMalloc(&sink, 1024 * sizeof(float));
  for (int threads : {64, 256}) {
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k, dim3(1), dim3(threads), 0, 0, out, sink, 0);
    hipDeviceSynchronize();
    long long h[64];
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[] = {"empty", "v_add indep", "s_nop 0", "s_nop 1", "s_nop 3", "nop1+dpp shr1", "nop1+dpp bcast15", "chain6+readlane",
                           "readlane+nop3+cmp+ff1", "m0+2 writelane", "dep ds_read", "taken branch(+cmp)", "untaken branch(+cmp)",
                           "v_cmp+vccz untaken", "ds_write+read+wait", "v_add dep", "4-op dist block"};
    printf("threads per block %d (cycles per snippet, loop overhead subtracted)\n", threads);
    for (int i = 0; i < 17; ++i) printf("  %-26s %8.1f\n", names[i], (double)(h[i] - h[0]) / 1024.0);
  }
  return 0;
}
