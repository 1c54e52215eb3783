// mfma_pattern.hip -- what one wave per SIMD sustains on v_mfma_f32_32x32x16_bf16 in the SHAPE of the ranking sweep: NACC
// independent accumulator chains taking turns, operands changing from MFMA to MFMA, VPER vector instructions (v_and_or /
// v_med3 / v_max on other registers) between two MFMAs.  hipcc --offload-arch=gfx950 -O3 mfma_pattern.hip -o mfma_pattern
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v16f32 __attribute__((ext_vector_type(16)));
typedef __bf16 v8bf16 __attribute__((ext_vector_type(8)));

template <int NACC, int VPER, int WAVES, int RCH = 1>
__global__ __launch_bounds__(WAVES * 64) void k_pat(float *out, int iters) {
  v16f32 acc[NACC];
  for (int i = 0; i < NACC; i++)
    for (int j = 0; j < 16; j++) acc[i][j] = 0;
  v8bf16 a[4], b[4];
  for (int q = 0; q < 4; q++)
    for (int j = 0; j < 8; j++) {
      a[q][j] = (__bf16)(1.0f + threadIdx.x * 1e-3f + j + q);
      b[q][j] = (__bf16)(1.0f - threadIdx.x * 1e-3f - q);
    }
  float m1[RCH], m2[RCH], v0 = threadIdx.x * 1.5f;
  for (int c = 0; c < RCH; c++) m1[c] = m2[c] = -1e30f;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
      for (int i = 0; i < NACC; i++) {
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[q], b[(q + i) & 3], acc[i], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < VPER; e++) {
          float v, m2n, m1n;
          asm volatile("v_and_or_b32 %0, %1, -16, 3" : "=v"(v) : "v"(v0));
          asm volatile("v_med3_f32 %0, %1, %2, %3" : "=v"(m2n) : "v"(m1[e % RCH]), "v"(m2[e % RCH]), "v"(v));
          asm volatile("v_max_f32 %0, %1, %2" : "=v"(m1n) : "v"(m1[e % RCH]), "v"(v));
          m2[e % RCH] = m2n;
          m1[e % RCH] = m1n;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
  }
  float s = 0;
  for (int c = 0; c < RCH; c++) s += m1[c] + m2[c];
  for (int i = 0; i < NACC; i++)
    for (int j = 0; j < 16; j++) s += acc[i][j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC, int VPER, int WAVES, int RCH = 1>
void run(const char *tag) {
  const int blocks = 256 * 4;  // WAVES waves per workgroup, 4 workgroups per CU's worth of grid; one workgroup per SIMD set below
  float *out;
  hipMalloc(&out, sizeof(float) * blocks * WAVES * 64);
  const int iters = 4000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k_pat<NACC, VPER, WAVES, RCH>), dim3(blocks), dim3(WAVES * 64), 0, 0, out, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k_pat<NACC, VPER, WAVES, RCH>), dim3(blocks), dim3(WAVES * 64), 0, 0, out, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)iters * 4 * NACC * (double)blocks * WAVES * 32768.0;
  printf("%-52s: %7.1f TFLOP/s chip (%.3f ms)\n", tag, flops / (ms * 1e-3) / 1e12, ms);
  hipFree(out);
}

int main() {
  // 1024 workgroups of 1 wave = one wave per SIMD; of 2 waves = two per SIMD (256 CUs x 4 SIMDs)
  run<1, 0, 1>("1 wave/SIMD, 1 chain, no VALU");
  run<2, 0, 1>("1 wave/SIMD, 2 chains, no VALU");
  run<4, 0, 1>("1 wave/SIMD, 4 chains, no VALU");
  run<2, 1, 1>("1 wave/SIMD, 2 chains, 3 VALU per MFMA");
  run<2, 2, 1>("1 wave/SIMD, 2 chains, 6 VALU per MFMA");
  run<1, 0, 2>("2 waves/SIMD, 1 chain, no VALU");
  run<1, 1, 2>("2 waves/SIMD, 1 chain, 3 VALU per MFMA");
  run<1, 2, 2>("2 waves/SIMD, 1 chain, 6 VALU per MFMA");
  run<2, 1, 2>("2 waves/SIMD, 2 chains, 3 VALU per MFMA");
  run<2, 2, 1, 2>("1 wave/SIMD, 2 chains, 6 VALU per MFMA, 2 ranking chains");
  run<2, 4, 1, 4>("1 wave/SIMD, 2 chains, 12 VALU per MFMA, 4 ranking chains");
  run<2, 2, 1, 1>("1 wave/SIMD, 2 chains, 6 VALU per MFMA, 1 ranking chain");
  run<1, 2, 2, 2>("2 waves/SIMD, 1 chain, 6 VALU per MFMA, 2 ranking chains");
  run<1, 2, 2, 1>("2 waves/SIMD, 1 chain, 6 VALU per MFMA, 1 ranking chain");
  run<1, 2, 3, 1>("3 waves/SIMD, 1 chain, 6 VALU per MFMA, 1 ranking chain");
  run<1, 2, 3, 2>("3 waves/SIMD, 1 chain, 6 VALU per MFMA, 2 ranking chains");
  run<1, 3, 3, 1>("3 waves/SIMD, 1 chain, 9 VALU per MFMA, 1 ranking chain");
  run<1, 3, 3, 3>("3 waves/SIMD, 1 chain, 9 VALU per MFMA, 3 ranking chains");
  return 0;
}
