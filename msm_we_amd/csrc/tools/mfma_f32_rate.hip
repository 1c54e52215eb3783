// mfma_f32_rate.hip -- micro-benchmark: sustained rate of the f32 and bf16 MFMAs on gfx950, next to the f64 one
// (mfma_f64_rate.hip): what a lower-precision distance FILTER in front of the exact f64 kernel could run at.
// Build: hipcc --offload-arch=gfx950 -O3 mfma_f32_rate.hip -o mfma_f32_rate ; run: ./mfma_f32_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v16f32 __attribute__((ext_vector_type(16)));
typedef float v4f32 __attribute__((ext_vector_type(4)));
typedef __bf16 v8bf16 __attribute__((ext_vector_type(8)));

template <int KIND, int NACC>
__global__ __launch_bounds__(256) void k_rate(float *out, int iters) {
  float s = 0;
  if constexpr (KIND == 0) {  // v_mfma_f32_32x32x2_f32: 4096 flop
    v16f32 acc[NACC];
    for (int i = 0; i < NACC; i++)
      for (int j = 0; j < 16; j++) acc[i][j] = 0;
    float a = threadIdx.x * 1e-3f + 1.0f, b = 1.0f - threadIdx.x * 1e-4f;
    for (int it = 0; it < iters; it++)
#pragma unroll
      for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    for (int i = 0; i < NACC; i++)
      for (int j = 0; j < 16; j++) s += acc[i][j];
  } else if constexpr (KIND == 1) {  // v_mfma_f32_16x16x4_f32: 2048 flop
    v4f32 acc[NACC];
    for (int i = 0; i < NACC; i++) acc[i] = (v4f32){0, 0, 0, 0};
    float a = threadIdx.x * 1e-3f + 1.0f, b = 1.0f - threadIdx.x * 1e-4f;
    for (int it = 0; it < iters; it++)
#pragma unroll
      for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  } else {  // v_mfma_f32_32x32x16_bf16: 32768 flop
    v16f32 acc[NACC];
    for (int i = 0; i < NACC; i++)
      for (int j = 0; j < 16; j++) acc[i][j] = 0;
    v8bf16 a, b;
    for (int j = 0; j < 8; j++) {
      a[j] = (__bf16)(1.0f + threadIdx.x * 1e-3f + j);
      b[j] = (__bf16)(1.0f - threadIdx.x * 1e-3f);
    }
    for (int it = 0; it < iters; it++)
#pragma unroll
      for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    for (int i = 0; i < NACC; i++)
      for (int j = 0; j < 16; j++) s += acc[i][j];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND, int NACC>
void run(int blocks, const char *tag, double flop_per) {
  float *out;
  hipMalloc(&out, sizeof(float) * blocks * 256);
  int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k_rate<KIND, NACC>), dim3(blocks), dim3(256), 0, 0, out, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k_rate<KIND, NACC>), dim3(blocks), dim3(256), 0, 0, out, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  double flops = (double)iters * NACC * (blocks * 4.0) * flop_per;
  printf("%-34s acc=%d blocks=%d : %.2f TFLOP/s chip, %.3f ms\n", tag, NACC, blocks, flops / (ms * 1e-3) / 1e12, ms);
  hipFree(out);
}

int main() {
  run<0, 4>(256, "f32 32x32x2, 1 wave/SIMD", 4096);
  run<0, 4>(512, "f32 32x32x2, 2 waves/SIMD", 4096);
  run<1, 4>(256, "f32 16x16x4, 1 wave/SIMD", 2048);
  run<1, 4>(512, "f32 16x16x4, 2 waves/SIMD", 2048);
  run<2, 4>(256, "bf16 32x32x16, 1 wave/SIMD", 32768);
  run<2, 4>(512, "bf16 32x32x16, 2 waves/SIMD", 32768);
  return 0;
}
