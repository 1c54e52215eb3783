// mfma_f64_rate.hip -- micro-benchmark: issue rate of v_mfma_f64_16x16x4_f64 on gfx950 (the roof of k_assign).
// Build: hipcc --offload-arch=gfx950 -O3 mfma_f64_rate.hip -o mfma_f64_rate ; run: ./mfma_f64_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double v4f64 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k_rate(double *out, int iters, unsigned long long *cycles) {
  v4f64 acc[NACC];
  for (int i = 0; i < NACC; i++) acc[i] = (v4f64){0, 0, 0, 0};
  double a = threadIdx.x * 1e-3 + 1.0, b = 1.0 - threadIdx.x * 1e-4;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

template <int NACC>
void run(int blocks, int threads, const char *tag) {
  double *out;
  unsigned long long *cyc, hc;
  hipMalloc(&out, sizeof(double) * blocks * threads);
  hipMalloc(&cyc, 8);
  int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k_rate<NACC>, dim3(blocks), dim3(threads), 0, 0, out, 100, cyc);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k_rate<NACC>, dim3(blocks), dim3(threads), 0, 0, out, iters, cyc);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
  double nm = (double)iters * NACC;
  double waves = (double)blocks * threads / 64;
  double flops = nm * waves * 2048.0;
  printf("%-28s acc=%d blocks=%d thr=%d : %.1f cycles/MFMA/wave (memtime), %.2f TFLOP/s chip, %.3f ms\n", tag, NACC, blocks,
         threads, (double)hc / nm, flops / (ms * 1e-3) / 1e12, ms);
  hipFree(out);
  hipFree(cyc);
}

int main() {
  run<1>(256, 256, "1 wave/SIMD dependent");
  run<2>(256, 256, "1 wave/SIMD 2 chains");
  run<4>(256, 256, "1 wave/SIMD 4 chains");
  run<8>(256, 256, "1 wave/SIMD 8 chains");
  run<1>(512, 256, "2 waves/SIMD dependent");
  run<4>(512, 256, "2 waves/SIMD 4 chains");
  run<4>(1024, 256, "4 waves/SIMD 4 chains");
  run<4>(1, 64, "single wave 4 chains");
  return 0;
}
