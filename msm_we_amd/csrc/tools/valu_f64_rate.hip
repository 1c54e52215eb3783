// valu_f64_rate.hip -- micro-benchmark: issue rate of v_fma_f64 on gfx950 (the roof of the k-means++ distance phase).
// Build: hipcc --offload-arch=gfx950 -O3 valu_f64_rate.hip -o valu_f64_rate ; run: ./valu_f64_rate
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int NCH>
__global__ __launch_bounds__(256) void k_rate(double *out, int iters, unsigned long long *cycles) {
  double acc[NCH];
  for (int i = 0; i < NCH; i++) acc[i] = threadIdx.x * 1e-9 + i;
  double a = 1.0 + threadIdx.x * 1e-12, b = 1e-9;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  unsigned long long c0 = wall_clock64();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < NCH; i++) acc[i] = __builtin_fma(acc[i], a, b);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  unsigned long long c1 = wall_clock64();
  double s = 0;
  for (int i = 0; i < NCH; i++) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    cycles[0] = t1 - t0;
    cycles[1] = c1 - c0;
  }
}

template <int NCH>
void run(int blocks, int threads, const char *tag) {
  double *out;
  unsigned long long *cyc, hc[2];
  hipMalloc(&out, sizeof(double) * blocks * threads);
  hipMalloc(&cyc, 16);
  int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k_rate<NCH>, dim3(blocks), dim3(threads), 0, 0, out, 100, cyc);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k_rate<NCH>, dim3(blocks), dim3(threads), 0, 0, out, iters, cyc);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  hipMemcpy(hc, cyc, 16, hipMemcpyDeviceToHost);
  double nf = (double)iters * NCH;
  double flops = nf * blocks * threads * 2.0;
  printf("%-26s chains=%2d blocks=%4d thr=%4d : %.2f memtime ticks / FMA / wave, %.2f ns / FMA / wave (100 MHz clock), %.2f TFLOP/s chip, %.3f ms\n",
         tag, NCH, blocks, threads, (double)hc[0] / nf, (double)hc[1] * 10.0 / nf, flops / (ms * 1e-3) / 1e12, ms);
  hipFree(out);
  hipFree(cyc);
}

int main() {
  run<8>(256, 256, "1 wave/SIMD, whole chip");
  run<8>(512, 256, "2 waves/SIMD, whole chip");
  run<8>(1024, 256, "4 waves/SIMD, whole chip");
  run<48>(512, 256, "2 waves/SIMD 48 chains");
  run<8>(1, 512, "one CU, 2 waves/SIMD");
  run<48>(1, 512, "one CU, 2 w/SIMD, 48 ch");
  run<8>(1, 1024, "one CU, 4 waves/SIMD");
  return 0;
}
