// mfma_corun.hip -- do waves that stream MFMAs slow down OTHER waves of the same SIMD, and does it matter whether the
// MFMAs form one dependent chain or two alternating ones?  Workgroups of 8 waves: waves 0-3 run f32 32x32x2 MFMAs
// (MODE 1: one accumulator, 2: two alternating, 0: idle), waves 4-7 run a latency-style loop (LDS read -> a few VALU ops
// -> LDS write) and report their cycles per iteration.
// Build: hipcc --offload-arch=gfx950 -O3 mfma_corun.hip -o mfma_corun ; run: ./mfma_corun
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v16f32 __attribute__((ext_vector_type(16)));

template <int MODE, int PRIO>
__global__ __launch_bounds__(512) void k(float *out, unsigned long long *cyc, int iters) {
  __shared__ float lds[4096];
  const int tid = threadIdx.x, wv = tid >> 6;
  lds[tid] = tid;
  lds[tid + 512] = tid;
  __syncthreads();
  if (wv < 4) {
    if (MODE == 0) return;
    unsigned long long tm0 = __builtin_amdgcn_s_memtime();
    v16f32 a0, a1;
    for (int r = 0; r < 16; r++) a0[r] = a1[r] = 0.f;
    float x = tid * 1e-3f, y = 1.0f - tid * 1e-4f;
    for (int it = 0; it < iters * 8; it++) {
      if (MODE == 1) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a0, 0, 0, 0);
      } else if (MODE == 2) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
      } else if (MODE == 3) {  // yield the issue port while the MFMA runs
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
        __builtin_amdgcn_s_sleep(1);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
        __builtin_amdgcn_s_sleep(1);
      } else if (MODE == 4) {  // s_nop padding instead (s_nop n = n + 1 quad-cycles)
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
        asm volatile("s_nop 15");
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
        asm volatile("s_nop 15");
      } else if (MODE == 5) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
        asm volatile("s_nop 13");
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
        asm volatile("s_nop 13");
      } else if (MODE == 6) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
        asm volatile("s_nop 11");
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
        asm volatile("s_nop 11");
      } else {  // 7
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
        asm volatile("s_nop 7");
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
        asm volatile("s_nop 7");
      }
    }
    float s = 0;
    for (int r = 0; r < 16; r++) s += a0[r] + a1[r];
    if ((tid & 63) == 0 && blockIdx.x == 0) cyc[4 + wv] = __builtin_amdgcn_s_memtime() - tm0;
    out[blockIdx.x * 512 + tid] = s;
  } else {
    if (PRIO) __builtin_amdgcn_s_setprio(3);
    float v = tid;
    int p = tid & 1023;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
      float r = lds[p];
      v = v * 1.0001f + r;
      p = ((int)v & 1023);
      lds[(p + 2048) & 4095] = v;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 512 + tid] = v;
    if ((tid & 63) == 0 && blockIdx.x == 0) cyc[wv - 4] = t1 - t0;
  }
}

template <int MODE, int PRIO>
void run(const char *tag) {
  float *out;
  unsigned long long *cyc, h[8];
  hipMalloc(&out, 256 * 512 * 4);
  hipMalloc(&cyc, 64);
  const int iters = 2000;
  hipLaunchKernelGGL((k<MODE, PRIO>), dim3(256), dim3(512), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
  printf("%-44s latency-loop waves: %.0f cycles per iteration; MFMA waves: %.1f cycles per MFMA\n", tag, (double)h[0] / iters,
         MODE ? (double)h[4] / (iters * 16.0) : 0.0);
  hipFree(out);
  hipFree(cyc);
}

int main() {
  run<0, 0>("alone (MFMA waves idle)");
  run<1, 0>("beside ONE dependent MFMA chain per wave");
  run<2, 0>("beside TWO alternating MFMA chains per wave");
  run<1, 1>("same, latency waves at s_setprio 3");
  run<2, 1>("two chains, latency waves at s_setprio 3");
  run<3, 0>("two chains, s_sleep 1 after every MFMA");
  run<4, 0>("two chains, s_nop 15 after every MFMA");
  run<5, 0>("two chains, s_nop 13 after every MFMA");
  run<6, 0>("two chains, s_nop 11 after every MFMA");
  run<7, 0>("two chains, s_nop 7 after every MFMA");
  return 0;
}
