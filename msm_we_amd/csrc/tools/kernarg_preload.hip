// kernarg_preload.hip -- does preloading kernel arguments into SGPRs (-mllvm -amdgpu-kernarg-preload-count=N) shorten a chain of
// short dependent kernels on this runtime?  Chain of launches; each kernel's first action depends on its arguments.
//   hipcc --offload-arch=gfx950 -O3 -o kp0 kernarg_preload.hip                                   (s_load of the arguments)
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-kernarg-preload-count=16 -o kp1 kernarg_preload.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

struct Pack {
  unsigned long long a[16];
};

// arguments as scalars (preloadable): the wave stores wall_clock at entry and after its first argument-dependent load
__global__ void k_scalar(unsigned long long *out, const unsigned *src, unsigned off0, unsigned off1, unsigned n0, unsigned n1,
                         unsigned long long p2, unsigned long long p3, unsigned long long p4) {
  const unsigned long long t0 = wall_clock64();
  const unsigned o = blockIdx.y ? off1 : off0;
  const unsigned v = src[o + threadIdx.x];
  asm volatile("s_waitcnt vmcnt(0)" ::"v"(v) : "memory");
  const unsigned long long t1 = wall_clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    out[0] += t1 - t0 + (v & 0) + ((n0 + n1 + p2 + p3 + p4) & 0);
    out[1] += 1;
  }
}

// arguments as a by-value struct indexed at run time (what the training kernels do today)
__global__ void k_struct(unsigned long long *out, const unsigned *src, const Pack pk) {
  const unsigned long long t0 = wall_clock64();
  const unsigned o = (unsigned)pk.a[blockIdx.y * 8];
  const unsigned v = src[o + threadIdx.x];
  asm volatile("s_waitcnt vmcnt(0)" ::"v"(v) : "memory");
  const unsigned long long t1 = wall_clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    out[0] += t1 - t0 + (v & 0);
    out[1] += 1;
  }
}

int main() {
  unsigned long long *out;
  unsigned *src;
  hipMalloc(&out, 64);
  hipMalloc(&src, 1 << 20);
  hipMemset(src, 0, 1 << 20);
  const int N = 4000;
  for (int mode = 0; mode < 2; mode++)
    for (int rep = 0; rep < 3; rep++) {
      hipMemset(out, 0, 64);
      hipDeviceSynchronize();
      auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < N; i++) {
        if (mode == 0) {
          Pack pk{};
          pk.a[0] = (i * 64) & 0xffff;
          pk.a[8] = (i * 64 + 4096) & 0xffff;
          hipLaunchKernelGGL(k_struct, dim3(160, 2), dim3(256), 0, 0, out, src, pk);
        } else {
          hipLaunchKernelGGL(k_scalar, dim3(160, 2), dim3(256), 0, 0, out, src, (unsigned)((i * 64) & 0xffff),
                             (unsigned)((i * 64 + 4096) & 0xffff), 1u, 2u, 3ull, 4ull, 5ull);
        }
      }
      hipDeviceSynchronize();
      double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      unsigned long long h[2];
      hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
      printf("%s: %.2f us per launch (chain of %d); entry -> first argument-dependent load done: %.0f ns (wave 0 of workgroup 0)\n",
             mode ? "scalar args" : "struct by value", us / N, N, 10.0 * (double)h[0] / (double)h[1]);
    }
  return 0;
}
