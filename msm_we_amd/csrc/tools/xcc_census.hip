// xcc_census.hip -- which XCD does workgroup i of a 1-D grid land on?  (round-robin is assumed by XCD-aware mappings)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k_census(unsigned *out) {
  unsigned xcc, hwid;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = xcc;
    out[2 * blockIdx.x + 1] = hwid;
  }
}
int main() {
  for (int nb : {256, 64, 1000}) {
    unsigned *d, h[2048];
    hipMalloc(&d, sizeof(h));
    hipMemset(d, 0, sizeof(h));
    hipLaunchKernelGGL(k_census, dim3(nb), dim3(256), 0, 0, d);
    hipMemcpy(h, d, sizeof(unsigned) * 2 * nb, hipMemcpyDeviceToHost);
    int ok = 0;
    printf("grid %d: xcc of blocks 0..23:", nb);
    for (int i = 0; i < nb; i++) {
      unsigned x = h[2 * i] & 0xf;
      if (i < 24) printf(" %u", x);
      ok += (x == (unsigned)(i % 8));
    }
    printf("  | blocks with xcc == i %% 8: %d / %d\n", ok, nb);
    hipFree(d);
  }
  return 0;
}
