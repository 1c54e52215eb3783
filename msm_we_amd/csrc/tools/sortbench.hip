// rocPRIM's onesweep at the shapes of the deterministic flux build: 10 M (key, f64) pairs sorted on the cell bits only (stable),
// default configuration against wider digits with the `match` rank algorithm (fewer passes: 20 bits = 3 x 8 or 2 x 10; 25 bits =
// 4 x 8 -- the fourth pass sorts ONE bit -- or 3 x 9).  Checks that every configuration returns the default's permutation.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o sortbench tools/sortbench.hip ; usage: ./sortbench
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <cstdio>
#include <random>
#include <vector>
template <int RB, int BS, int IPT>
using Cfg = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                       rocprim::radix_sort_onesweep_config<rocprim::kernel_config<256, 12>, rocprim::kernel_config<BS, IPT>, RB,
                                                                           rocprim::block_radix_rank_algorithm::match>>;
template <class C, class KT>
float run(const char *name, KT *k_in, KT *k_out, double *v_in, double *v_out, size_t n, int b0, int b1, std::vector<double> *ref) {
  size_t t = 0;
  rocprim::radix_sort_pairs<C>(nullptr, t, k_in, k_out, v_in, v_out, n, b0, b1, 0);
  void *tmp;
  hipMalloc(&tmp, t);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e9f;
  for (int r = 0; r < 6; r++) {
    hipEventRecord(e0, 0);
    rocprim::radix_sort_pairs<C>(tmp, t, k_in, k_out, v_in, v_out, n, b0, b1, 0);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (r && ms < best) best = ms;
  }
  std::vector<double> h(n);
  hipMemcpy(h.data(), v_out, n * 8, hipMemcpyDeviceToHost);
  bool same = true;
  if (ref->empty()) *ref = h;
  else same = memcmp(ref->data(), h.data(), n * 8) == 0;
  printf("  %-28s %7.3f ms  %s\n", name, best, same ? "same permutation" : "DIFFERENT");
  hipFree(tmp);
  return best;
}
template <class KT>
void shape(const char *what, size_t n, int ib, int cb, unsigned long long cells) {
  printf("%s: %zu pairs, %d-byte keys, sorted on bits [%d, %d)\n", what, n, (int)sizeof(KT), ib, ib + cb);
  std::vector<KT> hk(n);
  std::vector<double> hv(n);
  std::mt19937_64 g(7);
  for (size_t i = 0; i < n; i++) {
    hk[i] = (KT)(((g() % cells) << ib) | (i * 1000 / n));  // ascending iteration in the low bits, as k_flux_keys writes them
    hv[i] = (double)i;                                      // the value names the element: equal outputs = equal permutations
  }
  KT *k_in, *k_out;
  double *v_in, *v_out;
  hipMalloc(&k_in, n * sizeof(KT));
  hipMalloc(&k_out, n * sizeof(KT));
  hipMalloc(&v_in, n * 8);
  hipMalloc(&v_out, n * 8);
  hipMemcpy(k_in, hk.data(), n * sizeof(KT), hipMemcpyHostToDevice);
  hipMemcpy(v_in, hv.data(), n * 8, hipMemcpyHostToDevice);
  std::vector<double> ref;
  run<rocprim::default_config, KT>("default", k_in, k_out, v_in, v_out, n, ib, ib + cb, &ref);
  run<Cfg<8, 256, 12>, KT>("match, 8 bits, 256 x 12", k_in, k_out, v_in, v_out, n, ib, ib + cb, &ref);
  run<Cfg<9, 256, 12>, KT>("match, 9 bits, 256 x 12", k_in, k_out, v_in, v_out, n, ib, ib + cb, &ref);
  run<Cfg<10, 256, 12>, KT>("match, 10 bits, 256 x 12", k_in, k_out, v_in, v_out, n, ib, ib + cb, &ref);
  run<Cfg<10, 512, 8>, KT>("match, 10 bits, 512 x 8", k_in, k_out, v_in, v_out, n, ib, ib + cb, &ref);
  run<Cfg<9, 512, 8>, KT>("match, 9 bits, 512 x 8", k_in, k_out, v_in, v_out, n, ib, ib + cb, &ref);
  hipFree(k_in);
  hipFree(k_out);
  hipFree(v_in);
  hipFree(v_out);
}
int main() {
  shape<unsigned>("C3", 10000000, 10, 20, 1002ull * 1002);
  shape<unsigned long long>("C5 (10 M of its segments)", 10000000, 10, 25, 5002ull * 5002);
  return 0;
}
