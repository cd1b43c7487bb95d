// Yardstick: rocPRIM's radix_sort_pairs on the backward's key set (1.7 M (key, val) pairs, 23 key bits) against the
// in-tree 3-pass LSD sort (build_keys + 12 kernels).   hipcc -O3 --offload-arch=gfx950 rocprim_sort.hip -o rocprim_sort
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <cstdio>
#include <vector>
#include <random>

int main() {
  const unsigned n = 26u * 65536u;
  const unsigned rows = 5569296u;
  std::vector<unsigned> hk(n), hv(n);
  std::mt19937 g(1);
  for (unsigned i = 0; i < n; ++i) { hk[i] = g() % rows; hv[i] = i; }
  unsigned *k0, *k1, *v0, *v1;
  hipMalloc(&k0, n * 4); hipMalloc(&k1, n * 4); hipMalloc(&v0, n * 4); hipMalloc(&v1, n * 4);
  hipMemcpy(k0, hk.data(), n * 4, hipMemcpyHostToDevice);
  hipMemcpy(v0, hv.data(), n * 4, hipMemcpyHostToDevice);
  for (int bits : {23, 24, 32}) {
    size_t tmp_bytes = 0;
    rocprim::radix_sort_pairs(nullptr, tmp_bytes, k0, k1, v0, v1, n, 0, bits, 0);
    void* tmp;
    hipMalloc(&tmp, tmp_bytes);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int w = 0; w < 5; ++w) rocprim::radix_sort_pairs(tmp, tmp_bytes, k0, k1, v0, v1, n, 0, bits, 0);
    hipDeviceSynchronize();
    const int it = 50;
    hipEventRecord(a, 0);
    for (int i = 0; i < it; ++i) rocprim::radix_sort_pairs(tmp, tmp_bytes, k0, k1, v0, v1, n, 0, bits, 0);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    std::vector<unsigned> out(n);
    hipMemcpy(out.data(), k1, n * 4, hipMemcpyDeviceToHost);
    bool ok = true;
    for (unsigned i = 1; i < n; ++i) if (out[i - 1] > out[i]) { ok = false; break; }
    printf("rocprim radix_sort_pairs n=%u bits=%d: %.1f us per sort (temp %zu B, sorted=%d)\n", n, bits, ms * 1000.f / it,
           tmp_bytes, (int)ok);
    hipFree(tmp);
  }
  return 0;
}
