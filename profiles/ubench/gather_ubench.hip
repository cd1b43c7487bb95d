// gather_ubench.hip -- what random row gathers can reach on MI355X, as a yardstick for the
// embedding kernels.  Reads n random rows of `dim` floats (lane group of dim/4 lanes, float4 each,
// U rows in flight per lane) from a table far larger than the 256 MiB Infinity Cache and reports
// GB/s of row bytes.  Build: hipcc --offload-arch=gfx950 -O3 gather_ubench.hip -o gather_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int G, int U>
__global__ __launch_bounds__(256) void gather(const float* __restrict__ table, const unsigned* __restrict__ idx,
                                              long long n, int dim, float* __restrict__ out) {
  const int lane_g = threadIdx.x % G;
  const long long ngroups = (long long)gridDim.x * (blockDim.x / G);
  float4 acc = make_float4(0, 0, 0, 0);
  for (long long i0 = (long long)blockIdx.x * (blockDim.x / G) + threadIdx.x / G; i0 < n; i0 += ngroups * U) {
    unsigned id[U];
#pragma unroll
    for (int u = 0; u < U; ++u) id[u] = (i0 + u * ngroups < n) ? idx[i0 + u * ngroups] : 0u;
    float4 r[U];
#pragma unroll
    for (int u = 0; u < U; ++u) r[u] = *reinterpret_cast<const float4*>(table + (size_t)id[u] * dim + lane_g * 4);
#pragma unroll
    for (int u = 0; u < U; ++u) { acc.x += r[u].x; acc.y += r[u].y; acc.z += r[u].z; acc.w += r[u].w; }
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;   // keep the loads alive
}

__global__ void stream_copy(const float4* __restrict__ a, float4* __restrict__ b, long long n) {
  const long long s = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += s) b[i] = a[i];
}

template <int G>
static void run(const float* table, const unsigned* idx, long long n, int dim, float* out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int blocks = 256 * 8;
  for (int rep = 0; rep < 3; ++rep) gather<G, 8><<<blocks, 256>>>(table, idx, n, dim, out);
  hipEventRecord(e0);
  const int iters = 10;
  for (int rep = 0; rep < iters; ++rep) gather<G, 8><<<blocks, 256>>>(table, idx, n, dim, out);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double gbs = (double)n * dim * 4 * iters / (ms * 1e-3) / 1e9;
  printf("random rows of %4d B: %8.1f GB/s  (%.1f us per %lld rows, %.1f M rows/ms)\n", dim * 4, gbs, ms * 1e3 / iters,
         n, n / (ms / iters) / 1e3);
}

int main() {
  const size_t table_bytes = 4ull << 30;                 // 4 GiB >> 256 MiB MALL
  float* table;
  hipMalloc(&table, table_bytes);
  hipMemset(table, 0, table_bytes);
  const long long n = 1703936;                           // 65 536 x 26 lookups, the cfg-2 batch
  std::vector<unsigned> h(n);
  float* out;
  hipMalloc(&out, 64);
  unsigned* idx;
  hipMalloc(&idx, n * 4);
  for (int dim : {16, 32, 64, 128}) {
    const size_t rows = table_bytes / (dim * 4);
    unsigned long long s = 88172645463325252ull;
    for (auto& v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (unsigned)(s % rows); }
    hipMemcpy(idx, h.data(), n * 4, hipMemcpyHostToDevice);
    switch (dim) {
      case 16: run<4>(table, idx, n, dim, out); break;
      case 32: run<8>(table, idx, n, dim, out); break;
      case 64: run<16>(table, idx, n, dim, out); break;
      default: run<32>(table, idx, n, dim, out); break;
    }
  }
  {  // streaming copy yardstick
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const long long n4 = (1ll << 30) / 16;
    float4* a = reinterpret_cast<float4*>(table);
    float4* b = a + n4;
    for (int rep = 0; rep < 3; ++rep) stream_copy<<<256 * 8, 256>>>(a, b, n4);
    hipEventRecord(e0);
    for (int rep = 0; rep < 10; ++rep) stream_copy<<<256 * 8, 256>>>(a, b, n4);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("streaming float4 copy: %8.1f GB/s (read+write)\n", 2.0 * (1ll << 30) * 10 / (ms * 1e-3) / 1e9);
  }
  return 0;
}
