// gather_ceiling.hip -- round 3: how firm is "a random 64-byte row costs a 128-byte line" on MI355X?
// (the round-1 yardstick's own streaming copy reached 4.75 TB/s where MI355X_MICROARCH.md records
// 6.29; "0.38 of peak is the ceiling for 64-byte rows" rested on that kernel alone.)
//   1. streaming: float4 read-only sweep and read+write copy, 8 loads in flight per lane, non-temporal stores;
//   2. random rows of 64 / 128 / 256 B from a 4 GiB table (>> 256 MiB Infinity Cache), lane group of row/16 lanes:
//        register form, U = 8 / 16 / 32 rows in flight per lane (VGPRs cap the queue depth);
//        LDS-DMA form: global_load_lds_dwordx4 -- 16 bytes per lane straight into LDS, no VGPR per row in flight --
//        Q = 16 / 32 wave-instructions outstanding per wavefront (256 / 512 rows of 64 B per wavefront).
// Build: hipcc --offload-arch=gfx950 -O3 gather_ceiling.hip -o gather_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void stream_read(const v4f* __restrict__ a, long long n, float* __restrict__ out) {
  const long long s = (long long)gridDim.x * blockDim.x;
  v4f acc = {0, 0, 0, 0};
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 7 * s < n; i += 8 * s) {
    v4f r[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) r[u] = __builtin_nontemporal_load(a + i + u * s);
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += r[u];
  }
  for (; i < n; i += s) acc += a[i];
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}

__global__ __launch_bounds__(256) void stream_copy(const v4f* __restrict__ a, v4f* __restrict__ b, long long n) {
  const long long s = (long long)gridDim.x * blockDim.x;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 7 * s < n; i += 8 * s) {
    v4f r[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) r[u] = __builtin_nontemporal_load(a + i + u * s);
#pragma unroll
    for (int u = 0; u < 8; ++u) __builtin_nontemporal_store(r[u], b + i + u * s);
  }
  for (; i < n; i += s) b[i] = a[i];
}

template <int G, int U>
__global__ __launch_bounds__(256) void gather_reg(const float* __restrict__ table, const unsigned* __restrict__ idx,
                                                  long long n, int dim, float* __restrict__ out) {
  const int lane_g = threadIdx.x % G;
  const long long ngroups = (long long)gridDim.x * (blockDim.x / G);
  v4f acc = {0, 0, 0, 0};
  for (long long i0 = (long long)blockIdx.x * (blockDim.x / G) + threadIdx.x / G; i0 < n; i0 += ngroups * U) {
    unsigned id[U];
#pragma unroll
    for (int u = 0; u < U; ++u) id[u] = (i0 + u * ngroups < n) ? idx[i0 + u * ngroups] : 0u;
    v4f r[U];
#pragma unroll
    for (int u = 0; u < U; ++u) r[u] = *reinterpret_cast<const v4f*>(table + (size_t)id[u] * dim + lane_g * 4);
#pragma unroll
    for (int u = 0; u < U; ++u) acc += r[u];
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}

// LDS-DMA: each wave-instruction moves 64 x 16 B = 1 KiB (64 / G rows) from per-lane addresses into a wave-private LDS slot
template <int G, int Q>
__global__ __launch_bounds__(256) void gather_lds(const float* __restrict__ table, const unsigned* __restrict__ idx,
                                                  long long n, int dim, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  char* slot = smem + (size_t)wid * Q * 1024;                 // Q KiB per wavefront
  const int lane_g = lane % G, row_in = lane / G;              // 64 / G rows per instruction
  constexpr int RPI = 64 / G;
  const long long nwaves = (long long)gridDim.x * 4;
  const long long per_step = (long long)RPI * Q;
  v4f acc = {0, 0, 0, 0};
  for (long long base = ((long long)blockIdx.x * 4 + wid) * per_step; base < n; base += nwaves * per_step) {
    unsigned id[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const long long i = base + (long long)q * RPI + row_in;
      id[q] = (i < n) ? idx[i] : 0u;
    }
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const float* src = table + (size_t)id[q] * dim + lane_g * 4;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(slot + q * 1024), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int q = 0; q < Q; ++q) acc += *reinterpret_cast<const v4f*>(slot + q * 1024 + lane * 16);
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}

template <class L>
static float time_us(L launch, int iters = 10) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int r = 0; r < 3; ++r) launch();
  hipEventRecord(e0);
  for (int r = 0; r < iters; ++r) launch();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f / iters;
}

int main() {
  const size_t table_bytes = 4ull << 30;
  float* table;
  hipMalloc(&table, table_bytes);
  hipMemset(table, 0, table_bytes);
  float* out;
  hipMalloc(&out, 64);
  {
    const long long n4 = (2ll << 30) / 16;                 // 2 GiB read (+ 2 GiB written by the copy)
    v4f* a = reinterpret_cast<v4f*>(table);
    v4f* b = a + n4;
    for (int blocks : {2048, 4096, 8192}) {
      const float tr = time_us([&] { stream_read<<<blocks, 256>>>(a, n4, out); });
      const float tc = time_us([&] { stream_copy<<<blocks, 256>>>(a, b, n4); });
      printf("streaming, %5d workgroups: read-only %7.1f GB/s   copy %7.1f GB/s (read + write)\n", blocks,
             (double)n4 * 16 / (tr * 1e-6) / 1e9, 2.0 * n4 * 16 / (tc * 1e-6) / 1e9);
    }
  }
  const long long n = 1703936 * 4;                          // 4 x the cfg-2 batch's lookups: ~150 us launches
  std::vector<unsigned> h(n);
  unsigned* idx;
  hipMalloc(&idx, n * 4);
  hipFuncSetAttribute(reinterpret_cast<const void*>(gather_lds<4, 32>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  hipFuncSetAttribute(reinterpret_cast<const void*>(gather_lds<8, 32>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  for (int dim : {16, 32, 64}) {
    const size_t rows = table_bytes / (dim * 4);
    unsigned long long s = 88172645463325252ull;
    for (auto& v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (unsigned)(s % rows); }
    hipMemcpy(idx, h.data(), n * 4, hipMemcpyHostToDevice);
    auto report = [&](const char* what, float us) {
      printf("random rows of %3d B, %-34s %7.1f GB/s of row bytes  %6.1f G rows/s  (128-B lines: %6.1f GB/s)\n", dim * 4, what,
             (double)n * dim * 4 / (us * 1e-6) / 1e9, n / (us * 1e-6) / 1e9,
             (double)n * (dim * 4 < 128 ? 128 : dim * 4) / (us * 1e-6) / 1e9);
    };
    const int blocks = 256 * 8;
    if (dim == 16) {
      report("registers, 8 rows in flight / lane", time_us([&] { gather_reg<4, 8><<<blocks, 256>>>(table, idx, n, dim, out); }));
      report("registers, 16 rows in flight / lane", time_us([&] { gather_reg<4, 16><<<blocks, 256>>>(table, idx, n, dim, out); }));
      report("registers, 32 rows in flight / lane", time_us([&] { gather_reg<4, 32><<<blocks, 256>>>(table, idx, n, dim, out); }));
      report("LDS-DMA, 16 KiB in flight / wave", time_us([&] { gather_lds<4, 16><<<blocks, 256, 64 * 1024>>>(table, idx, n, dim, out); }));
      report("LDS-DMA, 32 KiB in flight / wave", time_us([&] { gather_lds<4, 32><<<256 * 4, 256, 128 * 1024>>>(table, idx, n, dim, out); }));
    } else if (dim == 32) {
      report("registers, 8 rows in flight / lane", time_us([&] { gather_reg<8, 8><<<blocks, 256>>>(table, idx, n, dim, out); }));
      report("registers, 16 rows in flight / lane", time_us([&] { gather_reg<8, 16><<<blocks, 256>>>(table, idx, n, dim, out); }));
      report("LDS-DMA, 16 KiB in flight / wave", time_us([&] { gather_lds<8, 16><<<blocks, 256, 64 * 1024>>>(table, idx, n, dim, out); }));
      report("LDS-DMA, 32 KiB in flight / wave", time_us([&] { gather_lds<8, 32><<<256 * 4, 256, 128 * 1024>>>(table, idx, n, dim, out); }));
    } else {
      report("registers, 8 rows in flight / lane", time_us([&] { gather_reg<16, 8><<<blocks, 256>>>(table, idx, n, dim, out); }));
      report("LDS-DMA, 16 KiB in flight / wave", time_us([&] { gather_lds<16, 16><<<blocks, 256, 64 * 1024>>>(table, idx, n, dim, out); }));
    }
  }
  return 0;
}
