// attn_parts.hip -- the MFMA attention forward kernel (L = 200, d = 64, 4096 sequences: SASRec's shape) timed alone, whole
// and with parts compiled out (-DRBX_ATTN_ABL=bits, see rbx_attn_mfma.hip), to see which part of a tile step the time is in.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -DRBX_ATTN_ABL=0 profiles/ubench/attn_parts.hip -o attn_parts_0
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../recbox_amd/csrc/rbx_common.hip"
#include "../../recbox_amd/csrc/rbx_attn_mfma.hip"

int main(int argc, char** argv) {
  const int BH = 4096, L = argc > 1 ? atoi(argv[1]) : 200, HD = 64;
  const size_t n = static_cast<size_t>(BH) * L * HD;
  std::vector<float> h(n);
  unsigned x = 12345u;
  for (size_t i = 0; i < n; ++i) { x = x * 1664525u + 1013904223u; h[i] = (static_cast<float>(x >> 8) / 8388608.f - 1.f); }
  float *q, *k, *v, *o, *lse;
  (void)hipMalloc(&q, n * 4); (void)hipMalloc(&k, n * 4); (void)hipMalloc(&v, n * 4); (void)hipMalloc(&o, n * 4);
  (void)hipMalloc(&lse, static_cast<size_t>(BH) * L * 4);
  (void)hipMemcpy(q, h.data(), n * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(k, h.data(), n * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(v, h.data(), n * 4, hipMemcpyHostToDevice);
  rbx::DropArgs drop{};
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) rbx::attn_mfma_fwd(q, k, v, BH, L, HD, 0.125f, 1, o, lse, drop, nullptr);
  (void)hipDeviceSynchronize();
  const int reps = 20;
  (void)hipEventRecord(a);
  for (int i = 0; i < reps; ++i) rbx::attn_mfma_fwd(q, k, v, BH, L, HD, 0.125f, 1, o, lse, drop, nullptr);
  (void)hipEventRecord(b);
  (void)hipEventSynchronize(b);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, a, b);
  printf("ABL=%d L=%d forward %.1f us (rc %s)\n", RBX_ATTN_ABL, L, ms * 1000.f / reps,
         hipGetErrorString(hipGetLastError()));
  return 0;
}
