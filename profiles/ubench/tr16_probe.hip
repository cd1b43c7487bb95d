// tr16_probe.hip -- what ds_read_b64_tr_b16 delivers: LDS holds its own element index (u16), every lane passes the address
// of 4 contiguous elements, the four 16-bit results of every lane are printed.  Address patterns: 0 = lane-linear (lane l ->
// elements 4 l ..), 1 = a [4][16] row-major block per 16-lane group with a row pitch of 32 elements, 2 = per-lane scattered.
//   hipcc --offload-arch=gfx950 -O2 profiles/ubench/tr16_probe.hip -o profiles/ubench/tr16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short v4i16 __attribute__((ext_vector_type(4)));
__global__ void probe(unsigned short* out, int pattern) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = static_cast<unsigned short>(i);
  __syncthreads();
  const int l = threadIdx.x, g = l >> 4, i = l & 15;
  int e = 4 * l;
  if (pattern == 1) e = g * 1024 + (i >> 2) * 32 + 4 * (i & 3);
  if (pattern == 2) e = ((l * 37) % 64) * 8 + 4 * (l & 1) + 1024;
  v4i16 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4i16*)(lds + e));
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = static_cast<unsigned short>(r[j]);
  out[256 + l] = static_cast<unsigned short>(e);
}
int main() {
  unsigned short* d;
  (void)hipMalloc(&d, 1024);
  for (int p = 0; p < 3; ++p) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, p);
    unsigned short h[320];
    (void)hipMemcpy(h, d, 640, hipMemcpyDeviceToHost);
    printf("pattern %d (lane: address -> 4 results)\n", p);
    for (int l = 0; l < 64; ++l) printf("%2d: %5d -> %5d %5d %5d %5d\n", l, h[256 + l], h[4 * l], h[4 * l + 1], h[4 * l + 2], h[4 * l + 3]);
  }
  return 0;
}
