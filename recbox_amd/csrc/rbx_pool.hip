// rbx_pool.hip -- sequence pooling of an already materialised [B, L, D] tensor
// (gfx950).  The fused gather+pool lives in rbx_embed_fwd.hip; this entry point
// keeps the standalone pooling modules drop-in:
//   recbox MaskedAveragePooling / MaskedSumPooling
//     (core/pytorch/layers/sequence.py:4-20, ranking/pytorch/layers/pooling.py:22-40)
//   rechub AveragePooling / SumPooling (third_party/rechub/basic/layers.py:176-230)
// numer_masked: numerator is sum_l mask[b,l]*E[b,l,:] (rechub bmm) instead of the plain sum.
// denom: 0 -> 1;  1 -> #rows with sum_d E != 0 (recbox value mask);  2 -> sum_l mask[b,l];  3 -> L.
// out = numer / (denom + eps); inv[b] = 1/(denom+eps) is kept for the backward:
//   dE[b,l,:] = (numer_masked ? mask[b,l] : 1) * inv[b] * dY[b,:]
#include "rbx_internal.h"

namespace rbx {

template <int G>
__global__ __launch_bounds__(256) void pool_fwd_kernel(const float* __restrict__ emb, const float* __restrict__ mask,
                                                       const long long B, const int L, const int D,
                                                       const int numer_masked, const int denom, const float eps,
                                                       float* __restrict__ out, float* __restrict__ inv) {
  const int lane_g = threadIdx.x % G;
  const long long ngroups = static_cast<long long>(gridDim.x) * (blockDim.x / G);
  for (long long b = static_cast<long long>(blockIdx.x) * (blockDim.x / G) + threadIdx.x / G; b < B; b += ngroups) {
    const float* base = emb + b * L * D;
    float cnt = 0.f;
    for (int d0 = 0; d0 < D; d0 += G) {         // D > G handled by looping the lane group over d
      const int d = d0 + lane_g;
      float acc = 0.f;
      for (int l = 0; l < L; ++l) {
        const float v = (d < D) ? base[l * D + d] : 0.f;
        const float m = (mask != nullptr) ? mask[b * L + l] : 1.f;
        acc += numer_masked ? m * v : v;
      }
      if (d < D) out[b * D + d] = acc;           // scaled below once the denominator is known
    }
    if (denom == 1) {
      for (int l = 0; l < L; ++l) {
        float rs = 0.f;
        for (int d = lane_g; d < D; d += G) rs += base[l * D + d];
        rs = group_sum<G>(rs);
        cnt += (rs != 0.f) ? 1.f : 0.f;
      }
    } else if (denom == 2) {
      for (int l = 0; l < L; ++l) cnt += mask[b * L + l];
    } else if (denom == 3) {
      cnt = static_cast<float>(L);
    }
    const float s = (denom == 0) ? 1.f : 1.f / (cnt + eps);
    if (denom != 0) {
      for (int d = lane_g; d < D; d += G) out[b * D + d] *= s;
    }
    if (inv != nullptr && lane_g == 0) inv[b] = s;
  }
}

__global__ __launch_bounds__(256) void pool_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ mask,
                                                       const float* __restrict__ inv, const long long total,
                                                       const int L, const int D, const int numer_masked,
                                                       float* __restrict__ demb) {
  const long long stride = static_cast<long long>(gridDim.x) * blockDim.x;
  for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += stride) {
    const long long bl = i / D;
    const int d = static_cast<int>(i - bl * D);
    const long long b = bl / L;
    float w = inv[b];
    if (numer_masked && mask != nullptr) w *= mask[bl];
    demb[i] = w * dout[b * D + d];
  }
}

}  // namespace rbx

extern "C" int rbx_pool_fwd(const float* d_emb, const float* d_mask, int64_t batch, int32_t seq_len, int32_t dim,
                            int32_t numer_masked, int32_t denom, float eps, float* d_out, float* d_inv, void* stream) {
  if (batch == 0) return RBX_OK;   // empty batch: nothing to do, pointers may be NULL
  using namespace rbx;
  if (d_emb == nullptr || d_out == nullptr || d_inv == nullptr) return fail(RBX_ERR_INVALID, "NULL tensor");
  if (batch < 0 || seq_len <= 0 || dim <= 0) return fail(RBX_ERR_INVALID, "bad shape");
  if (denom < 0 || denom > 3) return fail(RBX_ERR_INVALID, "bad denom mode %d", denom);
  if ((numer_masked || denom == 2) && d_mask == nullptr) return fail(RBX_ERR_INVALID, "mask required");
  if (batch == 0) return RBX_OK;
  int g = pow2_ceil(dim);
  if (g > 64) g = 64;
  const int gpb = 256 / g;
  long long blocks = (batch + gpb - 1) / gpb;
  if (blocks > kCUs * 8) blocks = kCUs * 8;
  hipStream_t s = as_stream(stream);
#define RBX_POOL_LAUNCH(GG)                                                                                        \
  hipLaunchKernelGGL((pool_fwd_kernel<GG>), dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s, d_emb, d_mask, \
                     static_cast<long long>(batch), seq_len, dim, numer_masked, denom, eps, d_out, d_inv)
  switch (g) {
    case 1: RBX_POOL_LAUNCH(1); break;
    case 2: RBX_POOL_LAUNCH(2); break;
    case 4: RBX_POOL_LAUNCH(4); break;
    case 8: RBX_POOL_LAUNCH(8); break;
    case 16: RBX_POOL_LAUNCH(16); break;
    case 32: RBX_POOL_LAUNCH(32); break;
    default: RBX_POOL_LAUNCH(64); break;
  }
#undef RBX_POOL_LAUNCH
  return check_launch("pool_fwd_kernel");
}

extern "C" int rbx_pool_bwd(const float* d_dout, const float* d_mask, const float* d_inv, int64_t batch,
                            int32_t seq_len, int32_t dim, int32_t numer_masked, float* d_demb, void* stream) {
  if (batch == 0) return RBX_OK;   // empty batch: nothing to do, pointers may be NULL
  using namespace rbx;
  if (d_dout == nullptr || d_inv == nullptr || d_demb == nullptr) return fail(RBX_ERR_INVALID, "NULL tensor");
  if (numer_masked && d_mask == nullptr) return fail(RBX_ERR_INVALID, "mask required");
  if (batch == 0) return RBX_OK;
  const long long total = static_cast<long long>(batch) * seq_len * dim;
  long long blocks = (total + 255) / 256;
  if (blocks > kCUs * 8) blocks = kCUs * 8;
  hipLaunchKernelGGL(pool_bwd_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, as_stream(stream), d_dout,
                     d_mask, d_inv, total, seq_len, dim, numer_masked, d_demb);
  return check_launch("pool_bwd_kernel");
}
