// rbx_tower.hip -- K5: two-tower scoring primitives (gfx950).
//
// Reference behaviour replaced (third_party/rechub/models/matching):
//   dssm.py:57,65 / youtube_dnn.py:56,65,70   F.normalize(x, p=2, dim=-1)   -> rbx_l2norm_fwd/bwd
//   dssm.py:48 / youtube_dnn.py:47-48          torch.mul(u, v).sum(-1) / T   -> rbx_pairdot_fwd/bwd
// One wavefront-slice (lane group of D/4 lanes, float4 per lane) owns one row / one
// (sample, candidate) pair; reductions over d are xor shuffles inside the group.  All four
// kernels are single-pass HBM streams: 2 reads + 1 write per element at most.
#include "rbx_internal.h"

namespace rbx {

// y = x / max(||x||, eps); inv[r] = 1/max(||x||, eps), stored NEGATIVE when the clamp was active
template <int G>
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const float* __restrict__ x, const long long rows,
                                                         const int D, const float eps, float* __restrict__ y,
                                                         float* __restrict__ inv) {
  const int lane_g = threadIdx.x % G;
  const long long ngroups = static_cast<long long>(gridDim.x) * (blockDim.x / G);
  for (long long r = static_cast<long long>(blockIdx.x) * (blockDim.x / G) + threadIdx.x / G; r < rows; r += ngroups) {
    const float* src = x + r * D;
    float ss = 0.f;
    for (int d = lane_g; d < D; d += G) ss += src[d] * src[d];
    ss = group_sum<G>(ss);
    const float nrm = sqrtf(ss);
    const bool clamped = nrm < eps;
    const float s = 1.0f / (clamped ? eps : nrm);
    for (int d = lane_g; d < D; d += G) y[r * D + d] = src[d] * s;
    if (lane_g == 0) inv[r] = clamped ? -s : s;
  }
}

// dx = inv * (dy - y * <y, dy>)   (plain dy * inv when the norm was clamped to eps)
template <int G>
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ y, const float* __restrict__ inv,
                                                         const float* __restrict__ dy, const long long rows,
                                                         const int D, float* __restrict__ dx) {
  const int lane_g = threadIdx.x % G;
  const long long ngroups = static_cast<long long>(gridDim.x) * (blockDim.x / G);
  for (long long r = static_cast<long long>(blockIdx.x) * (blockDim.x / G) + threadIdx.x / G; r < rows; r += ngroups) {
    const float s = inv[r];
    float dot = 0.f;
    if (s > 0.f) {
      for (int d = lane_g; d < D; d += G) dot += y[r * D + d] * dy[r * D + d];
    }
    dot = group_sum<G>(dot);
    const float a = fabsf(s);
    for (int d = lane_g; d < D; d += G) dx[r * D + d] = a * (dy[r * D + d] - y[r * D + d] * dot);
  }
}

// out[b, n] = scale * <u[b], v[b, n]>
template <int G>
__global__ __launch_bounds__(256) void pairdot_fwd_kernel(const float* __restrict__ u, const float* __restrict__ v,
                                                          const long long pairs, const int N, const int D,
                                                          const float scale, float* __restrict__ out) {
  const int lane_g = threadIdx.x % G;
  const long long ngroups = static_cast<long long>(gridDim.x) * (blockDim.x / G);
  for (long long p = static_cast<long long>(blockIdx.x) * (blockDim.x / G) + threadIdx.x / G; p < pairs; p += ngroups) {
    const long long b = p / N;
    float t = 0.f;
    for (int d = lane_g; d < D; d += G) t += u[b * D + d] * v[p * D + d];
    t = group_sum<G>(t);
    if (lane_g == 0) out[p] = t * scale;
  }
}

// dv[b,n] = scale * dout[b,n] * u[b];  du[b] = scale * sum_n dout[b,n] * v[b,n]  (one group per sample)
template <int G>
__global__ __launch_bounds__(256) void pairdot_bwd_kernel(const float* __restrict__ u, const float* __restrict__ v,
                                                          const float* __restrict__ dout, const long long B,
                                                          const int N, const int D, const float scale,
                                                          float* __restrict__ du, float* __restrict__ dv) {
  const int lane_g = threadIdx.x % G;
  const long long ngroups = static_cast<long long>(gridDim.x) * (blockDim.x / G);
  for (long long b = static_cast<long long>(blockIdx.x) * (blockDim.x / G) + threadIdx.x / G; b < B; b += ngroups) {
    for (int d = lane_g; d < D; d += G) {
      const float ud = u[b * D + d];
      float acc = 0.f;
      for (int n = 0; n < N; ++n) {
        const float g = dout[b * N + n] * scale;
        acc += g * v[(b * N + n) * D + d];
        if (dv != nullptr) dv[(b * N + n) * D + d] = g * ud;
      }
      if (du != nullptr) du[b * D + d] = acc;
    }
  }
}

static int pick_g(int D) {
  int g = pow2_ceil(D);
  return g > 64 ? 64 : g;
}

static unsigned grid_for(long long groups, int g) {
  const int gpb = 256 / g;
  long long blocks = (groups + gpb - 1) / gpb;
  if (blocks > kCUs * 8) blocks = kCUs * 8;
  if (blocks < 1) blocks = 1;
  return static_cast<unsigned>(blocks);
}

#define RBX_DISPATCH_G(g, CALL) \
  switch (g) {                  \
    case 1: CALL(1); break;     \
    case 2: CALL(2); break;     \
    case 4: CALL(4); break;     \
    case 8: CALL(8); break;     \
    case 16: CALL(16); break;   \
    case 32: CALL(32); break;   \
    default: CALL(64); break;   \
  }

}  // namespace rbx

extern "C" int rbx_l2norm_fwd(const float* d_x, int64_t rows, int32_t dim, float eps, float* d_y, float* d_inv,
                              void* stream) {
  using namespace rbx;
  if (d_x == nullptr || d_y == nullptr || d_inv == nullptr) return fail(RBX_ERR_INVALID, "l2norm: NULL tensor");
  if (rows < 0 || dim <= 0) return fail(RBX_ERR_INVALID, "l2norm: bad shape");
  if (rows == 0) return RBX_OK;
  const int g = pick_g(dim);
  hipStream_t s = as_stream(stream);
#define CALL(GG) hipLaunchKernelGGL((l2norm_fwd_kernel<GG>), dim3(grid_for(rows, GG)), dim3(256), 0, s, d_x, \
                                    static_cast<long long>(rows), dim, eps, d_y, d_inv)
  RBX_DISPATCH_G(g, CALL)
#undef CALL
  return check_launch("l2norm_fwd_kernel");
}

extern "C" int rbx_l2norm_bwd(const float* d_y, const float* d_inv, const float* d_dy, int64_t rows, int32_t dim,
                              float* d_dx, void* stream) {
  using namespace rbx;
  if (d_y == nullptr || d_inv == nullptr || d_dy == nullptr || d_dx == nullptr)
    return fail(RBX_ERR_INVALID, "l2norm_bwd: NULL tensor");
  if (rows == 0) return RBX_OK;
  const int g = pick_g(dim);
  hipStream_t s = as_stream(stream);
#define CALL(GG) hipLaunchKernelGGL((l2norm_bwd_kernel<GG>), dim3(grid_for(rows, GG)), dim3(256), 0, s, d_y, d_inv, \
                                    d_dy, static_cast<long long>(rows), dim, d_dx)
  RBX_DISPATCH_G(g, CALL)
#undef CALL
  return check_launch("l2norm_bwd_kernel");
}

extern "C" int rbx_pairdot_fwd(const float* d_u, const float* d_v, int64_t batch, int32_t n_cand, int32_t dim,
                               float scale, float* d_out, void* stream) {
  using namespace rbx;
  if (d_u == nullptr || d_v == nullptr || d_out == nullptr) return fail(RBX_ERR_INVALID, "pairdot: NULL tensor");
  if (batch < 0 || n_cand <= 0 || dim <= 0) return fail(RBX_ERR_INVALID, "pairdot: bad shape");
  if (batch == 0) return RBX_OK;
  const int g = pick_g(dim);
  const long long pairs = static_cast<long long>(batch) * n_cand;
  hipStream_t s = as_stream(stream);
#define CALL(GG) hipLaunchKernelGGL((pairdot_fwd_kernel<GG>), dim3(grid_for(pairs, GG)), dim3(256), 0, s, d_u, d_v, \
                                    pairs, n_cand, dim, scale, d_out)
  RBX_DISPATCH_G(g, CALL)
#undef CALL
  return check_launch("pairdot_fwd_kernel");
}

extern "C" int rbx_pairdot_bwd(const float* d_u, const float* d_v, const float* d_dout, int64_t batch,
                               int32_t n_cand, int32_t dim, float scale, float* d_du, float* d_dv, void* stream) {
  using namespace rbx;
  if (d_u == nullptr || d_v == nullptr || d_dout == nullptr) return fail(RBX_ERR_INVALID, "pairdot_bwd: NULL tensor");
  if (batch == 0) return RBX_OK;
  const int g = pick_g(dim);
  hipStream_t s = as_stream(stream);
#define CALL(GG) hipLaunchKernelGGL((pairdot_bwd_kernel<GG>), dim3(grid_for(batch, GG)), dim3(256), 0, s, d_u, d_v, \
                                    d_dout, static_cast<long long>(batch), n_cand, dim, scale, d_du, d_dv)
  RBX_DISPATCH_G(g, CALL)
#undef CALL
  return check_launch("pairdot_bwd_kernel");
}
