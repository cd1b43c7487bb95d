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
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const float* __restrict__ x, const long long inner,
                                                         const long long outer, const long long rows,
                                                         const int D, const float eps, float* __restrict__ y,
                                                         float* __restrict__ inv) {
  const int lane_g = threadIdx.x % G;
  const long long ngroups = static_cast<long long>(gridDim.x) * (blockDim.x / G);
  for (long long r = static_cast<long long>(blockIdx.x) * (blockDim.x / G) + threadIdx.x / G; r < rows; r += ngroups) {
    // row r = (r / inner, r % inner) of a [., inner, D] block whose outer stride is `outer` floats (inner == rows: plain)
    const float* src = (inner == 1) ? x + r * outer : x + (r / inner) * outer + (r % inner) * D;
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

// ---- normalise the candidates and score them in ONE pass (YoutubeDNN: youtube_dnn.py:56,65,70 -- F.normalize of the
// [B, 1 + n_neg, D] item rows, then the inner product with the normalised user vector) -------------------------------
// out[b, n] = scale * <u[b], v[b, n]> / max(|v[b, n]|, eps);  inv[b, n] = 1 / max(|v|, eps), NEGATIVE when the clamp was active.
// The normalised rows are never written (168 MB at cfg 3) nor read back by a second kernel.  v rows sit at
// v + b * v_outer + n * D: behind the user columns of the gathered block, read in place.
template <int G>
__global__ __launch_bounds__(256) void cosdot_fwd_kernel(const float* __restrict__ u, const float* __restrict__ v,
                                                         const long long v_outer, const long long pairs, const int N,
                                                         const int D, const float eps, const float scale,
                                                         float* __restrict__ out, float* __restrict__ inv) {
  const int lane_g = threadIdx.x % G;
  const long long ngroups = static_cast<long long>(gridDim.x) * (blockDim.x / G);
  for (long long p = static_cast<long long>(blockIdx.x) * (blockDim.x / G) + threadIdx.x / G; p < pairs; p += ngroups) {
    const long long b = p / N;
    const float* vr = v + b * v_outer + (p - b * N) * D;
    float ss = 0.f, t = 0.f;
    for (int d = lane_g; d < D; d += G) {
      const float x = vr[d];
      ss += x * x;
      t += u[b * D + d] * x;
    }
    ss = group_sum<G>(ss);
    t = group_sum<G>(t);
    const float nrm = sqrtf(ss);
    const bool clamped = nrm < eps;
    const float s = 1.0f / (clamped ? eps : nrm);
    if (lane_g == 0) {
      out[p] = t * s * scale;
      inv[p] = clamped ? -s : s;
    }
  }
}

// Backward, one lane group per sample (du is summed over its candidates in order):
//   vh = v * |inv|;  c = <u, vh>;  g = scale * dout[b, n]
//   dv = |inv| * g * (u - vh * c)      (|inv| * g * u when the norm was clamped: the l2norm backward's rule)
//   du[b] = sum_n g * vh
// dv rows are written at dv + b * dv_outer + n * D (the gradient block mirrors the layout the rows were read from).
template <int G>
__global__ __launch_bounds__(256) void cosdot_bwd_kernel(const float* __restrict__ u, const float* __restrict__ v,
                                                         const long long v_outer, const float* __restrict__ inv,
                                                         const float* __restrict__ dout, const long long B, const int N,
                                                         const int D, const float scale, float* __restrict__ du,
                                                         float* __restrict__ dv, const long long dv_outer) {
  const int lane_g = threadIdx.x % G;
  const long long ngroups = static_cast<long long>(gridDim.x) * (blockDim.x / G);
  for (long long b = static_cast<long long>(blockIdx.x) * (blockDim.x / G) + threadIdx.x / G; b < B; b += ngroups) {
    for (int n = 0; n < N; ++n) {
      const float si = inv[b * N + n];
      const float a = fabsf(si);
      const float* vr = v + b * v_outer + static_cast<long long>(n) * D;
      float c = 0.f;
      if (si > 0.f) {
        for (int d = lane_g; d < D; d += G) c += u[b * D + d] * (vr[d] * a);
        c = group_sum<G>(c);
      }
      const float g = dout[b * N + n] * scale;
      if (dv != nullptr) {
        float* dr = dv + b * dv_outer + static_cast<long long>(n) * D;
        for (int d = lane_g; d < D; d += G) dr[d] = a * g * (u[b * D + d] - (si > 0.f ? vr[d] * a * c : 0.f));
      }
    }
    if (du != nullptr) {
      for (int d = lane_g; d < D; d += G) {
        float acc = 0.f;
        for (int n = 0; n < N; ++n)
          acc += dout[b * N + n] * scale * (v[b * v_outer + static_cast<long long>(n) * D + d] * fabsf(inv[b * N + n]));
        du[b * D + d] = acc;
      }
    }
  }
}

// float4 variants (D % 4 == 0, rows 16-byte aligned): G = lanes per row, each lane owns D / (4 G) float4 of it (NV);
// a lane group takes one SAMPLE and issues the loads of all its candidate rows (CH at a time) before the first reduction.
// The scalar kernels above keep one row per group with two dependent 4-byte loads per lane and, in the backward, walk a
// sample's candidates one after the other: 1.7-1.8 TB/s at cfg 3 ([65 536, 5, 128]: 93 / 193 us).
typedef float cos_f4 __attribute__((ext_vector_type(4)));
template <int G, int NV>
__global__ __launch_bounds__(256) void cosdot_fwd_vec_kernel(const float* __restrict__ u, const float* __restrict__ v,
                                                             const long long v_outer, const long long B, const int N,
                                                             const int D, const float eps, const float scale,
                                                             float* __restrict__ out, float* __restrict__ inv) {
  constexpr int CH = 4;
  const int lane_g = threadIdx.x % G;
  const long long ngroups = static_cast<long long>(gridDim.x) * (blockDim.x / G);
  for (long long b = static_cast<long long>(blockIdx.x) * (blockDim.x / G) + threadIdx.x / G; b < B; b += ngroups) {
    cos_f4 uu[NV];
#pragma unroll
    for (int q = 0; q < NV; ++q) uu[q] = *reinterpret_cast<const cos_f4*>(u + b * D + (lane_g + q * G) * 4);
    for (int n0 = 0; n0 < N; n0 += CH) {
      cos_f4 x[CH][NV];
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const int n = n0 + c < N ? n0 + c : N - 1;
#pragma unroll
        for (int q = 0; q < NV; ++q)
          x[c][q] = *reinterpret_cast<const cos_f4*>(v + b * v_outer + static_cast<long long>(n) * D + (lane_g + q * G) * 4);
      }
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        float ss = 0.f, t = 0.f;
#pragma unroll
        for (int q = 0; q < NV; ++q)
#pragma unroll
          for (int j = 0; j < 4; ++j) { ss += x[c][q][j] * x[c][q][j]; t += uu[q][j] * x[c][q][j]; }
        ss = group_sum<G>(ss);
        t = group_sum<G>(t);
        const float nrm = sqrtf(ss);
        const bool clamped = nrm < eps;
        const float si = 1.0f / (clamped ? eps : nrm);
        if (lane_g == 0 && n0 + c < N) {
          out[b * N + n0 + c] = t * si * scale;
          inv[b * N + n0 + c] = clamped ? -si : si;
        }
      }
    }
  }
}

template <int G, int NV>
__global__ __launch_bounds__(256) void cosdot_bwd_vec_kernel(const float* __restrict__ u, const float* __restrict__ v,
                                                             const long long v_outer, const float* __restrict__ inv,
                                                             const float* __restrict__ dout, const long long B, const int N,
                                                             const int D, const float scale, float* __restrict__ du,
                                                             float* __restrict__ dv, const long long dv_outer) {
  constexpr int CH = 4;
  const int lane_g = threadIdx.x % G;
  const long long ngroups = static_cast<long long>(gridDim.x) * (blockDim.x / G);
  for (long long b = static_cast<long long>(blockIdx.x) * (blockDim.x / G) + threadIdx.x / G; b < B; b += ngroups) {
    cos_f4 uu[NV], acc[NV];
#pragma unroll
    for (int q = 0; q < NV; ++q) {
      uu[q] = *reinterpret_cast<const cos_f4*>(u + b * D + (lane_g + q * G) * 4);
      acc[q] = cos_f4{0.f, 0.f, 0.f, 0.f};
    }
    for (int n0 = 0; n0 < N; n0 += CH) {
      cos_f4 x[CH][NV];
      float si[CH], g[CH];
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const int n = n0 + c < N ? n0 + c : N - 1;
        si[c] = inv[b * N + n];
        g[c] = dout[b * N + n] * scale;
#pragma unroll
        for (int q = 0; q < NV; ++q)
          x[c][q] = *reinterpret_cast<const cos_f4*>(v + b * v_outer + static_cast<long long>(n) * D + (lane_g + q * G) * 4);
      }
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        if (n0 + c >= N) break;                     // (wave-uniform: N is)
        const float a = fabsf(si[c]);
        float cc = 0.f;
#pragma unroll
        for (int q = 0; q < NV; ++q)
#pragma unroll
          for (int j = 0; j < 4; ++j) cc += uu[q][j] * (x[c][q][j] * a);
        cc = group_sum<G>(cc);
        if (!(si[c] > 0.f)) cc = 0.f;
        const bool live = si[c] > 0.f;
#pragma unroll
        for (int q = 0; q < NV; ++q) {
          cos_f4 vh, d;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            vh[j] = x[c][q][j] * a;
            d[j] = a * g[c] * (uu[q][j] - (live ? vh[j] * cc : 0.f));
            acc[q][j] += g[c] * vh[j];
          }
          if (dv != nullptr)
            *reinterpret_cast<cos_f4*>(dv + b * dv_outer + static_cast<long long>(n0 + c) * D + (lane_g + q * G) * 4) = d;
        }
      }
    }
    if (du != nullptr) {
#pragma unroll
      for (int q = 0; q < NV; ++q) *reinterpret_cast<cos_f4*>(du + b * D + (lane_g + q * G) * 4) = acc[q];
    }
  }
}

static int pick_g(int D) {
  int g = pow2_ceil(D);
  return g > 64 ? 64 : g;
}

// lane group / float4-per-lane of the vector kernels for a row of D floats: (G, NV) with G * NV * 4 == D, or G = 0
static void pick_vec(int D, int* G, int* NV) {
  *G = 0;
  *NV = 0;
  if (D % 4 != 0) return;
  const int q = D / 4;                              // float4 per row
  if (q <= 64 && (q & (q - 1)) == 0) { *G = q; *NV = 1; }
  else if (q % 64 == 0 && q / 64 <= 4) { *G = 64; *NV = q / 64; }
}
static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

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
  if (rows == 0) return RBX_OK;   // empty batch: nothing to do, pointers may be NULL
  using namespace rbx;
  if (d_x == nullptr || d_y == nullptr || d_inv == nullptr) return fail(RBX_ERR_INVALID, "l2norm: NULL tensor");
  if (rows < 0 || dim <= 0) return fail(RBX_ERR_INVALID, "l2norm: bad shape");
  if (rows == 0) return RBX_OK;
  const int g = pick_g(dim);
  hipStream_t s = as_stream(stream);
#define CALL(GG) hipLaunchKernelGGL((l2norm_fwd_kernel<GG>), dim3(grid_for(rows, GG)), dim3(256), 0, s, d_x, 1LL, \
                                    static_cast<long long>(dim), static_cast<long long>(rows), dim, eps, d_y, d_inv)
  RBX_DISPATCH_G(g, CALL)
#undef CALL
  return check_launch("l2norm_fwd_kernel");
}

extern "C" int rbx_l2norm_fwd_strided(const float* d_x, int64_t inner, int64_t outer_stride, int64_t rows, int32_t dim,
                                      float eps, float* d_y, float* d_inv, void* stream) {
  if (rows == 0) return RBX_OK;
  using namespace rbx;
  if (d_x == nullptr || d_y == nullptr || d_inv == nullptr) return fail(RBX_ERR_INVALID, "l2norm: NULL tensor");
  if (rows < 0 || dim <= 0 || inner <= 0 || rows % inner != 0 || outer_stride < inner * dim)
    return fail(RBX_ERR_INVALID, "l2norm_strided: rows %lld must be whole groups of %lld rows, outer stride >= inner * dim",
                static_cast<long long>(rows), static_cast<long long>(inner));
  const int g = pick_g(dim);
  hipStream_t s = as_stream(stream);
#define CALL(GG) hipLaunchKernelGGL((l2norm_fwd_kernel<GG>), dim3(grid_for(rows, GG)), dim3(256), 0, s, d_x, \
                                    static_cast<long long>(inner), static_cast<long long>(outer_stride), \
                                    static_cast<long long>(rows), dim, eps, d_y, d_inv)
  RBX_DISPATCH_G(g, CALL)
#undef CALL
  return check_launch("l2norm_fwd_kernel");
}

extern "C" int rbx_l2norm_bwd(const float* d_y, const float* d_inv, const float* d_dy, int64_t rows, int32_t dim,
                              float* d_dx, void* stream) {
  if (rows == 0) return RBX_OK;   // empty batch: nothing to do, pointers may be NULL
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

extern "C" int rbx_cosdot_fwd(const float* d_u, const float* d_v, int64_t v_outer_stride, int64_t batch, int32_t n_cand,
                              int32_t dim, float eps, float scale, float* d_out, float* d_inv, void* stream) {
  if (batch == 0) return RBX_OK;
  using namespace rbx;
  if (!d_u || !d_v || !d_out || !d_inv) return fail(RBX_ERR_INVALID, "cosdot: NULL tensor");
  if (batch < 0 || n_cand <= 0 || dim <= 0 || v_outer_stride < static_cast<int64_t>(n_cand) * dim)
    return fail(RBX_ERR_INVALID, "cosdot: bad shape / outer stride shorter than n_cand * dim");
  const int g = pick_g(dim);
  const long long pairs = static_cast<long long>(batch) * n_cand;
  hipStream_t s = as_stream(stream);
  {
    int G, NV;
    pick_vec(dim, &G, &NV);
    if (G >= 2 && aligned16(d_u) && aligned16(d_v) && v_outer_stride % 4 == 0) {
#define VCALL(GG, VV) hipLaunchKernelGGL((cosdot_fwd_vec_kernel<GG, VV>), dim3(grid_for(batch, GG)), dim3(256), 0, s, d_u, d_v, \
                                         static_cast<long long>(v_outer_stride), static_cast<long long>(batch), n_cand, dim, eps, \
                                         scale, d_out, d_inv)
      if (NV == 1) { switch (G) { case 2: VCALL(2, 1); break; case 4: VCALL(4, 1); break; case 8: VCALL(8, 1); break;
                                  case 16: VCALL(16, 1); break; case 32: VCALL(32, 1); break; default: VCALL(64, 1); break; } }
      else if (NV == 2) VCALL(64, 2);
      else if (NV == 3) VCALL(64, 3);
      else VCALL(64, 4);
#undef VCALL
      return check_launch("cosdot_fwd_vec_kernel");
    }
  }
#define CALL(GG) hipLaunchKernelGGL((cosdot_fwd_kernel<GG>), dim3(grid_for(pairs, GG)), dim3(256), 0, s, d_u, d_v, \
                                    static_cast<long long>(v_outer_stride), pairs, n_cand, dim, eps, scale, d_out, d_inv)
  RBX_DISPATCH_G(g, CALL)
#undef CALL
  return check_launch("cosdot_fwd_kernel");
}

extern "C" int rbx_cosdot_bwd(const float* d_u, const float* d_v, int64_t v_outer_stride, const float* d_inv,
                              const float* d_dout, int64_t batch, int32_t n_cand, int32_t dim, float scale, float* d_du,
                              float* d_dv, int64_t dv_outer_stride, void* stream) {
  if (batch == 0) return RBX_OK;
  using namespace rbx;
  if (!d_u || !d_v || !d_inv || !d_dout) return fail(RBX_ERR_INVALID, "cosdot_bwd: NULL tensor");
  if (batch < 0 || n_cand <= 0 || dim <= 0 || v_outer_stride < static_cast<int64_t>(n_cand) * dim ||
      (d_dv != nullptr && dv_outer_stride < static_cast<int64_t>(n_cand) * dim))
    return fail(RBX_ERR_INVALID, "cosdot_bwd: bad shape / outer stride shorter than n_cand * dim");
  const int g = pick_g(dim);
  hipStream_t s = as_stream(stream);
  {
    int G, NV;
    pick_vec(dim, &G, &NV);
    if (G >= 2 && aligned16(d_u) && aligned16(d_v) && v_outer_stride % 4 == 0 && (d_du == nullptr || aligned16(d_du)) &&
        (d_dv == nullptr || (aligned16(d_dv) && dv_outer_stride % 4 == 0))) {
#define VCALL(GG, VV) hipLaunchKernelGGL((cosdot_bwd_vec_kernel<GG, VV>), dim3(grid_for(batch, GG)), dim3(256), 0, s, d_u, d_v, \
                                         static_cast<long long>(v_outer_stride), d_inv, d_dout, static_cast<long long>(batch), \
                                         n_cand, dim, scale, d_du, d_dv, static_cast<long long>(dv_outer_stride))
      if (NV == 1) { switch (G) { case 2: VCALL(2, 1); break; case 4: VCALL(4, 1); break; case 8: VCALL(8, 1); break;
                                  case 16: VCALL(16, 1); break; case 32: VCALL(32, 1); break; default: VCALL(64, 1); break; } }
      else if (NV == 2) VCALL(64, 2);
      else if (NV == 3) VCALL(64, 3);
      else VCALL(64, 4);
#undef VCALL
      return check_launch("cosdot_bwd_vec_kernel");
    }
  }
#define CALL(GG) hipLaunchKernelGGL((cosdot_bwd_kernel<GG>), dim3(grid_for(batch, GG)), dim3(256), 0, s, d_u, d_v, \
                                    static_cast<long long>(v_outer_stride), d_inv, d_dout, static_cast<long long>(batch), \
                                    n_cand, dim, scale, d_du, d_dv, static_cast<long long>(dv_outer_stride))
  RBX_DISPATCH_G(g, CALL)
#undef CALL
  return check_launch("cosdot_bwd_kernel");
}

extern "C" int rbx_pairdot_fwd(const float* d_u, const float* d_v, int64_t batch, int32_t n_cand, int32_t dim,
                               float scale, float* d_out, void* stream) {
  if (batch == 0) return RBX_OK;   // empty batch: nothing to do, pointers may be NULL
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
  if (batch == 0) return RBX_OK;   // empty batch: nothing to do, pointers may be NULL
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

// ---- cross layers: X_{i+1} = X_i + X_0 * h (+ bias) ------------------------------------------
// ranking/pytorch/layers/interactions/cross_net.py:48-59 (CrossNetV2: h = Linear_i(X_i), [B, dim]) and
// :22-46 (CrossNet: h = X_i w_i, [B, 1], plus a bias vector).  The Linear is the fp32 MFMA GEMM; this is the
// element-wise tail fused into one pass each way (the reference runs mul, add (, add) and their three backward
// kernels).  h_cols == 1 broadcasts h over the row.  HBM-bound streaming: 3 reads + 1 write forward.
namespace rbx {

template <bool VEC>
__global__ __launch_bounds__(256) void cross_fwd_kernel(const float* __restrict__ x0, const float* __restrict__ xi,
                                                        const float* __restrict__ h, const float* __restrict__ bias,
                                                        const long long rows, const int dim, const int h_cols,
                                                        float* __restrict__ out) {
  constexpr int W = VEC ? 4 : 1;
  const long long total = rows * dim / W;
  const long long step = static_cast<long long>(gridDim.x) * blockDim.x;
  for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += step) {
    const long long e = i * W;
    const long long r = e / dim;
    const int c = static_cast<int>(e - r * dim);
    float a[W], b[W], hv[W], o[W];
    if constexpr (VEC) {
      const float4 t0 = *reinterpret_cast<const float4*>(x0 + e);
      const float4 t1 = *reinterpret_cast<const float4*>(xi + e);
      a[0] = t0.x; a[1] = t0.y; a[2] = t0.z; a[3] = t0.w;
      b[0] = t1.x; b[1] = t1.y; b[2] = t1.z; b[3] = t1.w;
      if (h_cols == 1) {
        hv[0] = hv[1] = hv[2] = hv[3] = h[r];
      } else {
        const float4 t2 = *reinterpret_cast<const float4*>(h + e);
        hv[0] = t2.x; hv[1] = t2.y; hv[2] = t2.z; hv[3] = t2.w;
      }
    } else {
      a[0] = x0[e];
      b[0] = xi[e];
      hv[0] = (h_cols == 1) ? h[r] : h[e];
    }
#pragma unroll
    for (int k = 0; k < W; ++k) o[k] = b[k] + (a[k] * hv[k] + (bias != nullptr ? bias[c + k] : 0.f));
    if constexpr (VEC) *reinterpret_cast<float4*>(out + e) = make_float4(o[0], o[1], o[2], o[3]);
    else out[e] = o[0];
  }
}

// dx0 = g * h, dh = g * x0 (row-summed when h is [rows, 1]); dxi = g and dbias = colsum(g) need no kernel here
template <int G>
__global__ __launch_bounds__(256) void cross_bwd_kernel(const float* __restrict__ x0, const float* __restrict__ h,
                                                        const float* __restrict__ g, const long long rows, const int dim,
                                                        const int h_cols, float* __restrict__ dx0,
                                                        float* __restrict__ dh) {
  // a lane group of G lanes walks one row (needed for the row sum of the broadcast case; coalesced either way)
  const int lane_g = threadIdx.x % G;
  const long long ngroups = static_cast<long long>(gridDim.x) * (blockDim.x / G);
  for (long long r = static_cast<long long>(blockIdx.x) * (blockDim.x / G) + threadIdx.x / G; r < rows; r += ngroups) {
    const float hb = (h_cols == 1) ? h[r] : 0.f;
    float acc = 0.f;
    for (int c = lane_g; c < dim; c += G) {
      const long long e = r * dim + c;
      const float gv = g[e], xv = x0[e];
      const float hv = (h_cols == 1) ? hb : h[e];
      if (dx0 != nullptr) dx0[e] = gv * hv;
      if (h_cols == 1) acc += gv * xv;
      else dh[e] = gv * xv;
    }
    if (h_cols == 1) {
      acc = group_sum<G>(acc);
      if (lane_g == 0) dh[r] = acc;
    }
  }
}

}  // namespace rbx

extern "C" int rbx_cross_fwd(const float* d_x0, const float* d_xi, const float* d_h, const float* d_bias, int64_t rows,
                             int32_t dim, int32_t h_cols, float* d_out, void* stream) {
  if (rows == 0) return RBX_OK;   // empty batch: nothing to do, pointers may be NULL
  using namespace rbx;
  if (rows < 0 || dim <= 0 || (h_cols != 1 && h_cols != dim)) return fail(RBX_ERR_INVALID, "cross: bad shape");
  if (rows == 0) return RBX_OK;
  if (!d_x0 || !d_xi || !d_h || !d_out) return fail(RBX_ERR_INVALID, "cross: NULL tensor");
  const bool vec = dim % 4 == 0 && ((reinterpret_cast<uintptr_t>(d_x0) | reinterpret_cast<uintptr_t>(d_xi) |
                                     reinterpret_cast<uintptr_t>(d_h) | reinterpret_cast<uintptr_t>(d_out)) & 15) == 0;
  const long long total = static_cast<long long>(rows) * dim / (vec ? 4 : 1);
  long long blocks = (total + 255) / 256;
  if (blocks > kCUs * 16) blocks = kCUs * 16;
  if (vec)
    hipLaunchKernelGGL(cross_fwd_kernel<true>, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, as_stream(stream), d_x0, d_xi,
                       d_h, d_bias, static_cast<long long>(rows), dim, h_cols, d_out);
  else
    hipLaunchKernelGGL(cross_fwd_kernel<false>, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, as_stream(stream), d_x0, d_xi,
                       d_h, d_bias, static_cast<long long>(rows), dim, h_cols, d_out);
  return check_launch("cross_fwd_kernel");
}

extern "C" int rbx_cross_bwd(const float* d_x0, const float* d_h, const float* d_dout, int64_t rows, int32_t dim,
                             int32_t h_cols, float* d_dx0, float* d_dh, void* stream) {
  if (rows == 0) return RBX_OK;   // empty batch: nothing to do, pointers may be NULL
  using namespace rbx;
  if (rows < 0 || dim <= 0 || (h_cols != 1 && h_cols != dim)) return fail(RBX_ERR_INVALID, "cross_bwd: bad shape");
  if (rows == 0) return RBX_OK;
  if (!d_x0 || !d_h || !d_dout || !d_dh) return fail(RBX_ERR_INVALID, "cross_bwd: NULL tensor");
  long long blocks = (rows + 3) / 4;                       // 4 lane groups of 64 per workgroup
  if (blocks > kCUs * 16) blocks = kCUs * 16;
  hipLaunchKernelGGL(cross_bwd_kernel<64>, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, as_stream(stream), d_x0, d_h,
                     d_dout, static_cast<long long>(rows), dim, h_cols, d_dx0, d_dh);
  return check_launch("cross_bwd_kernel");
}

// ---- row scaling: out[r, :] = (alpha * x[r, :] + add[r, :]) * s[r] ----------------------------------------------
// SASRec's embedding prologue and per-block timeline mask (third_party/rechub/models/matching/sasrec.py:68-77, 92):
// `e *= sqrt(D); e += position_emb(...); e *= ~timeline_mask.unsqueeze(-1)` are three element-wise ATen passes, the
// broadcast one over [B, L, 1] x [B, L, D] un-vectorised (159 us at [819200, 64] against 64 us for a plain pass); here one.
namespace rbx {
// add_period > 0: `add` holds add_period floats that repeat (the position rows [L, D] of every sequence of a [B, L, D] block)
__global__ __launch_bounds__(256) void rowscale_kernel(const float* __restrict__ x, const float* __restrict__ add,
                                                       const float* __restrict__ s, const long long rows, const int dim,
                                                       const float alpha, float* __restrict__ out,
                                                       const long long add_period) {
  const long long total = rows * dim;
  const long long step = static_cast<long long>(gridDim.x) * blockDim.x * 4;
  for (long long i = (static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x) * 4; i < total; i += step) {
    if ((dim & 3) == 0) {                                  // a float4 never straddles two rows
      const float sc = s[i / dim];
      typedef float v4f __attribute__((ext_vector_type(4)));
      const v4f vv = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(x + i));       // streamed once
      const float4 v = make_float4(vv[0], vv[1], vv[2], vv[3]);
      float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
      if (add != nullptr) a = *reinterpret_cast<const float4*>(add + (add_period > 0 ? i % add_period : i));
      *reinterpret_cast<float4*>(out + i) = make_float4((alpha * v.x + a.x) * sc, (alpha * v.y + a.y) * sc,
                                                        (alpha * v.z + a.z) * sc, (alpha * v.w + a.w) * sc);
    } else {
      for (long long j = i; j < i + 4 && j < total; ++j)
        out[j] = (alpha * x[j] + (add != nullptr ? add[add_period > 0 ? j % add_period : j] : 0.f)) * s[j / dim];
    }
  }
}
}  // namespace rbx

extern "C" int rbx_rowscale(const float* d_x, const float* d_add, const float* d_scale, int64_t rows, int32_t dim, float alpha,
                            float* d_out, void* stream) {
  using namespace rbx;
  if (rows == 0 || dim == 0) return RBX_OK;
  if (rows < 0 || dim < 0) return fail(RBX_ERR_INVALID, "rowscale: negative sizes");
  if (!d_x || !d_scale || !d_out) return fail(RBX_ERR_INVALID, "rowscale: NULL tensor");
  if ((dim & 3) == 0 && (((reinterpret_cast<uintptr_t>(d_x) | reinterpret_cast<uintptr_t>(d_out) |
                           reinterpret_cast<uintptr_t>(d_add)) & 15) != 0))
    return fail(RBX_ERR_INVALID, "rowscale: tensors must be 16-byte aligned when dim is a multiple of 4");
  long long blocks = (rows * dim / 4 + 255) / 256;
  if (blocks > kCUs * 32) blocks = kCUs * 32;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(rowscale_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, as_stream(stream), d_x, d_add, d_scale,
                     static_cast<long long>(rows), dim, alpha, d_out, 0LL);
  return check_launch("rowscale_kernel");
}

// ---- SASRec's input block with the position rows read in place -------------------------------------------------------------
// sasrec.py:68-77: `positions = tile(arange(L), [B, 1]); seqs = item_emb(seq) * sqrt(D) + position_emb(positions);
// seqs *= ~timeline_mask`.  The positions of every sequence are 0 .. L-1, so the looked-up block is the table's first L rows
// B times over: the forward reads those rows (51 KB, cache resident) instead of a gathered [B, L, D] copy, and the table's
// gradient dP[l, :] = sum_b keep[b, l] g[b, l, :] is a column sum over the batch of the incoming gradient -- one streaming
// pass and a fixed-order sum of the block partials, where the generic embedding backward sorts B L equal-by-thousands ids and
// reduces 200 rows of 4096 duplicates each (sort 55 us + reduce and fix-ups 140 us + the pass that wrote keep * g 83 us).
namespace rbx {
constexpr int kSeqSumRows = 32;                 // sequences per block partial

template <int W>
__global__ __launch_bounds__(256) void seq_colsum_partial_kernel(const float* __restrict__ g, const float* __restrict__ s,
                                                                 const long long batch, const int L, const int dim,
                                                                 float* __restrict__ partial) {
  const long long cols = static_cast<long long>(L) * dim;
  const long long c = (static_cast<long long>(blockIdx.x) * 256 + threadIdx.x) * W;
  if (c >= cols) return;
  const int l = static_cast<int>(c / dim);                          // (W = 4: dim % 4 == 0, the four columns share a row)
  const long long b0 = static_cast<long long>(blockIdx.y) * kSeqSumRows;
  const long long b1 = b0 + kSeqSumRows < batch ? b0 + kSeqSumRows : batch;
  float acc[W];
#pragma unroll
  for (int k = 0; k < W; ++k) acc[k] = 0.f;
  constexpr int U = 8;
  for (long long b = b0; b < b1; b += U) {                          // U rows in flight, added in ascending order
    float v[U][W], sc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long bb = b + u < b1 ? b + u : b1 - 1;
      sc[u] = b + u < b1 ? s[bb * L + l] : 0.f;
      if constexpr (W == 4) {
        typedef float v4f __attribute__((ext_vector_type(4)));
        const v4f t = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(g + bb * cols + c));
        v[u][0] = t[0]; v[u][1] = t[1]; v[u][2] = t[2]; v[u][3] = t[3];
      } else {
        v[u][0] = g[bb * cols + c];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int k = 0; k < W; ++k) acc[k] += sc[u] * v[u][k];
  }
#pragma unroll
  for (int k = 0; k < W; ++k) partial[static_cast<long long>(blockIdx.y) * cols + c + k] = acc[k];
}

__global__ __launch_bounds__(256) void seq_colsum_final_kernel(const float* __restrict__ partial, const int n_blocks,
                                                               const long long cols, float* __restrict__ out) {
  const long long c = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;
  if (c >= cols) return;
  float t = 0.f;
  int k = 0;
  for (; k + 7 < n_blocks; k += 8) {                                // 8 partials in flight, added in ascending order
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = partial[static_cast<long long>(k + u) * cols + c];
#pragma unroll
    for (int u = 0; u < 8; ++u) t += v[u];
  }
  for (; k < n_blocks; ++k) t += partial[static_cast<long long>(k) * cols + c];
  out[c] = t;
}
}  // namespace rbx

extern "C" int rbx_rowscale_seq(const float* d_x, const float* d_add, int64_t add_rows, const float* d_scale, int64_t rows,
                                int32_t dim, float alpha, float* d_out, void* stream) {
  using namespace rbx;
  if (rows == 0 || dim == 0) return RBX_OK;
  if (rows < 0 || dim < 0 || add_rows <= 0 || rows % add_rows != 0)
    return fail(RBX_ERR_INVALID, "rowscale_seq: rows must be a multiple of add_rows > 0");
  if (!d_x || !d_add || !d_scale || !d_out) return fail(RBX_ERR_INVALID, "rowscale_seq: NULL tensor");
  if ((dim & 3) == 0 && (((reinterpret_cast<uintptr_t>(d_x) | reinterpret_cast<uintptr_t>(d_out) |
                           reinterpret_cast<uintptr_t>(d_add)) & 15) != 0))
    return fail(RBX_ERR_INVALID, "rowscale_seq: tensors must be 16-byte aligned when dim is a multiple of 4");
  long long blocks = (rows * dim / 4 + 255) / 256;
  if (blocks > kCUs * 32) blocks = kCUs * 32;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(rowscale_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, as_stream(stream), d_x, d_add, d_scale,
                     static_cast<long long>(rows), dim, alpha, d_out, static_cast<long long>(add_rows) * dim);
  return check_launch("rowscale_kernel (periodic add)");
}

extern "C" size_t rbx_seq_colsum_workspace_size(int64_t batch, int32_t seq_len, int32_t dim) {
  if (batch <= 0 || seq_len <= 0 || dim <= 0) return 0;
  const size_t nb = static_cast<size_t>((batch + rbx::kSeqSumRows - 1) / rbx::kSeqSumRows);
  return nb * static_cast<size_t>(seq_len) * dim * sizeof(float);
}

extern "C" int rbx_seq_colsum(const float* d_g, const float* d_scale, int64_t batch, int32_t seq_len, int32_t dim, float* d_out,
                              void* d_workspace, size_t workspace_bytes, void* stream) {
  using namespace rbx;
  if (seq_len <= 0 || dim <= 0 || batch < 0) return fail(RBX_ERR_INVALID, "seq_colsum: bad shape");
  if (!d_out) return fail(RBX_ERR_INVALID, "seq_colsum: NULL output");
  hipStream_t s = as_stream(stream);
  const long long cols = static_cast<long long>(seq_len) * dim;
  if (batch == 0) {
    (void)hipMemsetAsync(d_out, 0, cols * sizeof(float), s);
    return check_launch("seq_colsum (empty batch)");
  }
  if (!d_g || !d_scale) return fail(RBX_ERR_INVALID, "seq_colsum: NULL tensor");
  if (d_workspace == nullptr || workspace_bytes < rbx_seq_colsum_workspace_size(batch, seq_len, dim))
    return fail(RBX_ERR_WORKSPACE, "seq_colsum: workspace too small");
  const int nb = static_cast<int>((batch + kSeqSumRows - 1) / kSeqSumRows);
  float* partial = static_cast<float*>(d_workspace);
  const bool vec = (dim & 3) == 0 && (reinterpret_cast<uintptr_t>(d_g) & 15) == 0;
  if (vec)
    hipLaunchKernelGGL(seq_colsum_partial_kernel<4>, dim3(static_cast<unsigned>((cols / 4 + 255) / 256), nb), dim3(256), 0, s,
                       d_g, d_scale, static_cast<long long>(batch), seq_len, dim, partial);
  else
    hipLaunchKernelGGL(seq_colsum_partial_kernel<1>, dim3(static_cast<unsigned>((cols + 255) / 256), nb), dim3(256), 0, s,
                       d_g, d_scale, static_cast<long long>(batch), seq_len, dim, partial);
  hipLaunchKernelGGL(seq_colsum_final_kernel, dim3(static_cast<unsigned>((cols + 255) / 256)), dim3(256), 0, s, partial, nb,
                     cols, d_out);
  return check_launch("seq_colsum kernels");
}

// ---- gradient of a row block that is read whole AND through its leading columns ---------------------------------
// third_party/rechub/models/ranking/deepfm.py:34-39 feeds the SAME embeddings to the tower (whole row: embeddings | dense
// values), to FM and to the first-order Linear (leading F*D columns).  Autograd's glue for that fan-out is a zero fill, a
// strided copy and two adds over [B, F*D] (669 us at B = 65 536, F*D = 1664); here one pass:
//   out[r, c] = base[r, c] + (c < prefix_cols ? a[r, c] + b[r, c] : 0),   c < cols      (base / a / b optional)
namespace rbx {
template <bool VEC>
__global__ __launch_bounds__(256) void sum_prefix_kernel(const float* __restrict__ base, const long long base_stride,
                                                         const float* __restrict__ a, const long long a_stride,
                                                         const float* __restrict__ b, const long long b_stride,
                                                         const long long rows, const int cols, const int prefix_cols,
                                                         float* __restrict__ out, const long long out_stride) {
  constexpr int W = VEC ? 4 : 1;
  const unsigned per_row = static_cast<unsigned>((cols + W - 1) / W);
  const long long total = rows * per_row;
  const long long step = static_cast<long long>(gridDim.x) * blockDim.x;
  for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += step) {
    long long r;
    if (total < (1LL << 32)) r = static_cast<unsigned>(i) / per_row;       // 32-bit division whenever it fits
    else r = i / per_row;
    const int c = static_cast<int>(i - r * per_row) * W;
    float v[W];
#pragma unroll
    for (int k = 0; k < W; ++k) v[k] = 0.f;
    if (base != nullptr) {
      if constexpr (VEC) {
        const float4 t = *reinterpret_cast<const float4*>(base + r * base_stride + c);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
      } else {
        v[0] = base[r * base_stride + c];
      }
    }
    if (c < prefix_cols) {
      if (a != nullptr) {
        if constexpr (VEC) {
          const float4 t = *reinterpret_cast<const float4*>(a + r * a_stride + c);
          v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w;
        } else {
          v[0] += a[r * a_stride + c];
        }
      }
      if (b != nullptr) {
        if constexpr (VEC) {
          const float4 t = *reinterpret_cast<const float4*>(b + r * b_stride + c);
          v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w;
        } else {
          v[0] += b[r * b_stride + c];
        }
      }
    }
    if constexpr (VEC) *reinterpret_cast<float4*>(out + r * out_stride + c) = make_float4(v[0], v[1], v[2], v[3]);
    else out[r * out_stride + c] = v[0];
  }
}
}  // namespace rbx

extern "C" int rbx_sum_prefix(const float* d_base, int64_t base_stride, const float* d_a, int64_t a_stride, const float* d_b,
                              int64_t b_stride, int64_t rows, int32_t cols, int32_t prefix_cols, float* d_out,
                              int64_t out_stride, void* stream) {
  using namespace rbx;
  if (rows < 0 || cols < 0 || prefix_cols < 0 || prefix_cols > cols)
    return fail(RBX_ERR_INVALID, "sum_prefix: need rows >= 0 and 0 <= prefix_cols <= cols");
  if (rows == 0 || cols == 0) return RBX_OK;
  if (!d_out) return fail(RBX_ERR_INVALID, "sum_prefix: NULL output");
  if (out_stride < cols || (d_base && base_stride < cols) || (d_a && a_stride < prefix_cols) || (d_b && b_stride < prefix_cols))
    return fail(RBX_ERR_INVALID, "sum_prefix: a row stride is shorter than its row");
  // float4 sweep when every row starts 16-byte aligned, the prefix ends on a float4 and the padded tail of the last
  // float4 of a row lies inside the row stride of the tensors that hold whole rows
  const int padded = (cols + 3) / 4 * 4;
  const uintptr_t bits = reinterpret_cast<uintptr_t>(d_base) | reinterpret_cast<uintptr_t>(d_a) |
                         reinterpret_cast<uintptr_t>(d_b) | reinterpret_cast<uintptr_t>(d_out);
  const bool vec = (bits & 15) == 0 && (prefix_cols & 3) == 0 && (out_stride & 3) == 0 && out_stride >= padded &&
                   (!d_base || ((base_stride & 3) == 0 && base_stride >= padded)) && (!d_a || (a_stride & 3) == 0) &&
                   (!d_b || (b_stride & 3) == 0);
  const long long total = static_cast<long long>(rows) * (vec ? padded / 4 : cols);
  long long blocks = (total + 255) / 256;
  if (blocks > kCUs * 16) blocks = kCUs * 16;
  if (vec)
    hipLaunchKernelGGL(sum_prefix_kernel<true>, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, as_stream(stream), d_base,
                       static_cast<long long>(base_stride), d_a, static_cast<long long>(a_stride), d_b,
                       static_cast<long long>(b_stride), static_cast<long long>(rows), cols, prefix_cols, d_out,
                       static_cast<long long>(out_stride));
  else
    hipLaunchKernelGGL(sum_prefix_kernel<false>, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, as_stream(stream), d_base,
                       static_cast<long long>(base_stride), d_a, static_cast<long long>(a_stride), d_b,
                       static_cast<long long>(b_stride), static_cast<long long>(rows), cols, prefix_cols, d_out,
                       static_cast<long long>(out_stride));
  return check_launch("sum_prefix_kernel");
}

// ---- binary cross entropy on probabilities, mean-reduced (the ranking harness's loss) -------------------------
// ranking/pytorch/models/ranking_model.py:69 + ranking/pytorch/torch_utils.py:54-65: F.binary_cross_entropy(y_pred,
// y_true, reduction='mean') on SIGMOID OUTPUTS.  torch semantics: log terms clamped at -100; backward
// (p - y) / max(p (1 - p), 1e-12) * g / N.  ATen runs the element-wise loss, a 16 us generic reduction of 65 536
// values, a division and two backward kernels; here: one pass with block partials, a fixed-order final sum, one
// backward pass.  Deterministic (no atomics).
namespace rbx {

constexpr int kBceBlock = 1024;       // elements per workgroup of the forward pass

__global__ __launch_bounds__(256) void bce_partial_kernel(const float* __restrict__ p, const float* __restrict__ y,
                                                          const long long n, float* __restrict__ partial) {
  __shared__ float red[4];
  const long long base = static_cast<long long>(blockIdx.x) * kBceBlock;
  float acc = 0.f;
#pragma unroll
  for (int k = 0; k < kBceBlock / 256; ++k) {
    const long long i = base + k * 256 + threadIdx.x;
    if (i < n) {
      const float pi = p[i], yi = y[i];
      const float lp = fmaxf(logf(pi), -100.f), lq = fmaxf(log1pf(-pi), -100.f);
      acc -= yi * lp + (1.f - yi) * lq;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// one workgroup: fixed-order sum of the block partials, times 1/n
__global__ __launch_bounds__(256) void bce_final_kernel(const float* __restrict__ partial, const int nblocks, const float inv_n,
                                                        float* __restrict__ loss) {
  __shared__ float red[4];
  float acc = 0.f;
  for (int i = threadIdx.x; i < nblocks; i += 256) acc += partial[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) loss[0] = ((red[0] + red[1]) + (red[2] + red[3])) * inv_n;
}

__global__ __launch_bounds__(256) void bce_bwd_kernel(const float* __restrict__ p, const float* __restrict__ y,
                                                      const float* __restrict__ gloss, const long long n, const float inv_n,
                                                      float* __restrict__ dp) {
  const float g = gloss[0] * inv_n;
  const long long step = static_cast<long long>(gridDim.x) * blockDim.x;
  for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += step) {
    const float pi = p[i];
    dp[i] = g * (pi - y[i]) / fmaxf((1.f - pi) * pi, 1e-12f);
  }
}

// sigmoid + BCE + both backward steps in ONE pass (the step of recbox_amd.graph.ShardedFMStep owns its loss, so the
// nine small kernels of sigmoid -> BCE -> backward -> sigmoid backward collapse into this one and bce_final_kernel):
// p = 1 / (1 + exp(-x)); the loss term and dL/dp exactly as above; dL/dx = dL/dp * (1 - p) * p (torch's sigmoid backward).
__global__ __launch_bounds__(256) void sigmoid_bce_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                          const long long n, const float gscale, float* __restrict__ prob,
                                                          float* __restrict__ dx, float* __restrict__ partial) {
  __shared__ float red[4];
  const long long base = static_cast<long long>(blockIdx.x) * kBceBlock;
  float acc = 0.f;
#pragma unroll
  for (int k = 0; k < kBceBlock / 256; ++k) {
    const long long i = base + k * 256 + threadIdx.x;
    if (i < n) {
      const float pi = 1.f / (1.f + expf(-x[i])), yi = y[i];
      const float lp = fmaxf(logf(pi), -100.f), lq = fmaxf(log1pf(-pi), -100.f);
      acc -= yi * lp + (1.f - yi) * lq;
      if (prob != nullptr) prob[i] = pi;
      if (dx != nullptr) {
        const float dp = gscale * (pi - yi) / fmaxf((1.f - pi) * pi, 1e-12f);
        dx[i] = dp * (1.f - pi) * pi;
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}


// The same pass with the final sum folded in: the workgroup that arrives LAST (device-scope counter) adds the block partials
// in the fixed order bce_final_kernel uses -- the result does not depend on which workgroup that is -- and leaves the counter
// at zero for the next call.  One launch instead of two in a chain of 5 us kernels.
__global__ __launch_bounds__(256) void sigmoid_bce_onepass_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                                  const long long n, const float gscale, const float inv_n,
                                                                  float* __restrict__ prob, float* __restrict__ dx,
                                                                  float* __restrict__ partial, unsigned* __restrict__ counter,
                                                                  float* __restrict__ loss) {
  __shared__ float red[4];
  __shared__ bool last;
  const long long base = static_cast<long long>(blockIdx.x) * kBceBlock;
  float acc = 0.f;
#pragma unroll
  for (int k = 0; k < kBceBlock / 256; ++k) {
    const long long i = base + k * 256 + threadIdx.x;
    if (i < n) {
      const float pi = 1.f / (1.f + expf(-x[i])), yi = y[i];
      const float lp = fmaxf(logf(pi), -100.f), lq = fmaxf(log1pf(-pi), -100.f);
      acc -= yi * lp + (1.f - yi) * lq;
      if (prob != nullptr) prob[i] = pi;
      if (dx != nullptr) {
        const float dp = gscale * (pi - yi) / fmaxf((1.f - pi) * pi, 1e-12f);
        dx[i] = dp * (1.f - pi) * pi;
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_store(partial + blockIdx.x, (red[0] + red[1]) + (red[2] + red[3]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence();
    last = (atomicAdd(counter, 1u) == gridDim.x - 1);
  }
  __syncthreads();
  if (!last) return;
  __threadfence();
  float t = 0.f;
  for (int i = threadIdx.x; i < static_cast<int>(gridDim.x); i += 256)
    t += __hip_atomic_load(partial + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = t;
  __syncthreads();
  if (threadIdx.x == 0) {
    loss[0] = ((red[0] + red[1]) + (red[2] + red[3])) * inv_n;
    counter[0] = 0;
  }
}

}  // namespace rbx

extern "C" int rbx_sigmoid_bce_mean_onepass(const float* d_logit, const float* d_target, int64_t n, float grad_scale,
                                            float* d_prob, float* d_loss, float* d_dlogit, void* d_workspace,
                                            size_t workspace_bytes, uint32_t* d_counter, void* stream) {
  using namespace rbx;
  if (n <= 0) return fail(RBX_ERR_INVALID, "sigmoid_bce: empty input (the mean of no elements is undefined)");
  if (!d_logit || !d_target || !d_loss || !d_counter) return fail(RBX_ERR_INVALID, "sigmoid_bce: NULL tensor");
  if (d_workspace == nullptr || workspace_bytes < rbx_bce_workspace_size(n))
    return fail(RBX_ERR_WORKSPACE, "sigmoid_bce: workspace too small");
  const long long nb = (n + kBceBlock - 1) / kBceBlock;
  if (nb >= INT_MAX) return fail(RBX_ERR_UNSUPPORTED, "sigmoid_bce: too many elements");
  const float inv_n = 1.0f / static_cast<float>(n);
  hipLaunchKernelGGL(sigmoid_bce_onepass_kernel, dim3(static_cast<unsigned>(nb)), dim3(256), 0, as_stream(stream), d_logit,
                     d_target, static_cast<long long>(n), grad_scale * inv_n, inv_n, d_prob, d_dlogit,
                     static_cast<float*>(d_workspace), d_counter, d_loss);
  return check_launch("sigmoid_bce_onepass_kernel");
}

extern "C" int rbx_sigmoid_bce_mean(const float* d_logit, const float* d_target, int64_t n, float grad_scale, float* d_prob,
                                    float* d_loss, float* d_dlogit, void* d_workspace, size_t workspace_bytes,
                                    void* stream) {
  using namespace rbx;
  if (n <= 0) return fail(RBX_ERR_INVALID, "sigmoid_bce: empty input (the mean of no elements is undefined)");
  if (!d_logit || !d_target || !d_loss) return fail(RBX_ERR_INVALID, "sigmoid_bce: NULL tensor");
  if (d_workspace == nullptr || workspace_bytes < rbx_bce_workspace_size(n))
    return fail(RBX_ERR_WORKSPACE, "sigmoid_bce: workspace too small");
  const long long nb = (n + kBceBlock - 1) / kBceBlock;
  if (nb >= INT_MAX) return fail(RBX_ERR_UNSUPPORTED, "sigmoid_bce: too many elements");
  float* partial = static_cast<float*>(d_workspace);
  const float inv_n = 1.0f / static_cast<float>(n);
  hipLaunchKernelGGL(sigmoid_bce_kernel, dim3(static_cast<unsigned>(nb)), dim3(256), 0, as_stream(stream), d_logit, d_target,
                     static_cast<long long>(n), grad_scale * inv_n, d_prob, d_dlogit, partial);
  hipLaunchKernelGGL(bce_final_kernel, dim3(1), dim3(256), 0, as_stream(stream), partial, static_cast<int>(nb), inv_n, d_loss);
  return check_launch("sigmoid_bce kernels");
}

namespace rbx {
__global__ __launch_bounds__(256) void scale_by_scalar_kernel(const float* __restrict__ x, const float* __restrict__ scalar,
                                                              const long long n, float* __restrict__ y) {
  const float g = scalar[0];
  const long long step = static_cast<long long>(gridDim.x) * blockDim.x;
  for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += step) y[i] = g * x[i];
}
}  // namespace rbx

extern "C" int rbx_scale_by_scalar(const float* d_x, const float* d_scalar, int64_t n, float* d_y, void* stream) {
  using namespace rbx;
  if (n <= 0) return RBX_OK;
  if (!d_x || !d_scalar || !d_y) return fail(RBX_ERR_INVALID, "scale_by_scalar: NULL tensor");
  long long blocks = (n + 255) / 256;
  if (blocks > kCUs * 8) blocks = kCUs * 8;
  hipLaunchKernelGGL(scale_by_scalar_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, as_stream(stream), d_x,
                     d_scalar, static_cast<long long>(n), d_y);
  return check_launch("scale_by_scalar_kernel");
}

extern "C" size_t rbx_bce_workspace_size(int64_t n) {
  return n > 0 ? static_cast<size_t>((n + rbx::kBceBlock - 1) / rbx::kBceBlock) * sizeof(float) + 256 : 0;
}

extern "C" int rbx_bce_mean_fwd(const float* d_prob, const float* d_target, int64_t n, float* d_loss, void* d_workspace,
                                size_t workspace_bytes, void* stream) {
  using namespace rbx;
  if (n <= 0) return fail(RBX_ERR_INVALID, "bce: empty input (the mean of no elements is undefined)");
  if (!d_prob || !d_target || !d_loss) return fail(RBX_ERR_INVALID, "bce: NULL tensor");
  if (d_workspace == nullptr || workspace_bytes < rbx_bce_workspace_size(n)) return fail(RBX_ERR_WORKSPACE, "bce: workspace too small");
  const long long nb = (n + kBceBlock - 1) / kBceBlock;
  if (nb >= INT_MAX) return fail(RBX_ERR_UNSUPPORTED, "bce: too many elements");
  float* partial = static_cast<float*>(d_workspace);
  hipLaunchKernelGGL(bce_partial_kernel, dim3(static_cast<unsigned>(nb)), dim3(256), 0, as_stream(stream), d_prob, d_target,
                     static_cast<long long>(n), partial);
  hipLaunchKernelGGL(bce_final_kernel, dim3(1), dim3(256), 0, as_stream(stream), partial, static_cast<int>(nb),
                     1.0f / static_cast<float>(n), d_loss);
  return check_launch("bce forward kernels");
}

extern "C" int rbx_bce_mean_bwd(const float* d_prob, const float* d_target, const float* d_dloss, int64_t n, float* d_dprob,
                                void* stream) {
  using namespace rbx;
  if (n <= 0) return RBX_OK;
  if (!d_prob || !d_target || !d_dloss || !d_dprob) return fail(RBX_ERR_INVALID, "bce_bwd: NULL tensor");
  long long blocks = (n + 255) / 256;
  if (blocks > kCUs * 8) blocks = kCUs * 8;
  hipLaunchKernelGGL(bce_bwd_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, as_stream(stream), d_prob, d_target,
                     d_dloss, static_cast<long long>(n), 1.0f / static_cast<float>(n), d_dprob);
  return check_launch("bce_bwd_kernel");
}
