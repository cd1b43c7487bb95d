// rbx_norm.hip -- BatchNorm1d of the dense towers (gfx950).
//
// Reference op replaced: the nn.BatchNorm1d that rechub's MLP puts after EVERY Linear
// (third_party/rechub/basic/layers.py:255-263: Linear -> BatchNorm1d -> activation -> Dropout) and the optional
// one of MLP_Layer / MLP_Block (core/pytorch/layers/mlp.py:25-37, ranking/pytorch/layers/blocks/mlp_block.py:42-58).
// ATen's channels-last batch-norm kernels take 0.55 + 0.78 ms for a [65 536, 400] activation on MI355X
// (profiles/r01_models_bench.txt) -- 4.3 ms of a 15.5 ms DeepFM step -- for 105 MB of data.
//
// x is [rows, cols] row-major, statistics per column over the rows.
//   training forward : one sweep computes per-(row block, column) Welford partials (count, mean, M2), a second
//                      tiny kernel merges the blocks in a fixed order (Chan's formula: deterministic, no
//                      catastrophic cancellation as with E[x^2]-E[x]^2), updates the running statistics
//                      (unbiased variance, like torch) and leaves mean / rstd; a third sweep normalises.
//   backward         : dbeta = sum dy, dgamma = sum dy * xhat (two-stage, fixed order), then
//                      dx = gamma * rstd * (dy - dbeta/M - xhat * dgamma/M)   (eval mode: dx = gamma * rstd * dy).
// HBM-bound streaming: forward 2 reads + 1 write of x, backward 4 reads + 1 write; lanes run along the columns
// (coalesced), 4 row lanes per column inside a workgroup.
#include "rbx_internal.h"

namespace rbx {

constexpr int kBnRows = 256;      // rows per workgroup of the reduction sweeps

struct Welford {
  float n, mean, m2;
  __device__ __forceinline__ void add(float x) {
    n += 1.f;
    const float d = x - mean;
    mean += d / n;
    m2 += d * (x - mean);
  }
  __device__ __forceinline__ void merge(const Welford& o) {      // Chan et al.
    if (o.n == 0.f) return;
    const float tot = n + o.n;
    const float d = o.mean - mean;
    mean += d * (o.n / tot);
    m2 += o.m2 + d * d * (n * o.n / tot);
    n = tot;
  }
};

// grid (ceil(cols/64), ceil(rows/kBnRows)); partial[(rb * cols + c) * 3 + {0,1,2}] = (n, mean, M2)
__global__ __launch_bounds__(256) void bn_stats_partial_kernel(const float* __restrict__ x, const long long rows, const int cols,
                                                               float* __restrict__ partial) {
  __shared__ Welford red[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const long long r0 = static_cast<long long>(blockIdx.y) * kBnRows;
  const long long r1 = (r0 + kBnRows < rows) ? r0 + kBnRows : rows;
  Welford w = {0.f, 0.f, 0.f};
  if (c < cols)
    for (long long r = r0 + (threadIdx.x >> 6); r < r1; r += 4) w.add(x[r * cols + c]);
  red[threadIdx.x >> 6][threadIdx.x & 63] = w;
  __syncthreads();
  if (threadIdx.x < 64 && c < cols) {
    Welford t = red[0][threadIdx.x];
    t.merge(red[1][threadIdx.x]);
    t.merge(red[2][threadIdx.x]);
    t.merge(red[3][threadIdx.x]);
    float* dst = partial + (static_cast<long long>(blockIdx.y) * cols + c) * 3;
    dst[0] = t.n; dst[1] = t.mean; dst[2] = t.m2;
  }
}

// one thread per column: merge the row blocks in order, finish mean / rstd, update the running statistics
__global__ __launch_bounds__(256) void bn_stats_final_kernel(const float* __restrict__ partial, const int nblocks, const int cols,
                                                             const float eps, const float momentum,
                                                             float* __restrict__ running_mean,
                                                             float* __restrict__ running_var, float* __restrict__ mean,
                                                             float* __restrict__ rstd) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cols) return;
  Welford t = {0.f, 0.f, 0.f};
  for (int b = 0; b < nblocks; ++b) {
    const float* src = partial + (static_cast<long long>(b) * cols + c) * 3;
    const Welford o = {src[0], src[1], src[2]};
    t.merge(o);
  }
  const float var = t.m2 / t.n;                                  // biased: what normalises the batch
  mean[c] = t.mean;
  rstd[c] = 1.0f / sqrtf(var + eps);
  if (running_mean != nullptr) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * t.mean;
  if (running_var != nullptr) {
    const float unbiased = t.n > 1.f ? t.m2 / (t.n - 1.f) : var;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
  }
}

// eval mode: mean / rstd from the running statistics
__global__ __launch_bounds__(256) void bn_eval_stats_kernel(const float* __restrict__ running_mean,
                                                            const float* __restrict__ running_var, const int cols,
                                                            const float eps, float* __restrict__ mean, float* __restrict__ rstd) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cols) return;
  mean[c] = running_mean[c];
  rstd[c] = 1.0f / sqrtf(running_var[c] + eps);
}

// y = (x - mean) * rstd * gamma + beta, optional ReLU; 4 columns per lane when cols % 4 == 0
template <bool VEC>
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, const long long rows, const int cols,
                                                       const float* __restrict__ mean, const float* __restrict__ rstd,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const int relu, float* __restrict__ y) {
  constexpr int W = VEC ? 4 : 1;
  const long long total = rows * cols / W;
  const long long step = static_cast<long long>(gridDim.x) * blockDim.x;
  for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += step) {
    const long long e = i * W;
    const int c = static_cast<int>(e % cols);
    float v[W], o[W];
    if constexpr (VEC) {
      const float4 t = *reinterpret_cast<const float4*>(x + e);
      v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
      v[0] = x[e];
    }
#pragma unroll
    for (int k = 0; k < W; ++k) {
      const float g = gamma != nullptr ? gamma[c + k] : 1.f;
      const float b = beta != nullptr ? beta[c + k] : 0.f;
      o[k] = (v[k] - mean[c + k]) * rstd[c + k] * g + b;
      if (relu && o[k] < 0.f) o[k] = 0.f;
    }
    if constexpr (VEC) *reinterpret_cast<float4*>(y + e) = make_float4(o[0], o[1], o[2], o[3]);
    else y[e] = o[0];
  }
}

// partial[(rb * cols + c) * 2 + {0,1}] = (sum dy, sum dy * xhat) over the row block
__global__ __launch_bounds__(256) void bn_bwd_partial_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                             const float* __restrict__ y_relu, const long long rows,
                                                             const int cols,
                                                             const float* __restrict__ mean, const float* __restrict__ rstd,
                                                             float* __restrict__ partial) {
  __shared__ float red[2][4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const long long r0 = static_cast<long long>(blockIdx.y) * kBnRows;
  const long long r1 = (r0 + kBnRows < rows) ? r0 + kBnRows : rows;
  float s0 = 0.f, s1 = 0.f;
  if (c < cols) {
    const float m = mean[c], rs = rstd[c];
    for (long long r = r0 + (threadIdx.x >> 6); r < r1; r += 4) {
      float g = dy[r * cols + c];
      if (y_relu != nullptr && !(y_relu[r * cols + c] > 0.f)) g = 0.f;     // fused ReLU: its mask is y > 0
      s0 += g;
      s1 += g * ((x[r * cols + c] - m) * rs);
    }
  }
  red[0][threadIdx.x >> 6][threadIdx.x & 63] = s0;
  red[1][threadIdx.x >> 6][threadIdx.x & 63] = s1;
  __syncthreads();
  if (threadIdx.x < 64 && c < cols) {
    float* dst = partial + (static_cast<long long>(blockIdx.y) * cols + c) * 2;
    dst[0] = (red[0][0][threadIdx.x] + red[0][1][threadIdx.x]) + (red[0][2][threadIdx.x] + red[0][3][threadIdx.x]);
    dst[1] = (red[1][0][threadIdx.x] + red[1][1][threadIdx.x]) + (red[1][2][threadIdx.x] + red[1][3][threadIdx.x]);
  }
}

__global__ __launch_bounds__(256) void bn_bwd_final_kernel(const float* __restrict__ partial, const int nblocks, const int cols,
                                                           float* __restrict__ dbeta, float* __restrict__ dgamma) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cols) return;
  float s0 = 0.f, s1 = 0.f;
  for (int b = 0; b < nblocks; ++b) {
    const float* src = partial + (static_cast<long long>(b) * cols + c) * 2;
    s0 += src[0];
    s1 += src[1];
  }
  dbeta[c] = s0;
  dgamma[c] = s1;
}

template <bool VEC>
__global__ __launch_bounds__(256) void bn_bwd_dx_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                        const float* __restrict__ y_relu, const long long rows,
                                                        const int cols,
                                                        const float* __restrict__ mean, const float* __restrict__ rstd,
                                                        const float* __restrict__ gamma, const float* __restrict__ dbeta,
                                                        const float* __restrict__ dgamma, const int training,
                                                        float* __restrict__ dx) {
  constexpr int W = VEC ? 4 : 1;
  const long long total = rows * cols / W;
  const long long step = static_cast<long long>(gridDim.x) * blockDim.x;
  const float inv_m = 1.0f / static_cast<float>(rows);
  for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += step) {
    const long long e = i * W;
    const int c = static_cast<int>(e % cols);
    float xv[W], gv[W], o[W];
    if constexpr (VEC) {
      const float4 a = *reinterpret_cast<const float4*>(x + e);
      const float4 b = *reinterpret_cast<const float4*>(dy + e);
      xv[0] = a.x; xv[1] = a.y; xv[2] = a.z; xv[3] = a.w;
      gv[0] = b.x; gv[1] = b.y; gv[2] = b.z; gv[3] = b.w;
    } else {
      xv[0] = x[e];
      gv[0] = dy[e];
    }
    if (y_relu != nullptr) {
#pragma unroll
      for (int k = 0; k < W; ++k)
        if (!(y_relu[e + k] > 0.f)) gv[k] = 0.f;
    }
#pragma unroll
    for (int k = 0; k < W; ++k) {
      const float g = gamma != nullptr ? gamma[c + k] : 1.f;
      const float rs = rstd[c + k];
      if (training) {
        const float xhat = (xv[k] - mean[c + k]) * rs;
        o[k] = g * rs * (gv[k] - dbeta[c + k] * inv_m - xhat * dgamma[c + k] * inv_m);
      } else {
        o[k] = g * rs * gv[k];
      }
    }
    if constexpr (VEC) *reinterpret_cast<float4*>(dx + e) = make_float4(o[0], o[1], o[2], o[3]);
    else dx[e] = o[0];
  }
}

static int bn_blocks(int64_t rows) { return static_cast<int>((rows + kBnRows - 1) / kBnRows); }
static bool bn_vec(int cols, const void* a, const void* b, const void* c) {
  return cols % 4 == 0 && ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c)) & 15) == 0;
}

}  // namespace rbx

extern "C" size_t rbx_batchnorm_workspace_size(int64_t rows, int32_t cols) {
  if (rows <= 0 || cols <= 0) return 0;
  return static_cast<size_t>(rbx::bn_blocks(rows)) * cols * 3 * sizeof(float) + 256;
}

extern "C" int rbx_batchnorm_fwd(const float* d_x, int64_t rows, int32_t cols, const float* d_gamma, const float* d_beta,
                                 float eps, int32_t training, float momentum, float* d_running_mean, float* d_running_var,
                                 int32_t relu, float* d_mean, float* d_rstd, float* d_y, void* d_workspace,
                                 size_t workspace_bytes, void* stream) {
  using namespace rbx;
  if (rows < 0 || cols <= 0) return fail(RBX_ERR_INVALID, "batchnorm: bad shape");
  if (rows == 0) return RBX_OK;
  if (!d_x || !d_y || !d_mean || !d_rstd) return fail(RBX_ERR_INVALID, "batchnorm: NULL tensor");
  if (training && rows < 2) return fail(RBX_ERR_INVALID, "Expected more than 1 value per channel when training");
  hipStream_t s = as_stream(stream);
  const unsigned cb = (cols + 255) / 256;
  if (training) {
    if (d_workspace == nullptr || workspace_bytes < rbx_batchnorm_workspace_size(rows, cols))
      return fail(RBX_ERR_WORKSPACE, "batchnorm: workspace too small");
    float* partial = static_cast<float*>(d_workspace);
    const int nb = bn_blocks(rows);
    hipLaunchKernelGGL(bn_stats_partial_kernel, dim3((cols + 63) / 64, nb), dim3(256), 0, s, d_x, static_cast<long long>(rows),
                       cols, partial);
    hipLaunchKernelGGL(bn_stats_final_kernel, dim3(cb), dim3(256), 0, s, partial, nb, cols, eps, momentum, d_running_mean,
                       d_running_var, d_mean, d_rstd);
  } else {
    if (!d_running_mean || !d_running_var) return fail(RBX_ERR_INVALID, "batchnorm: eval mode needs the running statistics");
    hipLaunchKernelGGL(bn_eval_stats_kernel, dim3(cb), dim3(256), 0, s, d_running_mean, d_running_var, cols, eps, d_mean,
                       d_rstd);
  }
  const bool vec = bn_vec(cols, d_x, d_y, d_y);
  const long long total = static_cast<long long>(rows) * cols / (vec ? 4 : 1);
  long long blocks = (total + 255) / 256;
  if (blocks > kCUs * 16) blocks = kCUs * 16;
  if (vec)
    hipLaunchKernelGGL(bn_apply_kernel<true>, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s, d_x,
                       static_cast<long long>(rows), cols, d_mean, d_rstd, d_gamma, d_beta, relu, d_y);
  else
    hipLaunchKernelGGL(bn_apply_kernel<false>, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s, d_x,
                       static_cast<long long>(rows), cols, d_mean, d_rstd, d_gamma, d_beta, relu, d_y);
  return check_launch("batchnorm forward kernels");
}

extern "C" int rbx_batchnorm_bwd(const float* d_x, const float* d_dy, const float* d_y_relu, int64_t rows, int32_t cols,
                                 const float* d_gamma, const float* d_mean, const float* d_rstd, int32_t training,
                                 float* d_dx, float* d_dgamma,
                                 float* d_dbeta, void* d_workspace, size_t workspace_bytes, void* stream) {
  using namespace rbx;
  if (rows < 0 || cols <= 0) return fail(RBX_ERR_INVALID, "batchnorm_bwd: bad shape");
  if (rows == 0) return RBX_OK;
  if (!d_x || !d_dy || !d_mean || !d_rstd || !d_dgamma || !d_dbeta)
    return fail(RBX_ERR_INVALID, "batchnorm_bwd: NULL tensor (d_dgamma / d_dbeta are scratch even when unused)");
  if (d_workspace == nullptr || workspace_bytes < rbx_batchnorm_workspace_size(rows, cols))
    return fail(RBX_ERR_WORKSPACE, "batchnorm_bwd: workspace too small");
  hipStream_t s = as_stream(stream);
  float* partial = static_cast<float*>(d_workspace);
  const int nb = bn_blocks(rows);
  hipLaunchKernelGGL(bn_bwd_partial_kernel, dim3((cols + 63) / 64, nb), dim3(256), 0, s, d_x, d_dy, d_y_relu,
                     static_cast<long long>(rows), cols, d_mean, d_rstd, partial);
  hipLaunchKernelGGL(bn_bwd_final_kernel, dim3((cols + 255) / 256), dim3(256), 0, s, partial, nb, cols, d_dbeta, d_dgamma);
  if (d_dx != nullptr) {
    const bool vec = bn_vec(cols, d_x, d_dy, d_dx);
    const long long total = static_cast<long long>(rows) * cols / (vec ? 4 : 1);
    long long blocks = (total + 255) / 256;
    if (blocks > kCUs * 16) blocks = kCUs * 16;
    if (vec)
      hipLaunchKernelGGL(bn_bwd_dx_kernel<true>, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s, d_x, d_dy, d_y_relu,
                         static_cast<long long>(rows), cols, d_mean, d_rstd, d_gamma, d_dbeta, d_dgamma, training, d_dx);
    else
      hipLaunchKernelGGL(bn_bwd_dx_kernel<false>, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s, d_x, d_dy, d_y_relu,
                         static_cast<long long>(rows), cols, d_mean, d_rstd, d_gamma, d_dbeta, d_dgamma, training, d_dx);
  }
  return check_launch("batchnorm backward kernels");
}
