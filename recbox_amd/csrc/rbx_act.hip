// rbx_act.hip -- the activations of the dense towers that are not fused into a GEMM epilogue or a BatchNorm pass (gfx950):
//   nn.PReLU standing alone       core/pytorch/layers/mlp.py:25-37, ranking/pytorch/layers/blocks/mlp_block.py:42-58
//                                 (hidden_activations = "PReLU" without batch_norm), third_party/rechub/basic/layers.py:255-263
//   nn.Dropout(p > 0), training   the same three towers (dropout_rates / dropout)
//   Dice                          core/pytorch/layers/activations.py:23-33: p = sigmoid(BatchNorm1d(x, affine=False,
//                                 eps=1e-9, momentum=0.01)), y = p x + alpha (1 - p) x
// (paths relative to /root/reference/recbox).  All are HBM streams over [rows, cols] activations; the column reductions
// (Dice's statistics and its backward sums, the slope gradients) are two-stage in a fixed order -- partials per block of 256
// rows, merged in block order -- so results repeat bit for bit.
#include "rbx_internal.h"

namespace rbx {

constexpr int kActRows = 256;       // rows per block of the column reductions
constexpr int kActCols = 64;        // columns per workgroup: 64 column lanes x 4 row lanes

static int act_blocks(long long rows) { return static_cast<int>((rows + kActRows - 1) / kActRows); }

// ---- PReLU ----------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void prelu_fwd_kernel(const float* __restrict__ x, const long long n, const int cols,
                                                        const float* __restrict__ slope, const int n_slope,
                                                        float* __restrict__ y) {
  const long long stride = static_cast<long long>(gridDim.x) * blockDim.x;
  for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float v = x[i];
    const float a = slope[n_slope == 1 ? 0 : static_cast<int>(i % cols)];
    y[i] = v > 0.f ? v : a * v;
  }
}

// dx = dy (x > 0 ? 1 : a); partial[block][col] = sum over the block's rows of dy x [x <= 0]
__global__ __launch_bounds__(256) void prelu_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                        const long long rows, const int cols,
                                                        const float* __restrict__ slope, const int n_slope,
                                                        float* __restrict__ dx, float* __restrict__ partial) {
  __shared__ float red[4][kActCols];
  const int cl = threadIdx.x % kActCols, rl = threadIdx.x / kActCols;
  const int col = blockIdx.x * kActCols + cl;
  const long long r0 = static_cast<long long>(blockIdx.y) * kActRows;
  float acc = 0.f;
  if (col < cols) {
    const float a = slope[n_slope == 1 ? 0 : col];
    for (int r = rl; r < kActRows; r += 4) {
      const long long row = r0 + r;
      if (row >= rows) break;
      const float v = x[row * cols + col], g = dy[row * cols + col];
      if (dx != nullptr) dx[row * cols + col] = v > 0.f ? g : a * g;
      if (!(v > 0.f)) acc += g * v;
    }
  }
  red[rl][cl] = acc;
  __syncthreads();
  if (rl == 0 && col < cols && partial != nullptr)
    partial[static_cast<long long>(blockIdx.y) * cols + col] = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
}

// out[c] = sum over blocks (ascending) of partial[block][c]; n_out == 1: one workgroup adds all columns too (fixed order)
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ partial, const int n_blocks, const int cols,
                                                           const int n_out, float* __restrict__ out) {
  if (n_out != 1) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= cols) return;
    float s = 0.f;
    for (int b = 0; b < n_blocks; ++b) s += partial[static_cast<long long>(b) * cols + c];
    out[c] = s;
    return;
  }
  __shared__ float red[256];
  float s = 0.f;
  for (int c = threadIdx.x; c < cols; c += 256)
    for (int b = 0; b < n_blocks; ++b) s += partial[static_cast<long long>(b) * cols + c];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = red[0];
}

// ---- Dropout --------------------------------------------------------------------------------------------------------------
// keep(i) is a counter-based function of (seed, i): Philox4x32-10 on counter (i / 4), word i % 4, keep iff word >= thr
// (= round(p 2^32)); the backward applies the same call to dy -- the mask is never stored.
__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, const long long n, const unsigned thr,
                                                      const float scale, const unsigned long long seed,
                                                      const unsigned long long* __restrict__ seed_add, float* __restrict__ y) {
  const unsigned long long key = seed + (seed_add != nullptr ? seed_add[0] : 0ull);
  const unsigned k0 = static_cast<unsigned>(key), k1 = static_cast<unsigned>(key >> 32);
  const long long n4 = (n + 3) / 4;
  const long long stride = static_cast<long long>(gridDim.x) * blockDim.x;
  const bool vec = (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0;
  for (long long q = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; q < n4; q += stride) {
    unsigned c[4] = {static_cast<unsigned>(q), static_cast<unsigned>(static_cast<unsigned long long>(q) >> 32), 0u, 0u};
    Philox::run(c, k0, k1);
    const long long i = q * 4;
    if (vec && i + 3 < n) {
      const float4 v = *reinterpret_cast<const float4*>(x + i);
      *reinterpret_cast<float4*>(y + i) = make_float4(c[0] >= thr ? v.x * scale : 0.f, c[1] >= thr ? v.y * scale : 0.f,
                                                      c[2] >= thr ? v.z * scale : 0.f, c[3] >= thr ? v.w * scale : 0.f);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (i + j < n) y[i + j] = c[j] >= thr ? x[i + j] * scale : 0.f;
    }
  }
}

// ---- Dice -----------------------------------------------------------------------------------------------------------------
// statistics: per (block of 256 rows, column) the triple (count, mean, M2), merged with Chan's formula in block order
__global__ __launch_bounds__(256) void dice_stats_kernel(const float* __restrict__ x, const long long rows, const int cols,
                                                         float* __restrict__ partial) {
  __shared__ float s_n[4][kActCols], s_mean[4][kActCols], s_m2[4][kActCols];
  const int cl = threadIdx.x % kActCols, rl = threadIdx.x / kActCols;
  const int col = blockIdx.x * kActCols + cl;
  const long long r0 = static_cast<long long>(blockIdx.y) * kActRows;
  float n = 0.f, mean = 0.f, m2 = 0.f;
  if (col < cols) {
    for (int r = rl; r < kActRows; r += 4) {                 // Welford over this lane's rows
      const long long row = r0 + r;
      if (row >= rows) break;
      const float v = x[row * cols + col];
      n += 1.f;
      const float d = v - mean;
      mean += d / n;
      m2 += d * (v - mean);
    }
  }
  s_n[rl][cl] = n; s_mean[rl][cl] = mean; s_m2[rl][cl] = m2;
  __syncthreads();
  if (rl == 0 && col < cols) {
    float tn = s_n[0][cl], tm = s_mean[0][cl], t2 = s_m2[0][cl];
#pragma unroll
    for (int k = 1; k < 4; ++k) {
      const float on = s_n[k][cl];
      if (on > 0.f) {
        const float tot = tn + on, d = s_mean[k][cl] - tm;
        tm += d * (on / tot);
        t2 += s_m2[k][cl] + d * d * (tn * on / tot);
        tn = tot;
      }
    }
    float* dst = partial + (static_cast<long long>(blockIdx.y) * cols + col) * 3;
    dst[0] = tn; dst[1] = tm; dst[2] = t2;
  }
}

__global__ __launch_bounds__(256) void dice_stats_final_kernel(const float* __restrict__ partial, const int n_blocks,
                                                               const int cols, const float eps, const float momentum,
                                                               float* __restrict__ running_mean,
                                                               float* __restrict__ running_var, float* __restrict__ mean_out,
                                                               float* __restrict__ rstd_out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cols) return;
  float tn = 0.f, tm = 0.f, t2 = 0.f;
  for (int b = 0; b < n_blocks; ++b) {
    const float* p = partial + (static_cast<long long>(b) * cols + c) * 3;
    const float on = p[0];
    if (on > 0.f) {
      const float tot = tn + on, d = p[1] - tm;
      tm += d * (on / tot);
      t2 += p[2] + d * d * (tn * on / tot);
      tn = tot;
    }
  }
  const float var = tn > 0.f ? t2 / tn : 0.f;                 // the biased variance normalises
  mean_out[c] = tm;
  rstd_out[c] = rsqrtf(var + eps);
  if (running_mean != nullptr) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * tm;
  if (running_var != nullptr)
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (tn > 1.f ? t2 / (tn - 1.f) : var);
}

__global__ __launch_bounds__(256) void dice_eval_stats_kernel(const float* __restrict__ running_mean,
                                                              const float* __restrict__ running_var, const int cols,
                                                              const float eps, float* __restrict__ mean_out,
                                                              float* __restrict__ rstd_out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cols) return;
  mean_out[c] = running_mean[c];
  rstd_out[c] = rsqrtf(running_var[c] + eps);
}

__device__ __forceinline__ float sigmoidf(float t) { return 1.f / (1.f + expf(-t)); }

__global__ __launch_bounds__(256) void dice_apply_kernel(const float* __restrict__ x, const long long n, const int cols,
                                                         const float* __restrict__ alpha, const float* __restrict__ mean,
                                                         const float* __restrict__ rstd, float* __restrict__ y) {
  const long long stride = static_cast<long long>(gridDim.x) * blockDim.x;
  for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int c = static_cast<int>(i % cols);
    const float v = x[i];
    const float p = sigmoidf((v - mean[c]) * rstd[c]);
    y[i] = p * v + alpha[c] * (1.f - p) * v;
  }
}

// backward sums per (block, column): s1 = sum dxhat, s2 = sum dxhat xhat, sa = sum dy (1 - p) x,
// dxhat = dy x (1 - alpha) p (1 - p)
__global__ __launch_bounds__(256) void dice_bwd_partial_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                               const long long rows, const int cols,
                                                               const float* __restrict__ alpha, const float* __restrict__ mean,
                                                               const float* __restrict__ rstd, float* __restrict__ partial) {
  __shared__ float red[3][4][kActCols];
  const int cl = threadIdx.x % kActCols, rl = threadIdx.x / kActCols;
  const int col = blockIdx.x * kActCols + cl;
  const long long r0 = static_cast<long long>(blockIdx.y) * kActRows;
  float s1 = 0.f, s2 = 0.f, sa = 0.f;
  if (col < cols) {
    const float a = alpha[col], mu = mean[col], rs = rstd[col];
    for (int r = rl; r < kActRows; r += 4) {
      const long long row = r0 + r;
      if (row >= rows) break;
      const float v = x[row * cols + col], g = dy[row * cols + col];
      const float xh = (v - mu) * rs;
      const float p = sigmoidf(xh);
      const float dxh = g * v * (1.f - a) * p * (1.f - p);
      s1 += dxh;
      s2 += dxh * xh;
      sa += g * (1.f - p) * v;
    }
  }
  red[0][rl][cl] = s1; red[1][rl][cl] = s2; red[2][rl][cl] = sa;
  __syncthreads();
  if (rl == 0 && col < cols) {
    float* dst = partial + (static_cast<long long>(blockIdx.y) * cols + col) * 3;
#pragma unroll
    for (int k = 0; k < 3; ++k) dst[k] = (red[k][0][cl] + red[k][1][cl]) + (red[k][2][cl] + red[k][3][cl]);
  }
}

__global__ __launch_bounds__(256) void dice_bwd_final_kernel(const float* __restrict__ partial, const int n_blocks,
                                                             const int cols, float* __restrict__ sums /* [2, cols] */,
                                                             float* __restrict__ dalpha) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cols) return;
  float s1 = 0.f, s2 = 0.f, sa = 0.f;
  for (int b = 0; b < n_blocks; ++b) {
    const float* p = partial + (static_cast<long long>(b) * cols + c) * 3;
    s1 += p[0]; s2 += p[1]; sa += p[2];
  }
  sums[c] = s1;
  sums[cols + c] = s2;
  if (dalpha != nullptr) dalpha[c] = sa;
}

// dx = dy (p + alpha (1 - p)) + rstd (dxhat - s1 / n - xhat s2 / n)      (training; evaluation: + rstd dxhat)
__global__ __launch_bounds__(256) void dice_bwd_dx_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                          const long long n, const int cols, const float inv_rows,
                                                          const float* __restrict__ alpha, const float* __restrict__ mean,
                                                          const float* __restrict__ rstd, const float* __restrict__ sums,
                                                          const int training, float* __restrict__ dx) {
  const long long stride = static_cast<long long>(gridDim.x) * blockDim.x;
  for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int c = static_cast<int>(i % cols);
    const float v = x[i], g = dy[i], a = alpha[c], rs = rstd[c];
    const float xh = (v - mean[c]) * rs;
    const float p = sigmoidf(xh);
    const float dxh = g * v * (1.f - a) * p * (1.f - p);
    float t = dxh;
    if (training) t -= sums[c] * inv_rows + xh * sums[cols + c] * inv_rows;
    dx[i] = g * (p + a * (1.f - p)) + rs * t;
  }
}

static unsigned stream_blocks(long long n) {
  long long b = (n + 255) / 256;
  if (b > kCUs * 16) b = kCUs * 16;
  return static_cast<unsigned>(b < 1 ? 1 : b);
}

static int act_check(const float* x, long long rows, int cols, const char* what) {
  if (rows < 0 || cols <= 0) return fail(RBX_ERR_INVALID, "%s: bad shape [%lld, %d]", what, rows, cols);
  if (rows > 0 && x == nullptr) return fail(RBX_ERR_INVALID, "%s: d_x is NULL", what);
  return RBX_OK;
}

}  // namespace rbx

extern "C" size_t rbx_act_workspace_size(int64_t rows, int32_t cols) {
  if (rows <= 0 || cols <= 0) return 0;
  return (static_cast<size_t>(rbx::act_blocks(rows)) * cols * 3 + 2 * static_cast<size_t>(cols)) * sizeof(float);
}

extern "C" int rbx_prelu_fwd(const float* d_x, int64_t rows, int32_t cols, const float* d_slope, int32_t n_slope, float* d_y,
                             void* stream) {
  using namespace rbx;
  int rc = act_check(d_x, rows, cols, "prelu_fwd");
  if (rc != RBX_OK) return rc;
  if (n_slope != 1 && n_slope != cols) return fail(RBX_ERR_INVALID, "prelu: %d slopes for %d columns", n_slope, cols);
  if (d_slope == nullptr || d_y == nullptr) return fail(RBX_ERR_INVALID, "prelu_fwd: NULL argument");
  if (rows == 0) return RBX_OK;
  const long long n = rows * cols;
  hipLaunchKernelGGL(prelu_fwd_kernel, dim3(stream_blocks(n)), dim3(256), 0, as_stream(stream), d_x, n, cols, d_slope, n_slope,
                     d_y);
  return check_launch("prelu_fwd_kernel");
}

extern "C" int rbx_prelu_bwd(const float* d_x, const float* d_dy, int64_t rows, int32_t cols, const float* d_slope,
                             int32_t n_slope, float* d_dx, float* d_dslope, void* d_workspace, size_t workspace_bytes,
                             void* stream) {
  using namespace rbx;
  int rc = act_check(d_x, rows, cols, "prelu_bwd");
  if (rc != RBX_OK) return rc;
  if (n_slope != 1 && n_slope != cols) return fail(RBX_ERR_INVALID, "prelu: %d slopes for %d columns", n_slope, cols);
  if (d_slope == nullptr || d_dy == nullptr) return fail(RBX_ERR_INVALID, "prelu_bwd: NULL argument");
  hipStream_t s = as_stream(stream);
  if (rows == 0) {
    if (d_dslope != nullptr) return hipMemsetAsync(d_dslope, 0, sizeof(float) * n_slope, s) == hipSuccess ? RBX_OK : fail(RBX_ERR_LAUNCH, "memset");
    return RBX_OK;
  }
  if (d_dslope != nullptr && (d_workspace == nullptr || workspace_bytes < rbx_act_workspace_size(rows, cols)))
    return fail(RBX_ERR_WORKSPACE, "prelu_bwd: workspace too small");
  float* partial = d_dslope != nullptr ? static_cast<float*>(d_workspace) : nullptr;
  const int nb = act_blocks(rows);
  hipLaunchKernelGGL(prelu_bwd_kernel, dim3((cols + kActCols - 1) / kActCols, nb), dim3(256), 0, s, d_x, d_dy,
                     static_cast<long long>(rows), cols, d_slope, n_slope, d_dx, partial);
  rc = check_launch("prelu_bwd_kernel");
  if (rc != RBX_OK || d_dslope == nullptr) return rc;
  hipLaunchKernelGGL(colsum_final_kernel, dim3(n_slope == 1 ? 1 : (cols + 255) / 256), dim3(256), 0, s, partial, nb, cols,
                     n_slope, d_dslope);
  return check_launch("colsum_final_kernel");
}

extern "C" int rbx_dropout(const float* d_x, int64_t n, float p, uint64_t seed, const uint64_t* d_seed_add, float* d_y,
                           void* stream) {
  using namespace rbx;
  if (n < 0 || !(p >= 0.f) || !(p < 1.f)) return fail(RBX_ERR_INVALID, "dropout: n=%lld p=%g", (long long)n, p);
  if (n == 0) return RBX_OK;
  if (d_x == nullptr || d_y == nullptr) return fail(RBX_ERR_INVALID, "dropout: NULL argument");
  const double t = static_cast<double>(p) * 4294967296.0;
  const unsigned thr = t >= 4294967295.0 ? 4294967295u : static_cast<unsigned>(t + 0.5);
  // the kept values are scaled by the EXACT keep rate of the threshold: E[y] = x
  const float scale = static_cast<float>(4294967296.0 / (4294967296.0 - static_cast<double>(thr)));
  hipLaunchKernelGGL(dropout_kernel, dim3(stream_blocks((n + 3) / 4)), dim3(256), 0, as_stream(stream), d_x,
                     static_cast<long long>(n), thr, scale, static_cast<unsigned long long>(seed),
                     reinterpret_cast<const unsigned long long*>(d_seed_add), d_y);
  return check_launch("dropout_kernel");
}

extern "C" int rbx_dice_fwd(const float* d_x, int64_t rows, int32_t cols, const float* d_alpha, float eps, int32_t training,
                            float momentum, float* d_running_mean, float* d_running_var, float* d_mean, float* d_rstd,
                            float* d_y, void* d_workspace, size_t workspace_bytes, void* stream) {
  using namespace rbx;
  int rc = act_check(d_x, rows, cols, "dice_fwd");
  if (rc != RBX_OK) return rc;
  if (d_alpha == nullptr || d_mean == nullptr || d_rstd == nullptr || d_y == nullptr)
    return fail(RBX_ERR_INVALID, "dice_fwd: NULL argument");
  if (!training && (d_running_mean == nullptr || d_running_var == nullptr))
    return fail(RBX_ERR_INVALID, "dice_fwd: evaluation needs the running statistics");
  if (rows == 0) return RBX_OK;
  hipStream_t s = as_stream(stream);
  if (training) {
    if (d_workspace == nullptr || workspace_bytes < rbx_act_workspace_size(rows, cols))
      return fail(RBX_ERR_WORKSPACE, "dice_fwd: workspace too small");
    float* partial = static_cast<float*>(d_workspace);
    const int nb = act_blocks(rows);
    hipLaunchKernelGGL(dice_stats_kernel, dim3((cols + kActCols - 1) / kActCols, nb), dim3(256), 0, s, d_x,
                       static_cast<long long>(rows), cols, partial);
    hipLaunchKernelGGL(dice_stats_final_kernel, dim3((cols + 255) / 256), dim3(256), 0, s, partial, nb, cols, eps, momentum,
                       d_running_mean, d_running_var, d_mean, d_rstd);
  } else {
    hipLaunchKernelGGL(dice_eval_stats_kernel, dim3((cols + 255) / 256), dim3(256), 0, s, d_running_mean, d_running_var, cols,
                       eps, d_mean, d_rstd);
  }
  const long long n = rows * cols;
  hipLaunchKernelGGL(dice_apply_kernel, dim3(stream_blocks(n)), dim3(256), 0, s, d_x, n, cols, d_alpha, d_mean, d_rstd, d_y);
  return check_launch("dice_apply_kernel");
}

extern "C" int rbx_dice_bwd(const float* d_x, const float* d_dy, int64_t rows, int32_t cols, const float* d_alpha,
                            const float* d_mean, const float* d_rstd, int32_t training, float* d_dx, float* d_dalpha,
                            void* d_workspace, size_t workspace_bytes, void* stream) {
  using namespace rbx;
  int rc = act_check(d_x, rows, cols, "dice_bwd");
  if (rc != RBX_OK) return rc;
  if (d_dy == nullptr || d_alpha == nullptr || d_mean == nullptr || d_rstd == nullptr)
    return fail(RBX_ERR_INVALID, "dice_bwd: NULL argument");
  hipStream_t s = as_stream(stream);
  if (rows == 0) {
    if (d_dalpha != nullptr) return hipMemsetAsync(d_dalpha, 0, sizeof(float) * cols, s) == hipSuccess ? RBX_OK : fail(RBX_ERR_LAUNCH, "memset");
    return RBX_OK;
  }
  if (d_workspace == nullptr || workspace_bytes < rbx_act_workspace_size(rows, cols))
    return fail(RBX_ERR_WORKSPACE, "dice_bwd: workspace too small");
  float* partial = static_cast<float*>(d_workspace);
  const int nb = act_blocks(rows);
  float* sums = partial + static_cast<size_t>(nb) * cols * 3;
  hipLaunchKernelGGL(dice_bwd_partial_kernel, dim3((cols + kActCols - 1) / kActCols, nb), dim3(256), 0, s, d_x, d_dy,
                     static_cast<long long>(rows), cols, d_alpha, d_mean, d_rstd, partial);
  hipLaunchKernelGGL(dice_bwd_final_kernel, dim3((cols + 255) / 256), dim3(256), 0, s, partial, nb, cols, sums, d_dalpha);
  if (d_dx != nullptr) {
    const long long n = rows * cols;
    hipLaunchKernelGGL(dice_bwd_dx_kernel, dim3(stream_blocks(n)), dim3(256), 0, s, d_x, d_dy, n, cols,
                       1.f / static_cast<float>(rows), d_alpha, d_mean, d_rstd, sums, training, d_dx);
  }
  return check_launch("dice_bwd kernels");
}
