// rbx_loss.hip -- the loss epilogues of K7 (sampled-softmax / pos-neg logits -> scalar loss), gfx950.
//
// Reference behaviour replaced:
//   core/pytorch/losses/softmax_crossentropy_loss.py:14-22   -log softmax(y_pred)[:, 0], mean over the batch
//   third_party/rechub/trainers/match_trainer.py:59-60       CrossEntropyLoss on [B, 1 + n_neg] logits, label 0 (YoutubeDNN)
//   third_party/rechub/models/matching/sasrec.py:100-107     pos / neg logits of every position, trained with
//                                                            -log sigmoid(pos) - log(1 - sigmoid(neg)) over real positions
// ATen runs each as 4-8 element-wise / reduction kernels over the logit block (log_softmax, nll_loss, neg, mul, sum and
// their backward); here: ONE forward pass with block partials + the fixed-order final sum (bce_final's scheme: no
// atomics, deterministic), ONE backward pass.  HBM-stream bound, tiny next to the gathers that produce the logits.
#include "rbx_internal.h"

namespace rbx {

constexpr int kLossRows = 256;     // rows per workgroup (a thread owns a row)

// one workgroup: fixed-order sum of the block partials, times `scale`
__global__ __launch_bounds__(256) void loss_final_kernel(const float* __restrict__ partial, const int nblocks, const float scale,
                                                         float* __restrict__ loss) {
  __shared__ float red[4];
  float acc = 0.f;
  for (int i = threadIdx.x; i < nblocks; i += 256) acc += partial[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) loss[0] = ((red[0] + red[1]) + (red[2] + red[3])) * scale;
}

__device__ __forceinline__ float block_partial(float acc, float* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// loss_r = logsumexp(x[r, :]) - x[r, t_r]
__global__ __launch_bounds__(256) void softmax_ce_fwd_kernel(const float* __restrict__ x, const long long stride,
                                                             const long long rows, const int n,
                                                             const long long* __restrict__ target, float* __restrict__ lse,
                                                             float* __restrict__ partial, int* __restrict__ status) {
  __shared__ float red[4];
  const long long r = static_cast<long long>(blockIdx.x) * kLossRows + threadIdx.x;
  float acc = 0.f;
  if (r < rows) {
    const float* row = x + r * stride;
    float m = -INFINITY;
    for (int c = 0; c < n; ++c) m = fmaxf(m, row[c]);
    float s = 0.f;
    for (int c = 0; c < n; ++c) s += expf(row[c] - m);
    const float l = m + logf(s);
    long long t = (target != nullptr) ? target[r] : 0;
    if (t < 0 || t >= n) {
      if (status != nullptr) atomicOr(status, 1);
      t = 0;
    }
    lse[r] = l;
    acc = l - row[t];
  }
  const float tot = block_partial(acc, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

// dx[r, c] = g / rows * (softmax(x[r, :])[c] - [c == t_r])
__global__ __launch_bounds__(256) void softmax_ce_bwd_kernel(const float* __restrict__ x, const long long stride,
                                                             const long long rows, const int n,
                                                             const long long* __restrict__ target,
                                                             const float* __restrict__ lse, const float* __restrict__ gloss,
                                                             const float inv_rows, float* __restrict__ dx) {
  const float g = gloss[0] * inv_rows;
  const long long total = rows * n;
  const long long step = static_cast<long long>(gridDim.x) * blockDim.x;
  for (long long e = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; e < total; e += step) {
    const long long r = e / n;
    const int c = static_cast<int>(e - r * n);
    long long t = (target != nullptr) ? target[r] : 0;
    if (t < 0 || t >= n) t = 0;
    const float p = expf(x[r * stride + c] - lse[r]);
    dx[e] = g * (p - (c == t ? 1.f : 0.f));
  }
}

// torch's log_sigmoid: min(x, 0) - log1p(exp(-|x|))
__device__ __forceinline__ float log_sigmoid(float v) { return fminf(v, 0.f) - log1pf(expf(-fabsf(v))); }

// term_i = -w_i (log sigmoid(pos_i) + log sigmoid(-neg_i))
__global__ __launch_bounds__(256) void pair_logsig_fwd_kernel(const float* __restrict__ pos, const float* __restrict__ neg,
                                                              const float* __restrict__ w, const long long n,
                                                              float* __restrict__ partial) {
  __shared__ float red[4];
  const long long base = static_cast<long long>(blockIdx.x) * 1024;
  float acc = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const long long i = base + k * 256 + threadIdx.x;
    if (i < n) {
      const float wi = (w != nullptr) ? w[i] : 1.f;
      if (wi != 0.f) acc -= wi * (log_sigmoid(pos[i]) + log_sigmoid(-neg[i]));      // masked positions cost nothing
    }
  }
  const float tot = block_partial(acc, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

__global__ __launch_bounds__(256) void pair_logsig_bwd_kernel(const float* __restrict__ pos, const float* __restrict__ neg,
                                                              const float* __restrict__ w, const float* __restrict__ gloss,
                                                              const float scale, const long long n, float* __restrict__ dpos,
                                                              float* __restrict__ dneg) {
  const float g = gloss[0] * scale;
  const long long step = static_cast<long long>(gridDim.x) * blockDim.x;
  for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += step) {
    const float wi = g * ((w != nullptr) ? w[i] : 1.f);
    const float sp = 1.f / (1.f + expf(-pos[i])), sn = 1.f / (1.f + expf(-neg[i]));
    dpos[i] = -wi * (1.f - sp);                       // d/dx -log sigmoid(x) = -(1 - sigmoid(x))
    dneg[i] = wi * sn;                                // d/dx -log sigmoid(-x) = sigmoid(x)
  }
}

}  // namespace rbx

extern "C" size_t rbx_loss_workspace_size(int64_t n) {
  return n > 0 ? static_cast<size_t>((n + rbx::kLossRows - 1) / rbx::kLossRows) * sizeof(float) + 256 : 256;
}

extern "C" int rbx_softmax_ce_fwd(const float* d_logits, int64_t stride, int64_t rows, int32_t n_classes,
                                  const int64_t* d_target, float* d_loss, float* d_lse, int32_t* d_status,
                                  void* d_workspace, size_t workspace_bytes, void* stream) {
  using namespace rbx;
  if (rows <= 0 || n_classes <= 0) return fail(RBX_ERR_INVALID, "softmax_ce: empty input (the mean of no rows is undefined)");
  if (!d_logits || !d_loss || !d_lse) return fail(RBX_ERR_INVALID, "softmax_ce: NULL tensor");
  if (d_workspace == nullptr || workspace_bytes < rbx_loss_workspace_size(rows)) return fail(RBX_ERR_WORKSPACE, "softmax_ce: workspace too small");
  const long long nb = (rows + kLossRows - 1) / kLossRows;
  if (nb >= INT_MAX) return fail(RBX_ERR_UNSUPPORTED, "softmax_ce: too many rows");
  float* partial = static_cast<float*>(d_workspace);
  hipLaunchKernelGGL(softmax_ce_fwd_kernel, dim3(static_cast<unsigned>(nb)), dim3(256), 0, as_stream(stream), d_logits,
                     static_cast<long long>(stride), static_cast<long long>(rows), n_classes,
                     reinterpret_cast<const long long*>(d_target), d_lse, partial, d_status);
  hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(256), 0, as_stream(stream), partial, static_cast<int>(nb),
                     1.0f / static_cast<float>(rows), d_loss);
  return check_launch("softmax_ce forward kernels");
}

extern "C" int rbx_softmax_ce_bwd(const float* d_logits, int64_t stride, int64_t rows, int32_t n_classes,
                                  const int64_t* d_target, const float* d_lse, const float* d_dloss, float* d_dlogits,
                                  void* stream) {
  using namespace rbx;
  if (rows <= 0 || n_classes <= 0) return RBX_OK;
  if (!d_logits || !d_lse || !d_dloss || !d_dlogits) return fail(RBX_ERR_INVALID, "softmax_ce_bwd: NULL tensor");
  long long blocks = (rows * n_classes + 255) / 256;
  if (blocks > kCUs * 8) blocks = kCUs * 8;
  hipLaunchKernelGGL(softmax_ce_bwd_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, as_stream(stream), d_logits,
                     static_cast<long long>(stride), static_cast<long long>(rows), n_classes,
                     reinterpret_cast<const long long*>(d_target), d_lse, d_dloss, 1.0f / static_cast<float>(rows), d_dlogits);
  return check_launch("softmax_ce_bwd_kernel");
}

extern "C" int rbx_pair_logsigmoid_fwd(const float* d_pos, const float* d_neg, const float* d_weight, int64_t n, float scale,
                                       float* d_loss, void* d_workspace, size_t workspace_bytes, void* stream) {
  using namespace rbx;
  if (n <= 0) return fail(RBX_ERR_INVALID, "pair_logsigmoid: empty input");
  if (!d_pos || !d_neg || !d_loss) return fail(RBX_ERR_INVALID, "pair_logsigmoid: NULL tensor");
  const long long nb = (n + 1023) / 1024;
  if (d_workspace == nullptr || workspace_bytes < static_cast<size_t>(nb) * sizeof(float)) return fail(RBX_ERR_WORKSPACE, "pair_logsigmoid: workspace too small");
  if (nb >= INT_MAX) return fail(RBX_ERR_UNSUPPORTED, "pair_logsigmoid: too many elements");
  float* partial = static_cast<float*>(d_workspace);
  hipLaunchKernelGGL(pair_logsig_fwd_kernel, dim3(static_cast<unsigned>(nb)), dim3(256), 0, as_stream(stream), d_pos, d_neg,
                     d_weight, static_cast<long long>(n), partial);
  hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(256), 0, as_stream(stream), partial, static_cast<int>(nb), scale, d_loss);
  return check_launch("pair_logsigmoid forward kernels");
}

extern "C" int rbx_pair_logsigmoid_bwd(const float* d_pos, const float* d_neg, const float* d_weight, const float* d_dloss,
                                       int64_t n, float scale, float* d_dpos, float* d_dneg, void* stream) {
  using namespace rbx;
  if (n <= 0) return RBX_OK;
  if (!d_pos || !d_neg || !d_dloss || !d_dpos || !d_dneg) return fail(RBX_ERR_INVALID, "pair_logsigmoid_bwd: NULL tensor");
  long long blocks = (n + 255) / 256;
  if (blocks > kCUs * 8) blocks = kCUs * 8;
  hipLaunchKernelGGL(pair_logsig_bwd_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, as_stream(stream), d_pos, d_neg,
                     d_weight, d_dloss, scale, static_cast<long long>(n), d_dpos, d_dneg);
  return check_launch("pair_logsig_bwd_kernel");
}
