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

constexpr int kBnRows = 256;      // rows per workgroup of the reduction sweeps (a thread walks rows / 4 of them, four in flight)
// ... of a SHORT batch (fewer than 32 768 rows: the per-GPU batch of an 8-GPU strong-scaling run): 64, or the sweep is 128
// workgroups of sixteen dependent round trips each -- bn_bwd_partial took 24-38 us for 8 MB (profiles/r06/small_batches.txt)
static inline int bn_rows(long long rows) { return rows >= 32768 ? kBnRows : 64; }

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
                                                               float* __restrict__ partial, const int rpb) {
  __shared__ Welford red[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const long long r0 = static_cast<long long>(blockIdx.y) * rpb;
  const long long r1 = (r0 + rpb < rows) ? r0 + rpb : rows;
  Welford w = {0.f, 0.f, 0.f};
  if (c < cols) {
    long long r = r0 + (threadIdx.x >> 6);
    for (; r + 28 < r1; r += 32) {                           // 8 rows in flight, folded in ascending order
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = x[(r + 4 * u) * cols + c];
#pragma unroll
      for (int u = 0; u < 8; ++u) w.add(v[u]);
    }
    for (; r < r1; r += 4) w.add(x[r * cols + c]);
  }
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

// a workgroup owns 64 columns: its 16 wavefronts merge every 16th row block (in order), the sixteen results are merged
// in a fixed order; then mean / rstd and the running statistics.  (Four wavefronts walked 128 partials each at B = 65 536,
// eight loads in flight at a time: sixteen dependent round trips, 13-24 us for a few KB.)
constexpr int kBnFinalWaves = 16;
__global__ __launch_bounds__(64 * kBnFinalWaves) void bn_stats_final_kernel(const float* __restrict__ partial, const int nblocks,
                                                             const int cols,
                                                             const float eps, const float momentum,
                                                             float* __restrict__ running_mean,
                                                             float* __restrict__ running_var, float* __restrict__ mean,
                                                             float* __restrict__ rstd, float* __restrict__ raw) {
  __shared__ Welford red[kBnFinalWaves][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), zl = threadIdx.x >> 6;
  Welford t = {0.f, 0.f, 0.f};
  if (c < cols)
  {
    int b = zl;
    for (; b + 15 * kBnFinalWaves < nblocks; b += 16 * kBnFinalWaves) {    // 16 partials in flight, merged in ascending order
      Welford o[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const float* src = partial + (static_cast<long long>(b + kBnFinalWaves * u) * cols + c) * 3;
        o[u] = {src[0], src[1], src[2]};
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) t.merge(o[u]);
    }
    for (; b + 7 * kBnFinalWaves < nblocks; b += 8 * kBnFinalWaves) {      // 8 partials in flight
      Welford o[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const float* src = partial + (static_cast<long long>(b + kBnFinalWaves * u) * cols + c) * 3;
        o[u] = {src[0], src[1], src[2]};
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) t.merge(o[u]);
    }
    for (; b < nblocks; b += kBnFinalWaves) {
      const float* src = partial + (static_cast<long long>(b) * cols + c) * 3;
      const Welford o = {src[0], src[1], src[2]};
      t.merge(o);
    }
  }
  red[zl][threadIdx.x & 63] = t;
  __syncthreads();
  if (zl != 0 || c >= cols) return;
  t = red[0][threadIdx.x];
#pragma unroll
  for (int z = 1; z < kBnFinalWaves; ++z) t.merge(red[z][threadIdx.x]);
  if (raw != nullptr) {                                          // this rank's share of a synchronised BatchNorm: n, mean, M2
    raw[c] = t.n;
    raw[cols + c] = t.mean;
    raw[2 * cols + c] = t.m2;
    return;
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

// y = (x - mean) * rstd * gamma + beta, optional ReLU; 4 columns per lane when cols % 4 == 0.
// The host picks a grid whose stride is a multiple of `cols` whenever it can (bn_grid): a lane then stays on the same
// columns for the whole sweep, keeps their statistics in registers and has 4 independent loads in flight; other grids
// take the generic loop (a 64-bit modulo and 4 parameter loads per element).
template <bool VEC>
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, const long long rows, const int cols,
                                                       const float* __restrict__ mean, const float* __restrict__ rstd,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const int relu, float* __restrict__ y,
                                                       const float* __restrict__ slope, const int slope_n) {
  // slope != NULL: PReLU behind the normalisation (y = z > 0 ? z : a z; one a, or one per column) in the same pass
  constexpr int W = VEC ? 4 : 1;
  constexpr int U = 4;
  const long long total = rows * cols / W;
  const long long step = static_cast<long long>(gridDim.x) * blockDim.x;
  long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if ((step * W) % cols == 0) {
    if (i >= total) return;
    const int c = static_cast<int>((i * W) % cols);
    float m[W], rs[W], g[W], b[W], a[W];
#pragma unroll
    for (int k = 0; k < W; ++k) {
      m[k] = mean[c + k];
      rs[k] = rstd[c + k];
      g[k] = gamma != nullptr ? gamma[c + k] : 1.f;
      b[k] = beta != nullptr ? beta[c + k] : 0.f;
      a[k] = slope != nullptr ? slope[slope_n > 1 ? c + k : 0] : 0.f;
    }
    auto one = [&](const float (&v)[W], const long long e) {
      float o[W];
#pragma unroll
      for (int k = 0; k < W; ++k) {
        o[k] = (v[k] - m[k]) * rs[k] * g[k] + b[k];
        if (relu && o[k] < 0.f) o[k] = 0.f;
        if (slope != nullptr && o[k] < 0.f) o[k] *= a[k];
      }
      if constexpr (VEC) *reinterpret_cast<float4*>(y + e) = make_float4(o[0], o[1], o[2], o[3]);
      else y[e] = o[0];
    };
    for (; i + (U - 1) * step < total; i += U * step) {
      float v[U][W];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long long e = (i + u * step) * W;
        if constexpr (VEC) {
          const float4 t = *reinterpret_cast<const float4*>(x + e);
          v[u][0] = t.x; v[u][1] = t.y; v[u][2] = t.z; v[u][3] = t.w;
        } else {
          v[u][0] = x[e];
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) one(v[u], (i + u * step) * W);
    }
    for (; i < total; i += step) {
      float v[W];
      const long long e = i * W;
      if constexpr (VEC) {
        const float4 t = *reinterpret_cast<const float4*>(x + e);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
      } else {
        v[0] = x[e];
      }
      one(v, e);
    }
    return;
  }
  for (; i < total; i += step) {
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
      if (slope != nullptr && o[k] < 0.f) o[k] *= slope[slope_n > 1 ? c + k : 0];
    }
    if constexpr (VEC) *reinterpret_cast<float4*>(y + e) = make_float4(o[0], o[1], o[2], o[3]);
    else y[e] = o[0];
  }
}

// partial[(rb * cols + c) * 2 + {0,1}] = (sum dy, sum dy * xhat) over the row block
// PReLU behind the normalisation (slope != NULL; gamma / beta rebuild z = xhat gamma + beta, whose sign is the mask):
// dz = z > 0 ? dy : a dy takes dy's place, and a third partial, sum dy z over z <= 0 (d slope), goes to `partial3`
// [(rb * cols + c)].
struct BnPrelu {
  const float* slope;      // NULL: no PReLU
  int n;                   // 1, or cols
  const float* gamma;
  const float* beta;
};
__global__ __launch_bounds__(256) void bn_bwd_partial_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                             const float* __restrict__ y_relu, const long long rows,
                                                             const int cols,
                                                             const float* __restrict__ mean, const float* __restrict__ rstd,
                                                             float* __restrict__ partial, const BnPrelu pr,
                                                             float* __restrict__ partial3, const int rpb) {
  __shared__ float red[2][4][64];
  __shared__ float red3[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const long long r0 = static_cast<long long>(blockIdx.y) * rpb;
  const long long r1 = (r0 + rpb < rows) ? r0 + rpb : rows;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f;
  if (c < cols && pr.slope != nullptr) {
    const float m = mean[c], rs = rstd[c];
    const float gm = pr.gamma != nullptr ? pr.gamma[c] : 1.f, bt = pr.beta != nullptr ? pr.beta[c] : 0.f;
    const float a = pr.slope[pr.n > 1 ? c : 0];
    for (long long r = r0 + (threadIdx.x >> 6); r < r1; r += 4) {
      const float gy = dy[r * cols + c];
      const float xhat = (x[r * cols + c] - m) * rs;
      const float z = xhat * gm + bt;
      const float g = z > 0.f ? gy : a * gy;
      s0 += g;
      s1 += g * xhat;
      if (!(z > 0.f)) s2 += gy * z;
    }
  } else if (c < cols) {
    const float m = mean[c], rs = rstd[c];
    long long r = r0 + (threadIdx.x >> 6);
    for (; r + 12 < r1; r += 16) {                           // 4 rows in flight, added in ascending order
      float gg[4], xx[4], yy[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long long q = (r + 4 * u) * cols + c;
        gg[u] = dy[q];
        xx[u] = x[q];
        yy[u] = y_relu != nullptr ? y_relu[q] : 1.f;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float g = (yy[u] > 0.f) ? gg[u] : 0.f;         // fused ReLU: its mask is y > 0
        s0 += g;
        s1 += g * ((xx[u] - m) * rs);
      }
    }
    for (; r < r1; r += 4) {
      float g = dy[r * cols + c];
      if (y_relu != nullptr && !(y_relu[r * cols + c] > 0.f)) g = 0.f;     // fused ReLU: its mask is y > 0
      s0 += g;
      s1 += g * ((x[r * cols + c] - m) * rs);
    }
  }
  red[0][threadIdx.x >> 6][threadIdx.x & 63] = s0;
  red[1][threadIdx.x >> 6][threadIdx.x & 63] = s1;
  red3[threadIdx.x >> 6][threadIdx.x & 63] = s2;
  __syncthreads();
  if (threadIdx.x < 64 && c < cols) {
    float* dst = partial + (static_cast<long long>(blockIdx.y) * cols + c) * 2;
    dst[0] = (red[0][0][threadIdx.x] + red[0][1][threadIdx.x]) + (red[0][2][threadIdx.x] + red[0][3][threadIdx.x]);
    dst[1] = (red[1][0][threadIdx.x] + red[1][1][threadIdx.x]) + (red[1][2][threadIdx.x] + red[1][3][threadIdx.x]);
    if (partial3 != nullptr)
      partial3[static_cast<long long>(blockIdx.y) * cols + c] =
          (red3[0][threadIdx.x] + red3[1][threadIdx.x]) + (red3[2][threadIdx.x] + red3[3][threadIdx.x]);
  }
}

// a workgroup owns 64 columns; its 4 wavefronts take every 4th row block and meet in LDS (fixed order)
__global__ __launch_bounds__(256) void bn_bwd_final_kernel(const float* __restrict__ partial, const int nblocks, const int cols,
                                                           float* __restrict__ dbeta, float* __restrict__ dgamma,
                                                           const float* __restrict__ partial3, float* __restrict__ dslope) {
  __shared__ float red[2][4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), zl = threadIdx.x >> 6;
  if (partial3 != nullptr && dslope != nullptr) {            // d slope per column (PReLU): same walk, its own buffer
    __shared__ float r3[4][64];
    float s2 = 0.f;
    if (c < cols)
      for (int b = zl; b < nblocks; b += 4) s2 += partial3[static_cast<long long>(b) * cols + c];
    r3[zl][threadIdx.x & 63] = s2;
    __syncthreads();
    if (zl == 0 && c < cols) dslope[c] = (r3[0][threadIdx.x] + r3[1][threadIdx.x]) + (r3[2][threadIdx.x] + r3[3][threadIdx.x]);
    __syncthreads();
  }
  float s0 = 0.f, s1 = 0.f;
  if (c < cols) {
    int b = zl;
    for (; b + 28 < nblocks; b += 32) {                     // 8 partials in flight, added in the same (ascending) order
      float2 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        v[u] = *reinterpret_cast<const float2*>(partial + (static_cast<long long>(b + 4 * u) * cols + c) * 2);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        s0 += v[u].x;
        s1 += v[u].y;
      }
    }
    for (; b < nblocks; b += 4) {
      const float* src = partial + (static_cast<long long>(b) * cols + c) * 2;
      s0 += src[0];
      s1 += src[1];
    }
  }
  red[0][zl][threadIdx.x & 63] = s0;
  red[1][zl][threadIdx.x & 63] = s1;
  __syncthreads();
  if (zl == 0 && c < cols) {
    const int t = threadIdx.x;
    dbeta[c] = (red[0][0][t] + red[0][1][t]) + (red[0][2][t] + red[0][3][t]);
    dgamma[c] = (red[1][0][t] + red[1][1][t]) + (red[1][2][t] + red[1][3][t]);
  }
}

template <bool VEC>
__global__ __launch_bounds__(256) void bn_bwd_dx_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                        const float* __restrict__ y_relu, const long long rows,
                                                        const int cols,
                                                        const float* __restrict__ mean, const float* __restrict__ rstd,
                                                        const float* __restrict__ gamma, const float* __restrict__ dbeta,
                                                        const float* __restrict__ dgamma, const int training,
                                                        float* __restrict__ dx, const float inv_m, const BnPrelu pr) {
  constexpr int W = VEC ? 4 : 1;
  constexpr int U = 2;
  const long long total = rows * cols / W;
  const long long step = static_cast<long long>(gridDim.x) * blockDim.x;
  long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if ((step * W) % cols == 0) {                             // fixed columns per lane (bn_grid): statistics in registers
    if (i >= total) return;
    const int c = static_cast<int>((i * W) % cols);
    float m[W], rs[W], g[W], db[W], dg[W], bt[W], sl[W];
#pragma unroll
    for (int k = 0; k < W; ++k) {
      m[k] = mean[c + k];
      rs[k] = rstd[c + k];
      g[k] = gamma != nullptr ? gamma[c + k] : 1.f;
      db[k] = dbeta[c + k];
      dg[k] = dgamma[c + k];
      bt[k] = (pr.slope != nullptr && pr.beta != nullptr) ? pr.beta[c + k] : 0.f;
      sl[k] = pr.slope != nullptr ? pr.slope[pr.n > 1 ? c + k : 0] : 0.f;
    }
    auto load = [&](const float* __restrict__ src, const long long e, float (&v)[W]) {
      if constexpr (VEC) {
        const float4 t = *reinterpret_cast<const float4*>(src + e);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
      } else {
        v[0] = src[e];
      }
    };
    auto one = [&](const float (&xv)[W], float (&gv)[W], const float (&yv)[W], const long long e) {
      float o[W];
#pragma unroll
      for (int k = 0; k < W; ++k) {
        if (y_relu != nullptr && !(yv[k] > 0.f)) gv[k] = 0.f;
        if (pr.slope != nullptr && !((xv[k] - m[k]) * rs[k] * g[k] + bt[k] > 0.f)) gv[k] *= sl[k];
        if (training) {
          const float xhat = (xv[k] - m[k]) * rs[k];
          o[k] = g[k] * rs[k] * (gv[k] - db[k] * inv_m - xhat * dg[k] * inv_m);
        } else {
          o[k] = g[k] * rs[k] * gv[k];
        }
      }
      if constexpr (VEC) *reinterpret_cast<float4*>(dx + e) = make_float4(o[0], o[1], o[2], o[3]);
      else dx[e] = o[0];
    };
    for (; i + (U - 1) * step < total; i += U * step) {
      float xv[U][W], gv[U][W], yv[U][W];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long long e = (i + u * step) * W;
        load(x, e, xv[u]);
        load(dy, e, gv[u]);
        if (y_relu != nullptr) load(y_relu, e, yv[u]);
        else
#pragma unroll
          for (int k = 0; k < W; ++k) yv[u][k] = 1.f;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) one(xv[u], gv[u], yv[u], (i + u * step) * W);
    }
    for (; i < total; i += step) {
      float xv[W], gv[W], yv[W];
      const long long e = i * W;
      load(x, e, xv);
      load(dy, e, gv);
      if (y_relu != nullptr) load(y_relu, e, yv);
      else
#pragma unroll
        for (int k = 0; k < W; ++k) yv[k] = 1.f;
      one(xv, gv, yv, e);
    }
    return;
  }
  for (; i < total; i += step) {
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
      if (pr.slope != nullptr) {
        const float z = (xv[k] - mean[c + k]) * rs * g + (pr.beta != nullptr ? pr.beta[c + k] : 0.f);
        if (!(z > 0.f)) gv[k] *= pr.slope[pr.n > 1 ? c + k : 0];
      }
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

// grid of the element-wise sweeps: as many workgroups as the cap allows, rounded down to a count whose stride
// (blocks * 256 lanes * W columns) is a multiple of `cols`, so that every lane keeps its columns (see bn_apply_kernel)
static unsigned bn_grid(long long total_vecs, int cols, int w) {
  long long blocks = (total_vecs + 255) / 256;
  const long long cap = kCUs * 16;
  if (blocks > cap) blocks = cap;
  long long a = cols, b = 256LL * w;
  while (b != 0) { const long long t = a % b; a = b; b = t; }
  const long long unit = cols / a;                          // blocks must be a multiple of cols / gcd(cols, 256 * W)
  if (unit <= blocks) blocks = blocks / unit * unit;
  return static_cast<unsigned>(blocks);
}

static int bn_blocks(int64_t rows) { return static_cast<int>((rows + bn_rows(rows) - 1) / bn_rows(rows)); }
static bool bn_vec(int cols, const void* a, const void* b, const void* c) {
  return cols % 4 == 0 && ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c)) & 15) == 0;
}

}  // namespace rbx

extern "C" size_t rbx_batchnorm_workspace_size(int64_t rows, int32_t cols) {
  if (rows <= 0 || cols <= 0) return 0;
  return static_cast<size_t>(rbx::bn_blocks(rows)) * cols * 3 * sizeof(float) + 256;
}

namespace rbx {
static int bn_fwd_impl(const float* d_x, int64_t rows, int32_t cols, const float* d_gamma, const float* d_beta,
                       float eps, int32_t training, float momentum, float* d_running_mean, float* d_running_var,
                       int32_t relu, float* d_mean, float* d_rstd, float* d_y, void* d_workspace,
                       size_t workspace_bytes, void* stream, const float* d_slope, int slope_n) {
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
                       cols, partial, bn_rows(rows));
    hipLaunchKernelGGL(bn_stats_final_kernel, dim3((cols + 63) / 64), dim3(64 * kBnFinalWaves), 0, s, partial, nb, cols, eps, momentum, d_running_mean,
                       d_running_var, d_mean, d_rstd, static_cast<float*>(nullptr));
  } else {
    if (!d_running_mean || !d_running_var) return fail(RBX_ERR_INVALID, "batchnorm: eval mode needs the running statistics");
    hipLaunchKernelGGL(bn_eval_stats_kernel, dim3(cb), dim3(256), 0, s, d_running_mean, d_running_var, cols, eps, d_mean,
                       d_rstd);
  }
  const bool vec = bn_vec(cols, d_x, d_y, d_y);
  const long long total = static_cast<long long>(rows) * cols / (vec ? 4 : 1);
  const unsigned blocks = bn_grid(total, cols, vec ? 4 : 1);
  if (vec)
    hipLaunchKernelGGL(bn_apply_kernel<true>, dim3(blocks), dim3(256), 0, s, d_x,
                       static_cast<long long>(rows), cols, d_mean, d_rstd, d_gamma, d_beta, relu, d_y, d_slope, slope_n);
  else
    hipLaunchKernelGGL(bn_apply_kernel<false>, dim3(blocks), dim3(256), 0, s, d_x,
                       static_cast<long long>(rows), cols, d_mean, d_rstd, d_gamma, d_beta, relu, d_y, d_slope, slope_n);
  return check_launch("batchnorm forward kernels");
}
}  // namespace rbx

extern "C" int rbx_batchnorm_fwd(const float* d_x, int64_t rows, int32_t cols, const float* d_gamma, const float* d_beta,
                                 float eps, int32_t training, float momentum, float* d_running_mean, float* d_running_var,
                                 int32_t relu, float* d_mean, float* d_rstd, float* d_y, void* d_workspace,
                                 size_t workspace_bytes, void* stream) {
  return rbx::bn_fwd_impl(d_x, rows, cols, d_gamma, d_beta, eps, training, momentum, d_running_mean, d_running_var, relu,
                          d_mean, d_rstd, d_y, d_workspace, workspace_bytes, stream, nullptr, 0);
}

// nn.PReLU behind the BatchNorm of a tower layer (third_party/rechub/basic/layers.py:255-263 with activation="prelu":
// DSSM's towers, BASELINE cfg 1; activation.py:44-45) in the same passes: y = z > 0 ? z : a z with z the normalised
// value, a = d_slope[0] (slope_n == 1, nn.PReLU()'s default) or one per column.
extern "C" int rbx_batchnorm_prelu_fwd(const float* d_x, int64_t rows, int32_t cols, const float* d_gamma,
                                       const float* d_beta, const float* d_slope, int32_t slope_n, float eps,
                                       int32_t training, float momentum, float* d_running_mean, float* d_running_var,
                                       float* d_mean, float* d_rstd, float* d_y, void* d_workspace, size_t workspace_bytes,
                                       void* stream) {
  if (d_slope == nullptr || (slope_n != 1 && slope_n != cols))
    return rbx::fail(RBX_ERR_INVALID, "batchnorm_prelu: slope must hold 1 or cols values");
  return rbx::bn_fwd_impl(d_x, rows, cols, d_gamma, d_beta, eps, training, momentum, d_running_mean, d_running_var, 0,
                          d_mean, d_rstd, d_y, d_workspace, workspace_bytes, stream, d_slope, slope_n);
}

namespace rbx {
static int bn_bwd_reduce(const float* d_x, const float* d_dy, const float* d_y_relu, int64_t rows, int32_t cols,
                         const float* d_mean, const float* d_rstd, float* d_dgamma, float* d_dbeta, void* d_workspace,
                         size_t workspace_bytes, hipStream_t s, const BnPrelu pr = {nullptr, 0, nullptr, nullptr},
                         float* d_dslope = nullptr) {
  if (d_workspace == nullptr || workspace_bytes < rbx_batchnorm_workspace_size(rows, cols))
    return fail(RBX_ERR_WORKSPACE, "batchnorm_bwd: workspace too small");
  float* partial = static_cast<float*>(d_workspace);
  const int nb = bn_blocks(rows);
  float* partial3 = pr.slope != nullptr ? partial + static_cast<size_t>(nb) * cols * 2 : nullptr;   // (the workspace holds 3 per entry)
  hipLaunchKernelGGL(bn_bwd_partial_kernel, dim3((cols + 63) / 64, nb), dim3(256), 0, s, d_x, d_dy, d_y_relu,
                     static_cast<long long>(rows), cols, d_mean, d_rstd, partial, pr, partial3, bn_rows(rows));
  hipLaunchKernelGGL(bn_bwd_final_kernel, dim3((cols + 63) / 64), dim3(256), 0, s, partial, nb, cols, d_dbeta, d_dgamma,
                     static_cast<const float*>(partial3), d_dslope);
  return check_launch("batchnorm backward reductions");
}

static int bn_bwd_dx(const float* d_x, const float* d_dy, const float* d_y_relu, int64_t rows, int32_t cols,
                     const float* d_gamma, const float* d_mean, const float* d_rstd, const float* d_dgamma,
                     const float* d_dbeta, int32_t training, int64_t total_rows, float* d_dx, hipStream_t s,
                     const BnPrelu pr = {nullptr, 0, nullptr, nullptr}) {
  const bool vec = bn_vec(cols, d_x, d_dy, d_dx);
  const long long total = static_cast<long long>(rows) * cols / (vec ? 4 : 1);
  const unsigned blocks = bn_grid(total, cols, vec ? 4 : 1);
  const float inv_m = 1.0f / static_cast<float>(total_rows);
  if (vec)
    hipLaunchKernelGGL(bn_bwd_dx_kernel<true>, dim3(blocks), dim3(256), 0, s, d_x, d_dy, d_y_relu,
                       static_cast<long long>(rows), cols, d_mean, d_rstd, d_gamma, d_dbeta, d_dgamma, training, d_dx, inv_m, pr);
  else
    hipLaunchKernelGGL(bn_bwd_dx_kernel<false>, dim3(blocks), dim3(256), 0, s, d_x, d_dy, d_y_relu,
                       static_cast<long long>(rows), cols, d_mean, d_rstd, d_gamma, d_dbeta, d_dgamma, training, d_dx, inv_m, pr);
  return check_launch("batchnorm backward dx");
}
}  // namespace rbx

extern "C" int rbx_batchnorm_bwd(const float* d_x, const float* d_dy, const float* d_y_relu, int64_t rows, int32_t cols,
                                 const float* d_gamma, const float* d_mean, const float* d_rstd, int32_t training,
                                 float* d_dx, float* d_dgamma,
                                 float* d_dbeta, void* d_workspace, size_t workspace_bytes, void* stream) {
  using namespace rbx;
  if (rows < 0 || cols <= 0) return fail(RBX_ERR_INVALID, "batchnorm_bwd: bad shape");
  if (rows == 0) return RBX_OK;
  if (!d_x || !d_dy || !d_mean || !d_rstd || !d_dgamma || !d_dbeta)
    return fail(RBX_ERR_INVALID, "batchnorm_bwd: NULL tensor (d_dgamma / d_dbeta are scratch even when unused)");
  hipStream_t s = as_stream(stream);
  int rc = bn_bwd_reduce(d_x, d_dy, d_y_relu, rows, cols, d_mean, d_rstd, d_dgamma, d_dbeta, d_workspace, workspace_bytes, s);
  if (rc != RBX_OK || d_dx == nullptr) return rc;
  return bn_bwd_dx(d_x, d_dy, d_y_relu, rows, cols, d_gamma, d_mean, d_rstd, d_dgamma, d_dbeta, training, rows, d_dx, s);
}

// backward of rbx_batchnorm_prelu_fwd: d_dslope_cols[cols] = per-column sums of dy z over z <= 0 (the gradient of a
// per-column slope; summed over the columns by the caller for nn.PReLU()'s single parameter)
extern "C" int rbx_batchnorm_prelu_bwd(const float* d_x, const float* d_dy, int64_t rows, int32_t cols, const float* d_gamma,
                                       const float* d_beta, const float* d_slope, int32_t slope_n, const float* d_mean,
                                       const float* d_rstd, int32_t training, float* d_dx, float* d_dgamma, float* d_dbeta,
                                       float* d_dslope_cols, void* d_workspace, size_t workspace_bytes, void* stream) {
  using namespace rbx;
  if (rows < 0 || cols <= 0) return fail(RBX_ERR_INVALID, "batchnorm_prelu_bwd: bad shape");
  if (rows == 0) return RBX_OK;
  if (!d_x || !d_dy || !d_mean || !d_rstd || !d_dgamma || !d_dbeta || !d_slope || !d_dslope_cols)
    return fail(RBX_ERR_INVALID, "batchnorm_prelu_bwd: NULL tensor");
  if (slope_n != 1 && slope_n != cols) return fail(RBX_ERR_INVALID, "batchnorm_prelu: slope must hold 1 or cols values");
  hipStream_t s = as_stream(stream);
  const BnPrelu pr = {d_slope, slope_n, d_gamma, d_beta};
  int rc = bn_bwd_reduce(d_x, d_dy, nullptr, rows, cols, d_mean, d_rstd, d_dgamma, d_dbeta, d_workspace, workspace_bytes, s, pr,
                         d_dslope_cols);
  if (rc != RBX_OK || d_dx == nullptr) return rc;
  return bn_bwd_dx(d_x, d_dy, nullptr, rows, cols, d_gamma, d_mean, d_rstd, d_dgamma, d_dbeta, training, rows, d_dx, s, pr);
}

// ---- the pieces of a SYNCHRONISED BatchNorm (torch.nn.SyncBatchNorm; RecBole's DDP path converts every BatchNorm of the
// model, third_party/recbole/trainer/trainer.py:60-64): the statistics of one normalisation come from the batches of all
// ranks, so the collectives sit between these calls (recbox_amd/ops.py: _SyncBatchNorm).
extern "C" int rbx_batchnorm_stats(const float* d_x, int64_t rows, int32_t cols, float* d_raw, void* d_workspace,
                                   size_t workspace_bytes, void* stream) {
  using namespace rbx;
  if (rows <= 0 || cols <= 0) return fail(RBX_ERR_INVALID, "batchnorm_stats: bad shape");
  if (!d_x || !d_raw) return fail(RBX_ERR_INVALID, "batchnorm_stats: NULL tensor");
  if (d_workspace == nullptr || workspace_bytes < rbx_batchnorm_workspace_size(rows, cols))
    return fail(RBX_ERR_WORKSPACE, "batchnorm_stats: workspace too small");
  hipStream_t s = as_stream(stream);
  float* partial = static_cast<float*>(d_workspace);
  const int nb = bn_blocks(rows);
  hipLaunchKernelGGL(bn_stats_partial_kernel, dim3((cols + 63) / 64, nb), dim3(256), 0, s, d_x, static_cast<long long>(rows),
                     cols, partial, bn_rows(rows));
  hipLaunchKernelGGL(bn_stats_final_kernel, dim3((cols + 63) / 64), dim3(64 * kBnFinalWaves), 0, s, partial, nb, cols, 0.f, 0.f,
                     static_cast<float*>(nullptr), static_cast<float*>(nullptr), static_cast<float*>(nullptr),
                     static_cast<float*>(nullptr), d_raw);
  return check_launch("batchnorm statistics kernels");
}

extern "C" int rbx_batchnorm_apply(const float* d_x, int64_t rows, int32_t cols, const float* d_gamma, const float* d_beta,
                                   const float* d_mean, const float* d_rstd, int32_t relu, float* d_y, void* stream) {
  using namespace rbx;
  if (rows < 0 || cols <= 0) return fail(RBX_ERR_INVALID, "batchnorm_apply: bad shape");
  if (rows == 0) return RBX_OK;
  if (!d_x || !d_y || !d_mean || !d_rstd) return fail(RBX_ERR_INVALID, "batchnorm_apply: NULL tensor");
  const bool vec = bn_vec(cols, d_x, d_y, d_y);
  const long long total = static_cast<long long>(rows) * cols / (vec ? 4 : 1);
  const unsigned blocks = bn_grid(total, cols, vec ? 4 : 1);
  hipStream_t s = as_stream(stream);
  if (vec)
    hipLaunchKernelGGL(bn_apply_kernel<true>, dim3(blocks), dim3(256), 0, s, d_x, static_cast<long long>(rows), cols, d_mean,
                       d_rstd, d_gamma, d_beta, relu, d_y, static_cast<const float*>(nullptr), 0);
  else
    hipLaunchKernelGGL(bn_apply_kernel<false>, dim3(blocks), dim3(256), 0, s, d_x, static_cast<long long>(rows), cols, d_mean,
                       d_rstd, d_gamma, d_beta, relu, d_y, static_cast<const float*>(nullptr), 0);
  return check_launch("batchnorm apply kernel");
}

extern "C" int rbx_batchnorm_bwd_reduce(const float* d_x, const float* d_dy, const float* d_y_relu, int64_t rows,
                                        int32_t cols, const float* d_mean, const float* d_rstd, float* d_dgamma,
                                        float* d_dbeta, void* d_workspace, size_t workspace_bytes, void* stream) {
  using namespace rbx;
  if (rows <= 0 || cols <= 0) return fail(RBX_ERR_INVALID, "batchnorm_bwd_reduce: bad shape");
  if (!d_x || !d_dy || !d_mean || !d_rstd || !d_dgamma || !d_dbeta) return fail(RBX_ERR_INVALID, "batchnorm_bwd_reduce: NULL tensor");
  return bn_bwd_reduce(d_x, d_dy, d_y_relu, rows, cols, d_mean, d_rstd, d_dgamma, d_dbeta, d_workspace, workspace_bytes,
                       as_stream(stream));
}

extern "C" int rbx_batchnorm_bwd_dx(const float* d_x, const float* d_dy, const float* d_y_relu, int64_t rows, int32_t cols,
                                    const float* d_gamma, const float* d_mean, const float* d_rstd, const float* d_dgamma,
                                    const float* d_dbeta, int64_t total_rows, float* d_dx, void* stream) {
  using namespace rbx;
  if (rows <= 0 || cols <= 0 || total_rows < rows) return fail(RBX_ERR_INVALID, "batchnorm_bwd_dx: bad shape");
  if (!d_x || !d_dy || !d_mean || !d_rstd || !d_dgamma || !d_dbeta || !d_dx) return fail(RBX_ERR_INVALID, "batchnorm_bwd_dx: NULL tensor");
  return bn_bwd_dx(d_x, d_dy, d_y_relu, rows, cols, d_gamma, d_mean, d_rstd, d_dgamma, d_dbeta, 1, total_rows, d_dx,
                   as_stream(stream));
}

// ---- LayerNorm over the last dimension ----------------------------------------------------------------------
// Reference op replaced: the five nn.LayerNorm(D, eps=1e-8) of a SASRec block stack
// (third_party/rechub/models/matching/sasrec.py:52-63,81-94) applied to [B, L, D] activations (cfg 5:
// 819 200 rows of 64 floats, 210 MB).  ATen takes 0.39 ms forward + 1.69 ms backward per LayerNorm there
// (profiles/r01_models_bench.txt); the data could move in ~0.08 + 0.15 ms.
// A lane group of G = D/4 lanes (float4) owns one row: mean and variance by two shuffle reductions over
// registers (two-pass, no E[x^2] cancellation), y written once.  Backward: the row-wise part
//   dx = rstd * (g - mean_d(g) - xhat * mean_d(g * xhat)),  g = dy * gamma
// in the same mapping; dgamma = sum_rows dy * xhat and dbeta = sum_rows dy by the two-stage fixed-order column
// reduction used for BatchNorm.  HBM-bound streaming.
namespace rbx {

template <int G, int NV, bool VEC>
struct LnRow {
  static constexpr int W = VEC ? 4 : 1;
  float a[NV * W];
  template <bool NT = true>
  __device__ __forceinline__ void load(const float* row, int dim, int lane_g) {
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const int e = (lane_g + u * G) * W;
      if (e < dim) {
        // streamed (read once per pass, far larger than the caches): the non-temporal hint is worth a third of the rate on
        // this part (profiles/r03: a 420 MB read-only pass 4.2 -> 5.3 TB/s)
        // (NT = false: a gradient the previous kernel has just written -- measured slower with the hint)
        if constexpr (VEC && NT) {
          typedef float v4f __attribute__((ext_vector_type(4)));
          const v4f t = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(row + e));
          a[u * 4] = t[0]; a[u * 4 + 1] = t[1]; a[u * 4 + 2] = t[2]; a[u * 4 + 3] = t[3];
        } else if constexpr (VEC) {
          const float4 t = *reinterpret_cast<const float4*>(row + e);
          a[u * 4] = t.x; a[u * 4 + 1] = t.y; a[u * 4 + 2] = t.z; a[u * 4 + 3] = t.w;
        } else {
          a[u] = NT ? __builtin_nontemporal_load(row + e) : row[e];
        }
      } else {
#pragma unroll
        for (int k = 0; k < W; ++k) a[u * W + k] = 0.f;
      }
    }
  }
  __device__ __forceinline__ void store(float* row, int dim, int lane_g) const {
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const int e = (lane_g + u * G) * W;
      if (e < dim) {
        if constexpr (VEC) *reinterpret_cast<float4*>(row + e) = make_float4(a[u * 4], a[u * 4 + 1], a[u * 4 + 2], a[u * 4 + 3]);
        else row[e] = a[u];
      }
    }
  }
  // element index of slot i, or -1 when the slot lies beyond dim
  __device__ __forceinline__ static int index(int i, int dim, int lane_g) {
    const int e = (lane_g + (i / W) * G) * W + (i % W);
    return e < dim ? e : -1;
  }
};

template <int G, int NV, bool VEC>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const long long rows, const int dim,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     const float eps, float* __restrict__ mean, float* __restrict__ rstd,
                                                     float* __restrict__ y) {
  using Row = LnRow<G, NV, VEC>;
  constexpr int NA = NV * Row::W;
  const int lane_g = threadIdx.x % G;
  const long long ngroups = static_cast<long long>(gridDim.x) * (blockDim.x / G);
  const float inv_d = 1.0f / static_cast<float>(dim);
  // a lane's slice of gamma / beta, once: read where they are used, they were four pairs of loads with a wait each in every
  // row of the loop -- ten memory requests per row where two (the row in, the row out) move all of its bytes
  float gm[NA], bt[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int e = Row::index(i, dim, lane_g);
    gm[i] = (e >= 0 && gamma != nullptr) ? gamma[e] : 1.f;
    bt[i] = (e >= 0 && beta != nullptr) ? beta[e] : 0.f;
  }
  for (long long r = static_cast<long long>(blockIdx.x) * (blockDim.x / G) + threadIdx.x / G; r < rows; r += ngroups) {
    Row v;
    v.load(x + r * dim, dim, lane_g);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NA; ++i) s += v.a[i];
    const float m = group_sum<G>(s) * inv_d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const float d = (Row::index(i, dim, lane_g) >= 0) ? v.a[i] - m : 0.f;
      q += d * d;
    }
    const float rs = 1.0f / sqrtf(group_sum<G>(q) * inv_d + eps);
#pragma unroll
    for (int i = 0; i < NA; ++i) v.a[i] = (v.a[i] - m) * rs * gm[i] + bt[i];      // (slots beyond dim are not stored)
    v.store(y + r * dim, dim, lane_g);
    if (lane_g == 0) {
      mean[r] = m;
      rstd[r] = rs;
    }
  }
}

template <int G, int NV, bool VEC>
__global__ __launch_bounds__(256) void ln_bwd_dx_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                        const long long rows, const int dim,
                                                        const float* __restrict__ gamma, const float* __restrict__ mean,
                                                        const float* __restrict__ rstd, float* __restrict__ dx,
                                                        float* __restrict__ partial) {
  using Row = LnRow<G, NV, VEC>;
  constexpr int NA = NV * Row::W;
  const int lane_g = threadIdx.x % G;
  const long long ngroups = static_cast<long long>(gridDim.x) * (blockDim.x / G);
  const float inv_d = 1.0f / static_cast<float>(dim);
  // partial != NULL: the parameter gradients' block sums (sum dy, sum dy * xhat per column) fall out of the same pass --
  // dy and xhat are in registers anyway --, instead of a second kernel that reads x and dy again (210 + 210 MB at cfg 5).
  // A lane adds its rows in ascending order, the lane groups of the workgroup are added in a fixed order: deterministic.
  float pb[NA], pg[NA], gm[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    pb[i] = pg[i] = 0.f;
    const int e = Row::index(i, dim, lane_g);
    gm[i] = (e >= 0 && gamma != nullptr) ? gamma[e] : 1.f;          // (once, not per row: see ln_fwd_kernel)
  }
  for (long long r = static_cast<long long>(blockIdx.x) * (blockDim.x / G) + threadIdx.x / G; r < rows; r += ngroups) {
    Row xv, gv;
    xv.load(x + r * dim, dim, lane_g);
    gv.template load<false>(dy + r * dim, dim, lane_g);
    const float m = mean[r], rs = rstd[r];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NA; ++i) {                         // (slots beyond dim hold dy = 0: they add nothing)
      xv.a[i] = (xv.a[i] - m) * rs;                        // xhat
      pb[i] += gv.a[i];
      pg[i] += gv.a[i] * xv.a[i];
      gv.a[i] *= gm[i];                                    // g = dy * gamma
      s1 += gv.a[i];
      s2 += gv.a[i] * xv.a[i];
    }
    const float m1 = group_sum<G>(s1) * inv_d, m2 = group_sum<G>(s2) * inv_d;
#pragma unroll
    for (int i = 0; i < NA; ++i) gv.a[i] = rs * (gv.a[i] - m1 - xv.a[i] * m2);
    gv.store(dx + r * dim, dim, lane_g);
  }
  if (partial != nullptr) {
    __shared__ float red[2][256 * NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      red[0][threadIdx.x * NA + i] = pb[i];
      red[1][threadIdx.x * NA + i] = pg[i];
    }
    __syncthreads();
    for (int t = threadIdx.x; t < G * NA; t += 256) {
      const int lane = t / NA, i = t % NA;
      const int e = Row::index(i, dim, lane);
      if (e < 0) continue;
      float s0 = 0.f, s1 = 0.f;
      for (int gi = 0; gi < 256 / G; ++gi) {
        s0 += red[0][(gi * G + lane) * NA + i];
        s1 += red[1][(gi * G + lane) * NA + i];
      }
      float* dst = partial + (static_cast<long long>(blockIdx.x) * dim + e) * 2;
      dst[0] = s0;
      dst[1] = s1;
    }
  }
}

// partial[(rb * dim + c) * 2 + {0,1}] = (sum dy, sum dy * xhat) over a block of rows; statistics are per ROW here
__global__ __launch_bounds__(256) void ln_bwd_partial_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                             const long long rows, const int dim,
                                                             const float* __restrict__ mean, const float* __restrict__ rstd,
                                                             const int rows_per_block, float* __restrict__ partial) {
  __shared__ float red[2][4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const long long r0 = static_cast<long long>(blockIdx.y) * rows_per_block;
  const long long r1 = (r0 + rows_per_block < rows) ? r0 + rows_per_block : rows;
  float s0 = 0.f, s1 = 0.f;
  if (c < dim) {
    long long r = r0 + (threadIdx.x >> 6);
    for (; r + 12 < r1; r += 16) {                           // 4 rows in flight (16 loads), added in ascending order
      float gg[4], xx[4], mm[4], rr[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long long q = r + 4 * u;
        gg[u] = dy[q * dim + c];
        xx[u] = x[q * dim + c];
        mm[u] = mean[q];
        rr[u] = rstd[q];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        s0 += gg[u];
        s1 += gg[u] * ((xx[u] - mm[u]) * rr[u]);
      }
    }
    for (; r < r1; r += 4) {
      const float g = dy[r * dim + c];
      s0 += g;
      s1 += g * ((x[r * dim + c] - mean[r]) * rstd[r]);
    }
  }
  red[0][threadIdx.x >> 6][threadIdx.x & 63] = s0;
  red[1][threadIdx.x >> 6][threadIdx.x & 63] = s1;
  __syncthreads();
  if (threadIdx.x < 64 && c < dim) {
    float* dst = partial + (static_cast<long long>(blockIdx.y) * dim + c) * 2;
    dst[0] = (red[0][0][threadIdx.x] + red[0][1][threadIdx.x]) + (red[0][2][threadIdx.x] + red[0][3][threadIdx.x]);
    dst[1] = (red[1][0][threadIdx.x] + red[1][1][threadIdx.x]) + (red[1][2][threadIdx.x] + red[1][3][threadIdx.x]);
  }
}

constexpr int kLnRows = 1024;     // rows per workgroup of the dgamma / dbeta reduction
constexpr int kLnFusedBlocks = 4 * kCUs;   // workgroups of the dx kernel when it also leaves the parameter-gradient partials
static int ln_blocks(int64_t rows) { return static_cast<int>((rows + kLnRows - 1) / kLnRows); }

template <int G, int NV, bool VEC>
static int launch_ln(bool bwd, const float* x, const float* dy, int64_t rows, int dim, const float* gamma, const float* beta,
                     float eps, float* mean, float* rstd, float* out, hipStream_t s, float* partial = nullptr,
                     int* n_partial = nullptr) {
  const int gpb = 256 / G;
  long long blocks = (rows + gpb - 1) / gpb;
  if (blocks > kCUs * 16) blocks = kCUs * 16;
  if (partial != nullptr && blocks > kLnFusedBlocks) blocks = kLnFusedBlocks;   // (the final kernel adds one partial per workgroup)
  if (n_partial != nullptr) *n_partial = static_cast<int>(blocks);
  if (bwd)
    hipLaunchKernelGGL((ln_bwd_dx_kernel<G, NV, VEC>), dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s, x, dy,
                       static_cast<long long>(rows), dim, gamma, mean, rstd, out, partial);
  else
    hipLaunchKernelGGL((ln_fwd_kernel<G, NV, VEC>), dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s, x,
                       static_cast<long long>(rows), dim, gamma, beta, eps, mean, rstd, out);
  return check_launch("layernorm kernel");
}

template <bool VEC>
static int dispatch_ln(bool bwd, const float* x, const float* dy, int64_t rows, int dim, const float* gamma, const float* beta,
                       float eps, float* mean, float* rstd, float* out, hipStream_t s, float* partial = nullptr,
                       int* n_partial = nullptr) {
  switch (pow2_ceil(VEC ? dim / 4 : dim)) {
    case 1: return launch_ln<1, 1, VEC>(bwd, x, dy, rows, dim, gamma, beta, eps, mean, rstd, out, s, partial, n_partial);
    case 2: return launch_ln<2, 1, VEC>(bwd, x, dy, rows, dim, gamma, beta, eps, mean, rstd, out, s, partial, n_partial);
    case 4: return launch_ln<4, 1, VEC>(bwd, x, dy, rows, dim, gamma, beta, eps, mean, rstd, out, s, partial, n_partial);
    case 8: return launch_ln<8, 1, VEC>(bwd, x, dy, rows, dim, gamma, beta, eps, mean, rstd, out, s, partial, n_partial);
    case 16: return launch_ln<16, 1, VEC>(bwd, x, dy, rows, dim, gamma, beta, eps, mean, rstd, out, s, partial, n_partial);
    case 32: return launch_ln<32, 1, VEC>(bwd, x, dy, rows, dim, gamma, beta, eps, mean, rstd, out, s, partial, n_partial);
    case 64: return launch_ln<64, 1, VEC>(bwd, x, dy, rows, dim, gamma, beta, eps, mean, rstd, out, s, partial, n_partial);
    case 128: return launch_ln<64, 2, VEC>(bwd, x, dy, rows, dim, gamma, beta, eps, mean, rstd, out, s, partial, n_partial);
    case 256: return launch_ln<64, 4, VEC>(bwd, x, dy, rows, dim, gamma, beta, eps, mean, rstd, out, s, partial, n_partial);
    default: return fail(RBX_ERR_UNSUPPORTED, "layernorm: dim %d too large for one lane group", dim);
  }
}

static bool ln_vec(int dim, const void* a, const void* b, const void* c) {
  return dim % 4 == 0 && ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c)) & 15) == 0;
}

}  // namespace rbx

extern "C" int rbx_layernorm_fwd(const float* d_x, int64_t rows, int32_t dim, const float* d_gamma, const float* d_beta,
                                 float eps, float* d_mean, float* d_rstd, float* d_y, void* stream) {
  using namespace rbx;
  if (rows < 0 || dim <= 0) return fail(RBX_ERR_INVALID, "layernorm: bad shape");
  if (rows == 0) return RBX_OK;
  if (!d_x || !d_y || !d_mean || !d_rstd) return fail(RBX_ERR_INVALID, "layernorm: NULL tensor");
  return ln_vec(dim, d_x, d_y, d_y)
             ? dispatch_ln<true>(false, d_x, nullptr, rows, dim, d_gamma, d_beta, eps, d_mean, d_rstd, d_y, as_stream(stream))
             : dispatch_ln<false>(false, d_x, nullptr, rows, dim, d_gamma, d_beta, eps, d_mean, d_rstd, d_y, as_stream(stream));
}

extern "C" size_t rbx_layernorm_bwd_workspace_size(int64_t rows, int32_t dim) {
  if (rows <= 0 || dim <= 0) return 0;
  const size_t nb = static_cast<size_t>(rbx::ln_blocks(rows) > rbx::kLnFusedBlocks ? rbx::ln_blocks(rows) : rbx::kLnFusedBlocks);
  return nb * dim * 2 * sizeof(float) + 256;
}

extern "C" int rbx_layernorm_bwd(const float* d_x, const float* d_dy, int64_t rows, int32_t dim, const float* d_gamma,
                                 const float* d_mean, const float* d_rstd, float* d_dx, float* d_dgamma, float* d_dbeta,
                                 void* d_workspace, size_t workspace_bytes, void* stream) {
  using namespace rbx;
  if (rows < 0 || dim <= 0) return fail(RBX_ERR_INVALID, "layernorm_bwd: bad shape");
  if (rows == 0) return RBX_OK;
  if (!d_x || !d_dy || !d_mean || !d_rstd) return fail(RBX_ERR_INVALID, "layernorm_bwd: NULL tensor");
  hipStream_t s = as_stream(stream);
  int rc = RBX_OK;
  const bool want_p = d_dgamma != nullptr || d_dbeta != nullptr;
  if (want_p) {
    if (d_dgamma == nullptr || d_dbeta == nullptr) return fail(RBX_ERR_INVALID, "layernorm_bwd: d_dgamma and d_dbeta come together");
    if (d_workspace == nullptr || workspace_bytes < rbx_layernorm_bwd_workspace_size(rows, dim))
      return fail(RBX_ERR_WORKSPACE, "layernorm_bwd: workspace too small");
  }
  if (d_dx != nullptr) {
    // with the parameter gradients wanted too, their block partials come out of the dx pass (one read of x and dy)
    float* fused = want_p ? static_cast<float*>(d_workspace) : nullptr;
    int nb = 0;
    rc = ln_vec(dim, d_x, d_dy, d_dx)
             ? dispatch_ln<true>(true, d_x, d_dy, rows, dim, d_gamma, nullptr, 0.f, const_cast<float*>(d_mean),
                                 const_cast<float*>(d_rstd), d_dx, s, fused, &nb)
             : dispatch_ln<false>(true, d_x, d_dy, rows, dim, d_gamma, nullptr, 0.f, const_cast<float*>(d_mean),
                                  const_cast<float*>(d_rstd), d_dx, s, fused, &nb);
    if (rc != RBX_OK) return rc;
    if (want_p) {
      hipLaunchKernelGGL(bn_bwd_final_kernel, dim3((dim + 63) / 64), dim3(256), 0, s, fused, nb, dim, d_dbeta, d_dgamma,
                         static_cast<const float*>(nullptr), static_cast<float*>(nullptr));
      return check_launch("layernorm parameter-gradient kernel");
    }
    return rc;
  }
  if (want_p) {
    float* partial = static_cast<float*>(d_workspace);
    const int nb = ln_blocks(rows);
    hipLaunchKernelGGL(ln_bwd_partial_kernel, dim3((dim + 63) / 64, nb), dim3(256), 0, s, d_x, d_dy, static_cast<long long>(rows),
                       dim, d_mean, d_rstd, kLnRows, partial);
    hipLaunchKernelGGL(bn_bwd_final_kernel, dim3((dim + 63) / 64), dim3(256), 0, s, partial, nb, dim, d_dbeta, d_dgamma,
                       static_cast<const float*>(nullptr), static_cast<float*>(nullptr));
    rc = check_launch("layernorm parameter-gradient kernels");
  }
  return rc;
}
