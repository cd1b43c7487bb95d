// rbx_interaction.hip -- K4: feature-interaction layer on a materialised
// [B, F, D] embedding tensor (gfx950).
//
// Reference behaviour replaced: InnerProductInteraction.forward
// (ranking/pytorch/layers/interactions/inner_product.py:40-56) in its four output
// modes, and rechub FM (third_party/rechub/basic/layers.py:286-292), plus the
// autograd backward of each.
//   mode 0 product_sum        out[b]     = 0.5 * sum_d[(sum_f e)^2 - sum_f e^2]
//   mode 1 bi_interaction     out[b,d]   = 0.5 * [(sum_f e)^2 - sum_f e^2]
//   mode 2 inner_product      out[b,p]   = <e_i, e_j>, p over i<j row-major
//   mode 3 elementwise_product out[b,p,:] = e_i * e_j
// Modes 0/1 are HBM-streaming reductions: a lane group of D/4 lanes (float4 each)
// owns one sample and walks its F rows, so S = sum_f e lives in registers and the
// only cross-lane step is the final sum over d.  The backward recomputes S from the
// (L1/L2-hot) sample block instead of saving it: dE_f = g * (S - e_f).
// Modes 2/3 stage the sample's [F, D] block in LDS (one wavefront per sample).
#include "rbx_internal.h"

#define RBX_FM_ROW_UNROLL 4   // rows of a sample in flight per lane group: backward 120 -> 108 us at [65536, 39, 16] (8: 119)

namespace rbx {

template <int G, int NV, bool VEC>
__global__ __launch_bounds__(256) void fm_fwd_kernel(const float* __restrict__ emb, const long long sb, const long long B,
                                                     const int F, const int D, const int mode, float* __restrict__ out) {
  constexpr int W = VEC ? 4 : 1;
  const int lane_g = threadIdx.x % G;
  const long long ngroups = static_cast<long long>(gridDim.x) * (blockDim.x / G);
  for (long long b = static_cast<long long>(blockIdx.x) * (blockDim.x / G) + threadIdx.x / G; b < B; b += ngroups) {
    const float* base = emb + b * sb;
    float s[NV * W], q[NV * W];
#pragma unroll
    for (int i = 0; i < NV * W; ++i) s[i] = q[i] = 0.f;
#pragma unroll RBX_FM_ROW_UNROLL
    for (int f = 0; f < F; ++f) {
#pragma unroll
      for (int u = 0; u < NV; ++u) {
        const int e = (lane_g + u * G) * W;
        if (e < D) {
          if constexpr (VEC) {
            const float4 t = *reinterpret_cast<const float4*>(base + f * D + e);
            s[u * 4] += t.x; s[u * 4 + 1] += t.y; s[u * 4 + 2] += t.z; s[u * 4 + 3] += t.w;
            q[u * 4] += t.x * t.x; q[u * 4 + 1] += t.y * t.y; q[u * 4 + 2] += t.z * t.z; q[u * 4 + 3] += t.w * t.w;
          } else {
            const float t = base[f * D + e];
            s[u] += t;
            q[u] += t * t;
          }
        }
      }
    }
    if (mode == 1) {
#pragma unroll
      for (int u = 0; u < NV; ++u) {
        const int e = (lane_g + u * G) * W;
        if (e < D) {
#pragma unroll
          for (int k = 0; k < W; ++k) out[b * D + e + k] = (s[u * W + k] * s[u * W + k] - q[u * W + k]) * 0.5f;
        }
      }
    } else {
      float t = 0.f;
#pragma unroll
      for (int i = 0; i < NV * W; ++i) t += (s[i] * s[i] - q[i]) * 0.5f;
      t = group_sum<G>(t);
      if (lane_g == 0) out[b] = t;
    }
  }
}

template <int G, int NV, bool VEC>
__global__ __launch_bounds__(256) void fm_bwd_kernel(const float* __restrict__ emb, const long long sb,
                                                     const float* __restrict__ dout, const long long B, const int F,
                                                     const int D, const int mode, float* __restrict__ demb,
                                                     const long long dsb) {
  constexpr int W = VEC ? 4 : 1;
  const int lane_g = threadIdx.x % G;
  const long long ngroups = static_cast<long long>(gridDim.x) * (blockDim.x / G);
  for (long long b = static_cast<long long>(blockIdx.x) * (blockDim.x / G) + threadIdx.x / G; b < B; b += ngroups) {
    const float* base = emb + b * sb;
    float* dbase = demb + b * dsb;
    float s[NV * W], g[NV * W];
#pragma unroll
    for (int i = 0; i < NV * W; ++i) s[i] = 0.f;
#pragma unroll RBX_FM_ROW_UNROLL
    for (int f = 0; f < F; ++f) {
#pragma unroll
      for (int u = 0; u < NV; ++u) {
        const int e = (lane_g + u * G) * W;
        if (e < D) {
#pragma unroll
          for (int k = 0; k < W; ++k) s[u * W + k] += base[f * D + e + k];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const int e = (lane_g + u * G) * W;
#pragma unroll
      for (int k = 0; k < W; ++k) g[u * W + k] = (mode == 1) ? ((e < D) ? dout[b * D + e + k] : 0.f) : dout[b];
    }
#pragma unroll RBX_FM_ROW_UNROLL
    for (int f = 0; f < F; ++f) {
#pragma unroll
      for (int u = 0; u < NV; ++u) {
        const int e = (lane_g + u * G) * W;
        if (e < D) {
          if constexpr (VEC) {
            const float4 t = *reinterpret_cast<const float4*>(base + f * D + e);
            float4 r;
            r.x = g[u * 4] * (s[u * 4] - t.x);
            r.y = g[u * 4 + 1] * (s[u * 4 + 1] - t.y);
            r.z = g[u * 4 + 2] * (s[u * 4 + 2] - t.z);
            r.w = g[u * 4 + 3] * (s[u * 4 + 3] - t.w);
            *reinterpret_cast<float4*>(dbase + f * D + e) = r;
          } else {
            dbase[f * D + e] = g[u] * (s[u] - base[f * D + e]);
          }
        }
      }
    }
  }
}

__device__ __forceinline__ int pair_index(int i, int j, int F) { return i * F - (i * (i + 1)) / 2 + (j - i - 1); }

// one wavefront per sample; the sample's [F, D] block sits in LDS
__global__ __launch_bounds__(64) void pair_fwd_kernel(const float* __restrict__ emb, const long long sb, const int F,
                                                      const int D, const int mode, float* __restrict__ out) {
  extern __shared__ float se[];
  const long long b = blockIdx.x;
  const int FD = F * D, P = F * (F - 1) / 2;
  for (int i = threadIdx.x; i < FD; i += 64) se[i] = emb[b * sb + i];
  __syncthreads();
  if (mode == 2) {
    for (int i = 0; i < F - 1; ++i) {
      for (int j = i + 1 + threadIdx.x; j < F; j += 64) {
        float acc = 0.f;
        for (int d = 0; d < D; ++d) acc += se[i * D + d] * se[j * D + d];
        out[b * P + pair_index(i, j, F)] = acc;
      }
    }
  } else {
    for (int i = 0; i < F - 1; ++i) {
      const int n = (F - 1 - i) * D;                  // contiguous run of outputs for row i
      float* dst = out + (b * P + pair_index(i, i + 1, F)) * D;
      for (int t = threadIdx.x; t < n; t += 64) {
        const int j = i + 1 + t / D, d = t % D;
        dst[t] = se[i * D + d] * se[j * D + d];
      }
    }
  }
}

__global__ __launch_bounds__(64) void pair_bwd_kernel(const float* __restrict__ emb, const long long sb,
                                                      const float* __restrict__ dout, const int F, const int D,
                                                      const int mode, float* __restrict__ demb, const long long dsb) {
  extern __shared__ float se[];
  const long long b = blockIdx.x;
  const int FD = F * D, P = F * (F - 1) / 2;
  for (int i = threadIdx.x; i < FD; i += 64) se[i] = emb[b * sb + i];
  __syncthreads();
  for (int t = threadIdx.x; t < FD; t += 64) {
    const int i = t / D, d = t % D;
    float acc = 0.f;
    for (int j = 0; j < F; ++j) {
      if (j == i) continue;
      const int p = (i < j) ? pair_index(i, j, F) : pair_index(j, i, F);
      const float g = (mode == 2) ? dout[b * P + p] : dout[(b * P + p) * D + d];
      acc += g * se[j * D + d];
    }
    demb[b * dsb + t] = acc;
  }
}

// ---- bilinear pair products (BilinearInteraction / BilinearInteractionV2,
// ranking/pytorch/layers/interactions/bilinear_interaction.py:24-90) -------------------------------------------------
// out[b, p(i,j), :] = left(b, p, :) * right[b, j, :] for the F(F-1)/2 pairs i < j in triu order, where left is the
// field-wise transformed embedding hidden[b, i, :] = e_i W (per_pair == 0: field_all / field_each, left is [B, F, D])
// or the pair-wise one e_i W_p (per_pair == 1: field_interaction, left is [B, P, D]).  The matrix products are
// rbx_linear_fwd; this is the pairing.  One wavefront per sample, its [F, D] blocks in LDS.
__global__ __launch_bounds__(64) void pairmul_fwd_kernel(const float* __restrict__ left, const float* __restrict__ right,
                                                         const int F, const int D, const int per_pair,
                                                         float* __restrict__ out) {
  extern __shared__ float se[];                       // right [F, D], then left [F, D] when it is per field
  const long long b = blockIdx.x;
  const int FD = F * D, P = F * (F - 1) / 2;
  float* sl = se + FD;
  for (int i = threadIdx.x; i < FD; i += 64) {
    se[i] = right[b * FD + i];
    if (!per_pair) sl[i] = left[b * FD + i];
  }
  __syncthreads();
  for (int i = 0; i < F - 1; ++i) {
    const int n = (F - 1 - i) * D;                    // contiguous run of outputs for row i
    const long long o0 = (b * P + pair_index(i, i + 1, F)) * static_cast<long long>(D);
    for (int t = threadIdx.x; t < n; t += 64) {
      const int j = i + 1 + t / D, d = t % D;
      const float l = per_pair ? left[o0 + t] : sl[i * D + d];
      out[o0 + t] = l * se[j * D + d];
    }
  }
}

__global__ __launch_bounds__(64) void pairmul_bwd_kernel(const float* __restrict__ left, const float* __restrict__ right,
                                                         const float* __restrict__ g, const int F, const int D,
                                                         const int per_pair, float* __restrict__ dleft,
                                                         float* __restrict__ dright) {
  extern __shared__ float se[];
  const long long b = blockIdx.x;
  const int FD = F * D, P = F * (F - 1) / 2;
  float* sl = se + FD;
  for (int i = threadIdx.x; i < FD; i += 64) {
    se[i] = right[b * FD + i];
    if (!per_pair) sl[i] = left[b * FD + i];
  }
  __syncthreads();
  const float* gb = g + b * P * static_cast<long long>(D);
  const float* lb = left + b * P * static_cast<long long>(D);   // only read when per_pair
  if (per_pair) {                                      // d left[b, p, :] = g * right[j_p]
    for (int i = 0; i < F - 1; ++i) {
      const int n = (F - 1 - i) * D;
      const long long o0 = pair_index(i, i + 1, F) * static_cast<long long>(D);
      for (int t = threadIdx.x; t < n; t += 64) {
        const int j = i + 1 + t / D, d = t % D;
        dleft[b * P * static_cast<long long>(D) + o0 + t] = gb[o0 + t] * se[j * D + d];
      }
    }
  }
  for (int t = threadIdx.x; t < FD; t += 64) {
    const int x = t / D, d = t % D;
    float as_left = 0.f, as_right = 0.f;
    for (int j = x + 1; j < F; ++j) {                  // x is the left field of pair (x, j)
      const int p = pair_index(x, j, F);
      as_left += gb[p * D + d] * se[j * D + d];
    }
    for (int i = 0; i < x; ++i) {                      // x is the right field of pair (i, x)
      const int p = pair_index(i, x, F);
      as_right += gb[p * D + d] * (per_pair ? lb[p * D + d] : sl[i * D + d]);
    }
    if (!per_pair) dleft[b * FD + t] = as_left;
    dright[b * FD + t] = as_right;
  }
}

template <int G, int NV, bool VEC>
static int launch_fm(bool bwd, const float* emb, const float* dout, int64_t B, int F, int D, int mode, float* out,
                     long long sb, long long dsb, hipStream_t s) {
  const int gpb = 256 / G;
  long long blocks = (B + gpb - 1) / gpb;
  if (blocks > kCUs * 8) blocks = kCUs * 8;
  if (bwd)
    hipLaunchKernelGGL((fm_bwd_kernel<G, NV, VEC>), dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s, emb, sb, dout,
                       static_cast<long long>(B), F, D, mode, out, dsb);
  else
    hipLaunchKernelGGL((fm_fwd_kernel<G, NV, VEC>), dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s, emb, sb,
                       static_cast<long long>(B), F, D, mode, out);
  return check_launch("fm kernel");
}

template <bool VEC>
static int dispatch_fm(bool bwd, const float* emb, const float* dout, int64_t B, int F, int D, int mode, float* out,
                       long long sb, long long dsb, hipStream_t s) {
  const int units = VEC ? D / 4 : D;
  switch (pow2_ceil(units)) {
    case 1: return launch_fm<1, 1, VEC>(bwd, emb, dout, B, F, D, mode, out, sb, dsb, s);
    case 2: return launch_fm<2, 1, VEC>(bwd, emb, dout, B, F, D, mode, out, sb, dsb, s);
    case 4: return launch_fm<4, 1, VEC>(bwd, emb, dout, B, F, D, mode, out, sb, dsb, s);
    case 8: return launch_fm<8, 1, VEC>(bwd, emb, dout, B, F, D, mode, out, sb, dsb, s);
    case 16: return launch_fm<16, 1, VEC>(bwd, emb, dout, B, F, D, mode, out, sb, dsb, s);
    case 32: return launch_fm<32, 1, VEC>(bwd, emb, dout, B, F, D, mode, out, sb, dsb, s);
    case 64: return launch_fm<64, 1, VEC>(bwd, emb, dout, B, F, D, mode, out, sb, dsb, s);
    case 128: return launch_fm<64, 2, VEC>(bwd, emb, dout, B, F, D, mode, out, sb, dsb, s);
    case 256: return launch_fm<64, 4, VEC>(bwd, emb, dout, B, F, D, mode, out, sb, dsb, s);
    default: return fail(RBX_ERR_UNSUPPORTED, "interaction dim too large");
  }
}

static int run_interaction(bool bwd, const float* emb, long long sb, const float* dout, int64_t B, int F, int D, int mode,
                           float* out, long long dsb, void* stream) {
  if (B == 0) return RBX_OK;   // empty batch: nothing to do, pointers may be NULL
  if (emb == nullptr || out == nullptr || (bwd && dout == nullptr)) return fail(RBX_ERR_INVALID, "NULL tensor");
  if (B < 0 || F <= 0 || D <= 0) return fail(RBX_ERR_INVALID, "bad shape B=%lld F=%d D=%d", (long long)B, F, D);
  if (mode < 0 || mode > 3) return fail(RBX_ERR_INVALID, "InnerProductInteraction output mode %d is not supported", mode);
  if (B == 0) return RBX_OK;
  if (sb < static_cast<long long>(F) * D || (bwd && dsb < static_cast<long long>(F) * D))
    return fail(RBX_ERR_INVALID, "interaction: batch stride smaller than n_fields * dim");
  hipStream_t s = as_stream(stream);
  if (mode <= 1) {
    const bool vec = (D % 4 == 0) && ((reinterpret_cast<uintptr_t>(emb) & 15) == 0) &&
                     ((reinterpret_cast<uintptr_t>(out) & 15) == 0) && sb % 4 == 0 && (!bwd || dsb % 4 == 0);
    return vec ? dispatch_fm<true>(bwd, emb, dout, B, F, D, mode, out, sb, dsb, s)
               : dispatch_fm<false>(bwd, emb, dout, B, F, D, mode, out, sb, dsb, s);
  }
  const size_t lds = static_cast<size_t>(F) * D * sizeof(float);
  if (lds > 64 * 1024) return fail(RBX_ERR_UNSUPPORTED, "F*D=%d too large for the pairwise modes", F * D);
  if (F < 2) return RBX_OK;
  if (bwd)
    hipLaunchKernelGGL(pair_bwd_kernel, dim3(static_cast<unsigned>(B)), dim3(64), lds, s, emb, sb, dout, F, D, mode, out, dsb);
  else
    hipLaunchKernelGGL(pair_fwd_kernel, dim3(static_cast<unsigned>(B)), dim3(64), lds, s, emb, sb, F, D, mode, out);
  return check_launch("pair kernel");
}

}  // namespace rbx

namespace rbx {
// product_sum AND the field sum it is made of, one pass over the [B, F, D] block (D a multiple of 4, G = D / 4 lanes per
// sample): y[b] = 0.5 sum_d (S_d^2 - Q_d), S[b, d] = sum_f e[b, f, d].  S is what the backward of the term needs
// (d e[b, f, :] = g_b (S_b - e[b, f, :])): kept, the gradient can be formed wherever the block's gradient is assembled.
template <int G>
__global__ __launch_bounds__(256) void fm_sum_fwd_kernel(const float* __restrict__ emb, const long long sb, const long long B,
                                                         const int F, const int D, float* __restrict__ out,
                                                         float* __restrict__ sum, const float* __restrict__ lr_w,
                                                         const float* __restrict__ lr_b, float* __restrict__ lr_out) {
  const int lane_g = threadIdx.x % G;
  const long long ngroups = static_cast<long long>(gridDim.x) * (blockDim.x / G);
  const int e = lane_g * 4;
  // lr_w != NULL: the first-order Linear over the same F D columns (deepfm.py:37: LR reads the block FM reads) rides in this
  // pass -- y_lr[b] = <x[b, :F D], lr_w> + lr_b -- instead of a logit-head kernel that reads the 436 MB block once more
  const float bias = (lr_w != nullptr && lr_b != nullptr) ? lr_b[0] : 0.f;
  for (long long b = static_cast<long long>(blockIdx.x) * (blockDim.x / G) + threadIdx.x / G; b < B; b += ngroups) {
    const float* base = emb + b * sb;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f), q = make_float4(0.f, 0.f, 0.f, 0.f);
    float lr = 0.f;
    if (e < D) {
      if (lr_w != nullptr) {
#pragma unroll 8
        for (int f = 0; f < F; ++f) {
          const float4 t = *reinterpret_cast<const float4*>(base + f * D + e);
          const float4 w = *reinterpret_cast<const float4*>(lr_w + f * D + e);
          s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
          q.x += t.x * t.x; q.y += t.y * t.y; q.z += t.z * t.z; q.w += t.w * t.w;
          lr += (t.x * w.x + t.y * w.y) + (t.z * w.z + t.w * w.w);
        }
      } else {
#pragma unroll 8
        for (int f = 0; f < F; ++f) {
          const float4 t = *reinterpret_cast<const float4*>(base + f * D + e);
          s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
          q.x += t.x * t.x; q.y += t.y * t.y; q.z += t.z * t.z; q.w += t.w * t.w;
        }
      }
      *reinterpret_cast<float4*>(sum + b * D + e) = s;
    }
    float t = 0.5f * ((s.x * s.x - q.x) + (s.y * s.y - q.y) + (s.z * s.z - q.z) + (s.w * s.w - q.w));
    t = group_sum<G>(t);
    if (lr_w != nullptr) lr = group_sum<G>(lr);
    if (lane_g == 0) {
      out[b] = t;
      if (lr_w != nullptr) lr_out[b] = lr + bias;
    }
  }
}
}  // namespace rbx

extern "C" int rbx_fm_sum_fwd(const float* d_emb, int64_t emb_stride_b, int64_t batch, int32_t n_fields, int32_t dim,
                              float* d_out, float* d_sum, void* stream) {
  return rbx_fm_sum_lr_fwd(d_emb, emb_stride_b, batch, n_fields, dim, d_out, d_sum, nullptr, nullptr, nullptr, stream);
}

extern "C" int rbx_fm_sum_lr_fwd(const float* d_emb, int64_t emb_stride_b, int64_t batch, int32_t n_fields, int32_t dim,
                                 float* d_out, float* d_sum, const float* d_lr_w, const float* d_lr_b, float* d_lr_out,
                                 void* stream) {
  using namespace rbx;
  if (batch == 0) return RBX_OK;
  if (!d_emb || !d_out || !d_sum) return fail(RBX_ERR_INVALID, "fm_sum: NULL tensor");
  if (d_lr_w != nullptr && (d_lr_out == nullptr || (reinterpret_cast<uintptr_t>(d_lr_w) & 15) != 0))
    return fail(RBX_ERR_INVALID, "fm_sum: the first-order weights need an output and a 16-byte aligned base");
  if (batch < 0 || n_fields <= 0 || dim <= 0 || emb_stride_b < static_cast<int64_t>(n_fields) * dim)
    return fail(RBX_ERR_INVALID, "fm_sum: bad shape");
  if (dim % 4 != 0 || dim > 256 || emb_stride_b % 4 != 0 ||
      ((reinterpret_cast<uintptr_t>(d_emb) | reinterpret_cast<uintptr_t>(d_sum)) & 15) != 0)
    return fail(RBX_ERR_UNSUPPORTED, "fm_sum: needs 16-byte aligned rows and dim %% 4 == 0, dim <= 256 (got dim %d)", dim);
  const int g = pow2_ceil(dim / 4);
  const int gpb = 256 / g;
  long long blocks = (batch + gpb - 1) / gpb;
  if (blocks > kCUs * 8) blocks = kCUs * 8;
  hipStream_t s = as_stream(stream);
#define CALL(GG) hipLaunchKernelGGL((fm_sum_fwd_kernel<GG>), dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s, d_emb, \
                                    static_cast<long long>(emb_stride_b), static_cast<long long>(batch), n_fields, dim, d_out, d_sum, \
                                    d_lr_w, d_lr_b, d_lr_out)
  switch (g) {
    case 1: CALL(1); break;
    case 2: CALL(2); break;
    case 4: CALL(4); break;
    case 8: CALL(8); break;
    case 16: CALL(16); break;
    case 32: CALL(32); break;
    default: CALL(64); break;
  }
#undef CALL
  return check_launch("fm_sum_fwd_kernel");
}

extern "C" int rbx_interaction_fwd(const float* d_emb, int64_t emb_stride_b, int64_t batch, int32_t n_fields, int32_t dim,
                                   int32_t mode, float* d_out, void* stream) {
  return rbx::run_interaction(false, d_emb, emb_stride_b, nullptr, batch, n_fields, dim, mode, d_out, 0, stream);
}

extern "C" int rbx_interaction_bwd(const float* d_emb, int64_t emb_stride_b, const float* d_dout, int64_t batch,
                                   int32_t n_fields, int32_t dim, int32_t mode, float* d_demb, int64_t demb_stride_b,
                                   void* stream) {
  return rbx::run_interaction(true, d_emb, emb_stride_b, d_dout, batch, n_fields, dim, mode, d_demb, demb_stride_b, stream);
}

extern "C" int rbx_pairmul_fwd(const float* d_left, const float* d_right, int64_t batch, int32_t n_fields, int32_t dim,
                               int32_t per_pair, float* d_out, void* stream) {
  using namespace rbx;
  if (batch == 0) return RBX_OK;
  if (batch < 0 || n_fields <= 0 || dim <= 0) return fail(RBX_ERR_INVALID, "pairmul: bad shape");
  if (!d_left || !d_right || !d_out) return fail(RBX_ERR_INVALID, "pairmul: NULL tensor");
  if (n_fields < 2) return RBX_OK;
  const size_t lds = static_cast<size_t>(2) * n_fields * dim * sizeof(float);
  if (lds > 64 * 1024) return fail(RBX_ERR_UNSUPPORTED, "pairmul: F*D=%d too large", n_fields * dim);
  hipLaunchKernelGGL(pairmul_fwd_kernel, dim3(static_cast<unsigned>(batch)), dim3(64), lds, as_stream(stream), d_left, d_right,
                     n_fields, dim, per_pair, d_out);
  return check_launch("pairmul_fwd_kernel");
}

extern "C" int rbx_pairmul_bwd(const float* d_left, const float* d_right, const float* d_dout, int64_t batch, int32_t n_fields,
                               int32_t dim, int32_t per_pair, float* d_dleft, float* d_dright, void* stream) {
  using namespace rbx;
  if (batch == 0) return RBX_OK;
  if (batch < 0 || n_fields <= 0 || dim <= 0) return fail(RBX_ERR_INVALID, "pairmul_bwd: bad shape");
  if (!d_left || !d_right || !d_dout || !d_dleft || !d_dright) return fail(RBX_ERR_INVALID, "pairmul_bwd: NULL tensor");
  const size_t lds = static_cast<size_t>(2) * n_fields * dim * sizeof(float);
  if (lds > 64 * 1024) return fail(RBX_ERR_UNSUPPORTED, "pairmul_bwd: F*D=%d too large", n_fields * dim);
  hipLaunchKernelGGL(pairmul_bwd_kernel, dim3(static_cast<unsigned>(batch)), dim3(64), lds, as_stream(stream), d_left, d_right,
                     d_dout, n_fields, dim, per_pair, d_dleft, d_dright);
  return check_launch("pairmul_bwd_kernel");
}
