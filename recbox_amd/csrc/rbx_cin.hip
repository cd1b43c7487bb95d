// rbx_cin.hip -- the outer-product tensor of xDeepFM's Compressed Interaction Network (gfx950):
//   Z[b, (h, m), d] = X_0[b, h, d] * X_k[b, m, d]      ranking/pytorch/layers/interactions/compressed_interaction_net.py:35-48
//                                                      (torch.einsum("bhd,bmd->bhmd") + view, ahead of the 1x1 Conv1d)
// (path relative to /root/reference/recbox).  The 1x1 convolution that consumes Z runs as ONE GEMM over the channel axis for
// every (b, d) (rbx_linear_fwd), so Z is produced in that GEMM's A layout: z[(b, d), h * M + m], rows = B * D.  X_0 is read in
// the embedding layer's own [B, F, D] layout; X_k in the layout the previous layer's GEMM left it, [(b, d), M] -- or, for the
// first layer (X_k = X_0), from X_0 itself.  One workgroup per sample; the backward stages one row of dZ at a time in LDS.
#include "rbx_internal.h"

namespace rbx {

__global__ __launch_bounds__(256) void cin_outer_fwd_kernel(const float* __restrict__ x0, const float* __restrict__ xk,
                                                            const int F, const int M, const int D, float* __restrict__ z) {
  extern __shared__ float cin_lds[];
  float* s0 = cin_lds;              // [D][F]
  float* sk = cin_lds + D * F;      // [D][M]
  const long long b = blockIdx.x;
  for (int i = threadIdx.x; i < F * D; i += blockDim.x) {
    const int h = i / D, d = i % D;
    s0[d * F + h] = x0[b * F * D + i];
  }
  if (xk != nullptr) {
    for (int i = threadIdx.x; i < D * M; i += blockDim.x) sk[i] = xk[b * D * M + i];
  } else {
    for (int i = threadIdx.x; i < D * F; i += blockDim.x) sk[i] = 0.f;      // (filled below from s0: M == F)
  }
  __syncthreads();
  const float* k = xk != nullptr ? sk : s0;
  const int FM = F * M;
  float* out = z + b * D * FM;
  for (int e = threadIdx.x; e < D * FM; e += blockDim.x) {
    const int d = e / FM, c = e - d * FM;
    const int h = c / M, m = c - h * M;
    out[e] = s0[d * F + h] * k[d * M + m];
  }
}

// dx0[b, h, d] = sum_m dz[(b, d), h M + m] xk[(b, d), m]   (+ sum_h' dz[(b, d), h' M + h] x0[b, h', d] when X_k = X_0)
// dxk[(b, d), m] = sum_h dz[(b, d), h M + m] x0[b, h, d]
__global__ __launch_bounds__(256) void cin_outer_bwd_kernel(const float* __restrict__ x0, const float* __restrict__ xk,
                                                            const float* __restrict__ dz, const int F, const int M,
                                                            const int D, float* __restrict__ dx0, float* __restrict__ dxk) {
  extern __shared__ float cin_lds[];
  float* s0 = cin_lds;              // [D][F]
  float* sk = s0 + D * F;           // [D][M]
  float* srow = sk + D * M;         // [F * M]: one row of dz
  float* sacc = srow + F * M;       // [F][D] accumulated dx0 of this sample
  const long long b = blockIdx.x;
  for (int i = threadIdx.x; i < F * D; i += blockDim.x) {
    const int h = i / D, d = i % D;
    s0[d * F + h] = x0[b * F * D + i];
    sacc[i] = 0.f;
  }
  if (xk != nullptr)
    for (int i = threadIdx.x; i < D * M; i += blockDim.x) sk[i] = xk[b * D * M + i];
  __syncthreads();
  const float* k = xk != nullptr ? sk : s0;
  const int FM = F * M;
  for (int d = 0; d < D; ++d) {
    const float* row = dz + (b * D + d) * FM;
    for (int i = threadIdx.x; i < FM; i += blockDim.x) srow[i] = row[i];
    __syncthreads();
    for (int t = threadIdx.x; t < F + M; t += blockDim.x) {
      if (t < F) {
        const int h = t;
        float a = 0.f;
        for (int m = 0; m < M; ++m) a += srow[h * M + m] * k[d * M + m];
        sacc[h * D + d] += a;
      } else {
        const int m = t - F;
        float a = 0.f;
        for (int h = 0; h < F; ++h) a += srow[h * M + m] * s0[d * F + h];
        if (xk != nullptr) { if (dxk != nullptr) dxk[(b * D + d) * M + m] = a; }
        else sk[d * M + m] = a;                  // X_k = X_0: joins dx0 below (M == F)
      }
    }
    __syncthreads();
  }
  if (dx0 != nullptr)
    for (int i = threadIdx.x; i < F * D; i += blockDim.x) {
      const int h = i / D, d = i % D;
      dx0[b * F * D + i] = sacc[i] + (xk == nullptr ? sk[d * M + h] : 0.f);
    }
}

static size_t cin_lds_bytes(int F, int M, int D, bool bwd) {
  size_t n = static_cast<size_t>(D) * F + static_cast<size_t>(D) * M;
  if (bwd) n += static_cast<size_t>(F) * M + static_cast<size_t>(F) * D;
  return n * sizeof(float);
}

static int cin_check(const float* x0, long long B, int F, int M, int D, bool own, bool bwd) {
  if (B < 0 || F <= 0 || M <= 0 || D <= 0) return fail(RBX_ERR_INVALID, "cin_outer: bad shape B=%lld F=%d M=%d D=%d", B, F, M, D);
  if (own && M != F) return fail(RBX_ERR_INVALID, "cin_outer: X_k = X_0 needs M == F (got %d, %d)", M, F);
  if (B > 0 && x0 == nullptr) return fail(RBX_ERR_INVALID, "cin_outer: d_x0 is NULL");
  if (cin_lds_bytes(F, M, D, bwd) > 160 * 1024)
    return fail(RBX_ERR_UNSUPPORTED, "cin_outer: F=%d M=%d D=%d does not fit a workgroup's LDS", F, M, D);
  return RBX_OK;
}

}  // namespace rbx

extern "C" int rbx_cin_outer_fwd(const float* d_x0, const float* d_xk, int64_t batch, int32_t n_fields, int32_t m, int32_t dim,
                                 float* d_z, void* stream) {
  using namespace rbx;
  int rc = cin_check(d_x0, batch, n_fields, m, dim, d_xk == nullptr, false);
  if (rc != RBX_OK) return rc;
  if (batch == 0) return RBX_OK;
  if (d_z == nullptr) return fail(RBX_ERR_INVALID, "cin_outer_fwd: d_z is NULL");
  const size_t lds = cin_lds_bytes(n_fields, m, dim, false);
  if (lds > 64 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(cin_outer_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                          static_cast<int>(lds)) != hipSuccess)
    return fail(RBX_ERR_LAUNCH, "cin_outer_fwd: cannot reserve %zu bytes of LDS", lds);
  hipLaunchKernelGGL(cin_outer_fwd_kernel, dim3(static_cast<unsigned>(batch)), dim3(256), lds, as_stream(stream), d_x0, d_xk,
                     n_fields, m, dim, d_z);
  return check_launch("cin_outer_fwd_kernel");
}

extern "C" int rbx_cin_outer_bwd(const float* d_x0, const float* d_xk, const float* d_dz, int64_t batch, int32_t n_fields,
                                 int32_t m, int32_t dim, float* d_dx0, float* d_dxk, void* stream) {
  using namespace rbx;
  int rc = cin_check(d_x0, batch, n_fields, m, dim, d_xk == nullptr, true);
  if (rc != RBX_OK) return rc;
  if (batch == 0) return RBX_OK;
  if (d_dz == nullptr) return fail(RBX_ERR_INVALID, "cin_outer_bwd: d_dz is NULL");
  const size_t lds = cin_lds_bytes(n_fields, m, dim, true);
  if (lds > 64 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(cin_outer_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                          static_cast<int>(lds)) != hipSuccess)
    return fail(RBX_ERR_LAUNCH, "cin_outer_bwd: cannot reserve %zu bytes of LDS", lds);
  hipLaunchKernelGGL(cin_outer_bwd_kernel, dim3(static_cast<unsigned>(batch)), dim3(256), lds, as_stream(stream), d_x0, d_xk,
                     d_dz, n_fields, m, dim, d_dx0, d_dxk);
  return check_launch("cin_outer_bwd_kernel");
}
