// rbx_fm_quad.hip -- the forward of the fused FM body (rbx_fm_fwd) for the wire format the reference's ranking loader
// produces: every feature's id / value a column of ONE row-major batch tensor (any of the four id dtypes), dim 16 (gfx950).
//
// Reference op sequence replaced (paths relative to /root/reference/recbox), as rbx_fm_fused.hip:
//   FeatureEmbedding(X)               ranking/pytorch/layers/embeddings/feature_embedding.py:188-214
//   LogisticRegression(X)             ranking/pytorch/layers/blocks/logistic_regression.py:30-35
//   InnerProductInteraction           ranking/pytorch/layers/interactions/inner_product.py:41-48
//   lr_out + fm_out                   ranking/pytorch/layers/blocks/factorization_machine.py:30-34
//
// What bounds this gather on MI355X (profiles/r05/fm_fwd_limiter.md: counters of the round-4 kernel, and a laboratory of
// forms, profiles/ubench/fm_fwd_lab.hip): not HBM bytes -- the vector memory path of a CU.  Every lookup that misses L1 holds
// one of the CU's ~64 miss slots for the L2 / fabric latency (the round-4 kernel: 0.113 requests per cycle per CU x 547
// cycles = 62 in flight), every lookup at all costs a tag cycle, and every wave instruction issue slots: with all lookups
// hitting L1 the round-4 form still takes 16-19 us of its 35.  So this form spends as few instructions and lookups per
// sample as the wire format allows:
//   * a sample = 4 lanes (float4 each), 16 samples per wavefront; lane j of a group reads the sample's columns j, 4 + j,
//     8 + j, ...: a step of a group is 32 contiguous bytes, the NI steps of a sample its whole batch row, each line fetched
//     once (the round-4 form: 39 loads per lane group, every lane of a group the same address);
//   * the OWNER lane decodes its columns (range check, byte offset of the row from ONE base pointer `arena`, the lowest
//     address of the call's tables: 4 x fewer decodes than all lanes decoding every feature) and loads their LR weights;
//   * the feature loop needs no per-feature state: a DPP quad broadcast hands the owner's row offset to the group inside the
//     address add; UB rows (float4) in flight per lane, features summed in feature order (S, logit: the same floating-point
//     operations in the same order as fm_fused_fwd_kernel).
// Out-of-range ids read as zero rows (x = 0 on a valid row of the same table) and raise the status word, as there.
#include <stdlib.h>
#include "rbx_internal.h"

namespace rbx {

struct QuadMeta {            // first kernel argument: read through the kernarg segment pointer into LDS
  int voc[RBX_MAX_FIELDS];              // > 0: rows of a categorical column's table; 0: numeric column; -1: no such column
  unsigned eo[RBX_MAX_FIELDS];          // byte offset of the table / weight vector from arena
  unsigned es[RBX_MAX_FIELDS];          // byte stride of its rows (0 for numeric / padding columns)
  unsigned lo[RBX_MAX_FIELDS];          // the same for the first-order weights
  unsigned ls[RBX_MAX_FIELDS];
};
static_assert(sizeof(QuadMeta) == 5 * 4 * RBX_MAX_FIELDS && RBX_MAX_FIELDS == 64, "QuadMeta layout is read by index");

template <int J>
__device__ __forceinline__ unsigned qbcast(unsigned v) {         // lane J of every quad -> the quad
  return static_cast<unsigned>(__builtin_amdgcn_mov_dpp(static_cast<int>(v), J | (J << 2) | (J << 4) | (J << 6), 0xF, 0xF, true));
}
template <int J>
__device__ __forceinline__ float qbcastf(float v) { return __uint_as_float(qbcast<J>(__float_as_uint(v))); }

template <int F0, int U, int N>
__device__ __forceinline__ void quad_issue(const unsigned (&o)[16], const unsigned lane16, const char* __restrict__ arena,
                                           float4 (&e)[N]) {
  if constexpr (U < N) {
    constexpr int f = F0 + U;
    const unsigned off = qbcast<(f & 3)>(o[f >> 2]) + lane16;
    e[U] = *reinterpret_cast<const float4*>(arena + static_cast<size_t>(off));
    quad_issue<F0, U + 1, N>(o, lane16, arena, e);
  }
}
template <int F0, int U, int N>
__device__ __forceinline__ void quad_use(const float (&x)[16], const float4 (&e)[N], float (&s)[4], float (&q)[4]) {
  if constexpr (U < N) {
    constexpr int f = F0 + U;
    const float xb = qbcastf<(f & 3)>(x[f >> 2]);
    // explicitly rounded, as in fm_fused_fwd_kernel: the two kernels agree bit for bit
    const float t0 = mul_rn(e[U].x, xb), t1 = mul_rn(e[U].y, xb), t2 = mul_rn(e[U].z, xb), t3 = mul_rn(e[U].w, xb);
    s[0] = add_rn(s[0], t0); q[0] = fma_rn(t0, t0, q[0]);
    s[1] = add_rn(s[1], t1); q[1] = fma_rn(t1, t1, q[1]);
    s[2] = add_rn(s[2], t2); q[2] = fma_rn(t2, t2, q[2]);
    s[3] = add_rn(s[3], t3); q[3] = fma_rn(t3, t3, q[3]);
    quad_use<F0, U + 1, N>(x, e, s, q);
  }
}
template <int F0, int NF, int UB>
__device__ __forceinline__ void quad_batches(const unsigned (&o)[16], const float (&x)[16], const unsigned lane16,
                                             const char* __restrict__ arena, float (&s)[4], float (&q)[4]) {
  if constexpr (F0 < NF) {
    constexpr int N = (NF - F0 < UB) ? NF - F0 : UB;
    float4 e[N];
    quad_issue<F0, 0, N>(o, lane16, arena, e);
    quad_use<F0, 0, N>(x, e, s, q);
    __builtin_amdgcn_sched_barrier(0);          // (or the scheduler hoists the next batch's loads: registers, spills)
    quad_batches<F0 + N, NF, UB>(o, x, lane16, arena, s, q);
  }
}

// NI = column slots per lane (4 NI >= F), UB = rows in flight per lane, DT = the ids dtype of every column.
// 20 rows in flight at 3 wavefronts per SIMD is the measured optimum at the Criteo shape (profiles/r05/fm_fwd_lab.txt:
// 29.5 us; 10 / 14 in flight at 4 per SIMD 31.9 / 31.5, 8 at 5 per SIMD 41.0).
template <int NI, int UB, int DT>
__global__ __launch_bounds__(256, 3) void fm_quad_fwd_kernel(const QuadMeta M, const char* __restrict__ arena,
                                                            const void* __restrict__ X, const long long ldx, const int F,
                                                            const long long B, const float* __restrict__ bias,
                                                            float* __restrict__ logit, float* __restrict__ prob,
                                                            float* __restrict__ ssum, int* __restrict__ status) {
  __shared__ int s_voc[RBX_MAX_FIELDS];
  __shared__ unsigned s_eo[RBX_MAX_FIELDS], s_es[RBX_MAX_FIELDS], s_lo[RBX_MAX_FIELDS], s_ls[RBX_MAX_FIELDS];
  if (threadIdx.x < RBX_MAX_FIELDS) {
    // M is the first kernel argument: its bytes start the kernarg segment (indexing the by-value struct with a lane-varying
    // index would make the compiler copy it to scratch)
    const unsigned* km = reinterpret_cast<const unsigned*>(
        reinterpret_cast<uintptr_t>(__builtin_amdgcn_kernarg_segment_ptr()));
    s_voc[threadIdx.x] = static_cast<int>(km[threadIdx.x]);
    s_eo[threadIdx.x] = km[RBX_MAX_FIELDS + threadIdx.x];
    s_es[threadIdx.x] = km[2 * RBX_MAX_FIELDS + threadIdx.x];
    s_lo[threadIdx.x] = km[3 * RBX_MAX_FIELDS + threadIdx.x];
    s_ls[threadIdx.x] = km[4 * RBX_MAX_FIELDS + threadIdx.x];
  }
  __syncthreads();
  (void)M;
  const int lane_g = threadIdx.x & 3;
  const unsigned lane16 = lane_g * 16;
  const long long ngroups = static_cast<long long>(gridDim.x) * 64;
  for (long long b = static_cast<long long>(blockIdx.x) * 64 + threadIdx.x / 4; b < B; b += ngroups) {
    long long raw[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int col = 4 * i + lane_g;
      raw[i] = fm_load_raw<DT>(X, b * ldx + (col < F ? col : F - 1));
    }
    unsigned o[16];
    float x[16], l1[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int col = 4 * i + lane_g;
      const int voc = s_voc[col];
      int v = 0;
      x[i] = 0.f;
      if (voc > 0) {
        if (fm_decode_id<DT>(raw[i], voc, &v)) x[i] = 1.f;
        else { v = 0; if (status != nullptr) atomicOr(status, 1); }       // out of range: reads as a zero row
      } else if (voc == 0) {
        x[i] = fm_decode_value<DT>(raw[i]);
      }
      o[i] = s_eo[col] + static_cast<unsigned>(v) * s_es[col];
      l1[i] = *reinterpret_cast<const float*>(arena + static_cast<size_t>(s_lo[col] + static_cast<unsigned>(v) * s_ls[col]));
    }
    float s[4] = {0.f, 0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f};
    quad_batches<0, 4 * NI, UB>(o, x, lane16, arena, s, q);
    float lr = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) lr = fma_rn(l1[i], x[i], lr);
    float fm = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) fm = fma_rn(fma_rn(s[i], s[i], -q[i]), 0.5f, fm);
    const float total = group_sum<4>(add_rn(fm, lr));
    if (lane_g == 0) {
      const float z = total + (bias != nullptr ? bias[0] : 0.f);
      logit[b] = z;
      if (prob != nullptr) prob[b] = 1.f / (1.f + expf(-z));
    }
    if (ssum != nullptr) *reinterpret_cast<float4*>(ssum + b * 16 + lane_g * 4) = make_float4(s[0], s[1], s[2], s[3]);
  }
}

static int g_fm_quad = -1;                  // -1: read RBX_FM_QUAD on first use (0 = the general kernel for every call)
static bool fm_quad_on() {
  if (g_fm_quad < 0) { const char* e = getenv("RBX_FM_QUAD"); g_fm_quad = (e == nullptr || e[0] != '0') ? 1 : 0; }
  return g_fm_quad != 0;
}

template <int NI, int DT>
static void quad_launch(const QuadMeta& m, const char* arena, const void* X, long long ldx, int F, long long B,
                        const float* bias, float* logit, float* prob, float* ssum, int* status, hipStream_t s) {
  long long blocks = (B + 63) / 64;
  if (blocks > kCUs * 16) blocks = kCUs * 16;
  constexpr int UB = NI > 13 ? 16 : 20;       // (64 columns: 20 float4 in flight no longer fit 168 registers)
  hipLaunchKernelGGL((fm_quad_fwd_kernel<NI, UB, DT>), dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s, m, arena, X, ldx,
                     F, B, bias, logit, prob, ssum, status);
}

template <int DT>
static void quad_dispatch(int ni, const QuadMeta& m, const char* arena, const void* X, long long ldx, int F, long long B,
                          const float* bias, float* logit, float* prob, float* ssum, int* status, hipStream_t s) {
  if (ni <= 4) quad_launch<4, DT>(m, arena, X, ldx, F, B, bias, logit, prob, ssum, status, s);
  else if (ni <= 7) quad_launch<7, DT>(m, arena, X, ldx, F, B, bias, logit, prob, ssum, status, s);
  else if (ni <= 10) quad_launch<10, DT>(m, arena, X, ldx, F, B, bias, logit, prob, ssum, status, s);
  else if (ni <= 13) quad_launch<13, DT>(m, arena, X, ldx, F, B, bias, logit, prob, ssum, status, s);
  else quad_launch<16, DT>(m, arena, X, ldx, F, B, bias, logit, prob, ssum, status, s);
}

// RBX_OK: launched.  1: the call does not have the shape this kernel is written for (the caller launches the general one).
int fm_quad_fwd(const FmPack& pack, int F, int D, bool has_emb, bool has_lr, int uniform_dt, long long B, const float* bias,
                float* logit, float* prob, float* ssum, int* status, hipStream_t s) {
  if (!fm_quad_on() || D != 16 || !has_emb || !has_lr || uniform_dt < 0 || F < 1 || F > RBX_MAX_FIELDS) return 1;
  if (ssum != nullptr && (reinterpret_cast<uintptr_t>(ssum) & 15) != 0) return 1;
  const size_t esz = (uniform_dt == RBX_I64 || uniform_dt == RBX_F64) ? 8 : 4;
  // the columns of ONE row-major matrix, in feature order
  const char* x0 = static_cast<const char*>(pack.f[0].ids);
  const long long ldx = pack.f[0].stride_b;
  if (ldx < F || (reinterpret_cast<uintptr_t>(x0) & (esz - 1)) != 0) return 1;
  uintptr_t lo = UINTPTR_MAX, hi = 0;
  for (int f = 0; f < F; ++f) {
    const FmField& k = pack.f[f];
    if (static_cast<const char*>(k.ids) != x0 + static_cast<size_t>(f) * esz || k.stride_b != ldx) return 1;
    if (k.emb == nullptr || k.lr == nullptr || (reinterpret_cast<uintptr_t>(k.emb) & 15) != 0 || k.emb_stride % 4 != 0) return 1;
    const size_t rows = k.kind == RBX_FIELD_CATEGORICAL ? static_cast<size_t>(k.vocab) : 1;
    const uintptr_t e0 = reinterpret_cast<uintptr_t>(k.emb), l0 = reinterpret_cast<uintptr_t>(k.lr);
    const uintptr_t e1 = e0 + ((rows - 1) * static_cast<size_t>(k.emb_stride) + 16) * 4;
    const uintptr_t l1 = l0 + ((rows - 1) * static_cast<size_t>(k.lr_stride) + 1) * 4;
    lo = e0 < lo ? e0 : lo; lo = l0 < lo ? l0 : lo;
    hi = e1 > hi ? e1 : hi; hi = l1 > hi ? l1 : hi;
  }
  if (hi - lo >= (1ull << 32)) return 1;                 // 32-bit byte offsets from one base pointer
  QuadMeta m;
  for (int c = 0; c < RBX_MAX_FIELDS; ++c) {
    const FmField& k = pack.f[c < F ? c : F - 1];        // a column past F: row 0 of the last table, scaled by x = 0
    const bool cat = k.kind == RBX_FIELD_CATEGORICAL;
    m.voc[c] = c < F ? (cat ? k.vocab : 0) : -1;
    m.eo[c] = static_cast<unsigned>(reinterpret_cast<uintptr_t>(k.emb) - lo);
    m.lo[c] = static_cast<unsigned>(reinterpret_cast<uintptr_t>(k.lr) - lo);
    m.es[c] = (c < F && cat) ? static_cast<unsigned>(k.emb_stride) * 4u : 0u;
    m.ls[c] = (c < F && cat) ? static_cast<unsigned>(k.lr_stride) * 4u : 0u;
  }
  const char* arena = reinterpret_cast<const char*>(lo);
  const int ni = (F + 3) / 4;
  switch (uniform_dt) {
    case RBX_I32: quad_dispatch<RBX_I32>(ni, m, arena, x0, ldx, F, B, bias, logit, prob, ssum, status, s); break;
    case RBX_I64: quad_dispatch<RBX_I64>(ni, m, arena, x0, ldx, F, B, bias, logit, prob, ssum, status, s); break;
    case RBX_F32: quad_dispatch<RBX_F32>(ni, m, arena, x0, ldx, F, B, bias, logit, prob, ssum, status, s); break;
    default: quad_dispatch<RBX_F64>(ni, m, arena, x0, ldx, F, B, bias, logit, prob, ssum, status, s); break;
  }
  return check_launch("fm_quad_fwd_kernel");
}

}  // namespace rbx

extern "C" int rbx_fm_quad(int32_t enable) {
  using namespace rbx;
  const int was = fm_quad_on() ? 1 : 0;
  if (enable >= 0) g_fm_quad = enable ? 1 : 0;
  return was;
}
