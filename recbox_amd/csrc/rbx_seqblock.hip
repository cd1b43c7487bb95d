// rbx_seqblock.hip -- the row-local chains of a SASRec block as single passes over [B L, 64] (gfx950).
//
// Reference behaviour replaced: third_party/rechub/models/matching/sasrec.py:81-94 (one block of seq_forward) and
// :110-124 (PointWiseFeedForward):
//     Q = attention_layernorm(seqs); mha = MultiheadAttention(Q, seqs, seqs); seqs = Q + mha
//     seqs = forward_layernorm(seqs); seqs = seqs + conv2(relu(conv1(seqs))); seqs *= ~timeline_mask
// Everything except the attention itself is local to a ROW of the [B L, 64] activation: LayerNorm, the 64 -> 64
// projections, the residual adds, the ReLU, the timeline mask.  As separate launches (LayerNorm, one slab GEMM per
// projection, rbx_dense.hip) each of them is a pass over 210 MB at cfg 5 that runs at the copy ceiling -- the NUMBER of
// passes was the cost (profiles/r04/sasrec_kernel_stats.txt: 10 LayerNorm passes, 18 GEMM passes, 12 dW passes per step).
// Here a wavefront owns 32-row slabs and carries a slab through the whole chain in registers:
//
//   * Row layout "pi": lane (m, h) = (lane % 32, lane / 32) holds 32 floats of row m, register r <-> column
//     pi(r, h) = 32 (r >> 4) + 8 ((r >> 2) & 3) + 4 h + (r & 3).  This is exactly how v_mfma_f32_32x32x2_f32 hands back
//     the TRANSPOSED product Y^T = W X^T (A operand = weights, B operand = the slab): output register r of tile t is
//     Y[m, 32 t + (r & 3) + 8 (r >> 2) + 4 h].  The k index of an MFMA step is free as long as both operands agree, so a
//     slab in layout pi is a valid B operand of the NEXT product without any shuffle: a chain of GEMMs never leaves the
//     registers.  Inputs are brought into pi by the LDS turn that the coalesced slab loads need anyway (rbx_dense.hip's
//     slab requests: eight 1 KB requests per slab), outputs leave through the same turn backwards.
//   * Weights sit in LDS once per workgroup in the order the MFMA steps read them (one ds_read_b128 = four steps of one
//     output tile, conflict-free); the same image serves the transposed product (row layout out) and the natural one
//     (column layout out: the operands the weight-gradient products want).
//   * LayerNorm statistics of a row: 32 adds in the lane, one exchange with lane ^ 32.
//
// Forward kernels (this file): sb_qkv_fwd (LayerNorm + the three in-projections: 1 read, 4 writes instead of
// 3 reads + 4 writes in three launches) and sb_ffn_fwd (out-projection + residual + LayerNorm + conv1 + ReLU + conv2 +
// residual + mask: 2 reads, 4 writes instead of 7 reads + 5 writes in four launches).
#include <stdlib.h>
#include "rbx_internal.h"

namespace rbx {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kSbWaves = 8;                 // wavefronts per workgroup (independent of each other after the staging)
constexpr int kSbLd = 64 + 4;               // LDS row pitch of a slab (floats)
constexpr int kSbSlab = 32 * kSbLd;         // floats of a wavefront's slab
constexpr int kSbW = 64 * 64;               // floats of a staged weight

// Weight image for MFMA operand reads.  The product is Y[m, out] = sum_in X[m, in] Wmat[out][in] with
// Wmat[out][in] = trans ? W[in * ld + out] : W[out * ld + in].  Image word ((t * 8 + g) * 64 + lane) * 4 + e holds
// Wmat[32 t + lane % 32][pi(4 g + e, lane / 32)]: what lane feeds to step 4 g + e of output tile t.
__device__ __forceinline__ void sb_stage_weight(float* __restrict__ wl, const float* __restrict__ W, const long long ld,
                                                const bool trans) {
  for (int idx = threadIdx.x; idx < kSbW; idx += static_cast<int>(blockDim.x)) {
    const int e = idx & 3, lane = (idx >> 2) & 63, g = (idx >> 8) & 7, t = idx >> 11;
    const int out = 32 * t + (lane & 31);
    const int in = 32 * (g >> 2) + 8 * (g & 3) + 4 * (lane >> 5) + e;
    wl[idx] = trans ? W[static_cast<long long>(in) * ld + out] : W[static_cast<long long>(out) * ld + in];
  }
}
__device__ __forceinline__ void sb_stage_vec(float* __restrict__ vl, const float* __restrict__ v, const float fill) {
  if (threadIdx.x < 64) vl[threadIdx.x] = v != nullptr ? v[threadIdx.x] : fill;
}
// a lane's 32 entries of a 64-vector in layout pi
__device__ __forceinline__ void sb_vec(const float* __restrict__ vl, const int h, float (&o)[32]) {
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    const f32x4 u = *reinterpret_cast<const f32x4*>(vl + 32 * (g >> 2) + 8 * (g & 3) + 4 * h);
    o[4 * g] = u[0]; o[4 * g + 1] = u[1]; o[4 * g + 2] = u[2]; o[4 * g + 3] = u[3];
  }
}

// ---- slab requests (coalesced: request p covers rows 4 p .. 4 p + 3, lane l floats [4 (l % 16), + 4) of row 4 p + l / 16) ----
__device__ __forceinline__ void sb_offsets(const long long ld, const int rows_left, const int lane, unsigned (&off)[8]) {
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    int row = 4 * p + (lane >> 4);
    row = row < rows_left ? row : rows_left - 1;
    off[p] = static_cast<unsigned>((static_cast<long long>(row) * ld + 4 * (lane & 15)) * 4);
  }
}
__device__ __forceinline__ void sb_issue(const float* base, const unsigned (&off)[8], f32x4 (&v)[8]) {
#pragma unroll
  for (int p = 0; p < 8; ++p) asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=v"(v[p]) : "v"(off[p]), "s"(base));
}
// The same request with the accumulation registers as its destination (gfx950 loads into AGPRs directly): a kernel that
// lives on the whole 512-register file parks values in AGPRs on its own, and the compiler -- which does not know that an
// inline-asm load is still in flight -- was seen copying the freshly "defined" VGPRs there right behind the request
// (profiles/scripts/check_inflight.py).  Registers that start out as AGPRs are left where they are until they are used.
__device__ __forceinline__ void sb_issue_a(const float* base, const unsigned (&off)[8], f32x4 (&v)[8]) {
#pragma unroll
  for (int p = 0; p < 8; ++p) asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=a"(v[p]) : "v"(off[p]), "s"(base));
}
__device__ __forceinline__ void sb_arrived(f32x4 (&v)[8]) {
  asm volatile("s_waitcnt vmcnt(0)"
               : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7])
               :
               : "memory");
}
__device__ __forceinline__ void sb_issue_word(const float* base, const unsigned off, float& v) {
  asm volatile("global_load_dword %0, %1, %2" : "=v"(v) : "v"(off), "s"(base));
}
__device__ __forceinline__ void sb_arrived1(f32x4 (&v)[8], float& u) {
  asm volatile("s_waitcnt vmcnt(0)"
               : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(u)
               :
               : "memory");
}
__device__ __forceinline__ void sb_arrived2(f32x4 (&v)[8], f32x4 (&w)[8], float& u, float& u1, float& u2) {
  asm volatile("s_waitcnt vmcnt(0)"
               : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(w[0]),
                 "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), "+v"(w[7]), "+v"(u), "+v"(u1), "+v"(u2)
               :
               : "memory");
}
__device__ __forceinline__ void sb_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// request registers -> layout pi, through the wavefront's LDS slab
__device__ __forceinline__ void sb_turn_in(float* __restrict__ lds, const int lane, const f32x4 (&v)[8], float (&a)[32]) {
  float* dst = lds + (lane >> 4) * kSbLd + 4 * (lane & 15);
#pragma unroll
  for (int p = 0; p < 8; ++p) *reinterpret_cast<f32x4*>(dst + 4 * p * kSbLd) = v[p];
  sb_wave_sync();
  const float* src = lds + (lane & 31) * kSbLd + 4 * (lane >> 5);
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    const f32x4 u = *reinterpret_cast<const f32x4*>(src + 32 * (g >> 2) + 8 * (g & 3));
    a[4 * g] = u[0]; a[4 * g + 1] = u[1]; a[4 * g + 2] = u[2]; a[4 * g + 3] = u[3];
  }
  sb_wave_sync();
}
// layout pi -> request registers -> global (rows beyond `left` are not stored)
__device__ __forceinline__ void sb_store(float* __restrict__ lds, const int lane, const float (&a)[32], float* base,
                                         const unsigned (&off)[8], const int left) {
  float* dst = lds + (lane & 31) * kSbLd + 4 * (lane >> 5);
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    f32x4 u;
    u[0] = a[4 * g]; u[1] = a[4 * g + 1]; u[2] = a[4 * g + 2]; u[3] = a[4 * g + 3];
    *reinterpret_cast<f32x4*>(dst + 32 * (g >> 2) + 8 * (g & 3)) = u;
  }
  sb_wave_sync();
  const float* src = lds + (lane >> 4) * kSbLd + 4 * (lane & 15);
  char* cb = reinterpret_cast<char*>(base);
  if (left >= 32) {
#pragma unroll
    for (int p = 0; p < 8; ++p) *reinterpret_cast<f32x4*>(cb + off[p]) = *reinterpret_cast<const f32x4*>(src + 4 * p * kSbLd);
  } else {
#pragma unroll
    for (int p = 0; p < 8; ++p)
      if (4 * p + (lane >> 4) < left) *reinterpret_cast<f32x4*>(cb + off[p]) = *reinterpret_cast<const f32x4*>(src + 4 * p * kSbLd);
  }
  sb_wave_sync();
}

// the same for a tensor of pitch 64 with the requests' offsets rebuilt from the lane's part (request p = lane_off + 1024 p
// bytes: constants the stores fold into their offset fields, no array of eight offsets kept alive through a long loop)
__device__ __forceinline__ void sb_store_lin(float* __restrict__ lds, const int lane, const float (&a)[32], float* base,
                                             const unsigned lane_off, const int left) {
  float* dst = lds + (lane & 31) * kSbLd + 4 * (lane >> 5);
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    f32x4 u;
    u[0] = a[4 * g]; u[1] = a[4 * g + 1]; u[2] = a[4 * g + 2]; u[3] = a[4 * g + 3];
    *reinterpret_cast<f32x4*>(dst + 32 * (g >> 2) + 8 * (g & 3)) = u;
  }
  sb_wave_sync();
  const float* src = lds + (lane >> 4) * kSbLd + 4 * (lane & 15);
  char* cb = reinterpret_cast<char*>(base) + lane_off;
  if (left >= 32) {
#pragma unroll
    for (int p = 0; p < 8; ++p) *reinterpret_cast<f32x4*>(cb + 1024 * p) = *reinterpret_cast<const f32x4*>(src + 4 * p * kSbLd);
  } else {
#pragma unroll
    for (int p = 0; p < 8; ++p)
      if (4 * p + (lane >> 4) < left) *reinterpret_cast<f32x4*>(cb + 1024 * p) = *reinterpret_cast<const f32x4*>(src + 4 * p * kSbLd);
  }
  sb_wave_sync();
}

// y (layout pi) = x (layout pi) Wmat^T: the transposed product, weights as the A operand
__device__ __forceinline__ void sb_gemm_row(const float* __restrict__ wl, const int lane, const float (&x)[32], float (&y)[32]) {
  f32x16 a0, a1;
#pragma unroll
  for (int i = 0; i < 16; ++i) { a0[i] = 0.f; a1[i] = 0.f; }
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    const f32x4 w0 = *reinterpret_cast<const f32x4*>(wl + (g * 64 + lane) * 4);
    const f32x4 w1 = *reinterpret_cast<const f32x4*>(wl + ((8 + g) * 64 + lane) * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w0[e], x[4 * g + e], a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w1[e], x[4 * g + e], a1, 0, 0, 0);
    }
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) { y[i] = a0[i]; y[16 + i] = a1[i]; }
}

// LayerNorm of a row held in layout pi (rbx_norm.hip's arithmetic: two-pass variance, 1 / sqrtf)
__device__ __forceinline__ void sb_layernorm(float (&x)[32], const float* __restrict__ gl, const float* __restrict__ bl,
                                             const int h, const float eps, float* mean, float* rstd) {
  float s = 0.f;
#pragma unroll
  for (int r = 0; r < 32; ++r) s += x[r];
  s += __shfl_xor(s, 32, 64);
  const float m = s * (1.0f / 64.0f);
  float q = 0.f;
#pragma unroll
  for (int r = 0; r < 32; ++r) {
    x[r] -= m;
    q += x[r] * x[r];
  }
  q += __shfl_xor(q, 32, 64);
  const float rs = 1.0f / sqrtf(q * (1.0f / 64.0f) + eps);
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    const int c = 32 * (g >> 2) + 8 * (g & 3) + 4 * h;
    const f32x4 gm = *reinterpret_cast<const f32x4*>(gl + c), bt = *reinterpret_cast<const f32x4*>(bl + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) x[4 * g + e] = x[4 * g + e] * rs * gm[e] + bt[e];
  }
  *mean = m;
  *rstd = rs;
}
__device__ __forceinline__ void sb_add_vec(float (&y)[32], const float* __restrict__ vl, const int h) {
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    const f32x4 b = *reinterpret_cast<const f32x4*>(vl + 32 * (g >> 2) + 8 * (g & 3) + 4 * h);
#pragma unroll
    for (int e = 0; e < 4; ++e) y[4 * g + e] += b[e];
  }
}

struct SbQkvArgs {
  const float* x;          // [M, 64] block input (seqs)
  const float *ln_w, *ln_b, *in_w, *in_b;
  float eps;
  float *mean, *rstd, *q, *Q, *KV;
  int M;
};

// q = LayerNorm(x); Q = q Wq^T + bq; K | V = x [Wk; Wv]^T + [bk; bv]          (sasrec.py:82-84 + in_proj of the MHA)
__global__ __launch_bounds__(64 * kSbWaves, 1) void sb_qkv_fwd_kernel(const SbQkvArgs A) {
  extern __shared__ float sb_lds[];
  float* wq = sb_lds;
  float* wk = wq + kSbW;
  float* wv = wk + kSbW;
  float* vec = wv + kSbW;                     // gamma, beta, bq, bk, bv
  float* slabs = vec + 5 * 64;
  sb_stage_weight(wq, A.in_w, 64, false);
  sb_stage_weight(wk, A.in_w + 64 * 64, 64, false);
  sb_stage_weight(wv, A.in_w + 2 * 64 * 64, 64, false);
  sb_stage_vec(vec, A.ln_w, 1.f);
  sb_stage_vec(vec + 64, A.ln_b, 0.f);
  sb_stage_vec(vec + 128, A.in_b, 0.f);
  sb_stage_vec(vec + 192, A.in_b != nullptr ? A.in_b + 64 : nullptr, 0.f);
  sb_stage_vec(vec + 256, A.in_b != nullptr ? A.in_b + 128 : nullptr, 0.f);
  __syncthreads();
  const int lane = threadIdx.x & 63, m = lane & 31, h = lane >> 5;
  const int wid = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  const int nw = static_cast<int>(gridDim.x) * kSbWaves;
  const int slabs_n = (A.M + 31) >> 5;
  int s = static_cast<int>(blockIdx.x) * kSbWaves + wid;
  if (s >= slabs_n) return;
  float* lds = slabs + wid * kSbSlab;
  unsigned off64[8], off128[8], off[8];
  sb_offsets(64, 32, lane, off64);
  sb_offsets(128, 32, lane, off128);
  f32x4 nx[8];
  {
    const int left = A.M - s * 32;
#pragma unroll
    for (int p = 0; p < 8; ++p) off[p] = off64[p];
    if (left < 32) sb_offsets(64, left, lane, off);
    sb_issue(A.x + static_cast<long long>(s) * 32 * 64, off, nx);
  }
  sb_arrived(nx);
  for (;;) {
    float x[32];
    sb_turn_in(lds, lane, nx, x);
    const int r0 = s * 32;
    const int left = A.M - r0;
    int sn = s + nw;
    const bool more = sn < slabs_n;
    sn = more ? sn : s;
    {
      const int ln = A.M - sn * 32;
#pragma unroll
      for (int p = 0; p < 8; ++p) off[p] = off64[p];
      if (ln < 32) sb_offsets(64, ln, lane, off);
      sb_issue(A.x + static_cast<long long>(sn) * 32 * 64, off, nx);
    }
    float y[32];
    // K and V read the block input itself
    sb_gemm_row(wk, lane, x, y);
    sb_add_vec(y, vec + 192, h);
    sb_store(lds, lane, y, A.KV + static_cast<long long>(r0) * 128, off128, left);
    sb_gemm_row(wv, lane, x, y);
    sb_add_vec(y, vec + 256, h);
    sb_store(lds, lane, y, A.KV + static_cast<long long>(r0) * 128 + 64, off128, left);
    float mu, rs;
    sb_layernorm(x, vec, vec + 64, h, A.eps, &mu, &rs);
    if (h == 0 && m < left) {
      A.mean[r0 + m] = mu;
      A.rstd[r0 + m] = rs;
    }
    if (A.q != nullptr) sb_store(lds, lane, x, A.q + static_cast<long long>(r0) * 64, off64, left);
    sb_gemm_row(wq, lane, x, y);
    sb_add_vec(y, vec + 128, h);
    sb_arrived(nx);          // in front of the last store: behind it the wait (vmcnt counts stores) would hold the next slab
    sb_store(lds, lane, y, A.Q + static_cast<long long>(r0) * 64, off64, left);
    if (!more) break;
    s = sn;
  }
}

struct SbFfnArgs {
  // prologue (attn != NULL): x = res + attn Wo^T + bo is WRITTEN; without it x is the input
  const float *attn, *res, *wo, *bo;
  float* x;
  const float *ln_w, *ln_b, *w1, *b1, *w2, *b2, *keep;
  float eps;
  float *mean, *rstd, *n, *h, *out;           // n may be NULL (not stored)
  int M;
  // res_mean != NULL: `res` is the BLOCK INPUT e and the residual is q = LayerNorm(e), rebuilt from these statistics and
  // parameters (sasrec.py:82,86: Q = attention_layernorm(seqs) ... seqs = Q + mha) -- the first chain then need not store q
  const float *res_mean, *res_rstd, *res_ln_w, *res_ln_b;
};

// [x = res + attn Wo^T + bo;]  n = LayerNorm(x); h = relu(n W1^T + b1); out = (n + h W2^T + b2) * keep[row]
// (sasrec.py:86-92, PointWiseFeedForward :119-123 without dropout)
template <bool PRO>
__global__ __launch_bounds__(64 * kSbWaves, 1) void sb_ffn_fwd_kernel(const SbFfnArgs A) {
  extern __shared__ float sb_lds[];
  float* w1 = sb_lds;
  float* w2 = w1 + kSbW;
  float* wo = w2 + kSbW;                       // (PRO only)
  float* vec = PRO ? wo + kSbW : wo;           // gamma, beta, b1, b2, bo, the residual LayerNorm's gamma, beta
  float* slabs = vec + 7 * 64;
  sb_stage_weight(w1, A.w1, 64, false);
  sb_stage_weight(w2, A.w2, 64, false);
  if constexpr (PRO) sb_stage_weight(wo, A.wo, 64, false);
  sb_stage_vec(vec, A.ln_w, 1.f);
  sb_stage_vec(vec + 64, A.ln_b, 0.f);
  sb_stage_vec(vec + 128, A.b1, 0.f);
  sb_stage_vec(vec + 192, A.b2, 0.f);
  sb_stage_vec(vec + 256, PRO ? A.bo : nullptr, 0.f);
  sb_stage_vec(vec + 320, PRO ? A.res_ln_w : nullptr, 1.f);
  sb_stage_vec(vec + 384, PRO ? A.res_ln_b : nullptr, 0.f);
  __syncthreads();
  const int lane = threadIdx.x & 63, m = lane & 31, h = lane >> 5;
  const int wid = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  const int nw = static_cast<int>(gridDim.x) * kSbWaves;
  const int slabs_n = (A.M + 31) >> 5;
  int s = static_cast<int>(blockIdx.x) * kSbWaves + wid;
  if (s >= slabs_n) return;
  float* lds = slabs + wid * kSbSlab;
  unsigned off64[8], off[8];
  sb_offsets(64, 32, lane, off64);
  const float* in0 = PRO ? A.attn : A.x;
  const bool rebuild = PRO && A.res_mean != nullptr;
  f32x4 nx[8], nr[8];
  float nkp, nrmu = 0.f, nrrs = 0.f;            // the rows' keep (and the residual's statistics) travel with the slab's requests
  const float* keep_p = A.keep != nullptr ? A.keep : in0;
  const float* rmu_p = rebuild ? A.res_mean : in0;
  const float* rrs_p = rebuild ? A.res_rstd : in0;
  {
    const int left = A.M - s * 32;
#pragma unroll
    for (int p = 0; p < 8; ++p) off[p] = off64[p];
    if (left < 32) sb_offsets(64, left, lane, off);
    sb_issue(in0 + static_cast<long long>(s) * 32 * 64, off, nx);
    if constexpr (PRO) sb_issue(A.res + static_cast<long long>(s) * 32 * 64, off, nr);
    sb_issue_word(keep_p, 4u * static_cast<unsigned>(s * 32 + m < A.M ? s * 32 + m : A.M - 1), nkp);
    if constexpr (PRO) {
      sb_issue_word(rmu_p, 4u * static_cast<unsigned>(s * 32 + m < A.M ? s * 32 + m : A.M - 1), nrmu);
      sb_issue_word(rrs_p, 4u * static_cast<unsigned>(s * 32 + m < A.M ? s * 32 + m : A.M - 1), nrrs);
    }
  }
  if constexpr (PRO) sb_arrived2(nx, nr, nkp, nrmu, nrrs);
  else sb_arrived1(nx, nkp);
  for (;;) {
    float x[32];
    const int r0 = s * 32;
    const int left = A.M - r0;
    int sn = s + nw;
    const bool more = sn < slabs_n;
    sn = more ? sn : s;
    if constexpr (PRO) {
      float o[32];
      sb_turn_in(lds, lane, nx, o);
      sb_turn_in(lds, lane, nr, x);
      if (rebuild) {                           // the residual is LayerNorm(e): its rows' statistics are two L2-resident words
        const float rmu = nrmu, rrs = nrrs;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
          const int c = 32 * (g >> 2) + 8 * (g & 3) + 4 * h;
          const f32x4 gm = *reinterpret_cast<const f32x4*>(vec + 320 + c), bt = *reinterpret_cast<const f32x4*>(vec + 384 + c);
#pragma unroll
          for (int e = 0; e < 4; ++e) x[4 * g + e] = (x[4 * g + e] - rmu) * rrs * gm[e] + bt[e];
        }
      }
      {
        const int ln = A.M - sn * 32;
#pragma unroll
        for (int p = 0; p < 8; ++p) off[p] = off64[p];
        if (ln < 32) sb_offsets(64, ln, lane, off);
        sb_issue(in0 + static_cast<long long>(sn) * 32 * 64, off, nx);
        sb_issue(A.res + static_cast<long long>(sn) * 32 * 64, off, nr);
      }
      float y[32];
      sb_gemm_row(wo, lane, o, y);
      sb_add_vec(y, vec + 256, h);
#pragma unroll
      for (int r = 0; r < 32; ++r) x[r] += y[r];
      sb_store(lds, lane, x, A.x + static_cast<long long>(r0) * 64, off64, left);
    } else {
      sb_turn_in(lds, lane, nx, x);
      const int ln = A.M - sn * 32;
#pragma unroll
      for (int p = 0; p < 8; ++p) off[p] = off64[p];
      if (ln < 32) sb_offsets(64, ln, lane, off);
      sb_issue(in0 + static_cast<long long>(sn) * 32 * 64, off, nx);
    }
    const float kp = A.keep != nullptr ? nkp : 1.f;
    sb_issue_word(keep_p, 4u * static_cast<unsigned>(sn * 32 + m < A.M ? sn * 32 + m : A.M - 1), nkp);
    if constexpr (PRO) {
      sb_issue_word(rmu_p, 4u * static_cast<unsigned>(sn * 32 + m < A.M ? sn * 32 + m : A.M - 1), nrmu);
      sb_issue_word(rrs_p, 4u * static_cast<unsigned>(sn * 32 + m < A.M ? sn * 32 + m : A.M - 1), nrrs);
    }
    float mu, rs;
    sb_layernorm(x, vec, vec + 64, h, A.eps, &mu, &rs);
    if (h == 0 && m < left) {
      A.mean[r0 + m] = mu;
      A.rstd[r0 + m] = rs;
    }
    if (A.n != nullptr) sb_store(lds, lane, x, A.n + static_cast<long long>(r0) * 64, off64, left);
    float hh[32];
    sb_gemm_row(w1, lane, x, hh);
    sb_add_vec(hh, vec + 128, h);
#pragma unroll
    for (int r = 0; r < 32; ++r) hh[r] = hh[r] > 0.f ? hh[r] : 0.f;
    sb_store(lds, lane, hh, A.h + static_cast<long long>(r0) * 64, off64, left);
    float y[32];
    sb_gemm_row(w2, lane, hh, y);
    sb_add_vec(y, vec + 192, h);
#pragma unroll
    for (int r = 0; r < 32; ++r) y[r] = (y[r] + x[r]) * kp;
    if constexpr (PRO) sb_arrived2(nx, nr, nkp, nrmu, nrrs);   // in front of the last store (see sb_qkv_fwd_kernel)
    else sb_arrived1(nx, nkp);
    sb_store(lds, lane, y, A.out + static_cast<long long>(r0) * 64, off64, left);
    if (!more) break;
    s = sn;
  }
}

// ---- backward of the feed-forward sub-layer as ONE pass ----------------------------------------------------------------
// dout -> g = dout * keep;  dW2 += g^T h, db2 += colsum g;  dh = (g W2) o [h > 0];  dW1 += dh^T n, db1 += colsum dh;
// dn = dh W1 + g;  LayerNorm backward of dn (dgamma, dbeta, dx).  As separate launches: two dW slab passes, two dx GEMMs,
// the LayerNorm backward and its final kernel -- 13 reads and 3 writes of [M, 64]; here dout, h and x are read once and dx
// is written once.  The chain (dh, dn, dx) stays in the row layout; the weight-gradient products contract over ROWS, so
// their operands are wanted in the column layout (lane = column, register s <-> row rho(s, h) = (s & 3) + 8 (s >> 2) + 4 h:
// the layout the natural MFMA product hands out) -- a tensor gets there by one trip through the wavefront's LDS slab
// (8 b128 writes, 32 b32 reads: nothing beside the 64 MFMAs of a product).  n = xhat gamma + beta is rebuilt from the
// block input and the saved statistics, so the forward does not have to store it.  Four products per slab and four
// [M, 64] streams: the matrix pipe (f32: 43 us per product at cfg 5) and HBM are balanced.  One wavefront per SIMD with the
// whole 512-register file: 128 accumulators for dW1 | dW2 and three slabs of prefetch live through the loop.
// Partial sums: per workgroup (its four wavefronts added through LDS in a fixed order), reduced by sb_reduce_kernel in
// a fixed order -- deterministic, no float atomics.
constexpr int kSbBwdWaves = 4;
constexpr int kSbFfnPart = 2 * kSbW + 4 * 64;          // dW2 | dW1 | db2 | db1 | dgamma | dbeta

__device__ __forceinline__ void sb_row_to_lds(float* __restrict__ lds, const int lane, const float (&a)[32]) {
  float* dst = lds + (lane & 31) * kSbLd + 4 * (lane >> 5);
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    f32x4 u;
    u[0] = a[4 * g]; u[1] = a[4 * g + 1]; u[2] = a[4 * g + 2]; u[3] = a[4 * g + 3];
    *reinterpret_cast<f32x4*>(dst + 32 * (g >> 2) + 8 * (g & 3)) = u;
  }
}
// c[16 t + s] = tile[rho(s, h)][32 t + m]
__device__ __forceinline__ void sb_lds_to_col(const float* __restrict__ lds, const int lane, float (&c)[32]) {
  const float* src = lds + 4 * (lane >> 5) * kSbLd + (lane & 31);
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int s = 0; s < 16; ++s) c[16 * t + s] = src[((s & 3) + 8 * (s >> 2)) * kSbLd + 32 * t];
}
// row layout registers -> column layout registers through the slab
__device__ __forceinline__ void sb_row_to_col(float* __restrict__ lds, const int lane, const float (&a)[32], float (&c)[32]) {
  sb_row_to_lds(lds, lane, a);
  sb_wave_sync();
  sb_lds_to_col(lds, lane, c);
  sb_wave_sync();
}
// acc[tn][tk] += G^T X over the slab's 32 rows (operands in the column layout)
__device__ __forceinline__ void sb_dw_acc(const float (&gc)[32], const float (&xc)[32], f32x16 (&acc)[2][2]) {
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(gc[s], xc[s], acc[0][0], 0, 0, 0);
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(gc[s], xc[16 + s], acc[0][1], 0, 0, 0);
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(gc[16 + s], xc[s], acc[1][0], 0, 0, 0);
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(gc[16 + s], xc[16 + s], acc[1][1], 0, 0, 0);
  }
}
__device__ __forceinline__ void sb_arrived3(f32x4 (&a)[8], f32x4 (&b)[8], f32x4 (&c)[8], float& u0, float& u1, float& u2) {
  asm volatile("s_waitcnt vmcnt(0)"
               : "+a"(a[0]), "+a"(a[1]), "+a"(a[2]), "+a"(a[3]), "+a"(a[4]), "+a"(a[5]), "+a"(a[6]), "+a"(a[7]), "+a"(b[0]),
                 "+a"(b[1]), "+a"(b[2]), "+a"(b[3]), "+a"(b[4]), "+a"(b[5]), "+a"(b[6]), "+a"(b[7]), "+v"(c[0]), "+v"(c[1]),
                 "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]), "+v"(u0), "+v"(u1), "+v"(u2)
               :
               : "memory");
}
// a wavefront's accumulators into the workgroup's LDS sum (first: store, later: add)
__device__ __forceinline__ void sb_acc_to_lds(float* __restrict__ red, const int lane, const f32x16 (&acc)[2][2], const bool first) {
  const int m = lane & 31, h = lane >> 5;
#pragma unroll
  for (int tn = 0; tn < 2; ++tn)
#pragma unroll
    for (int tk = 0; tk < 2; ++tk)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float* d = red + (32 * tn + (r & 3) + 8 * (r >> 2) + 4 * h) * 64 + 32 * tk + m;
        *d = first ? acc[tn][tk][r] : *d + acc[tn][tk][r];
      }
}
__device__ __forceinline__ void sb_colsum_to_lds(float* __restrict__ red, const int lane, const float (&v)[2], const bool first) {
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const float w = v[t] + __shfl_xor(v[t], 32, 64);
    if (lane < 32) red[32 * t + lane] = first ? w : red[32 * t + lane] + w;
  }
}

struct SbFfnBwdArgs {
  const float *g0, *keep, *h, *x, *mean, *rstd, *ln_w, *ln_b, *w1, *w2;
  float *dx, *part;
  int M;
};

// ORDER 1: the two chain products are written in FRONT of the LDS trips that do not depend on them (dh = g W2 before the
// column-layout trips of g and h, dn = dh W1 before dh's), so that the scheduler has independent work to put between
// the MFMAs of a product: one wavefront per SIMD hides nothing by itself.
template <int ORDER>
__global__ __launch_bounds__(64 * kSbBwdWaves, 1) void sb_ffn_bwd_kernel(const SbFfnBwdArgs A) {
  extern __shared__ float sb_lds[];
  float* w2t = sb_lds;                         // dh = g W2:  Wmat[k][n] = W2[n][k]
  float* w1t = w2t + kSbW;                     // dn = dh W1
  float* vec = w1t + kSbW;                     // gamma, beta
  float* slabs = vec + 2 * 64;
  sb_stage_weight(w2t, A.w2, 64, true);
  sb_stage_weight(w1t, A.w1, 64, true);
  sb_stage_vec(vec, A.ln_w, 1.f);
  sb_stage_vec(vec + 64, A.ln_b, 0.f);
  __syncthreads();
  const int lane = threadIdx.x & 63, m = lane & 31, h = lane >> 5;
  const int wid = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  const int nw = static_cast<int>(gridDim.x) * kSbBwdWaves;
  const int slabs_n = (A.M + 31) >> 5;
  int s = static_cast<int>(blockIdx.x) * kSbBwdWaves + wid;
  float* lds = slabs + wid * 3 * kSbSlab;       // the wavefront's scratch slab, then g and xhat parked for the slab's lifetime
  float* lds_g = lds + kSbSlab;
  float* lds_x = lds_g + kSbSlab;
  f32x16 acc2[2][2], acc1[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) { acc2[a][b][i] = 0.f; acc1[a][b][i] = 0.f; }
  float db2[2] = {0.f, 0.f}, db1[2] = {0.f, 0.f}, dgm[2] = {0.f, 0.f}, dbt[2] = {0.f, 0.f};
  const float gmc[2] = {vec[m], vec[32 + m]}, btc[2] = {vec[64 + m], vec[96 + m]};      // column layout: columns 32 t + m
  if (s < slabs_n) {
    const unsigned lane_part = static_cast<unsigned>(16 * (lane & 15));
    const unsigned lane_off = static_cast<unsigned>(256 * (lane >> 4)) + lane_part;      // request p: lane_off + 1024 p
    f32x4 ng[8], nh[8], nx[8];
    float nkp, nmu, nrs;                                  // the rows' keep / mean / rstd travel with the slab's requests
    const float* keep_p = A.keep != nullptr ? A.keep : A.mean;
    {
      // a ragged last slab re-reads its last row: min() with that row's offset clamps a request (row * 256 + lane part)
      unsigned off[8];
      const int left = A.M - s * 32;
      const unsigned lim = static_cast<unsigned>((left < 32 ? left : 32) - 1) * 256u + lane_part;
#pragma unroll
      for (int p = 0; p < 8; ++p) off[p] = lane_off + 1024u * p < lim ? lane_off + 1024u * p : lim;
      const long long o = static_cast<long long>(s) * 32 * 64;
      sb_issue_a(A.g0 + o, off, ng);
      sb_issue_a(A.h + o, off, nh);
      sb_issue(A.x + o, off, nx);
      const unsigned ro = 4u * static_cast<unsigned>(s * 32 + m < A.M ? s * 32 + m : A.M - 1);
      sb_issue_word(keep_p, ro, nkp);
      sb_issue_word(A.mean, ro, nmu);
      sb_issue_word(A.rstd, ro, nrs);
    }
    sb_arrived3(ng, nh, nx, nkp, nmu, nrs);
    for (;;) {
      const int r0 = s * 32;
      const int left = A.M - r0;
      int sn = s + nw;
      const bool more = sn < slabs_n;
      sn = more ? sn : s;
      const float kp = (m < left) ? (A.keep != nullptr ? nkp : 1.f) : 0.f;             // rows beyond the end add nothing
      const float mu = nmu, rs = nrs;
      float g[32], xc[32];
      unsigned hmask = 0u;
      sb_turn_in(lds, lane, ng, g);
#pragma unroll
      for (int r = 0; r < 32; ++r) g[r] *= kp;
      float dh[32], dn[32];
      if constexpr (ORDER == 1) sb_gemm_row(w2t, lane, g, dh);
      {
        float gc[32], hr[32], hc[32];
        sb_row_to_lds(lds_g, lane, g);                   // (kept there: read back for dn = dh W1 + g)
        sb_wave_sync();
        sb_lds_to_col(lds_g, lane, gc);
        sb_turn_in(lds, lane, nh, hr);                   // (the slab keeps the tile: the column layout is read from it)
#pragma unroll
        for (int r = 0; r < 32; ++r) hmask |= (hr[r] > 0.f ? 1u : 0u) << r;
        sb_lds_to_col(lds, lane, hc);
        sb_wave_sync();
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int q = 0; q < 16; ++q) db2[t] += gc[16 * t + q];
        sb_dw_acc(gc, hc, acc2);
      }
      {
        float xh[32];
        sb_turn_in(lds, lane, nx, xh);
#pragma unroll
        for (int r = 0; r < 32; ++r) xh[r] = (xh[r] - mu) * rs;
        sb_row_to_lds(lds_x, lane, xh);                  // (kept there: the LayerNorm backward reads it at the end)
        sb_wave_sync();
        sb_lds_to_col(lds_x, lane, xc);
      }
      {
        unsigned off[8];
        const int ln = A.M - sn * 32;
        const unsigned lim = static_cast<unsigned>((ln < 32 ? ln : 32) - 1) * 256u + lane_part;
#pragma unroll
        for (int p = 0; p < 8; ++p) off[p] = lane_off + 1024u * p < lim ? lane_off + 1024u * p : lim;
        const long long o = static_cast<long long>(sn) * 32 * 64;
        sb_issue_a(A.g0 + o, off, ng);
        sb_issue_a(A.h + o, off, nh);
        sb_issue(A.x + o, off, nx);
        const unsigned ro = 4u * static_cast<unsigned>(sn * 32 + m < A.M ? sn * 32 + m : A.M - 1);
        sb_issue_word(keep_p, ro, nkp);
        sb_issue_word(A.mean, ro, nmu);
        sb_issue_word(A.rstd, ro, nrs);
      }
      if constexpr (ORDER == 0) sb_gemm_row(w2t, lane, g, dh);
#pragma unroll
      for (int r = 0; r < 32; ++r) dh[r] = ((hmask >> r) & 1u) ? dh[r] : 0.f;
      if constexpr (ORDER == 1) sb_gemm_row(w1t, lane, dh, dn);
      {
        float dc[32];
        sb_row_to_col(lds, lane, dh, dc);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int q = 0; q < 16; ++q) db1[t] += dc[16 * t + q];
        // dW1 += dh^T n with n = xhat gamma + beta rebuilt per operand
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const float n0 = xc[q] * gmc[0] + btc[0], n1 = xc[16 + q] * gmc[1] + btc[1];
          acc1[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(dc[q], n0, acc1[0][0], 0, 0, 0);
          acc1[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(dc[q], n1, acc1[0][1], 0, 0, 0);
          acc1[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(dc[16 + q], n0, acc1[1][0], 0, 0, 0);
          acc1[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(dc[16 + q], n1, acc1[1][1], 0, 0, 0);
        }
      }
      if constexpr (ORDER == 0) sb_gemm_row(w1t, lane, dh, dn);
      {
        const float* src = lds_g + m * kSbLd + 4 * h;
#pragma unroll
        for (int gq = 0; gq < 8; ++gq) {
          const f32x4 u = *reinterpret_cast<const f32x4*>(src + 32 * (gq >> 2) + 8 * (gq & 3));
#pragma unroll
          for (int e = 0; e < 4; ++e) dn[4 * gq + e] += u[e];
        }
      }
      {
        float dc[32];
        sb_row_to_col(lds, lane, dn, dc);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            dbt[t] += dc[16 * t + q];
            dgm[t] += dc[16 * t + q] * xc[16 * t + q];
          }
      }
      // LayerNorm backward of the row (rbx_norm.hip's arithmetic)
      float xh[32];
      {
        const float* src = lds_x + m * kSbLd + 4 * h;
#pragma unroll
        for (int gq = 0; gq < 8; ++gq) {
          const f32x4 u = *reinterpret_cast<const f32x4*>(src + 32 * (gq >> 2) + 8 * (gq & 3));
          xh[4 * gq] = u[0]; xh[4 * gq + 1] = u[1]; xh[4 * gq + 2] = u[2]; xh[4 * gq + 3] = u[3];
        }
      }
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int gq = 0; gq < 8; ++gq) {
        const f32x4 gm = *reinterpret_cast<const f32x4*>(vec + 32 * (gq >> 2) + 8 * (gq & 3) + 4 * h);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          dn[4 * gq + e] *= gm[e];
          s1 += dn[4 * gq + e];
          s2 += dn[4 * gq + e] * xh[4 * gq + e];
        }
      }
      s1 += __shfl_xor(s1, 32, 64);
      s2 += __shfl_xor(s2, 32, 64);
      const float m1 = s1 * (1.0f / 64.0f), m2 = s2 * (1.0f / 64.0f);
#pragma unroll
      for (int r = 0; r < 32; ++r) dn[r] = rs * (dn[r] - m1 - xh[r] * m2);
      // the next slab's streams are waited for HERE, in front of the store: behind it, the wait (vmcnt counts stores
      // too) would hold the next slab until this one's rows are in L2
      sb_arrived3(ng, nh, nx, nkp, nmu, nrs);
      sb_store_lin(lds, lane, dn, A.dx + static_cast<long long>(r0) * 64, lane_off, left);
      if (!more) break;
      s = sn;
    }
  }
  __syncthreads();                               // every wavefront is done with the weights: the LDS becomes the sum
  float* red = sb_lds;
  for (int w = 0; w < kSbBwdWaves; ++w) {
    if (wid == w) {
      sb_acc_to_lds(red, lane, acc2, w == 0);
      sb_acc_to_lds(red + kSbW, lane, acc1, w == 0);
      sb_colsum_to_lds(red + 2 * kSbW, lane, db2, w == 0);
      sb_colsum_to_lds(red + 2 * kSbW + 64, lane, db1, w == 0);
      sb_colsum_to_lds(red + 2 * kSbW + 128, lane, dgm, w == 0);
      sb_colsum_to_lds(red + 2 * kSbW + 192, lane, dbt, w == 0);
    }
    __syncthreads();
  }
  float* dst = A.part + static_cast<long long>(blockIdx.x) * kSbFfnPart;
  for (int i = threadIdx.x; i < kSbFfnPart; i += 64 * kSbBwdWaves) dst[i] = red[i];
}

// ---- backward of the attention sub-layer's input side as ONE pass ----------------------------------------------------------
// Behind the attention's backward: dq = dQ Wq + g (q = LayerNorm(e) feeds the query projection and the residual),
// de = LayerNormBackward(dq) + dK Wk + dV Wv (e feeds the LayerNorm and the key / value projection), dgamma, dbeta.
// As separate launches: a dx GEMM, the LayerNorm backward + its final kernel, the 128 -> 64 slab GEMM -- 8 reads and 3
// writes of [M, 64]; here dQ, dK | dV, g and e are read once and de is written once (the three weight gradients stay
// with the slab dW kernels).  No accumulators to speak of, so two wavefronts per SIMD share the pipes; a slab's five
// requests are issued together and waited for one by one (vmcnt counts down in order), the first product runs while the
// rest is still on its way.
constexpr int kSbLnPart = 128;                         // dgamma | dbeta

template <int N>
__device__ __forceinline__ void sb_wait_tile(f32x4 (&v)[8]) {
  asm volatile("s_waitcnt vmcnt(%8)"
               : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7])
               : "n"(N)
               : "memory");
}
template <int N>
__device__ __forceinline__ void sb_wait_tile2(f32x4 (&v)[8], float& u0, float& u1) {
  asm volatile("s_waitcnt vmcnt(%10)"
               : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(u0), "+v"(u1)
               : "n"(N)
               : "memory");
}

struct SbAttnInBwdArgs {
  const float *dQ, *dKV, *g, *x, *mean, *rstd, *ln_w, *in_w;
  float *de, *part;
  int M;
  const float* row_scale;      // optional: de is stored as de * row_scale[row] * alpha (the backward of SASRec's input stage,
  float alpha;                 // (alpha e + position) * keep, sasrec.py:68-77, folded into the block in front of it)
};
template <int N>
__device__ __forceinline__ void sb_wait_tile3(f32x4 (&v)[8], float& u0, float& u1, float& u2) {
  asm volatile("s_waitcnt vmcnt(%11)"
               : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(u0), "+v"(u1),
                 "+v"(u2)
               : "n"(N)
               : "memory");
}

__global__ __launch_bounds__(64 * kSbWaves, 1) void sb_attn_in_bwd_kernel(const SbAttnInBwdArgs A) {
  extern __shared__ float sb_lds[];
  float* wq = sb_lds;                          // dq = dQ Wq: Wmat[k][n] = Wq[n][k]
  float* wk = wq + kSbW;
  float* wv = wk + kSbW;
  float* vec = wv + kSbW;                      // gamma
  float* slabs = vec + 64;
  sb_stage_weight(wq, A.in_w, 64, true);
  sb_stage_weight(wk, A.in_w + 64 * 64, 64, true);
  sb_stage_weight(wv, A.in_w + 2 * 64 * 64, 64, true);
  sb_stage_vec(vec, A.ln_w, 1.f);
  __syncthreads();
  const int lane = threadIdx.x & 63, m = lane & 31, h = lane >> 5;
  const int wid = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  const int nw = static_cast<int>(gridDim.x) * kSbWaves;
  const int slabs_n = (A.M + 31) >> 5;
  float* lds = slabs + wid * kSbSlab;
  float dgm[2] = {0.f, 0.f}, dbt[2] = {0.f, 0.f};
  const unsigned lane_part = static_cast<unsigned>(16 * (lane & 15));
  const unsigned lane_off = static_cast<unsigned>(256 * (lane >> 4)) + lane_part;        // request p: lane_off + 1024 p
  for (int s = static_cast<int>(blockIdx.x) * kSbWaves + wid; s < slabs_n; s += nw) {
    const int r0 = s * 32;
    const int left = A.M - r0;
    f32x4 tq[8], tg[8], tx[8], tk[8], tv[8];
    float mu, rs, rsc;
    unsigned off[8];
    const unsigned lim = static_cast<unsigned>((left < 32 ? left : 32) - 1) * 256u + lane_part;
#pragma unroll
    for (int p = 0; p < 8; ++p) off[p] = lane_off + 1024u * p < lim ? lane_off + 1024u * p : lim;
    const long long o = static_cast<long long>(r0) * 64;
    {
      const unsigned ro = 4u * static_cast<unsigned>(r0 + m < A.M ? r0 + m : A.M - 1);
      sb_issue_word(A.mean, ro, mu);
      sb_issue_word(A.rstd, ro, rs);
      sb_issue_word(A.row_scale != nullptr ? A.row_scale : A.mean, ro, rsc);
      sb_issue(A.dQ + o, off, tq);
      sb_issue(A.g + o, off, tg);
      sb_issue(A.x + o, off, tx);
    }
    float a[32], dq[32], xh[32];
    sb_wait_tile3<16>(tq, mu, rs, rsc);
    sb_turn_in(lds, lane, tq, a);
    sb_gemm_row(wq, lane, a, dq);
    sb_wait_tile<8>(tg);
    sb_turn_in(lds, lane, tg, a);
    {
      // dK | dV are requested once two of the first three slabs have left their registers (160 registers of requests in
      // flight at once spilled); they arrive behind x: vmcnt counts down in order
#pragma unroll
      for (int p = 0; p < 8; ++p) off[p] = 2u * off[p] - lane_part;      // the same rows at a pitch of 128 floats
      sb_issue(A.dKV + 2 * o, off, tk);
      sb_issue(A.dKV + 2 * o + 64, off, tv);
    }
    const bool live = m < left;                           // rows beyond the end add nothing to the column sums
#pragma unroll
    for (int r = 0; r < 32; ++r) dq[r] = live ? dq[r] + a[r] : 0.f;
    sb_wait_tile<16>(tx);                                 // (the 16 requests of dK | dV may still be out)
    sb_turn_in(lds, lane, tx, xh);
#pragma unroll
    for (int r = 0; r < 32; ++r) xh[r] = (xh[r] - mu) * rs;
    {
      float dc[32], xc[32];
      sb_row_to_col(lds, lane, dq, dc);
      sb_row_to_col(lds, lane, xh, xc);
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          dbt[t] += dc[16 * t + q];
          dgm[t] += dc[16 * t + q] * xc[16 * t + q];
        }
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int gq = 0; gq < 8; ++gq) {
      const f32x4 gm = *reinterpret_cast<const f32x4*>(vec + 32 * (gq >> 2) + 8 * (gq & 3) + 4 * h);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        dq[4 * gq + e] *= gm[e];
        s1 += dq[4 * gq + e];
        s2 += dq[4 * gq + e] * xh[4 * gq + e];
      }
    }
    s1 += __shfl_xor(s1, 32, 64);
    s2 += __shfl_xor(s2, 32, 64);
    const float m1 = s1 * (1.0f / 64.0f), m2 = s2 * (1.0f / 64.0f);
#pragma unroll
    for (int r = 0; r < 32; ++r) dq[r] = rs * (dq[r] - m1 - xh[r] * m2);
    float y[32];
    sb_wait_tile<8>(tk);
    sb_turn_in(lds, lane, tk, a);
    sb_gemm_row(wk, lane, a, y);
#pragma unroll
    for (int r = 0; r < 32; ++r) dq[r] += y[r];
    sb_wait_tile<0>(tv);
    sb_turn_in(lds, lane, tv, a);
    sb_gemm_row(wv, lane, a, y);
    const float osc = A.row_scale != nullptr ? rsc * A.alpha : 1.f;
#pragma unroll
    for (int r = 0; r < 32; ++r) dq[r] = (dq[r] + y[r]) * osc;
    sb_store_lin(lds, lane, dq, A.de + static_cast<long long>(r0) * 64, lane_off, left);
  }
  __syncthreads();
  float* red = sb_lds;
  for (int w = 0; w < kSbWaves; ++w) {
    if (wid == w) {
      sb_colsum_to_lds(red, lane, dgm, w == 0);
      sb_colsum_to_lds(red + 64, lane, dbt, w == 0);
    }
    __syncthreads();
  }
  if (threadIdx.x < kSbLnPart) A.part[static_cast<long long>(blockIdx.x) * kSbLnPart + threadIdx.x] = red[threadIdx.x];
}

// ---- backward of the out-projection as ONE pass: dO = g Wo, dWo = g^T O, dbo = colsum g ------------------------------------
// (a dW slab pass + a dx GEMM before: g read twice.)  64 accumulators: two wavefronts per SIMD still fit.
constexpr int kSbOutPart = kSbW + 64;                  // dWo | dbo

struct SbAttnOutBwdArgs {
  const float *g, *O, *wo;
  float *dO, *part;
  int M;
};

__global__ __launch_bounds__(64 * kSbWaves, 1) void sb_attn_out_bwd_kernel(const SbAttnOutBwdArgs A) {
  extern __shared__ float sb_lds[];
  float* wot = sb_lds;                         // dO = g Wo: Wmat[k][n] = Wo[n][k]
  float* slabs = wot + kSbW + 64;              // (+ 64: the workgroup's sum, kSbOutPart floats, ends in front of the slabs)
  sb_stage_weight(wot, A.wo, 64, true);
  __syncthreads();
  const int lane = threadIdx.x & 63, m = lane & 31;
  const int wid = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  const int nw = static_cast<int>(gridDim.x) * kSbWaves;
  const int slabs_n = (A.M + 31) >> 5;
  float* lds = slabs + wid * kSbSlab;
  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
  float dbo[2] = {0.f, 0.f};
  const unsigned lane_part = static_cast<unsigned>(16 * (lane & 15));
  const unsigned lane_off = static_cast<unsigned>(256 * (lane >> 4)) + lane_part;
  for (int s = static_cast<int>(blockIdx.x) * kSbWaves + wid; s < slabs_n; s += nw) {
    const int r0 = s * 32;
    const int left = A.M - r0;
    f32x4 tg[8], to[8];
    {
      unsigned off[8];
      const unsigned lim = static_cast<unsigned>((left < 32 ? left : 32) - 1) * 256u + lane_part;
#pragma unroll
      for (int p = 0; p < 8; ++p) off[p] = lane_off + 1024u * p < lim ? lane_off + 1024u * p : lim;
      const long long o = static_cast<long long>(r0) * 64;
      sb_issue(A.g + o, off, tg);
      sb_issue(A.O + o, off, to);
    }
    float g[32], gc[32], oc[32];
    sb_wait_tile<8>(tg);
    sb_turn_in(lds, lane, tg, g);
    if (left < 32) {                                      // rows beyond the end add nothing (wave-uniform test)
      const bool live = m < left;
#pragma unroll
      for (int r = 0; r < 32; ++r) g[r] = live ? g[r] : 0.f;
      sb_row_to_lds(lds, lane, g);
      sb_wave_sync();
    }
    sb_lds_to_col(lds, lane, gc);                         // (the slab still holds the tile)
    sb_wave_sync();
    {
      float y[32];
      sb_gemm_row(wot, lane, g, y);
      sb_wait_tile<0>(to);                                // in front of the store (vmcnt counts stores too)
      sb_store_lin(lds, lane, y, A.dO + static_cast<long long>(r0) * 64, lane_off, left);
    }
    {
      float* dst = lds + (lane >> 4) * kSbLd + 4 * (lane & 15);
#pragma unroll
      for (int p = 0; p < 8; ++p) *reinterpret_cast<f32x4*>(dst + 4 * p * kSbLd) = to[p];
      sb_wave_sync();
      sb_lds_to_col(lds, lane, oc);
      sb_wave_sync();
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int q = 0; q < 16; ++q) dbo[t] += gc[16 * t + q];
    sb_dw_acc(gc, oc, acc);
  }
  __syncthreads();
  float* red = sb_lds;
  for (int w = 0; w < kSbWaves; ++w) {
    if (wid == w) {
      sb_acc_to_lds(red, lane, acc, w == 0);
      sb_colsum_to_lds(red + kSbW, lane, dbo, w == 0);
    }
    __syncthreads();
  }
  float* dstp = A.part + static_cast<long long>(blockIdx.x) * kSbOutPart;
  for (int i = threadIdx.x; i < kSbOutPart; i += 64 * kSbWaves) dstp[i] = red[i];
}

// ---- the three in-projection weight gradients as ONE pass: dWq = dQ^T q, dWk = dK^T e, dWv = dV^T e (+ their column sums) -----
// Three slab dW launches read e twice and the stored q once; here e is read once and q = LayerNorm(e) is rebuilt in the
// column layout from the saved statistics (16 rows per lane: their mean / rstd come through a 64-float LDS vector).
// 192 accumulators: one wavefront per SIMD, dQ and e prefetched into AGPRs, dK and dV into VGPRs.
constexpr int kSbInPart = 3 * kSbW + 3 * 64;           // dWq | dWk | dWv | dbq | dbk | dbv

struct SbInDwArgs {
  const float *dQ, *dKV, *x, *mean, *rstd, *ln_w, *ln_b;
  float* part;
  int M;
};
__device__ __forceinline__ void sb_arrived4(f32x4 (&a)[8], f32x4 (&b)[8], f32x4 (&c)[8], f32x4 (&d)[8], float& u0, float& u1) {
  asm volatile("s_waitcnt vmcnt(0)"
               : "+a"(a[0]), "+a"(a[1]), "+a"(a[2]), "+a"(a[3]), "+a"(a[4]), "+a"(a[5]), "+a"(a[6]), "+a"(a[7]), "+a"(b[0]),
                 "+a"(b[1]), "+a"(b[2]), "+a"(b[3]), "+a"(b[4]), "+a"(b[5]), "+a"(b[6]), "+a"(b[7])
               :
               : "memory");
  asm volatile("" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]), "+v"(d[0]),
                    "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]), "+v"(u0), "+v"(u1)
               :
               : "memory");
}
// request registers -> the slab -> column layout (the row layout is not wanted here)
__device__ __forceinline__ void sb_req_to_col(float* __restrict__ lds, const int lane, const f32x4 (&v)[8], float (&c)[32]) {
  float* dst = lds + (lane >> 4) * kSbLd + 4 * (lane & 15);
#pragma unroll
  for (int p = 0; p < 8; ++p) *reinterpret_cast<f32x4*>(dst + 4 * p * kSbLd) = v[p];
  sb_wave_sync();
  sb_lds_to_col(lds, lane, c);
  sb_wave_sync();
}

__global__ __launch_bounds__(64 * kSbBwdWaves, 1) void sb_inproj_dw_kernel(const SbInDwArgs A) {
  extern __shared__ float sb_lds[];
  float* vec = sb_lds;                         // gamma, beta
  float* stats = vec + 128;                    // per wavefront: mean[32] | rstd[32] of the slab's rows
  float* slabs = sb_lds + kSbInPart;           // (the workgroup's sum takes the front of the LDS at the end)
  sb_stage_vec(vec, A.ln_w, 1.f);
  sb_stage_vec(vec + 64, A.ln_b, 0.f);
  __syncthreads();
  const int lane = threadIdx.x & 63, m = lane & 31, h = lane >> 5;
  const int wid = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  const int nw = static_cast<int>(gridDim.x) * kSbBwdWaves;
  const int slabs_n = (A.M + 31) >> 5;
  int s = static_cast<int>(blockIdx.x) * kSbBwdWaves + wid;
  float* lds = slabs + wid * kSbSlab;
  float* st = stats + wid * 64;
  f32x16 aq[2][2], ak[2][2], av[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) { aq[a][b][i] = 0.f; ak[a][b][i] = 0.f; av[a][b][i] = 0.f; }
  float dbq[2] = {0.f, 0.f}, dbk[2] = {0.f, 0.f}, dbv[2] = {0.f, 0.f};
  const float gmc[2] = {vec[m], vec[32 + m]}, btc[2] = {vec[64 + m], vec[96 + m]};
  if (s < slabs_n) {
    const unsigned lane_part = static_cast<unsigned>(16 * (lane & 15));
    const unsigned lane_off = static_cast<unsigned>(256 * (lane >> 4)) + lane_part;
    f32x4 nq[8], nx[8], nk[8], nv[8];
    float nmu, nrs;
    // which = 0: e + the rows' statistics, 1: dQ, 2: dK, 3: dV -- each stream is requested again as soon as its registers
    // are free, so the requests of the next slab run under this slab's three products
    auto issue = [&](int sl, int which) {
      unsigned off[8];
      const int left = A.M - sl * 32;
      const unsigned lim = static_cast<unsigned>((left < 32 ? left : 32) - 1) * 256u + lane_part;
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        off[p] = lane_off + 1024u * p < lim ? lane_off + 1024u * p : lim;
        if (which >= 2) off[p] = 2u * off[p] - lane_part;  // the same rows at a pitch of 128 floats
      }
      const long long o = static_cast<long long>(sl) * 32 * 64;
      if (which == 0) {
        const unsigned ro = 4u * static_cast<unsigned>(sl * 32 + m < A.M ? sl * 32 + m : A.M - 1);
        sb_issue_a(A.x + o, off, nx);
        sb_issue_word(A.mean, ro, nmu);
        sb_issue_word(A.rstd, ro, nrs);
      } else if (which == 1) {
        sb_issue_a(A.dQ + o, off, nq);
      } else if (which == 2) {
        sb_issue(A.dKV + 2 * o, off, nk);
      } else {
        sb_issue(A.dKV + 2 * o + 64, off, nv);
      }
    };
    issue(s, 0); issue(s, 1); issue(s, 2); issue(s, 3);
    sb_arrived4(nq, nx, nk, nv, nmu, nrs);
    for (;;) {
      const int left = A.M - s * 32;
      int sn = s + nw;
      const bool more = sn < slabs_n;
      sn = more ? sn : s;
      float xc[32], mu[16], rs[16];
      // the rows' statistics in the column layout: register q <-> rows 8 (q >> 2) + 4 h + (q & 3)
      if (lane < 32) {
        st[lane] = nmu;
        st[32 + lane] = lane < left ? nrs : 0.f;          // rows beyond the end: xhat = 0 -- and their dQ / dK / dV are zeroed below
      }
      sb_req_to_col(lds, lane, nx, xc);                   // (its wave syncs also publish the statistics)
      issue(sn, 0);
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(st + 8 * q4 + 4 * h);
        const f32x4 b = *reinterpret_cast<const f32x4*>(st + 32 + 8 * q4 + 4 * h);
#pragma unroll
        for (int e = 0; e < 4; ++e) { mu[4 * q4 + e] = a[e]; rs[4 * q4 + e] = b[e]; }
      }
      // rows beyond the end of a ragged last slab re-read the last row: their gradients must not count
      float live[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) live[q] = ((q & 3) + 8 * (q >> 2) + 4 * h) < left ? 1.f : 0.f;
      {
        float dc[32];
        sb_req_to_col(lds, lane, nq, dc);
        issue(sn, 1);
        if (left < 32) {
#pragma unroll
          for (int q = 0; q < 32; ++q) dc[q] *= live[q & 15];
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int q = 0; q < 16; ++q) dbq[t] += dc[16 * t + q];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const float q0 = (xc[q] - mu[q]) * rs[q] * gmc[0] + btc[0], q1 = (xc[16 + q] - mu[q]) * rs[q] * gmc[1] + btc[1];
          aq[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(dc[q], q0, aq[0][0], 0, 0, 0);
          aq[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(dc[q], q1, aq[0][1], 0, 0, 0);
          aq[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(dc[16 + q], q0, aq[1][0], 0, 0, 0);
          aq[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(dc[16 + q], q1, aq[1][1], 0, 0, 0);
        }
      }
      {
        float dc[32];
        sb_req_to_col(lds, lane, nk, dc);
        issue(sn, 2);
        if (left < 32) {
#pragma unroll
          for (int q = 0; q < 32; ++q) dc[q] *= live[q & 15];
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int q = 0; q < 16; ++q) dbk[t] += dc[16 * t + q];
        sb_dw_acc(dc, xc, ak);
      }
      {
        float dc[32];
        sb_req_to_col(lds, lane, nv, dc);
        issue(sn, 3);
        if (left < 32) {
#pragma unroll
          for (int q = 0; q < 32; ++q) dc[q] *= live[q & 15];
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int q = 0; q < 16; ++q) dbv[t] += dc[16 * t + q];
        sb_dw_acc(dc, xc, av);
      }
      sb_arrived4(nq, nx, nk, nv, nmu, nrs);
      if (!more) break;
      s = sn;
    }
  }
  __syncthreads();
  float* red = sb_lds;
  for (int w = 0; w < kSbBwdWaves; ++w) {
    if (wid == w) {
      sb_acc_to_lds(red, lane, aq, w == 0);
      sb_acc_to_lds(red + kSbW, lane, ak, w == 0);
      sb_acc_to_lds(red + 2 * kSbW, lane, av, w == 0);
      sb_colsum_to_lds(red + 3 * kSbW, lane, dbq, w == 0);
      sb_colsum_to_lds(red + 3 * kSbW + 64, lane, dbk, w == 0);
      sb_colsum_to_lds(red + 3 * kSbW + 128, lane, dbv, w == 0);
    }
    __syncthreads();
  }
  float* dst = A.part + static_cast<long long>(blockIdx.x) * kSbInPart;
  for (int i = threadIdx.x; i < kSbInPart; i += 64 * kSbBwdWaves) dst[i] = red[i];
}

// out segment j (offset seg_off[j], length seg_len[j]) = sum over the workgroups' partials, in order
struct SbReduceArgs {
  const float* part;
  int nparts, stride, nseg;
  int seg_off[8], seg_len[8];
  float* dst[8];
};
__global__ __launch_bounds__(256) void sb_reduce_kernel(const SbReduceArgs A) {
  // 32 outputs per workgroup, the partials dealt to 8 groups of lanes (group pg adds partials pg, pg + 8, ... in order),
  // the 8 group sums added in order: fixed association, 264 workgroups instead of 33 long chains of dependent adds
  __shared__ float red[8][32];
  const int o = threadIdx.x & 31, pg = threadIdx.x >> 5;
  const int i = static_cast<int>(blockIdx.x) * 32 + o;
  float acc = 0.f;
  if (i < A.stride)
    for (int p = pg; p < A.nparts; p += 8) acc += A.part[static_cast<long long>(p) * A.stride + i];
  red[pg][o] = acc;
  __syncthreads();
  if (pg != 0 || i >= A.stride) return;
  float* d = nullptr;
#pragma unroll
  for (int j = 0; j < 8; ++j)
    if (j < A.nseg && i >= A.seg_off[j] && i < A.seg_off[j] + A.seg_len[j] && A.dst[j] != nullptr) d = A.dst[j] + (i - A.seg_off[j]);
  if (d == nullptr) return;
  float t = red[0][o];
#pragma unroll
  for (int k = 1; k < 8; ++k) t += red[k][o];
  *d = t;
}

static int sb_bwd_grid(long long m) {
  const long long slabs = (m + 31) / 32;
  long long wgs = (slabs + kSbBwdWaves - 1) / kSbBwdWaves;
  if (wgs > kCUs) wgs = kCUs;
  return static_cast<int>(wgs < 1 ? 1 : wgs);
}

static bool sb_aligned(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

static int sb_grid(long long m) {
  const long long slabs = (m + 31) / 32;
  long long wgs = (slabs + kSbWaves - 1) / kSbWaves;
  if (wgs > kCUs) wgs = kCUs;
  return static_cast<int>(wgs < 1 ? 1 : wgs);
}

}  // namespace rbx

using namespace rbx;

extern "C" int rbx_seqblock_qkv_fwd(const float* d_x, int64_t m, const float* d_ln_w, const float* d_ln_b, float eps,
                                    const float* d_in_w, const float* d_in_b, float* d_mean, float* d_rstd, float* d_q,
                                    float* d_Q, float* d_KV, void* stream) {
  if (m < 0 || m > (1LL << 30)) return fail(RBX_ERR_INVALID, "rbx_seqblock_qkv_fwd: m = %lld", static_cast<long long>(m));
  if (m == 0) return RBX_OK;
  if (!d_x || !d_in_w || !d_mean || !d_rstd || !d_Q || !d_KV)
    return fail(RBX_ERR_INVALID, "rbx_seqblock_qkv_fwd: NULL operand");
  if (!sb_aligned(d_x) || !sb_aligned(d_q) || !sb_aligned(d_Q) || !sb_aligned(d_KV))
    return fail(RBX_ERR_UNSUPPORTED, "rbx_seqblock_qkv_fwd: activations must be 16-byte aligned");
  SbQkvArgs a{d_x, d_ln_w, d_ln_b, d_in_w, d_in_b, eps, d_mean, d_rstd, d_q, d_Q, d_KV, static_cast<int>(m)};
  const size_t lds = sizeof(float) * (3 * kSbW + 5 * 64 + kSbWaves * kSbSlab);
  static bool once = false;
  if (!once) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(sb_qkv_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            static_cast<int>(lds)) != hipSuccess)
      return fail(RBX_ERR_LAUNCH, "rbx_seqblock_qkv_fwd: %zu bytes of LDS refused", lds);
    once = true;
  }
  hipLaunchKernelGGL(sb_qkv_fwd_kernel, dim3(sb_grid(m)), dim3(64 * kSbWaves), lds, as_stream(stream), a);
  return check_launch("sb_qkv_fwd_kernel");
}

extern "C" int rbx_seqblock_ffn_fwd(const float* d_attn, const float* d_res, const float* d_wo, const float* d_bo, float* d_x,
                                    int64_t m, const float* d_ln_w, const float* d_ln_b, float eps, const float* d_w1,
                                    const float* d_b1, const float* d_w2, const float* d_b2, const float* d_keep,
                                    float* d_mean, float* d_rstd, float* d_n, float* d_h, float* d_out,
                                    const float* d_res_mean, const float* d_res_rstd, const float* d_res_ln_w,
                                    const float* d_res_ln_b, void* stream) {
  if (m < 0 || m > (1LL << 30)) return fail(RBX_ERR_INVALID, "rbx_seqblock_ffn_fwd: m = %lld", static_cast<long long>(m));
  if ((d_res_mean != nullptr) != (d_res_rstd != nullptr) || (d_res_mean != nullptr && d_attn == nullptr))
    return fail(RBX_ERR_INVALID, "rbx_seqblock_ffn_fwd: the rebuilt residual needs mean AND rstd, and the prologue");
  if (m == 0) return RBX_OK;
  if (!d_x || !d_w1 || !d_w2 || !d_mean || !d_rstd || !d_h || !d_out)
    return fail(RBX_ERR_INVALID, "rbx_seqblock_ffn_fwd: NULL operand");
  const bool pro = d_attn != nullptr;
  if (pro && (!d_res || !d_wo)) return fail(RBX_ERR_INVALID, "rbx_seqblock_ffn_fwd: the prologue needs d_res and d_wo");
  if (!sb_aligned(d_x) || !sb_aligned(d_h) || !sb_aligned(d_out) || !sb_aligned(d_n) || !sb_aligned(d_attn) ||
      !sb_aligned(d_res))
    return fail(RBX_ERR_UNSUPPORTED, "rbx_seqblock_ffn_fwd: activations must be 16-byte aligned");
  SbFfnArgs a{d_attn, d_res, d_wo, d_bo, d_x, d_ln_w, d_ln_b, d_w1, d_b1, d_w2, d_b2, d_keep, eps,
              d_mean, d_rstd, d_n, d_h, d_out, static_cast<int>(m), d_res_mean, d_res_rstd, d_res_ln_w, d_res_ln_b};
  const size_t lds = sizeof(float) * ((pro ? 3 : 2) * kSbW + 7 * 64 + kSbWaves * kSbSlab);
  static bool once[2] = {false, false};
  const void* fn = pro ? reinterpret_cast<const void*>(sb_ffn_fwd_kernel<true>) : reinterpret_cast<const void*>(sb_ffn_fwd_kernel<false>);
  if (!once[pro]) {
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)) != hipSuccess)
      return fail(RBX_ERR_LAUNCH, "rbx_seqblock_ffn_fwd: %zu bytes of LDS refused", lds);
    once[pro] = true;
  }
  if (pro) hipLaunchKernelGGL(sb_ffn_fwd_kernel<true>, dim3(sb_grid(m)), dim3(64 * kSbWaves), lds, as_stream(stream), a);
  else hipLaunchKernelGGL(sb_ffn_fwd_kernel<false>, dim3(sb_grid(m)), dim3(64 * kSbWaves), lds, as_stream(stream), a);
  return check_launch("sb_ffn_fwd_kernel");
}

extern "C" size_t rbx_seqblock_ffn_bwd_workspace_size(int64_t m) {
  return m <= 0 ? 0 : sizeof(float) * static_cast<size_t>(sb_bwd_grid(m)) * kSbFfnPart;
}

extern "C" int rbx_seqblock_ffn_bwd(const float* d_dout, const float* d_keep, const float* d_h, const float* d_x,
                                    const float* d_mean, const float* d_rstd, int64_t m, const float* d_ln_w,
                                    const float* d_ln_b, const float* d_w1, const float* d_w2, float* d_dx, float* d_dw1,
                                    float* d_db1, float* d_dw2, float* d_db2, float* d_dgamma, float* d_dbeta,
                                    void* d_workspace, size_t workspace_bytes, void* stream) {
  if (m < 0 || m > (1LL << 30)) return fail(RBX_ERR_INVALID, "rbx_seqblock_ffn_bwd: m = %lld", static_cast<long long>(m));
  if (m == 0) return RBX_OK;
  if (!d_dout || !d_h || !d_x || !d_mean || !d_rstd || !d_w1 || !d_w2 || !d_dx)
    return fail(RBX_ERR_INVALID, "rbx_seqblock_ffn_bwd: NULL operand");
  if (!sb_aligned(d_dout) || !sb_aligned(d_h) || !sb_aligned(d_x) || !sb_aligned(d_dx))
    return fail(RBX_ERR_UNSUPPORTED, "rbx_seqblock_ffn_bwd: activations must be 16-byte aligned");
  const size_t need = rbx_seqblock_ffn_bwd_workspace_size(m);
  if (d_workspace == nullptr || workspace_bytes < need)
    return fail(RBX_ERR_WORKSPACE, "rbx_seqblock_ffn_bwd: workspace %zu < %zu bytes", workspace_bytes, need);
  const int grid = sb_bwd_grid(m);
  SbFfnBwdArgs a{d_dout, d_keep, d_h, d_x, d_mean, d_rstd, d_ln_w, d_ln_b, d_w1, d_w2, d_dx, static_cast<float*>(d_workspace),
                 static_cast<int>(m)};
  const size_t lds = sizeof(float) * (2 * kSbW + 2 * 64 + 3 * kSbBwdWaves * kSbSlab);
  static bool attr = false;                 // (the other order of the four products measured 353 vs 339 us: profiles/r04/INDEX.md)
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(sb_ffn_bwd_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            static_cast<int>(lds)) != hipSuccess)
      return fail(RBX_ERR_LAUNCH, "rbx_seqblock_ffn_bwd: %zu bytes of LDS refused", lds);
    attr = true;
  }
  hipLaunchKernelGGL(sb_ffn_bwd_kernel<0>, dim3(grid), dim3(64 * kSbBwdWaves), lds, as_stream(stream), a);
  int rc = check_launch("sb_ffn_bwd_kernel");
  if (rc != RBX_OK) return rc;
  SbReduceArgs r{};
  r.part = static_cast<const float*>(d_workspace);
  r.nparts = grid;
  r.stride = kSbFfnPart;
  r.nseg = 6;
  const int offs[6] = {0, kSbW, 2 * kSbW, 2 * kSbW + 64, 2 * kSbW + 128, 2 * kSbW + 192};
  const int lens[6] = {kSbW, kSbW, 64, 64, 64, 64};
  float* dsts[6] = {d_dw2, d_dw1, d_db2, d_db1, d_dgamma, d_dbeta};
  for (int j = 0; j < 6; ++j) { r.seg_off[j] = offs[j]; r.seg_len[j] = lens[j]; r.dst[j] = dsts[j]; }
  hipLaunchKernelGGL(sb_reduce_kernel, dim3((kSbFfnPart + 31) / 32), dim3(256), 0, as_stream(stream), r);
  return check_launch("sb_reduce_kernel");
}

extern "C" size_t rbx_seqblock_attn_in_bwd_workspace_size(int64_t m) {
  return m <= 0 ? 0 : sizeof(float) * static_cast<size_t>(sb_grid(m)) * kSbLnPart;
}

extern "C" int rbx_seqblock_attn_in_bwd(const float* d_dQ, const float* d_dKV, const float* d_g, const float* d_x,
                                        const float* d_mean, const float* d_rstd, int64_t m, const float* d_ln_w,
                                        const float* d_in_w, const float* d_row_scale, float alpha, float* d_de,
                                        float* d_dgamma, float* d_dbeta, void* d_workspace, size_t workspace_bytes,
                                        void* stream) {
  if (m < 0 || m > (1LL << 30)) return fail(RBX_ERR_INVALID, "rbx_seqblock_attn_in_bwd: m = %lld", static_cast<long long>(m));
  if (m == 0) return RBX_OK;
  if (!d_dQ || !d_dKV || !d_g || !d_x || !d_mean || !d_rstd || !d_in_w || !d_de)
    return fail(RBX_ERR_INVALID, "rbx_seqblock_attn_in_bwd: NULL operand");
  if (!sb_aligned(d_dQ) || !sb_aligned(d_dKV) || !sb_aligned(d_g) || !sb_aligned(d_x) || !sb_aligned(d_de))
    return fail(RBX_ERR_UNSUPPORTED, "rbx_seqblock_attn_in_bwd: activations must be 16-byte aligned");
  const size_t need = rbx_seqblock_attn_in_bwd_workspace_size(m);
  if (d_workspace == nullptr || workspace_bytes < need)
    return fail(RBX_ERR_WORKSPACE, "rbx_seqblock_attn_in_bwd: workspace %zu < %zu bytes", workspace_bytes, need);
  const int grid = sb_grid(m);
  SbAttnInBwdArgs a{d_dQ, d_dKV, d_g, d_x, d_mean, d_rstd, d_ln_w, d_in_w, d_de, static_cast<float*>(d_workspace),
                    static_cast<int>(m), d_row_scale, alpha};
  const size_t lds = sizeof(float) * (3 * kSbW + 64 + kSbWaves * kSbSlab);
  static bool once = false;
  if (!once) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(sb_attn_in_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            static_cast<int>(lds)) != hipSuccess)
      return fail(RBX_ERR_LAUNCH, "rbx_seqblock_attn_in_bwd: %zu bytes of LDS refused", lds);
    once = true;
  }
  hipLaunchKernelGGL(sb_attn_in_bwd_kernel, dim3(grid), dim3(64 * kSbWaves), lds, as_stream(stream), a);
  int rc = check_launch("sb_attn_in_bwd_kernel");
  if (rc != RBX_OK) return rc;
  if (d_dgamma == nullptr && d_dbeta == nullptr) return RBX_OK;
  SbReduceArgs r{};
  r.part = static_cast<const float*>(d_workspace);
  r.nparts = grid;
  r.stride = kSbLnPart;
  r.nseg = 2;
  r.seg_off[0] = 0; r.seg_len[0] = 64; r.dst[0] = d_dgamma;
  r.seg_off[1] = 64; r.seg_len[1] = 64; r.dst[1] = d_dbeta;
  hipLaunchKernelGGL(sb_reduce_kernel, dim3((kSbLnPart + 31) / 32), dim3(256), 0, as_stream(stream), r);
  return check_launch("sb_reduce_kernel");
}

extern "C" size_t rbx_seqblock_attn_out_bwd_workspace_size(int64_t m) {
  return m <= 0 ? 0 : sizeof(float) * static_cast<size_t>(sb_grid(m)) * kSbOutPart;
}

extern "C" int rbx_seqblock_attn_out_bwd(const float* d_g, const float* d_O, int64_t m, const float* d_wo, float* d_dO,
                                         float* d_dwo, float* d_dbo, void* d_workspace, size_t workspace_bytes,
                                         void* stream) {
  if (m < 0 || m > (1LL << 30)) return fail(RBX_ERR_INVALID, "rbx_seqblock_attn_out_bwd: m = %lld", static_cast<long long>(m));
  if (m == 0) return RBX_OK;
  if (!d_g || !d_O || !d_wo || !d_dO) return fail(RBX_ERR_INVALID, "rbx_seqblock_attn_out_bwd: NULL operand");
  if (!sb_aligned(d_g) || !sb_aligned(d_O) || !sb_aligned(d_dO))
    return fail(RBX_ERR_UNSUPPORTED, "rbx_seqblock_attn_out_bwd: activations must be 16-byte aligned");
  const size_t need = rbx_seqblock_attn_out_bwd_workspace_size(m);
  if (d_workspace == nullptr || workspace_bytes < need)
    return fail(RBX_ERR_WORKSPACE, "rbx_seqblock_attn_out_bwd: workspace %zu < %zu bytes", workspace_bytes, need);
  const int grid = sb_grid(m);
  SbAttnOutBwdArgs a{d_g, d_O, d_wo, d_dO, static_cast<float*>(d_workspace), static_cast<int>(m)};
  const size_t lds = sizeof(float) * (kSbW + 64 + kSbWaves * kSbSlab);
  static bool once = false;
  if (!once) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(sb_attn_out_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            static_cast<int>(lds)) != hipSuccess)
      return fail(RBX_ERR_LAUNCH, "rbx_seqblock_attn_out_bwd: %zu bytes of LDS refused", lds);
    once = true;
  }
  hipLaunchKernelGGL(sb_attn_out_bwd_kernel, dim3(grid), dim3(64 * kSbWaves), lds, as_stream(stream), a);
  int rc = check_launch("sb_attn_out_bwd_kernel");
  if (rc != RBX_OK) return rc;
  if (d_dwo == nullptr && d_dbo == nullptr) return RBX_OK;
  SbReduceArgs r{};
  r.part = static_cast<const float*>(d_workspace);
  r.nparts = grid;
  r.stride = kSbOutPart;
  r.nseg = 2;
  r.seg_off[0] = 0; r.seg_len[0] = kSbW; r.dst[0] = d_dwo;
  r.seg_off[1] = kSbW; r.seg_len[1] = 64; r.dst[1] = d_dbo;
  hipLaunchKernelGGL(sb_reduce_kernel, dim3((kSbOutPart + 31) / 32), dim3(256), 0, as_stream(stream), r);
  return check_launch("sb_reduce_kernel");
}

extern "C" size_t rbx_seqblock_inproj_dw_workspace_size(int64_t m) {
  return m <= 0 ? 0 : sizeof(float) * static_cast<size_t>(sb_bwd_grid(m)) * kSbInPart;
}

extern "C" int rbx_seqblock_inproj_dw(const float* d_dQ, const float* d_dKV, const float* d_x, const float* d_mean,
                                      const float* d_rstd, int64_t m, const float* d_ln_w, const float* d_ln_b, float* d_dw,
                                      float* d_db, void* d_workspace, size_t workspace_bytes, void* stream) {
  if (m < 0 || m > (1LL << 30)) return fail(RBX_ERR_INVALID, "rbx_seqblock_inproj_dw: m = %lld", static_cast<long long>(m));
  if (m == 0) return RBX_OK;
  if (!d_dQ || !d_dKV || !d_x || !d_mean || !d_rstd) return fail(RBX_ERR_INVALID, "rbx_seqblock_inproj_dw: NULL operand");
  if (!sb_aligned(d_dQ) || !sb_aligned(d_dKV) || !sb_aligned(d_x))
    return fail(RBX_ERR_UNSUPPORTED, "rbx_seqblock_inproj_dw: activations must be 16-byte aligned");
  const size_t need = rbx_seqblock_inproj_dw_workspace_size(m);
  if (d_workspace == nullptr || workspace_bytes < need)
    return fail(RBX_ERR_WORKSPACE, "rbx_seqblock_inproj_dw: workspace %zu < %zu bytes", workspace_bytes, need);
  const int grid = sb_bwd_grid(m);
  SbInDwArgs a{d_dQ, d_dKV, d_x, d_mean, d_rstd, d_ln_w, d_ln_b, static_cast<float*>(d_workspace), static_cast<int>(m)};
  const size_t lds = sizeof(float) * (kSbInPart + kSbBwdWaves * kSbSlab);
  static bool once = false;
  if (!once) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(sb_inproj_dw_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            static_cast<int>(lds)) != hipSuccess)
      return fail(RBX_ERR_LAUNCH, "rbx_seqblock_inproj_dw: %zu bytes of LDS refused", lds);
    once = true;
  }
  hipLaunchKernelGGL(sb_inproj_dw_kernel, dim3(grid), dim3(64 * kSbBwdWaves), lds, as_stream(stream), a);
  int rc = check_launch("sb_inproj_dw_kernel");
  if (rc != RBX_OK) return rc;
  if (d_dw == nullptr && d_db == nullptr) return RBX_OK;
  SbReduceArgs r{};
  r.part = static_cast<const float*>(d_workspace);
  r.nparts = grid;
  r.stride = kSbInPart;
  r.nseg = 2;
  r.seg_off[0] = 0; r.seg_len[0] = 3 * kSbW; r.dst[0] = d_dw;            // [192, 64] = dWq | dWk | dWv, in_proj_weight's layout
  r.seg_off[1] = 3 * kSbW; r.seg_len[1] = 192; r.dst[1] = d_db;
  hipLaunchKernelGGL(sb_reduce_kernel, dim3((kSbInPart + 31) / 32), dim3(256), 0, as_stream(stream), r);
  return check_launch("sb_reduce_kernel");
}
