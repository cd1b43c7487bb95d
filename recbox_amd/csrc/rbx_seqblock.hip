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
  for (int idx = threadIdx.x; idx < kSbW; idx += 64 * kSbWaves) {
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
__device__ __forceinline__ void sb_arrived(f32x4 (&v)[8]) {
  asm volatile("s_waitcnt vmcnt(0)"
               : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7])
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
  for (;;) {
    float x[32];
    sb_arrived(nx);
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
    sb_store(lds, lane, x, A.q + static_cast<long long>(r0) * 64, off64, left);
    sb_gemm_row(wq, lane, x, y);
    sb_add_vec(y, vec + 128, h);
    sb_store(lds, lane, y, A.Q + static_cast<long long>(r0) * 64, off64, left);
    if (!more) break;
    s = sn;
  }
  sb_arrived(nx);
}

struct SbFfnArgs {
  // prologue (attn != NULL): x = res + attn Wo^T + bo is WRITTEN; without it x is the input
  const float *attn, *res, *wo, *bo;
  float* x;
  const float *ln_w, *ln_b, *w1, *b1, *w2, *b2, *keep;
  float eps;
  float *mean, *rstd, *n, *h, *out;           // n may be NULL (not stored)
  int M;
};

// [x = res + attn Wo^T + bo;]  n = LayerNorm(x); h = relu(n W1^T + b1); out = (n + h W2^T + b2) * keep[row]
// (sasrec.py:86-92, PointWiseFeedForward :119-123 without dropout)
template <bool PRO>
__global__ __launch_bounds__(64 * kSbWaves, 1) void sb_ffn_fwd_kernel(const SbFfnArgs A) {
  extern __shared__ float sb_lds[];
  float* w1 = sb_lds;
  float* w2 = w1 + kSbW;
  float* wo = w2 + kSbW;                       // (PRO only)
  float* vec = PRO ? wo + kSbW : wo;           // gamma, beta, b1, b2, bo
  float* slabs = vec + 5 * 64;
  sb_stage_weight(w1, A.w1, 64, false);
  sb_stage_weight(w2, A.w2, 64, false);
  if constexpr (PRO) sb_stage_weight(wo, A.wo, 64, false);
  sb_stage_vec(vec, A.ln_w, 1.f);
  sb_stage_vec(vec + 64, A.ln_b, 0.f);
  sb_stage_vec(vec + 128, A.b1, 0.f);
  sb_stage_vec(vec + 192, A.b2, 0.f);
  sb_stage_vec(vec + 256, PRO ? A.bo : nullptr, 0.f);
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
  f32x4 nx[8], nr[8];
  {
    const int left = A.M - s * 32;
#pragma unroll
    for (int p = 0; p < 8; ++p) off[p] = off64[p];
    if (left < 32) sb_offsets(64, left, lane, off);
    sb_issue(in0 + static_cast<long long>(s) * 32 * 64, off, nx);
    if constexpr (PRO) sb_issue(A.res + static_cast<long long>(s) * 32 * 64, off, nr);
  }
  for (;;) {
    float x[32];
    const int r0 = s * 32;
    const int left = A.M - r0;
    int sn = s + nw;
    const bool more = sn < slabs_n;
    sn = more ? sn : s;
    if constexpr (PRO) {
      float o[32];
      sb_arrived(nx);                          // (vmcnt(0): both tensors have arrived)
      sb_turn_in(lds, lane, nx, o);
      sb_turn_in(lds, lane, nr, x);
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
      sb_arrived(nx);
      sb_turn_in(lds, lane, nx, x);
      const int ln = A.M - sn * 32;
#pragma unroll
      for (int p = 0; p < 8; ++p) off[p] = off64[p];
      if (ln < 32) sb_offsets(64, ln, lane, off);
      sb_issue(in0 + static_cast<long long>(sn) * 32 * 64, off, nx);
    }
    const int rr = r0 + m;
    const float kp = A.keep != nullptr ? A.keep[rr < A.M ? rr : A.M - 1] : 1.f;
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
    sb_store(lds, lane, y, A.out + static_cast<long long>(r0) * 64, off64, left);
    if (!more) break;
    s = sn;
  }
  sb_arrived(nx);
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
  if (!d_x || !d_in_w || !d_mean || !d_rstd || !d_q || !d_Q || !d_KV)
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
                                    float* d_mean, float* d_rstd, float* d_n, float* d_h, float* d_out, void* stream) {
  if (m < 0 || m > (1LL << 30)) return fail(RBX_ERR_INVALID, "rbx_seqblock_ffn_fwd: m = %lld", static_cast<long long>(m));
  if (m == 0) return RBX_OK;
  if (!d_x || !d_w1 || !d_w2 || !d_mean || !d_rstd || !d_h || !d_out)
    return fail(RBX_ERR_INVALID, "rbx_seqblock_ffn_fwd: NULL operand");
  const bool pro = d_attn != nullptr;
  if (pro && (!d_res || !d_wo)) return fail(RBX_ERR_INVALID, "rbx_seqblock_ffn_fwd: the prologue needs d_res and d_wo");
  if (!sb_aligned(d_x) || !sb_aligned(d_h) || !sb_aligned(d_out) || !sb_aligned(d_n) || !sb_aligned(d_attn) ||
      !sb_aligned(d_res))
    return fail(RBX_ERR_UNSUPPORTED, "rbx_seqblock_ffn_fwd: activations must be 16-byte aligned");
  SbFfnArgs a{d_attn, d_res, d_wo, d_bo, d_x, d_ln_w, d_ln_b, d_w1, d_b1, d_w2, d_b2, d_keep, eps,
              d_mean, d_rstd, d_n, d_h, d_out, static_cast<int>(m)};
  const size_t lds = sizeof(float) * ((pro ? 3 : 2) * kSbW + 5 * 64 + kSbWaves * kSbSlab);
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
