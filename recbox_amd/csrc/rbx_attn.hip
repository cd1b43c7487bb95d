// rbx_attn.hip -- K6: fused masked softmax attention for short sequences (gfx950).
//
// Reference behaviour replaced:
//   first-party ScaledDotProductAttention.forward
//     (ranking/pytorch/layers/attentions/dot_product_attention.py:31-43: QK^T, /scale,
//      masked_fill(mask == 0, -1e9), softmax, .V; returns (output, attention))
//   the core of nn.MultiheadAttention inside rechub SASRec
//     (third_party/rechub/models/matching/sasrec.py:81-87: boolean causal mask -> -inf)
// and their autograd backward.  The [B, H, L, L] score matrix never reaches HBM unless the
// caller asks for the attention probabilities.
//
// One workgroup of 256 threads owns 256 query rows of one (sample, head): thread i owns query
// row i with q_i, o_i in registers and streams the keys, which pass through LDS in chunks of
// C rows (K and V of a chunk: 2 C head_dim floats <= 96 KB; C = 192 at head_dim 64) -- every
// lane of a wave reads the SAME k_j / v_j row (LDS broadcast, conflict free).  Any sequence
// length runs (the reference's nn.MultiheadAttention has no limit, sasrec.py:81-94): the
// online softmax simply continues over the chunks.  SASRec cfg 5 (L = 200, d = 64, no
// explicit mask) takes the matrix-core kernels of rbx_attn_mfma.hip instead.  Exact fp32 on the
// VALU: at fp32 the matrix cores run at the vector rate, and the 1e-4 parity bar rules out
// bf16.  Softmax is online over blocks of 8 keys (one rescale per block).
// Backward recomputes p from the saved log-sum-exp: phase A (thread = query row) produces
// dQ and D_i = <dO_i, O_i>; phase B (thread = key row, Q and dO staged in LDS) produces
// dK, dV -- no atomics, deterministic.
#include "rbx_internal.h"

namespace rbx {

constexpr int kAttnThreads = 256;
constexpr int kKeyBlock = 8;

// mask: optional float [BH, Lq, Lk]; an entry == 0 means "masked": score := fill
template <int HD>
__global__ __launch_bounds__(kAttnThreads) void attn_fwd_kernel(const float* __restrict__ Q, const float* __restrict__ K,
                                                                const float* __restrict__ V,
                                                                const float* __restrict__ mask, const int Lq,
                                                                const int Lk, const float scale, const int causal,
                                                                const float fill, float* __restrict__ O,
                                                                float* __restrict__ LSE, float* __restrict__ P,
                                                                const DropArgs drop, const int C) {
  extern __shared__ float lds[];
  unsigned dk0 = 0, dk1 = 0;
  if (drop.thr16 != 0) drop_seed(drop, &dk0, &dk1);
  float* Ks = lds;                 // [C][HD]: the current chunk of keys
  float* Vs = lds + C * HD;        // [C][HD]
  const long long bh = blockIdx.x;
  const float* kg = K + bh * Lk * HD;
  const float* vg = V + bh * Lk * HD;
  // stage keys [c0, c0 + C) (rows beyond Lk are never read); the barrier in front protects the previous chunk's readers
  auto stage = [&](int c0) {
    __syncthreads();
    const int n = ((Lk - c0 < C) ? Lk - c0 : C) * HD;
    for (int i = threadIdx.x * 4; i < n; i += kAttnThreads * 4) {
      *reinterpret_cast<float4*>(Ks + i) = *reinterpret_cast<const float4*>(kg + static_cast<long long>(c0) * HD + i);
      *reinterpret_cast<float4*>(Vs + i) = *reinterpret_cast<const float4*>(vg + static_cast<long long>(c0) * HD + i);
    }
    __syncthreads();
  };
  {
    const int i0 = blockIdx.y * kAttnThreads;
    const int i = i0 + threadIdx.x;
    const bool live = i < Lq;
    float q[HD], o[HD];
    const float* qg = Q + (bh * Lq + (live ? i : 0)) * HD;
#pragma unroll
    for (int d = 0; d < HD; d += 4) {
      const float4 t = *reinterpret_cast<const float4*>(qg + d);
      q[d] = t.x * scale; q[d + 1] = t.y * scale; q[d + 2] = t.z * scale; q[d + 3] = t.w * scale;
    }
#pragma unroll
    for (int d = 0; d < HD; ++d) o[d] = 0.f;
    float m = -INFINITY, l = 0.f;
    const int jmax = live ? (causal ? (i < Lk - 1 ? i : Lk - 1) : Lk - 1) : -1;
    // wave-uniform trip count so that the LDS reads stay broadcasts
    int jend = jmax;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const int other = __shfl_xor(jend, off, 64);
      jend = other > jend ? other : jend;
    }
    const float* mrow = (mask != nullptr && live) ? mask + (bh * Lq + i) * Lk : nullptr;
    // the last key any row of this WORKGROUP sees: every wavefront walks the same chunks (the staging barriers are
    // workgroup-wide), inside a chunk each stops at its own jend
    const int last = causal ? ((i0 + kAttnThreads - 1 < Lk - 1) ? i0 + kAttnThreads - 1 : Lk - 1) : Lk - 1;
    for (int c0 = 0; c0 <= last; c0 += C) {
    stage(c0);
    const int cend = (c0 + C - 1 < jend) ? c0 + C - 1 : jend;
    for (int j0 = c0; j0 <= cend; j0 += kKeyBlock) {
      float s[kKeyBlock];
      float mb = -INFINITY;
#pragma unroll
      for (int k = 0; k < kKeyBlock; ++k) {
        const int j = j0 + k;
        float acc = 0.f;
        if (j < Lk && j < c0 + C) {
          const float* kr = Ks + (j - c0) * HD;
#pragma unroll
          for (int d = 0; d < HD; ++d) acc += q[d] * kr[d];
        }
        if (mrow != nullptr && j <= jmax && mrow[j] == 0.f) acc = fill;
        s[k] = (j <= jmax && j < c0 + C) ? acc : -INFINITY;
        mb = fmaxf(mb, s[k]);
      }
      if (mb == -INFINITY) continue;                 // nothing visible in this block for this row
      const float mn = fmaxf(m, mb);
      const float alpha = __expf(m - mn);            // m == -inf -> 0
      l *= alpha;
#pragma unroll
      for (int d = 0; d < HD; ++d) o[d] *= alpha;
      unsigned dc[2][4];
      if (drop.thr16 != 0) {                         // two 2 x 4 blocks of the dropout mask cover the 8 keys of this step
        drop_block(static_cast<unsigned>(i) >> 2, static_cast<unsigned>(j0) >> 2, static_cast<unsigned long long>(bh),
                   (i & 3) >> 1, dk0, dk1, dc[0]);
        drop_block(static_cast<unsigned>(i) >> 2, (static_cast<unsigned>(j0) >> 2) + 1, static_cast<unsigned long long>(bh),
                   (i & 3) >> 1, dk0, dk1, dc[1]);
      }
#pragma unroll
      for (int k = 0; k < kKeyBlock; ++k) {
        const int j = j0 + k;
        float p = (s[k] == -INFINITY) ? 0.f : __expf(s[k] - mn);
        l += p;                                      // the normaliser is the undropped sum
        if (drop.thr16 != 0) p = drop_keep(dc[k >> 2], i & 1, k & 3, drop.thr16) ? p * drop.scale : 0.f;
        if (j < Lk && j < c0 + C) {
          const float* vr = Vs + (j - c0) * HD;
#pragma unroll
          for (int d = 0; d < HD; ++d) o[d] += p * vr[d];
        }
      }
      m = mn;
    }
    }
    const float invl = 1.0f / l;                     // l == 0 (fully -inf row) -> NaN like the reference
    if (live) {
      float* og = O + (bh * Lq + i) * HD;
#pragma unroll
      for (int d = 0; d < HD; d += 4)
        *reinterpret_cast<float4*>(og + d) = make_float4(o[d] * invl, o[d + 1] * invl, o[d + 2] * invl, o[d + 3] * invl);
      if (LSE != nullptr) LSE[bh * Lq + i] = m + __logf(l);
    }
    if (P != nullptr) {                              // attention probabilities requested: second sweep over the chunks
      for (int c0 = 0; c0 < Lk; c0 += C) {
        if (Lk > C) stage(c0);                       // (one chunk: it is still there)
        if (!live) continue;
        float* pg = P + (bh * Lq + i) * Lk;
        const int ce = (c0 + C < Lk) ? c0 + C : Lk;
        for (int j = c0; j < ce; ++j) {
          float acc = 0.f;
          const float* kr = Ks + (j - c0) * HD;
#pragma unroll
          for (int d = 0; d < HD; ++d) acc += q[d] * kr[d];
          if (mrow != nullptr && mrow[j] == 0.f) acc = fill;
          float pj = (j <= jmax) ? __expf(acc - m) * invl : 0.f;
          if (drop.thr16 != 0) {                     // the reference returns the DROPPED probabilities (dot_product_attention.py:40-43)
            unsigned c[4];
            drop_block(static_cast<unsigned>(i) >> 2, static_cast<unsigned>(j) >> 2, static_cast<unsigned long long>(bh),
                       (i & 3) >> 1, dk0, dk1, c);
            pj = drop_keep(c, i & 1, j & 3, drop.thr16) ? pj * drop.scale : 0.f;
          }
          pg[j] = pj;
        }
      }
    }
  }
}

// phase A: dQ and D = <dO, O>
template <int HD>
__global__ __launch_bounds__(kAttnThreads) void attn_bwd_dq_kernel(const float* __restrict__ Q, const float* __restrict__ K,
                                                                   const float* __restrict__ V,
                                                                   const float* __restrict__ mask,
                                                                   const float* __restrict__ O,
                                                                   const float* __restrict__ dO,
                                                                   const float* __restrict__ LSE, const int Lq,
                                                                   const int Lk, const float scale, const int causal,
                                                                   const float fill, float* __restrict__ dQ,
                                                                   float* __restrict__ Dv, const DropArgs drop, const int C) {
  extern __shared__ float lds[];
  unsigned dk0 = 0, dk1 = 0;
  if (drop.thr16 != 0) drop_seed(drop, &dk0, &dk1);
  float* Ks = lds;                 // [C][HD]: the current chunk of keys
  float* Vs = lds + C * HD;
  const long long bh = blockIdx.x;
  const float* kg = K + bh * Lk * HD;
  const float* vg = V + bh * Lk * HD;
  auto stage = [&](int c0) {
    __syncthreads();
    const int n = ((Lk - c0 < C) ? Lk - c0 : C) * HD;
    for (int i = threadIdx.x * 4; i < n; i += kAttnThreads * 4) {
      *reinterpret_cast<float4*>(Ks + i) = *reinterpret_cast<const float4*>(kg + static_cast<long long>(c0) * HD + i);
      *reinterpret_cast<float4*>(Vs + i) = *reinterpret_cast<const float4*>(vg + static_cast<long long>(c0) * HD + i);
    }
    __syncthreads();
  };
  {
    const int i0 = blockIdx.y * kAttnThreads;
    const int i = i0 + threadIdx.x;
    const bool live = i < Lq;
    const long long row = bh * Lq + (live ? i : 0);
    float q[HD], go[HD], dq[HD];
    float Di = 0.f;
#pragma unroll
    for (int d = 0; d < HD; ++d) {
      q[d] = Q[row * HD + d] * scale;
      go[d] = dO[row * HD + d];
      Di += go[d] * O[row * HD + d];
      dq[d] = 0.f;
    }
    const float lse = LSE[row];
    const int jmax = live ? (causal ? (i < Lk - 1 ? i : Lk - 1) : Lk - 1) : -1;
    int jend = jmax;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const int other = __shfl_xor(jend, off, 64);
      jend = other > jend ? other : jend;
    }
    const float* mrow = (mask != nullptr && live) ? mask + row * Lk : nullptr;
    unsigned dc[4] = {0u, 0u, 0u, 0u};
    const int last = causal ? ((i0 + kAttnThreads - 1 < Lk - 1) ? i0 + kAttnThreads - 1 : Lk - 1) : Lk - 1;
    for (int c0 = 0; c0 <= last; c0 += C) {
    stage(c0);
    const int cend = (c0 + C - 1 < jend) ? c0 + C - 1 : jend;
    for (int j = c0; j <= cend; ++j) {
      const float* kr = Ks + (j - c0) * HD;
      const float* vr = Vs + (j - c0) * HD;
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int d = 0; d < HD; ++d) {
        s += q[d] * kr[d];
        dp += go[d] * vr[d];
      }
      if (drop.thr16 != 0) {
        if ((j & 3) == 0)
          drop_block(static_cast<unsigned>(i) >> 2, static_cast<unsigned>(j) >> 2, static_cast<unsigned long long>(bh),
                     (i & 3) >> 1, dk0, dk1, dc);
        dp = drop_keep(dc, i & 1, j & 3, drop.thr16) ? dp * drop.scale : 0.f;
      }
      if (mrow != nullptr && j <= jmax && mrow[j] == 0.f) s = fill;
      const float p = (j <= jmax) ? __expf(s - lse) : 0.f;
      const float ds = p * (dp - Di);
#pragma unroll
      for (int d = 0; d < HD; ++d) dq[d] += ds * kr[d];
    }
    }
    if (live) {
#pragma unroll
      for (int d = 0; d < HD; ++d) dQ[row * HD + d] = dq[d] * scale;
      Dv[row] = Di;
    }
  }
}

// phase B: dK, dV (thread = key row; Q (pre-scaled) and dO staged in LDS)
template <int HD>
__global__ __launch_bounds__(kAttnThreads) void attn_bwd_dkv_kernel(const float* __restrict__ Q, const float* __restrict__ K,
                                                                    const float* __restrict__ V,
                                                                    const float* __restrict__ mask,
                                                                    const float* __restrict__ dO,
                                                                    const float* __restrict__ LSE,
                                                                    const float* __restrict__ Dv, const int Lq,
                                                                    const int Lk, const float scale, const int causal,
                                                                    const float fill, float* __restrict__ dK,
                                                                    float* __restrict__ dV, const DropArgs drop, const int C) {
  extern __shared__ float lds[];
  unsigned dk0 = 0, dk1 = 0;
  if (drop.thr16 != 0) drop_seed(drop, &dk0, &dk1);
  float* Qs = lds;                       // [C][HD]: the current chunk of queries, pre-scaled
  float* Gs = Qs + C * HD;               // [C][HD] dO
  float* Ls = Gs + C * HD;               // [C] lse
  float* Ds = Ls + C;                    // [C] D
  const long long bh = blockIdx.x;
  auto stage = [&](int c0) {
    __syncthreads();
    const int rows = (Lq - c0 < C) ? Lq - c0 : C;
    const long long base = (bh * Lq + c0) * HD;
    for (int i = threadIdx.x; i < rows * HD; i += kAttnThreads) {
      Qs[i] = Q[base + i] * scale;
      Gs[i] = dO[base + i];
    }
    for (int i = threadIdx.x; i < rows; i += kAttnThreads) {
      Ls[i] = LSE[bh * Lq + c0 + i];
      Ds[i] = Dv[bh * Lq + c0 + i];
    }
    __syncthreads();
  };
  {
    const int j0 = blockIdx.y * kAttnThreads;
    const int j = j0 + threadIdx.x;
    const bool live = j < Lk;
    const long long row = bh * Lk + (live ? j : 0);
    float kx[HD], vx[HD], dk[HD], dv[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) {
      kx[d] = K[row * HD + d];
      vx[d] = V[row * HD + d];
      dk[d] = 0.f;
      dv[d] = 0.f;
    }
    // causal: query i sees key j iff j <= i; the wave starts at its smallest j
    int istart = causal ? (live ? j : Lq) : 0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const int other = __shfl_xor(istart, off, 64);
      istart = other < istart ? other : istart;
    }
    unsigned dc[4] = {0u, 0u, 0u, 0u};
    int dc_for = -1;                     // the (i >> 1) the cached Philox block belongs to
    // the first query any key of this WORKGROUP is seen by: every wavefront walks the same chunks
    const int first = causal ? (j0 < Lq ? j0 : Lq) : 0;
    for (int c0 = first / C * C; c0 < Lq; c0 += C) {
    stage(c0);
    const int cend = (c0 + C < Lq) ? c0 + C : Lq;
    for (int i = (istart > c0 ? istart : c0); i < cend; ++i) {
      const float* qr = Qs + (i - c0) * HD;
      const float* gr = Gs + (i - c0) * HD;
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int d = 0; d < HD; ++d) {
        s += qr[d] * kx[d];
        dp += gr[d] * vx[d];
      }
      const bool vis = live && (!causal || j <= i);
      if (mask != nullptr && vis && mask[(bh * Lq + i) * Lk + j] == 0.f) s = fill;
      const float p = vis ? __expf(s - Ls[i - c0]) : 0.f;
      float pd = p;
      if (drop.thr16 != 0) {
        if ((i >> 1) != dc_for) {                  // a block covers two consecutive queries
          drop_block(static_cast<unsigned>(i) >> 2, static_cast<unsigned>(j) >> 2, static_cast<unsigned long long>(bh),
                     (i & 3) >> 1, dk0, dk1, dc);
          dc_for = i >> 1;
        }
        const bool keep = drop_keep(dc, i & 1, j & 3, drop.thr16);
        pd = keep ? p * drop.scale : 0.f;
        dp = keep ? dp * drop.scale : 0.f;
      }
      const float ds = p * (dp - Ds[i - c0]);
#pragma unroll
      for (int d = 0; d < HD; ++d) {
        dv[d] += pd * gr[d];
        dk[d] += ds * qr[d];
      }
    }
    }
    if (live) {
#pragma unroll
      for (int d = 0; d < HD; ++d) {
        dK[row * HD + d] = dk[d];
        dV[row * HD + d] = dv[d];
      }
    }
  }
}

// fast path on the fp32 matrix cores (rbx_attn_mfma.hip)
bool attn_mfma_supported(int lq, int lk, int hd, const float* mask, const float* probs);
int attn_mfma_fwd(const float* q, const float* k, const float* v, long long bh, int L, int hd, float scale, int causal,
                  float* o, float* lse, const DropArgs& drop, hipStream_t s);
int attn_mfma_bwd(const float* q, const float* k, const float* v, const float* o, const float* go, const float* lse,
                  long long bh, int L, int hd, float scale, int causal, float* dq, float* dk, float* dv, float* scratch,
                  const DropArgs& drop, hipStream_t s);

// p in [0, 1) -> 16-bit threshold and the scale of the kept entries (exactly 1 / (1 - thr16 / 65536))
static int make_drop(float p, uint64_t seed, const uint64_t* d_seed_add, DropArgs* d) {
  if (!(p >= 0.f) || p >= 1.f) return fail(RBX_ERR_INVALID, "attention: dropout probability %g not in [0, 1)", static_cast<double>(p));
  unsigned thr = static_cast<unsigned>(p * 65536.0f + 0.5f);
  if (thr > 65535u) thr = 65535u;
  d->thr16 = thr;
  d->scale = 65536.0f / static_cast<float>(65536u - thr);
  d->k0 = static_cast<unsigned>(seed);
  d->k1 = static_cast<unsigned>(seed >> 32);
  d->seed_add = reinterpret_cast<const unsigned long long*>(d_seed_add);
  return RBX_OK;
}

__global__ __launch_bounds__(256) void attn_dropout_mask_kernel(const long long total, const int lq, const int lk,
                                                                const DropArgs drop, unsigned char* __restrict__ out) {
  unsigned k0, k1;
  drop_seed(drop, &k0, &k1);
  const long long e = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;
  if (e >= total) return;
  const int j = static_cast<int>(e % lk);
  const long long t = e / lk;
  const int i = static_cast<int>(t % lq);
  const long long bh = t / lq;
  unsigned c[4];
  drop_block(static_cast<unsigned>(i) >> 2, static_cast<unsigned>(j) >> 2, static_cast<unsigned long long>(bh), (i & 3) >> 1, k0, k1, c);
  out[e] = drop_keep(c, i & 1, j & 3, drop.thr16) ? 1 : 0;
}

static int attn_check(int64_t bh, int lq, int lk, int hd) {
  if (bh < 0 || lq <= 0 || lk <= 0) return fail(RBX_ERR_INVALID, "attention: bad shape");
  if (hd != 4 && hd != 8 && hd != 16 && hd != 32 && hd != 64)
    return fail(RBX_ERR_UNSUPPORTED, "attention: head_dim %d not in {4,8,16,32,64}", hd);
  return RBX_OK;
}

// rows of the streamed operand per LDS chunk: as many as 96 KB hold (two [C, hd] tiles + two [C] vectors), a multiple of the
// forward's 8-key step, at most the operand itself
static int attn_chunk(int rows, int hd) {
  int c = static_cast<int>((96 * 1024) / (sizeof(float) * (2 * hd + 2))) / kKeyBlock * kKeyBlock;
  const int need = (rows + kKeyBlock - 1) / kKeyBlock * kKeyBlock;
  return c < need ? c : need;
}

}  // namespace rbx

#define RBX_ATTN_HD(hd, CALL) \
  switch (hd) {               \
    case 4: CALL(4); break;   \
    case 8: CALL(8); break;   \
    case 16: CALL(16); break; \
    case 32: CALL(32); break; \
    default: CALL(64); break; \
  }

extern "C" int rbx_attn_fwd(const float* d_q, const float* d_k, const float* d_v, const float* d_mask, int64_t bh,
                            int32_t lq, int32_t lk, int32_t head_dim, float scale, int32_t causal, float mask_fill,
                            float* d_o, float* d_lse, float* d_p, void* stream) {
  return rbx_attn_dropout_fwd(d_q, d_k, d_v, d_mask, bh, lq, lk, head_dim, scale, causal, mask_fill, 0.f, 0, nullptr, d_o,
                              d_lse, d_p, stream);
}

extern "C" int rbx_attn_dropout_mask(int64_t bh, int32_t lq, int32_t lk, float p_drop, uint64_t seed,
                                     const uint64_t* d_seed_add, uint8_t* d_keep, void* stream) {
  using namespace rbx;
  if (bh < 0 || lq <= 0 || lk <= 0 || d_keep == nullptr) return fail(RBX_ERR_INVALID, "attn_dropout_mask: bad arguments");
  DropArgs drop;
  int rc = make_drop(p_drop, seed, d_seed_add, &drop);
  if (rc != RBX_OK) return rc;
  const long long total = static_cast<long long>(bh) * lq * lk;
  if (total == 0) return RBX_OK;
  hipLaunchKernelGGL(attn_dropout_mask_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, as_stream(stream),
                     total, lq, lk, drop, d_keep);
  return check_launch("attn_dropout_mask_kernel");
}

extern "C" int rbx_attn_dropout_fwd(const float* d_q, const float* d_k, const float* d_v, const float* d_mask, int64_t bh,
                                    int32_t lq, int32_t lk, int32_t head_dim, float scale, int32_t causal, float mask_fill,
                                    float p_drop, uint64_t seed, const uint64_t* d_seed_add, float* d_o, float* d_lse,
                                    float* d_p, void* stream) {
  if (bh == 0) return RBX_OK;   // empty batch: nothing to do, pointers may be NULL
  using namespace rbx;
  int rc = attn_check(bh, lq, lk, head_dim);
  if (rc != RBX_OK) return rc;
  DropArgs drop;
  rc = make_drop(p_drop, seed, d_seed_add, &drop);
  if (rc != RBX_OK) return rc;
  if (d_q == nullptr || d_k == nullptr || d_v == nullptr || d_o == nullptr) return fail(RBX_ERR_INVALID, "attention: NULL tensor");
  if (bh == 0) return RBX_OK;
  if (d_lse != nullptr && attn_mfma_supported(lq, lk, head_dim, d_mask, d_p))
    return attn_mfma_fwd(d_q, d_k, d_v, bh, lq, head_dim, scale, causal, d_o, d_lse, drop, as_stream(stream));
  const int C = attn_chunk(lk, head_dim);
  const size_t lds = static_cast<size_t>(2) * C * head_dim * sizeof(float);
  const unsigned qb = static_cast<unsigned>((lq + kAttnThreads - 1) / kAttnThreads);
  hipStream_t s = as_stream(stream);
#define CALL(HD)                                                                                                  \
  do {                                                                                                            \
    if (lds > 64 * 1024)                                                                                          \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_kernel<HD>),                                    \
                          hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));                     \
    hipLaunchKernelGGL((attn_fwd_kernel<HD>), dim3(static_cast<unsigned>(bh), qb), dim3(kAttnThreads), lds, s, d_q, \
                       d_k, d_v, d_mask, lq, lk, scale, causal, mask_fill, d_o, d_lse, d_p, drop, C);             \
  } while (0)
  RBX_ATTN_HD(head_dim, CALL)
#undef CALL
  return check_launch("attn_fwd_kernel");
}

extern "C" int rbx_attn_bwd(const float* d_q, const float* d_k, const float* d_v, const float* d_mask,
                            const float* d_o, const float* d_do, const float* d_lse, int64_t bh, int32_t lq,
                            int32_t lk, int32_t head_dim, float scale, int32_t causal, float mask_fill, float* d_dq,
                            float* d_dk, float* d_dv, float* d_scratch, void* stream) {
  return rbx_attn_dropout_bwd(d_q, d_k, d_v, d_mask, d_o, d_do, d_lse, bh, lq, lk, head_dim, scale, causal, mask_fill, 0.f, 0,
                              nullptr, d_dq, d_dk, d_dv, d_scratch, stream);
}

extern "C" int rbx_attn_dropout_bwd(const float* d_q, const float* d_k, const float* d_v, const float* d_mask,
                                    const float* d_o, const float* d_do, const float* d_lse, int64_t bh, int32_t lq,
                                    int32_t lk, int32_t head_dim, float scale, int32_t causal, float mask_fill,
                                    float p_drop, uint64_t seed, const uint64_t* d_seed_add, float* d_dq, float* d_dk,
                                    float* d_dv, float* d_scratch, void* stream) {
  if (bh == 0) return RBX_OK;   // empty batch: nothing to do, pointers may be NULL
  using namespace rbx;
  int rc = attn_check(bh, lq, lk, head_dim);
  if (rc != RBX_OK) return rc;
  DropArgs drop;
  rc = make_drop(p_drop, seed, d_seed_add, &drop);
  if (rc != RBX_OK) return rc;
  if (d_q == nullptr || d_k == nullptr || d_v == nullptr || d_o == nullptr || d_do == nullptr || d_lse == nullptr ||
      d_dq == nullptr || d_dk == nullptr || d_dv == nullptr || d_scratch == nullptr)
    return fail(RBX_ERR_INVALID, "attention backward: NULL tensor");
  if (bh == 0) return RBX_OK;
  if (attn_mfma_supported(lq, lk, head_dim, d_mask, nullptr))
    return attn_mfma_bwd(d_q, d_k, d_v, d_o, d_do, d_lse, bh, lq, head_dim, scale, causal, d_dq, d_dk, d_dv, d_scratch,
                         drop, as_stream(stream));
  const int Ca = attn_chunk(lk, head_dim), Cb = attn_chunk(lq, head_dim);
  const size_t lds_a = static_cast<size_t>(2) * Ca * head_dim * sizeof(float);
  const size_t lds_b = (static_cast<size_t>(2) * Cb * head_dim + 2 * Cb) * sizeof(float);
  const unsigned qb = static_cast<unsigned>((lq + kAttnThreads - 1) / kAttnThreads);
  const unsigned kb = static_cast<unsigned>((lk + kAttnThreads - 1) / kAttnThreads);
  hipStream_t s = as_stream(stream);
#define CALL(HD)                                                                                                   \
  do {                                                                                                             \
    if (lds_a > 64 * 1024)                                                                                         \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dq_kernel<HD>),                                  \
                          hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_a));                    \
    if (lds_b > 64 * 1024)                                                                                         \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dkv_kernel<HD>),                                 \
                          hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_b));                    \
    hipLaunchKernelGGL((attn_bwd_dq_kernel<HD>), dim3(static_cast<unsigned>(bh), qb), dim3(kAttnThreads), lds_a, s, \
                       d_q, d_k, d_v, d_mask, d_o, d_do, d_lse, lq, lk, scale, causal, mask_fill, d_dq,            \
                       d_scratch, drop, Ca);                                                                       \
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<HD>), dim3(static_cast<unsigned>(bh), kb), dim3(kAttnThreads), lds_b, s, \
                       d_q, d_k, d_v, d_mask, d_do, d_lse, d_scratch, lq, lk, scale, causal, mask_fill, d_dk,      \
                       d_dv, drop, Cb);                                                                            \
  } while (0)
  RBX_ATTN_HD(head_dim, CALL)
#undef CALL
  return check_launch("attn_bwd kernels");
}
