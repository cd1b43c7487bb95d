// rbx_fm_fused.hip -- K1+K4 fused: the whole FM model body (embedding gather,
// first-order LR term, second-order interaction) in one forward kernel that never
// materialises [B, F, D], and its backward fused into the sorted segmented
// scatter-add (gfx950).
//
// Reference op sequence replaced (paths relative to /root/reference/recbox):
//   feature_emb = FeatureEmbedding(X)            ranking/pytorch/layers/embeddings/feature_embedding.py:188-214
//   lr_out = sum_f LR_f(X) + bias                ranking/pytorch/layers/blocks/logistic_regression.py:30-35
//   fm_out = 0.5 sum_d[(sum_f e)^2 - sum_f e^2]  ranking/pytorch/layers/interactions/inner_product.py:41-48
//   y = fm_out + lr_out                          ranking/pytorch/layers/blocks/factorization_machine.py:30-34
// and the autograd backward of all of it.
//
// Forward.  A lane group of D/4 lanes owns one sample and walks the F features; each
// lane keeps its float4 slice of S = sum_f e_f and Q = sum_f e_f^2 in registers, so
// the only cross-lane traffic is the final sum over d.  Feature descriptors are
// wave-uniform (scalar loads from the kernarg segment); features are processed 8 at a
// time: 8 id loads, then 8 independent row loads in flight per lane.  HBM traffic is
// the table rows + ids + 4 B/sample of logit + D*4 B/sample of S (kept for backward).
//
// Backward.  dL/de_f[b] = g_b (S_b - e_f[b]) and e_f[b] is the ROW w_r itself, so for a
// table row r hit by samples b_1..b_k:
//     dW[r]  = sum_j g_bj S_bj  -  w_r * sum_j g_bj          dW_lr[r] = sum_j g_bj
// The sorted (row, sample) pairs drive a segmented reduction whose per-lookup inputs are
// g[b] (4 B) and S[b] (D*4 B) -- 4.3 MB at B=65 536, D=16, L2-resident -- instead of a
// 163 MB dE tensor.  Numeric features x_f * w_f reduce over the batch:
//     dw_f = sum_b g x S_b - w_f sum_b g x^2,   dw_lr_f = sum_b g x,   dbias = sum_b g.
#include <stdlib.h>
#include "rbx_segreduce.h"
#include "rbx_tiera.h"

namespace rbx {

template <int G, int NV, bool VEC, int DT>
__global__ __launch_bounds__(256) void fm_fused_fwd_kernel(const FmPack P, const int F, const long long B, const int D,
                                                           const bool has_emb, const bool has_lr,
                                                           const float* __restrict__ bias,
                                                           const float* __restrict__ xe, const int n_extra,
                                                           const int x_stride, const int x_lr_off,
                                                           const int* __restrict__ xidx, const long long x_rows,
                                                           float* __restrict__ logit, float* __restrict__ prob,
                                                           float* __restrict__ ssum, int* __restrict__ status) {
  constexpr int W = VEC ? 4 : 1;
  constexpr int NA = NV * W;
  // features in flight per lane.  8 is the measured optimum at D=16 on MI355X: 16 in flight ran 1.6x
  // SLOWER (94 vs 58 us at B=65 536) -- the gather is bound by the random-row request rate, not latency.
  // More wavefronts do not help either: a sample's feature batches dealt to 2 / 4 neighbouring lane groups (2x / 4x
  // the wavefronts, partial sums joined by a shuffle) took 102 / 149 us instead of 46.
#define RBX_FM_U 8
  constexpr int U = (NA <= 4) ? RBX_FM_U : 4;
  const int lane_g = threadIdx.x % G;
  const long long ngroups = static_cast<long long>(gridDim.x) * (blockDim.x / G);
  for (long long b = static_cast<long long>(blockIdx.x) * (blockDim.x / G) + threadIdx.x / G; b < B; b += ngroups) {
    float s[NA], q[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) s[i] = q[i] = 0.f;
    float lr = 0.f;
    for (int f0 = 0; f0 < F; f0 += U) {
      long long id[U];
      float x[U];
      // phase 1: issue the batch's id / value loads, nothing consumes them yet
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const FmField& fd = P.f[(f0 + u < F) ? f0 + u : F - 1];
        if constexpr (DT >= 0) id[u] = fm_load_raw<DT>(fd.ids, b * fd.stride_b);
        else id[u] = load_raw(fd.ids, b * fd.stride_b, fd.dtype);
      }
      // phase 2: decode and range-check
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const FmField& fd = P.f[(f0 + u < F) ? f0 + u : F - 1];
        x[u] = 1.f;
        if constexpr (DT >= 0) {
          if (fd.kind == RBX_FIELD_CATEGORICAL) {
            int v;
            if (!fm_decode_id<DT>(id[u], fd.vocab, &v)) {
              if (status != nullptr && f0 + u < F) atomicOr(status, 1);
              v = 0;
              x[u] = 0.f;                       // out-of-range lookups read as zero rows
            }
            id[u] = v;
          } else {
            x[u] = fm_decode_value<DT>(id[u]);
            id[u] = 0;
          }
        } else if (fd.kind == RBX_FIELD_CATEGORICAL) {
          id[u] = decode_id(id[u], fd.dtype);
          if (id[u] < 0 || id[u] >= fd.vocab) {
            if (status != nullptr && f0 + u < F) atomicOr(status, 1);
            id[u] = 0;
            x[u] = 0.f;                         // out-of-range lookups read as zero rows
          }
        } else {
          x[u] = decode_value(id[u], fd.dtype);
          id[u] = 0;
        }
      }
      float e[U][NA];
      float l1[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int i = 0; i < NA; ++i) e[u][i] = 0.f;
        l1[u] = 0.f;
        if (f0 + u < F) {
          const FmField& fd = P.f[f0 + u];
          if (has_emb) {
            const float* row = (DT >= 0) ? fd.emb + static_cast<unsigned long long>(static_cast<unsigned>(id[u])) *
                                                        static_cast<unsigned>(fd.emb_stride)
                                         : fd.emb + id[u] * fd.emb_stride;
#pragma unroll
            for (int v = 0; v < NV; ++v) {
              const int d = (lane_g + v * G) * W;
              if (d < D) {
                if constexpr (VEC) {
                  const float4 t = *reinterpret_cast<const float4*>(row + d);
                  e[u][v * 4] = t.x; e[u][v * 4 + 1] = t.y; e[u][v * 4 + 2] = t.z; e[u][v * 4 + 3] = t.w;
                } else {
                  e[u][v] = row[d];
                }
              }
            }
          }
          if (has_lr && lane_g == ((f0 + u) % G))                                       // one lane per feature fetches the LR weight
            l1[u] = (DT >= 0) ? fd.lr[static_cast<unsigned long long>(static_cast<unsigned>(id[u])) * static_cast<unsigned>(fd.lr_stride)]
                              : fd.lr[id[u] * fd.lr_stride];
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
          // (explicitly rounded operations: the compiler may not contract them differently here and in rbx_fm_quad.hip,
          // whose results are tested bit for bit against this kernel's)
          const float t = mul_rn(e[u][i], x[u]);
          s[i] = add_rn(s[i], t);
          q[i] = fma_rn(t, t, q[i]);
        }
        lr = fma_rn(l1[u], x[u], lr);
      }
    }
    // rows that were fetched elsewhere (row-sharded tables: the owners sent them back).  Without an index the
    // sample's [n_extra, stride] block is contiguous (coalesced reads); with one, row (b, t) sits at wire
    // slot xidx[b, t] of the exchange buffer (slots outside [0, x_rows) = lookups that did not fit: zero rows),
    // which saves the un-permute pass over the buffer.  4 rows in flight per lane.
    for (int t0 = 0; t0 < n_extra; t0 += 4) {
      long long ridx[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int t = t0 + u;
        ridx[u] = -1;
        if (t < n_extra) ridx[u] = (xidx != nullptr) ? static_cast<long long>(xidx[b * n_extra + t]) : b * n_extra + t;
        if (xidx != nullptr && ridx[u] >= x_rows) ridx[u] = -1;
      }
      float e[4][NA];
      float l1[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
#pragma unroll
        for (int i = 0; i < NA; ++i) e[u][i] = 0.f;
        l1[u] = 0.f;
        if (ridx[u] >= 0) {
          const float* row = xe + ridx[u] * static_cast<long long>(x_stride);
          if (has_emb) {
#pragma unroll
            for (int v = 0; v < NV; ++v) {
              const int d = (lane_g + v * G) * W;
              if (d < D) {
                if constexpr (VEC) {
                  const float4 tt = *reinterpret_cast<const float4*>(row + d);     // stride % 4 == 0 is validated
                  e[u][v * 4] = tt.x; e[u][v * 4 + 1] = tt.y; e[u][v * 4 + 2] = tt.z; e[u][v * 4 + 3] = tt.w;
                } else {
                  e[u][v] = row[d];
                }
              }
            }
          }
          if (x_lr_off >= 0 && lane_g == ((t0 + u) % G)) l1[u] = row[x_lr_off];
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
          s[i] += e[u][i];
          q[i] += e[u][i] * e[u][i];
        }
        lr += l1[u];
      }
    }
    float fm = 0.f;
#pragma unroll
    for (int i = 0; i < NA; ++i) fm = fma_rn(fma_rn(s[i], s[i], -q[i]), 0.5f, fm);
    float total = group_sum<G>(add_rn(fm, lr));
    if (lane_g == 0) {
      const float z = total + (bias != nullptr ? bias[0] : 0.f);
      logit[b] = z;
      if (prob != nullptr) prob[b] = 1.f / (1.f + expf(-z));      // the model's y_pred = sigmoid(logit), same pass
    }
    if (has_emb && ssum != nullptr) {
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int d = (lane_g + v * G) * W;
        if (d < D) {
          if constexpr (VEC) {
            *reinterpret_cast<float4*>(ssum + b * D + d) = make_float4(s[v * 4], s[v * 4 + 1], s[v * 4 + 2], s[v * 4 + 3]);
          } else {
            ssum[b * D + d] = s[v];
          }
        }
      }
    }
  }
}

// ---- backward policy: contribution g_b * S_b, cnt += g_b; flush dW = A - cnt*w ------------
#define RBX_ABL 0        // ablation mask for measurements only (profiles/scripts): 1 no LR-gradient store, 2 no w_r read,
struct FmPolicy {
  static constexpr bool kHasCount = true;
  struct Args {
    const float* g;        // [B] upstream grad of the logit
    const float* ssum;     // [B, D]
    int D;
    int accumulate;        // 0: grads pre-zeroed -> store; 1: read-modify-write
  };
  template <class F>
  static __device__ __forceinline__ void contribute(const Args& a, const RedField& fd, unsigned local, int lane_g,
                                                    F& frag, float& cnt) {
    const float g = a.g[local];
    cnt = g;
    if (!(RBX_ABL & 4) && a.ssum != nullptr) frag.fma_from(a.ssum + static_cast<size_t>(local) * a.D, fd.dim, lane_g, g);
  }
  template <class F>
  static __device__ __forceinline__ void prefetch(const Args&, const RedField& fd, unsigned row, int lane_g, F& pre) {
    if (!(RBX_ABL & 2) && fd.grad != nullptr)
      pre.add_from_nt(fd.table + static_cast<size_t>(row) * fd.table_stride, fd.dim, lane_g);   // w_r
  }
  // the two-phase form of contribute / prefetch (segment_reduce_kernel): loads only, then frag *= weight(w)
  template <class F>
  static __device__ __forceinline__ void fetch(const Args& a, const RedField& fd, unsigned local, int lane_g, F& frag, float& w) {
    w = a.g[local];
    if (!(RBX_ABL & 4) && a.ssum != nullptr) frag.load_from(a.ssum + static_cast<size_t>(local) * a.D, fd.dim, lane_g);
    else frag.zero();
  }
  static __device__ __forceinline__ float weight(const Args&, float w) { return w; }
  template <class F>
  static __device__ __forceinline__ void prefetch_raw(const Args&, const RedField& fd, unsigned row, int lane_g, F& pre) {
    if (!(RBX_ABL & 2) && fd.grad != nullptr)
      pre.load_from_nt(fd.table + static_cast<size_t>(row) * fd.table_stride, fd.dim, lane_g);   // w_r
  }
  template <class F>
  static __device__ __forceinline__ void flush(const Args& a, const RedField& fd, unsigned row, const F& acc, float cnt,
                                               const F& pre, int lane_g) {
    if (!(RBX_ABL & 8) && fd.grad != nullptr) {
      F out = acc;                                               // A - cnt * w_r
#pragma unroll
      for (int q = 0; q < static_cast<int>(sizeof(out.a) / sizeof(float)); ++q) out.a[q] -= cnt * pre.a[q];
      float* dst = fd.grad + static_cast<size_t>(row) * fd.dim;
      if (a.accumulate) out.accumulate_into(dst, fd.dim, lane_g); else out.store_nt(dst, fd.dim, lane_g);
    }
    if (!(RBX_ABL & 1) && fd.grad2 != nullptr && lane_g == 0) {
      if (a.accumulate) fd.grad2[row] += cnt; else fd.grad2[row] = cnt;
    }
  }
};

// ---- numeric features + bias: batch reductions ------------------------------------------------
struct FmNumField {          // 48 B
  const void* ids;
  const float* w;            // emb weight [D] (or NULL)
  float* gw;                 // grad of emb weight [D] (or NULL)
  float* glr;                // grad of LR weight [1] (or NULL)
  long long stride_b;
  int dtype;
  int reserved;
};
struct FmNumPack { FmNumField f[RBX_MAX_FIELDS]; };

// samples staged per workgroup: as many as fit 48 KB of LDS, at most 128 (D = 16 with 13 numeric features: 128 samples,
// 22 KB, 512 workgroups at B = 65 536; the 256 threads split a sample's features between them), at least 32
static int fm_num_samples(int D, int n_num) {
  int ns = 128;
  while (ns > 32 && static_cast<size_t>(ns) * (D + 2 * n_num + 1) * sizeof(float) > 48 * 1024) ns >>= 1;
  return ns;
}

// partial layout: output o = feature * (D+2) + slot, slots = [D] sum g x S_d | [1] sum g x^2 | [1] sum g x; feature index
// n_num holds sum g in slot 0 (bias).  Stored [output][block], so that the final kernel's lanes read neighbouring floats.
// A workgroup stages `ns` samples (S rows, g*x and x per numeric feature, g) in LDS, then
// every thread owns one output and sums over the samples in a fixed order: this is a
// [n_num, ns] x [ns, D] product per workgroup, deterministic and sync-free after staging.
__global__ __launch_bounds__(256) void fm_numeric_partial_kernel(const FmNumPack P, const int n_num, const long long B,
                                                                 const int D, const int ns,
                                                                 const float* __restrict__ g,
                                                                 const float* __restrict__ ssum,
                                                                 float* __restrict__ partial, const bool vec_s) {
  extern __shared__ float lds[];
  const int nsp = ns + 1;                   // row pitch of the per-feature arrays (staging writes walk the features)
  float* sS = lds;                          // [ns][D]
  float* sgx = sS + ns * D;                 // [n_num][ns + 1]
  float* sx = sgx + n_num * nsp;            // [n_num][ns + 1]
  float* sg = sx + n_num * nsp;             // [ns]
  const long long b0 = static_cast<long long>(blockIdx.x) * ns;
  const int live = static_cast<int>((B - b0 < ns) ? (B - b0) : ns);
  // Staging: every thread ISSUES all of its loads before it touches the first value (raw loads, converted afterwards) --
  // written as load / convert / store per element this was 15 dependent round trips to memory per thread and the kernel
  // took 23 us for 11 MB.
  constexpr int U = 8;
  if (ssum != nullptr) {
    const float* src = ssum + b0 * D;
    if (vec_s) {                                      // D % 4 == 0 and S is 16-byte aligned: float4
      const int n4 = ns * D / 4, live4 = live * D / 4;
      for (int base = 0; base < n4; base += U * 256) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int i = base + u * 256 + threadIdx.x;
          v[u] = (i < live4) ? reinterpret_cast<const float4*>(src)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int i = base + u * 256 + threadIdx.x;
          if (i < n4) reinterpret_cast<float4*>(sS)[i] = v[u];
        }
      }
    } else {
      for (int i = threadIdx.x; i < ns * D; i += blockDim.x) sS[i] = (i < live * D) ? src[i] : 0.f;
    }
  }
  // (sample, feature) pairs dealt to the threads with the FEATURE fastest: the numeric columns of the reference's batch
  // tensor sit next to each other in a sample's row, so a wavefront's 64 values come from ~5 rows (a dozen cache lines)
  // instead of 64 rows
  for (int i = threadIdx.x; i < ns; i += blockDim.x) sg[i] = (i < live) ? g[b0 + i] : 0.f;
  for (int base = 0; base < ns * n_num; base += U * 256) {
    long long raw[U];
    float gb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int idx = base + u * 256 + threadIdx.x;
      const int i = idx / n_num, f = idx - i * n_num;
      raw[u] = 0;
      gb[u] = 0.f;
      if (idx < ns * n_num && i < live) {
        const FmNumField& fd = P.f[f];
        raw[u] = load_raw(fd.ids, (b0 + i) * fd.stride_b, fd.dtype);
        gb[u] = g[b0 + i];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int idx = base + u * 256 + threadIdx.x;
      const int i = idx / n_num, f = idx - i * n_num;
      if (idx < ns * n_num) {
        const float x = (i < live) ? decode_value(raw[u], P.f[f].dtype) : 0.f;
        sx[f * nsp + i] = x;
        sgx[f * nsp + i] = gb[u] * x;
      }
    }
  }
  __syncthreads();
  const int stride = D + 2;
  const int n_out = n_num * stride + 1;
  float* out = partial + blockIdx.x;
  const size_t nb = gridDim.x;
  for (int o = threadIdx.x; o < n_out; o += blockDim.x) {
    const int f = o / stride, slot = o - f * stride;
    // four interleaved partial sums (samples i, i+1, i+2, i+3 of every group of four), combined in a fixed order: the
    // dependent add chain is a quarter as long and the LDS reads of four samples are in flight together
    float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
    if (f == n_num) {
      for (int i = 0; i < ns; i += 4) { t0 += sg[i]; t1 += sg[i + 1]; t2 += sg[i + 2]; t3 += sg[i + 3]; }
    } else if (slot < D) {
      if (ssum != nullptr) {
        const float* a = sgx + f * nsp;
        const float* c = sS + slot;
        for (int i = 0; i < ns; i += 4) {
          t0 += a[i] * c[i * D];
          t1 += a[i + 1] * c[(i + 1) * D];
          t2 += a[i + 2] * c[(i + 2) * D];
          t3 += a[i + 3] * c[(i + 3) * D];
        }
      }
    } else if (slot == D) {
      const float* a = sgx + f * nsp;
      const float* c = sx + f * nsp;
      for (int i = 0; i < ns; i += 4) { t0 += a[i] * c[i]; t1 += a[i + 1] * c[i + 1]; t2 += a[i + 2] * c[i + 2]; t3 += a[i + 3] * c[i + 3]; }
    } else {
      const float* a = sgx + f * nsp;
      for (int i = 0; i < ns; i += 4) { t0 += a[i]; t1 += a[i + 1]; t2 += a[i + 2]; t3 += a[i + 3]; }
    }
    out[static_cast<size_t>(o) * nb] = (t0 + t1) + (t2 + t3);
  }
}

// one wavefront per (feature, slot): lanes stride over the workgroup partials, fixed shuffle tree.
// grid (n_num + 1, D + 2); scratch[f] keeps sum g x^2 for the w-correction applied by slot < D blocks,
// so the correction term is recomputed per block (cheap) instead of synchronising blocks.
__global__ __launch_bounds__(64) void fm_numeric_final_kernel(const FmNumPack P, const int n_num, const int D,
                                                              const unsigned num_blocks,
                                                              const float* __restrict__ partial,
                                                              float* __restrict__ dbias, const bool store) {
  const int f = blockIdx.x, slot = blockIdx.y;
  const int stride = D + 2;
  auto total = [&](int sl) -> float {
    float t = 0.f;
    const float* src = partial + static_cast<size_t>(f * stride + sl) * num_blocks;
    for (unsigned k = threadIdx.x; k < num_blocks; k += 64) t += src[k];
    return group_sum<64>(t);
  };
  if (f == n_num) {
    if (slot != 0) return;
    const float tg = total(0);
    if (dbias != nullptr && threadIdx.x == 0) dbias[0] = store ? tg : dbias[0] + tg;
    return;
  }
  const FmNumField& fd = P.f[f];
  if (slot == D) return;                          // sum g x^2 is only a correction term (read below)
  if (slot == D + 1) {
    const float t1 = total(D + 1);
    if (fd.glr != nullptr && threadIdx.x == 0) fd.glr[0] = store ? t1 : fd.glr[0] + t1;
    return;
  }
  if (fd.gw == nullptr) return;
  const float a = total(slot), t2 = total(D);
  if (threadIdx.x == 0) {
    const float v = a - fd.w[slot] * t2;
    fd.gw[slot] = store ? v : fd.gw[slot] + v;
  }
}

// The numeric partials in the form tier A's launch computes them (rbx_tiera.h: ta_numeric_block), as a launch of their own:
// for the callers that run the numeric reductions apart from tier A (ahead of a sort still in flight, no tier-A table at
// all).  Same units, same summation order: the two placements leave the same bits.
__global__ __launch_bounds__(512) void fm_numeric_blocks_kernel(const TaNumPack N, const long long B, const int D,
                                                                const float* __restrict__ g,
                                                                const float* __restrict__ ssum,
                                                                float* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) float num_stage[];
  ta_numeric_block(N, blockIdx.x, gridDim.x, B, g, ssum, D, num_stage, partial);
}

// d row(b,t) = [ g_b (S_b - e[b,t,:]) | ... g_b at the LR slot ... | 0 ]  in the rows' own packed layout
// (these rows belong to another rank: the gradient block is sent back to it as is)
__global__ __launch_bounds__(256) void fm_extra_bwd_kernel(const float* __restrict__ g, const float* __restrict__ ssum,
                                                           const float* __restrict__ x, const long long B,
                                                           const int n_extra, const int D, const int stride,
                                                           const int lr_off, const bool has_emb,
                                                           const int* __restrict__ xidx, const long long x_rows,
                                                           float* __restrict__ dx) {
  const long long total = B * n_extra * stride;
  const long long step = static_cast<long long>(gridDim.x) * blockDim.x;
  for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += step) {
    const long long bt = i / stride;
    const int c = static_cast<int>(i - bt * stride);
    const long long b = bt / n_extra;
    long long at = i;                              // position of this float in x / dx
    if (xidx != nullptr) {
      const long long r = xidx[bt];
      if (r < 0 || r >= x_rows) continue;          // lookup without a wire slot: no gradient
      at = r * stride + c;
    }
    float v = 0.f;
    if (has_emb && c < D) v = g[b] * (ssum[b * D + c] - x[at]);
    else if (c == lr_off) v = g[b];
    dx[at] = v;
  }
}

// ---- host side -------------------------------------------------------------------------------
static bool fm_fast_dtype() { return true; }   // (the generic decode for every call was the A/B arm: profiles/r02)

struct FmHost {
  int F = 0, D = 0;
  bool has_emb = false, has_lr = false, vec = false;
  int uniform_dt = -1;         // the ids dtype every feature shares, -1 when they differ
  FmPack pack;
};

static int fm_validate(const rbx_field_t* emb, const rbx_field_t* lr, int n, int64_t B, FmHost* h) {
  if (emb == nullptr && lr == nullptr) return fail(RBX_ERR_INVALID, "fm: both field arrays are NULL");
  if (n <= 0 || n > RBX_MAX_FIELDS) return fail(RBX_ERR_INVALID, "fm: n_fields=%d not in [1,%d]", n, RBX_MAX_FIELDS);
  h->F = n;
  h->has_emb = emb != nullptr;
  h->has_lr = lr != nullptr;
  h->D = h->has_emb ? emb[0].dim : 1;
  h->vec = h->has_emb && (h->D % 4 == 0);
  const rbx_field_t* lead = h->has_emb ? emb : lr;
  for (int i = 0; i < n; ++i) {
    const rbx_field_t& a = lead[i];
    if (a.kind != RBX_FIELD_CATEGORICAL && a.kind != RBX_FIELD_NUMERIC)
      return fail(RBX_ERR_UNSUPPORTED, "fm: feature %d: only categorical / numeric features can be fused", i);
    if (a.seq_len != 1 || a.pool != RBX_POOL_NONE)
      return fail(RBX_ERR_UNSUPPORTED, "fm: feature %d: sequence features are not fused (use the layer path)", i);
    if (a.ids == nullptr) return fail(RBX_ERR_INVALID, "fm: feature %d: ids is NULL", i);
    if (a.ids_dtype < RBX_I32 || a.ids_dtype > RBX_F64) return fail(RBX_ERR_INVALID, "fm: feature %d: bad ids_dtype", i);
    if (a.kind == RBX_FIELD_CATEGORICAL && (a.vocab <= 0 || a.vocab > INT_MAX))
      return fail(RBX_ERR_INVALID, "fm: feature %d: bad vocab", i);
    FmField& k = h->pack.f[i];
    k.ids = a.ids;
    k.stride_b = a.ids_stride_b;
    k.vocab = static_cast<int>(a.vocab);
    k.dtype = static_cast<unsigned char>(a.ids_dtype);
    k.kind = static_cast<unsigned char>(a.kind);
    k.r0 = k.r1 = 0;
    k.emb = nullptr;
    k.lr = nullptr;
    k.emb_stride = h->D;
    k.lr_stride = 1;
    if (h->has_emb) {
      if (emb[i].dim != h->D) return fail(RBX_ERR_UNSUPPORTED, "fm: all embedding dims must be equal to fuse");
      if (emb[i].table == nullptr) return fail(RBX_ERR_INVALID, "fm: feature %d: table is NULL", i);
      if ((reinterpret_cast<uintptr_t>(emb[i].table) & 15) != 0) h->vec = false;
      k.emb = emb[i].table;
      if (a.kind == RBX_FIELD_CATEGORICAL && emb[i].table_stride != 0) {
        if (emb[i].table_stride < h->D || emb[i].table_stride > INT_MAX)
          return fail(RBX_ERR_INVALID, "fm: feature %d: table_stride %lld < dim %d", i, (long long)emb[i].table_stride, h->D);
        k.emb_stride = static_cast<int>(emb[i].table_stride);
        if (k.emb_stride % 4 != 0) h->vec = false;
      }
    }
    if (h->has_lr) {
      const rbx_field_t& l = lr[i];
      if (l.dim != 1 || l.table == nullptr) return fail(RBX_ERR_INVALID, "fm: feature %d: LR tables must have dim 1", i);
      if (h->has_emb && (l.ids != a.ids || l.kind != a.kind || l.vocab != a.vocab || l.ids_stride_b != a.ids_stride_b))
        return fail(RBX_ERR_INVALID, "fm: feature %d: LR and embedding descriptors must describe the same ids", i);
      k.lr = l.table;
      if (l.kind == RBX_FIELD_CATEGORICAL && l.table_stride != 0) {
        if (l.table_stride < 1 || l.table_stride > INT_MAX) return fail(RBX_ERR_INVALID, "fm: feature %d: bad LR table_stride", i);
        k.lr_stride = static_cast<int>(l.table_stride);
      }
    }
  }
  h->uniform_dt = lead[0].ids_dtype;
  for (int i = 1; i < n; ++i)
    if (lead[i].ids_dtype != h->uniform_dt) h->uniform_dt = -1;
  (void)B;
  return RBX_OK;
}

template <int G, int NV, bool VEC>
static int launch_fm_fwd(const FmHost& h, int64_t B, const float* bias, const float* xe, int n_extra, int x_stride,
                         int x_lr_off, const int* xidx, long long x_rows, float* logit, float* prob, float* ssum,
                         int* status, hipStream_t s) {
  const int gpb = 256 / G;
  long long blocks = (B + gpb - 1) / gpb;
  if (blocks > kCUs * 16) blocks = kCUs * 16;
#define RBX_FM_LAUNCH(DT)                                                                                              \
  hipLaunchKernelGGL((fm_fused_fwd_kernel<G, NV, VEC, DT>), dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s, h.pack,   \
                     h.F, static_cast<long long>(B), h.D, h.has_emb, h.has_lr, bias, xe, n_extra, x_stride, x_lr_off, xidx, \
                     x_rows, logit, prob, ssum, status)
  // the reference's loader hands over one float64 [B, cols] tensor; int64 ids are the other common case
  if (h.uniform_dt == RBX_F64 && fm_fast_dtype()) RBX_FM_LAUNCH(RBX_F64);
  else if (h.uniform_dt == RBX_I64 && fm_fast_dtype()) RBX_FM_LAUNCH(RBX_I64);
  else RBX_FM_LAUNCH(-1);
#undef RBX_FM_LAUNCH
  return check_launch("fm_fused_fwd_kernel");
}

template <bool VEC>
static int dispatch_fm_fwd(const FmHost& h, int64_t B, const float* bias, const float* xe, int n_extra, int x_stride,
                           int x_lr_off, const int* xidx, long long x_rows, float* logit, float* prob, float* ssum,
                           int* status, hipStream_t s) {
  const int units = VEC ? h.D / 4 : h.D;
  switch (pow2_ceil(units)) {
    case 1: return launch_fm_fwd<1, 1, VEC>(h, B, bias, xe, n_extra, x_stride, x_lr_off, xidx, x_rows, logit, prob, ssum, status, s);
    case 2: return launch_fm_fwd<2, 1, VEC>(h, B, bias, xe, n_extra, x_stride, x_lr_off, xidx, x_rows, logit, prob, ssum, status, s);
    case 4: return launch_fm_fwd<4, 1, VEC>(h, B, bias, xe, n_extra, x_stride, x_lr_off, xidx, x_rows, logit, prob, ssum, status, s);
    case 8: return launch_fm_fwd<8, 1, VEC>(h, B, bias, xe, n_extra, x_stride, x_lr_off, xidx, x_rows, logit, prob, ssum, status, s);
    case 16: return launch_fm_fwd<16, 1, VEC>(h, B, bias, xe, n_extra, x_stride, x_lr_off, xidx, x_rows, logit, prob, ssum, status, s);
    case 32: return launch_fm_fwd<32, 1, VEC>(h, B, bias, xe, n_extra, x_stride, x_lr_off, xidx, x_rows, logit, prob, ssum, status, s);
    case 64: return launch_fm_fwd<64, 1, VEC>(h, B, bias, xe, n_extra, x_stride, x_lr_off, xidx, x_rows, logit, prob, ssum, status, s);
    default: return fail(RBX_ERR_UNSUPPORTED, "fm: embedding dim %d too large to fuse", h.D);
  }
}

// Which tables take the sort-free path of rbx_tiera.h.  The rule looks at the tables only (rows, dim), never at the
// batch: a persistent gradient buffer is cleared by the PREVIOUS step's plan (rbx_fm_rezero), which must agree with this
// step's about who writes which table in full.  Tables are admitted in ascending row count while the gradients of the
// admitted ones stay under 4 MB (the partial arrays are that times the number of 2048-sample blocks).
static int fm_tier_a_max_vocab() {
  // 4096 rows: measured at the Criteo shape (profiles/r03), 2200 / 4096 / 16384 give 0.257 / 0.251 / 0.253 ms per step -- a
  // table of 5 000-15 000 rows costs as much either way (block partials that hardly merge vs L2-resident rows in the sorted
  // path), and the partial arrays of the 16384 setting are 3x the workspace
  // (round 6, with the large tables' chain bounding the step: 2200 / 4096 / 6000 / 13000 / 16384 rows give 0.191 / 0.190 /
  //  0.204 / 0.207 / 0.213 ms on uniform ids and 0.234 / 0.227 / 0.231 / 0.222 / 0.215 on Zipf-like ids:
  //  profiles/r06/fm_tier_a_row_limit.txt)
  return 4096 < kTaMaxVocab ? 4096 : kTaMaxVocab;
}

// plan over the categorical features (keys from `lead`, grads from emb and lr): the tables of tier A go to `ta`, the
// others to the sort plan `p`, whose id columns are the rows of the compact id matrix (patched in by fm_bind_cid once
// the workspace is known)
static int fm_plan(const rbx_field_t* emb, const rbx_field_t* lr, int n, int64_t B, BwdPlan* p, FmNumPack* np,
                   int* n_num, TaPlan* ta, int* cid_of_key, int* key_src = nullptr, int* tab_src = nullptr) {
  rbx_field_t tmp[RBX_MAX_FIELDS];
  const rbx_field_t* lead = (emb != nullptr) ? emb : lr;
  *n_num = 0;
  int n_cat = 0;
  int cat_src[RBX_MAX_FIELDS];
  const int D = emb ? emb[0].dim : 1;
  *ta = TaPlan();
  ta->D = D;
  ta->has_emb = emb != nullptr;
  for (int i = 0; i < n; ++i) {
    if (lead[i].kind == RBX_FIELD_NUMERIC) {
      FmNumField& f = np->f[(*n_num)++];
      f.ids = lead[i].ids;
      f.stride_b = lead[i].ids_stride_b;
      f.dtype = lead[i].ids_dtype;
      f.w = emb ? emb[i].table : nullptr;
      f.gw = emb ? emb[i].grad : nullptr;
      f.glr = lr ? lr[i].grad : nullptr;
      f.reserved = 0;
      continue;
    }
    // a feature whose tables are all frozen still needs no sort entry
    float* g1 = emb ? emb[i].grad : nullptr;
    float* g2 = lr ? lr[i].grad : nullptr;
    if (g1 == nullptr && g2 == nullptr) continue;
    cat_src[n_cat++] = i;
  }
  // ---- tier split over the tables (features that share a table go together) ----
  bool in_a[RBX_MAX_FIELDS];
  for (int c = 0; c < n_cat; ++c) in_a[c] = false;
  // (a call without embedding tables -- the LogisticRegression layer of a layer-composed model -- keeps the one-sort form:
  //  its pairs can then be a copy of the FeatureEmbedding lookup's, rbx_sort_share)
  const int vmax = (emb != nullptr && D <= kTaMaxDim && B > 0) ? fm_tier_a_max_vocab() : 0;
  int tab_first[RBX_MAX_FIELDS], n_tabs = 0, tab_of[RBX_MAX_FIELDS];
  for (int c = 0; c < n_cat; ++c) {
    const rbx_field_t& a = lead[cat_src[c]];
    int hit = -1;
    for (int t = 0; t < n_tabs; ++t)
      if (lead[cat_src[tab_first[t]]].table == a.table) hit = t;
    if (hit < 0) { hit = n_tabs; tab_first[n_tabs++] = c; }
    tab_of[c] = hit;
  }
  bool tab_a[RBX_MAX_FIELDS];
  {
    int order[RBX_MAX_FIELDS];
    for (int t = 0; t < n_tabs; ++t) { order[t] = t; tab_a[t] = false; }
    for (int x = 1; x < n_tabs; ++x)               // insertion sort by (vocab, first field): stable, tiny
      for (int y = x; y > 0 && lead[cat_src[tab_first[order[y]]]].vocab < lead[cat_src[tab_first[order[y - 1]]]].vocab; --y) {
        const int t = order[y]; order[y] = order[y - 1]; order[y - 1] = t;
      }
    long long budget = 1ll << 20;                   // floats of gradient the admitted tables may hold
    for (int x = 0; x < n_tabs; ++x) {
      const int t = order[x];
      const rbx_field_t& a = lead[cat_src[tab_first[t]]];
      if (a.vocab > vmax) break;
      bool same = true;                             // (features of one table must agree on it, as make_plan demands)
      for (int c = 0; c < n_cat; ++c)
        if (tab_of[c] == t && lead[cat_src[c]].vocab != a.vocab) same = false;
      const long long cost = static_cast<long long>(a.vocab) * (D + 1);
      if (!same || cost > budget) continue;
      budget -= cost;
      tab_a[t] = true;
    }
    for (int c = 0; c < n_cat; ++c) in_a[c] = tab_a[tab_of[c]];
  }
  // ---- the compact id matrix: one row per categorical feature with a gradient (only when some table is in tier A;
  //      otherwise the sort reads the id columns where they are, as rbx_embed_sort does) ----
  bool tiered = false;
  for (int t = 0; t < n_tabs; ++t) tiered = tiered || tab_a[t];
  ta->n_cid = tiered ? n_cat : 0;
  bool strided = false;
  for (int c = 0; c < n_cat; ++c) {
    const rbx_field_t& a = lead[cat_src[c]];
    if (a.seq_len != 1 || a.pool != RBX_POOL_NONE) return fail(RBX_ERR_UNSUPPORTED, "fm: sequence features are not fused");
    if (a.ids == nullptr) return fail(RBX_ERR_INVALID, "fm: feature %d: ids is NULL", cat_src[c]);
    if (a.ids_dtype < RBX_I32 || a.ids_dtype > RBX_F64) return fail(RBX_ERR_INVALID, "fm: feature %d: bad ids_dtype", cat_src[c]);
    if (a.vocab <= 0 || a.vocab > INT_MAX) return fail(RBX_ERR_INVALID, "fm: feature %d: bad vocab", cat_src[c]);
    CidField& cf = ta->cid.f[c];
    cf.ids = a.ids;
    cf.stride_b = a.ids_stride_b;
    cf.vocab = static_cast<int>(a.vocab);
    cf.dtype = a.ids_dtype;
    if (a.ids_stride_b != 1) strided = true;
  }
  ta->field_fast = strided;
  ta->cid_ts = 32;                                 // samples per tile of compact_ids_kernel: a power of two, tile <= 2048 ids
  while (ta->cid_ts < 256 && 2 * ta->cid_ts * n_cat <= 256 * kCidPerThread) ta->cid_ts *= 2;
  ta->NB = static_cast<unsigned>((B + kTaBlock - 1) / kTaBlock);
  // ---- tier A descriptors: fields grouped by table ----
  ta->vec = emb != nullptr && (D % 4 == 0);
  for (int t = 0; t < n_tabs; ++t) {
    if (!tab_a[t]) continue;
    const int i0 = cat_src[tab_first[t]];
    if (tab_src != nullptr) tab_src[ta->n_tab] = i0;
    TaTable& tb = ta->tab.t[ta->n_tab++];
    tb.grad = emb ? emb[i0].grad : nullptr;
    tb.grad2 = lr ? lr[i0].grad : nullptr;
    tb.table = emb ? emb[i0].table : nullptr;
    tb.stride = (emb && emb[i0].table_stride != 0) ? static_cast<int>(emb[i0].table_stride) : D;
    tb.vocab = static_cast<int>(lead[i0].vocab);
    tb.pad = (lead[i0].padding_idx == RBX_NO_ID || lead[i0].padding_idx < INT_MIN || lead[i0].padding_idx > INT_MAX)
                 ? kNoId : static_cast<int>(lead[i0].padding_idx);
    tb.f_begin = static_cast<short>(ta->n_fld);
    tb.f_count = 0;
    tb.row0 = ta->rows;
    tb.reserved = 0;
    ta->rows += static_cast<unsigned>(tb.vocab);
    if (tb.stride % 4 != 0) ta->vec = false;
    if (tb.grad != nullptr && (reinterpret_cast<uintptr_t>(tb.grad) & 15) != 0) ta->vec = false;
    if (tb.table != nullptr && (reinterpret_cast<uintptr_t>(tb.table) & 15) != 0) ta->vec = false;
    for (int c = 0; c < n_cat; ++c) {
      if (tab_of[c] != t) continue;
      TaField& fd = ta->fld.f[ta->n_fld++];
      fd.cid_row = c;
      fd.vocab = tb.vocab;
      fd.frow0 = ta->frows;
      fd.fword0 = ta->fwords;
      ta->frows += static_cast<unsigned>(tb.vocab);
      ta->fwords += static_cast<unsigned>((tb.vocab + 31) / 32);
      ++tb.f_count;
    }
  }
  ta_layout(ta, B);
  // ---- tier B: the sort plan ----
  int n_b = 0;
  int b_src[RBX_MAX_FIELDS];
  for (int c = 0; c < n_cat; ++c) {
    if (in_a[c]) continue;
    const int i = cat_src[c];
    tmp[n_b] = lead[i];
    float* g1 = emb ? emb[i].grad : nullptr;
    float* g2 = lr ? lr[i].grad : nullptr;
    tmp[n_b].grad = g1 ? g1 : g2;            // make_plan skips fields without a grad
    tmp[n_b].out_off = 0;
    tmp[n_b].table_stride = 0;               // (the plan's keys do not depend on the storage; the stride is set below)
    b_src[n_b] = i;
    cid_of_key[n_b] = c;
    if (key_src != nullptr) key_src[n_b] = i;
    ++n_b;
  }
  const int ns = fm_num_samples(D, *n_num);
  if (n_b == 0) {
    *p = BwdPlan();
    p->n_lookups = 0;
    p->bytes = 256;
    p->num_blocks = static_cast<unsigned>((B + ns - 1) / ns);
    return RBX_OK;
  }
  int rc = make_plan(tmp, n_b, B, nullptr, 0, p, /*extra_dim=*/4);
  if (rc != RBX_OK) return rc;
  if (p->n_cat != n_b) return fail(RBX_ERR_INVALID, "fm: internal plan mismatch");
  p->vec = emb != nullptr && (D % 4 == 0);
  for (int c = 0; c < n_b; ++c) {
    const int i = b_src[c];
    RedField& rf = p->red.f[c];
    rf.grad = emb ? emb[i].grad : nullptr;
    rf.grad2 = lr ? lr[i].grad : nullptr;
    rf.table = emb ? emb[i].table : nullptr;
    rf.dim = static_cast<short>(D);
    rf.table_stride = (emb && emb[i].table_stride != 0) ? static_cast<int>(emb[i].table_stride) : D;
    if (rf.table_stride % 4 != 0) p->vec = false;          // float4 row reads need 16-byte aligned rows
    if (rf.grad != nullptr && (reinterpret_cast<uintptr_t>(rf.grad) & 15) != 0) p->vec = false;
    if (rf.table != nullptr && (reinterpret_cast<uintptr_t>(rf.table) & 15) != 0) p->vec = false;
  }
  p->max_dim = D;
  p->num_blocks = static_cast<unsigned>((B + ns - 1) / ns);
  // (the long fix-up launch keeps the plan's 2 x CUs workgroups: through round 5 the tiered plan launched 32 -- "runs of
  //  hundreds of pairs come from the small tables" -- which holds for uniform ids only: Zipf-like ids leave ~300 long chains in
  //  the large tables, ten per workgroup in a row, 56 us at the end of the step's critical chain.  Workgroups without a
  //  chain now leave at once, so the empty launch of a uniform batch costs the same: profiles/r06/fm_long_fixup_cap.txt)
  return RBX_OK;
}

static size_t fm_num_bytes(const BwdPlan& p, int n_num, int D) {
  return static_cast<size_t>(p.num_blocks) * (n_num + 1) * (D + 2) * sizeof(float) + 256;
}

// Everything one fused FM backward needs, derived from the descriptor arrays alone.  Workspace layout:
//   [ sort plan of tier B: p.bytes | numeric partials: fm_num_bytes | tier A region (compact ids first): ta.bytes ]
struct FmFull {
  BwdPlan p;
  FmNumPack np;
  TaPlan ta;
  int n_num = 0;
  int D = 1;
  int cid_of_key[RBX_MAX_FIELDS];
  size_t off_ta = 0, bytes = 0;
};

static int fm_full_plan(const rbx_field_t* emb, const rbx_field_t* lr, int n, int64_t B, FmFull* f) {
  if (emb == nullptr && lr == nullptr) return fail(RBX_ERR_INVALID, "fm: both field arrays are NULL");
  if (n <= 0 || n > RBX_MAX_FIELDS) return fail(RBX_ERR_INVALID, "fm: n_fields=%d not in [1,%d]", n, RBX_MAX_FIELDS);
  int rc = fm_plan(emb, lr, n, B, &f->p, &f->np, &f->n_num, &f->ta, f->cid_of_key);
  if (rc != RBX_OK) return rc;
  f->D = emb ? emb[0].dim : 1;
  f->off_ta = ta_align(f->p.bytes + fm_num_bytes(f->p, f->n_num, f->D));
  f->bytes = f->off_ta + f->ta.bytes;
  return RBX_OK;
}

// what rbx_fm_sparse_update (rbx_optim.hip) needs of a call's plan: the tier B sort plan, the tier A description, where the
// tier A region starts, the workspace size, and which feature of the caller's arrays every plan slot / tier A table is
int fm_update_plan(const rbx_field_t* emb, const rbx_field_t* lr, int n, int64_t B, BwdPlan* p, TaPlan* ta, size_t* off_ta,
                   size_t* bytes, int* src_of_key, int* src_of_tab) {
  if (emb == nullptr && lr == nullptr) return fail(RBX_ERR_INVALID, "fm: both field arrays are NULL");
  if (n <= 0 || n > RBX_MAX_FIELDS) return fail(RBX_ERR_INVALID, "fm: n_fields=%d not in [1,%d]", n, RBX_MAX_FIELDS);
  FmNumPack np;
  int n_num = 0, cid_of_key[RBX_MAX_FIELDS];
  const int rc = fm_plan(emb, lr, n, B, p, &np, &n_num, ta, cid_of_key, src_of_key, src_of_tab);
  if (rc != RBX_OK) return rc;
  const int D = emb ? emb[0].dim : 1;
  *off_ta = ta_align(p->bytes + fm_num_bytes(*p, n_num, D));
  *bytes = *off_ta + ta->bytes;
  return RBX_OK;
}

// the sort plan reads its ids from the compact matrix inside THIS workspace
static void fm_bind_cid(FmFull* f, char* ws, int64_t B) {
  if (f->ta.n_cid == 0) return;
  const int* cid = reinterpret_cast<const int*>(ws + f->off_ta + f->ta.off_cid);
  for (int c = 0; c < f->p.n_cat; ++c) {
    KeyField& kf = f->p.keys.f[c];
    kf.ids = cid + static_cast<size_t>(f->cid_of_key[c]) * static_cast<size_t>(B);
    kf.stride_b = 1;
    kf.stride_l = 0;
    kf.dtype = RBX_I32;
  }
}


}  // namespace rbx

extern "C" int rbx_fm_rezero(const rbx_field_t* emb, const rbx_field_t* lr, int32_t n_fields, int64_t batch,
                             void* d_workspace, size_t workspace_bytes, void* stream) {
  using namespace rbx;
  if (batch == 0) return RBX_OK;
  FmFull f;
  int rc = fm_full_plan(emb, lr, n_fields, batch, &f);
  if (rc != RBX_OK) return rc;
  // (tier-A tables are written in full by every backward: nothing to clear)
  if (f.p.n_lookups == 0) return RBX_OK;
  if (d_workspace == nullptr || workspace_bytes < f.bytes) return fail(RBX_ERR_WORKSPACE, "fm_rezero: workspace too small");
  return launch_rezero(f.p, static_cast<const char*>(d_workspace), as_stream(stream));
}

extern "C" int rbx_fm_fwd(const rbx_field_t* emb, const rbx_field_t* lr, int32_t n_fields, int64_t batch,
                          const float* d_lr_bias, const float* d_extra, int32_t n_extra, int32_t extra_stride,
                          int32_t extra_lr_off, const int32_t* d_extra_index, int64_t extra_rows, float* d_logit,
                          float* d_prob, float* d_sum, int32_t* d_status, void* stream) {
  using namespace rbx;
  FmHost h;
  int rc = fm_validate(emb, lr, n_fields, batch, &h);
  if (rc != RBX_OK) return rc;
  if (batch < 0) return fail(RBX_ERR_INVALID, "negative batch");
  if (batch == 0) return RBX_OK;
  if (d_logit == nullptr) return fail(RBX_ERR_INVALID, "d_logit is NULL");
  if (h.vec && d_sum != nullptr && (reinterpret_cast<uintptr_t>(d_sum) & 15) != 0) h.vec = false;
  if (n_extra < 0 || (n_extra > 0 && d_extra == nullptr)) return fail(RBX_ERR_INVALID, "fm: n_extra=%d without rows", n_extra);
  if (n_extra > 0) {
    const int need = (h.has_emb ? h.D : 0);
    if (extra_stride < need || extra_lr_off >= extra_stride || (extra_lr_off >= 0 && extra_lr_off < need))
      return fail(RBX_ERR_INVALID, "fm: extra rows: stride %d / lr offset %d do not fit dim %d", extra_stride,
                  extra_lr_off, need);
    if (d_extra_index != nullptr && extra_rows <= 0) return fail(RBX_ERR_INVALID, "fm: indexed extra rows need extra_rows > 0");
    if (h.vec && (extra_stride % 4 != 0 || (reinterpret_cast<uintptr_t>(d_extra) & 15) != 0)) h.vec = false;
  }
  if (n_extra == 0 && h.vec) {          // every feature a column of one batch tensor, dim 16: rbx_fm_quad.hip
    rc = fm_quad_fwd(h.pack, h.F, h.D, h.has_emb, h.has_lr, fm_fast_dtype() ? h.uniform_dt : -1, batch, d_lr_bias, d_logit,
                     d_prob, d_sum, d_status, as_stream(stream));
    if (rc != 1) return rc;
  }
  return h.vec ? dispatch_fm_fwd<true>(h, batch, d_lr_bias, d_extra, n_extra, extra_stride, extra_lr_off, d_extra_index,
                                       extra_rows, d_logit, d_prob, d_sum, d_status, as_stream(stream))
               : dispatch_fm_fwd<false>(h, batch, d_lr_bias, d_extra, n_extra, extra_stride, extra_lr_off, d_extra_index,
                                        extra_rows, d_logit, d_prob, d_sum, d_status, as_stream(stream));
}

extern "C" size_t rbx_fm_bwd_workspace_size(const rbx_field_t* emb, const rbx_field_t* lr, int32_t n_fields,
                                            int64_t batch) {
  using namespace rbx;
  FmFull f;
  if (fm_full_plan(emb, lr, n_fields, batch, &f) != RBX_OK) return 0;
  return f.bytes;
}

extern "C" int rbx_fm_sort_phases(const rbx_field_t* emb, const rbx_field_t* lr, int32_t n_fields, int64_t batch,
                                  void* d_workspace, size_t workspace_bytes, int32_t* d_status, int32_t phases,
                                  void* stream) {
  using namespace rbx;
  if (batch <= 0) return RBX_OK;
  FmFull f;
  int rc = fm_full_plan(emb, lr, n_fields, batch, &f);
  if (rc != RBX_OK) return rc;
  if (f.ta.n_cid == 0 && f.p.n_lookups == 0) return RBX_OK;
  if (d_workspace == nullptr || workspace_bytes < f.bytes) return fail(RBX_ERR_WORKSPACE, "fm: workspace too small");
  char* ws = static_cast<char*>(d_workspace);
  hipStream_t s = as_stream(stream);
  if (phases & 1) {      // ids of every dtype / stride -> int32 [feature][B], range-checked once; both tiers read that
    rc = ta_launch_compact(f.ta, batch, ws + f.off_ta, d_status, s);
    if (rc != RBX_OK) return rc;
  }
  if (phases & 4) {      // tier A: per-block sorts in LDS
    rc = ta_launch_blocksort(f.ta, batch, ws + f.off_ta, s);
    if (rc != RBX_OK) return rc;
  }
  if ((phases & 2) && f.p.n_lookups > 0) {      // tier B: global segmented radix sort (a one-tier call reads the id columns where they are)
    fm_bind_cid(&f, ws, batch);
    rc = run_sort(f.p, ws, f.ta.n_cid == 0 ? d_status : nullptr, s);
    if (rc != RBX_OK) return rc;
  }
  return RBX_OK;
}

extern "C" int rbx_fm_sort(const rbx_field_t* emb, const rbx_field_t* lr, int32_t n_fields, int64_t batch,
                           void* d_workspace, size_t workspace_bytes, int32_t* d_status, void* stream) {
  return rbx_fm_sort_phases(emb, lr, n_fields, batch, d_workspace, workspace_bytes, d_status, 7, stream);
}

namespace rbx {

// Two plans sort the same (key, val) pairs when they look up the same id tensors with the same key construction
// (table order and sizes, padding / mask ids, sequence lengths): e.g. the embedding tables and the dim-1 LR tables of
// one batch.  Member-wise: the packs are not zero-initialised.
static bool same_pairs(const BwdPlan& a, const BwdPlan& b) {
  if (a.n_cat != b.n_cat || a.n_lookups != b.n_lookups || a.total_rows != b.total_rows || a.passes != b.passes ||
      a.radix_bits != b.radix_bits || a.n_tiles != b.n_tiles) return false;
  for (int i = 0; i < a.n_cat; ++i) {
    const KeyField& x = a.keys.f[i];
    const KeyField& y = b.keys.f[i];
    if (x.ids != y.ids || x.stride_b != y.stride_b || x.stride_l != y.stride_l || x.vocab != y.vocab ||
        x.mask_id != y.mask_id || x.pad_id != y.pad_id || x.row_base != y.row_base || x.lk_off != y.lk_off ||
        x.seq_len != y.seq_len || x.dtype != y.dtype || x.pool != y.pool)
      return false;
  }
  return true;
}

__global__ __launch_bounds__(256) void copy_pairs_kernel(const unsigned* __restrict__ sk, const unsigned* __restrict__ sv,
                                                         unsigned* __restrict__ dk, unsigned* __restrict__ dv,
                                                         const unsigned n, unsigned* __restrict__ fin) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {          // what build_keys_kernel does for a sort of its own
    fin[0] = 0;
    fin[1] = 0;
    fin[2] = 0;
  }
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    dk[i] = sk[i];
    dv[i] = sv[i];
  }
}

static int plan_of(const rbx_field_t* a, const rbx_field_t* b, int n, int is_fm, int64_t B, BwdPlan* p) {
  if (is_fm) {
    // a tiered fused FM sort is more than sorted pairs (compact ids, block sorts of the small tables) and its keys live
    // in its own workspace: it is neither a source nor a destination of a copy
    FmFull f;
    const int rc = fm_full_plan(a, b, n, B, &f);
    if (rc != RBX_OK) return rc;
    if (f.ta.n_cid != 0) return RBX_ERR_UNSUPPORTED;
    *p = f.p;
    return RBX_OK;
  }
  return make_plan(a, n, B, nullptr, 0, p);
}

}  // namespace rbx

extern "C" int rbx_sort_share(const rbx_field_t* src_a, const rbx_field_t* src_b, int32_t src_n, int32_t src_is_fm,
                              const void* d_src_workspace, const rbx_field_t* dst_a, const rbx_field_t* dst_b,
                              int32_t dst_n, int32_t dst_is_fm, void* d_dst_workspace, size_t dst_workspace_bytes,
                              int64_t batch, void* stream) {
  using namespace rbx;
  if (batch <= 0 || d_src_workspace == nullptr || d_dst_workspace == nullptr) return RBX_ERR_UNSUPPORTED;
  BwdPlan src, dst;
  if (plan_of(src_a, src_b, src_n, src_is_fm, batch, &src) != RBX_OK) return RBX_ERR_UNSUPPORTED;
  int rc = plan_of(dst_a, dst_b, dst_n, dst_is_fm, batch, &dst);
  if (rc == RBX_ERR_UNSUPPORTED) return rc;
  if (rc != RBX_OK) return rc;
  if (dst.n_lookups == 0 || !same_pairs(src, dst)) return RBX_ERR_UNSUPPORTED;
  if (dst_workspace_bytes < dst.bytes) return fail(RBX_ERR_WORKSPACE, "sort_share: workspace too small");
  const char* sws = static_cast<const char*>(d_src_workspace);
  char* dws = static_cast<char*>(d_dst_workspace);
  const int cur = dst.passes & 1;
  unsigned blocks = (dst.n_lookups + 1023) / 1024;
  if (blocks > static_cast<unsigned>(kCUs) * 8) blocks = kCUs * 8;
  hipLaunchKernelGGL(copy_pairs_kernel, dim3(blocks), dim3(256), 0, as_stream(stream),
                     reinterpret_cast<const unsigned*>(sws + src.off_keys[cur]),
                     reinterpret_cast<const unsigned*>(sws + src.off_vals[cur]),
                     reinterpret_cast<unsigned*>(dws + dst.off_keys[cur]), reinterpret_cast<unsigned*>(dws + dst.off_vals[cur]),
                     dst.n_lookups, reinterpret_cast<unsigned*>(dws + dst.off_fin));
  return check_launch("copy_pairs_kernel");
}

extern "C" int rbx_fm_bwd(const rbx_field_t* emb, const rbx_field_t* lr, int32_t n_fields, int64_t batch,
                          const float* d_dlogit, const float* d_sum, float* d_dbias, int32_t accumulate,
                          int32_t phases, void* d_workspace, size_t workspace_bytes, void* stream) {
  using namespace rbx;
  if (d_dlogit == nullptr) return fail(RBX_ERR_INVALID, "fm: d_dlogit is NULL");
  if (emb != nullptr && d_sum == nullptr) return fail(RBX_ERR_INVALID, "fm: d_sum from the forward is required");
  if (batch == 0) return RBX_OK;
  FmFull f;
  int rc = fm_full_plan(emb, lr, n_fields, batch, &f);
  if (rc != RBX_OK) return rc;
  const BwdPlan& p = f.p;
  const int D = f.D, n_num = f.n_num;
  const size_t need = f.bytes;
  if (d_workspace == nullptr || workspace_bytes < need)
    return fail(RBX_ERR_WORKSPACE, "fm: workspace %zu B < required %zu B", workspace_bytes, need);
  char* ws = static_cast<char*>(d_workspace);
  hipStream_t s = as_stream(stream);
  if (p.n_lookups > 0 && (phases & 1) && !(phases & 16)) {
    const int cur = p.passes & 1;
    const unsigned* keys = reinterpret_cast<const unsigned*>(ws + p.off_keys[cur]);
    const unsigned* vals = reinterpret_cast<const unsigned*>(ws + p.off_vals[cur]);
    const FmPolicy::Args args = {d_dlogit, emb ? d_sum : nullptr, D, accumulate};
    const bool vec = p.vec && ((reinterpret_cast<uintptr_t>(d_sum) & 15) == 0);
    rc = vec ? dispatch_reduce<FmPolicy, true>(p, args, keys, vals, ws, s)
             : dispatch_reduce<FmPolicy, false>(p, args, keys, vals, ws, s);
    if (rc != RBX_OK) return rc;
  }
  // numeric partials in 512-sample units (ta_numeric_block) -- inside tier A's first launch when this call runs tier A too
  TaNumPack np;
  np.n = 0;
  // (512 samples' S rows + x columns must fit a workgroup's LDS with room for a second one: dims up to ~40; wider rows keep
  //  fm_numeric_partial_kernel, whose sample count shrinks with the dim)
  const bool num_blocks_form = (phases & 2) && emb != nullptr && n_num > 0 && n_num <= kTaNumMax && D % 4 == 0 &&
                               (reinterpret_cast<uintptr_t>(d_sum) & 15) == 0 && ta_num_lds_bytes(n_num, D) <= 96 * 1024;
  np.reserved = 0;
  if (num_blocks_form) {
    np.n = n_num;
    for (int i = 0; i < n_num; ++i) {
      np.ids[i] = f.np.f[i].ids; np.stride_b[i] = f.np.f[i].stride_b; np.dtype[i] = f.np.f[i].dtype;
    }
  }
  bool num_fused = false;                            // ... they rode in tier A's first launch
  if ((phases & 1) && !(phases & 8)) {               // the tables of tier A: block partials, then every row written once
    rc = ta_dispatch_bwd(f.ta, batch, d_dlogit, emb ? d_sum : nullptr, accumulate, ws + f.off_ta, s,
                         np.n > 0 ? &np : nullptr, reinterpret_cast<float*>(ws + p.bytes), &num_fused);
    if (rc != RBX_OK) return rc;
  }
  if ((phases & 2) && (n_num > 0 || d_dbias != nullptr)) {
    float* partial = reinterpret_cast<float*>(ws + p.bytes);
    unsigned nblk = p.num_blocks;
    if (num_blocks_form) {
      nblk = static_cast<unsigned>((batch + kTaNumBlock - 1) / kTaNumBlock);
      if (!num_fused) {
        const size_t lds = ta_num_lds_bytes(n_num, D);
        static bool attr_set = false;
        if (!attr_set) {
          (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fm_numeric_blocks_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
          attr_set = true;
        }
        hipLaunchKernelGGL(fm_numeric_blocks_kernel, dim3(nblk), dim3(512), lds, s, np, static_cast<long long>(batch), D,
                           d_dlogit, d_sum, partial);
      }
    } else {
      const int ns = fm_num_samples(D, n_num);
      const size_t lds = (static_cast<size_t>(ns) * (D + 1) + 2 * static_cast<size_t>(n_num) * (ns + 1)) * sizeof(float);
      hipLaunchKernelGGL(fm_numeric_partial_kernel, dim3(p.num_blocks), dim3(256), lds, s, f.np, n_num,
                         static_cast<long long>(batch), D, ns, d_dlogit, emb ? d_sum : nullptr, partial,
                         emb != nullptr && D % 4 == 0 && (reinterpret_cast<uintptr_t>(d_sum) & 15) == 0);
    }
    hipLaunchKernelGGL(fm_numeric_final_kernel, dim3(n_num + 1, D + 2), dim3(64), 0, s, f.np, n_num, D, nblk,
                       partial, d_dbias, (phases & 4) != 0);
    rc = check_launch("fm numeric kernels");
    if (rc != RBX_OK) return rc;
  }
  return RBX_OK;
}

extern "C" int rbx_fm_extra_bwd(const float* d_dlogit, const float* d_sum, const float* d_extra, int64_t batch,
                                int32_t n_extra, int32_t dim, int32_t extra_stride, int32_t extra_lr_off,
                                const int32_t* d_extra_index, int64_t extra_rows, float* d_dextra, void* stream) {
  using namespace rbx;
  if (d_dlogit == nullptr || d_dextra == nullptr) return fail(RBX_ERR_INVALID, "fm_extra_bwd: NULL tensor");
  const bool has_emb = dim > 0;
  if (has_emb && (d_sum == nullptr || d_extra == nullptr))
    return fail(RBX_ERR_INVALID, "fm_extra_bwd: S and the extra rows are needed for dE");
  if (batch <= 0 || n_extra <= 0) return RBX_OK;
  const long long total = static_cast<long long>(batch) * n_extra * extra_stride;
  long long blocks = (total + 255) / 256;
  if (blocks > kCUs * 8) blocks = kCUs * 8;
  hipLaunchKernelGGL(fm_extra_bwd_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, as_stream(stream), d_dlogit,
                     d_sum, d_extra, static_cast<long long>(batch), n_extra, dim, extra_stride, extra_lr_off, has_emb,
                     d_extra_index, static_cast<long long>(extra_rows), d_dextra);
  return check_launch("fm_extra_bwd_kernel");
}
