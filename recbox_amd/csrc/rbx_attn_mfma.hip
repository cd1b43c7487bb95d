// rbx_attn_mfma.hip -- K6 on the fp32 matrix cores: fused (causal) softmax attention for the
// SASRec shape regime (L <= 256, head_dim 32 or 64, no explicit mask), forward and backward.
//
// Same reference behaviour as rbx_attn.hip (nn.MultiheadAttention core of
// third_party/rechub/models/matching/sasrec.py:81-87); this file is the fast path, the VALU
// kernels there keep serving explicit masks, returned probabilities and small head dims.
//
// One workgroup (4 waves) per (sample, head); all products run on v_mfma_f32_32x32x2_f32 (exact
// fp32, so the 1e-4 parity bar holds).  The trick that removes every register shuffle: tiles are
// computed TRANSPOSED.  S^T = K Q^T has C-layout "lane & 31 = query, 16 registers x 2 half-waves
// = 32 keys", so softmax statistics are per LANE (one xor-32 shuffle joins the halves) and the
// probabilities p[r] are ALREADY the B operand of the next product O^T = V^T P^T (B[k][col]: k
// selected by the half-wave, col = lane & 31) -- register r of S^T feeds MFMA step r of O^T.
// The per-row operands (K, V, or Q, dO in the second backward phase) sit in LDS as row-major
// [L][HD + kPad] (kPad = 1, odd stride: both "lanes = rows" and "lanes = columns" reads are conflict free);
// the per-lane operands (the Q / dO / K / V tile of the wave) live in registers.
//   forward : lane = query.  S^T -> online softmax -> O^T accumulators (rescaled per lane).
//   backward A (lane = query): S^T, dP^T = V dO^T, dS^T = P^T o (dP^T - D) -> dQ^T += K^T dS^T.
//   backward B (lane = key)  : S = Q K^T, dP = dO V^T -> dV^T += dO^T P, dK^T += Q^T dS.
// No atomics, deterministic.  Query/key tiles are dealt to the 4 waves in a zig-zag (heavy tile
// + light tile) so the causal triangle is balanced.
#include <stdlib.h>
#include "rbx_internal.h"

namespace rbx {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kT = 32;                      // tile edge (queries or keys)
#define RBX_ATTN_PAD 1     // floats of padding per LDS row of the resident kernels: 1 = odd pitch, scalar LDS accesses; 4 = 16-byte
                           // of the LDS instructions of a tile_dot).  Measured, profiles/r04/INDEX.md: dQ kernel 555 vs 567 us,
                           // dK | dV 661 vs 663, resident forward 427 vs 408 (spills): the LDS round trips are not what these
                           // kernels wait for.
constexpr int kPad = RBX_ATTN_PAD;
constexpr int kAttnThreads = 512;           // 8 wavefronts: two per SIMD, so one hides the other's MFMA / LDS latency

__device__ __forceinline__ int tile_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// [rows][HD] global -> LDS [rows_pad][HD + kPad], scaled, zero beyond `rows`.  float4 global reads, four
// in flight per thread (one workgroup per CU: nothing else hides this latency); the odd LDS row stride
// forces scalar LDS writes.
template <int HD>
__device__ __forceinline__ void stage_rows(const float* __restrict__ g, const long long ld, float* __restrict__ lds, int rows,
                                           int rows_pad, float scale) {
  constexpr int Q4 = HD / 4;                               // float4 per row
  const int total = rows_pad * Q4;
  for (int i0 = threadIdx.x; i0 < total; i0 += 4 * kAttnThreads) {
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * kAttnThreads;
      const int r = i / Q4;
      v[u] = (i < total && r < rows) ? *reinterpret_cast<const float4*>(g + r * ld + (i - r * Q4) * 4)
                                     : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * kAttnThreads;
      if (i < total) {
        const int r = i / Q4, d = (i - r * Q4) * 4;
        float* dst = lds + r * (HD + kPad) + d;
        if constexpr (kPad == 4) {
          *reinterpret_cast<float4*>(dst) = make_float4(v[u].x * scale, v[u].y * scale, v[u].z * scale, v[u].w * scale);
        } else {
          dst[0] = v[u].x * scale; dst[1] = v[u].y * scale; dst[2] = v[u].z * scale; dst[3] = v[u].w * scale;
        }
      }
    }
  }
}

// Two blocks of a long sequence at once (more than 6 x 512 float4 each: head_dim 64, L > 192): all 16 requests of a thread in
// flight before the first LDS write, at clamped addresses with no branch around a load.  stage_rows above keeps four in flight
// and staged K, then V: four dependent round trips per sequence -- 63-71 us of the backward kernels' 566 / 666 us
// (profiles/r04/attn_bwd_parts.txt) with one workgroup per CU and nothing else to hide them.
template <int HD>
__device__ __forceinline__ void stage_rows_pair(const float* __restrict__ ga, const long long lda, float* __restrict__ la,
                                                const float sa, const float* __restrict__ gb, const long long ldb,
                                                float* __restrict__ lb, const float sb, int rows, int rows_pad) {
  constexpr int Q4 = HD / 4, N = 8;
  const int total = rows_pad * Q4;
  float4 va[N], vb[N];
#pragma unroll
  for (int u = 0; u < N; ++u) {
    int i = threadIdx.x + u * kAttnThreads;
    i = i < total ? i : total - 1;
    int r = i / Q4;
    const int c = (i - r * Q4) * 4;
    r = r < rows ? r : rows - 1;
    va[u] = *reinterpret_cast<const float4*>(ga + r * lda + c);
    vb[u] = *reinterpret_cast<const float4*>(gb + r * ldb + c);
  }
#pragma unroll
  for (int u = 0; u < N; ++u) {
    const int i = threadIdx.x + u * kAttnThreads;
    if (i < total) {
      const int r = i / Q4, d = (i - r * Q4) * 4;
      const float ka = r < rows ? sa : 0.f, kb = r < rows ? sb : 0.f;
      float* da = la + r * (HD + kPad) + d;
      float* db = lb + r * (HD + kPad) + d;
      da[0] = va[u].x * ka; da[1] = va[u].y * ka; da[2] = va[u].z * ka; da[3] = va[u].w * ka;
      db[0] = vb[u].x * kb; db[1] = vb[u].y * kb; db[2] = vb[u].z * kb; db[3] = vb[u].w * kb;
    }
  }
}
#define RBX_ATTN_STAGE_PAIR 1
template <int HD>
__device__ __forceinline__ void stage_two(const float* __restrict__ ga, const long long lda, float* __restrict__ la, const float sa,
                                          const float* __restrict__ gb, const long long ldb, float* __restrict__ lb,
                                          const float sb, int rows, int rows_pad) {
  const int total = rows_pad * (HD / 4);
  if (RBX_ATTN_STAGE_PAIR && total > 6 * kAttnThreads && total <= 8 * kAttnThreads) {
    stage_rows_pair<HD>(ga, lda, la, sa, gb, ldb, lb, sb, rows, rows_pad);
  } else {
    stage_rows<HD>(ga, lda, la, rows, rows_pad, sa);
    stage_rows<HD>(gb, ldb, lb, rows, rows_pad, sb);
  }
}

// The same block of the NEXT sequence, fetched into registers while the current one is being computed (a workgroup that
// loops over sequences: one workgroup per CU means nothing else hides these loads -- the forward at L = 200 spent about as
// long waiting for K and V as multiplying them).  Issued as inline assembly so that the compiler cannot sink the loads to
// their use at the top of the next iteration; rows beyond `rows` are clamped here and zeroed when they are written to LDS.
typedef float attn_f4 __attribute__((ext_vector_type(4)));
template <int HD, int N>
__device__ __forceinline__ void prefetch_rows(const float* g, const long long ld, int rows, attn_f4 (&v)[N]) {
  constexpr int Q4 = HD / 4;
#pragma unroll
  for (int u = 0; u < N; ++u) {
    const int i = threadIdx.x + u * kAttnThreads;
    int r = i / Q4;
    const int c = i - r * Q4;
    r = r < rows ? r : rows - 1;
    const float* src = g + static_cast<long long>(r) * ld + c * 4;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v[u]) : "v"(src));
  }
}
template <int HD, int N>
__device__ __forceinline__ void commit_rows(float* __restrict__ lds, int rows, int rows_pad, float scale,
                                            const attn_f4 (&v)[N]) {
  constexpr int Q4 = HD / 4;
  const int total = rows_pad * Q4;
#pragma unroll
  for (int u = 0; u < N; ++u) {
    const int i = threadIdx.x + u * kAttnThreads;
    if (i < total) {
      const int r = i / Q4, d = (i - r * Q4) * 4;
      const float k = r < rows ? scale : 0.f;
      float* dst = lds + r * (HD + kPad) + d;
      if constexpr (kPad == 4) {
        *reinterpret_cast<float4*>(dst) = make_float4(v[u][0] * k, v[u][1] * k, v[u][2] * k, v[u][3] * k);
      } else {
        dst[0] = v[u][0] * k; dst[1] = v[u][1] * k; dst[2] = v[u][2] * k; dst[3] = v[u][3] * k;
      }
    }
  }
}

// the wave's own tile as B operands: reg[s] = g[row0 + (lane & 31)][(lane >> 5) * HD/2 + s] * scale.
// MFMA step s contracts over TWO columns, one per half-wave; which two is free as long as both operands agree, so the
// halves take the two contiguous halves of a row (columns [0, HD/2) and [HD/2, HD)): a lane reads HD/8 float4 that it
// uses entirely, all in flight.  (The first version paired columns 2 s and 2 s + 1: every lane loaded the WHOLE row,
// HD/4 float4, and kept every other element -- the compiler issued those loads two at a time, ~8 dependent round trips per
// operand tile, which is where half of the wavefronts' cycles were parked: profiles/r02/sq_stalls.txt.)
template <int HD>
__device__ __forceinline__ void load_tile_regs(const float* __restrict__ g, const long long ld, int row0, int rows, float scale,
                                               float (&reg)[HD / 2]) {
  const int lane = threadIdx.x & 63;
  const int row = row0 + (lane & 31), half = lane >> 5;
  const bool ok = row < rows;
  const float4* src = reinterpret_cast<const float4*>(g + static_cast<long long>(ok ? row : 0) * ld + half * (HD / 2));
  float4 v[HD / 8];
#pragma unroll
  for (int q = 0; q < HD / 8; ++q) v[q] = ok ? src[q] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int q = 0; q < HD / 8; ++q) {
    reg[4 * q] = v[q].x * scale; reg[4 * q + 1] = v[q].y * scale; reg[4 * q + 2] = v[q].z * scale; reg[4 * q + 3] = v[q].w * scale;
  }
}

// the same tile with NO arithmetic on the loaded values and rows beyond `rows` read from row rows - 1 (finite values that the
// masks of the backward kernels keep out of every result): nothing uses the registers here, so the loads stay in flight until
// their first use -- the backward kernels request their first tile BEFORE they stage the sequence's rows into LDS.
template <int HD>
__device__ __forceinline__ void load_tile_clamped(const float* __restrict__ g, const long long ld, int row0, int rows,
                                                  float (&reg)[HD / 2]) {
  const int lane = threadIdx.x & 63;
  int row = row0 + (lane & 31);
  row = row < rows ? row : rows - 1;
  const float4* src = reinterpret_cast<const float4*>(g + static_cast<long long>(row) * ld + (lane >> 5) * (HD / 2));
#pragma unroll
  for (int q = 0; q < HD / 8; ++q) {
    const float4 v = src[q];
    reg[4 * q] = v.x; reg[4 * q + 1] = v.y; reg[4 * q + 2] = v.z; reg[4 * q + 3] = v.w;
  }
}
#define RBX_ATTN_EARLY_TILE 0   // backward kernels: 1 = the first job's own tiles are requested in front of the staging (measured slower)

#define RBX_ATTN_QFIRST 1  // looping forward kernel: the K / V prefetch is issued once the wavefront's Q tile has arrived
#define RBX_ATTN_ABL 0     // profiles/ubench/attn_parts.hip: the forward kernel without 1 = S^T, 2 = the softmax, 4 = O^T += V^T P^T, 8 = the Q tile loads, 16 = the stores of unsplit tiles; the backward kernels without 32 = the staging of a sequence's rows, 64 = the loads of a wavefront's own tiles
#define RBX_ATTN_BF16X6 44 // which tile products run on the bf16 matrix cores, operands split three ways: bits 0, 1 the forward
                           // (0: v_mfma_f32_32x32x2_f32 everywhere)
// f32 products on the bf16 pipes (the recipe of rbx_dense.hip's gemm_bx6_kernel): x = h + m + l with bf16 h = rn(x), m = rn(x - h),
// l = rn(x - h - m); a b = ah bh + (ah bm + am bh) + (ah bl + al bh + am bm) + O(2^-24 |a b|): six v_mfma_f32_32x32x16_bf16 per
// 16 k (192 pipe cycles) where eight v_mfma_f32_32x32x2_f32 take 512, at an error per product of ONE f32 rounding -- the parity
// tests keep their tolerances.  The operands stay f32 in LDS and in global memory (same layouts, same staging): a lane splits the
// eight values of its A fragment between the LDS read and the MFMAs (4.5 VALU operations per element, on the vector pipe that the
// f32 form left idle), the wavefront's own tile is split once per tile.  A 32x32x16 step contracts over 8 k per half-wave where
// the f32 step took one: WHICH eight is free as long as both operands agree, so the halves keep the pairing of load_tile_regs
// (k = half * HD/2 + 8 step + e) and, in tile_accumulate, register r of the weights keeps row tile_row(r, half).
typedef __bf16 abf16x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 abf16x8_t __attribute__((ext_vector_type(8)));
typedef float af32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned au32x4_t __attribute__((ext_vector_type(4)));
struct Split8 {
  abf16x8_t h, m, l;
};
__device__ __forceinline__ void attn_split2(af32x2_t x, unsigned& h, unsigned& m, unsigned& l) {
  const abf16x2_t hb = __builtin_convertvector(x, abf16x2_t);
  x -= __builtin_convertvector(hb, af32x2_t);
  const abf16x2_t mb = __builtin_convertvector(x, abf16x2_t);
  x -= __builtin_convertvector(mb, af32x2_t);
  const abf16x2_t lb = __builtin_convertvector(x, abf16x2_t);
  h = __builtin_bit_cast(unsigned, hb);
  m = __builtin_bit_cast(unsigned, mb);
  l = __builtin_bit_cast(unsigned, lb);
}
__device__ __forceinline__ Split8 split8(const float (&x)[8]) {
  au32x4_t h, m, l;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    unsigned hh, mm, ll;
    attn_split2(af32x2_t{x[2 * p], x[2 * p + 1]}, hh, mm, ll);
    h[p] = hh; m[p] = mm; l[p] = ll;
  }
  Split8 o;
  o.h = __builtin_bit_cast(abf16x8_t, h);
  o.m = __builtin_bit_cast(abf16x8_t, m);
  o.l = __builtin_bit_cast(abf16x8_t, l);
  return o;
}
// acc += A B to one f32 rounding per product; two chains so that consecutive MFMAs do not wait for each other
__device__ __forceinline__ void mfma6(f32x16& c0, f32x16& c1, const Split8& a, const Split8& b) {
  c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.l, b.h, c0, 0, 0, 0);
  c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.l, c1, 0, 0, 0);
  c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.m, b.m, c0, 0, 0, 0);
  c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.m, b.h, c1, 0, 0, 0);
  c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.m, c0, 0, 0, 0);
  c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.h, c1, 0, 0, 0);
}
// c0 += A0 B, c1 += A1 B, the two chains interleaved
__device__ __forceinline__ void mfma6x2(f32x16& c0, f32x16& c1, const Split8& a0, const Split8& a1, const Split8& b) {
  c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0.l, b.h, c0, 0, 0, 0);
  c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1.l, b.h, c1, 0, 0, 0);
  c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0.h, b.l, c0, 0, 0, 0);
  c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1.h, b.l, c1, 0, 0, 0);
  c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0.m, b.m, c0, 0, 0, 0);
  c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1.m, b.m, c1, 0, 0, 0);
  c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0.m, b.h, c0, 0, 0, 0);
  c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1.m, b.h, c1, 0, 0, 0);
  c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0.h, b.m, c0, 0, 0, 0);
  c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1.h, b.m, c1, 0, 0, 0);
  c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0.h, b.h, c0, 0, 0, 0);
  c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1.h, b.h, c1, 0, 0, 0);
}

// the wavefront's own tile (load_tile_regs) in the form the tile products take it
template <int HD, bool X6>
struct TileOp {
  const float* v;            // f32 form: the registers load_tile_regs filled, as they are
};
template <int HD>
struct TileOp<HD, true> {
  Split8 p[HD / 16];
};
template <int HD, bool X6>
__device__ __forceinline__ void make_op(const float (&reg)[HD / 2], TileOp<HD, X6>& op) {
  if constexpr (X6) {
#pragma unroll
    for (int s = 0; s < HD / 16; ++s) {
      float x[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] = reg[8 * s + e];
      op.p[s] = split8(x);
    }
  } else {
    op.v = reg;
  }
}

// (Measured and not kept, profiles/r03/INDEX.md: the LDS reads of a tile product issued four ahead of the MFMAs that use them
//  with __builtin_amdgcn_sched_group_barrier: the forward kernel alone 456 vs 459 us.  profiles/r04: s_setprio 1 around the
//  tile products: forward 395.7 vs 394.2 us, backward 1285.8 vs 1284.4 us.)
// acc[row = li of `rows_lds`][col = lane] = sum_d rows_lds[row0 + li][d] * reg[d]   (column pairing as in load_tile_regs)
template <int HD, bool X6>
__device__ __forceinline__ f32x16 tile_dot(const float* __restrict__ rows_lds, int row0, const TileOp<HD, X6>& op) {
  const int lane = threadIdx.x & 63;
  const float* a = rows_lds + (row0 + (lane & 31)) * (HD + kPad) + (lane >> 5) * (HD / 2);
  // two accumulator chains (even / odd k steps): a single chain makes every MFMA wait for the previous one
  f32x16 acc, acc1;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = acc1[r] = 0.f;
  if constexpr (X6) {
#pragma unroll
    for (int s = 0; s < HD / 16; ++s) {
      float x[8];
      if constexpr (kPad == 4) {
        const float4 u0 = *reinterpret_cast<const float4*>(a + 8 * s), u1 = *reinterpret_cast<const float4*>(a + 8 * s + 4);
        x[0] = u0.x; x[1] = u0.y; x[2] = u0.z; x[3] = u0.w; x[4] = u1.x; x[5] = u1.y; x[6] = u1.z; x[7] = u1.w;
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = a[8 * s + e];
      }
      mfma6(acc, acc1, split8(x), op.p[s]);
    }
  } else if constexpr (kPad == 4) {
#pragma unroll
    for (int s = 0; s < HD / 2; s += 4) {
      const float4 u = *reinterpret_cast<const float4*>(a + s);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(u.x, op.v[s], acc, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(u.y, op.v[s + 1], acc1, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(u.z, op.v[s + 2], acc, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(u.w, op.v[s + 3], acc1, 0, 0, 0);
    }
  } else {
#pragma unroll
    for (int s = 0; s < HD / 2; s += 2) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], op.v[s], acc, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s + 1], op.v[s + 1], acc1, 0, 0, 0);
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] += acc1[r];
  return acc;
}

// out[dt][row = d][col = lane] += sum_r rows_lds[row0 + tile_row(r)][dt*32 + li] * w[r]
template <int HD, bool X6>
__device__ __forceinline__ void tile_accumulate(const float* __restrict__ rows_lds, int row0, const f32x16& w,
                                                f32x16 (&out)[HD / 32]) {
  const int lane = threadIdx.x & 63;
  const int li = lane & 31, half = lane >> 5;
  if constexpr (X6) {
  static_assert(HD == 32 || HD == 64, "one or two column tiles");
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    float x[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = w[8 * s + e];
    const Split8 b = split8(x);
    Split8 a[HD / 32];
#pragma unroll
    for (int dt = 0; dt < HD / 32; ++dt) {
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] = rows_lds[(row0 + tile_row(8 * s + e, half)) * (HD + kPad) + dt * 32 + li];
      a[dt] = split8(x);
    }
    if constexpr (HD == 64) {
      mfma6x2(out[0], out[1], a[0], a[1], b);
    } else {
      f32x16 t;
#pragma unroll
      for (int r = 0; r < 16; ++r) t[r] = 0.f;
      mfma6(out[0], t, a[0], b);
#pragma unroll
      for (int r = 0; r < 16; ++r) out[0][r] += t[r];
    }
  }
  } else {
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float* a = rows_lds + (row0 + tile_row(r, half)) * (HD + kPad) + li;
#pragma unroll
    for (int dt = 0; dt < HD / 32; ++dt) out[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[dt * 32], w[r], out[dt], 0, 0, 0);
  }
  }
}

// transposed accumulators (row = d, col = lane's row) -> g[row][d] * scale
template <int HD>
__device__ __forceinline__ void store_transposed(float* __restrict__ g, const long long ld, int row0, int rows, float scale,
                                                 const f32x16 (&acc)[HD / 32]) {
  const int lane = threadIdx.x & 63;
  const int row = row0 + (lane & 31), half = lane >> 5;
  if (row >= rows) return;
#pragma unroll
  for (int dt = 0; dt < HD / 32; ++dt)
#pragma unroll
    for (int q = 0; q < 4; ++q)
      *reinterpret_cast<float4*>(g + static_cast<long long>(row) * ld + dt * 32 + 8 * q + 4 * half) =
          make_float4(acc[dt][4 * q] * scale, acc[dt][4 * q + 1] * scale, acc[dt][4 * q + 2] * scale,
                      acc[dt][4 * q + 3] * scale);
}

// Operand layout: element (b, l, h, d) of a tensor sits at base + (b * L + l) * ld + h * HD + d.  The contiguous
// [BH, L, HD] form is ld = HD with one "head" per block; ld = E or 2 E with heads > 1 reads the projections' output
// [B, L, E] / the fused K|V projection [B, L, 2 E] in place and writes O / dQ / dK|dV in the layout the next GEMM reads
// (no transposes, no K / V split copies, no concatenation of dK and dV).
struct AttnLd {
  long long q, k, v, o, go, dq, dk, dv;
  int heads;
};

__device__ __forceinline__ long long attn_base(long long bh, int heads, int L, long long ld, int hd) {
  const long long b = bh / heads;
  return b * L * ld + (bh - b * heads) * hd;
}

// Tile schedule: with L <= 256 there are at most 8 tiles and the workgroup has 8 wavefronts.  Numbered by cost -- tile u
// takes u+1 tile steps (query tile u in the forward / dQ kernels, key tile nT-1-u in the dK|dV kernel) -- SIMD s
// (wavefronts s and s+4) gets the heavy tile nT-1-s and a light one: tile s when nT is even (nT+1 steps per SIMD), tile
// s-1 when nT is odd (SIMD 0 keeps the heaviest tile alone: nT steps per SIMD.  L = 200 is 7 tiles: paired 6+0, 5+1, 4+2, 3
// the SIMDs carried 8, 8, 8, 4 steps; 6, 5+0, 4+1, 3+2 is 7 each).
//
// Within a SIMD the steps are then dealt to its TWO wavefronts (`split`, causal only): wavefront s+4 takes the light tile
// and the first x = (heavy - light) / 2 steps of the heavy one, wavefront s the rest of the heavy tile; the two partial
// results of the heavy tile meet through LDS (a sum for dQ / dK / dV; (m, l, O^T) rescaled to the common maximum in the
// forward) and wavefront s writes the tile.  At 7 tiles that is 4 + 3 steps on every SIMD where it was 7 + 0, 6 + 1, 5 + 2,
// 4 + 3: a wavefront issues its MFMAs and its softmax arithmetic one after the other, so a SIMD whose second wavefront has
// run out of work (or never had any) leaves the matrix core idle for half of every step -- the lone 7-step wavefront of
// SIMD 0 was the length of every sequence.
struct WavePlan {
  int n;                       // jobs of this wavefront (0, 1 or 2)
  int tile0, beg0, end0;       // job 0: cost index of the tile, its steps [beg, end)
  int tile1, beg1, end1;       // job 1
  int partial;                 // the job whose result is a partial for the SIMD's other wavefront (-1: none)
  int merge;                   // this wavefront adds the other one's partial to its (only) job before it writes the tile
  int simd;
  __device__ __forceinline__ int tile(int j) const { return j == 0 ? tile0 : tile1; }   // (scalars, not arrays: an array
  __device__ __forceinline__ int beg(int j) const { return j == 0 ? beg0 : beg1; }      //  indexed by the job would live
  __device__ __forceinline__ int end(int j) const { return j == 0 ? end0 : end1; }      //  in scratch memory)
};

__device__ __forceinline__ WavePlan wave_plan(const int nT, const int wid, const int causal, const int split) {
  WavePlan p;
  p.n = 0; p.partial = -1; p.merge = 0; p.simd = wid & 3;
  p.tile0 = p.tile1 = 0; p.beg0 = p.beg1 = 0; p.end0 = p.end1 = 0;
  const int s = wid & 3;
  const int uh = nT - 1 - s, ul = s - (nT & 1);
  const bool heavy = s <= uh, light = ul >= 0 && ul < uh;
  if (!heavy) return p;
  const int sh = causal ? uh + 1 : nT, sl = light ? (causal ? ul + 1 : nT) : 0;
  const int x = (split && causal && sh - sl >= 2) ? (sh - sl) / 2 : 0;
  if (wid < 4) {
    p.n = 1; p.tile0 = uh; p.beg0 = x; p.end0 = sh; p.merge = x > 0;
  } else if (light) {
    p.n = 1; p.tile0 = ul; p.end0 = sl;
    if (x > 0) { p.n = 2; p.tile1 = uh; p.end1 = x; p.partial = 1; }
  } else if (x > 0) {
    p.n = 1; p.tile0 = uh; p.end0 = x; p.partial = 0;
  }
  return p;
}

// the accumulators of a tile through LDS, a register a row of 64 lanes (conflict free)
template <int HD>
__device__ __forceinline__ void park_acc(float* __restrict__ buf, const f32x16 (&acc)[HD / 32]) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int dt = 0; dt < HD / 32; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) buf[(dt * 16 + r) * 64 + lane] = acc[dt][r];
}
template <int HD>
__device__ __forceinline__ void add_parked(const float* __restrict__ buf, f32x16 (&acc)[HD / 32], const float mine = 1.f,
                                           const float theirs = 1.f) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int dt = 0; dt < HD / 32; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[dt][r] = acc[dt][r] * mine + buf[(dt * 16 + r) * 64 + lane] * theirs;
}
template <int HD>
constexpr int merge_floats() { return HD * 32 + 128; }      // one SIMD's slot: O^T accumulators + (m, l) per lane

// NPF > 0 (float4 per thread that hold one [Lp, HD] block: Lp * HD / 4 / 512): a workgroup per CU that loops over sequences (gridDim.x <= BH) and fetches the next sequence's K and V into registers
// while it computes the current one; used where only one workgroup fits a CU anyway (K, V of a sequence > 80 KB of LDS).
// `split`: 0 = every tile by one wavefront; 1 = the SIMDs' heavy tiles by both wavefronts, their partials meet in LDS slots
// of their own behind K and V; 2 = the same through the K rows, once every wavefront is done with them (no LDS of its own:
// the form for short sequences, where several workgroups share a CU).
template <int HD, bool DROP, int NPF>
__global__ __launch_bounds__(kAttnThreads) void attn_mfma_fwd_kernel(const float* __restrict__ Q0, const float* __restrict__ K0,
                                                            const float* __restrict__ V0, const int L,
                                                            const float scale, const int causal,
                                                            float* __restrict__ O0, float* __restrict__ LSE,
                                                            const DropArgs drop, const AttnLd ld, const long long BH,
                                                            const int split) {
  extern __shared__ float lds[];
  const int nT = (L + kT - 1) / kT, Lp = nT * kT;
  float* Ks = lds;
  float* Vs = lds + Lp * (HD + kPad);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, li = lane & 31, half = lane >> 5;
  const WavePlan pl = wave_plan(nT, wid, causal, split);
  float* slot = (split == 1 ? Vs + Lp * (HD + kPad) : lds) + pl.simd * merge_floats<HD>();
  unsigned dk0 = 0, dk1 = 0;
  if (DROP) drop_seed(drop, &dk0, &dk1);
  constexpr bool PF = NPF > 0;
  attn_f4 kpf[PF ? NPF : 1], vpf[PF ? NPF : 1];
  for (long long bh = blockIdx.x; bh < BH; bh += gridDim.x) {
    const float* Q = Q0 + attn_base(bh, ld.heads, L, ld.q, HD);
    const float* K = K0 + attn_base(bh, ld.heads, L, ld.k, HD);
    const float* V = V0 + attn_base(bh, ld.heads, L, ld.v, HD);
    float* O = O0 + attn_base(bh, ld.heads, L, ld.o, HD);
    if (!PF || bh == static_cast<long long>(blockIdx.x)) {
      stage_rows<HD>(K, ld.k, Ks, L, Lp, 1.0f);
      stage_rows<HD>(V, ld.v, Vs, L, Lp, 1.0f);
    } else {
      if constexpr (PF) {
        // the matching wait: names the registers read-write so that nothing uses them above it
        if constexpr (NPF == 8)
          asm volatile("s_waitcnt vmcnt(0)" : "+v"(kpf[0]), "+v"(kpf[1]), "+v"(kpf[2]), "+v"(kpf[3]), "+v"(kpf[4]), "+v"(kpf[5]),
                       "+v"(kpf[6]), "+v"(kpf[NPF - 1]), "+v"(vpf[0]), "+v"(vpf[1]), "+v"(vpf[2]), "+v"(vpf[3]), "+v"(vpf[4]),
                       "+v"(vpf[5]), "+v"(vpf[6]), "+v"(vpf[NPF - 1]) : : "memory");
        else if constexpr (NPF == 7)
          asm volatile("s_waitcnt vmcnt(0)" : "+v"(kpf[0]), "+v"(kpf[1]), "+v"(kpf[2]), "+v"(kpf[3]), "+v"(kpf[4]), "+v"(kpf[5]),
                       "+v"(kpf[NPF - 1]), "+v"(vpf[0]), "+v"(vpf[1]), "+v"(vpf[2]), "+v"(vpf[3]), "+v"(vpf[4]), "+v"(vpf[5]),
                       "+v"(vpf[NPF - 1]) : : "memory");
        else
          asm volatile("s_waitcnt vmcnt(0)" : "+v"(kpf[0]), "+v"(kpf[1]), "+v"(kpf[2]), "+v"(kpf[3]), "+v"(kpf[4]),
                       "+v"(kpf[NPF - 1]), "+v"(vpf[0]), "+v"(vpf[1]), "+v"(vpf[2]), "+v"(vpf[3]), "+v"(vpf[4]),
                       "+v"(vpf[NPF - 1]) : : "memory");
        static_assert(!PF || (NPF >= 6 && NPF <= 8), "operand lists above");
        commit_rows<HD, PF ? NPF : 1>(Ks, L, Lp, 1.0f, kpf);
        commit_rows<HD, PF ? NPF : 1>(Vs, L, Lp, 1.0f, vpf);
      }
    }
    __syncthreads();
    bool fetched = false;
    auto fetch_next = [&]() {
      if constexpr (PF) {
        const long long nb = bh + gridDim.x;
        if (nb < BH) {
          prefetch_rows<HD, PF ? NPF : 1>(K0 + attn_base(nb, ld.heads, L, ld.k, HD), ld.k, L, kpf);
          prefetch_rows<HD, PF ? NPF : 1>(V0 + attn_base(nb, ld.heads, L, ld.v, HD), ld.v, L, vpf);
        }
        fetched = true;
      }
    };
    // the state of the wavefront's last job outlives the loop: a partial to hand over, or the tile that takes one in
    f32x16 oacc[HD / 32];
    float m = -INFINITY, lsum = 0.f;
    int i0 = 0, qi = li;
    for (int jb = 0; jb < pl.n; ++jb) {
      const int qt = pl.tile(jb);
      i0 = qt * kT;
      qi = i0 + li;
      float qreg[HD / 2];
      if constexpr ((RBX_ATTN_ABL & 8) != 0) {
#pragma unroll
        for (int q = 0; q < HD / 2; ++q) qreg[q] = scale * static_cast<float>(q + li);
        if (!fetched) fetch_next();
      } else if constexpr (PF && HD == 64 && RBX_ATTN_QFIRST) {
        // The next sequence's K and V are requested only once the wavefront's Q tile is THERE.  Requested behind the tile's
        // loads (loads return in order, "the tile is back first") they were also waited for with it: the compiler's wait in
        // front of the first MFMA is vmcnt(0) -- it does not count the prefetch's asm loads -- so every wavefront sat out its
        // share of the prefetch before it computed anything (profiles/r03/attn_parts.txt: 65 of 436 us belong to the Q tile
        // loads).  (The tile by asm loads and a counted wait, vmcnt(2 NPF), measured the same and is not safe: the
        // compiler may spill a register whose load is still in flight -- NaNs in the dropout variant.)
        load_tile_regs<HD>(Q, ld.q, i0, L, scale, qreg);
        asm volatile("" : "+v"(qreg[0]), "+v"(qreg[4]), "+v"(qreg[8]), "+v"(qreg[12]), "+v"(qreg[16]), "+v"(qreg[20]),
                     "+v"(qreg[24]), "+v"(qreg[28]) : : "memory");
        if (!fetched) fetch_next();
      } else {
        load_tile_regs<HD>(Q, ld.q, i0, L, scale, qreg);
        if (!fetched) fetch_next();                            // (behind the wave's own operand loads: those return first)
      }
#pragma unroll
      for (int dt = 0; dt < HD / 32; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;
      m = -INFINITY;
      lsum = 0.f;
      constexpr bool X6 = HD == 64 && (RBX_ATTN_BF16X6 & 1) != 0, XA = HD == 64 && (RBX_ATTN_BF16X6 & 2) != 0;
      TileOp<HD, X6> qop;
      make_op<HD, X6>(qreg, qop);
      for (int kt = pl.beg(jb); kt < pl.end(jb); ++kt) {
        const int j0 = kt * kT;
        f32x16 s;
        if constexpr ((RBX_ATTN_ABL & 1) != 0) {
#pragma unroll
          for (int r = 0; r < 16; ++r) s[r] = qreg[r] * Ks[j0 + lane];
        } else {
          s = tile_dot<HD, X6>(Ks, j0, qop);                   // S^T[key][query]
        }
        if constexpr ((RBX_ATTN_ABL & 2) != 0) {
          lsum += s[0];
          tile_accumulate<HD, XA>(Vs, j0, s, oacc);
          continue;
        }
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int kj = j0 + tile_row(r, half);
          if (kj >= L || (causal && kj > qi)) s[r] = -INFINITY;
          mx = fmaxf(mx, s[r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mn = fmaxf(m, mx);
        const float alpha = (mn == -INFINITY) ? 1.f : __expf(m - mn);
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          s[r] = (s[r] == -INFINITY) ? 0.f : __expf(s[r] - mn);
          ps += s[r];
        }
        ps += __shfl_xor(ps, 32, 64);
        lsum = lsum * alpha + ps;
#pragma unroll
        for (int dt = 0; dt < HD / 32; ++dt)
#pragma unroll
          for (int r = 0; r < 16; ++r) oacc[dt][r] *= alpha;
        m = mn;
        if (DROP) {        // dropout on the probabilities (nn.MultiheadAttention / ScaledDotProductAttention): the
                           // normaliser lsum stays the undropped sum, kept entries are scaled by 1 / (1 - p)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            unsigned c[4];
            drop_block(static_cast<unsigned>(qi) >> 2, static_cast<unsigned>(j0 + 8 * g + 4 * half) >> 2,
                       static_cast<unsigned long long>(bh), (qi & 3) >> 1, dk0, dk1, c);
#pragma unroll
            for (int q = 0; q < 4; ++q) s[4 * g + q] = drop_keep(c, qi & 1, q, drop.thr16) ? s[4 * g + q] * drop.scale : 0.f;
          }
        }
        if constexpr ((RBX_ATTN_ABL & 4) != 0) {
#pragma unroll
          for (int r = 0; r < 16; ++r) oacc[0][r] += s[r];
        } else {
          tile_accumulate<HD, XA>(Vs, j0, s, oacc);            // O^T[d][query] += V^T P^T
        }
      }
      if ((RBX_ATTN_ABL & 16) != 0 && lsum != 12345.f) continue;            // (no O / LSE stores of unsplit tiles)
      if (jb != pl.partial && !pl.merge) {
        store_transposed<HD>(O, ld.o, i0, L, 1.0f / lsum, oacc);
        if (half == 0 && qi < L) LSE[bh * L + qi] = m + __logf(lsum);
      }
    }
    if (PF && !fetched) fetch_next();                          // (a wavefront without a tile still fetches its share)
    if (split == 2) __syncthreads();                           // the slots lie over K: every wavefront is done with it
    if (pl.partial >= 0) {
      park_acc<HD>(slot, oacc);
      slot[HD * 32 + lane] = m;
      slot[HD * 32 + 64 + lane] = lsum;
    }
    if (PF || split != 0) __syncthreads();                     // every wavefront is done with this sequence's K, V rows
    if (pl.merge) {
      const float mo = slot[HD * 32 + lane], lo = slot[HD * 32 + 64 + lane];
      const float mn = fmaxf(m, mo);
      const float a = (m == -INFINITY) ? 0.f : __expf(m - mn), b = (mo == -INFINITY) ? 0.f : __expf(mo - mn);
      add_parked<HD>(slot, oacc, a, b);
      lsum = lsum * a + lo * b;
      store_transposed<HD>(O, ld.o, i0, L, 1.0f / lsum, oacc);
      if (half == 0 && qi < L) LSE[bh * L + qi] = mn + __logf(lsum);
    }
    if (split == 2) __syncthreads();                           // (the K rows are free for the next sequence's)
  }
}

// backward phase A: lane = query.  dQ and D = <dO, O>.
template <int HD, bool DROP>
__global__ __launch_bounds__(kAttnThreads) void attn_mfma_bwd_q_kernel(const float* __restrict__ Q, const float* __restrict__ K,
                                                              const float* __restrict__ V,
                                                              const float* __restrict__ O,
                                                              const float* __restrict__ dO,
                                                              const float* __restrict__ LSE, const int L,
                                                              const float scale, const int causal,
                                                              float* __restrict__ dQ, float* __restrict__ Dv,
                                                              const DropArgs drop, const AttnLd ld, const int split) {
  extern __shared__ float lds[];
  const int nT = (L + kT - 1) / kT, Lp = nT * kT;
  float* Ks = lds;
  float* Vs = lds + Lp * (HD + kPad);
  const long long bh = blockIdx.x;
  Q += attn_base(bh, ld.heads, L, ld.q, HD);
  K += attn_base(bh, ld.heads, L, ld.k, HD);
  V += attn_base(bh, ld.heads, L, ld.v, HD);
  O += attn_base(bh, ld.heads, L, ld.o, HD);
  dO += attn_base(bh, ld.heads, L, ld.go, HD);
  dQ += attn_base(bh, ld.heads, L, ld.dq, HD);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, li = lane & 31, half = lane >> 5;
  const WavePlan pl = wave_plan(nT, wid, causal, split);
  // The wavefront's own tiles (Q, dO, O rows of its query tile).  Alone the kernel ran 566 us; 495 without the staging below,
  // 426 without these loads, 337 without both (profiles/r04/attn_bwd_parts.txt): exposed round trips, one workgroup per CU.
  // The branch-free loader took 24 us of that; requesting the first job's tiles in FRONT of the staging
  // (RBX_ATTN_EARLY_TILE=1) keeps 96 registers live across it and spills (702 us).
  float qreg[HD / 2], greg[HD / 2], oreg[HD / 2];
  auto load_job = [&](const int jb) {
    const int r0 = pl.tile(jb) * kT;
    if constexpr ((RBX_ATTN_ABL & 64) != 0) {                // (... and without its own tiles' loads)
#pragma unroll
      for (int q = 0; q < HD / 2; ++q) { qreg[q] = 0.01f * q + lane; greg[q] = 0.02f * q; oreg[q] = 0.5f; }
    } else {
      load_tile_clamped<HD>(Q, ld.q, r0, L, qreg);
      load_tile_clamped<HD>(dO, ld.go, r0, L, greg);
      load_tile_clamped<HD>(O, ld.o, r0, L, oreg);
    }
  };
  if (RBX_ATTN_EARLY_TILE && pl.n > 0) load_job(0);
  if constexpr ((RBX_ATTN_ABL & 32) == 0) {                 // (profiles/ubench/attn_stream.hip: the kernel without its staging)
    stage_two<HD>(K, ld.k, Ks, 1.0f, V, ld.v, Vs, 1.0f, L, Lp);
  }
  __syncthreads();
  unsigned dk0 = 0, dk1 = 0;
  if (DROP) drop_seed(drop, &dk0, &dk1);
  f32x16 dq[HD / 32];
  int i0 = 0;
  for (int jb = 0; jb < pl.n; ++jb) {
    const int qt = pl.tile(jb);
    i0 = qt * kT;
    const int qi = i0 + li;
    if (!RBX_ATTN_EARLY_TILE || jb > 0) load_job(jb);
#pragma unroll
    for (int q = 0; q < HD / 2; ++q) qreg[q] *= scale;
    float Di = 0.f;
#pragma unroll
    for (int s = 0; s < HD / 2; ++s) Di += greg[s] * oreg[s];
    Di += __shfl_xor(Di, 32, 64);
    const float lse = (qi < L) ? LSE[bh * L + qi] : 0.f;
    if (half == 0 && qi < L && jb != pl.partial) Dv[bh * L + qi] = Di;
#pragma unroll
    for (int dt = 0; dt < HD / 32; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) dq[dt][r] = 0.f;
    // (profiles/r04/attn_forms.txt, kernels alone at 4096 x 200 x 64: this kernel 579 us with all three products on the bf16
    //  pipes, 599 / 606 with the tile_dots / the tile_accumulate only, 622 f32; the dK | dV kernel 679 with its two
    //  tile_accumulates, 772 with its two tile_dots (both: 744, its four operand sets spill), 704 f32; the streamed forward
    //  348 f32, 354 with tile_accumulate, 504 with tile_dot.  A lane's splits cost ~10 VALU operations per pair of elements:
    //  ~900 per tile step here, the kernels become VALU-bound where the f32 MFMAs had them wait for the matrix pipe.  Planes
    //  split ONCE per tile instead of once per wavefront need the tiled (streamed) form: DESIGN.md section 8.)
    constexpr bool X6 = HD == 64 && (RBX_ATTN_BF16X6 & 4) != 0, XA = HD == 64 && (RBX_ATTN_BF16X6 & 8) != 0;
    TileOp<HD, X6> qop, gop;
    make_op<HD, X6>(qreg, qop);
    make_op<HD, X6>(greg, gop);
    for (int kt = pl.beg(jb); kt < pl.end(jb); ++kt) {
      const int j0 = kt * kT;
      f32x16 s = tile_dot<HD, X6>(Ks, j0, qop);              // S^T
      f32x16 dp = tile_dot<HD, X6>(Vs, j0, gop);             // dP^T[key][query] = <V_key, dO_query>
      if (DROP) {                                            // d(dropped P) -> dP: the same mask and scale
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          unsigned c[4];
          drop_block(static_cast<unsigned>(qi) >> 2, static_cast<unsigned>(j0 + 8 * g + 4 * half) >> 2,
                     static_cast<unsigned long long>(bh), (qi & 3) >> 1, dk0, dk1, c);
#pragma unroll
          for (int q = 0; q < 4; ++q) dp[4 * g + q] = drop_keep(c, qi & 1, q, drop.thr16) ? dp[4 * g + q] * drop.scale : 0.f;
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int kj = j0 + tile_row(r, half);
        const bool vis = kj < L && qi < L && !(causal && kj > qi);
        const float p = vis ? __expf(s[r] - lse) : 0.f;
        s[r] = p * (dp[r] - Di);                             // dS^T
      }
      tile_accumulate<HD, XA>(Ks, j0, s, dq);                // dQ^T[d][query] += K^T dS^T
    }
    if (jb != pl.partial && !pl.merge) store_transposed<HD>(dQ, ld.dq, i0, L, scale, dq);
  }
  if (split) {                                               // the two halves of the heavy tiles meet through the K rows
    float* slot = lds + pl.simd * merge_floats<HD>();
    __syncthreads();
    if (pl.partial >= 0) park_acc<HD>(slot, dq);
    __syncthreads();
    if (pl.merge) {
      add_parked<HD>(slot, dq);
      store_transposed<HD>(dQ, ld.dq, i0, L, scale, dq);
    }
  }
}

// backward phase B: lane = key.  dK, dV.
template <int HD, bool DROP>
__global__ __launch_bounds__(kAttnThreads) void attn_mfma_bwd_kv_kernel(const float* __restrict__ Q, const float* __restrict__ K,
                                                               const float* __restrict__ V,
                                                               const float* __restrict__ dO,
                                                               const float* __restrict__ LSE,
                                                               const float* __restrict__ Dv, const int L,
                                                               const float scale, const int causal,
                                                               float* __restrict__ dK, float* __restrict__ dV,
                                                               const DropArgs drop, const AttnLd ld, const int split) {
  extern __shared__ float lds[];
  const int nT = (L + kT - 1) / kT, Lp = nT * kT;
  float* Qs = lds;                          // scale * Q
  float* Gs = Qs + Lp * (HD + kPad);           // dO
  float* Ls = Gs + Lp * (HD + kPad);           // lse[Lp]
  float* Ds = Ls + Lp;                      // D[Lp]
  const long long bh = blockIdx.x;
  Q += attn_base(bh, ld.heads, L, ld.q, HD);
  K += attn_base(bh, ld.heads, L, ld.k, HD);
  V += attn_base(bh, ld.heads, L, ld.v, HD);
  dO += attn_base(bh, ld.heads, L, ld.go, HD);
  dK += attn_base(bh, ld.heads, L, ld.dk, HD);
  dV += attn_base(bh, ld.heads, L, ld.dv, HD);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, li = lane & 31, half = lane >> 5;
  const WavePlan pl = wave_plan(nT, wid, causal, split);
  float kreg[HD / 2], vreg[HD / 2];                          // the wavefront's own K and V tile (as in the dQ kernel: 666 us
  auto load_job = [&](const int jb) {                        //  alone, 603 / 596 / 508 without the staging / these loads / both)
    const int r0 = (nT - 1 - pl.tile(jb)) * kT;
    if constexpr ((RBX_ATTN_ABL & 64) != 0) {
#pragma unroll
      for (int q = 0; q < HD / 2; ++q) { kreg[q] = 0.01f * q + lane; vreg[q] = 0.02f * q; }
    } else {
      load_tile_clamped<HD>(K, ld.k, r0, L, kreg);
      load_tile_clamped<HD>(V, ld.v, r0, L, vreg);
    }
  };
  if (RBX_ATTN_EARLY_TILE && pl.n > 0) load_job(0);
  if constexpr ((RBX_ATTN_ABL & 32) == 0) {
    stage_two<HD>(Q, ld.q, Qs, scale, dO, ld.go, Gs, 1.0f, L, Lp);
  }
  for (int i = threadIdx.x; i < Lp; i += blockDim.x) {
    Ls[i] = (i < L) ? LSE[bh * L + i] : 0.f;
    Ds[i] = (i < L) ? Dv[bh * L + i] : 0.f;
  }
  __syncthreads();
  unsigned dk0 = 0, dk1 = 0;
  if (DROP) drop_seed(drop, &dk0, &dk1);
  f32x16 dk[HD / 32], dv[HD / 32];
  int j0 = 0;
  for (int jb = 0; jb < pl.n; ++jb) {
    const int jt = nT - 1 - pl.tile(jb);            // (key tile jt meets nT - jt query tiles: cost index nT - 1 - jt)
    j0 = jt * kT;
    const int kj = j0 + li;
    if (!RBX_ATTN_EARLY_TILE || jb > 0) load_job(jb);
#pragma unroll
    for (int dt = 0; dt < HD / 32; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) dk[dt][r] = dv[dt][r] = 0.f;
    const int it_beg = (causal ? jt : 0) + pl.beg(jb), it_end = (causal ? jt : 0) + pl.end(jb);
    constexpr bool X6 = HD == 64 && (RBX_ATTN_BF16X6 & 16) != 0, XA = HD == 64 && (RBX_ATTN_BF16X6 & 32) != 0;
    TileOp<HD, X6> kop, vop;
    make_op<HD, X6>(kreg, kop);
    make_op<HD, X6>(vreg, vop);
    for (int it = it_beg; it < it_end; ++it) {
      const int i0 = it * kT;
      f32x16 s = tile_dot<HD, X6>(Qs, i0, kop);              // S[query][key] (already scaled)
      f32x16 dp = tile_dot<HD, X6>(Gs, i0, vop);             // dP[query][key]
      f32x16 p;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int qi = i0 + tile_row(r, half);
        const bool vis = kj < L && qi < L && !(causal && kj > qi);
        p[r] = vis ? __expf(s[r] - Ls[qi]) : 0.f;
      }
      if (DROP) {
        // lane = key: registers 4g..4g+3 are four consecutive queries, i.e. both 2 x 4 blocks of one (i >> 2, j >> 2)
        f32x16 pd;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            unsigned c[4];
            drop_block(static_cast<unsigned>(i0 + 8 * g + 4 * half) >> 2, static_cast<unsigned>(kj) >> 2,
                       static_cast<unsigned long long>(bh), h, dk0, dk1, c);
#pragma unroll
            for (int q = 0; q < 2; ++q) {
              const int r = 4 * g + 2 * h + q;
              const bool keep = drop_keep(c, q, kj & 3, drop.thr16);
              pd[r] = keep ? p[r] * drop.scale : 0.f;
              dp[r] = keep ? dp[r] * drop.scale : 0.f;
            }
          }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = p[r] * (dp[r] - Ds[i0 + tile_row(r, half)]);     // dS
        tile_accumulate<HD, XA>(Gs, i0, pd, dv);             // dV^T[d][key] += dO^T (dropped P)
        tile_accumulate<HD, XA>(Qs, i0, s, dk);
        continue;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = p[r] * (dp[r] - Ds[i0 + tile_row(r, half)]);       // dS
      tile_accumulate<HD, XA>(Gs, i0, p, dv);                // dV^T[d][key] += dO^T P
      tile_accumulate<HD, XA>(Qs, i0, s, dk);                // dK^T[d][key] += (scale Q)^T dS
    }
    if (jb != pl.partial && !pl.merge) {
      store_transposed<HD>(dK, ld.dk, j0, L, 1.0f, dk);
      store_transposed<HD>(dV, ld.dv, j0, L, 1.0f, dv);
    }
  }
  if (split) {                 // the two halves of the heavy tiles meet through the Q rows: dV, then dK through the same slots
    float* slot = lds + pl.simd * merge_floats<HD>();
    __syncthreads();
    if (pl.partial >= 0) park_acc<HD>(slot, dv);
    __syncthreads();
    if (pl.merge) {
      add_parked<HD>(slot, dv);
      store_transposed<HD>(dV, ld.dv, j0, L, 1.0f, dv);
    }
    __syncthreads();
    if (pl.partial >= 0) park_acc<HD>(slot, dk);
    __syncthreads();
    if (pl.merge) {
      add_parked<HD>(slot, dk);
      store_transposed<HD>(dK, ld.dk, j0, L, 1.0f, dk);
    }
  }
}

bool attn_mfma_supported(int lq, int lk, int hd, const float* mask, const float* probs) {
  return mask == nullptr && probs == nullptr && lq == lk && lq <= 256 && (hd == 32 || hd == 64);
}

}  // namespace rbx
#include "rbx_attn_stream.h"
#include "rbx_attn_planes.h"
namespace rbx {

// heavy tiles dealt to both wavefronts of their SIMD (wave_plan): causal sequences of at least three tiles (below that no
// tile is two steps heavier than its partner).  The four merge slots fit in
// the operand rows they reuse: 4 * (32 HD + 128) floats <= 2 * 96 * (HD + kPad) for HD = 32 and 64.
static bool attn_split(int L, int causal) {
  return causal != 0 && L > 2 * kT;
}

template <int HD>
static size_t lds_bytes(int L, bool phase_b) {
  const int Lp = (L + kT - 1) / kT * kT;
  return (static_cast<size_t>(2) * Lp * (HD + kPad) + (phase_b ? 2 * Lp : 0)) * sizeof(float);
}

template <int HD, bool DROP>
static int run_fwd(const float* q, const float* k, const float* v, long long bh, int L, float scale, int causal, float* o,
                   float* lse, const DropArgs& drop, const AttnLd& ld, hipStream_t s) {
  if constexpr (HD == 64) {
    // K / V streamed through a ring of 32-key tiles, a pair of sequences per workgroup (rbx_attn_stream.h)
    const int nT = (L + kT - 1) / kT;
    if (causal != 0 && nT == 7) {
      // seven tiles (192 < L <= 224: BASELINE cfg 5's L = 200): K / V as three bf16 planes split once per tile by the loader
      // (rbx_attn_planes.h): 339 vs 373 us at L = 200, 4096 sequences, the same 1e-7 error against float64
      // (profiles/r04/attn_planes.txt); the default since round 5
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_planes_fwd_kernel<DROP>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kPlanesLds));
      const long long pairs = (bh + 1) / 2;
      hipLaunchKernelGGL((attn_planes_fwd_kernel<DROP>), dim3(static_cast<unsigned>(pairs < kCUs ? pairs : kCUs)), dim3(512),
                         kPlanesLds, s, q, k, v, L, scale, o, lse, drop, ld, bh);
      return check_launch("attn_planes_fwd_kernel");
    }
    if (causal != 0 && nT >= 3 && nT <= 7) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_stream_fwd_kernel<DROP>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kStreamLds));
      hipLaunchKernelGGL((attn_stream_fwd_kernel<DROP>), dim3(static_cast<unsigned>((bh + 1) / 2)), dim3((nT + 1) * 64),
                         kStreamLds, s, q, k, v, L, scale, o, lse, drop, ld, bh);
      return check_launch("attn_stream_fwd_kernel");
    }
  }
  const size_t lds = lds_bytes<HD>(L, false);
  // (the looping form keeps the heavy tiles' partials in LDS slots of their own behind K and V: where those do not fit --
  //  HD = 64, L > 224 -- neither form splits, so that a sequence gets the same arithmetic from both)
  const size_t lds_split = lds + 4 * merge_floats<HD>() * sizeof(float);
  const bool split_on = attn_split(L, causal) && lds_split <= 160 * 1024;
  if constexpr (HD == 64) {
    if (lds > 80 * 1024 && bh > kCUs) {             // one workgroup per CU either way: loop over sequences, prefetch
      const int npf = ((L + kT - 1) / kT * kT) * (HD / 4) / kAttnThreads;      // 6, 7 or 8 at Lp = 192, 224, 256
      const int split = split_on ? 1 : 0;
      const size_t lds_pf = split ? lds_split : lds;
#define RBX_ATTN_PF(N)                                                                                                  \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_mfma_fwd_kernel<HD, DROP, N>),                        \
                                hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_pf));                   \
      hipLaunchKernelGGL((attn_mfma_fwd_kernel<HD, DROP, N>), dim3(kCUs), dim3(kAttnThreads), lds_pf, s, q, k, v, L, scale, \
                         causal, o, lse, drop, ld, bh, split)
      if (npf == 6) { RBX_ATTN_PF(6); } else if (npf == 7) { RBX_ATTN_PF(7); } else { RBX_ATTN_PF(8); }
#undef RBX_ATTN_PF
      return check_launch("attn_mfma_fwd_kernel");
    }
  }
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_mfma_fwd_kernel<HD, DROP, 0>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
  hipLaunchKernelGGL((attn_mfma_fwd_kernel<HD, DROP, 0>), dim3(static_cast<unsigned>(bh)), dim3(kAttnThreads), lds, s, q, k, v,
                     L, scale, causal, o, lse, drop, ld, bh, split_on ? 2 : 0);
  return check_launch("attn_mfma_fwd_kernel");
}

template <int HD, bool DROP>
static int run_bwd(const float* q, const float* k, const float* v, const float* o, const float* go, const float* lse,
                   long long bh, int L, float scale, int causal, float* dq, float* dk, float* dv, float* scratch,
                   const DropArgs& drop, const AttnLd& ld, hipStream_t s) {
  const size_t la = lds_bytes<HD>(L, false), lb = lds_bytes<HD>(L, true);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_mfma_bwd_q_kernel<HD, DROP>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(la));
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_mfma_bwd_kv_kernel<HD, DROP>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lb));
  const int split = attn_split(L, causal) ? 1 : 0;
  hipLaunchKernelGGL((attn_mfma_bwd_q_kernel<HD, DROP>), dim3(static_cast<unsigned>(bh)), dim3(kAttnThreads), la, s, q, k, v, o,
                     go, lse, L, scale, causal, dq, scratch, drop, ld, split);
  hipLaunchKernelGGL((attn_mfma_bwd_kv_kernel<HD, DROP>), dim3(static_cast<unsigned>(bh)), dim3(kAttnThreads), lb, s, q, k, v, go,
                     lse, scratch, L, scale, causal, dk, dv, drop, ld, split);
  return check_launch("attn_mfma_bwd kernels");
}

static AttnLd contiguous_ld(int hd) {
  const long long h = hd;
  return AttnLd{h, h, h, h, h, h, h, h, 1};
}

int attn_mfma_fwd_ld(const float* q, const float* k, const float* v, long long bh, int L, int hd, float scale, int causal,
                     float* o, float* lse, const DropArgs& drop, const AttnLd& ld, hipStream_t s) {
  if (drop.thr16 != 0)
    return hd == 64 ? run_fwd<64, true>(q, k, v, bh, L, scale, causal, o, lse, drop, ld, s)
                    : run_fwd<32, true>(q, k, v, bh, L, scale, causal, o, lse, drop, ld, s);
  return hd == 64 ? run_fwd<64, false>(q, k, v, bh, L, scale, causal, o, lse, drop, ld, s)
                  : run_fwd<32, false>(q, k, v, bh, L, scale, causal, o, lse, drop, ld, s);
}

int attn_mfma_bwd_ld(const float* q, const float* k, const float* v, const float* o, const float* go, const float* lse,
                     long long bh, int L, int hd, float scale, int causal, float* dq, float* dk, float* dv, float* scratch,
                     const DropArgs& drop, const AttnLd& ld, hipStream_t s) {
  if (drop.thr16 != 0)
    return hd == 64 ? run_bwd<64, true>(q, k, v, o, go, lse, bh, L, scale, causal, dq, dk, dv, scratch, drop, ld, s)
                    : run_bwd<32, true>(q, k, v, o, go, lse, bh, L, scale, causal, dq, dk, dv, scratch, drop, ld, s);
  return hd == 64 ? run_bwd<64, false>(q, k, v, o, go, lse, bh, L, scale, causal, dq, dk, dv, scratch, drop, ld, s)
                  : run_bwd<32, false>(q, k, v, o, go, lse, bh, L, scale, causal, dq, dk, dv, scratch, drop, ld, s);
}

int attn_mfma_fwd(const float* q, const float* k, const float* v, long long bh, int L, int hd, float scale, int causal,
                  float* o, float* lse, const DropArgs& drop, hipStream_t s) {
  return attn_mfma_fwd_ld(q, k, v, bh, L, hd, scale, causal, o, lse, drop, contiguous_ld(hd), s);
}

int attn_mfma_bwd(const float* q, const float* k, const float* v, const float* o, const float* go, const float* lse,
                  long long bh, int L, int hd, float scale, int causal, float* dq, float* dk, float* dv, float* scratch,
                  const DropArgs& drop, hipStream_t s) {
  return attn_mfma_bwd_ld(q, k, v, o, go, lse, bh, L, hd, scale, causal, dq, dk, dv, scratch, drop, contiguous_ld(hd), s);
}

}  // namespace rbx

/* ---- packed-operand entry points: read Q [B, L, ldq], K / V inside the fused projection, write O / dQ / dK / dV strided */
extern "C" int rbx_attn_packed_fwd(const float* d_q, int64_t ldq, const float* d_k, int64_t ldk, const float* d_v, int64_t ldv,
                                   int64_t batch, int32_t heads, int32_t seq_len, int32_t head_dim, float scale,
                                   int32_t causal, float p_drop, uint64_t seed, const uint64_t* d_seed_add, float* d_o,
                                   int64_t ldo, float* d_lse, void* stream) {
  using namespace rbx;
  if (batch == 0) return RBX_OK;
  if (batch < 0 || heads <= 0 || seq_len <= 0) return fail(RBX_ERR_INVALID, "attn_packed: bad shape");
  if (!attn_mfma_supported(seq_len, seq_len, head_dim, nullptr, nullptr))
    return fail(RBX_ERR_UNSUPPORTED, "attn_packed: needs seq_len <= 256 and head_dim in {32, 64} (got %d, %d)", seq_len, head_dim);
  if (!d_q || !d_k || !d_v || !d_o || !d_lse) return fail(RBX_ERR_INVALID, "attn_packed: NULL tensor");
  const long long need = static_cast<long long>(heads) * head_dim;
  if (ldq < need || ldk < need || ldv < need || ldo < need || ((ldq | ldk | ldv | ldo) & 3) ||
      ((reinterpret_cast<uintptr_t>(d_q) | reinterpret_cast<uintptr_t>(d_k) | reinterpret_cast<uintptr_t>(d_v) |
        reinterpret_cast<uintptr_t>(d_o)) & 15))
    return fail(RBX_ERR_INVALID, "attn_packed: row strides must cover heads * head_dim, be multiples of 4 floats, pointers 16-byte aligned");
  if (!(p_drop >= 0.f) || p_drop >= 1.f) return fail(RBX_ERR_INVALID, "attn_packed: dropout probability %g not in [0, 1)", static_cast<double>(p_drop));
  DropArgs drop;
  unsigned thr = static_cast<unsigned>(p_drop * 65536.0f + 0.5f);
  if (thr > 65535u) thr = 65535u;
  drop.thr16 = thr;
  drop.scale = 65536.0f / static_cast<float>(65536u - thr);
  drop.k0 = static_cast<unsigned>(seed);
  drop.k1 = static_cast<unsigned>(seed >> 32);
  drop.seed_add = reinterpret_cast<const unsigned long long*>(d_seed_add);
  const AttnLd ld{ldq, ldk, ldv, ldo, ldo, ldq, ldk, ldv, heads};
  return attn_mfma_fwd_ld(d_q, d_k, d_v, static_cast<long long>(batch) * heads, seq_len, head_dim, scale, causal, d_o, d_lse,
                          drop, ld, as_stream(stream));
}

extern "C" int rbx_attn_packed_bwd(const float* d_q, int64_t ldq, const float* d_k, int64_t ldk, const float* d_v, int64_t ldv,
                                   const float* d_o, int64_t ldo, const float* d_do, int64_t lddo, const float* d_lse,
                                   int64_t batch, int32_t heads, int32_t seq_len, int32_t head_dim, float scale,
                                   int32_t causal, float p_drop, uint64_t seed, const uint64_t* d_seed_add, float* d_dq,
                                   int64_t lddq, float* d_dk, int64_t lddk, float* d_dv, int64_t lddv, float* d_scratch,
                                   void* stream) {
  using namespace rbx;
  if (batch == 0) return RBX_OK;
  if (batch < 0 || heads <= 0 || seq_len <= 0) return fail(RBX_ERR_INVALID, "attn_packed_bwd: bad shape");
  if (!attn_mfma_supported(seq_len, seq_len, head_dim, nullptr, nullptr))
    return fail(RBX_ERR_UNSUPPORTED, "attn_packed_bwd: needs seq_len <= 256 and head_dim in {32, 64}");
  if (!d_q || !d_k || !d_v || !d_o || !d_do || !d_lse || !d_dq || !d_dk || !d_dv || !d_scratch)
    return fail(RBX_ERR_INVALID, "attn_packed_bwd: NULL tensor");
  const long long need = static_cast<long long>(heads) * head_dim;
  const long long lds_[8] = {ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv};
  for (int i = 0; i < 8; ++i)
    if (lds_[i] < need || (lds_[i] & 3)) return fail(RBX_ERR_INVALID, "attn_packed_bwd: bad row stride");
  if ((reinterpret_cast<uintptr_t>(d_q) | reinterpret_cast<uintptr_t>(d_k) | reinterpret_cast<uintptr_t>(d_v) |
       reinterpret_cast<uintptr_t>(d_o) | reinterpret_cast<uintptr_t>(d_do) | reinterpret_cast<uintptr_t>(d_dq) |
       reinterpret_cast<uintptr_t>(d_dk) | reinterpret_cast<uintptr_t>(d_dv)) & 15)
    return fail(RBX_ERR_INVALID, "attn_packed_bwd: pointers must be 16-byte aligned");
  if (!(p_drop >= 0.f) || p_drop >= 1.f) return fail(RBX_ERR_INVALID, "attn_packed_bwd: bad dropout probability");
  DropArgs drop;
  unsigned thr = static_cast<unsigned>(p_drop * 65536.0f + 0.5f);
  if (thr > 65535u) thr = 65535u;
  drop.thr16 = thr;
  drop.scale = 65536.0f / static_cast<float>(65536u - thr);
  drop.k0 = static_cast<unsigned>(seed);
  drop.k1 = static_cast<unsigned>(seed >> 32);
  drop.seed_add = reinterpret_cast<const unsigned long long*>(d_seed_add);
  const AttnLd ld{ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv, heads};
  return attn_mfma_bwd_ld(d_q, d_k, d_v, d_o, d_do, d_lse, static_cast<long long>(batch) * heads, seq_len, head_dim, scale,
                          causal, d_dq, d_dk, d_dv, d_scratch, drop, ld, as_stream(stream));
}
