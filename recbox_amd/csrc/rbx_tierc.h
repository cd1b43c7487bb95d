// rbx_tierc.h -- the large tables of the fused FM backward without a global sort ("tier C", round 4).
//
// Reference behaviour replaced: autograd's embedding_dense_backward behind the nn.Embedding tables of
// ranking/pytorch/layers/embeddings/feature_embedding.py:89-103 (dense [V, D] gradient, padding_idx row zero) for the
// HIGH-cardinality fields of a CTR batch, fused with the FM interaction's backward (interactions/inner_product.py:40-47):
//     dW[r] = sum_{b: id_b = r} g_b (S_b - w_r),   dLR[r] = sum g_b.
//
// Why a third path.  A field of a fused FM call looks up ONE id per sample, so a table of tier B collects B pairs per step
// (65 536 at the Criteo shape) and almost all of its touched rows are unique.  The sorted path spent, for the 12 such
// tables, a global segmented LSD radix sort (build_keys + 3 x (hist, scan, scatter): ten latency-bound launches, ~80 us),
// a segmented reduce that gathers S rows in row order (63-73 us, 1.40x its algorithmic bytes), two fix-up launches and a
// re-zero launch -- fifteen kernels of the step's 23.  Here ONE launch does it, with nothing global to synchronise on:
//
//   a workgroup owns (table, partition p of P = 2^k); it SCANS the table's compact id column (int32 [B], L2-resident:
//   256 KB per field), keeps the ids whose row hashes to p -- ~B / P = 1024 of them, written to an LDS list in an order
//   that is a pure function of the data (tile, then thread, then slot: an exclusive prefix over the workgroup, no atomics) --,
//   sorts that list by row with stable 8-bit LSD passes in LDS (ta_radix_pass of rbx_tiera.h; keys row << 11 | list
//   position, so equal rows stay in list order), and walks it: a lane group per run of equal rows adds g_b S_b in that
//   fixed order, subtracts cnt * w_r and STORES the row.  No row is shared by two workgroups (rows are partitioned by
//   hash), so there are no fix-ups; every sum has a fixed order, so the gradient is bit-identical from run to run.
//   Runs longer than 32 pairs (a hot id of a skewed batch) are summed by the whole workgroup, element k by lane group
//   k mod NG, partials added in lane-group order.  A partition that selects more than 2048 pairs (all ids equal, say)
//   is processed in several list fills; fills after the first add into the rows (read-modify-write; the rows of one
//   workgroup are its own).
//
//   The rows a workgroup wrote go to its row list in the workspace ([partition][cap] + a count): rbx_fm_rezero clears
//   exactly those rows of a persistent gradient buffer (tc_rezero_kernel), as rezero_rows_kernel does from sorted keys.
//
// Cost model at the Criteo shape (12 tables, 768 workgroups): 64 workgroups scan each 256 KB column = 200 MB of L2 reads;
// per pair 64 B of S (L2 / MALL), 64 B of w_r and a 64-byte store to HBM at random -- the same random-row traffic as the
// sorted reduce, minus the sort's 10 launches, the fix-ups, and the key / value arrays (12.6 MB written and read 3.5 times).
#pragma once
#include "rbx_tiera.h"

namespace rbx {

constexpr int kTcList = 2048;                     // pairs a workgroup sorts at a time: ta_radix_pass's tile
constexpr int kTcIdxBits = 11;
constexpr unsigned kTcIdxMask = (1u << kTcIdxBits) - 1u;
constexpr int kTcTarget = 1024;                   // expected pairs per partition (half a list: skew rarely needs a second fill)
constexpr int kTcMaxLogP = 8;                     // at most 256 partitions per table
constexpr int kTcMaxVocab = (1 << 21) - 1;        // row << 11 | position must stay below the all-ones filler key
constexpr int kTcLong = 32;                       // runs longer than this are summed by the whole workgroup
constexpr int kTcMaxLong = 64;                    // long runs one list fill can hold (more: handled by their head's lane group)
constexpr unsigned kTcOverflow = 0xFFFFFFFFu;     // row-list count: "more rows than the list holds: clear by hash"

struct TcTable {             // 48 B
  float* grad;               // [V, D]
  float* grad2;              // [V] (LR weight gradient) or NULL
  const float* table;        // [V, stride] embedding rows (dW = A - cnt * w)
  int stride;
  int vocab;
  int pad;                   // padding_idx (kNoId when unset)
  int cid_row;               // row of the compact id matrix
  unsigned part0;            // first workgroup of this table
  int log_p;                 // the table has 1 << log_p partitions
};
struct TcPack { TcTable t[RBX_MAX_FIELDS]; };
static_assert(sizeof(TcPack) + 128 <= 4096, "tc_reduce_kernel's arguments must fit the kernarg segment");

struct TcPlan {
  int n_tab = 0;
  TcPack tab;
  unsigned n_parts = 0;      // workgroups of a launch
  unsigned cap = 0;          // ints per row list
  int D = 1;
  size_t off_counts = 0, off_lists = 0, bytes = 0;      // relative to the region
};

static inline void tc_layout(TcPlan* t) {
  size_t o = 0;
  t->off_counts = o; o += ta_align(static_cast<size_t>(t->n_parts) * 4);
  t->off_lists = o; o += ta_align(static_cast<size_t>(t->n_parts) * t->cap * 4);
  t->bytes = t->n_tab ? o : 0;
}

__device__ __forceinline__ unsigned tc_part_of(unsigned row, int log_p) {
  return log_p == 0 ? 0u : (row * 2654435761u) >> (32 - log_p);
}

__device__ __forceinline__ int tc_table_of(const TcPack& P, int n_tab, unsigned wg) {
  int t = 0;
  while (t + 1 < n_tab && P.t[t + 1].part0 <= wg) ++t;
  return t;
}

// a float4 of a gradient row read around the L1 (a row this workgroup stored in an earlier list fill of the same launch)
__device__ __forceinline__ float tc_load_coherent(const float* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int G>
__global__ __launch_bounds__(256, 3) void tc_reduce_kernel(const TcPack P, const int n_tab, const long long B, const int D,
                                                        const int* __restrict__ cid, const float* __restrict__ g,
                                                        const float* __restrict__ ssum, const int accumulate,
                                                        unsigned* __restrict__ counts, unsigned* __restrict__ lists,
                                                        const unsigned cap) {
  using F = Frag<G, 1, true>;
  constexpr int NG = 256 / G;
  __shared__ unsigned buf[kTcList];               // rows of the list being filled, then the sorted keys
  __shared__ unsigned lb[kTcList];                // sample of list entry i
  __shared__ unsigned wcnt[4][256];
  __shared__ unsigned dstart[256];
  __shared__ unsigned wtot[4];
  __shared__ unsigned s_wave[4];
  __shared__ unsigned s_out;                      // rows this workgroup has written so far (row-list fill)
  __shared__ unsigned s_nlong;
  __shared__ unsigned s_long[kTcMaxLong][2];      // (first sorted position, length) of the long runs of this fill
  __shared__ float s_part[NG][G * 4 + 1];         // partials of a long run, one per lane group

  const int t = tc_table_of(P, n_tab, blockIdx.x);
  const TcTable tb = P.t[t];
  const unsigned part = blockIdx.x - tb.part0;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int lane_g = threadIdx.x % G, group = threadIdx.x / G;
  const int* col = cid + static_cast<size_t>(tb.cid_row) * B;
  unsigned* mylist = lists + static_cast<size_t>(blockIdx.x) * cap;
  int nbits = 1;
  while ((1ll << nbits) < tb.vocab) ++nbits;
  const int passes = (nbits + 7) / 8;
  if (threadIdx.x == 0) s_out = 0;
  unsigned n_list = 0;                            // (uniform over the workgroup)
  bool later = false;                             // a list of this workgroup has been reduced already

  // ---- reduce the list that is in buf / lb: sort by row, then one lane group per run ----
  auto flush = [&](const unsigned n) {
    unsigned key[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const unsigned off = wid * 512 + s * 64 + lane;
      key[s] = (off < n) ? ((buf[off] << kTcIdxBits) | off) : 0xFFFFFFFFu;
    }
    if (threadIdx.x == 0) s_nlong = 0;
    for (int q = 0; q < passes; ++q) ta_radix_pass(key, kTcIdxBits + 8 * q, wcnt, dstart, wtot, buf);
    // (ta_radix_pass ends on a barrier: buf holds the keys in sorted order)
    const bool rmw = accumulate != 0 || later;
    constexpr int U = 4;
    for (unsigned i0 = group; i0 < n; i0 += U * NG) {
      unsigned row[U], b[U], len[U];
      bool head[U];
      float gg[U];
      F acc[U], w[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const unsigned i = i0 + u * NG;
        head[u] = false;
        len[u] = 0;
        row[u] = 0;
        b[u] = 0;
        if (i < n) {
          const unsigned k = buf[i];
          row[u] = k >> kTcIdxBits;
          b[u] = lb[k & kTcIdxMask];
          head[u] = (i == 0) || ((buf[i - 1] >> kTcIdxBits) != row[u]);
          if (head[u]) {
            unsigned j = i + 1;
            while (j < n && (buf[j] >> kTcIdxBits) == row[u]) ++j;
            len[u] = j - i;
          }
        }
      }
      // every first-element load of the batch is issued before the first use
#pragma unroll
      for (int u = 0; u < U; ++u) {
        acc[u].zero();
        w[u].zero();
        gg[u] = 0.f;
        if (head[u] && len[u] <= kTcLong) {
          gg[u] = g[b[u]];
          acc[u].fma_from(ssum + static_cast<size_t>(b[u]) * D, D, lane_g, 1.0f);
          w[u].add_from_nt(tb.table + static_cast<size_t>(row[u]) * tb.stride, D, lane_g);
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (!head[u]) continue;
        const unsigned i = i0 + u * NG;
        if (len[u] > kTcLong) {
          unsigned slot = kTcMaxLong;
          if (lane_g == 0) slot = atomicAdd(&s_nlong, 1u);
          slot = __shfl(slot, (lane / G) * G, 64);
          if (slot < kTcMaxLong) {
            if (lane_g == 0) { s_long[slot][0] = i; s_long[slot][1] = len[u]; }
            continue;
          }
          // (more long runs than the table holds: this lane group walks the run itself)
          gg[u] = g[b[u]];
          acc[u].fma_from(ssum + static_cast<size_t>(b[u]) * D, D, lane_g, 1.0f);
          w[u].add_from_nt(tb.table + static_cast<size_t>(row[u]) * tb.stride, D, lane_g);
        }
        float cnt = gg[u];
#pragma unroll
        for (int q = 0; q < G * 0 + 4; ++q) acc[u].a[q] *= gg[u];
        for (unsigned j = i + 1; j < i + len[u]; ++j) {
          const unsigned bj = lb[buf[j] & kTcIdxMask];
          const float gj = g[bj];
          acc[u].fma_from(ssum + static_cast<size_t>(bj) * D, D, lane_g, gj);
          cnt += gj;
        }
        F out = acc[u];
#pragma unroll
        for (int q = 0; q < 4; ++q) out.a[q] -= cnt * w[u].a[q];
        float* dst = tb.grad + static_cast<size_t>(row[u]) * D;
        if (rmw) {
          const int e = lane_g * 4;
          if (e < D) {
            float4 v;
            v.x = tc_load_coherent(dst + e) + out.a[0];
            v.y = tc_load_coherent(dst + e + 1) + out.a[1];
            v.z = tc_load_coherent(dst + e + 2) + out.a[2];
            v.w = tc_load_coherent(dst + e + 3) + out.a[3];
            *reinterpret_cast<float4*>(dst + e) = v;
          }
        } else {
          out.store_nt(dst, D, lane_g);
        }
        if (lane_g == 0) {
          if (tb.grad2 != nullptr) {
            if (rmw) tb.grad2[row[u]] = tc_load_coherent(tb.grad2 + row[u]) + cnt; else tb.grad2[row[u]] = cnt;
          }
          const unsigned pos = atomicAdd(&s_out, 1u);
          if (pos < cap) mylist[pos] = row[u];
        }
      }
    }
    __syncthreads();
    // ---- long runs: the whole workgroup, element k of the run by lane group k mod NG, partials added in group order ----
    const unsigned nl = (s_nlong < static_cast<unsigned>(kTcMaxLong)) ? s_nlong : static_cast<unsigned>(kTcMaxLong);
    for (unsigned q = 0; q < nl; ++q) {
      // (the slots were handed out by an atomic: walk them in the order of their first position, so that the row list
      //  and nothing else depends on arrival order -- the sums do not: each run is reduced on its own)
      const unsigned first = s_long[q][0], len = s_long[q][1];
      const unsigned row = buf[first] >> kTcIdxBits;
      F acc;
      acc.zero();
      float cnt = 0.f;
      for (unsigned j = first + group; j < first + len; j += NG) {
        const unsigned bj = lb[buf[j] & kTcIdxMask];
        const float gj = g[bj];
        acc.fma_from(ssum + static_cast<size_t>(bj) * D, D, lane_g, gj);
        cnt += gj;
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) s_part[group][lane_g * 4 + c] = acc.a[c];
      if (lane_g == 0) s_part[group][G * 4] = cnt;
      __syncthreads();
      if (group == 0) {
        F tot;
        tot.zero();
        float ctot = 0.f;
        for (int k = 0; k < NG; ++k) {
#pragma unroll
          for (int c = 0; c < 4; ++c) tot.a[c] += s_part[k][lane_g * 4 + c];
          ctot += s_part[k][G * 4];
        }
        F w;
        w.zero();
        w.add_from_nt(tb.table + static_cast<size_t>(row) * tb.stride, D, lane_g);
#pragma unroll
        for (int c = 0; c < 4; ++c) tot.a[c] -= ctot * w.a[c];
        float* dst = tb.grad + static_cast<size_t>(row) * D;
        const int e = lane_g * 4;
        if (rmw) {
          if (e < D) {
            float4 v;
            v.x = tc_load_coherent(dst + e) + tot.a[0];
            v.y = tc_load_coherent(dst + e + 1) + tot.a[1];
            v.z = tc_load_coherent(dst + e + 2) + tot.a[2];
            v.w = tc_load_coherent(dst + e + 3) + tot.a[3];
            *reinterpret_cast<float4*>(dst + e) = v;
          }
        } else {
          tot.store_nt(dst, D, lane_g);
        }
        if (lane_g == 0) {
          if (tb.grad2 != nullptr) {
            if (rmw) tb.grad2[row] = tc_load_coherent(tb.grad2 + row) + ctot; else tb.grad2[row] = ctot;
          }
          const unsigned pos = atomicAdd(&s_out, 1u);
          if (pos < cap) mylist[pos] = row;
        }
      }
      __syncthreads();
    }
    __threadfence();                 // the rows stored by this fill are visible to the read-modify-writes of the next one
    __syncthreads();
  };

  // ---- scan the id column in tiles of 2048; selected ids go to the list in (tile, thread, slot) order ----
  // (one call site of the list reduce: a tile that does not fit is put back and read again after the flush)
  long long base = 0;
  while (true) {
    bool do_flush = false;
    if (base < B) {
      int id[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const long long b = base + u * 256 + threadIdx.x;
        id[u] = (b < B) ? col[b] : -1;
      }
      unsigned mask = 0;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const bool sel = id[u] >= 0 && id[u] != tb.pad && id[u] < tb.vocab &&
                         tc_part_of(static_cast<unsigned>(id[u]), tb.log_p) == part;
        mask |= (sel ? 1u : 0u) << u;
      }
      const unsigned mine = __popc(mask);
      unsigned inc = mine;                                           // inclusive prefix over the wavefront
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const unsigned v = __shfl_up(inc, o, 64);
        if (lane >= o) inc += v;
      }
      if (lane == 63) s_wave[wid] = inc;
      __syncthreads();
      unsigned before = inc - mine, total = 0;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const unsigned c = s_wave[w];
        if (w < wid) before += c;
        total += c;
      }
      __syncthreads();                                               // (s_wave is rewritten by the next tile)
      if (n_list + total > static_cast<unsigned>(kTcList)) {         // (uniform) no room: reduce what is there, then retry
        do_flush = true;
      } else {
        unsigned pos = n_list + before;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if ((mask >> u) & 1u) {
            buf[pos] = static_cast<unsigned>(id[u]);
            lb[pos] = static_cast<unsigned>(base + u * 256 + threadIdx.x);
            ++pos;
          }
        }
        n_list += total;
        base += kTcList;
      }
    } else {
      do_flush = n_list > 0;
    }
    if (do_flush) {
      __syncthreads();
      flush(n_list);
      n_list = 0;
      later = true;
    }
    if (base >= B && n_list == 0) break;
  }
  __syncthreads();
  if (threadIdx.x == 0) counts[blockIdx.x] = (s_out <= cap) ? s_out : kTcOverflow;
}

// ---- clear the rows the previous backward wrote (persistent gradient buffers) ------------------------------------------
template <int G>
__global__ __launch_bounds__(256) void tc_rezero_kernel(const TcPack P, const int n_tab, const int D,
                                                        unsigned* __restrict__ counts, const unsigned* __restrict__ lists,
                                                        const unsigned cap) {
  constexpr int NG = 256 / G;
  const int t = tc_table_of(P, n_tab, blockIdx.x);
  const TcTable tb = P.t[t];
  const unsigned part = blockIdx.x - tb.part0;
  const int lane_g = threadIdx.x % G, group = threadIdx.x / G;
  const unsigned c = counts[blockIdx.x];
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  const int e = lane_g * 4;
  if (c == kTcOverflow) {                        // the list did not hold every row: every row of this partition
    for (unsigned r = group; r < static_cast<unsigned>(tb.vocab); r += NG) {
      if (tc_part_of(r, tb.log_p) != part) continue;
      if (e < D) *reinterpret_cast<float4*>(tb.grad + static_cast<size_t>(r) * D + e) = z;
      if (lane_g == 0 && tb.grad2 != nullptr) tb.grad2[r] = 0.f;
    }
  } else {
    const unsigned* mylist = lists + static_cast<size_t>(blockIdx.x) * cap;
    for (unsigned i = group; i < c; i += NG) {
      const unsigned r = mylist[i];
      if (e < D) *reinterpret_cast<float4*>(tb.grad + static_cast<size_t>(r) * D + e) = z;
      if (lane_g == 0 && tb.grad2 != nullptr) tb.grad2[r] = 0.f;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) counts[blockIdx.x] = 0;
}

template <int G>
static int tc_launch_bwd(const TcPlan& t, int64_t B, const int* cid, const float* g, const float* ssum, int accumulate,
                         char* region, hipStream_t s) {
  hipLaunchKernelGGL(tc_reduce_kernel<G>, dim3(t.n_parts), dim3(256), 0, s, t.tab, t.n_tab, static_cast<long long>(B), t.D,
                     cid, g, ssum, accumulate, reinterpret_cast<unsigned*>(region + t.off_counts),
                     reinterpret_cast<unsigned*>(region + t.off_lists), t.cap);
  return check_launch("tc_reduce_kernel");
}

static inline int tc_dispatch_bwd(const TcPlan& t, int64_t B, const int* cid, const float* g, const float* ssum,
                                  int accumulate, char* region, hipStream_t s) {
  if (t.n_tab == 0) return RBX_OK;
  switch (pow2_ceil((t.D + 3) / 4)) {
    case 1: return tc_launch_bwd<1>(t, B, cid, g, ssum, accumulate, region, s);
    case 2: return tc_launch_bwd<2>(t, B, cid, g, ssum, accumulate, region, s);
    case 4: return tc_launch_bwd<4>(t, B, cid, g, ssum, accumulate, region, s);
    case 8: return tc_launch_bwd<8>(t, B, cid, g, ssum, accumulate, region, s);
    default: return tc_launch_bwd<16>(t, B, cid, g, ssum, accumulate, region, s);
  }
}

template <int G>
static int tc_launch_rezero(const TcPlan& t, char* region, hipStream_t s) {
  hipLaunchKernelGGL(tc_rezero_kernel<G>, dim3(t.n_parts), dim3(256), 0, s, t.tab, t.n_tab, t.D,
                     reinterpret_cast<unsigned*>(region + t.off_counts),
                     reinterpret_cast<const unsigned*>(region + t.off_lists), t.cap);
  return check_launch("tc_rezero_kernel");
}

static inline int tc_dispatch_rezero(const TcPlan& t, char* region, hipStream_t s) {
  if (t.n_tab == 0) return RBX_OK;
  switch (pow2_ceil((t.D + 3) / 4)) {
    case 1: return tc_launch_rezero<1>(t, region, s);
    case 2: return tc_launch_rezero<2>(t, region, s);
    case 4: return tc_launch_rezero<4>(t, region, s);
    case 8: return tc_launch_rezero<8>(t, region, s);
    default: return tc_launch_rezero<16>(t, region, s);
  }
}

}  // namespace rbx
