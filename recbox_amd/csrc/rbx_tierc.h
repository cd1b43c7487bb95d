// rbx_tierc.h -- the large tables of the fused FM backward without a global multi-pass sort ("tier C", round 4).
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
// re-zero launch -- fifteen kernels of the step's 23.  Here:
//
//   ids only (side stream, beside the forward kernel):  ONE partition pass over the compact id matrix --
//     tc_count    a workgroup per (table, block of 2048 samples): histogram of the partition  p = row & (P - 1)
//     tc_scan     a workgroup per table: where every (block, partition) run starts inside the table's bucket array
//     tc_scatter  the same grid: (row, sample) pairs to their buckets; inside a bucket the pairs stay in
//                 (block, wavefront, step, lane) order -- a pure function of the data, no atomics decide a position
//   after the loss:
//     tc_reduce   a workgroup per (table, partition), ~B / P = 1024 pairs: loads its bucket, sorts it by row >> log2 P
//                 with stable 8-bit LSD passes in LDS (ta_radix_pass of rbx_tiera.h: keys  row' << 11 | position, two
//                 passes for a 1 M-row table), then lane groups walk chunks of 16 sorted pairs with every S row, w row
//                 and g of a batch of 8 requested before the first is used: a run of equal rows inside a chunk is
//                 summed, w_r * cnt subtracted and the row STORED; the first and the last run of every chunk go to an
//                 LDS list whose stretches of equal rows are closed after a barrier (the scheme of ta_reduce_kernel).
//                 No row is shared by two workgroups, so there are no fix-up launches; every sum has a fixed order, so
//                 the gradient is bit-identical from run to run.  A bucket of more than 2048 pairs (a hot id, a skewed
//                 batch) is taken 2048 pairs at a time; fills after the first add into the rows (read-modify-write: the
//                 rows of one workgroup are its own).
//   persistent gradient buffers: the bucket arrays still name every touched row when the next step begins --
//     tc_rezero   clears them (rbx_fm_rezero), as rezero_rows_kernel does from sorted keys.
//
// (First form of this tier, measured and replaced -- profiles/r04/INDEX.md: every (table, partition) workgroup SCANNED the
//  table's whole id column -- 60 us of the launch -- and walked its runs with loads issued behind branches, one round trip
//  per pair: 279 us per launch where the sorted reduce took 73.)
#pragma once
#include "rbx_tiera.h"

namespace rbx {

constexpr int kTcList = 2048;                     // pairs a workgroup sorts at a time: ta_radix_pass's tile
constexpr int kTcIdxBits = 11;
constexpr unsigned kTcIdxMask = (1u << kTcIdxBits) - 1u;
constexpr int kTcTarget = 1024;                   // expected pairs per partition
constexpr int kTcMaxLogP = 7;                     // at most 128 partitions per table (digit 255 = "no pair")
constexpr int kTcMaxVocab = (1 << 21) - 1;        // row' << 11 | position must stay below the all-ones filler key
constexpr int kTcChunk = 16;                      // sorted pairs one lane group walks in sequence
constexpr int kTcChunks = kTcList / kTcChunk;     // 128
constexpr unsigned kTcNone = 0xFFFFFFFEu;         // "no element" in the boundary-run list
constexpr unsigned kTcNoRow = 0x7FFFFFFFu;        // row of a filler key

struct TcTable {             // 48 B
  float* grad;               // [V, D]
  float* grad2;              // [V] (LR weight gradient) or NULL
  const float* table;        // [V, stride] embedding rows (dW = A - cnt * w)
  int stride;
  int vocab;
  int pad;                   // padding_idx (kNoId when unset)
  int cid_row;               // row of the compact id matrix
  unsigned part0;            // first workgroup of this table in tc_reduce's grid
  int reserved;
};
struct TcPack { TcTable t[RBX_MAX_FIELDS]; };
static_assert(sizeof(TcPack) + 128 <= 4096, "the tier-C kernels' arguments must fit the kernarg segment");

struct TcPlan {
  int n_tab = 0;
  TcPack tab;
  int log_p = 0;             // every table has 1 << log_p partitions
  unsigned n_parts = 0;      // workgroups of tc_reduce
  unsigned NB = 0;           // blocks of 2048 samples
  int D = 1;
  // region layout (bytes, relative): hist u32 [n_tab][NB][P] (after tc_scan: the start of every (block, partition) run),
  // start u32 [n_tab][P], count u32 [n_tab][P], prev u32 [n_tab], rows u32 [n_tab][B], smp u32 [n_tab][B]
  size_t off_hist = 0, off_start = 0, off_count = 0, off_prev = 0, off_rows = 0, off_smp = 0, bytes = 0;
};

static inline void tc_layout(TcPlan* t, int64_t B) {
  const size_t P = static_cast<size_t>(1) << t->log_p;
  size_t o = 0;
  t->off_hist = o; o += ta_align(static_cast<size_t>(t->n_tab) * t->NB * P * 4);
  t->off_start = o; o += ta_align(static_cast<size_t>(t->n_tab) * P * 4);
  t->off_count = o; o += ta_align(static_cast<size_t>(t->n_tab) * P * 4);
  t->off_prev = o; o += ta_align(static_cast<size_t>(t->n_tab) * 4);      // pairs the PREVIOUS partition pass placed, per table
  t->off_rows = o; o += ta_align(static_cast<size_t>(t->n_tab) * static_cast<size_t>(B) * 4);
  t->off_smp = o; o += ta_align(static_cast<size_t>(t->n_tab) * static_cast<size_t>(B) * 4);
  t->bytes = t->n_tab ? o : 0;
}

// ---- the 2048 ids of a (table, block) unit as partition digits, in (wavefront, step, lane) order ---------------------
__device__ __forceinline__ void tc_load_digits(const TcTable& tb, const int* __restrict__ cid, const long long B,
                                               const unsigned k, const int log_p, int (&id)[8], unsigned (&dg)[8]) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int* col = cid + static_cast<size_t>(tb.cid_row) * B;
  const long long b0 = static_cast<long long>(k) * kTcList;
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const long long b = b0 + wid * 512 + s * 64 + lane;
    id[s] = (b < B) ? col[b] : -1;
  }
  const unsigned mask = (1u << log_p) - 1u;
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const bool ok = id[s] >= 0 && id[s] != tb.pad && id[s] < tb.vocab;
    dg[s] = ok ? (static_cast<unsigned>(id[s]) & mask) : 255u;
  }
}

// rank of every key among the keys of its digit inside this wavefront (steps in order, lanes in order), and the
// wavefront's count per digit in wcnt[wid][digit] -- the first half of ta_radix_pass
__device__ __forceinline__ void tc_rank_digits(const unsigned (&dg)[8], unsigned (*wcnt)[256], unsigned (&rank)[8]) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 4 * 256; i += 256) (&wcnt[0][0])[i] = 0;
  __syncthreads();
  const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const unsigned d = dg[s];
    unsigned long long peers = ~0ull;
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const unsigned long long m = __ballot((d >> b) & 1u);
      peers &= ((d >> b) & 1u) ? m : ~m;
    }
    const unsigned before = __popcll(peers & lt);
    volatile unsigned* wc = wcnt[wid];
    const unsigned prev = wc[d];
    rank[s] = prev + before;
    if (before == 0) wc[d] = prev + __popcll(peers);
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void tc_count_kernel(const TcPack P, const long long B, const int log_p, const unsigned NB,
                                                       const int* __restrict__ cid, unsigned* __restrict__ hist) {
  __shared__ unsigned wcnt[4][256];
  const int t = blockIdx.y;
  const unsigned k = blockIdx.x;
  const TcTable tb = P.t[t];
  int id[8];
  unsigned dg[8], rank[8];
  tc_load_digits(tb, cid, B, k, log_p, id, dg);
  tc_rank_digits(dg, wcnt, rank);
  const unsigned np = 1u << log_p;
  if (threadIdx.x < np)
    hist[(static_cast<size_t>(t) * NB + k) * np + threadIdx.x] =
        wcnt[0][threadIdx.x] + wcnt[1][threadIdx.x] + wcnt[2][threadIdx.x] + wcnt[3][threadIdx.x];
}

// one workgroup per table: hist[t][k][p] -> where the run of (block k, partition p) starts inside ITS PARTITION's bucket
// (blocks in order); start[t][p] = first pair of partition p inside the table's bucket array, count[t][p]
__global__ __launch_bounds__(256) void tc_scan_kernel(const int log_p, const unsigned NB, unsigned* __restrict__ hist,
                                                      unsigned* __restrict__ start, unsigned* __restrict__ count,
                                                      unsigned* __restrict__ prev, const int keep_prev) {
  __shared__ unsigned tot[256];
  const int t = blockIdx.x;
  const unsigned np = 1u << log_p;
  const unsigned p = threadIdx.x;
  if (threadIdx.x == 0)      // how many pairs the previous pass placed (tc_scatter clears their rows while it overwrites them)
    prev[t] = keep_prev ? start[static_cast<size_t>(t) * np + np - 1] + count[static_cast<size_t>(t) * np + np - 1] : 0u;
  __syncthreads();
  unsigned* h = hist + static_cast<size_t>(t) * NB * np;
  unsigned run = 0;
  if (p < np) {
    for (unsigned k0 = 0; k0 < NB; k0 += 8) {                         // eight loads in flight, then their prefixes
      unsigned c[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) c[u] = (k0 + u < NB) ? h[static_cast<size_t>(k0 + u) * np + p] : 0u;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (k0 + u < NB) h[static_cast<size_t>(k0 + u) * np + p] = run;
        run += c[u];
      }
    }
  }
  tot[threadIdx.x] = (p < np) ? run : 0u;
  __syncthreads();
  unsigned base = 0;
  for (unsigned q = 0; q < p && q < np; ++q) base += tot[q];          // (np <= 128: a short serial prefix per thread)
  if (p < np) {
    start[static_cast<size_t>(t) * np + p] = base;
    count[static_cast<size_t>(t) * np + p] = run;
  }
}

__global__ __launch_bounds__(256) void tc_scatter_kernel(const TcPack P, const long long B, const int log_p,
                                                         const unsigned NB, const int* __restrict__ cid,
                                                         const unsigned* __restrict__ hist, const unsigned* __restrict__ start,
                                                         const unsigned* __restrict__ count, const unsigned* __restrict__ prev,
                                                         unsigned* __restrict__ rows, unsigned* __restrict__ smp,
                                                         const int clear_prev, const int D) {
  __shared__ unsigned wcnt[4][256];
  __shared__ unsigned first[256];                    // where this block's run of partition d starts in the bucket array
  const int t = blockIdx.y;
  const unsigned k = blockIdx.x;
  const TcTable tb = P.t[t];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const unsigned np = 1u << log_p;
  {                                                  // (requested first: the ids' and these loads' latencies overlap)
    const unsigned d = threadIdx.x < np ? threadIdx.x : 0u;
    first[threadIdx.x] = start[static_cast<size_t>(t) * np + d] + hist[(static_cast<size_t>(t) * NB + k) * np + d];
  }
  int id[8];
  unsigned dg[8], rank[8];
  tc_load_digits(tb, cid, B, k, log_p, id, dg);
  tc_rank_digits(dg, wcnt, rank);
  {                                                  // exclusive prefix over the wavefronts, per digit
    const int d = threadIdx.x;
    unsigned run = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const unsigned c = wcnt[w][d];
      wcnt[w][d] = run;
      run += c;
    }
  }
  __syncthreads();
  unsigned* rdst = rows + static_cast<size_t>(t) * B;
  unsigned* sdst = smp + static_cast<size_t>(t) * B;
  // clear_prev (persistent gradient buffers): the bucket array still names the rows the PREVIOUS backward stored.  Every
  // position this pass overwrites is read first and that row cleared here -- beside the forward kernel, instead of a
  // re-zero launch of its own at the end of the step's critical path; positions the previous pass filled and this one
  // does not reach (it placed fewer pairs) are cleared by the table's first block.
  const unsigned n_prev = clear_prev ? prev[t] : 0u;
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  unsigned pos[8], old[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    pos[s] = (dg[s] == 255u) ? 0xFFFFFFFFu : first[dg[s]] + wcnt[wid][dg[s]] + rank[s];
    old[s] = (pos[s] < n_prev) ? rdst[pos[s]] : 0xFFFFFFFFu;
  }
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    if (old[s] != 0xFFFFFFFFu) {
      float4* g4 = reinterpret_cast<float4*>(tb.grad + static_cast<size_t>(old[s]) * D);
      for (int q = 0; q < D / 4; ++q) g4[q] = z;
      if (tb.grad2 != nullptr) tb.grad2[old[s]] = 0.f;
    }
  }
  if (clear_prev && k == 0) {
    const unsigned n_now = start[static_cast<size_t>(t) * np + np - 1] + count[static_cast<size_t>(t) * np + np - 1];
    for (unsigned i = n_now + threadIdx.x; i < n_prev; i += 256) {
      const unsigned r = rdst[i];
      float4* g4 = reinterpret_cast<float4*>(tb.grad + static_cast<size_t>(r) * D);
      for (int q = 0; q < D / 4; ++q) g4[q] = z;
      if (tb.grad2 != nullptr) tb.grad2[r] = 0.f;
    }
  }
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    if (pos[s] == 0xFFFFFFFFu) continue;
    rdst[pos[s]] = static_cast<unsigned>(id[s]);
    sdst[pos[s]] = static_cast<unsigned>(static_cast<unsigned long long>(k) * kTcList + wid * 512 + s * 64 + lane);
  }
}

// a float of a gradient row read around the L1 (a row this workgroup stored in an earlier list fill of the same launch)
__device__ __forceinline__ float tc_load_coherent(const float* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int G>
__device__ __forceinline__ void tc_finish_row(const TcTable& tb, const unsigned row, const Frag<G, 1, true>& acc, const float cnt,
                                              const Frag<G, 1, true>& w, const int D, const int lane_g, const bool rmw) {
  Frag<G, 1, true> out = acc;
#pragma unroll
  for (int q = 0; q < 4; ++q) out.a[q] -= cnt * w.a[q];
  float* dst = tb.grad + static_cast<size_t>(row) * D;
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
  if (lane_g == 0 && tb.grad2 != nullptr) {
    if (rmw) tb.grad2[row] = tc_load_coherent(tb.grad2 + row) + cnt; else tb.grad2[row] = cnt;
  }
}

template <int G>
__global__ __launch_bounds__(256, 2) void tc_reduce_kernel(const TcPack P, const int n_tab, const long long B, const int D,
                                                        const int log_p, const float* __restrict__ g,
                                                        const float* __restrict__ ssum, const int accumulate,
                                                        const unsigned* __restrict__ start, const unsigned* __restrict__ count,
                                                        const unsigned* __restrict__ rows, const unsigned* __restrict__ smp) {
  using F = Frag<G, 1, true>;
  constexpr int NG = 256 / G;
  constexpr int NE = 2 * kTcChunks;
  __shared__ unsigned buf[kTcList];               // sorted keys of the fill
  __shared__ unsigned lb[kTcList];                // sample of the fill's entry i
  __shared__ unsigned wcnt[4][256];
  __shared__ unsigned dstart[256];
  __shared__ unsigned wtot[4];
  __shared__ __attribute__((aligned(16))) float esum[NE][G * 4];
  __shared__ float ecnt[NE];
  __shared__ unsigned erow[NE];

  const unsigned np = 1u << log_p;
  const int t = blockIdx.x >> log_p;
  const unsigned part = blockIdx.x & (np - 1u);
  const TcTable tb = P.t[t];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int lane_g = threadIdx.x % G, group = threadIdx.x / G;
  const unsigned first = start[static_cast<size_t>(t) * np + part];
  const unsigned n_all = count[static_cast<size_t>(t) * np + part];
  const unsigned* rsrc = rows + static_cast<size_t>(t) * B + first;
  const unsigned* ssrc = smp + static_cast<size_t>(t) * B + first;
  int nbits = 1;
  while ((1ll << nbits) < tb.vocab) ++nbits;
  const int passes = (nbits - log_p + 7) / 8 > 0 ? (nbits - log_p + 7) / 8 : 1;

  for (unsigned c0 = 0; c0 < n_all; c0 += kTcList) {
    const unsigned n = (n_all - c0 < static_cast<unsigned>(kTcList)) ? n_all - c0 : static_cast<unsigned>(kTcList);
    const bool rmw = accumulate != 0 || c0 > 0;
    // ---- the fill: keys row' << 11 | position into registers, samples into LDS ----
    // (no branch in front of a load -- the compiler waits for a load inside the block that issues it: all sixteen of a
    //  thread are requested, at a clamped position, before the first is looked at)
    unsigned key[8], rv[8], sv[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const unsigned off = wid * 512 + s * 64 + lane;
      const unsigned at = c0 + (off < n ? off : 0u);
      rv[s] = rsrc[at];
      sv[s] = ssrc[at];
    }
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const unsigned off = wid * 512 + s * 64 + lane;
      key[s] = (off < n) ? (((rv[s] >> log_p) << kTcIdxBits) | off) : 0xFFFFFFFFu;
      if (off < n) lb[off] = sv[s];
    }
    for (int q = 0; q < passes; ++q) ta_radix_pass(key, kTcIdxBits + 8 * q, wcnt, dstart, wtot, buf);
    // (ta_radix_pass ends on a barrier: buf holds the keys in sorted order, fillers behind the n real ones)
    const unsigned n_chunks = (n + kTcChunk - 1) / kTcChunk;
    for (unsigned c = group; c < n_chunks; c += NG) {
      const unsigned* e = buf + c * kTcChunk;
      unsigned cur = (e[0] == 0xFFFFFFFFu) ? kTcNoRow : (((e[0] >> kTcIdxBits) << log_p) | part);
      F acc, wcur;
      acc.zero();
      wcur.zero();
      float cnt = 0.f;
      int nrun = 0;
      bool have_w = false;
      constexpr int U = 8;
      typedef float v4f __attribute__((ext_vector_type(4)));
      const int ec = (lane_g * 4 < D) ? lane_g * 4 : 0;             // (a lane beyond the row reads its head: never stored)
      for (int i0 = 0; i0 < kTcChunk; i0 += U) {
        unsigned row[U];
        float gg[U];
        v4f sr[U], wr[U];
        // every load of the batch is requested before the first use: plain loads at clamped addresses, no branch and no
        // arithmetic on a loaded value in between (the compiler waits for a load inside the block that issues it -- the
        // first version of this loop ran load / wait / load / wait, one round trip per pair: 214 us per launch)
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const unsigned k = e[i0 + u];
          const bool live = k != 0xFFFFFFFFu;
          row[u] = live ? (((k >> kTcIdxBits) << log_p) | part) : kTcNoRow;
          const unsigned b = lb[live ? (k & kTcIdxMask) : 0u];
          gg[u] = g[b];
          sr[u] = *reinterpret_cast<const v4f*>(ssum + static_cast<size_t>(b) * D + ec);
          wr[u] = __builtin_nontemporal_load(
              reinterpret_cast<const v4f*>(tb.table + static_cast<size_t>(live ? row[u] : 0u) * tb.stride + ec));
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (row[u] != cur) {                                   // the run of `cur` ends in front of this pair
            if (nrun == 0) {
#pragma unroll
              for (int q = 0; q < 4; ++q) esum[2 * c][lane_g * 4 + q] = acc.a[q];
              if (lane_g == 0) { ecnt[2 * c] = cnt; erow[2 * c] = cur; }
            } else if (cur != kTcNoRow) {
              tc_finish_row<G>(tb, cur, acc, cnt, wcur, D, lane_g, rmw);
            }
            ++nrun;
            acc.zero();
            cnt = 0.f;
            cur = row[u];
            have_w = false;
          }
          if (!have_w) {
#pragma unroll
            for (int q = 0; q < 4; ++q) wcur.a[q] = wr[u][q];
            have_w = true;
          }
          const float gu = (row[u] != kTcNoRow) ? gg[u] : 0.f;
#pragma unroll
          for (int q = 0; q < 4; ++q) acc.a[q] += gu * sr[u][q];
          cnt += gu;
        }
      }
      const int slot = (nrun == 0) ? 2 * c : 2 * c + 1;          // the run that is open at the end of the chunk
#pragma unroll
      for (int q = 0; q < 4; ++q) esum[slot][lane_g * 4 + q] = acc.a[q];
      if (lane_g == 0) {
        ecnt[slot] = cnt;
        erow[slot] = cur;
        if (nrun == 0) erow[2 * c + 1] = kTcNone;
      }
    }
    __syncthreads();
    // ---- the boundary runs: a stretch of equal rows is closed by its first element, in list order ----
    const unsigned ne = 2 * n_chunks;
    for (unsigned i = group; i < ne; i += NG) {
      const unsigned r = erow[i];
      if (r == kTcNone || r == kTcNoRow) continue;
      int p = static_cast<int>(i) - 1;
      if (p >= 0 && erow[p] == kTcNone) --p;                     // (element 2c always exists: at most one gap)
      if (p >= 0 && erow[p] == r) continue;                      // the stretch of this row started earlier
      F acc, w;
      acc.zero();
      w.zero();
      w.add_from_nt(tb.table + static_cast<size_t>(r) * tb.stride, D, lane_g);
#pragma unroll
      for (int q = 0; q < 4; ++q) acc.a[q] = esum[i][lane_g * 4 + q];
      float cnt = ecnt[i];
      for (unsigned j = i + 1; j < ne; ++j) {
        const unsigned rj = erow[j];
        if (rj == kTcNone) continue;
        if (rj != r) break;
#pragma unroll
        for (int q = 0; q < 4; ++q) acc.a[q] += esum[j][lane_g * 4 + q];
        cnt += ecnt[j];
      }
      tc_finish_row<G>(tb, r, acc, cnt, w, D, lane_g, rmw);
    }
    // The read-modify-writes of a later fill see the rows this fill stored: both come from this workgroup -- one CU, one
    // L2 -- so the barrier (which waits for the workgroup's outstanding stores) orders them, and the later loads go around
    // the L1 (tc_load_coherent).  NOT __threadfence(): a device-scope release makes every workgroup write its XCD's L2
    // back -- 768 workgroups doing that were 185 of the first two forms' 200 us per launch (profiles/r04/INDEX.md).
    __syncthreads();
  }
}

// ---- clear the rows the previous backward wrote (persistent gradient buffers): the bucket arrays still name them -------
template <int G>
__global__ __launch_bounds__(256) void tc_rezero_kernel(const TcPack P, const long long B, const int D, const int log_p,
                                                        const unsigned* __restrict__ start, const unsigned* __restrict__ count,
                                                        const unsigned* __restrict__ rows) {
  constexpr int NG = 256 / G;
  const int t = blockIdx.y;
  const TcTable tb = P.t[t];
  const unsigned np = 1u << log_p;
  const unsigned total = start[static_cast<size_t>(t) * np + np - 1] + count[static_cast<size_t>(t) * np + np - 1];
  const int lane_g = threadIdx.x % G, group = threadIdx.x / G;
  const unsigned* src = rows + static_cast<size_t>(t) * B;
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  const int e = lane_g * 4;
  for (unsigned i = blockIdx.x * NG + group; i < total; i += gridDim.x * NG) {
    const unsigned r = src[i];
    if (e < D) *reinterpret_cast<float4*>(tb.grad + static_cast<size_t>(r) * D + e) = z;
    if (lane_g == 0 && tb.grad2 != nullptr) tb.grad2[r] = 0.f;
  }
}

// ---- launches ---------------------------------------------------------------------------------------------------------
static inline int tc_launch_partition(const TcPlan& t, int64_t B, const int* cid, char* region, int clear_prev,
                                      hipStream_t s) {
  if (t.n_tab == 0) return RBX_OK;
  unsigned* hist = reinterpret_cast<unsigned*>(region + t.off_hist);
  hipLaunchKernelGGL(tc_count_kernel, dim3(t.NB, t.n_tab), dim3(256), 0, s, t.tab, static_cast<long long>(B), t.log_p, t.NB,
                     cid, hist);
  hipLaunchKernelGGL(tc_scan_kernel, dim3(t.n_tab), dim3(256), 0, s, t.log_p, t.NB, hist,
                     reinterpret_cast<unsigned*>(region + t.off_start), reinterpret_cast<unsigned*>(region + t.off_count),
                     reinterpret_cast<unsigned*>(region + t.off_prev), clear_prev);
  hipLaunchKernelGGL(tc_scatter_kernel, dim3(t.NB, t.n_tab), dim3(256), 0, s, t.tab, static_cast<long long>(B), t.log_p,
                     t.NB, cid, hist, reinterpret_cast<const unsigned*>(region + t.off_start),
                     reinterpret_cast<const unsigned*>(region + t.off_count),
                     reinterpret_cast<const unsigned*>(region + t.off_prev),
                     reinterpret_cast<unsigned*>(region + t.off_rows), reinterpret_cast<unsigned*>(region + t.off_smp),
                     clear_prev, t.D);
  return check_launch("tier-C partition kernels");
}

template <int G>
static int tc_launch_bwd(const TcPlan& t, int64_t B, const float* g, const float* ssum, int accumulate, char* region,
                         hipStream_t s) {
  hipLaunchKernelGGL(tc_reduce_kernel<G>, dim3(t.n_parts), dim3(256), 0, s, t.tab, t.n_tab, static_cast<long long>(B), t.D,
                     t.log_p, g, ssum, accumulate, reinterpret_cast<const unsigned*>(region + t.off_start),
                     reinterpret_cast<const unsigned*>(region + t.off_count),
                     reinterpret_cast<const unsigned*>(region + t.off_rows),
                     reinterpret_cast<const unsigned*>(region + t.off_smp));
  return check_launch("tc_reduce_kernel");
}

static inline int tc_dispatch_bwd(const TcPlan& t, int64_t B, const float* g, const float* ssum, int accumulate,
                                  char* region, hipStream_t s) {
  if (t.n_tab == 0) return RBX_OK;
  switch (pow2_ceil((t.D + 3) / 4)) {
    case 1: return tc_launch_bwd<1>(t, B, g, ssum, accumulate, region, s);
    case 2: return tc_launch_bwd<2>(t, B, g, ssum, accumulate, region, s);
    case 4: return tc_launch_bwd<4>(t, B, g, ssum, accumulate, region, s);
    case 8: return tc_launch_bwd<8>(t, B, g, ssum, accumulate, region, s);
    default: return tc_launch_bwd<16>(t, B, g, ssum, accumulate, region, s);
  }
}

template <int G>
static int tc_launch_rezero(const TcPlan& t, int64_t B, char* region, hipStream_t s) {
  constexpr int NG = 256 / G;
  unsigned blocks = static_cast<unsigned>((B + NG * 4 - 1) / (NG * 4));
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(tc_rezero_kernel<G>, dim3(blocks, t.n_tab), dim3(256), 0, s, t.tab, static_cast<long long>(B), t.D,
                     t.log_p, reinterpret_cast<const unsigned*>(region + t.off_start),
                     reinterpret_cast<const unsigned*>(region + t.off_count),
                     reinterpret_cast<const unsigned*>(region + t.off_rows));
  return check_launch("tc_rezero_kernel");
}

static inline int tc_dispatch_rezero(const TcPlan& t, int64_t B, char* region, hipStream_t s) {
  if (t.n_tab == 0) return RBX_OK;
  switch (pow2_ceil((t.D + 3) / 4)) {
    case 1: return tc_launch_rezero<1>(t, B, region, s);
    case 2: return tc_launch_rezero<2>(t, B, region, s);
    case 4: return tc_launch_rezero<4>(t, B, region, s);
    case 8: return tc_launch_rezero<8>(t, B, region, s);
    default: return tc_launch_rezero<16>(t, B, region, s);
  }
}

}  // namespace rbx
