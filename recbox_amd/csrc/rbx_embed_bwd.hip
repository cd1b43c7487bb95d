// rbx_embed_bwd.hip -- K3: embedding backward as a deterministic, sorted,
// segmented scatter-add (gfx950).
//
// Reference behaviour replaced: autograd's embedding_dense_backward behind every
// nn.Embedding of core/pytorch/layers/embedding.py:61-75, ranking/pytorch/layers/
// embeddings/feature_embedding.py:89-103 and third_party/rechub/basic/features.py:
// 43-46,75-78, together with the backward of the pooling / stack ops around it
// (SURVEY.md a-6): dW[i,:] = sum_{lookups of row i} w * dY[b,:], dW[padding_idx]=0,
// w = 1/(count+eps) under mean pooling, shared tables accumulate from every
// aliasing feature, result is a DENSE [V,D] gradient.
//
// Why a sort.  Criteo-shaped batches put 65 536 updates on tables with 3..30 rows;
// float atomics would serialise on a handful of L2 lines and make the result
// order dependent.  Instead every lookup becomes a (global row, lookup id) pair,
// the pairs are LSD radix sorted (8-bit digits, stable), and each run of equal
// rows is summed by ONE lane group in a fixed order and written ONCE:
//   build_keys -> [radix_hist -> radix_scan -> radix_scatter] x passes
//   -> segment_reduce (interior runs) -> segment_fixup (runs crossing chunks)
// No float atomics anywhere: run-to-run bit-identical gradients.
// All phases stream their inputs once; the random traffic is the dY row gather
// (L2/MALL resident for [B,F,D] <= 256 MiB) and one RMW per touched row.
#include <stdlib.h>
#include "rbx_segreduce.h"

namespace rbx {

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// BwdPlan::chained on (the default; RBX_SORT_CHAINED=0 or rbx_sort_chained(0): histogram + scan + scatter launches per pass)
static int g_sort_chained = -1;
static bool sort_chained_on() {
  if (g_sort_chained < 0) {
    const char* e = getenv("RBX_SORT_CHAINED");
    g_sort_chained = (e != nullptr && e[0] == '0') ? 0 : 1;
  }
  return g_sort_chained != 0;
}

int make_plan(const rbx_field_t* fields, int n, int64_t B, const float* dout, int64_t stride_b, BwdPlan* p,
              int extra_dim) {
  if (fields == nullptr || n <= 0 || n > RBX_MAX_FIELDS) return fail(RBX_ERR_INVALID, "bad field array");
  FieldPack tmp;
  int rc = pack_fields(fields, n, B, true, &tmp);
  if (rc != RBX_OK) return rc;
  unsigned long long lookups = 0, rows = 0;
  const float* seen_table[RBX_MAX_FIELDS];
  unsigned seen_base[RBX_MAX_FIELDS];
  int seen_dim[RBX_MAX_FIELDS];
  long long seen_vocab[RBX_MAX_FIELDS];
  int n_seen = 0;
  int field_table[RBX_MAX_FIELDS];         // table index of every categorical field, in KeyPack order
  unsigned field_lookups[RBX_MAX_FIELDS];
  p->vec = (stride_b % 4 == 0) && ((reinterpret_cast<uintptr_t>(dout) & 15) == 0);
  for (int i = 0; i < n; ++i) {
    const rbx_field_t& f = fields[i];
    if (f.kind == RBX_FIELD_DENSE || f.grad == nullptr) continue;   // no parameters / frozen
    if (f.dim > p->max_dim) p->max_dim = f.dim;
    if (f.dim % 4 != 0 || f.out_off % 4 != 0 || (reinterpret_cast<uintptr_t>(f.grad) & 15) != 0) p->vec = false;
    if (f.kind == RBX_FIELD_NUMERIC) {
      NumField& nf = p->num.f[p->n_num++];
      nf.ids = f.ids;
      nf.grad = f.grad;
      nf.stride_b = f.ids_stride_b;
      nf.out_off = static_cast<int>(f.out_off);
      nf.dim = static_cast<short>(f.dim);
      nf.dtype = static_cast<unsigned char>(f.ids_dtype);
      continue;
    }
    const unsigned long long n_lk = static_cast<unsigned long long>(B) * f.seq_len;
    if (n_lk > kLocalMask) return fail(RBX_ERR_UNSUPPORTED, "field %d: B*seq_len=%llu exceeds 2^26 per call", i, n_lk);
    int hit = -1;
    for (int s = 0; s < n_seen; ++s)
      if (seen_table[s] == f.table) hit = s;
    if (hit >= 0 && (seen_dim[hit] != f.dim || seen_vocab[hit] != f.vocab))
      return fail(RBX_ERR_INVALID, "field %d shares a table with different vocab/dim", i);
    if (hit < 0) {
      hit = n_seen++;
      seen_table[hit] = f.table;
      seen_base[hit] = static_cast<unsigned>(rows);
      seen_dim[hit] = f.dim;
      seen_vocab[hit] = f.vocab;
      rows += static_cast<unsigned long long>(f.vocab);
    }
    const int c = p->n_cat++;
    field_table[c] = hit;
    field_lookups[c] = static_cast<unsigned>(n_lk);
    KeyField& kf = p->keys.f[c];
    kf.ids = f.ids;
    kf.stride_b = f.ids_stride_b;
    kf.stride_l = static_cast<int>(f.ids_stride_l);
    kf.vocab = static_cast<int>(f.vocab);
    kf.mask_id = tmp.f[i].mask_id;
    kf.pad_id = tmp.f[i].pad_id;
    kf.row_base = seen_base[hit];
    kf.lk_off = static_cast<unsigned>(lookups);
    kf.seq_len = static_cast<short>(f.seq_len);
    kf.dtype = static_cast<unsigned char>(f.ids_dtype);
    kf.pool = static_cast<unsigned char>(f.pool);
    kf.reserved = 0;
    RedField& rf = p->red.f[c];
    rf.grad = f.grad;
    rf.grad2 = nullptr;
    rf.table = f.table;
    rf.row_base = seen_base[hit];
    rf.out_off = static_cast<int>(f.out_off);
    rf.dim = static_cast<short>(f.dim);
    rf.seq_len = static_cast<short>(f.seq_len);
    rf.pool = static_cast<unsigned char>(f.pool);
    rf.slot = static_cast<unsigned char>(i);
    rf.reserved = 0;
    rf.table_stride = f.dim;
    rf.reserved2 = 0;
    lookups += n_lk;
  }
  if (lookups >= (1ull << 31) || rows >= (1ull << 31))
    return fail(RBX_ERR_UNSUPPORTED, "too many lookups/rows for one call (%llu / %llu)", lookups, rows);
  p->n_lookups = static_cast<unsigned>(lookups);
  p->total_rows = static_cast<unsigned>(rows);
  // segments: maximal runs of consecutive fields closed under table sharing.  Tables are numbered in first-seen order,
  // so the tables first seen inside a segment own one contiguous row range.
  {
    int last_field_of_table[RBX_MAX_FIELDS];
    for (int c = 0; c < p->n_cat; ++c) last_field_of_table[field_table[c]] = c;
    SegPack& S = p->segs;
    S.n = 0;
    unsigned tiles = 0, lk = 0;
    unsigned long long max_rows = 1;
    unsigned long long rows_of_seg[RBX_MAX_FIELDS];
    int c = 0;
    while (c < p->n_cat) {
      int end = last_field_of_table[field_table[c]];
      unsigned long long seg_rows = 0;
      unsigned seg_lk = 0;
      const unsigned row0 = p->keys.f[c].row_base;
      int t_lo = field_table[c], t_hi = field_table[c];
      for (int k = c; k <= end; ++k) {
        if (last_field_of_table[field_table[k]] > end) end = last_field_of_table[field_table[k]];
        if (field_table[k] > t_hi) t_hi = field_table[k];
        if (field_table[k] < t_lo) t_lo = field_table[k];
        seg_lk += field_lookups[k];
      }
      for (int t = t_lo; t <= t_hi; ++t) seg_rows += static_cast<unsigned long long>(seen_vocab[t]);
      if (seg_lk > 0) {                                  // (an empty batch has no tiles)
        S.tile0[S.n] = tiles;
        S.lk0[S.n] = lk;
        S.row0[S.n] = row0;
        rows_of_seg[S.n] = seg_rows;
        ++S.n;
        tiles += (seg_lk + kSortTile - 1) / kSortTile;
        lk += seg_lk;
        if (seg_rows > max_rows) max_rows = seg_rows;
      }
      c = end + 1;
    }
    S.tile0[S.n] = tiles;
    S.lk0[S.n] = lk;
    p->n_tiles = tiles;
    unsigned max_nt = 0;
    for (int g = 0; g < S.n; ++g)
      if (S.tile0[g + 1] - S.tile0[g] > max_nt) max_nt = S.tile0[g + 1] - S.tile0[g];
    p->chained = max_nt <= static_cast<unsigned>(kChainTiles);
    int bits = 1;
    while ((1ull << bits) <= max_rows) ++bits;     // 2^bits > rows of the largest segment: the all-ones key is free for masked lookups
    p->passes = (bits + kMaxRadixBits - 1) / kMaxRadixBits;
    const int per = (bits + p->passes - 1) / p->passes;
    p->radix_bits = per <= 8 ? 8 : (per <= 10 ? 10 : 11);
    for (int g = 0; g < S.n; ++g) {
      int sb = 1;
      while ((1ull << sb) <= rows_of_seg[g]) ++sb;
      const int need = (sb + p->radix_bits - 1) / p->radix_bits;
      S.first_pass[g] = static_cast<unsigned char>(p->passes - need);
    }
  }
  // Pairs per lane group of the reduce.  40 is the measured optimum when the batch yields ~100 000 chunks (1.7 M pairs at the
  // Criteo shape: 666 workgroups, all resident).  With fewer pairs -- the 0.52 M of the eight large tables once the small ones
  // take the sort-free path -- 40 left 205 workgroups on 256 CUs, each walking 5 dependent rounds of gathers: 69-80 us;
  // 16 pairs per group: 49 us (profiles/r03).  Rule: at least 32 768 chunks, between 16 and 40 pairs, a multiple of 8.
  {
    int c = static_cast<int>(p->n_lookups / 32768u) / 8 * 8;
    if (c < 16) c = 16;
    if (c > kChunk) c = kChunk;
    p->chunk = c;
  }
  p->n_chunks = (p->n_lookups + p->chunk - 1) / p->chunk;
  p->num_blocks = static_cast<unsigned>((B + kNumSamples - 1) / kNumSamples);
  size_t o = 0;
  const size_t nl = align_up(static_cast<size_t>(p->n_lookups) * 4, 256);
  p->off_keys[0] = o; o += nl;
  p->off_keys[1] = o; o += nl;
  p->off_vals[0] = o; o += nl;
  p->off_vals[1] = o; o += nl;
  const size_t radix = static_cast<size_t>(1) << p->radix_bits;
  p->chained = sort_chained_on() && p->chained && p->passes <= kChainPasses && p->radix_bits == 8;
  p->off_hist = o;
  o += p->chained ? align_up((2 * static_cast<size_t>(p->n_tiles) * radix + p->n_tiles + static_cast<size_t>(p->segs.n) * radix) *
                             p->passes * 4, 256)
                  : align_up(static_cast<size_t>(p->n_tiles) * radix * 4 + 4, 256);
  p->off_ssum = o; o += align_up((static_cast<size_t>(p->n_tiles) * radix / 4096 + 2) * 4, 256);
  p->sum_stride = p->max_dim + extra_dim;
  p->off_head = o; o += align_up(static_cast<size_t>(p->n_chunks) * p->sum_stride * 4, 256);
  p->off_tail = o; o += align_up(static_cast<size_t>(p->n_chunks) * p->sum_stride * 4, 256);
  p->off_flags = o; o += align_up(static_cast<size_t>(p->n_chunks) * 4, 256);
  // 3 counters, short list, long list, arrival counters of long chains that several workgroups share
  p->off_fin = o; o += align_up((2 * static_cast<size_t>(p->n_chunks) + 3 + kLongSlots) * 4, 256);
  p->off_long = o; o += align_up(static_cast<size_t>(kLongSlots) * p->sum_stride * 4, 256);      // ... and their partials
  p->off_num = o; o += align_up(static_cast<size_t>(p->n_num) * p->num_blocks * p->max_dim * 4, 256);
  p->bytes = o + 256;
  return RBX_OK;
}

// ---- build_keys ----------------------------------------------------------------
// digit `shift / RB` of a key inside its segment; masked lookups (key == sentinel) are all ones: last in every pass
template <int RB>
__device__ __forceinline__ unsigned seg_digit(unsigned key, unsigned sentinel, unsigned row0, int shift) {
  const unsigned local = (key == sentinel) ? 0xFFFFFFFFu : key - row0;
  return (local >> shift) & ((1u << RB) - 1u);
}

template <int RB>
__global__ __launch_bounds__(kSortThreads) void build_keys_kernel(const KeyPack P, const int n_cat, const SegPack S,
                                                         const unsigned sentinel, unsigned* __restrict__ keys0,
                                                         unsigned* __restrict__ vals0, unsigned* __restrict__ keys1,
                                                         unsigned* __restrict__ vals1, int* __restrict__ status,
                                                         unsigned* __restrict__ fin, unsigned* __restrict__ hist,
                                                         const int chain_passes, const unsigned n_tiles) {
  constexpr int R = 1 << RB;
  constexpr int kCP = (RB == 8) ? kChainPasses : 1;      // (the chained sort runs 8-bit digits only)
  if (blockIdx.x == 0 && threadIdx.x == 0) {          // fix-up work-list length and arrival counter (rbx_segreduce.h)
    fin[0] = 0;
    fin[1] = 0;
    fin[2] = 0;
  }
  __shared__ KeyField sf[RBX_MAX_FIELDS];
  {
    const int words = n_cat * static_cast<int>(sizeof(KeyField) / 4);
    const int* src = reinterpret_cast<const int*>(&P);
    int* dst = reinterpret_cast<int*>(sf);
    for (int i = threadIdx.x; i < words; i += blockDim.x) dst[i] = src[i];
  }
  // one workgroup per sort tile, so that the tile's histogram of the FIRST radix digit falls out of the same pass
  // (the keys are in registers anyway): the first radix_hist_kernel launch of the sort is not needed
  __shared__ unsigned cnt[kCP][R];
  for (int d = threadIdx.x; d < kCP * R; d += kSortThreads) (&cnt[0][0])[d] = 0;
  __syncthreads();
  int seg;
  unsigned tile0, tile_n;
  seg_of_tile(S, blockIdx.x, &seg, &tile0, &tile_n);
  const unsigned row0 = S.row0[seg];
  // the pairs go where the segment's first pass reads them; only a segment that starts in pass 0 needs its first
  // digit's histogram from here (the others get it from radix_hist_kernel when their turn comes)
  const int fp = S.first_pass[seg];
  unsigned* __restrict__ keys = (fp & 1) ? keys1 : keys0;
  unsigned* __restrict__ vals = (fp & 1) ? vals1 : vals0;
  // (a two-sweep variant -- all raw id loads first, decoding afterwards -- was measured slower: 19.3 vs 16.3 us)
#pragma unroll
  for (int it = 0; it < kSortItems; ++it) {
    const unsigned off = it * kSortThreads + threadIdx.x;
    if (off >= tile_n) break;
    const unsigned j = tile0 + off;
    int lo = 0, hi = n_cat - 1;                      // last field with lk_off <= j
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (sf[mid].lk_off <= j) lo = mid; else hi = mid - 1;
    }
    const KeyField& fd = sf[lo];
    const unsigned local = j - fd.lk_off;
    const unsigned L = static_cast<unsigned>(fd.seq_len);
    const unsigned b = local / L;
    const unsigned l = local - b * L;
    const long long id = load_id(fd.ids, static_cast<long long>(b) * fd.stride_b + static_cast<long long>(l) * fd.stride_l,
                                 fd.dtype);
    unsigned key = sentinel;
    if (id < 0 || id >= fd.vocab) {
      if (status != nullptr) atomicOr(status, 1);
    } else {
      const bool id_pool = fd.pool == RBX_POOL_MEAN_ID || fd.pool == RBX_POOL_SUM_ID;
      if (id != fd.pad_id && !(id_pool && id == fd.mask_id)) key = fd.row_base + static_cast<unsigned>(id);
    }
    keys[j] = key;
    vals[j] = (static_cast<unsigned>(lo) << kLocalBits) | local;
    if (chain_passes > 0) {
      // BwdPlan::chained: this tile's count of every digit the segment will be sorted by
#pragma unroll
      for (int k = 0; k < kCP; ++k)
        if (fp + k < chain_passes) atomicAdd(&cnt[k][seg_digit<RB>(key, sentinel, row0, k * RB)], 1u);
    } else if (fp == 0) {
      atomicAdd(&cnt[0][seg_digit<RB>(key, sentinel, row0, 0)], 1u);
    }
  }
  if (chain_passes > 0) {
    // [pass][tile][digit] counts, then the same again (what the scatter workgroups publish), then [pass][tile] flags
    __syncthreads();
    const size_t plane = static_cast<size_t>(n_tiles) * R;
    unsigned* flags = hist + 2 * plane * chain_passes;
    for (int k = 0; fp + k < chain_passes; ++k) {
      unsigned* h = hist + plane * (fp + k) + static_cast<size_t>(blockIdx.x) * R;
      for (int d = threadIdx.x; d < R; d += kSortThreads) h[d] = cnt[k < kCP ? k : 0][d];
      if (threadIdx.x == 0) flags[static_cast<size_t>(fp + k) * n_tiles + blockIdx.x] = 0u;
    }
    return;
  }
  if (fp != 0) return;
  __syncthreads();
  // histogram layout: segment, then digit, then tile of the segment -- one flat exclusive scan then yields, for every
  // (digit, tile), the global position of its first pair
  const unsigned t_in = blockIdx.x - S.tile0[seg], nt = S.tile0[seg + 1] - S.tile0[seg];
  unsigned* h = hist + static_cast<size_t>(S.tile0[seg]) * R + t_in;
  for (int d = threadIdx.x; d < R; d += kSortThreads) h[static_cast<size_t>(d) * nt] = cnt[0][d];
}

// ---- radix sort: per-tile digit histogram ----------------------------------------
template <int RB>
__global__ __launch_bounds__(kSortThreads) void radix_hist_kernel(const unsigned* __restrict__ keys, const SegPack S,
                                                                  const unsigned sentinel, const int pass,
                                                                  unsigned* __restrict__ hist) {
  constexpr int R = 1 << RB;
  __shared__ unsigned cnt[R];
  int seg;
  unsigned tile0, tile_n;
  seg_of_tile(S, blockIdx.x, &seg, &tile0, &tile_n);
  if (pass < S.first_pass[seg]) return;               // the segment joins the sort in a later pass
  const int shift = (pass - S.first_pass[seg]) * RB;
  for (int d = threadIdx.x; d < R; d += kSortThreads) cnt[d] = 0;
  __syncthreads();
  const unsigned row0 = S.row0[seg];
  // (loading the 8 keys of a thread before the first LDS atomic was measured slower: 11.3 vs 9.1 us)
#pragma unroll
  for (int i = 0; i < kSortItems; ++i) {
    const unsigned off = i * kSortThreads + threadIdx.x;
    if (off < tile_n) atomicAdd(&cnt[seg_digit<RB>(keys[tile0 + off], sentinel, row0, shift)], 1u);
  }
  __syncthreads();
  const unsigned t_in = blockIdx.x - S.tile0[seg], nt = S.tile0[seg + 1] - S.tile0[seg];
  unsigned* h = hist + static_cast<size_t>(S.tile0[seg]) * R + t_in;
  for (int d = threadIdx.x; d < R; d += kSortThreads) h[static_cast<size_t>(d) * nt] = cnt[d];
}

// ---- exclusive scan of hist[256 * n_tiles], two levels -------------------------------
// level 1: every workgroup scans its own 4096-entry slice in place and emits the slice
// total; level 2: one workgroup scans the slice totals; the scatter kernel adds the
// slice prefix when it picks up its 256 (digit, tile) offsets.
constexpr int kScanSlice = 4096;
constexpr int kMaxFusedSlices = 4 * kSortThreads;      // slice totals one scatter workgroup can scan by itself

__device__ __forceinline__ unsigned block_scan_1024x4(unsigned (&v)[4], unsigned* wave_tot, unsigned* total) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const unsigned mine = v[0] + v[1] + v[2] + v[3];
  unsigned inc = mine;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  if (lane == 63) wave_tot[wid] = inc;
  __syncthreads();
  unsigned wbase = 0, all = 0;
#pragma unroll
  for (int w = 0; w < 16; ++w) {
    const unsigned t = wave_tot[w];
    if (w < wid) wbase += t;
    all += t;
  }
  *total = all;
  return wbase + inc - mine;      // exclusive prefix of this thread's first element
}

__global__ __launch_bounds__(1024) void radix_scan_local_kernel(unsigned* __restrict__ hist, const unsigned len,
                                                                unsigned* __restrict__ slice_sum) {
  __shared__ unsigned wave_tot[16];
  const unsigned i0 = blockIdx.x * kScanSlice + threadIdx.x * 4;
  unsigned v[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] = (i0 + k < len) ? hist[i0 + k] : 0u;
  unsigned total;
  unsigned run = block_scan_1024x4(v, wave_tot, &total);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (i0 + k < len) hist[i0 + k] = run;
    run += v[k];
  }
  if (threadIdx.x == 0) slice_sum[blockIdx.x] = total;
}

__global__ __launch_bounds__(1024) void radix_scan_sums_kernel(unsigned* __restrict__ sums, const unsigned len) {
  __shared__ unsigned wave_tot[16];
  unsigned carry = 0;
  for (unsigned base = 0; base < len; base += kScanSlice) {
    const unsigned i0 = base + threadIdx.x * 4;
    unsigned v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = (i0 + k < len) ? sums[i0 + k] : 0u;
    unsigned total;
    unsigned run = carry + block_scan_1024x4(v, wave_tot, &total);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (i0 + k < len) sums[i0 + k] = run;
      run += v[k];
    }
    carry += total;
    __syncthreads();
  }
}

// ---- stable scatter of one tile -----------------------------------------------------
// Wave w owns the contiguous quarter [w*512, (w+1)*512) of the tile and walks it in
// 64-item steps, so tile order == (wave, step, lane) and ranks respect it.
template <int RB, bool CHAIN>
__global__ __launch_bounds__(kSortThreads) void radix_scatter_kernel(const unsigned* __restrict__ keys_in,
                                                                     const unsigned* __restrict__ vals_in,
                                                                     unsigned* __restrict__ keys_out,
                                                                     unsigned* __restrict__ vals_out, const SegPack S,
                                                                     const unsigned sentinel, const int pass,
                                                                     const unsigned* __restrict__ hist,
                                                                     const unsigned* __restrict__ slice_sum,
                                                                     const unsigned n_slices, const bool raw_sums,
                                                                     unsigned* __restrict__ chain, const int passes,
                                                                     const unsigned n_tiles) {
  constexpr int R = 1 << RB;
  constexpr int DPT = R / kSortThreads;              // digits per thread in the per-digit steps
  constexpr int kWaves = kSortThreads / 64;
  constexpr int kPerWave = kSortTile / kWaves;       // 512
  constexpr int kSteps = kPerWave / 64;              // 8
  __shared__ unsigned wcnt[kWaves][R];               // per-wave digit counts, then running offsets
  __shared__ unsigned dstart[R];                     // tile-local start of each digit
  __shared__ unsigned gbase[R];                      // global start of (digit, tile)
  __shared__ unsigned skey[kSortTile];
  __shared__ unsigned sval[kSortTile];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  int seg;
  unsigned tile0, tile_n;
  seg_of_tile(S, blockIdx.x, &seg, &tile0, &tile_n);
  if (pass < S.first_pass[seg]) return;               // the segment joins the sort in a later pass
  const int shift = (pass - S.first_pass[seg]) * RB;
  const unsigned row0 = S.row0[seg];
  const unsigned t_in = blockIdx.x - S.tile0[seg], nt = S.tile0[seg + 1] - S.tile0[seg];
  // Positions are taken RELATIVE to the segment's first histogram entry: the flat scan runs over every segment's
  // entries, and those of a segment that is not in this pass are stale -- differences inside one segment do not see
  // them (unsigned wrap-around included).
  const size_t hfirst = static_cast<size_t>(S.tile0[seg]) * R;
  const size_t hbase = hfirst + t_in;
  const unsigned seg_lk0 = S.lk0[seg];
  for (int i = threadIdx.x; i < kWaves * R; i += kSortThreads) (&wcnt[0][0])[i] = 0;
  // BwdPlan::chained.  Start of this tile's run of digit d = pairs of the segment with a smaller digit (from build_keys'
  // per-tile counts of this pass's digit: a total does not depend on the order) + pairs with digit d in the tiles in front
  // of this one -- in the segment's first pass those are build_keys' counts too (the order is still the original one),
  // later they are what those tiles publish below.
  unsigned chain_excl[DPT], chain_before[DPT];
  const size_t plane = static_cast<size_t>(n_tiles) * R;
  const bool first_of_seg = pass == S.first_pass[seg];
  if constexpr (CHAIN) {
    __shared__ unsigned ctot[kWaves];
    const unsigned* T = chain + plane * pass + static_cast<size_t>(S.tile0[seg]) * R;
    // (after the flags: [pass][segment][digit] totals, written in the segment's first pass by its first tile for the passes
    //  that follow -- one row to read there instead of one per tile)
    unsigned* G = chain + (2 * plane + n_tiles) * passes;
    unsigned tot[DPT];
    unsigned mine = 0;
    if (first_of_seg) {
#pragma unroll
      for (int q = 0; q < DPT; ++q) {
        const int d = threadIdx.x * DPT + q;
        unsigned a = 0, b = 0;
#pragma unroll 8
        for (unsigned t = 0; t < nt; ++t) {
          const unsigned c = T[static_cast<size_t>(t) * R + d];
          a += c;
          b += (t < t_in) ? c : 0u;
        }
        tot[q] = a;
        chain_before[q] = b;
        mine += a;
      }
      if (t_in == 0) {
        for (int k = pass + 1; k < passes; ++k) {
          const unsigned* Tk = chain + plane * k + static_cast<size_t>(S.tile0[seg]) * R;
#pragma unroll
          for (int q = 0; q < DPT; ++q) {
            const int d = threadIdx.x * DPT + q;
            unsigned a = 0;
#pragma unroll 8
            for (unsigned t = 0; t < nt; ++t) a += Tk[static_cast<size_t>(t) * R + d];
            G[(static_cast<size_t>(k) * S.n + seg) * R + d] = a;
          }
        }
      }
    } else {
#pragma unroll
      for (int q = 0; q < DPT; ++q) {
        tot[q] = G[(static_cast<size_t>(pass) * S.n + seg) * R + threadIdx.x * DPT + q];
        chain_before[q] = 0;
        mine += tot[q];
      }
    }
    unsigned inc = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const unsigned u = __shfl_up(inc, o, 64);
      if (lane >= o) inc += u;
    }
    if (lane == 63) ctot[wid] = inc;
    __syncthreads();
    unsigned run = inc - mine;
    for (int w = 0; w < wid; ++w) run += ctot[w];
#pragma unroll
    for (int q = 0; q < DPT; ++q) {
      chain_excl[q] = run;
      run += tot[q];
    }
  } else if (raw_sums) {
    // slice_sum holds the slice TOTALS as radix_scan_local_kernel left them (at most kMaxFusedSlices of them): every
    // workgroup scans them itself (a few dozen values at the bench shape) instead of waiting for a separate
    // one-workgroup kernel per pass
    __shared__ unsigned spre[kMaxFusedSlices];
    __shared__ unsigned stot[kWaves];
    const unsigned i0 = threadIdx.x * 4;
    unsigned t[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) t[q] = (i0 + q < n_slices) ? slice_sum[i0 + q] : 0u;
    const unsigned mine = t[0] + t[1] + t[2] + t[3];
    unsigned inc = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const unsigned u = __shfl_up(inc, o, 64);
      if (lane >= o) inc += u;
    }
    if (lane == 63) stot[wid] = inc;
    __syncthreads();
    unsigned run = inc - mine;
    for (int w = 0; w < wid; ++w) run += stot[w];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      spre[i0 + q] = run;
      run += t[q];
    }
    __syncthreads();
    const unsigned first = hist[hfirst] + spre[hfirst / kScanSlice];
    for (int d = threadIdx.x; d < R; d += kSortThreads) {
      const size_t hi = hbase + static_cast<size_t>(d) * nt;
      gbase[d] = seg_lk0 + (hist[hi] + spre[hi / kScanSlice] - first);
    }
  } else {
    const unsigned first = hist[hfirst] + slice_sum[hfirst / kScanSlice];
    for (int d = threadIdx.x; d < R; d += kSortThreads) {
      const size_t hi = hbase + static_cast<size_t>(d) * nt;
      gbase[d] = seg_lk0 + (hist[hi] + slice_sum[hi / kScanSlice] - first);
    }
  }
  __syncthreads();

  unsigned k[kSteps], v[kSteps];
  unsigned rank[kSteps];                             // rank inside (wave, digit)
  const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
  for (int s = 0; s < kSteps; ++s) {                 // all 16 loads of the thread in flight before the first is used
    const unsigned off = wid * kPerWave + s * 64 + lane;
    const bool ok = off < tile_n;
    k[s] = ok ? __builtin_nontemporal_load(keys_in + tile0 + off) : 0xFFFFFFFFu;
    v[s] = ok ? __builtin_nontemporal_load(vals_in + tile0 + off) : 0u;
  }
#pragma unroll
  for (int s = 0; s < kSteps; ++s) {
    const unsigned off = wid * kPerWave + s * 64 + lane;
    const bool ok = off < tile_n;
    const unsigned d = ok ? seg_digit<RB>(k[s], sentinel, row0, shift) : 0u;
    unsigned long long peers = __ballot(ok);
#pragma unroll
    for (int b = 0; b < RB; ++b) {
      const unsigned long long m = __ballot((d >> b) & 1u);
      peers &= ((d >> b) & 1u) ? m : ~m;
    }
    if (!ok) peers = 0ull;
    const unsigned before = __popcll(peers & lt);
    // LDS ops of one wave issue in program order: every peer reads the running count
    // before the run leader (lowest peer lane) bumps it.  volatile: no caching across steps.
    volatile unsigned* wc = wcnt[wid];
    unsigned prev = 0;
    if (ok) prev = wc[d];
    rank[s] = prev + before;
    if (ok && before == 0) wc[d] = prev + __popcll(peers);
  }
  __syncthreads();
  // per digit: exclusive prefix over waves, then over digits (a thread owns DPT consecutive digits)
  {
    unsigned runs[DPT];
    unsigned mine = 0;
#pragma unroll
    for (int q = 0; q < DPT; ++q) {
      const int d = threadIdx.x * DPT + q;
      unsigned run = 0;
#pragma unroll
      for (int w = 0; w < kWaves; ++w) {
        const unsigned c = wcnt[w][d];
        wcnt[w][d] = run;
        run += c;
      }
      runs[q] = run;
      mine += run;
    }
    unsigned inc = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const unsigned t = __shfl_up(inc, o, 64);
      if (lane >= o) inc += t;
    }
    __shared__ unsigned wtot[kWaves];
    if (lane == 63) wtot[wid] = inc;
    __syncthreads();
    unsigned base = inc - mine;
    for (int w = 0; w < wid; ++w) base += wtot[w];
#pragma unroll
    for (int q = 0; q < DPT; ++q) {
      dstart[threadIdx.x * DPT + q] = base;
      base += runs[q];
    }
    if constexpr (CHAIN) {
      if (!first_of_seg) {
        // publish this tile's counts, then wait for the tiles in front of it in the segment (lower workgroup indices:
        // dispatched before this one, so they are running or done) and add theirs up
        unsigned* L = chain + plane * (passes + pass);
        unsigned* flags = chain + 2 * plane * passes + static_cast<size_t>(pass) * n_tiles;
        // (everything that crosses workgroups here is an agent-scope atomic access -- written through to / read from the
        //  memory side, where the eight L2s agree: a release / acquire pair would write back and invalidate a whole L2 per
        //  workgroup.  The counts are acknowledged -- vmcnt(0) -- before the barrier, the flag goes out after it.)
#pragma unroll
        for (int q = 0; q < DPT; ++q)
          __hip_atomic_store(L + static_cast<size_t>(blockIdx.x) * R + threadIdx.x * DPT + q, runs[q], __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(flags + blockIdx.x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (threadIdx.x < t_in) {
          const unsigned* f = flags + S.tile0[seg] + threadIdx.x;
          while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
        const unsigned* Ls = L + static_cast<size_t>(S.tile0[seg]) * R;
#pragma unroll
        for (int q = 0; q < DPT; ++q) {
          unsigned b = 0;
#pragma unroll 8
          for (unsigned t = 0; t < t_in; ++t)
            b += __hip_atomic_load(Ls + static_cast<size_t>(t) * R + threadIdx.x * DPT + q, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
          chain_before[q] = b;
        }
      }
#pragma unroll
      for (int q = 0; q < DPT; ++q) gbase[threadIdx.x * DPT + q] = seg_lk0 + chain_excl[q] + chain_before[q];
    }
  }
  __syncthreads();
#pragma unroll
  for (int s = 0; s < kSteps; ++s) {
    const unsigned off = wid * kPerWave + s * 64 + lane;
    if (off < tile_n) {
      const unsigned d = seg_digit<RB>(k[s], sentinel, row0, shift);
      const unsigned pos = dstart[d] + wcnt[wid][d] + rank[s];
      skey[pos] = k[s];
      sval[pos] = v[s];
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < kSortItems; ++i) {
    const unsigned pos = i * kSortThreads + threadIdx.x;
    if (pos < tile_n) {
      const unsigned key = skey[pos];
      const unsigned d = seg_digit<RB>(key, sentinel, row0, shift);
      const unsigned g = gbase[d] + (pos - dstart[d]);
      keys_out[g] = key;
      vals_out[g] = sval[pos];
    }
  }
}

// ---- generic policy: a lookup contributes w * dY[b, slot] (the kernels are in rbx_segreduce.h) ----
struct GenericPolicy {
  static constexpr bool kHasCount = false;
  struct Args {
    const float* dout;
    long long stride_b;
    const float* row_scale;
    long long B;
    int accumulate;        // 0: grads were pre-zeroed by the caller -> store; 1: read-modify-write
    const int* index;      // optional: sample b reads row index[b] of dout (rbx_embed_bwd_indexed)
  };
  template <class F>
  static __device__ __forceinline__ void contribute(const Args& a, const RedField& fd, unsigned local, int lane_g,
                                                    F& frag, float& cnt) {
    const unsigned L = static_cast<unsigned>(fd.seq_len);
    const unsigned b = local / L;
    const unsigned l = local - b * L;
    const long long brow = (a.index != nullptr) ? static_cast<long long>(a.index[b]) : static_cast<long long>(b);
    const float* src = a.dout + brow * a.stride_b + fd.out_off +
                       (fd.pool == RBX_POOL_CONCAT ? static_cast<long long>(l) * fd.dim : 0ll);
    float w = 1.0f;
    if (fd.pool == RBX_POOL_MEAN_VALUE || fd.pool == RBX_POOL_MEAN_ID)
      w = a.row_scale[static_cast<long long>(fd.slot) * a.B + b];
    frag.fma_from(src, fd.dim, lane_g, w);
    (void)cnt;
  }
  template <class F>
  static __device__ __forceinline__ void prefetch(const Args& a, const RedField& fd, unsigned row, int lane_g, F& pre) {
    if (a.accumulate) pre.add_from(fd.grad + static_cast<size_t>(row) * fd.dim, fd.dim, lane_g);   // old grad (RMW)
  }
  // the two-phase form (segment_reduce_kernel): loads only, then frag *= weight(w)
  template <class F>
  static __device__ __forceinline__ void fetch(const Args& a, const RedField& fd, unsigned local, int lane_g, F& frag, float& w) {
    const unsigned L = static_cast<unsigned>(fd.seq_len);
    const unsigned b = local / L;
    const unsigned l = local - b * L;
    const long long brow = (a.index != nullptr) ? static_cast<long long>(a.index[b]) : static_cast<long long>(b);
    const float* src = a.dout + brow * a.stride_b + fd.out_off +
                       (fd.pool == RBX_POOL_CONCAT ? static_cast<long long>(l) * fd.dim : 0ll);
    w = 1.0f;
    if (fd.pool == RBX_POOL_MEAN_VALUE || fd.pool == RBX_POOL_MEAN_ID)
      w = a.row_scale[static_cast<long long>(fd.slot) * a.B + b];
    frag.load_from(src, fd.dim, lane_g);
  }
  static __device__ __forceinline__ float weight(const Args&, float w) { return w; }
  template <class F>
  static __device__ __forceinline__ void prefetch_raw(const Args& a, const RedField& fd, unsigned row, int lane_g, F& pre) {
    if (a.accumulate) pre.load_from(fd.grad + static_cast<size_t>(row) * fd.dim, fd.dim, lane_g);   // old grad (RMW)
  }
  template <class F>
  static __device__ __forceinline__ void flush(const Args&, const RedField& fd, unsigned row, const F& acc, float,
                                               const F& pre, int lane_g) {
    // every touched row is written by exactly one lane group per call; pre = old grad or zeros
    F out = acc;
    frag_add(out, pre);
    out.store_nt(fd.grad + static_cast<size_t>(row) * fd.dim, fd.dim, lane_g);
  }
};

// ---- numeric features: grad[d] += sum_b x_b * dY[b, off + d] -----------------------------
// grid (num_blocks, n_num); a workgroup reduces kNumSamples samples of one feature:
// 256 threads = 16 sample lanes x 16 dim lanes per step, LDS tree across sample lanes.
__global__ __launch_bounds__(256) void numeric_partial_kernel(const NumPack P, const long long B,
                                                              const float* __restrict__ dout, const long long stride_b,
                                                              float* __restrict__ partial, const int max_dim,
                                                              const unsigned num_blocks) {
  const NumField fd = P.f[blockIdx.y];
  const int dim = fd.dim;
  __shared__ float red[256];
  const long long b0 = static_cast<long long>(blockIdx.x) * kNumSamples;
  const long long b1 = (b0 + kNumSamples < B) ? b0 + kNumSamples : B;
  for (int dbase = 0; dbase < dim; dbase += 16) {
    const int d = dbase + (threadIdx.x & 15);
    float acc = 0.f;
    if (d < dim) {
      long long b = b0 + (threadIdx.x >> 4);
      for (; b + 48 < b1; b += 64) {                         // 4 samples in flight, added in ascending order
        float xv[4], gv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          xv[u] = load_value(fd.ids, (b + 16 * u) * fd.stride_b, fd.dtype);
          gv[u] = dout[(b + 16 * u) * stride_b + fd.out_off + d];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += xv[u] * gv[u];
      }
      for (; b < b1; b += 16) {
        const float x = load_value(fd.ids, b * fd.stride_b, fd.dtype);
        acc += x * dout[b * stride_b + fd.out_off + d];
      }
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x < 16) {
      float t = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) t += red[r * 16 + threadIdx.x];
      if (d < dim) partial[(static_cast<size_t>(blockIdx.y) * num_blocks + blockIdx.x) * max_dim + d] = t;
    }
    __syncthreads();
  }
}

// one wavefront per (feature, d): lanes stride over the block partials, fixed shuffle tree
__global__ __launch_bounds__(64) void numeric_final_kernel(const NumPack P, const float* __restrict__ partial,
                                                           const int max_dim, const unsigned num_blocks) {
  const NumField fd = P.f[blockIdx.x];
  const int d = blockIdx.y;
  if (d >= fd.dim) return;
  float t = 0.f;
  for (unsigned k = threadIdx.x; k < num_blocks; k += 64)
    t += partial[(static_cast<size_t>(blockIdx.x) * num_blocks + k) * max_dim + d];
  t = group_sum<64>(t);
  if (threadIdx.x == 0) fd.grad[d] += t;
}

}  // namespace rbx

extern "C" int rbx_sort_chained(int32_t enable) {
  const int was = rbx::sort_chained_on() ? 1 : 0;
  if (enable >= 0) rbx::g_sort_chained = enable ? 1 : 0;
  return was;
}

extern "C" size_t rbx_embed_bwd_workspace_size(const rbx_field_t* fields, int32_t n_fields, int64_t batch) {
  rbx::BwdPlan p;
  if (rbx::make_plan(fields, n_fields, batch, nullptr, 0, &p) != RBX_OK) return 0;
  return p.bytes;
}

namespace rbx {
template <int RB>
static int run_sort_rb(const BwdPlan& p, char* ws, int* d_status, hipStream_t s) {
  constexpr int R = 1 << RB;
  unsigned* keys[2] = {reinterpret_cast<unsigned*>(ws + p.off_keys[0]), reinterpret_cast<unsigned*>(ws + p.off_keys[1])};
  unsigned* vals[2] = {reinterpret_cast<unsigned*>(ws + p.off_vals[0]), reinterpret_cast<unsigned*>(ws + p.off_vals[1])};
  unsigned* hist = reinterpret_cast<unsigned*>(ws + p.off_hist);
  unsigned* ssum = reinterpret_cast<unsigned*>(ws + p.off_ssum);
  const bool chained = p.chained && RB == 8;
  hipLaunchKernelGGL(build_keys_kernel<RB>, dim3(p.n_tiles), dim3(kSortThreads), 0, s, p.keys, p.n_cat, p.segs, p.total_rows,
                     keys[0], vals[0], keys[1], vals[1], d_status, reinterpret_cast<unsigned*>(ws + p.off_fin), hist,
                     chained ? p.passes : 0, p.n_tiles);
  int rc = check_launch("build_keys_kernel");
  if (rc != RBX_OK) return rc;
  int cur = 0;
  for (int pass = 0; pass < p.passes; ++pass) {
    if (chained) {
      hipLaunchKernelGGL((radix_scatter_kernel<RB, true>), dim3(p.n_tiles), dim3(kSortThreads), 0, s, keys[cur], vals[cur],
                         keys[cur ^ 1], vals[cur ^ 1], p.segs, p.total_rows, pass, static_cast<const unsigned*>(nullptr),
                         static_cast<const unsigned*>(nullptr), 0u, false, hist, p.passes, p.n_tiles);
      rc = check_launch("radix pass (chained)");
      if (rc != RBX_OK) return rc;
      cur ^= 1;
      continue;
    }
    if (pass > 0)        // (pass 0's histograms come out of build_keys_kernel)
      hipLaunchKernelGGL(radix_hist_kernel<RB>, dim3(p.n_tiles), dim3(kSortThreads), 0, s, keys[cur], p.segs, p.total_rows,
                         pass, hist);
    const unsigned hist_len = p.n_tiles * R;
    const unsigned n_slices = (hist_len + kScanSlice - 1) / kScanSlice;
    hipLaunchKernelGGL(radix_scan_local_kernel, dim3(n_slices), dim3(1024), 0, s, hist, hist_len, ssum);
    const bool raw_sums = n_slices <= static_cast<unsigned>(kMaxFusedSlices);
    if (!raw_sums) hipLaunchKernelGGL(radix_scan_sums_kernel, dim3(1), dim3(1024), 0, s, ssum, n_slices);
    hipLaunchKernelGGL((radix_scatter_kernel<RB, false>), dim3(p.n_tiles), dim3(kSortThreads), 0, s, keys[cur], vals[cur],
                       keys[cur ^ 1], vals[cur ^ 1], p.segs, p.total_rows, pass, hist, ssum, n_slices, raw_sums,
                       static_cast<unsigned*>(nullptr), p.passes, p.n_tiles);
    rc = check_launch("radix pass");
    if (rc != RBX_OK) return rc;
    cur ^= 1;
  }
  return RBX_OK;
}

int run_sort(const BwdPlan& p, char* ws, int* d_status, hipStream_t s) {
  if (p.n_lookups == 0) return RBX_OK;
  switch (p.radix_bits) {
    case 8: return run_sort_rb<8>(p, ws, d_status, s);
    case 10: return run_sort_rb<10>(p, ws, d_status, s);
    default: return run_sort_rb<11>(p, ws, d_status, s);
  }
}
}  // namespace rbx

extern "C" int rbx_embed_sort(const rbx_field_t* fields, int32_t n_fields, int64_t batch, void* d_workspace,
                              size_t workspace_bytes, int32_t* d_status, void* stream) {
  using namespace rbx;
  BwdPlan p;
  int rc = make_plan(fields, n_fields, batch, nullptr, 0, &p);
  if (rc != RBX_OK) return rc;
  if (p.n_lookups == 0) return RBX_OK;
  if (d_workspace == nullptr || workspace_bytes < p.bytes)
    return fail(RBX_ERR_WORKSPACE, "workspace %zu B < required %zu B", workspace_bytes, p.bytes);
  return run_sort(p, static_cast<char*>(d_workspace), d_status, as_stream(stream));
}

extern "C" int rbx_embed_rezero(const rbx_field_t* fields, int32_t n_fields, int64_t batch, void* d_workspace,
                                size_t workspace_bytes, void* stream) {
  using namespace rbx;
  if (batch == 0) return RBX_OK;
  BwdPlan p;
  int rc = make_plan(fields, n_fields, batch, nullptr, 0, &p);
  if (rc != RBX_OK) return rc;
  if (p.n_lookups == 0) return RBX_OK;
  if (d_workspace == nullptr || workspace_bytes < p.bytes) return fail(RBX_ERR_WORKSPACE, "embed_rezero: workspace too small");
  return launch_rezero(p, static_cast<const char*>(d_workspace), as_stream(stream));
}

extern "C" int rbx_embed_bwd(const rbx_field_t* fields, int32_t n_fields, int64_t batch, const float* d_dout,
                             int64_t out_stride_b, const float* d_row_scale, int32_t accumulate, void* d_workspace,
                             size_t workspace_bytes, void* stream) {
  return rbx_embed_bwd_indexed(fields, n_fields, batch, d_dout, out_stride_b, nullptr, d_row_scale, accumulate, d_workspace,
                               workspace_bytes, stream);
}

extern "C" int rbx_embed_bwd_indexed(const rbx_field_t* fields, int32_t n_fields, int64_t batch, const float* d_dout,
                                     int64_t out_stride_b, const int32_t* d_dout_index, const float* d_row_scale,
                                     int32_t accumulate, void* d_workspace, size_t workspace_bytes, void* stream) {
  using namespace rbx;
  if (d_dout == nullptr) return fail(RBX_ERR_INVALID, "d_dout is NULL");
  if (batch == 0) return RBX_OK;
  BwdPlan p;
  int rc = make_plan(fields, n_fields, batch, d_dout, out_stride_b, &p);
  if (rc != RBX_OK) return rc;
  if (d_workspace == nullptr || workspace_bytes < p.bytes)
    return fail(RBX_ERR_WORKSPACE, "workspace %zu B < required %zu B", workspace_bytes, p.bytes);
  char* ws = static_cast<char*>(d_workspace);
  hipStream_t s = as_stream(stream);
  if (p.n_lookups > 0) {
    for (int i = 0; i < p.n_cat; ++i) {
      const unsigned char pool = p.red.f[i].pool;
      if ((pool == RBX_POOL_MEAN_VALUE || pool == RBX_POOL_MEAN_ID) && d_row_scale == nullptr)
        return fail(RBX_ERR_INVALID, "mean pooling backward needs d_row_scale from the forward");
    }
    const int cur = p.passes & 1;
    const unsigned* keys = reinterpret_cast<const unsigned*>(ws + p.off_keys[cur]);
    const unsigned* vals = reinterpret_cast<const unsigned*>(ws + p.off_vals[cur]);
    const GenericPolicy::Args args = {d_dout, static_cast<long long>(out_stride_b), d_row_scale,
                                      static_cast<long long>(batch), accumulate, d_dout_index};
    rc = p.vec ? dispatch_reduce<GenericPolicy, true>(p, args, keys, vals, ws, s)
               : dispatch_reduce<GenericPolicy, false>(p, args, keys, vals, ws, s);
    if (rc != RBX_OK) return rc;
  }
  if (p.n_num > 0 && d_dout_index != nullptr) return fail(RBX_ERR_UNSUPPORTED, "embed_bwd_indexed: numeric features are not indexed");
  if (p.n_num > 0) {
    float* partial = reinterpret_cast<float*>(ws + p.off_num);
    hipLaunchKernelGGL(numeric_partial_kernel, dim3(p.num_blocks, p.n_num), dim3(256), 0, s, p.num,
                       static_cast<long long>(batch), d_dout, static_cast<long long>(out_stride_b), partial, p.max_dim,
                       p.num_blocks);
    hipLaunchKernelGGL(numeric_final_kernel, dim3(p.n_num, p.max_dim), dim3(64), 0, s, p.num, partial, p.max_dim,
                       p.num_blocks);
    rc = check_launch("numeric grad kernels");
    if (rc != RBX_OK) return rc;
  }
  return RBX_OK;
}
