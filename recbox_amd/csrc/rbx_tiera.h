// rbx_tiera.h -- the sort-free half of the fused FM backward: tables of up to 16 384 rows ("tier A").
//
// Reference behaviour replaced: autograd's embedding_dense_backward behind the nn.Embedding tables of
// ranking/pytorch/layers/embeddings/feature_embedding.py:89-103 (dense [V, D] gradient, padding_idx row zero) for the
// low-cardinality fields of a CTR batch.
//
// Why a second path.  At the Criteo shape 18 of the 26 tables have <= 15 000 rows; they carry 69 % of the (row, sample)
// pairs, all of the long runs of equal rows (65 536 lookups on 3..30 rows), and 3 MB of gradient in total.  Sending them
// through the global LSD radix sort made that sort (three dependent kernels per 8-bit pass, latency-bound) and the
// run fix-up launches the critical path of the step.  Here nothing global is sorted:
//
//   compact_ids   (ids only)  id columns of any dtype / stride -> int32 [field][B], range-checked once
//   ta_blocksort  (ids only)  one workgroup per (field, block of 2048 samples): stable LSD radix sort of the block's
//                             (row, sample offset) pairs entirely in LDS; leaves the sorted pairs and a presence
//                             bitmap [block][row]
//   ta_reduce     (g, S)      same grid: segmented sum of g_b * S_b and g_b over the sorted pairs of the block, one
//                             partial per (block, present row), written into a dense [block][row] array
//   ta_final      (tables)    one lane group per table row: adds the row's partials over fields and blocks in ascending
//                             order, subtracts w_r * sum g, and WRITES the row -- every row of a tier-A table is written
//                             every step, so these tables need no re-zeroing and no fix-up launches
//
// Every sum has a fixed order (sample order inside a chunk of 32 sorted pairs, chunk order inside a block, block order
// inside a table): gradients are bit-identical run to run, no float atomics.  The two ids-only kernels run on the side
// stream beside the forward (rbx_fm_sort), the other two after the loss (rbx_fm_bwd).
#pragma once
#include "rbx_bwd_common.h"

namespace rbx {

constexpr int kTaBlock = 2048;                       // samples per (field, block) unit = one LDS sort tile
constexpr int kTaOffBits = 11;
constexpr unsigned kTaOffMask = (1u << kTaOffBits) - 1u;
#define RBX_TA_CHUNK 16
#define RBX_TA_THREADS 512
constexpr int kTaChunk = RBX_TA_CHUNK;               // sorted pairs one lane group walks in sequence
constexpr int kTaChunks = kTaBlock / kTaChunk;       // 128
constexpr int kTaThreads = RBX_TA_THREADS;           // of ta_reduce_kernel: 8 wavefronts per (field, block) unit
constexpr int kTaMaxVocab = 16384;
constexpr int kTaMaxDim = 64;
constexpr unsigned kTaNone = 0xFFFFFFFEu;            // "no element" in the boundary-run list

struct CidField {            // 24 B: how to read one categorical id column
  const void* ids;
  long long stride_b;
  int vocab;
  int dtype;
};
struct CidPack { CidField f[RBX_MAX_FIELDS]; };

struct TaField {             // 16 B
  int cid_row;               // row of the compact id matrix
  int vocab;
  unsigned frow0;            // rows of the tier-A fields before this one (partial-array addressing)
  unsigned fword0;           // bitmap words per block of the fields before this one
};
struct TaFieldPack { TaField f[RBX_MAX_FIELDS]; };

struct TaTable {             // 48 B
  float* grad;               // [V, D] or NULL
  float* grad2;              // [V] (LR weight gradient) or NULL
  const float* table;        // [V, stride] embedding rows (dW = A - cnt * w) or NULL
  int stride;
  int vocab;
  int pad;                   // padding_idx (kNoId when unset): that row's gradient is zero
  short f_begin, f_count;    // its fields in the TaFieldPack (contiguous)
  unsigned row0;             // rows of the tier-A tables before this one (grid mapping of ta_final)
  int reserved;
};
struct TaTablePack { TaTable t[RBX_MAX_FIELDS]; };
struct TaFieldRef { unsigned frow0, fword0; };   // what ta_final_kernel needs of a TaField
struct TaFieldRefPack { TaFieldRef f[RBX_MAX_FIELDS]; };
static_assert(sizeof(TaTablePack) + sizeof(TaFieldRefPack) + 128 <= 4096, "ta_final_kernel's arguments must fit the kernarg segment");

// id of any dtype -> int32 row number; false when it lies outside [0, vocab) (NaN included).  An id that passes is
// < 2^31, so one v_cvt_i32_f64 replaces the ~20 emulated instructions of static_cast<long long>(double).
__device__ __forceinline__ bool ta_decode_id(long long raw, int dt, int vocab, int* id) {
  switch (dt) {
    case RBX_I32:
      *id = static_cast<int>(raw);
      return static_cast<unsigned>(*id) < static_cast<unsigned>(vocab);
    case RBX_I64:
      *id = static_cast<int>(raw);
      return static_cast<unsigned long long>(raw) < static_cast<unsigned long long>(vocab);
    case RBX_F32: {
      const float f = __int_as_float(static_cast<int>(raw));
      *id = __float2int_rz(f);                               // .long() truncates towards zero
      return (f == f) && static_cast<unsigned>(*id) < static_cast<unsigned>(vocab);
    }
    default: {
      const double d = __longlong_as_double(raw);
      *id = __double2int_rz(d);
      return (d == d) && static_cast<unsigned>(*id) < static_cast<unsigned>(vocab);
    }
  }
}

// ---- ids -> int32 [n][B] ----------------------------------------------------------------------------------------
// A workgroup transposes a tile of `ts` samples x n columns through LDS; ts is sized so that the tile is at most 2048
// ids: every thread issues ALL of its (up to 8) loads before it touches the first value -- one memory round trip per
// workgroup, and B / ts workgroups (1024 at the Criteo shape) keep the chip busy.  (256-sample tiles walked in 7 batches
// of 4 loads by 256 workgroups took 28 us for the same 21 MB.)  FIELD_FAST: consecutive threads read consecutive
// COLUMNS of one sample (the reference's loader hands over one [B, cols] tensor: a sample's ids are neighbours in
// memory); otherwise consecutive samples of one column (separate contiguous id tensors).  Either way the stores are `ts`
// consecutive ints per column.  Out-of-range ids become -1 and raise the status word.
constexpr int kCidPerThread = 8;
template <bool FIELD_FAST>
__global__ __launch_bounds__(256) void compact_ids_kernel(const CidPack P, const int n, const long long B, const int ts,
                                                          int* __restrict__ cid, int* __restrict__ status) {
  extern __shared__ int ta_lds[];
  CidField* sf = reinterpret_cast<CidField*>(ta_lds);                       // [n]
  int* tile = ta_lds + (RBX_MAX_FIELDS * sizeof(CidField)) / sizeof(int);   // [n][ts + 1]
  {
    const int words = n * static_cast<int>(sizeof(CidField) / 4);
    const int* src = reinterpret_cast<const int*>(&P);
    int* dst = reinterpret_cast<int*>(sf);
    for (int i = threadIdx.x; i < words; i += blockDim.x) dst[i] = src[i];
  }
  __syncthreads();
  const long long b0 = static_cast<long long>(blockIdx.x) * ts;
  const int live = static_cast<int>((B - b0 < ts) ? (B - b0) : ts);
  const int pitch = ts + 1;
  const int total = n * ts;                             // <= 256 * kCidPerThread
  constexpr int U = kCidPerThread;
  long long raw[U];
  int cc[U], ss[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int idx = u * 256 + threadIdx.x;
    int c, s;
    if (FIELD_FAST) { s = idx / n; c = idx - s * n; } else { c = idx / ts; s = idx - c * ts; }
    cc[u] = c;
    ss[u] = s;
    raw[u] = 0;
    if (idx < total && s < live) raw[u] = load_raw(sf[c].ids, (b0 + s) * sf[c].stride_b, sf[c].dtype);
  }
  bool bad = false;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int idx = u * 256 + threadIdx.x;
    if (idx < total && ss[u] < live) {
      int v;
      if (!ta_decode_id(raw[u], sf[cc[u]].dtype, sf[cc[u]].vocab, &v)) {
        v = -1;
        bad = true;
      }
      tile[cc[u] * pitch + ss[u]] = v;
    }
  }
  if (bad && status != nullptr) atomicOr(status, 1);
  __syncthreads();
  for (int idx = threadIdx.x; idx < total; idx += 256) {
    const int c = idx / ts, s = idx - c * ts;
    if (s < live) cid[static_cast<size_t>(c) * B + b0 + s] = tile[c * pitch + s];
  }
}

// ---- one stable 8-bit LSD pass over the 2048 keys of a workgroup, in LDS -------------------------------------------
// Wave w owns the contiguous quarter [w*512, (w+1)*512) and walks it in 64-key steps, so key order == (wave, step,
// lane) and the ranks respect it (same scheme as radix_scatter_kernel in rbx_embed_bwd.hip, without the global part).
__device__ __forceinline__ void ta_radix_pass(unsigned (&key)[8], const int shift, unsigned (*wcnt)[256],
                                              unsigned* dstart, unsigned* wtot, unsigned* out) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 4 * 256; i += 256) (&wcnt[0][0])[i] = 0;
  __syncthreads();
  unsigned rank[8];
  const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const unsigned d = (key[s] >> shift) & 255u;
    unsigned long long peers = ~0ull;
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const unsigned long long m = __ballot((d >> b) & 1u);
      peers &= ((d >> b) & 1u) ? m : ~m;
    }
    const unsigned before = __popcll(peers & lt);
    // LDS ops of one wave issue in program order: every peer reads the running count before the run leader (lowest
    // peer lane) bumps it.  volatile: no caching across steps.
    volatile unsigned* wc = wcnt[wid];
    const unsigned prev = wc[d];
    rank[s] = prev + before;
    if (before == 0) wc[d] = prev + __popcll(peers);
  }
  __syncthreads();
  {
    const int d = threadIdx.x;                       // one digit per thread: exclusive prefix over waves, then over digits
    unsigned run = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const unsigned c = wcnt[w][d];
      wcnt[w][d] = run;
      run += c;
    }
    unsigned inc = run;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const unsigned t = __shfl_up(inc, o, 64);
      if (lane >= o) inc += t;
    }
    if (lane == 63) wtot[wid] = inc;
    __syncthreads();
    unsigned base = inc - run;
    for (int w = 0; w < wid; ++w) base += wtot[w];
    dstart[d] = base;
  }
  __syncthreads();
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const unsigned d = (key[s] >> shift) & 255u;
    out[dstart[d] + wcnt[wid][d] + rank[s]] = key[s];
  }
  __syncthreads();
#pragma unroll
  for (int s = 0; s < 8; ++s) key[s] = out[wid * 512 + s * 64 + lane];
}

// ---- ids only: sort one (field, block) unit by row, stable in the sample -------------------------------------------
// key = row << 11 | sample offset inside the block; an id outside the table sorts behind every row (row = vocab), the
// filler of a short last block behind that (all ones).  Only the row bits are sorted: the keys start in sample order
// and the passes are stable.
template <int kUnused = 0>              // (a template only so that the header may be included by several translation units)
__global__ __launch_bounds__(256) void ta_blocksort_kernel(const TaFieldPack P, const int n_fld, const long long B,
                                                           const int* __restrict__ cid, unsigned* __restrict__ sorted,
                                                           unsigned* __restrict__ bitmap, const unsigned NB) {
  __shared__ unsigned buf[kTaBlock];
  __shared__ unsigned wcnt[4][256];
  __shared__ unsigned dstart[256];
  __shared__ unsigned wtot[4];
  __shared__ unsigned bm[kTaMaxVocab / 32];
  const int f = blockIdx.x % n_fld;
  const unsigned k = blockIdx.x / n_fld;
  const TaField fd = P.f[f];
  const int V = fd.vocab;
  const long long b0 = static_cast<long long>(k) * kTaBlock;
  const unsigned n = static_cast<unsigned>((B - b0 < kTaBlock) ? (B - b0) : kTaBlock);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int words = (V + 31) >> 5;
  for (int i = threadIdx.x; i < words; i += 256) bm[i] = 0;
  const int* src = cid + static_cast<size_t>(fd.cid_row) * B + b0;
  unsigned key[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const unsigned off = wid * 512 + s * 64 + lane;
    key[s] = 0xFFFFFFFFu;
    if (off < n) {
      const int id = src[off];
      const unsigned row = (id >= 0 && id < V) ? static_cast<unsigned>(id) : static_cast<unsigned>(V);
      key[s] = (row << kTaOffBits) | off;
    }
  }
  ta_radix_pass(key, kTaOffBits, wcnt, dstart, wtot, buf);
  if (V > 255) ta_radix_pass(key, kTaOffBits + 8, wcnt, dstart, wtot, buf);   // rows 0..V (V = the out-of-range bucket)
  unsigned* dst = sorted + static_cast<size_t>(blockIdx.x) * kTaBlock;
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    dst[wid * 512 + s * 64 + lane] = key[s];
    const unsigned row = key[s] >> kTaOffBits;
    if (row < static_cast<unsigned>(V)) atomicOr(&bm[row >> 5], 1u << (row & 31u));
  }
  __syncthreads();
  unsigned* bdst = bitmap + static_cast<size_t>(fd.fword0) * NB + static_cast<size_t>(k) * words;
  for (int i = threadIdx.x; i < words; i += 256) bdst[i] = bm[i];
}

// ---- g, S: segmented sum over the sorted pairs of one (field, block) unit -------------------------------------------
// Lane group j walks the chunks j, j + NG, ... of kTaChunk sorted pairs.  A run of equal rows that lies inside a chunk and is
// neither its first nor its last run is complete: its sum goes straight to the partial array.  The first and the last
// run of every chunk go to an LDS list of 128 elements (chunk order); after a barrier the element that starts a row's
// stretch adds the following elements of the same row in list order and writes the partial.  (Chunks of 32 pairs walked
// by 4 wavefronts per unit: 35 us at the Criteo shape -- 2.25 wavefronts per SIMD, each waiting for 4 dependent rounds of
// gathers; 16 pairs and 8 wavefronts: half the rounds, twice the wavefronts.)  A row of the block is
// written exactly once, absent rows are not written (the bitmap of ta_blocksort_kernel names the present ones).
template <int G, bool VEC>
__global__ __launch_bounds__(kTaThreads) void ta_reduce_kernel(const TaFieldPack P, const int n_fld, const unsigned NB,
                                                        const float* __restrict__ g, const float* __restrict__ ssum,
                                                        const int D, const unsigned* __restrict__ sorted,
                                                        float* __restrict__ psum, float* __restrict__ pcnt) {
  using F = Frag<G, 1, VEC>;
  constexpr int NG = kTaThreads / G;
  constexpr int NE = 2 * kTaChunks;
  extern __shared__ float ta_red[];
  float* esum = ta_red;                                        // [NE][D]
  float* ecnt = esum + NE * D;                                 // [NE]
  unsigned* erow = reinterpret_cast<unsigned*>(ecnt + NE);    // [NE]
  const int f = blockIdx.x % n_fld;
  const unsigned k = blockIdx.x / n_fld;
  const TaField fd = P.f[f];
  const unsigned V = static_cast<unsigned>(fd.vocab);
  const size_t b0 = static_cast<size_t>(k) * kTaBlock;
  const unsigned* src = sorted + static_cast<size_t>(blockIdx.x) * kTaBlock;
  const size_t pbase = static_cast<size_t>(fd.frow0) * NB + static_cast<size_t>(k) * V;
  float* ps = psum + pbase * D;
  float* pc = pcnt + pbase;
  const int lane_g = threadIdx.x % G, group = threadIdx.x / G;
  constexpr int U = 8;
  for (int c = group; c < kTaChunks; c += NG) {
    const unsigned* e = src + c * kTaChunk;
    unsigned cur = e[0] >> kTaOffBits;
    F acc;
    acc.zero();
    float cnt = 0.f;
    int nrun = 0;
    for (int i0 = 0; i0 < kTaChunk; i0 += U) {
      unsigned ent[U];
#pragma unroll
      for (int u = 0; u < U; ++u) ent[u] = e[i0 + u];
      F rows[U];
      float gg[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        rows[u].zero();
        gg[u] = 0.f;
        if ((ent[u] >> kTaOffBits) < V) {
          const size_t b = b0 + (ent[u] & kTaOffMask);
          gg[u] = g[b];
          if (ssum != nullptr) rows[u].fma_from(ssum + b * D, D, lane_g, gg[u]);
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const unsigned row = ent[u] >> kTaOffBits;
        if (row != cur) {                                      // the run of `cur` ends in front of this pair
          if (nrun == 0) {
            acc.store(esum + (2 * c) * D, D, lane_g);
            if (lane_g == 0) { ecnt[2 * c] = cnt; erow[2 * c] = cur; }
          } else if (cur < V) {
            acc.store(ps + static_cast<size_t>(cur) * D, D, lane_g);
            if (lane_g == 0) pc[cur] = cnt;
          }
          ++nrun;
          acc.zero();
          cnt = 0.f;
          cur = row;
        }
        frag_add(acc, rows[u]);
        cnt += gg[u];
      }
    }
    const int slot = (nrun == 0) ? 2 * c : 2 * c + 1;           // the run that is open at the end of the chunk
    acc.store(esum + slot * D, D, lane_g);
    if (lane_g == 0) {
      ecnt[slot] = cnt;
      erow[slot] = cur;
      if (nrun == 0) erow[2 * c + 1] = kTaNone;
    }
  }
  __syncthreads();
  for (int i = group; i < NE; i += NG) {
    const unsigned r = erow[i];
    if (r == kTaNone || r >= V) continue;
    int p = i - 1;
    if (p >= 0 && erow[p] == kTaNone) --p;                      // (element 2c always exists: at most one gap)
    if (p >= 0 && erow[p] == r) continue;                       // the stretch of this row started earlier
    F acc;
    acc.zero();
    acc.add_from(esum + i * D, D, lane_g);
    float cnt = ecnt[i];
    for (int j = i + 1; j < NE; ++j) {
      const unsigned rj = erow[j];
      if (rj == kTaNone) continue;
      if (rj != r) break;
      acc.add_from(esum + j * D, D, lane_g);
      cnt += ecnt[j];
    }
    acc.store(ps + static_cast<size_t>(r) * D, D, lane_g);
    if (lane_g == 0) pc[r] = cnt;
  }
}

// ---- numeric features of the fused FM backward as extra workgroups of ta_reduce_lds_kernel's launch -------------------
// The batch reductions of the numeric features' weights (sum_b g_b x_bf S_b, sum g x^2, sum g x) and of the bias (sum g) were
// a launch of their own at the tail of the backward's main chain (fm_numeric_partial_kernel, ~21 us: the chain then ended
// with the large tables' chain on the other hardware queue, and the next step's first kernel paid a cross-queue edge).
// Here they are the LAST workgroups of tier A's first launch: that launch holds one 157 KB workgroup per CU and at the
// Criteo shape 224 of them, so the numeric workgroups -- one per 512 samples, ~5 us each -- flow through the CUs the
// tables leave free while the tables' workgroups run.  A workgroup issues every x load of its samples at once (the
// columns of a sample's row, feature fastest), stages the samples' S rows and g meanwhile, then thread (o, h) sums
// output o over half h of the samples in sample order (four interleaved sums), the halves meet in a fixed order.
// Partial layout as fm_numeric_partial_kernel's ([output][block]): fm_numeric_final_kernel reads it.
constexpr int kTaNumMax = 32;               // numeric features this form serves (kernarg space: 20 B each)
constexpr int kTaNumBlock = 512;            // samples per numeric workgroup
struct TaNumPack {
  const void* ids[kTaNumMax];
  long long stride_b[kTaNumMax];
  int dtype[kTaNumMax];
  int n;
  int reserved;
};
static inline size_t ta_num_lds_bytes(int n_num, int D) {
  return (static_cast<size_t>(kTaNumBlock) * D + kTaNumBlock + static_cast<size_t>(n_num) * (kTaNumBlock + 1) + 256) * 4;
}

__device__ __forceinline__ void ta_numeric_block(const TaNumPack& N, const unsigned k, const unsigned nblk, const long long B,
                                                 const float* __restrict__ g, const float* __restrict__ ssum, const int D,
                                                 float* lds, float* __restrict__ partial) {
  constexpr int SB = kTaNumBlock, SP = SB + 1;
  const int n_num = N.n;
  float* sS = lds;                                  // [SB][D]
  float* sg = sS + SB * D;                          // [SB]
  float* sx = sg + SB;                              // [n_num][SB + 1]
  float* meet = sx + n_num * SP;                    // [256]
  const size_t b0 = static_cast<size_t>(k) * SB;
  const int n = static_cast<int>((static_cast<size_t>(B) - b0 < SB) ? (static_cast<size_t>(B) - b0) : SB);
  const int tid = static_cast<int>(threadIdx.x);    // == the thread's sample (SB == the workgroup's 512 threads)
  // every x load of the thread's sample in flight at once; the feature index is a compile-time constant (a lane-varying
  // index into the by-value pack would send it to scratch), the columns of a row share its cache lines
  long long raw[kTaNumMax];
#pragma unroll
  for (int f = 0; f < kTaNumMax; ++f) {
    raw[f] = 0;
    if (f < n_num && tid < n) raw[f] = load_raw(N.ids[f], static_cast<long long>(b0 + tid) * N.stride_b[f], N.dtype[f]);
  }
  {
    const int n4 = SB * D / 4, live4 = n * D / 4;
    const float4* src = reinterpret_cast<const float4*>(ssum + b0 * D);
    for (int i = tid; i < n4; i += 512) reinterpret_cast<float4*>(sS)[i] = (i < live4) ? src[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    sg[tid] = (tid < n) ? g[b0 + tid] : 0.f;
  }
#pragma unroll
  for (int f = 0; f < kTaNumMax; ++f)
    if (f < n_num) sx[f * SP + tid] = (tid < n) ? decode_value(raw[f], N.dtype[f]) : 0.f;
  __syncthreads();
  const int stride = D + 2;
  const int n_out = n_num * stride + 1;
  const int h = tid >> 8;
  // outputs in rounds of 256: thread (o, h) sums output o over half h of the samples, four interleaved sums
  for (int o0 = 0; o0 < n_out; o0 += 256) {
    const int o = o0 + (tid & 255);
    float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
    if (o < n_out) {
      const int f = o / stride, slot = o - f * stride;
      const int i0 = h * (SB / 2);
      const float* gg = sg + i0;
      if (f == n_num) {
        for (int i = 0; i < SB / 2; i += 4) { t0 += gg[i]; t1 += gg[i + 1]; t2 += gg[i + 2]; t3 += gg[i + 3]; }
      } else {
        const float* a = sx + f * SP + i0;
        if (slot < D) {
          const float* c = sS + static_cast<size_t>(i0) * D + slot;
          for (int i = 0; i < SB / 2; i += 4) {
            t0 += gg[i] * a[i] * c[i * D];
            t1 += gg[i + 1] * a[i + 1] * c[(i + 1) * D];
            t2 += gg[i + 2] * a[i + 2] * c[(i + 2) * D];
            t3 += gg[i + 3] * a[i + 3] * c[(i + 3) * D];
          }
        } else if (slot == D) {
          for (int i = 0; i < SB / 2; i += 4) {
            t0 += gg[i] * a[i] * a[i];
            t1 += gg[i + 1] * a[i + 1] * a[i + 1];
            t2 += gg[i + 2] * a[i + 2] * a[i + 2];
            t3 += gg[i + 3] * a[i + 3] * a[i + 3];
          }
        } else {
          for (int i = 0; i < SB / 2; i += 4) {
            t0 += gg[i] * a[i]; t1 += gg[i + 1] * a[i + 1]; t2 += gg[i + 2] * a[i + 2]; t3 += gg[i + 3] * a[i + 3];
          }
        }
      }
    }
    const float mine = (t0 + t1) + (t2 + t3);
    if (h == 1) meet[tid & 255] = mine;
    __syncthreads();
    if (h == 0 && o < n_out) partial[static_cast<size_t>(o) * nblk + k] = mine + meet[tid & 255];
    __syncthreads();                                 // `meet` is rewritten by the next round
  }
}

// ---- the same for rows of up to 16 floats: the block's S rows staged in LDS ----------------------------------------
// A (field, block) unit reads every S row of its block exactly once, in sorted-by-row (= random) order: 18 fields x
// 65 536 gathers of 64 bytes at the Criteo shape, which the L2s (4 MB each, S is 4.2 MB) do not hold -- the gather
// kernel above takes 43 us alone and slows whatever runs beside it.  Here a workgroup owns one block of 2048 samples and a
// GROUP of fields: it streams the block's S rows (128 KB at D = 16) and g into LDS once, coalesced, then serves the fields
// one after the other from LDS.  The boundary runs of a field's 128 chunks are combined by a segmented Hillis-Steele scan
// over the 256-element list (8 steps; element i adds element i - d iff both carry the same row -- rows are sorted, so
// equal rows are contiguous): a fixed tree over list positions, whatever the run lengths, instead of one lane group
// walking the list of a 700-pair run.  157 KB of LDS: one such workgroup per CU (kernels with small LDS needs still fit
// beside it).
constexpr int kTaLdsMaxDim = 16;
template <int G>
__global__ __launch_bounds__(512) void ta_reduce_lds_kernel(const TaFieldPack P, const int n_fld, const int fpg,
                                                            const int n_grp, const unsigned NB, const long long B,
                                                            const float* __restrict__ g, const float* __restrict__ ssum,
                                                            const int D, const unsigned* __restrict__ sorted,
                                                            float* __restrict__ psum, float* __restrict__ pcnt,
                                                            const TaNumPack N, float* __restrict__ num_partial) {
  using F = Frag<G, 1, true>;
  constexpr int NE = 2 * kTaChunks;                  // 256 boundary elements; 512 / G >= 128 lane groups
  static_assert(kTaChunk == 16 && kTaChunks == 128, "ta_reduce_lds_kernel is written for 128 chunks of 16 pairs");
  extern __shared__ __attribute__((aligned(16))) float ta_stage[];
  float* sS = ta_stage;                              // [2048][D]
  float* sg = sS + kTaBlock * D;                     // [2048]
  float* esum = sg + kTaBlock;                       // [NE][D]
  float* ecnt = esum + NE * D;                       // [NE]
  unsigned* erow = reinterpret_cast<unsigned*>(ecnt + NE);   // [NE]
  const unsigned n_tabwg = NB * static_cast<unsigned>(n_grp);
  if (blockIdx.x >= n_tabwg) {       // the launch's last workgroups: the numeric features (N.n > 0).  (Dispatched FIRST they
    ta_numeric_block(N, blockIdx.x - n_tabwg, gridDim.x - n_tabwg, B, g, ssum, D, ta_stage, num_partial);    // shorten this
    return;                          // launch, 51 vs 57 us, and the step not at all: profiles/r06/fm_numeric_in_tier_a.txt)
  }
  const unsigned bid = blockIdx.x;
  const unsigned k = bid / n_grp;
  const int grp = bid % n_grp;
  const size_t b0 = static_cast<size_t>(k) * kTaBlock;
  const int n = static_cast<int>((static_cast<size_t>(B) - b0 < kTaBlock) ? (static_cast<size_t>(B) - b0) : kTaBlock);
  // ---- stage S[b0 .. b0 + n) and g: every thread's loads in flight together ----
  {
    const int n4 = kTaBlock * D / 4, live4 = n * D / 4;            // float4 units
    const float4* src = reinterpret_cast<const float4*>(ssum + b0 * D);
    float4* dst = reinterpret_cast<float4*>(sS);
    for (int base = 0; base < n4; base += 8 * 512) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = base + u * 512 + threadIdx.x;
        v[u] = (i < live4) ? src[i] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = base + u * 512 + threadIdx.x;
        if (i < n4) dst[i] = v[u];
      }
    }
    for (int i = threadIdx.x; i < kTaBlock; i += 512) sg[i] = (i < n) ? g[b0 + i] : 0.f;
  }
  __syncthreads();
  const int lane_g = threadIdx.x % G, group = threadIdx.x / G;
  const int f_end = ((grp + 1) * fpg < n_fld) ? (grp + 1) * fpg : n_fld;
  for (int f = grp * fpg; f < f_end; ++f) {
    const TaField fd = P.f[f];
    const unsigned V = static_cast<unsigned>(fd.vocab);
    const unsigned* src = sorted + (static_cast<size_t>(k) * n_fld + f) * kTaBlock;
    const size_t pbase = static_cast<size_t>(fd.frow0) * NB + static_cast<size_t>(k) * V;
    float* ps = psum + pbase * D;
    float* pc = pcnt + pbase;
    if (group < kTaChunks) {
      const int c = group;
      const uint4* e4 = reinterpret_cast<const uint4*>(src + c * kTaChunk);
      unsigned ent[kTaChunk];
#pragma unroll
      for (int q = 0; q < kTaChunk / 4; ++q) {
        const uint4 t = e4[q];
        ent[q * 4] = t.x; ent[q * 4 + 1] = t.y; ent[q * 4 + 2] = t.z; ent[q * 4 + 3] = t.w;
      }
      unsigned cur = ent[0] >> kTaOffBits;
      F acc;
      acc.zero();
      float cnt = 0.f;
      int nrun = 0;
#pragma unroll
      for (int u = 0; u < kTaChunk; ++u) {
        const unsigned row = ent[u] >> kTaOffBits;
        const unsigned off = ent[u] & kTaOffMask;
        if (row != cur) {                                        // the run of `cur` ends in front of this pair
          if (nrun == 0) {
            acc.store(esum + (2 * c) * D, D, lane_g);
            if (lane_g == 0) { ecnt[2 * c] = cnt; erow[2 * c] = cur; }
          } else if (cur < V) {
            acc.store(ps + static_cast<size_t>(cur) * D, D, lane_g);
            if (lane_g == 0) pc[cur] = cnt;
          }
          ++nrun;
          acc.zero();
          cnt = 0.f;
          cur = row;
        }
        if (row < V) {
          const float gv = sg[off];
          acc.fma_from(sS + off * D, D, lane_g, gv);
          cnt += gv;
        }
      }
      if (nrun == 0) {                                            // one run fills the chunk: its sum in 2c, nothing in 2c + 1
        acc.store(esum + (2 * c) * D, D, lane_g);
        F z;
        z.zero();
        z.store(esum + (2 * c + 1) * D, D, lane_g);
        if (lane_g == 0) { ecnt[2 * c] = cnt; erow[2 * c] = cur; ecnt[2 * c + 1] = 0.f; erow[2 * c + 1] = cur; }
      } else {
        acc.store(esum + (2 * c + 1) * D, D, lane_g);
        if (lane_g == 0) { ecnt[2 * c + 1] = cnt; erow[2 * c + 1] = cur; }
      }
    }
    __syncthreads();
    // ---- segmented inclusive scan over the NE boundary elements; lane group j holds elements j and j + 128 ----
    F v0, v1;
    v0.zero();
    v1.zero();
    float c0 = 0.f, c1 = 0.f;
    unsigned r0 = kTaNone, r1 = kTaNone;
    const int i0 = group, i1 = group + kTaChunks;
    if (group < kTaChunks) {
      v0.add_from(esum + i0 * D, D, lane_g);
      v1.add_from(esum + i1 * D, D, lane_g);
      c0 = ecnt[i0]; c1 = ecnt[i1];
      r0 = erow[i0]; r1 = erow[i1];
    }
#pragma unroll 1
    for (int d = 1; d < NE; d <<= 1) {
      F a0, a1;
      a0.zero();
      a1.zero();
      float ac0 = 0.f, ac1 = 0.f;
      if (group < kTaChunks) {
        if (i0 >= d && erow[i0 - d] == r0) { a0.add_from(esum + (i0 - d) * D, D, lane_g); ac0 = ecnt[i0 - d]; }
        if (i1 >= d && erow[i1 - d] == r1) { a1.add_from(esum + (i1 - d) * D, D, lane_g); ac1 = ecnt[i1 - d]; }
      }
      __syncthreads();
      if (group < kTaChunks) {
        frag_add(v0, a0); c0 += ac0;
        frag_add(v1, a1); c1 += ac1;
        v0.store(esum + i0 * D, D, lane_g);
        v1.store(esum + i1 * D, D, lane_g);
        if (lane_g == 0) { ecnt[i0] = c0; ecnt[i1] = c1; }
      }
      __syncthreads();
    }
    if (group < kTaChunks) {                                       // the last element of a row's stretch holds the row's sum
      if (r0 < V && erow[i0 + 1] != r0) {                         // (i0 + 1 <= 128 < NE)
        v0.store(ps + static_cast<size_t>(r0) * D, D, lane_g);
        if (lane_g == 0) pc[r0] = c0;
      }
      if (r1 < V && (i1 + 1 == NE || erow[i1 + 1] != r1)) {
        v1.store(ps + static_cast<size_t>(r1) * D, D, lane_g);
        if (lane_g == 0) pc[r1] = c1;
      }
    }
    __syncthreads();                                               // the list is rewritten by the next field
  }
}

// ---- tables: combine the block partials of every row, apply dW = A - cnt * w, write the row -----------------------
// A row is served by KQ lane groups of G lanes: group q adds the partials of the blocks q, q + KQ, ... (fields in
// order, blocks ascending, 8 bitmap words and then up to 8 partial rows in flight), the KQ sums meet in a fixed xor
// butterfly.  (One lane group per row walking all blocks: 23 us at the Criteo shape, four dependent rounds of two
// memory trips each at 3 wavefronts per SIMD.)
template <int G, bool VEC>
__global__ __launch_bounds__(256) void ta_final_kernel(const TaTablePack T, const TaFieldRefPack P, const int n_tab,
                                                       const int n_fld, const unsigned NB, const int D,
                                                       const unsigned total_rows, const float* __restrict__ psum,
                                                       const float* __restrict__ pcnt,
                                                       const unsigned* __restrict__ bitmap, const int accumulate,
                                                       const bool has_emb) {
  using F = Frag<G, 1, VEC>;
  constexpr int KQ = (G >= 16) ? 1 : 16 / G;
  constexpr int LR = G * KQ;                         // lanes per row
  constexpr int NR = 256 / LR;                       // rows per workgroup
  __shared__ TaTable st[RBX_MAX_FIELDS];
  __shared__ TaFieldRef sfd[RBX_MAX_FIELDS];
  {
    const int* src = reinterpret_cast<const int*>(&T);
    int* dst = reinterpret_cast<int*>(st);
    for (int i = threadIdx.x; i < n_tab * static_cast<int>(sizeof(TaTable) / 4); i += 256) dst[i] = src[i];
    const int* src2 = reinterpret_cast<const int*>(&P);
    int* dst2 = reinterpret_cast<int*>(sfd);
    for (int i = threadIdx.x; i < n_fld * static_cast<int>(sizeof(TaFieldRef) / 4); i += 256) dst2[i] = src2[i];
  }
  __syncthreads();
  const int lane_r = threadIdx.x % LR;
  const int lane_g = lane_r % G;
  const unsigned kq = static_cast<unsigned>(lane_r / G);
  const unsigned R = blockIdx.x * NR + threadIdx.x / LR;
  if (R >= total_rows) return;                        // (all lanes of a row leave together: the shuffles below stay inside a row)
  int lo = 0, hi = n_tab - 1;                         // last table with row0 <= R
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (st[mid].row0 <= R) lo = mid; else hi = mid - 1;
  }
  const TaTable& tb = st[lo];
  const unsigned r = R - tb.row0;
  const unsigned V = static_cast<unsigned>(tb.vocab);
  F wrow;                                             // w_r, fetched ahead of the walk
  wrow.zero();
  if (kq == 0 && tb.grad != nullptr) wrow.add_from(tb.table + static_cast<size_t>(r) * tb.stride, D, lane_g);
  F acc;
  acc.zero();
  float cnt = 0.f;
  int any = 0;
  constexpr int U = 8;
  const unsigned words = (V + 31u) >> 5;
  for (int f = tb.f_begin; f < tb.f_begin + tb.f_count; ++f) {
    const TaFieldRef& fd = sfd[f];
    const unsigned* bw = bitmap + static_cast<size_t>(fd.fword0) * NB + (r >> 5);
    const size_t pbase = static_cast<size_t>(fd.frow0) * NB + r;
    for (unsigned k0 = kq; k0 < NB; k0 += KQ * U) {
      unsigned w[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const unsigned k = k0 + u * KQ;
        w[u] = (k < NB) ? bw[static_cast<size_t>(k) * words] : 0u;
      }
      F part[U];
      float pc[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        part[u].zero();
        pc[u] = 0.f;
        if ((w[u] >> (r & 31u)) & 1u) {
          const size_t idx = pbase + static_cast<size_t>(k0 + u * KQ) * V;
          if (has_emb) part[u].add_from(psum + idx * D, D, lane_g);
          pc[u] = pcnt[idx];
          any = 1;
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        frag_add(acc, part[u]);
        cnt += pc[u];
      }
    }
  }
#pragma unroll
  for (int o = G; o < LR; o <<= 1) {                  // the KQ partial sums of the row, fixed butterfly (a + b == b + a bit for bit)
#pragma unroll
    for (int q = 0; q < static_cast<int>(sizeof(acc.a) / sizeof(float)); ++q) acc.a[q] += __shfl_xor(acc.a[q], o, 64);
    cnt += __shfl_xor(cnt, o, 64);
    any |= __shfl_xor(any, o, 64);
  }
  if (kq != 0) return;
  if (static_cast<int>(r) == tb.pad) any = 0;         // nn.Embedding(padding_idx): that row's gradient stays zero
  if (tb.grad != nullptr) {
    F out;
    out.zero();
    if (any) {
      out = acc;
#pragma unroll
      for (int q = 0; q < static_cast<int>(sizeof(out.a) / sizeof(float)); ++q) out.a[q] -= cnt * wrow.a[q];
    }
    float* dst = tb.grad + static_cast<size_t>(r) * D;
    if (accumulate) out.accumulate_into(dst, D, lane_g); else out.store(dst, D, lane_g);
  }
  if (tb.grad2 != nullptr && lane_g == 0) {
    const float v = any ? cnt : 0.f;
    if (accumulate) tb.grad2[r] += v; else tb.grad2[r] = v;
  }
}

// ---- host-side description of the tier-A part of one fused FM call -------------------------------------------------
struct TaPlan {
  int n_cid = 0;               // categorical fields with a gradient: rows of the compact id matrix
  CidPack cid;
  bool field_fast = false;     // how compact_ids_kernel walks a tile
  int cid_ts = 256;            // samples per tile of compact_ids_kernel
  int n_fld = 0, n_tab = 0;
  TaFieldPack fld;
  TaTablePack tab;
  unsigned NB = 0;             // blocks of 2048 samples
  unsigned rows = 0;           // rows of the tier-A tables
  unsigned frows = 0;          // rows summed over the tier-A FIELDS (shared tables count once per field)
  unsigned fwords = 0;         // bitmap words per block, summed over the fields
  int D = 1;
  bool has_emb = false, vec = false;
  size_t off_cid = 0, off_sorted = 0, off_bitmap = 0, off_psum = 0, off_pcnt = 0, bytes = 0;   // relative to the region
};

static inline size_t ta_align(size_t v) { return (v + 255) / 256 * 256; }

static inline void ta_layout(TaPlan* t, int64_t B) {
  size_t o = 0;
  t->off_cid = o; o += ta_align(static_cast<size_t>(t->n_cid) * static_cast<size_t>(B) * 4);
  const size_t units = static_cast<size_t>(t->n_fld) * t->NB;
  t->off_sorted = o; o += ta_align(units * kTaBlock * 4);
  t->off_bitmap = o; o += ta_align(static_cast<size_t>(t->fwords) * t->NB * 4);
  t->off_psum = o; o += ta_align(t->has_emb ? static_cast<size_t>(t->frows) * t->NB * t->D * 4 : 0);
  t->off_pcnt = o; o += ta_align(static_cast<size_t>(t->frows) * t->NB * 4);
  t->bytes = o;
}

static inline int ta_launch_compact(const TaPlan& t, int64_t B, char* region, int* d_status, hipStream_t s) {
  if (t.n_cid == 0) return RBX_OK;
  const int ts = t.cid_ts;
  const size_t lds = RBX_MAX_FIELDS * sizeof(CidField) + static_cast<size_t>(t.n_cid) * (ts + 1) * 4;
  const unsigned blocks = static_cast<unsigned>((B + ts - 1) / ts);
  int* cid = reinterpret_cast<int*>(region + t.off_cid);
  if (t.field_fast)
    hipLaunchKernelGGL(compact_ids_kernel<true>, dim3(blocks), dim3(256), lds, s, t.cid, t.n_cid, static_cast<long long>(B),
                       ts, cid, d_status);
  else
    hipLaunchKernelGGL(compact_ids_kernel<false>, dim3(blocks), dim3(256), lds, s, t.cid, t.n_cid, static_cast<long long>(B),
                       ts, cid, d_status);
  return check_launch("compact_ids_kernel");
}

static inline int ta_launch_blocksort(const TaPlan& t, int64_t B, char* region, hipStream_t s) {
  if (t.n_fld == 0) return RBX_OK;
  hipLaunchKernelGGL(ta_blocksort_kernel<0>, dim3(t.n_fld * t.NB), dim3(256), 0, s, t.fld, t.n_fld, static_cast<long long>(B),
                     reinterpret_cast<const int*>(region + t.off_cid), reinterpret_cast<unsigned*>(region + t.off_sorted),
                     reinterpret_cast<unsigned*>(region + t.off_bitmap), t.NB);
  return check_launch("ta_blocksort_kernel");
}

static inline bool ta_use_lds() { return true; }     // (the gather form for every shape was the A/B arm: profiles/r03)

template <int G, bool VEC>
static int ta_launch_bwd(const TaPlan& t, int64_t B, const float* g, const float* ssum, int accumulate, char* region,
                         hipStream_t s, const TaNumPack* num, float* num_partial, bool* num_done) {
  float* psum = reinterpret_cast<float*>(region + t.off_psum);
  float* pcnt = reinterpret_cast<float*>(region + t.off_pcnt);
  const unsigned* sorted = reinterpret_cast<const unsigned*>(region + t.off_sorted);
  int rc;
  if constexpr (VEC && G <= 4) {
    if (t.has_emb && ssum != nullptr && t.D <= kTaLdsMaxDim && kTaChunk == 16 && ta_use_lds()) {
      // fields per workgroup: about one workgroup per CU
      int fpg = static_cast<int>((static_cast<long long>(t.n_fld) * t.NB + kCUs - 1) / kCUs);
      if (fpg < 1) fpg = 1;
      if (fpg > 4) fpg = 4;
      const int n_grp = (t.n_fld + fpg - 1) / fpg;
      const size_t lds = (static_cast<size_t>(kTaBlock) * t.D + kTaBlock + static_cast<size_t>(2 * kTaChunks) * t.D +
                          4 * kTaChunks) * 4;
      static bool attr_set[8] = {false, false, false, false, false, false, false, false};
      if (!attr_set[G]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ta_reduce_lds_kernel<G>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set[G] = true;
      }
      TaNumPack none;
      none.n = 0;
      none.reserved = 0;
      size_t lds_num = lds;
      const bool with_num = num != nullptr && num->n > 0 && num->n <= kTaNumMax && num_partial != nullptr;
      unsigned n_numblk = 0;
      if (with_num) {
        n_numblk = static_cast<unsigned>((B + kTaNumBlock - 1) / kTaNumBlock);
        const size_t need = ta_num_lds_bytes(num->n, t.D);
        if (need > lds_num) lds_num = need;          // (<= 100 KB; only a small dim with many numeric features gets here)
      }
      hipLaunchKernelGGL((ta_reduce_lds_kernel<G>), dim3(t.NB * n_grp + n_numblk), dim3(512), lds_num, s, t.fld,
                         t.n_fld, fpg, n_grp, t.NB, static_cast<long long>(B), g, ssum, t.D, sorted, psum, pcnt,
                         with_num ? *num : none, num_partial);
      if (with_num && num_done != nullptr) *num_done = true;
      rc = check_launch("ta_reduce_lds_kernel");
      if (rc != RBX_OK) return rc;
      goto combine;
    }
  }
  {
    const size_t lds = (static_cast<size_t>(2 * kTaChunks) * t.D + 4 * kTaChunks) * 4;
    hipLaunchKernelGGL((ta_reduce_kernel<G, VEC>), dim3(t.n_fld * t.NB), dim3(kTaThreads), lds, s, t.fld, t.n_fld, t.NB, g,
                       t.has_emb ? ssum : nullptr, t.D, sorted, psum, pcnt);
    rc = check_launch("ta_reduce_kernel");
    if (rc != RBX_OK) return rc;
  }
combine:
  constexpr int NR = 256 / (G >= 16 ? G : 16);    // rows per workgroup of ta_final_kernel
  const unsigned blocks = (t.rows + NR - 1) / NR;
  TaFieldRefPack refs;
  for (int i = 0; i < t.n_fld; ++i) refs.f[i] = {t.fld.f[i].frow0, t.fld.f[i].fword0};
  hipLaunchKernelGGL((ta_final_kernel<G, VEC>), dim3(blocks), dim3(256), 0, s, t.tab, refs, t.n_tab, t.n_fld, t.NB, t.D,
                     t.rows, psum, pcnt, reinterpret_cast<const unsigned*>(region + t.off_bitmap), accumulate, t.has_emb);
  return check_launch("ta_final_kernel");
}

// num / num_partial / num_done: the fused FM backward's numeric reductions as extra workgroups of the LDS form's launch
// (ta_numeric_block); *num_done says whether they were taken (the caller otherwise launches fm_numeric_partial_kernel).
static inline int ta_dispatch_bwd(const TaPlan& t, int64_t B, const float* g, const float* ssum, int accumulate,
                                  char* region, hipStream_t s, const TaNumPack* num = nullptr, float* num_partial = nullptr,
                                  bool* num_done = nullptr) {
  if (t.n_fld == 0) return RBX_OK;
  const bool vec = t.vec && (ssum == nullptr || (reinterpret_cast<uintptr_t>(ssum) & 15) == 0);
  const int units = vec ? t.D / 4 : t.D;
  if (vec) {
    switch (pow2_ceil(units)) {
      case 1: return ta_launch_bwd<1, true>(t, B, g, ssum, accumulate, region, s, num, num_partial, num_done);
      case 2: return ta_launch_bwd<2, true>(t, B, g, ssum, accumulate, region, s, num, num_partial, num_done);
      case 4: return ta_launch_bwd<4, true>(t, B, g, ssum, accumulate, region, s, num, num_partial, num_done);
      case 8: return ta_launch_bwd<8, true>(t, B, g, ssum, accumulate, region, s, num, num_partial, num_done);
      default: return ta_launch_bwd<16, true>(t, B, g, ssum, accumulate, region, s, num, num_partial, num_done);
    }
  }
  switch (pow2_ceil(units)) {
    case 1: return ta_launch_bwd<1, false>(t, B, g, ssum, accumulate, region, s, num, num_partial, num_done);
    case 2: return ta_launch_bwd<2, false>(t, B, g, ssum, accumulate, region, s, num, num_partial, num_done);
    case 4: return ta_launch_bwd<4, false>(t, B, g, ssum, accumulate, region, s, num, num_partial, num_done);
    case 8: return ta_launch_bwd<8, false>(t, B, g, ssum, accumulate, region, s, num, num_partial, num_done);
    case 16: return ta_launch_bwd<16, false>(t, B, g, ssum, accumulate, region, s, num, num_partial, num_done);
    case 32: return ta_launch_bwd<32, false>(t, B, g, ssum, accumulate, region, s, num, num_partial, num_done);
    default: return ta_launch_bwd<64, false>(t, B, g, ssum, accumulate, region, s, num, num_partial, num_done);
  }
}

}  // namespace rbx
