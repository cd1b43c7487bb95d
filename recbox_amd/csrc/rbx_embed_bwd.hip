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
#include "rbx_internal.h"

namespace rbx {

constexpr int kSortThreads = 256;
constexpr int kSortItems = 8;                              // per thread
constexpr int kSortTile = kSortThreads * kSortItems;       // 2048 pairs per workgroup
constexpr int kRadix = 256;
constexpr int kChunk = 64;                                 // sorted pairs per lane group in the reduce
constexpr unsigned kLocalBits = 26;                        // val = slot << 26 | (b*L + l)
constexpr unsigned kLocalMask = (1u << kLocalBits) - 1u;

struct KeyField {            // 48 B
  const void* ids;
  long long stride_b;
  int stride_l;
  int vocab;
  int mask_id;
  int pad_id;
  unsigned row_base;
  unsigned lk_off;           // first lookup index of this field
  short seq_len;
  unsigned char dtype, pool;
  int reserved;
};
struct KeyPack { KeyField f[RBX_MAX_FIELDS]; };

struct RedField {            // 24 B
  float* grad;
  unsigned row_base;
  int out_off;
  short dim;
  short seq_len;
  unsigned char pool, slot;
  short reserved;
};
struct RedPack { RedField f[RBX_MAX_FIELDS]; };

struct NumField {            // numeric features: grad[d] += sum_b x_b * dY[b, off+d]
  const void* ids;
  float* grad;
  long long stride_b;
  int out_off;
  short dim;
  unsigned char dtype, reserved;
};
struct NumPack { NumField f[RBX_MAX_FIELDS]; };

// ---- host-side plan: everything derived from the descriptor array -------------
struct BwdPlan {
  int n_cat = 0, n_num = 0;
  KeyPack keys;
  RedPack red;
  NumPack num;
  unsigned n_lookups = 0;      // pairs to sort
  unsigned total_rows = 0;     // sentinel key
  int passes = 0;
  int max_dim = 1;
  bool vec = true;
  // workspace layout (byte offsets)
  size_t off_keys[2], off_vals[2], off_hist, off_head, off_tail, off_flags, off_num, bytes;
  unsigned n_tiles = 0, n_chunks = 0, num_blocks = 0;
};

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static int make_plan(const rbx_field_t* fields, int n, int64_t B, const float* dout, int64_t stride_b, BwdPlan* p) {
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
    rf.row_base = seen_base[hit];
    rf.out_off = static_cast<int>(f.out_off);
    rf.dim = static_cast<short>(f.dim);
    rf.seq_len = static_cast<short>(f.seq_len);
    rf.pool = static_cast<unsigned char>(f.pool);
    rf.slot = static_cast<unsigned char>(i);
    rf.reserved = 0;
    lookups += n_lk;
  }
  if (lookups >= (1ull << 31) || rows >= (1ull << 31))
    return fail(RBX_ERR_UNSUPPORTED, "too many lookups/rows for one call (%llu / %llu)", lookups, rows);
  p->n_lookups = static_cast<unsigned>(lookups);
  p->total_rows = static_cast<unsigned>(rows);
  int bits = 1;
  while ((1ull << bits) <= rows) ++bits;           // the sentinel key == rows must be representable
  p->passes = (bits + 7) / 8;
  p->n_tiles = (p->n_lookups + kSortTile - 1) / kSortTile;
  p->n_chunks = (p->n_lookups + kChunk - 1) / kChunk;
  p->num_blocks = static_cast<unsigned>((B + 1023) / 1024);
  size_t o = 0;
  const size_t nl = align_up(static_cast<size_t>(p->n_lookups) * 4, 256);
  p->off_keys[0] = o; o += nl;
  p->off_keys[1] = o; o += nl;
  p->off_vals[0] = o; o += nl;
  p->off_vals[1] = o; o += nl;
  p->off_hist = o; o += align_up(static_cast<size_t>(p->n_tiles) * kRadix * 4 + 4, 256);
  p->off_head = o; o += align_up(static_cast<size_t>(p->n_chunks) * p->max_dim * 4, 256);
  p->off_tail = o; o += align_up(static_cast<size_t>(p->n_chunks) * p->max_dim * 4, 256);
  p->off_flags = o; o += align_up(static_cast<size_t>(p->n_chunks) * 4, 256);
  p->off_num = o; o += align_up(static_cast<size_t>(p->n_num) * p->num_blocks * p->max_dim * 4, 256);
  p->bytes = o + 256;
  return RBX_OK;
}

// ---- build_keys ----------------------------------------------------------------
__global__ __launch_bounds__(256) void build_keys_kernel(const KeyPack P, const int n_cat, const unsigned n_lookups,
                                                         const unsigned sentinel, unsigned* __restrict__ keys,
                                                         unsigned* __restrict__ vals, int* __restrict__ status) {
  __shared__ KeyField sf[RBX_MAX_FIELDS];
  {
    const int words = n_cat * static_cast<int>(sizeof(KeyField) / 4);
    const int* src = reinterpret_cast<const int*>(&P);
    int* dst = reinterpret_cast<int*>(sf);
    for (int i = threadIdx.x; i < words; i += blockDim.x) dst[i] = src[i];
  }
  __syncthreads();
  const unsigned stride = gridDim.x * blockDim.x;
  for (unsigned j = blockIdx.x * blockDim.x + threadIdx.x; j < n_lookups; j += stride) {
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
  }
}

// ---- radix sort: per-tile digit histogram ----------------------------------------
__global__ __launch_bounds__(kSortThreads) void radix_hist_kernel(const unsigned* __restrict__ keys, const unsigned n,
                                                                  const int shift, unsigned* __restrict__ hist,
                                                                  const unsigned n_tiles) {
  __shared__ unsigned cnt[kRadix];
  cnt[threadIdx.x] = 0;
  __syncthreads();
  const unsigned base = blockIdx.x * kSortTile;
#pragma unroll
  for (int i = 0; i < kSortItems; ++i) {
    const unsigned j = base + i * kSortThreads + threadIdx.x;
    if (j < n) atomicAdd(&cnt[(keys[j] >> shift) & 0xFFu], 1u);
  }
  __syncthreads();
  hist[threadIdx.x * n_tiles + blockIdx.x] = cnt[threadIdx.x];   // digit-major for the scan
}

// ---- exclusive scan of hist[256 * n_tiles] by one workgroup ------------------------
__global__ __launch_bounds__(1024) void radix_scan_kernel(unsigned* __restrict__ hist, const unsigned len) {
  __shared__ unsigned wave_tot[16];
  __shared__ unsigned carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  for (unsigned base = 0; base < len; base += 1024 * 4) {
    const unsigned i0 = base + threadIdx.x * 4;
    unsigned v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = (i0 + k < len) ? hist[i0 + k] : 0u;
    const unsigned mine = v[0] + v[1] + v[2] + v[3];
    unsigned inc = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const unsigned t = __shfl_up(inc, o, 64);
      if (lane >= o) inc += t;
    }
    if (lane == 63) wave_tot[wid] = inc;
    __syncthreads();
    unsigned wbase = 0;
    for (int w = 0; w < wid; ++w) wbase += wave_tot[w];
    unsigned run = carry + wbase + inc - mine;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (i0 + k < len) hist[i0 + k] = run;
      run += v[k];
    }
    __syncthreads();
    if (threadIdx.x == 1023) carry = run;
    __syncthreads();
  }
}

// ---- stable scatter of one tile -----------------------------------------------------
// Wave w owns the contiguous quarter [w*512, (w+1)*512) of the tile and walks it in
// 64-item steps, so tile order == (wave, step, lane) and ranks respect it.
__global__ __launch_bounds__(kSortThreads) void radix_scatter_kernel(const unsigned* __restrict__ keys_in,
                                                                     const unsigned* __restrict__ vals_in,
                                                                     unsigned* __restrict__ keys_out,
                                                                     unsigned* __restrict__ vals_out, const unsigned n,
                                                                     const int shift, const unsigned* __restrict__ hist,
                                                                     const unsigned n_tiles) {
  constexpr int kWaves = kSortThreads / 64;
  constexpr int kPerWave = kSortTile / kWaves;       // 512
  constexpr int kSteps = kPerWave / 64;              // 8
  __shared__ unsigned wcnt[kWaves][kRadix];          // per-wave digit counts, then running offsets
  __shared__ unsigned dstart[kRadix];                // tile-local start of each digit
  __shared__ unsigned gbase[kRadix];                 // global start of (digit, tile)
  __shared__ unsigned skey[kSortTile];
  __shared__ unsigned sval[kSortTile];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const unsigned tile0 = blockIdx.x * kSortTile;
  for (int i = threadIdx.x; i < kWaves * kRadix; i += kSortThreads) (&wcnt[0][0])[i] = 0;
  gbase[threadIdx.x] = hist[threadIdx.x * n_tiles + blockIdx.x];
  __syncthreads();

  unsigned k[kSteps], v[kSteps];
  unsigned rank[kSteps];                             // rank inside (wave, digit)
  const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
  for (int s = 0; s < kSteps; ++s) {
    const unsigned j = tile0 + wid * kPerWave + s * 64 + lane;
    const bool ok = j < n;
    k[s] = ok ? keys_in[j] : 0xFFFFFFFFu;
    v[s] = ok ? vals_in[j] : 0u;
    const unsigned d = ok ? ((k[s] >> shift) & 0xFFu) : 0x100u;   // out-of-range lanes match nobody real
    unsigned long long peers = __ballot(ok);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
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
  // per digit: exclusive prefix over waves, then over digits (one thread per digit)
  {
    const int d = threadIdx.x;
    unsigned run = 0;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) {
      const unsigned c = wcnt[w][d];
      wcnt[w][d] = run;
      run += c;
    }
    // exclusive scan of `run` over the 256 digits
    unsigned inc = run;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const unsigned t = __shfl_up(inc, o, 64);
      if (lane >= o) inc += t;
    }
    __shared__ unsigned wtot[kWaves];
    if (lane == 63) wtot[wid] = inc;
    __syncthreads();
    unsigned wb = 0;
    for (int w = 0; w < wid; ++w) wb += wtot[w];
    dstart[d] = wb + inc - run;
  }
  __syncthreads();
#pragma unroll
  for (int s = 0; s < kSteps; ++s) {
    const unsigned j = tile0 + wid * kPerWave + s * 64 + lane;
    if (j < n) {
      const unsigned d = (k[s] >> shift) & 0xFFu;
      const unsigned pos = dstart[d] + wcnt[wid][d] + rank[s];
      skey[pos] = k[s];
      sval[pos] = v[s];
    }
  }
  __syncthreads();
  const unsigned tile_n = (n - tile0 < static_cast<unsigned>(kSortTile)) ? (n - tile0) : kSortTile;
#pragma unroll
  for (int i = 0; i < kSortItems; ++i) {
    const unsigned pos = i * kSortThreads + threadIdx.x;
    if (pos < tile_n) {
      const unsigned key = skey[pos];
      const unsigned d = (key >> shift) & 0xFFu;
      const unsigned g = gbase[d] + (pos - dstart[d]);
      keys_out[g] = key;
      vals_out[g] = sval[pos];
    }
  }
}

// ---- segment reduce ------------------------------------------------------------------
template <int G, int NV, bool VEC>
struct Frag {
  static constexpr int W = VEC ? 4 : 1;
  float a[NV * W];
  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int i = 0; i < NV * W; ++i) a[i] = 0.f;
  }
  __device__ __forceinline__ void fma_from(const float* row, int dim, int lane_g, float w) {
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const int e = (lane_g + u * G) * W;
      if (e < dim) {
        if constexpr (VEC) {
          const float4 t = *reinterpret_cast<const float4*>(row + e);
          a[u * 4 + 0] += w * t.x; a[u * 4 + 1] += w * t.y; a[u * 4 + 2] += w * t.z; a[u * 4 + 3] += w * t.w;
        } else {
          a[u] += w * row[e];
        }
      }
    }
  }
  __device__ __forceinline__ void add_from(const float* row, int dim, int lane_g) { fma_from(row, dim, lane_g, 1.0f); }
  __device__ __forceinline__ void store(float* row, int dim, int lane_g) const {
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const int e = (lane_g + u * G) * W;
      if (e < dim) {
        if constexpr (VEC) {
          *reinterpret_cast<float4*>(row + e) = make_float4(a[u * 4], a[u * 4 + 1], a[u * 4 + 2], a[u * 4 + 3]);
        } else {
          row[e] = a[u];
        }
      }
    }
  }
  // row[e] += a  (each touched row is owned by exactly one lane group per call)
  __device__ __forceinline__ void accumulate_into(float* row, int dim, int lane_g) const {
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const int e = (lane_g + u * G) * W;
      if (e < dim) {
        if constexpr (VEC) {
          float4 t = *reinterpret_cast<float4*>(row + e);
          t.x += a[u * 4]; t.y += a[u * 4 + 1]; t.z += a[u * 4 + 2]; t.w += a[u * 4 + 3];
          *reinterpret_cast<float4*>(row + e) = t;
        } else {
          row[e] += a[u];
        }
      }
    }
  }
};

constexpr int kFlagFin = 1;    // chunk finalises a run that started in an earlier chunk
constexpr int kFlagPass = 2;   // whole chunk is the middle of one run

template <int G, int NV, bool VEC>
__global__ __launch_bounds__(256) void segment_reduce_kernel(const RedPack P, const int n_cat, const long long B,
                                                             const unsigned* __restrict__ keys,
                                                             const unsigned* __restrict__ vals, const unsigned n,
                                                             const unsigned sentinel, const float* __restrict__ dout,
                                                             const long long stride_b,
                                                             const float* __restrict__ row_scale,
                                                             float* __restrict__ head, float* __restrict__ tail,
                                                             int* __restrict__ flags, const int max_dim,
                                                             const unsigned n_chunks) {
  __shared__ RedField sf[RBX_MAX_FIELDS];
  {
    const int words = n_cat * static_cast<int>(sizeof(RedField) / 4);
    const int* src = reinterpret_cast<const int*>(&P);
    int* dst = reinterpret_cast<int*>(sf);
    for (int i = threadIdx.x; i < words; i += blockDim.x) dst[i] = src[i];
  }
  __syncthreads();
  using F = Frag<G, NV, VEC>;
  const int lane_g = threadIdx.x % G;
  const unsigned c = blockIdx.x * (blockDim.x / G) + threadIdx.x / G;
  if (c >= n_chunks) return;
  const unsigned s = c * kChunk;
  const unsigned e = (s + kChunk < n) ? s + kChunk : n;
  const unsigned key_before = (s > 0) ? keys[s - 1] : sentinel;
  const unsigned key_after = (e < n) ? keys[e] : sentinel;
  unsigned cur = keys[s];
  const bool open_in = (s > 0) && (cur == key_before) && (cur != sentinel);
  bool seen_boundary = false;
  unsigned cur_val = vals[s];
  F acc;
  acc.zero();
  constexpr int U = 4;
  for (unsigned i0 = s; i0 < e; i0 += U) {
    unsigned kk[U], vv[U];
    const float* src[U];
    float w[U];
    int dims[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const unsigned i = i0 + u;
      const bool ok = i < e;
      kk[u] = ok ? keys[i] : sentinel;
      vv[u] = ok ? vals[i] : 0u;
      const RedField& fd = sf[vv[u] >> kLocalBits];
      const unsigned local = vv[u] & kLocalMask;
      const unsigned L = static_cast<unsigned>(fd.seq_len);
      const unsigned b = local / L;
      const unsigned l = local - b * L;
      dims[u] = fd.dim;
      src[u] = dout + static_cast<long long>(b) * stride_b + fd.out_off +
               (fd.pool == RBX_POOL_CONCAT ? static_cast<long long>(l) * fd.dim : 0ll);
      w[u] = 1.0f;
      if (ok && kk[u] != sentinel && (fd.pool == RBX_POOL_MEAN_VALUE || fd.pool == RBX_POOL_MEAN_ID))
        w[u] = row_scale[static_cast<long long>(fd.slot) * B + b];
    }
    F rows[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      rows[u].zero();
      if (kk[u] != sentinel) rows[u].fma_from(src[u], dims[u], lane_g, w[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (i0 + u >= e) break;
      if (kk[u] != cur) {                               // run boundary
        if (cur != sentinel) {
          const RedField& fd = sf[cur_val >> kLocalBits];
          if (!seen_boundary && open_in) {
            acc.store(head + static_cast<size_t>(c) * max_dim, fd.dim, lane_g);
          } else {
            acc.accumulate_into(fd.grad + static_cast<size_t>(cur - fd.row_base) * fd.dim, fd.dim, lane_g);
          }
        }
        seen_boundary = true;
        acc.zero();
        cur = kk[u];
        cur_val = vv[u];
      }
#pragma unroll
      for (int q = 0; q < NV * F::W; ++q) acc.a[q] += rows[u].a[q];
    }
  }
  int flag = 0;
  if (cur != sentinel) {
    const RedField& fd = sf[cur_val >> kLocalBits];
    const bool open_out = (e < n) && (key_after == cur);
    const bool is_head = !seen_boundary && open_in;
    if (open_out) {
      acc.store(tail + static_cast<size_t>(c) * max_dim, fd.dim, lane_g);
      if (is_head) flag |= kFlagPass;
    } else if (is_head) {
      acc.store(head + static_cast<size_t>(c) * max_dim, fd.dim, lane_g);
      flag |= kFlagFin;
    } else {
      acc.accumulate_into(fd.grad + static_cast<size_t>(cur - fd.row_base) * fd.dim, fd.dim, lane_g);
    }
  }
  if (seen_boundary && open_in) flag |= kFlagFin;
  if (lane_g == 0) flags[c] = flag;
}

template <int G, int NV, bool VEC>
__global__ __launch_bounds__(256) void segment_fixup_kernel(const RedPack P, const int n_cat,
                                                            const unsigned* __restrict__ keys,
                                                            const unsigned* __restrict__ vals,
                                                            const float* __restrict__ head,
                                                            const float* __restrict__ tail,
                                                            const int* __restrict__ flags, const int max_dim,
                                                            const unsigned n_chunks) {
  using F = Frag<G, NV, VEC>;
  const int lane_g = threadIdx.x % G;
  const unsigned c = blockIdx.x * (blockDim.x / G) + threadIdx.x / G;
  if (c >= n_chunks) return;
  if (!(flags[c] & kFlagFin)) return;
  const unsigned s = c * kChunk;
  const unsigned key = keys[s];
  const RedField fd = P.f[vals[s] >> kLocalBits];
  F acc;
  acc.zero();
  acc.add_from(head + static_cast<size_t>(c) * max_dim, fd.dim, lane_g);
  for (unsigned j = c; j-- > 0;) {
    acc.add_from(tail + static_cast<size_t>(j) * max_dim, fd.dim, lane_g);
    if (!(flags[j] & kFlagPass)) break;
  }
  acc.accumulate_into(fd.grad + static_cast<size_t>(key - fd.row_base) * fd.dim, fd.dim, lane_g);
}

// ---- numeric features: grad[d] += sum_b x_b * dY[b, off + d] -----------------------------
// grid (num_blocks, n_num); block: 256 threads = 16 sample lanes x 16 dim lanes per step.
__global__ __launch_bounds__(256) void numeric_partial_kernel(const NumPack P, const long long B,
                                                              const float* __restrict__ dout, const long long stride_b,
                                                              float* __restrict__ partial, const int max_dim,
                                                              const unsigned num_blocks) {
  const NumField fd = P.f[blockIdx.y];
  const int dim = fd.dim;
  __shared__ float red[256];
  const long long b0 = static_cast<long long>(blockIdx.x) * 1024;
  const long long b1 = (b0 + 1024 < B) ? b0 + 1024 : B;
  for (int dbase = 0; dbase < dim; dbase += 16) {
    const int d = dbase + (threadIdx.x & 15);
    float acc = 0.f;
    if (d < dim) {
      for (long long b = b0 + (threadIdx.x >> 4); b < b1; b += 16) {
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

__global__ void numeric_final_kernel(const NumPack P, const float* __restrict__ partial, const int max_dim,
                                     const unsigned num_blocks) {
  const NumField fd = P.f[blockIdx.x];
  for (int d = threadIdx.x; d < fd.dim; d += blockDim.x) {
    float t = 0.f;
    for (unsigned k = 0; k < num_blocks; ++k) t += partial[(static_cast<size_t>(blockIdx.x) * num_blocks + k) * max_dim + d];
    fd.grad[d] += t;
  }
}

template <int G, int NV, bool VEC>
static int launch_reduce(const BwdPlan& p, int64_t B, const unsigned* keys, const unsigned* vals, const float* dout,
                         int64_t stride_b, const float* row_scale, char* ws, hipStream_t s) {
  const int groups_per_block = 256 / G;
  const unsigned blocks = (p.n_chunks + groups_per_block - 1) / groups_per_block;
  float* head = reinterpret_cast<float*>(ws + p.off_head);
  float* tail = reinterpret_cast<float*>(ws + p.off_tail);
  int* flags = reinterpret_cast<int*>(ws + p.off_flags);
  hipLaunchKernelGGL((segment_reduce_kernel<G, NV, VEC>), dim3(blocks), dim3(256), 0, s, p.red, p.n_cat,
                     static_cast<long long>(B), keys, vals, p.n_lookups, p.total_rows, dout,
                     static_cast<long long>(stride_b), row_scale, head, tail, flags, p.max_dim, p.n_chunks);
  int rc = check_launch("segment_reduce_kernel");
  if (rc != RBX_OK) return rc;
  hipLaunchKernelGGL((segment_fixup_kernel<G, NV, VEC>), dim3(blocks), dim3(256), 0, s, p.red, p.n_cat, keys, vals, head,
                     tail, flags, p.max_dim, p.n_chunks);
  return check_launch("segment_fixup_kernel");
}

template <bool VEC>
static int dispatch_reduce(const BwdPlan& p, int64_t B, const unsigned* keys, const unsigned* vals, const float* dout,
                           int64_t stride_b, const float* row_scale, char* ws, hipStream_t s) {
  const int units = VEC ? p.max_dim / 4 : p.max_dim;
  switch (pow2_ceil(units)) {
    case 1: return launch_reduce<1, 1, VEC>(p, B, keys, vals, dout, stride_b, row_scale, ws, s);
    case 2: return launch_reduce<2, 1, VEC>(p, B, keys, vals, dout, stride_b, row_scale, ws, s);
    case 4: return launch_reduce<4, 1, VEC>(p, B, keys, vals, dout, stride_b, row_scale, ws, s);
    case 8: return launch_reduce<8, 1, VEC>(p, B, keys, vals, dout, stride_b, row_scale, ws, s);
    case 16: return launch_reduce<16, 1, VEC>(p, B, keys, vals, dout, stride_b, row_scale, ws, s);
    case 32: return launch_reduce<32, 1, VEC>(p, B, keys, vals, dout, stride_b, row_scale, ws, s);
    case 64: return launch_reduce<64, 1, VEC>(p, B, keys, vals, dout, stride_b, row_scale, ws, s);
    case 128: return launch_reduce<64, 2, VEC>(p, B, keys, vals, dout, stride_b, row_scale, ws, s);
    case 256: return launch_reduce<64, 4, VEC>(p, B, keys, vals, dout, stride_b, row_scale, ws, s);
    default: return fail(RBX_ERR_UNSUPPORTED, "embedding dim too large for one lane group");
  }
}

}  // namespace rbx

extern "C" size_t rbx_embed_bwd_workspace_size(const rbx_field_t* fields, int32_t n_fields, int64_t batch) {
  rbx::BwdPlan p;
  if (rbx::make_plan(fields, n_fields, batch, nullptr, 0, &p) != RBX_OK) return 0;
  return p.bytes;
}

extern "C" int rbx_embed_sort(const rbx_field_t* fields, int32_t n_fields, int64_t batch, void* d_workspace,
                              size_t workspace_bytes, int32_t* d_status, void* stream) {
  using namespace rbx;
  BwdPlan p;
  int rc = make_plan(fields, n_fields, batch, nullptr, 0, &p);
  if (rc != RBX_OK) return rc;
  if (p.n_lookups == 0) return RBX_OK;
  if (d_workspace == nullptr || workspace_bytes < p.bytes)
    return fail(RBX_ERR_WORKSPACE, "workspace %zu B < required %zu B", workspace_bytes, p.bytes);
  char* ws = static_cast<char*>(d_workspace);
  hipStream_t s = as_stream(stream);
  unsigned* keys[2] = {reinterpret_cast<unsigned*>(ws + p.off_keys[0]), reinterpret_cast<unsigned*>(ws + p.off_keys[1])};
  unsigned* vals[2] = {reinterpret_cast<unsigned*>(ws + p.off_vals[0]), reinterpret_cast<unsigned*>(ws + p.off_vals[1])};
  unsigned* hist = reinterpret_cast<unsigned*>(ws + p.off_hist);
  {
    unsigned blocks = (p.n_lookups + 255) / 256;
    if (blocks > static_cast<unsigned>(kCUs * 8)) blocks = kCUs * 8;
    hipLaunchKernelGGL(build_keys_kernel, dim3(blocks), dim3(256), 0, s, p.keys, p.n_cat, p.n_lookups, p.total_rows,
                       keys[0], vals[0], d_status);
    rc = check_launch("build_keys_kernel");
    if (rc != RBX_OK) return rc;
  }
  int cur = 0;
  for (int pass = 0; pass < p.passes; ++pass) {
    const int shift = pass * 8;
    hipLaunchKernelGGL(radix_hist_kernel, dim3(p.n_tiles), dim3(kSortThreads), 0, s, keys[cur], p.n_lookups, shift, hist,
                       p.n_tiles);
    hipLaunchKernelGGL(radix_scan_kernel, dim3(1), dim3(1024), 0, s, hist, p.n_tiles * kRadix);
    hipLaunchKernelGGL(radix_scatter_kernel, dim3(p.n_tiles), dim3(kSortThreads), 0, s, keys[cur], vals[cur],
                       keys[cur ^ 1], vals[cur ^ 1], p.n_lookups, shift, hist, p.n_tiles);
    rc = check_launch("radix pass");
    if (rc != RBX_OK) return rc;
    cur ^= 1;
  }
  return RBX_OK;   // sorted pairs live in buffer (passes & 1)
}

extern "C" int rbx_embed_bwd(const rbx_field_t* fields, int32_t n_fields, int64_t batch, const float* d_dout,
                             int64_t out_stride_b, const float* d_row_scale, void* d_workspace, size_t workspace_bytes,
                             void* stream) {
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
    rc = p.vec ? dispatch_reduce<true>(p, batch, keys, vals, d_dout, out_stride_b, d_row_scale, ws, s)
               : dispatch_reduce<false>(p, batch, keys, vals, d_dout, out_stride_b, d_row_scale, ws, s);
    if (rc != RBX_OK) return rc;
  }
  if (p.n_num > 0) {
    float* partial = reinterpret_cast<float*>(ws + p.off_num);
    hipLaunchKernelGGL(numeric_partial_kernel, dim3(p.num_blocks, p.n_num), dim3(256), 0, s, p.num,
                       static_cast<long long>(batch), d_dout, static_cast<long long>(out_stride_b), partial, p.max_dim,
                       p.num_blocks);
    hipLaunchKernelGGL(numeric_final_kernel, dim3(p.n_num), dim3(64), 0, s, p.num, partial, p.max_dim, p.num_blocks);
    rc = check_launch("numeric grad kernels");
    if (rc != RBX_OK) return rc;
  }
  return RBX_OK;
}
