// rbx_shard.hip -- C1: one exchange each way for row-sharded embedding tables (gfx950).
//
// The reference has no model-parallel embedding (single torch.device, nn.DataParallel / DDP only in vendored
// trainers: SURVEY.md 2.1, 8e); this is the build's own scaling path for tables that should not be replicated
// (BASELINE.json configs 3 and 4): owner(id) = id mod W, local row = base[owner][table] + id div W.
//
// Per step and rank there is ONE all-to-all of int32 requests, ONE all-to-all of fp32 rows back, and ONE
// all-to-all of fp32 gradient rows in the backward (the collectives themselves are issued by the host side,
// recbox_amd/sharded.py, over RCCL).  A sample asks for
//   T single-row lookups  (one-hot features, and the columns of a pooling='concat' sequence: YoutubeDNN's positive
//                          and negative items, youtube_dnn.py:57-70; DeepFM's large categorical fields), and
//   P <= 1 pooled lookup  (a padded id sequence that the model mean-/sum-pools with an id mask: rechub's
//                          AveragePooling / SumPooling behind InputMask, third_party/rechub/basic/layers.py:135-148,
//                          187-210 -- YoutubeDNN's history, youtube_dnn.py:46-56).
// The pooled lookup is REDUCED AT THE OWNER: an owner returns one partial sum per (source rank, sample) instead of
// the raw rows (4 MB instead of ~16 MB per GPU pair at cfg 3), the requester adds the W partial
// sums and scales by 1 / (count + eps).
//
// Wire format, per (requester -> owner) pair, static sizes (no host sync, graph-capturable):
//   int32 chunk  [ cap_pool row numbers, grouped by sample | P * (B + 1) offsets into them | cap_rows row numbers ]
//                (row -1 = empty single-row slot; pooled rows beyond offsets[B] are not read)
//   fp32 chunk   [ P * B partial sums | cap_rows rows ] x D     (forward: owner -> requester)
//   fp32 chunk   [ P * B upstream gradients of the pooled output, already scaled by 1/(count+eps) | cap_rows
//                  gradient rows ] x D                          (backward: requester -> owner)
// Slots inside an owner's block are assigned by a stable counting sort over the lookups in (sample, position)
// order: deterministic, and the pooled rows of one sample are contiguous.  Lookups that do not fit raise the
// overflow byte (never dropped silently); ids outside [0, vocab) raise the status word (the reference raises
// IndexError) and are treated as absent.
//
// Owner-side backward: rbx_shard_serve leaves a flat key array (local row of every received lookup, -1 = none) and
// for each key the fp32 row of the gradient buffer that holds its upstream gradient; rbx_embed_sort +
// rbx_embed_bwd_indexed then run the same sorted, segmented, deterministic scatter-add as everywhere else (K3).
#include "rbx_internal.h"

namespace rbx {

constexpr int kShardTile = 2048;      // lookups per workgroup: 8 rounds of 256
constexpr int kShardMaxW = 64;

struct ShardCol {            // ids of one lookup column: a strided, typed column read in place
  const void* ids;
  long long stride_b;
  long long stride_l;
  int vocab;
  int mask_id;               // pooled lookup: ids equal to it are padding (kNoId: none)
  int dtype;
  int pad;
};
struct ShardCols { ShardCol f[RBX_MAX_FIELDS]; };

struct ShardOffs { long long off[RBX_MAX_FIELDS + 1]; };      // float offset of each lookup's slot in an output row

__device__ __forceinline__ int shard_owner(long long id, int W) { return static_cast<int>(id % W); }

// lookup i -> (owner or -1, id).  POOL: i = b * L + l over ONE column descriptor; rows: i = b * T + t over T columns.
template <bool POOL>
__device__ __forceinline__ int shard_lookup(const ShardCols& C, long long i, int per_sample, int W, long long* id_out,
                                            int* t_out, int* status) {
  const long long b = i / per_sample;
  const int k = static_cast<int>(i - b * per_sample);
  const ShardCol& c = C.f[POOL ? 0 : k];
  const long long id = load_id(c.ids, b * c.stride_b + (POOL ? k * c.stride_l : 0), c.dtype);
  *id_out = id;
  *t_out = POOL ? 0 : k;
  if (POOL && c.mask_id != kNoId && id == c.mask_id) return -1;           // padding position: no lookup
  if (id < 0 || id >= c.vocab) {
    if (status != nullptr) atomicOr(status, 1);
    return -1;
  }
  return shard_owner(id, W);
}

template <bool POOL>
__global__ __launch_bounds__(256) void shard_count_kernel(const ShardCols C, const int per_sample, const long long n,
                                                          const int W, int* __restrict__ hist /*[W][tiles]*/,
                                                          const int tiles, int* __restrict__ cnt /*[B], POOL*/,
                                                          int* __restrict__ status) {
  __shared__ int s_cnt[kShardMaxW];
  if (threadIdx.x < W) s_cnt[threadIdx.x] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const long long first = static_cast<long long>(blockIdx.x) * kShardTile;
#pragma unroll
  for (int j = 0; j < kShardTile / 256; ++j) {
    const long long i = first + j * 256 + threadIdx.x;
    long long id;
    int t;
    const int own = (i < n) ? shard_lookup<POOL>(C, i, per_sample, W, &id, &t, status) : -1;
    if (own >= 0) atomicAdd(&s_cnt[own], 1);                               // integer LDS atomics: order-independent
    if (POOL) {
      // valid ids of a sample: the lanes of one sample are consecutive, the first of them inside this wavefront
      // adds the wavefront's share (integer atomics: exact)
      const unsigned long long m = __ballot(own >= 0);
      if (i < n) {
        const long long b = i / per_sample;
        const long long wave0 = i - lane;
        long long lo = b * per_sample - wave0, hi = (b + 1) * per_sample - 1 - wave0;
        if (lo < 0) lo = 0;
        if (hi > 63) hi = 63;
        if (lane == lo) {
          const unsigned long long span = ((hi >= 63) ? ~0ull : ((1ull << (hi + 1)) - 1ull)) & ~((1ull << lo) - 1ull);
          const int c = __popcll(m & span);
          if (c > 0) atomicAdd(&cnt[b], c);
        }
      }
    }
  }
  __syncthreads();
  if (threadIdx.x < W) hist[static_cast<long long>(threadIdx.x) * tiles + blockIdx.x] = s_cnt[threadIdx.x];
}

// one workgroup per owner: exclusive scan of its tile counts (in place); total vs capacity; POOL: offsets[owner][B]
__global__ __launch_bounds__(256) void shard_scan_kernel(int* __restrict__ hist, const int tiles, const long long capacity,
                                                         unsigned char* __restrict__ overflow, int* __restrict__ send,
                                                         const long long ichunk, const long long end_off /* <0: none */) {
  __shared__ int s_wave[4];
  __shared__ int s_carry;
  int* col = hist + static_cast<long long>(blockIdx.x) * tiles;
  if (threadIdx.x == 0) s_carry = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int t0 = 0; t0 < tiles; t0 += 256) {
    const int t = t0 + threadIdx.x;
    const int v = (t < tiles) ? col[t] : 0;
    int incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int up = __shfl_up(incl, o, 64);
      if (lane >= o) incl += up;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int before = s_carry;
    for (int w = 0; w < wave; ++w) before += s_wave[w];
    if (t < tiles) col[t] = before + incl - v;
    __syncthreads();
    if (threadIdx.x == 255) s_carry = before + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    if (s_carry > capacity && overflow != nullptr) *overflow = 1;
    if (end_off >= 0) send[blockIdx.x * ichunk + end_off] = static_cast<int>(s_carry < capacity ? s_carry : capacity);
  }
}

template <bool POOL>
__global__ __launch_bounds__(256) void shard_assign_kernel(const ShardCols C, const int per_sample, const long long n,
                                                           const int W, const long long capacity,
                                                           const long long* __restrict__ base /*[W][n_base]*/,
                                                           const int n_base, const int base_col0,
                                                           const int* __restrict__ hist, const int tiles,
                                                           int* __restrict__ send, const long long ichunk,
                                                           const long long rows_off, const long long offs_off,
                                                           int* __restrict__ slot /*[B][T], rows*/,
                                                           const int* __restrict__ cnt, float* __restrict__ inv,
                                                           const int mean, const float eps) {
  __shared__ int s_run[kShardMaxW];                       // owner's lookups before the current round
  __shared__ int s_wave[4][kShardMaxW];                   // per wavefront counts of the current round
  __shared__ unsigned long long s_mask[4][kShardMaxW];    // POOL: which lanes of the wavefront go to owner w
  if (threadIdx.x < W) s_run[threadIdx.x] = hist[static_cast<long long>(threadIdx.x) * tiles + blockIdx.x];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned long long below = (1ull << lane) - 1ull;
  const long long first = static_cast<long long>(blockIdx.x) * kShardTile;
  const long long dump = capacity * W;
  for (int j = 0; j < kShardTile / 256; ++j) {
    const long long i = first + j * 256 + threadIdx.x;
    const bool valid = i < n;
    long long id = 0;
    int t = 0;
    const int own = valid ? shard_lookup<POOL>(C, i, per_sample, W, &id, &t, nullptr) : -1;
    int in_wave = 0;
    for (int w = 0; w < W; ++w) {                          // W is small (GPUs of one node)
      const unsigned long long m = __ballot(own == w);
      if (own == w) in_wave = __popcll(m & below);
      if (lane == 0) {
        s_wave[wave][w] = __popcll(m);
        if (POOL) s_mask[wave][w] = m;
      }
    }
    __syncthreads();
    if (own >= 0) {
      int rank = s_run[own] + in_wave;
      for (int v = 0; v < wave; ++v) rank += s_wave[v][own];
      const long long row = (base != nullptr ? base[static_cast<long long>(own) * n_base + base_col0 + t] : 0ll) + id / W;
      if (rank < capacity) send[own * ichunk + rows_off + rank] = static_cast<int>(row);
      if (!POOL) slot[i] = (rank < capacity) ? static_cast<int>(own * capacity + rank) : static_cast<int>(dump);
    } else if (!POOL && valid) {
      slot[i] = static_cast<int>(dump);                   // absent (out-of-range id): reads as a zero row
    }
    if (POOL && valid && (i % per_sample) == 0) {
      // first position of sample b: every owner's block has `before` lookups of earlier samples
      const long long b = i / per_sample;
      for (int w = 0; w < W; ++w) {
        int before = s_run[w] + __popcll(s_mask[wave][w] & below);
        for (int v = 0; v < wave; ++v) before += s_wave[v][w];
        send[w * ichunk + offs_off + b] = static_cast<int>(before < capacity ? before : capacity);
      }
      inv[b] = mean ? 1.0f / (static_cast<float>(cnt[b]) + eps) : 1.0f;
    }
    __syncthreads();
    if (threadIdx.x < W) s_run[threadIdx.x] += s_wave[0][threadIdx.x] + s_wave[1][threadIdx.x] + s_wave[2][threadIdx.x] +
                                               s_wave[3][threadIdx.x];
    __syncthreads();
  }
}

// ---- geometry ---------------------------------------------------------------------------------------
struct Geom {
  int W, D, T, P;
  long long B, cap_rows, cap_pool;
  long long ichunk;          // int32 per (requester, owner) pair
  long long off_offs;        // start of the offsets inside an int32 chunk
  long long off_rows;        // start of the single-row numbers inside an int32 chunk
  long long frows;           // fp32 rows per (requester, owner) pair: P * B + cap_rows
};

static int make_geom(const rbx_shard_geom_t* g, Geom* out) {
  if (g == nullptr) return fail(RBX_ERR_INVALID, "shard: geometry is NULL");
  if (g->world <= 0 || g->world > kShardMaxW) return fail(RBX_ERR_UNSUPPORTED, "shard: world=%d not in [1,%d]", g->world, kShardMaxW);
  if (g->dim <= 0 || g->dim % 4 != 0 || g->dim > 1024) return fail(RBX_ERR_UNSUPPORTED, "shard: dim=%d must be a multiple of 4 (<= 1024)", g->dim);
  if (g->n_rows < 0 || g->n_rows > RBX_MAX_FIELDS || g->n_pool < 0 || g->n_pool > 1 || g->n_rows + g->n_pool == 0)
    return fail(RBX_ERR_UNSUPPORTED, "shard: %d single-row + %d pooled lookups per sample", g->n_rows, g->n_pool);
  if (g->batch < 0 || g->cap_rows < 0 || g->cap_pool < 0) return fail(RBX_ERR_INVALID, "shard: negative size");
  if ((g->n_rows > 0) != (g->cap_rows > 0) || (g->n_pool > 0) != (g->cap_pool > 0))
    return fail(RBX_ERR_INVALID, "shard: capacity must be positive exactly for the lookup kinds in use");
  out->W = g->world; out->D = g->dim; out->T = g->n_rows; out->P = g->n_pool;
  out->B = g->batch; out->cap_rows = g->cap_rows; out->cap_pool = g->cap_pool;
  out->off_offs = g->cap_pool;
  out->off_rows = g->cap_pool + static_cast<long long>(g->n_pool) * (g->batch + 1);
  out->ichunk = (out->off_rows + g->cap_rows + 3) / 4 * 4;
  out->frows = static_cast<long long>(g->n_pool) * g->batch + g->cap_rows;
  if (out->ichunk * g->world >= INT_MAX || out->frows * g->world >= INT_MAX || (g->cap_pool + g->cap_rows) * g->world >= INT_MAX)
    return fail(RBX_ERR_UNSUPPORTED, "shard: exchange too large for 32-bit slot numbers");
  return RBX_OK;
}

static long long shard_tiles(long long n) { return (n + kShardTile - 1) / kShardTile; }

// ---- owner: gather + pool -------------------------------------------------------------------------------
// unit u < W * P * B: pooled segment (source w, sample b); then W * cap_rows single-row slots.
// A lane group of G lanes (one float4 each, NV float4 per lane when D > 4 G) owns a unit.
template <int G, int NV>
__global__ __launch_bounds__(256) void shard_serve_kernel(const Geom g, const int* __restrict__ recv,
                                                          const float* __restrict__ weight, const long long n_local,
                                                          float* __restrict__ back, int* __restrict__ keys,
                                                          int* __restrict__ src, int* __restrict__ status) {
  const int lane_g = threadIdx.x % G;
  const long long u = static_cast<long long>(blockIdx.x) * (256 / G) + threadIdx.x / G;
  const long long n_pool_units = static_cast<long long>(g.W) * g.P * g.B;
  const long long n_units = n_pool_units + static_cast<long long>(g.W) * g.cap_rows;
  if (u >= n_units) return;
  const int D = g.D;
  if (u < n_pool_units) {
    const int w = static_cast<int>(u / g.B);
    const long long b = u - static_cast<long long>(w) * g.B;
    const int* chunk = recv + w * g.ichunk;
    int lo = chunk[g.off_offs + b], hi = chunk[g.off_offs + b + 1];
    if (lo < 0) lo = 0;
    if (hi > g.cap_pool) hi = static_cast<int>(g.cap_pool);
    float4 acc[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) acc[v] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int j0 = lo; j0 < hi; j0 += 4) {                  // 4 rows in flight, added in slot order
      int r[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) r[q] = (j0 + q < hi) ? chunk[j0 + q] : -1;
      float4 x[4][NV];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const bool ok = r[q] >= 0 && r[q] < n_local;
        if (r[q] >= n_local && status != nullptr && lane_g == 0) atomicOr(status, 2);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          const int e = (lane_g + v * G) * 4;
          x[q][v] = (ok && e < D) ? *reinterpret_cast<const float4*>(weight + static_cast<long long>(r[q]) * D + e)
                                  : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          acc[v].x += x[q][v].x; acc[v].y += x[q][v].y; acc[v].z += x[q][v].z; acc[v].w += x[q][v].w;
        }
      // keys of the backward: the lanes of the group share the (at most 4) slots of this step
      if (lane_g < 4 && j0 + lane_g < hi) {
        const int rr = chunk[j0 + lane_g];
        const long long k = static_cast<long long>(w) * g.cap_pool + j0 + lane_g;
        keys[k] = (rr >= 0 && rr < n_local) ? rr : -1;
        src[k] = static_cast<int>(w * g.frows + b);
      }
    }
    float* dst = back + (w * g.frows + b) * D;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int e = (lane_g + v * G) * 4;
      if (e < D) *reinterpret_cast<float4*>(dst + e) = acc[v];
    }
  } else {
    const long long s = u - n_pool_units;
    const int w = static_cast<int>(s / g.cap_rows);
    const long long j = s - static_cast<long long>(w) * g.cap_rows;
    const int r = recv[w * g.ichunk + g.off_rows + j];
    const bool ok = r >= 0 && r < n_local;
    if (r >= n_local && status != nullptr && lane_g == 0) atomicOr(status, 2);
    const long long k = static_cast<long long>(g.W) * g.cap_pool + s;
    const long long frow = w * g.frows + static_cast<long long>(g.P) * g.B + j;
    if (lane_g == 0) {
      keys[k] = ok ? r : -1;
      src[k] = static_cast<int>(frow);
    }
    if (ok) {
      float* dst = back + frow * D;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int e = (lane_g + v * G) * 4;
        if (e < D) *reinterpret_cast<float4*>(dst + e) = *reinterpret_cast<const float4*>(weight + static_cast<long long>(r) * D + e);
      }
    }
  }
}

// ---- requester: place the returned rows / partial sums into the layer's output rows --------------------------
// unit = (sample b, lookup f): f < T single-row lookups (row at wire slot slot[b][f]), then the pooled one.
template <int G, int NV>
__global__ __launch_bounds__(256) void shard_combine_fwd_kernel(const Geom g, const float* __restrict__ back,
                                                                const int* __restrict__ slot,
                                                                const float* __restrict__ inv, float* __restrict__ out,
                                                                const long long out_stride_b, const ShardOffs offs) {
  const int lane_g = threadIdx.x % G;
  const int F = g.T + g.P;
  const long long u = static_cast<long long>(blockIdx.x) * (256 / G) + threadIdx.x / G;
  if (u >= g.B * F) return;
  const long long b = u / F;
  const int f = static_cast<int>(u - b * F);
  const int D = g.D;
  float* dst = out + b * out_stride_b + offs.off[f];
  float4 acc[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) acc[v] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (f < g.T) {
    const long long s = slot[b * g.T + f];
    if (s < g.cap_rows * g.W) {
      const int w = static_cast<int>(s / g.cap_rows);
      const float* srcp = back + (w * g.frows + static_cast<long long>(g.P) * g.B + (s - w * g.cap_rows)) * D;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int e = (lane_g + v * G) * 4;
        if (e < D) acc[v] = *reinterpret_cast<const float4*>(srcp + e);
      }
    }
  } else {
    for (int w = 0; w < g.W; ++w) {                        // fixed owner order: deterministic
      const float* srcp = back + (w * g.frows + b) * D;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int e = (lane_g + v * G) * 4;
        if (e < D) {
          const float4 t = *reinterpret_cast<const float4*>(srcp + e);
          acc[v].x += t.x; acc[v].y += t.y; acc[v].z += t.z; acc[v].w += t.w;
        }
      }
    }
    const float sc = inv[b];
#pragma unroll
    for (int v = 0; v < NV; ++v) { acc[v].x *= sc; acc[v].y *= sc; acc[v].z *= sc; acc[v].w *= sc; }
  }
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int e = (lane_g + v * G) * 4;
    if (e < D) {                                           // output rows need not be 16-byte aligned (rechub squeeze layout)
      dst[e] = acc[v].x; dst[e + 1] = acc[v].y; dst[e + 2] = acc[v].z; dst[e + 3] = acc[v].w;
    }
  }
}

// upstream gradient of the layer's output rows -> fp32 wire chunks for the owners
template <int G, int NV>
__global__ __launch_bounds__(256) void shard_combine_bwd_kernel(const Geom g, const float* __restrict__ dout,
                                                                const long long stride_b, const ShardOffs offs,
                                                                const int* __restrict__ slot,
                                                                const float* __restrict__ inv, float* __restrict__ gsend) {
  const int lane_g = threadIdx.x % G;
  const int F = g.T + g.P;
  const long long u = static_cast<long long>(blockIdx.x) * (256 / G) + threadIdx.x / G;
  if (u >= g.B * F) return;
  const long long b = u / F;
  const int f = static_cast<int>(u - b * F);
  const int D = g.D;
  const float* srcp = dout + b * stride_b + offs.off[f];
  float4 x[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int e = (lane_g + v * G) * 4;
    x[v] = (e < D) ? make_float4(srcp[e], srcp[e + 1], srcp[e + 2], srcp[e + 3]) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  if (f < g.T) {
    const long long s = slot[b * g.T + f];
    if (s >= g.cap_rows * g.W) return;                     // dump slot: the lookup never left this rank
    const int w = static_cast<int>(s / g.cap_rows);
    float* dst = gsend + (w * g.frows + static_cast<long long>(g.P) * g.B + (s - w * g.cap_rows)) * D;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int e = (lane_g + v * G) * 4;
      if (e < D) *reinterpret_cast<float4*>(dst + e) = x[v];
    }
  } else {
    const float sc = inv[b];
#pragma unroll
    for (int v = 0; v < NV; ++v) { x[v].x *= sc; x[v].y *= sc; x[v].z *= sc; x[v].w *= sc; }
    for (int w = 0; w < g.W; ++w) {                        // every owner that holds rows of the sample needs it
      float* dst = gsend + (w * g.frows + b) * D;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int e = (lane_g + v * G) * 4;
        if (e < D) *reinterpret_cast<float4*>(dst + e) = x[v];
      }
    }
  }
}

#define RBX_SHARD_DISPATCH(KERNEL, units, s, ...)                                                              \
  do {                                                                                                         \
    const int lanes = (geo.D + 3) / 4;                                                                         \
    if (lanes <= 4) {                                                                                          \
      hipLaunchKernelGGL((KERNEL<4, 1>), dim3(static_cast<unsigned>(((units) + 63) / 64)), dim3(256), 0, s, __VA_ARGS__);  \
    } else if (lanes <= 8) {                                                                                   \
      hipLaunchKernelGGL((KERNEL<8, 1>), dim3(static_cast<unsigned>(((units) + 31) / 32)), dim3(256), 0, s, __VA_ARGS__);  \
    } else if (lanes <= 16) {                                                                                  \
      hipLaunchKernelGGL((KERNEL<16, 1>), dim3(static_cast<unsigned>(((units) + 15) / 16)), dim3(256), 0, s, __VA_ARGS__); \
    } else if (lanes <= 32) {                                                                                  \
      hipLaunchKernelGGL((KERNEL<32, 1>), dim3(static_cast<unsigned>(((units) + 7) / 8)), dim3(256), 0, s, __VA_ARGS__);   \
    } else if (lanes <= 64) {                                                                                  \
      hipLaunchKernelGGL((KERNEL<64, 1>), dim3(static_cast<unsigned>(((units) + 3) / 4)), dim3(256), 0, s, __VA_ARGS__);   \
    } else {                                                                                                   \
      hipLaunchKernelGGL((KERNEL<64, 4>), dim3(static_cast<unsigned>(((units) + 3) / 4)), dim3(256), 0, s, __VA_ARGS__);   \
    }                                                                                                          \
  } while (0)

static int pack_col(const rbx_field_t& f, int W, ShardCol* c, const char* what, int i) {
  if (f.ids_dtype < RBX_I32 || f.ids_dtype > RBX_F64) return fail(RBX_ERR_INVALID, "shard_route: %s %d: bad ids_dtype", what, i);
  if (f.vocab <= 0 || f.vocab / W + 1 >= INT_MAX) return fail(RBX_ERR_INVALID, "shard_route: %s %d: bad vocab", what, i);
  c->ids = f.ids;
  c->stride_b = f.ids_stride_b;
  c->stride_l = f.ids_stride_l;
  c->vocab = static_cast<int>(f.vocab > INT_MAX ? INT_MAX : f.vocab);
  c->mask_id = (f.mask_id == RBX_NO_ID) ? kNoId : static_cast<int>(f.mask_id);
  c->dtype = f.ids_dtype;
  c->pad = 0;
  return RBX_OK;
}

}  // namespace rbx

extern "C" size_t rbx_shard_route_workspace_size(const rbx_shard_geom_t* geom, int32_t pool_seq_len) {
  rbx::Geom g;
  if (rbx::make_geom(geom, &g) != RBX_OK) return 0;
  const long long tr = rbx::shard_tiles(g.B * g.T), tp = rbx::shard_tiles(g.B * (g.P ? pool_seq_len : 0));
  return static_cast<size_t>((tr + tp) * g.W + g.B + 64) * sizeof(int);
}

extern "C" size_t rbx_shard_int_chunk(const rbx_shard_geom_t* geom) {
  rbx::Geom g;
  return rbx::make_geom(geom, &g) == RBX_OK ? static_cast<size_t>(g.ichunk) : 0;
}

extern "C" size_t rbx_shard_float_rows(const rbx_shard_geom_t* geom) {
  rbx::Geom g;
  return rbx::make_geom(geom, &g) == RBX_OK ? static_cast<size_t>(g.frows) : 0;
}

extern "C" int rbx_shard_route(const rbx_shard_geom_t* geom, const rbx_field_t* row_fields, const rbx_field_t* pool_field,
                               const int64_t* d_base, int32_t* d_send, int32_t* d_slot, float* d_inv,
                               uint8_t* d_overflow, int32_t* d_status, void* d_workspace, size_t workspace_bytes,
                               void* stream) {
  using namespace rbx;
  Geom g;
  int rc = make_geom(geom, &g);
  if (rc != RBX_OK) return rc;
  hipStream_t s = as_stream(stream);
  if (d_send == nullptr) return fail(RBX_ERR_INVALID, "shard_route: d_send is NULL");
  if (g.T > 0 && (row_fields == nullptr || (g.B > 0 && d_slot == nullptr))) return fail(RBX_ERR_INVALID, "shard_route: single-row lookups need row_fields and d_slot");
  if (g.P > 0 && (pool_field == nullptr || (g.B > 0 && d_inv == nullptr))) return fail(RBX_ERR_INVALID, "shard_route: the pooled lookup needs pool_field and d_inv");
  const int L = g.P ? pool_field->seq_len : 0;
  if (g.P && (L <= 0 || (pool_field->pool != RBX_POOL_MEAN_ID && pool_field->pool != RBX_POOL_SUM_ID)))
    return fail(RBX_ERR_UNSUPPORTED, "shard_route: the pooled lookup must be an id-masked mean or sum over seq_len >= 1");
  const size_t need = rbx_shard_route_workspace_size(geom, L);
  if (d_workspace == nullptr || workspace_bytes < need) return fail(RBX_ERR_WORKSPACE, "shard_route: workspace %zu B < %zu B", workspace_bytes, need);
  // empty single-row slots carry row -1; the offsets of an empty batch are all zero
  if (hipMemsetAsync(d_send, 0xFF, static_cast<size_t>(g.ichunk) * g.W * sizeof(int), s) != hipSuccess)
    return fail(RBX_ERR_LAUNCH, "shard_route: memset failed");
  if (g.B == 0) {
    if (g.P)
      for (int w = 0; w < g.W; ++w)
        if (hipMemsetAsync(d_send + w * g.ichunk + g.off_offs, 0, sizeof(int), s) != hipSuccess) return fail(RBX_ERR_LAUNCH, "shard_route: memset failed");
    return RBX_OK;
  }
  const int n_base = g.T + g.P;
  int* ws = static_cast<int*>(d_workspace);
  const long long tr = shard_tiles(g.B * g.T), tp = shard_tiles(g.B * L);
  int* hist_r = ws;
  int* hist_p = ws + tr * g.W;
  int* cnt = hist_p + tp * g.W;
  if (g.T > 0) {
    ShardCols cols;
    for (int t = 0; t < g.T; ++t) {
      if (row_fields[t].ids == nullptr) return fail(RBX_ERR_INVALID, "shard_route: row lookup %d: ids is NULL", t);
      rc = pack_col(row_fields[t], g.W, &cols.f[t], "row lookup", t);
      if (rc != RBX_OK) return rc;
      cols.f[t].mask_id = kNoId;
    }
    const long long n = g.B * g.T;
    hipLaunchKernelGGL(shard_count_kernel<false>, dim3(static_cast<unsigned>(tr)), dim3(256), 0, s, cols, g.T, n, g.W, hist_r,
                       static_cast<int>(tr), static_cast<int*>(nullptr), d_status);
    hipLaunchKernelGGL(shard_scan_kernel, dim3(g.W), dim3(256), 0, s, hist_r, static_cast<int>(tr), g.cap_rows, d_overflow,
                       d_send, g.ichunk, -1ll);
    hipLaunchKernelGGL(shard_assign_kernel<false>, dim3(static_cast<unsigned>(tr)), dim3(256), 0, s, cols, g.T, n, g.W,
                       g.cap_rows, reinterpret_cast<const long long*>(d_base), n_base, 0, hist_r, static_cast<int>(tr),
                       d_send, g.ichunk, g.off_rows, 0ll, d_slot, static_cast<const int*>(nullptr),
                       static_cast<float*>(nullptr), 0, 0.f);
  }
  if (g.P > 0) {
    if (pool_field->ids == nullptr) return fail(RBX_ERR_INVALID, "shard_route: pooled lookup: ids is NULL");
    ShardCols cols;
    rc = pack_col(*pool_field, g.W, &cols.f[0], "pooled lookup", 0);
    if (rc != RBX_OK) return rc;
    const long long n = g.B * L;
    if (hipMemsetAsync(cnt, 0, static_cast<size_t>(g.B) * sizeof(int), s) != hipSuccess) return fail(RBX_ERR_LAUNCH, "shard_route: memset failed");
    hipLaunchKernelGGL(shard_count_kernel<true>, dim3(static_cast<unsigned>(tp)), dim3(256), 0, s, cols, L, n, g.W, hist_p,
                       static_cast<int>(tp), cnt, d_status);
    hipLaunchKernelGGL(shard_scan_kernel, dim3(g.W), dim3(256), 0, s, hist_p, static_cast<int>(tp), g.cap_pool, d_overflow,
                       d_send, g.ichunk, g.off_offs + g.B);
    hipLaunchKernelGGL(shard_assign_kernel<true>, dim3(static_cast<unsigned>(tp)), dim3(256), 0, s, cols, L, n, g.W,
                       g.cap_pool, reinterpret_cast<const long long*>(d_base), n_base, g.T, hist_p, static_cast<int>(tp),
                       d_send, g.ichunk, 0ll, g.off_offs, static_cast<int*>(nullptr), cnt, d_inv,
                       pool_field->pool == RBX_POOL_MEAN_ID ? 1 : 0, pool_field->eps);
  }
  return check_launch("shard_route kernels");
}

extern "C" int rbx_shard_serve(const rbx_shard_geom_t* geom, const int32_t* d_recv, const float* d_weight,
                               int64_t n_local_rows, float* d_back, int32_t* d_keys, int32_t* d_src,
                               int32_t* d_status, void* stream) {
  using namespace rbx;
  Geom geo;
  int rc = make_geom(geom, &geo);
  if (rc != RBX_OK) return rc;
  if (d_recv == nullptr || d_weight == nullptr || d_back == nullptr || d_keys == nullptr || d_src == nullptr)
    return fail(RBX_ERR_INVALID, "shard_serve: NULL buffer");
  if ((reinterpret_cast<uintptr_t>(d_weight) | reinterpret_cast<uintptr_t>(d_back)) & 15)
    return fail(RBX_ERR_INVALID, "shard_serve: weight / wire buffers must be 16-byte aligned");
  hipStream_t s = as_stream(stream);
  const long long n_keys = (geo.cap_pool + geo.cap_rows) * geo.W;
  if (n_keys == 0) return RBX_OK;
  if (geo.P && hipMemsetAsync(d_keys, 0xFF, static_cast<size_t>(geo.cap_pool) * geo.W * sizeof(int), s) != hipSuccess)
    return fail(RBX_ERR_LAUNCH, "shard_serve: memset failed");          // pooled slots past offsets[B]: no key
  const long long units = static_cast<long long>(geo.W) * (geo.P * geo.B + geo.cap_rows);
  if (units == 0) return RBX_OK;
  RBX_SHARD_DISPATCH(shard_serve_kernel, units, s, geo, d_recv, d_weight, static_cast<long long>(n_local_rows), d_back, d_keys,
                     d_src, d_status);
  return check_launch("shard_serve_kernel");
}

namespace rbx {
static int pack_offs(const Geom& g, const int64_t* col_off, int64_t stride_b, ShardOffs* o) {
  if (col_off == nullptr) return fail(RBX_ERR_INVALID, "shard_combine: col_off is NULL");
  for (int f = 0; f < g.T + g.P; ++f) {
    if (col_off[f] < 0 || col_off[f] + g.D > stride_b) return fail(RBX_ERR_INVALID, "shard_combine: slot %d does not fit the output row", f);
    o->off[f] = col_off[f];
  }
  return RBX_OK;
}
}  // namespace rbx

extern "C" int rbx_shard_combine_fwd(const rbx_shard_geom_t* geom, const float* d_back, const int32_t* d_slot,
                                     const float* d_inv, float* d_out, int64_t out_stride_b, const int64_t* col_off,
                                     void* stream) {
  using namespace rbx;
  Geom geo;
  int rc = make_geom(geom, &geo);
  if (rc != RBX_OK) return rc;
  if (geo.B == 0) return RBX_OK;
  if (d_back == nullptr || d_out == nullptr || (geo.T && d_slot == nullptr) || (geo.P && d_inv == nullptr))
    return fail(RBX_ERR_INVALID, "shard_combine_fwd: NULL buffer");
  ShardOffs offs;
  rc = pack_offs(geo, col_off, out_stride_b, &offs);
  if (rc != RBX_OK) return rc;
  const long long units = geo.B * (geo.T + geo.P);
  hipStream_t s = as_stream(stream);
  RBX_SHARD_DISPATCH(shard_combine_fwd_kernel, units, s, geo, d_back, d_slot, d_inv, d_out, static_cast<long long>(out_stride_b), offs);
  return check_launch("shard_combine_fwd_kernel");
}

extern "C" int rbx_shard_combine_bwd(const rbx_shard_geom_t* geom, const float* d_dout, int64_t dout_stride_b,
                                     const int64_t* col_off, const int32_t* d_slot, const float* d_inv, float* d_gsend,
                                     void* stream) {
  using namespace rbx;
  Geom geo;
  int rc = make_geom(geom, &geo);
  if (rc != RBX_OK) return rc;
  if (geo.B == 0) return RBX_OK;
  if (d_dout == nullptr || d_gsend == nullptr || (geo.T && d_slot == nullptr) || (geo.P && d_inv == nullptr))
    return fail(RBX_ERR_INVALID, "shard_combine_bwd: NULL buffer");
  ShardOffs offs;
  rc = pack_offs(geo, col_off, dout_stride_b, &offs);
  if (rc != RBX_OK) return rc;
  const long long units = geo.B * (geo.T + geo.P);
  hipStream_t s = as_stream(stream);
  RBX_SHARD_DISPATCH(shard_combine_bwd_kernel, units, s, geo, d_dout, static_cast<long long>(dout_stride_b), offs, d_slot, d_inv, d_gsend);
  return check_launch("shard_combine_bwd_kernel");
}
