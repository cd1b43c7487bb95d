// rbx_sampler.hip -- SURVEY 8f-1: negative sampling and the item-corpus feature gather of the
// two-tower training loader, on the GPU (gfx950).
//
// Reference behaviour replaced (paths relative to /root/reference/recbox):
//   sampling_block            matching/pytorch/dataloaders/h5_generator.py:61-84
//       np.random.choice(num_items, size=(n_samples, num_negs), replace=True)        uniform, with replacement
//       ignore_pos_items: the items the query interacted with get probability 0 (renormalised), i.e.
//       uniform over the complement -- rejection sampling draws from exactly that distribution
//   negative_sampling         h5_generator.py:144-181   all_item_indexes = hstack([pos, negs])
//   TrainDataset.__getitem__  h5_generator.py:23-28     item_dict[k] = item_corpus[k][item_indexes]
//   collate_fn                h5_generator.py:49-58     item rows flattened to [B * (1 + num_negs), ...]
//
// The reference draws from numpy's MT19937 stream on the host (one serial stream per worker process);
// that stream cannot be reproduced by a parallel kernel, so the build defines its own counter-based
// generator: Philox4x32-10 (Salmon et al., SC'11), key = seed, counter = (element, attempt).  The
// distribution is the reference's; the bits are the build's and are pinned by the C oracle
// (oracle/recbox_oracle.c: orc_negsample) and the published Philox known-answer vectors.
//
// Integer work, HBM-bound (8 B written per draw; the gather moves row_bytes per index and column): no LDS
// tiling, no MFMA -- coalesced stores, one draw / one 16-byte unit per lane.
#include "rbx_internal.h"

namespace rbx {

constexpr int kMaxAttempts = 64;     // rejection rounds; after them ONE exact draw over the complement of the exclusion list

// uniform item in [0, num_items): high 64 bits of (64 random bits) x num_items
__device__ __forceinline__ long long draw_item(unsigned long long element, unsigned attempt, unsigned long long seed,
                                               unsigned long long num_items) {
  unsigned c[4] = {static_cast<unsigned>(element), static_cast<unsigned>(element >> 32), attempt, 0u};
  Philox::run(c, static_cast<unsigned>(seed), static_cast<unsigned>(seed >> 32));
  const unsigned long long r = (static_cast<unsigned long long>(c[1]) << 32) | c[0];
  return static_cast<long long>(__umul64hi(r, num_items));
}

// is `item` in the sorted list excl[lo, hi)?
__device__ __forceinline__ bool excluded(const long long* __restrict__ excl, long long lo, long long hi, long long item) {
  while (lo < hi) {
    const long long mid = (lo + hi) >> 1;
    const long long v = excl[mid];
    if (v == item) return true;
    if (v < item) lo = mid + 1; else hi = mid;
  }
  return false;
}

__global__ __launch_bounds__(256) void negsample_kernel(const long long num_items, const long long rows, const int num_negs,
                                                        const unsigned long long seed, const unsigned long long offset,
                                                        const long long* __restrict__ pos,
                                                        const long long* __restrict__ query,
                                                        const long long* __restrict__ excl_off,
                                                        const long long* __restrict__ excl_items,
                                                        const long long n_queries, int* __restrict__ status,
                                                        long long* __restrict__ out) {
  const int width = num_negs + (pos != nullptr ? 1 : 0);
  const long long total = rows * width;
  const long long step = static_cast<long long>(gridDim.x) * blockDim.x;
  for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += step) {
    const long long r = i / width;
    const int j = static_cast<int>(i - r * width);
    if (pos != nullptr && j == 0) {
      out[i] = pos[r];                                          // column 0 = the positive item (hstack([pos, negs]))
      continue;
    }
    const int jn = j - (pos != nullptr ? 1 : 0);
    const unsigned long long element = offset + static_cast<unsigned long long>(r) * num_negs + jn;
    long long lo = 0, hi = 0;
    if (excl_off != nullptr) {
      const long long q = query[r];
      if (q < 0 || q >= n_queries) {                              // (numpy would raise IndexError: flagged, no exclusion)
        if (status != nullptr) atomicOr(status, 1);
      } else {
        lo = excl_off[q];
        hi = excl_off[q + 1];
      }
    }
    long long item = 0;
    bool found = false;
    for (unsigned a = 0; a < kMaxAttempts && !found; ++a) {
      item = draw_item(element, a, seed, static_cast<unsigned long long>(num_items));
      found = hi <= lo || !excluded(excl_items, lo, hi, item);
    }
    if (!found && hi - lo < num_items) {
      // the query interacted with (almost) the whole corpus: draw the k-th item of the complement exactly -- k uniform in
      // [0, num_items - n_excl), then step over the sorted exclusion list (every excluded id <= the running item shifts
      // it by one).  Still the reference's distribution (uniform over the items the query did not interact with).
      item = draw_item(element, kMaxAttempts, seed, static_cast<unsigned long long>(num_items - (hi - lo)));
      for (long long k = lo; k < hi && excl_items[k] <= item; ++k) ++item;
    }
    out[i] = item;
  }
}

// ---- row gather over several columns of a corpus ---------------------------------------------------------
struct RowCopy {
  const char* src;
  char* dst;
  long long row_bytes;
  int unit;               // bytes moved per lane step: 16, 8, 4, 2 or 1 (alignment of src, dst and row_bytes)
  int pad;
};
struct RowCopyPack { RowCopy c[RBX_MAX_FIELDS]; };

template <class T>
__device__ __forceinline__ void copy_units(const char* __restrict__ src, char* __restrict__ dst, long long k) {
  reinterpret_cast<T*>(dst)[k] = reinterpret_cast<const T*>(src)[k];
}

// grid.y = column; lanes walk (index q, unit u) pairs, unit fastest: contiguous stores, row-sized random reads
__global__ __launch_bounds__(256) void gather_rows_kernel(const RowCopyPack P, const long long* __restrict__ index,
                                                          const long long n_index, const long long n_rows,
                                                          int* __restrict__ status) {
  const RowCopy c = P.c[blockIdx.y];
  const long long upr = c.row_bytes / c.unit;                  // units per row
  const long long total = n_index * upr;
  const long long step = static_cast<long long>(gridDim.x) * blockDim.x;
  for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += step) {
    const long long q = i / upr;
    const long long u = i - q * upr;
    long long row = index[q];
    if (row < 0 || row >= n_rows) {
      if (status != nullptr) atomicOr(status, 1);
      row = 0;                                                  // numpy would raise IndexError; flagged, not silent
    }
    const char* s = c.src + row * c.row_bytes;
    char* d = c.dst + q * c.row_bytes;
    switch (c.unit) {
      case 16: copy_units<uint4>(s, d, u); break;
      case 8: copy_units<unsigned long long>(s, d, u); break;
      case 4: copy_units<unsigned>(s, d, u); break;
      case 2: copy_units<unsigned short>(s, d, u); break;
      default: copy_units<unsigned char>(s, d, u); break;
    }
  }
}

}  // namespace rbx

extern "C" int rbx_negsample(int64_t num_items, int64_t rows, int32_t num_negs, uint64_t seed, uint64_t offset,
                             const int64_t* d_pos, const int64_t* d_query, const int64_t* d_excl_offsets,
                             const int64_t* d_excl_items, int64_t* d_out, void* stream) {
  return rbx_negsample_checked(num_items, rows, num_negs, seed, offset, d_pos, d_query, d_excl_offsets, d_excl_items, INT64_MAX,
                               nullptr, d_out, stream);
}

extern "C" int rbx_negsample_checked(int64_t num_items, int64_t rows, int32_t num_negs, uint64_t seed, uint64_t offset,
                                     const int64_t* d_pos, const int64_t* d_query, const int64_t* d_excl_offsets,
                                     const int64_t* d_excl_items, int64_t n_queries, int32_t* d_status, int64_t* d_out,
                                     void* stream) {
  using namespace rbx;
  if (num_items <= 0) return fail(RBX_ERR_INVALID, "negsample: num_items=%lld", static_cast<long long>(num_items));
  if (rows < 0 || num_negs < 0) return fail(RBX_ERR_INVALID, "negsample: negative sizes");
  const int width = num_negs + (d_pos != nullptr ? 1 : 0);
  if (rows == 0 || width == 0) return RBX_OK;
  if (d_out == nullptr) return fail(RBX_ERR_INVALID, "negsample: d_out is NULL");
  if ((d_excl_offsets != nullptr) != (d_excl_items != nullptr) || (d_excl_offsets != nullptr && d_query == nullptr))
    return fail(RBX_ERR_INVALID, "negsample: exclusion needs d_query, d_excl_offsets and d_excl_items together");
  const long long total = static_cast<long long>(rows) * width;
  long long blocks = (total + 255) / 256;
  if (blocks > kCUs * 16) blocks = kCUs * 16;
  hipLaunchKernelGGL(negsample_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, as_stream(stream),
                     static_cast<long long>(num_items), static_cast<long long>(rows), num_negs,
                     static_cast<unsigned long long>(seed), static_cast<unsigned long long>(offset),
                     reinterpret_cast<const long long*>(d_pos), reinterpret_cast<const long long*>(d_query),
                     reinterpret_cast<const long long*>(d_excl_offsets), reinterpret_cast<const long long*>(d_excl_items),
                     static_cast<long long>(n_queries), d_status, reinterpret_cast<long long*>(d_out));
  return check_launch("negsample_kernel");
}

extern "C" int rbx_gather_rows(const rbx_rowcopy_t* cols, int32_t n_cols, const int64_t* d_index, int64_t n_index,
                               int64_t n_src_rows, int32_t* d_status, void* stream) {
  using namespace rbx;
  if (cols == nullptr || n_cols <= 0 || n_cols > RBX_MAX_FIELDS) return fail(RBX_ERR_INVALID, "gather_rows: bad column array");
  if (n_index < 0 || n_src_rows < 0) return fail(RBX_ERR_INVALID, "gather_rows: negative sizes");
  if (n_index == 0) return RBX_OK;
  if (d_index == nullptr) return fail(RBX_ERR_INVALID, "gather_rows: d_index is NULL");
  RowCopyPack pack;
  long long max_units = 1;
  for (int i = 0; i < n_cols; ++i) {
    if (cols[i].src == nullptr || cols[i].dst == nullptr || cols[i].row_bytes <= 0)
      return fail(RBX_ERR_INVALID, "gather_rows: column %d: NULL pointer or row_bytes <= 0", i);
    RowCopy& c = pack.c[i];
    c.src = static_cast<const char*>(cols[i].src);
    c.dst = static_cast<char*>(cols[i].dst);
    c.row_bytes = cols[i].row_bytes;
    const uintptr_t bits = reinterpret_cast<uintptr_t>(c.src) | reinterpret_cast<uintptr_t>(c.dst) |
                           static_cast<uintptr_t>(c.row_bytes);
    c.unit = (bits % 16 == 0) ? 16 : (bits % 8 == 0) ? 8 : (bits % 4 == 0) ? 4 : (bits % 2 == 0) ? 2 : 1;
    c.pad = 0;
    const long long units = c.row_bytes / c.unit;
    if (units > max_units) max_units = units;
  }
  long long blocks = (static_cast<long long>(n_index) * max_units + 255) / 256;
  if (blocks > kCUs * 16) blocks = kCUs * 16;
  hipLaunchKernelGGL(gather_rows_kernel, dim3(static_cast<unsigned>(blocks), n_cols), dim3(256), 0, as_stream(stream), pack,
                     reinterpret_cast<const long long*>(d_index), static_cast<long long>(n_index),
                     static_cast<long long>(n_src_rows), d_status);
  return check_launch("gather_rows_kernel");
}
