// rbx_bwd_common.h -- plan, sort entry and register-fragment helpers shared by the
// generic embedding backward (rbx_embed_bwd.hip) and the fused FM backward
// (rbx_fm_fused.hip).
#pragma once
#include "rbx_internal.h"

namespace rbx {

constexpr int kSortThreads = 256;
#define RBX_SORT_ITEMS 8
constexpr int kSortItems = RBX_SORT_ITEMS;                 // per thread
constexpr int kSortTile = kSortThreads * kSortItems;       // 2048 pairs per workgroup
constexpr int kChainTiles = 64;                            // (BwdPlan::chained)
constexpr int kChainPasses = 4;                            // 32-bit keys in digits of >= 8 bits
constexpr int kRadix = 256;                                // bins of an 8-bit digit (the sort also runs 10- and 11-bit digits)
// Widest digit the plan may choose.  10- and 11-bit digits (2 passes instead of 3 for 1 M-row tables) were measured on
// the Criteo shape and LOST: a 2048-pair tile then scatters into 1024 buckets of ~2 pairs (8-byte runs instead of
// 32-byte ones) and its per-digit steps are 4x longer -- radix_scatter 32.5 us per pass instead of 17.8, the sort 114 us
// instead of 105 (profiles/r02/sort_variants.txt).  The kernels stay templated on the digit width.
#define RBX_MAX_RADIX_BITS 8
constexpr int kMaxRadixBits = RBX_MAX_RADIX_BITS;
// sorted pairs per lane group in the reduce.  Measured on the Criteo shape (profiles/r02/reduce_variants.txt): the kernel
// sits at 77-86 us whatever the chunk, the lookups in flight and the occupancy are -- 40 pairs with 3 waves per SIMD
// (no register spill, 666 workgroups: all resident at once) is the best of them at 80 us; 16, 48, 56 and 64 are slower.
#define RBX_CHUNK 40
constexpr int kChunk = RBX_CHUNK;                          // the LARGEST chunk; a plan with few pairs takes a shorter one (BwdPlan::chunk)
constexpr unsigned kLocalBits = 26;                        // val = slot << 26 | (b*L + l)
constexpr unsigned kLocalMask = (1u << kLocalBits) - 1u;
constexpr int kNumSamples = 256;                            // samples per workgroup in the numeric-feature reduction

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

struct RedField {            // 48 B
  float* grad;
  float* grad2;              // fused FM: the dim-1 LR table's grad (same ids)
  const float* table;        // fused FM: the embedding table (dW = A - cnt * w)
  unsigned row_base;
  int out_off;
  short dim;
  short seq_len;
  unsigned char pool, slot;
  short reserved;
  int table_stride;          // fused FM: floats between rows of `table` (packed [vocab, stride] storage); grads are [vocab, dim]
  int reserved2;
};
struct RedPack { RedField f[RBX_MAX_FIELDS]; };

// The sort is SEGMENTED: lookups are laid out field after field, and the fields of one table (or of tables that share
// fields with it) form a contiguous segment with its own contiguous row range -- so only the row number INSIDE the
// segment has to be sorted, in ceil(bits / digit) passes over digit bits each (Criteo: 1 M-row tables -> 2 passes of
// 10 bits instead of 3 passes of 8 over the 23-bit global row).  A sort tile never straddles two segments.
struct SegPack {
  unsigned tile0[RBX_MAX_FIELDS + 1];   // first tile of every segment (+ total)
  unsigned lk0[RBX_MAX_FIELDS + 1];     // first lookup of every segment (+ total)
  unsigned row0[RBX_MAX_FIELDS];        // first global row of the segment
  // A segment only takes part in the LAST ceil(bits of its row range / digit) passes of the sort: build_keys puts its
  // pairs into the buffer pass `first_pass` reads, and the tiles of the segment leave the earlier passes at once.
  // Criteo shape: 8 tables of <= 255 rows sort in 1 pass, 10 of <= 65 535 rows in 2, the 8 large ones in 3 -- 52
  // (table, pass) units of work instead of 78.
  unsigned char first_pass[RBX_MAX_FIELDS];
  int n;
};
static_assert(sizeof(KeyPack) + sizeof(SegPack) + 96 <= 4096, "build_keys_kernel's arguments must fit the 4 KiB kernarg segment");

// (segment, first lookup, lookups) of the tile a workgroup owns
__device__ __forceinline__ void seg_of_tile(const SegPack& S, unsigned tile, int* seg, unsigned* first, unsigned* count) {
  int lo = 0, hi = S.n - 1;                         // last segment with tile0 <= tile
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (S.tile0[mid] <= tile) lo = mid; else hi = mid - 1;
  }
  *seg = lo;
  const unsigned f = S.lk0[lo] + (tile - S.tile0[lo]) * static_cast<unsigned>(kSortTile);
  const unsigned end = S.lk0[lo + 1];
  *first = f;
  *count = (end - f < static_cast<unsigned>(kSortTile)) ? end - f : static_cast<unsigned>(kSortTile);
}

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
  int radix_bits = 8;          // digit width of a pass: 8, 10 or 11
  SegPack segs;
  int max_dim = 1;
  bool vec = true;
  // workspace layout (byte offsets)
  size_t off_keys[2], off_vals[2], off_hist, off_ssum, off_head, off_tail, off_flags, off_fin, off_long, off_num, bytes;
  // Round 5: when no segment has more than kChainTiles tiles, the sort is 1 + passes launches -- build_keys counts the digits
  // of EVERY pass per tile (a segment's digit totals do not depend on the order of its pairs), and a scatter workgroup gets
  // the start of its runs from those totals plus the counts the tiles in front of it in its segment publish as they go
  // (off_hist then holds [pass][tile][digit] counts of build_keys, the same again for the published counts, and
  // [pass][tile] flags) -- instead of 1 + 3 x passes with histogram and scan kernels in between.
  bool chained = false;
  unsigned n_tiles = 0, n_chunks = 0, num_blocks = 0;
  unsigned long_cap = kCUs * 2;  // workgroups of the long fix-up launch (one per long chain, grid-stride): every one of them
                               // arrives at one counter, so a plan that expects no long chains (fused FM with its small tables
                               // on the sort-free path) launches few
  int chunk = kChunk;          // sorted pairs per lane group of the reduce: kChunk, shorter (>= 16) when that leaves fewer than 32 768 chunks
  int sum_stride = 1;          // floats per chunk summary (max_dim + extra)
};


// Build the plan for `fields` (definition in rbx_embed_bwd.hip).  `extra_dim` floats are
// reserved behind every chunk summary (the fused FM backward keeps sum(g) there).
int make_plan(const rbx_field_t* fields, int n, int64_t B, const float* dout, int64_t stride_b, BwdPlan* p,
              int extra_dim = 0);
// build_keys + LSD radix passes; sorted pairs end up in key/val buffer (passes & 1).
int run_sort(const BwdPlan& p, char* ws, int* d_status, hipStream_t s);

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
  // The row as it is: no arithmetic on the loaded values and no branch around a load -- lanes beyond `dim` re-read the row's
  // last vector (every consumer stores lanes below `dim` only).  fma_from's `if (e < dim) a += w * load` is a branch per load
  // with the wait for it inside: a batch of U lookups written with it is U dependent round trips (segment_reduce_kernel).
  __device__ __forceinline__ void load_from(const float* row, int dim, int lane_g) {
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      int e = (lane_g + u * G) * W;
      e = e < dim ? e : dim - W;
      if constexpr (VEC) {
        const float4 t = *reinterpret_cast<const float4*>(row + e);
        a[u * 4 + 0] = t.x; a[u * 4 + 1] = t.y; a[u * 4 + 2] = t.z; a[u * 4 + 3] = t.w;
      } else {
        a[u] = row[e];
      }
    }
  }
  __device__ __forceinline__ void load_from_nt(const float* row, int dim, int lane_g) {
    typedef float v4f __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      int e = (lane_g + u * G) * W;
      e = e < dim ? e : dim - W;
      if constexpr (VEC) {
        const v4f t = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(row + e));
        a[u * 4 + 0] = t.x; a[u * 4 + 1] = t.y; a[u * 4 + 2] = t.z; a[u * 4 + 3] = t.w;
      } else {
        a[u] = __builtin_nontemporal_load(row + e);
      }
    }
  }
  __device__ __forceinline__ void scale(float w) {
#pragma unroll
    for (int i = 0; i < NV * W; ++i) a[i] *= w;
  }
  // streaming variants: rows that are touched once per call should not evict the L2-resident
  // gather operands (S, g), so they use the non-temporal cache policy
  __device__ __forceinline__ void add_from_nt(const float* row, int dim, int lane_g) {
    typedef float v4f __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const int e = (lane_g + u * G) * W;
      if (e < dim) {
        if constexpr (VEC) {
          const v4f t = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(row + e));
          a[u * 4 + 0] += t.x; a[u * 4 + 1] += t.y; a[u * 4 + 2] += t.z; a[u * 4 + 3] += t.w;
        } else {
          a[u] += __builtin_nontemporal_load(row + e);
        }
      }
    }
  }
  __device__ __forceinline__ void store_nt(float* row, int dim, int lane_g) const {
    typedef float v4f __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const int e = (lane_g + u * G) * W;
      if (e < dim) {
        if constexpr (VEC) {
          v4f t;
          t.x = a[u * 4]; t.y = a[u * 4 + 1]; t.z = a[u * 4 + 2]; t.w = a[u * 4 + 3];
          __builtin_nontemporal_store(t, reinterpret_cast<v4f*>(row + e));
        } else {
          __builtin_nontemporal_store(a[u], row + e);
        }
      }
    }
  }
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
constexpr int kFlagZero = 4;   // the chunk's tail summary is exactly zero (the fix-ups need not read it)


}  // namespace rbx
