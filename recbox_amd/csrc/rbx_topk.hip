// rbx_topk.hip -- SURVEY 8f-2: exact top-k retrieval for the evaluation of two-tower models
// (gfx950): per-row top-k selection over a score matrix, the "seen items" penalty and the
// hit flags the ranking metrics are computed from.
//
// Reference behaviour replaced (paths relative to /root/reference/recbox):
//   FaissIndex(IndexFlatIP).search(user_embs, topk=500)      utils/ann/faiss.py:3-15, core/metrics.py:56
//       exact inner-product search: scores = U I^T in fp32 (the fp32-MFMA GEMM rbx_linear_fwd), then the
//       500 largest per row, sorted descending  -> rbx_topk
//   mask[i, train_user2items[q_i]] = 1; scores += -1e9 * mask      core/metrics.py:57-62   -> rbx_penalize_members
//   np.argsort(-scores)[:, :max_topk]                              core/metrics.py:63-64   -> rbx_topk again
//   item in set(true_items)                                        core/metrics.py:78-190  -> rbx_membership
//
// Selection = MSB-first radix select on order-preserving keys (4 passes of 8 bits over the row find the
// exact k-th largest key), one ordered pass collects the winners (ties at the threshold go to the LOWEST
// index -- deterministic; faiss/numpy leave ties unspecified), a bitonic sort in LDS orders the k winners
// by (score descending, index ascending).  A workgroup stages the keys of its 8 192-score segment in LDS, so
// HBM is read once and the five sweeps run out of LDS.  Longer rows are cut into segments (a single query
// still fills the chip): every segment keeps its k best and further launches select among the survivors
// until one segment per row is left.  HBM-bound streaming read + LDS integer work: no MFMA.
#include "rbx_internal.h"

namespace rbx {

constexpr int kTopkMaxK = 1024;
constexpr int kTopkSeg = 8192;        // scores per workgroup: their keys are staged in LDS once (32 KB)

__device__ __forceinline__ unsigned key_of(float v) {
  const unsigned u = __float_as_uint(v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);          // larger float <=> larger key
}
__device__ __forceinline__ unsigned long long pack_winner(unsigned key, long long index) {
  return (static_cast<unsigned long long>(key) << 32) | static_cast<unsigned>(~static_cast<unsigned>(index));
}
__device__ __forceinline__ float value_of(unsigned k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

// One workgroup per (query, segment).  vals: query u at vals + u * row_stride; the segment covers elements
// [s * seg, min(n, (s+1) * seg)).  idx_in == nullptr: the index of an element is its position in the row; otherwise
// idx_in holds the row positions of a previous level's survivors.
// Output slot (u * nseg + s) * K .. + K: the segment's top K (padded with -FLT_MAX / -1 when it is shorter).
__global__ __launch_bounds__(256) void topk_kernel(const float* __restrict__ vals, const long long* __restrict__ idx_in,
                                                   const long long row_stride, const long long n, const int seg,
                                                   const int nseg, const int K, float* __restrict__ out_vals,
                                                   long long* __restrict__ out_idx,
                                                   const long long* __restrict__ remap, const long long remap_stride,
                                                   const bool do_sort, const unsigned* __restrict__ row_len,
                                                   const int* __restrict__ row_state, const int run_state,
                                                   const long long sample_stride, unsigned* __restrict__ thr_out) {
  __shared__ unsigned s_hist[256];
  // a winner = (key << 32) | ~index: ONE 64-bit word per element, so that the sort compares and swaps single words
  // (larger key first, then the smaller index; 0 = empty slot, sinks to the end).  Indices are < 2^32 - 1 (checked).
  __shared__ unsigned long long s_win[kTopkMaxK];
  __shared__ unsigned s_prefix, s_remaining, s_count, s_bin_count;
  __shared__ unsigned s_wave[4];
  const long long u = blockIdx.x / nseg;
  if (row_state != nullptr && row_state[u] != run_state) return;   // this row is served by the other path
  const int s = static_cast<int>(blockIdx.x - u * nseg);
  const long long first = static_cast<long long>(s) * seg;
  const long long n_row = row_len != nullptr ? (static_cast<long long>(row_len[u]) < n ? row_len[u] : n) : n;
  const int len = static_cast<int>((n_row - first < seg) ? (n_row - first > 0 ? n_row - first : 0) : seg);
  const float* row = vals + u * row_stride + first;
  const long long* irow = idx_in != nullptr ? idx_in + u * row_stride + first : nullptr;
  const int want = (K < len) ? K : len;
  const int tid = threadIdx.x;
  __shared__ unsigned s_row[kTopkSeg];
  for (int i = tid; i < len; i += 256) s_row[i] = key_of(row[i * sample_stride]);     // the only pass over HBM

  // ---- radix select on the 64-bit word (key << 32 | ~position): 4 passes over the keys find the key T of the want-th
  // largest element; if more elements equal T than places are left, 4 more passes over THOSE elements' ~position find
  // which of them win (the lowest positions).  All words are distinct, so the winners are a set that does not depend
  // on the order the elements are visited in: the levels need no common order between them.
  auto position_of = [&](int i) -> long long { return irow != nullptr ? irow[i] : first + i; };
  auto radix_pass = [&](const int shift, const bool on_pos, const unsigned key_T) {
    // histogram of one digit over the elements still in play; thread t then owns bin t
    s_hist[tid] = 0u;
    __syncthreads();
    const unsigned prefix = s_prefix;
    const unsigned mask = (shift == 24) ? 0u : (0xFFFFFFFFu << (shift + 8));
    for (int i = tid; i < len; i += 256) {
      const unsigned k = s_row[i];
      if (!on_pos) {
        if ((k & mask) == prefix) atomicAdd(&s_hist[(k >> shift) & 255u], 1u);
      } else if (k == key_T) {
        const unsigned q = ~static_cast<unsigned>(position_of(i));
        if ((q & mask) == prefix) atomicAdd(&s_hist[(q >> shift) & 255u], 1u);
      }
    }
    __syncthreads();
    const unsigned v = s_hist[tid];
    const int ln = tid & 63, wv = tid >> 6;
    unsigned incl = v;                                       // suffix sum inside the wave: elements in bins >= t
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const unsigned dn = __shfl_down(incl, o, 64);
      if (ln + o < 64) incl += dn;
    }
    if (ln == 0) s_wave[wv] = incl;
    const unsigned rem = s_remaining;
    __syncthreads();
    unsigned above = incl - v;
    for (int w = wv + 1; w < 4; ++w) above += s_wave[w];
    if (above < rem && above + v >= rem) {                   // exactly one bin holds the rem-th largest
      s_remaining = rem - above;
      s_prefix = prefix | (static_cast<unsigned>(tid) << shift);
      s_bin_count = v;
    }
    __syncthreads();
  };
  if (tid == 0) { s_prefix = 0u; s_remaining = static_cast<unsigned>(want); s_bin_count = 0u; }
  __syncthreads();
  unsigned T = 0u, T2 = 0u;
  bool tie = false;
  if (want > 0) {
    for (int shift = 24; shift >= 0; shift -= 8) radix_pass(shift, false, 0u);
    T = s_prefix;                                            // exact key of the want-th largest
    if (thr_out != nullptr) {                                // threshold estimation on a sample: that is all
      if (tid == 0) thr_out[u] = T;
      return;
    }
    tie = s_bin_count > s_remaining;                         // more elements equal to T than places left
    __syncthreads();
    if (tie) {
      if (tid == 0) s_prefix = 0u;                           // s_remaining = places left among the elements equal to T
      __syncthreads();
      for (int shift = 24; shift >= 0; shift -= 8) radix_pass(shift, true, T);
      T2 = s_prefix;                                         // ~position of the last winner among them
    }
  }

  // ---- collect the winners (any order) ---------------------------------------------------------------------
  if (tid == 0) s_count = 0u;
  for (int i = tid; i < kTopkMaxK; i += 256) s_win[i] = 0ull;
  __syncthreads();
  if (want > 0) {
    for (int i = tid; i < len; i += 256) {
      const unsigned k = s_row[i];
      if (k < T) continue;
      const long long pos = position_of(i);
      if (k > T || !tie || ~static_cast<unsigned>(pos) >= T2) s_win[atomicAdd(&s_count, 1u)] = pack_winner(k, pos);
    }
  }

  // ---- bitonic sort of the winners, final level only: larger word first = (key descending, position ascending);
  // empty slots (0) sink.  The levels before it hand their winners on unsorted.
  int P = 1;
  while (P < K) P <<= 1;
  for (int size = 2; do_sort && size <= P; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      __syncthreads();
      for (int t = tid; t < P / 2; t += 256) {
        const int lo = 2 * t - (t & (stride - 1));
        const int hi = lo + stride;
        const bool up = (lo & size) == 0;                   // "up" blocks end with the better element first
        const unsigned long long a = s_win[lo], b = s_win[hi];
        if ((a >= b) != up) {                               // a is better than b when its word is larger
          s_win[lo] = b;
          s_win[hi] = a;
        }
      }
    }
  }
  __syncthreads();
  float* ov = out_vals + static_cast<long long>(blockIdx.x) * K;
  long long* oi = out_idx + static_cast<long long>(blockIdx.x) * K;
  for (int t = tid; t < K; t += 256) {
    const unsigned long long w = s_win[t];
    ov[t] = w == 0ull ? -3.402823466e+38f : value_of(static_cast<unsigned>(w >> 32));
    // inside the kernel an element is named by its POSITION in the row (< 2^32 - 1); caller-supplied 64-bit ids are
    // looked up only here, by the launch that writes the final result
    const long long pos = static_cast<long long>(~static_cast<unsigned>(w));
    oi[t] = w == 0ull ? -1ll : (remap != nullptr ? remap[u * remap_stride + pos] : pos);
  }
}

// is `item` in the sorted CSR list of query q?
__device__ __forceinline__ bool csr_member(const long long* __restrict__ off, const long long* __restrict__ items,
                                           long long q, long long item) {
  long long lo = off[q], hi = off[q + 1];
  while (lo < hi) {
    const long long mid = (lo + hi) >> 1;
    const long long v = items[mid];
    if (v == item) return true;
    if (v < item) lo = mid + 1; else hi = mid;
  }
  return false;
}

// mode 0: flags[r, j] = member;  mode 1: scores[r, j] = (float)((double)scores[r, j] + penalty * member)
__global__ __launch_bounds__(256) void membership_kernel(const long long* __restrict__ cand, const long long rows, const int k,
                                                         const long long* __restrict__ query,
                                                         const long long* __restrict__ off,
                                                         const long long* __restrict__ items, const int mode,
                                                         const double penalty, float* __restrict__ scores,
                                                         unsigned char* __restrict__ flags) {
  const long long total = rows * k;
  const long long step = static_cast<long long>(gridDim.x) * blockDim.x;
  for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += step) {
    const long long r = i / k;
    const long long c = cand[i];
    const bool hit = c >= 0 && csr_member(off, items, query[r], c);
    if (mode == 0) flags[i] = hit ? 1 : 0;
    else if (hit) scores[i] = static_cast<float>(static_cast<double>(scores[i]) + penalty);
  }
}

// ---- fast path for long rows -------------------------------------------------------------------------------
// A threshold estimated from a strided sample of the row (topk_kernel in its thr_out mode: the key of the r-th largest
// of 8 192 samples, r chosen so that ~4k + a margin elements of the row are expected above it) turns the first,
// dominant level into ONE streaming sweep that keeps the elements >= threshold: kTopkCand slots per row.  The exact
// selection then runs on those candidates.  A row whose candidates are fewer than k, or do not fit (ties, a skewed
// sample), is flagged and served by the exact multi-level path instead -- both paths check the flag on the device.
constexpr int kTopkCand = kTopkSeg;          // candidate slots per row (one segment of the final selection)
constexpr int kTopkStage = 512;              // candidates one workgroup can hand over from its 8 192 scores

__global__ __launch_bounds__(256) void topk_filter_kernel(const float* __restrict__ vals, const long long row_stride,
                                                          const long long n, const int seg, const int nseg,
                                                          const unsigned* __restrict__ thr, unsigned* __restrict__ cnt,
                                                          unsigned* __restrict__ fail, float* __restrict__ cand_vals,
                                                          long long* __restrict__ cand_pos) {
  __shared__ float s_val[kTopkStage];
  __shared__ unsigned s_pos[kTopkStage];
  __shared__ unsigned s_n, s_base;
  const long long u = blockIdx.x / nseg;
  const int s = static_cast<int>(blockIdx.x - u * nseg);
  const long long first = static_cast<long long>(s) * seg;
  const int len = static_cast<int>((n - first < seg) ? (n - first) : seg);
  const float* row = vals + u * row_stride + first;
  const unsigned T = thr[u];
  if (threadIdx.x == 0) s_n = 0u;
  __syncthreads();
  for (int i = threadIdx.x; i < len; i += 256) {
    const float v = row[i];
    if (key_of(v) >= T) {
      const unsigned slot = atomicAdd(&s_n, 1u);
      if (slot < static_cast<unsigned>(kTopkStage)) {
        s_val[slot] = v;
        s_pos[slot] = static_cast<unsigned>(first + i);
      }
    }
  }
  __syncthreads();
  const unsigned mine = s_n;
  if (mine == 0u) return;
  if (threadIdx.x == 0) {
    unsigned base = 0xFFFFFFFFu;
    if (mine > static_cast<unsigned>(kTopkStage)) atomicOr(&fail[u], 1u);
    else base = atomicAdd(&cnt[u], mine);
    if (base != 0xFFFFFFFFu && base + mine > static_cast<unsigned>(kTopkCand)) {
      atomicOr(&fail[u], 1u);
      base = 0xFFFFFFFFu;
    }
    s_base = base;
  }
  __syncthreads();
  const unsigned base = s_base;
  if (base == 0xFFFFFFFFu) return;
  for (unsigned j = threadIdx.x; j < mine; j += 256) {
    cand_vals[u * kTopkCand + base + j] = s_val[j];
    cand_pos[u * kTopkCand + base + j] = static_cast<long long>(s_pos[j]);
  }
}

// state[u] = 1 when the fast path has everything it needs for row u, else 0 (the exact path takes the row)
__global__ __launch_bounds__(256) void topk_state_kernel(const unsigned* __restrict__ cnt, const unsigned* __restrict__ fail,
                                                         const long long rows, const unsigned need, int* __restrict__ state) {
  const long long u = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (u < rows) state[u] = (fail[u] == 0u && cnt[u] >= need && cnt[u] <= static_cast<unsigned>(kTopkCand)) ? 1 : 0;
}

static int topk_nseg(long long n) { return static_cast<int>((n + kTopkSeg - 1) / kTopkSeg); }

// sample rank for the threshold estimate, or 0 when the fast path does not apply (short rows, k not selective)
static int topk_sample_rank(long long n, int k, long long* m_out, long long* stride_out) {
  if (n <= 2ll * kTopkSeg) return 0;
  const long long m = kTopkSeg, stride = n / m;
  const long long r = (4ll * k * m + n - 1) / n + 8;
  *m_out = m;
  *stride_out = stride;
  return (r < m / 4) ? static_cast<int>(r) : 0;
}

static size_t topk_fast_bytes(long long rows) {
  return static_cast<size_t>(rows) * (3 * sizeof(unsigned) + sizeof(int)) + 256 +
         static_cast<size_t>(rows) * kTopkCand * (sizeof(float) + sizeof(long long)) + 256;
}

}  // namespace rbx

extern "C" size_t rbx_topk_workspace_size(int64_t rows, int64_t n, int32_t k) {
  if (rows <= 0 || n <= 0 || k <= 0) return 0;
  const int nseg = rbx::topk_nseg(n);
  if (nseg <= 1) return 0;
  // two survivor buffers (levels ping-pong between them); the first level is the largest
  const size_t level = static_cast<size_t>(rows) * nseg * k * (sizeof(float) + sizeof(int64_t)) + 256;
  long long m, stride;
  return 2 * level + (rbx::topk_sample_rank(n, k, &m, &stride) > 0 ? rbx::topk_fast_bytes(rows) : 0);
}

extern "C" int rbx_topk(const float* d_scores, const int64_t* d_index, int64_t rows, int64_t n, int64_t row_stride,
                        int32_t k, float* d_out_scores, int64_t* d_out_index, void* d_workspace, size_t workspace_bytes,
                        void* stream) {
  using namespace rbx;
  if (rows < 0 || n < 0) return fail(RBX_ERR_INVALID, "topk: negative sizes");
  if (k <= 0 || k > kTopkMaxK) return fail(RBX_ERR_UNSUPPORTED, "topk: k=%d not in [1,%d]", k, kTopkMaxK);
  if (rows == 0) return RBX_OK;
  if (d_out_scores == nullptr || d_out_index == nullptr) return fail(RBX_ERR_INVALID, "topk: NULL output");
  if (n > 0 && d_scores == nullptr) return fail(RBX_ERR_INVALID, "topk: d_scores is NULL");
  if (row_stride < n) return fail(RBX_ERR_INVALID, "topk: row_stride < n");
  if (n >= 0xFFFFFFFFll) return fail(RBX_ERR_UNSUPPORTED, "topk: rows of 2^32 - 1 scores or more are not supported");
  int nseg = n > 0 ? topk_nseg(n) : 1;
  if (rows * static_cast<long long>(nseg) >= INT_MAX) return fail(RBX_ERR_UNSUPPORTED, "topk: too many rows");
  hipStream_t s = as_stream(stream);
  long long* oidx = reinterpret_cast<long long*>(d_out_index);
  if (nseg > 1 && (d_workspace == nullptr || workspace_bytes < rbx_topk_workspace_size(rows, n, k)))
    return fail(RBX_ERR_WORKSPACE, "topk: workspace too small");
  const size_t level_bytes = nseg > 1 ? static_cast<size_t>(rows) * nseg * k * (sizeof(float) + sizeof(int64_t)) + 256 : 0;
  const size_t half = level_bytes;
  const unsigned* no_len = nullptr;
  unsigned* no_thr = nullptr;

  // ---- fast path: sample threshold -> one filtering sweep -> exact selection among the candidates ------------
  long long m = 0, sstride = 1;
  const int rank = (nseg > 1) ? topk_sample_rank(n, k, &m, &sstride) : 0;
  int* state = nullptr;
  if (rank > 0) {
    char* fast = static_cast<char*>(d_workspace) + 2 * level_bytes;
    unsigned* thr = reinterpret_cast<unsigned*>(fast);
    unsigned* cnt = thr + rows;
    unsigned* failed = cnt + rows;
    state = reinterpret_cast<int*>(failed + rows);
    char* cbase = fast + ((static_cast<size_t>(rows) * (3 * sizeof(unsigned) + sizeof(int)) + 255) / 256) * 256;
    long long* cpos = reinterpret_cast<long long*>(cbase);
    float* cval = reinterpret_cast<float*>(cpos + rows * kTopkCand);
    if (hipMemsetAsync(cnt, 0, static_cast<size_t>(rows) * 2 * sizeof(unsigned), s) != hipSuccess)
      return fail(RBX_ERR_LAUNCH, "topk: memset of the candidate counters failed");
    // (1) threshold = key of the rank-th largest of m strided samples of the row
    hipLaunchKernelGGL(topk_kernel, dim3(static_cast<unsigned>(rows)), dim3(256), 0, s, d_scores,
                       static_cast<const long long*>(nullptr), static_cast<long long>(row_stride), m, kTopkSeg, 1, rank,
                       static_cast<float*>(nullptr), static_cast<long long*>(nullptr), static_cast<const long long*>(nullptr),
                       0ll, false, no_len, static_cast<const int*>(nullptr), 0, sstride, thr);
    // (2) one sweep over the scores keeps what is >= threshold
    hipLaunchKernelGGL(topk_filter_kernel, dim3(static_cast<unsigned>(rows * nseg)), dim3(256), 0, s, d_scores,
                       static_cast<long long>(row_stride), static_cast<long long>(n), kTopkSeg, nseg, thr, cnt, failed, cval,
                       cpos);
    // (3) which rows have what they need
    hipLaunchKernelGGL(topk_state_kernel, dim3(static_cast<unsigned>((rows + 255) / 256)), dim3(256), 0, s, cnt, failed,
                       static_cast<long long>(rows), static_cast<unsigned>(k), state);
    // (4) exact selection + sort among the candidates of those rows
    hipLaunchKernelGGL(topk_kernel, dim3(static_cast<unsigned>(rows)), dim3(256), 0, s, cval, cpos,
                       static_cast<long long>(kTopkCand), static_cast<long long>(kTopkCand), kTopkSeg, 1, k, d_out_scores, oidx,
                       reinterpret_cast<const long long*>(d_index), static_cast<long long>(row_stride), true, cnt, state, 1, 1ll,
                       no_thr);
    int rc = check_launch("topk fast path");
    if (rc != RBX_OK) return rc;
  }

  // ---- exact multi-level path: every row without the fast path, or the rows it had to give up ------------------
  const float* vals = d_scores;
  const long long* idx = nullptr;                             // level 1 names elements by their position in the row
  long long stride = row_stride, len = n;
  int level = 0;
  while (nseg > 1) {                                          // every level keeps k survivors per segment
    char* buf = static_cast<char*>(d_workspace) + (level & 1) * half;
    const long long cand = static_cast<long long>(nseg) * k;
    long long* cidx = reinterpret_cast<long long*>(buf);
    float* cval = reinterpret_cast<float*>(cidx + rows * cand);
    hipLaunchKernelGGL(topk_kernel, dim3(static_cast<unsigned>(rows * nseg)), dim3(256), 0, s, vals, idx, stride, len,
                       kTopkSeg, nseg, k, cval, cidx, static_cast<const long long*>(nullptr), 0ll, false, no_len, state, 0, 1ll,
                       no_thr);
    vals = cval;
    idx = cidx;
    stride = len = cand;
    nseg = topk_nseg(cand);
    ++level;
    if (k >= kTopkSeg) return fail(RBX_ERR_UNSUPPORTED, "topk: k too large for the segment size");
  }
  hipLaunchKernelGGL(topk_kernel, dim3(static_cast<unsigned>(rows)), dim3(256), 0, s, vals, idx, stride, len, kTopkSeg, 1, k,
                     d_out_scores, oidx, reinterpret_cast<const long long*>(d_index), static_cast<long long>(row_stride), true,
                     no_len, state, 0, 1ll, no_thr);
  return check_launch("topk_kernel");
}

extern "C" int rbx_membership(const int64_t* d_candidates, int64_t rows, int32_t k, const int64_t* d_query,
                              const int64_t* d_offsets, const int64_t* d_items, uint8_t* d_flags, void* stream) {
  using namespace rbx;
  if (rows < 0 || k <= 0) return fail(RBX_ERR_INVALID, "membership: bad sizes");
  if (rows == 0) return RBX_OK;
  if (!d_candidates || !d_query || !d_offsets || !d_items || !d_flags) return fail(RBX_ERR_INVALID, "membership: NULL tensor");
  long long blocks = (static_cast<long long>(rows) * k + 255) / 256;
  if (blocks > kCUs * 16) blocks = kCUs * 16;
  hipLaunchKernelGGL(membership_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, as_stream(stream),
                     reinterpret_cast<const long long*>(d_candidates), static_cast<long long>(rows), k,
                     reinterpret_cast<const long long*>(d_query), reinterpret_cast<const long long*>(d_offsets),
                     reinterpret_cast<const long long*>(d_items), 0, 0.0, static_cast<float*>(nullptr), d_flags);
  return check_launch("membership_kernel");
}

extern "C" int rbx_penalize_members(const int64_t* d_candidates, int64_t rows, int32_t k, const int64_t* d_query,
                                    const int64_t* d_offsets, const int64_t* d_items, double penalty, float* d_scores,
                                    void* stream) {
  using namespace rbx;
  if (rows < 0 || k <= 0) return fail(RBX_ERR_INVALID, "penalize_members: bad sizes");
  if (rows == 0) return RBX_OK;
  if (!d_candidates || !d_query || !d_offsets || !d_items || !d_scores)
    return fail(RBX_ERR_INVALID, "penalize_members: NULL tensor");
  long long blocks = (static_cast<long long>(rows) * k + 255) / 256;
  if (blocks > kCUs * 16) blocks = kCUs * 16;
  hipLaunchKernelGGL(membership_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, as_stream(stream),
                     reinterpret_cast<const long long*>(d_candidates), static_cast<long long>(rows), k,
                     reinterpret_cast<const long long*>(d_query), reinterpret_cast<const long long*>(d_offsets),
                     reinterpret_cast<const long long*>(d_items), 1, penalty, d_scores,
                     static_cast<unsigned char*>(nullptr));
  return check_launch("membership_kernel");
}
