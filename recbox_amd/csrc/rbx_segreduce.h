// rbx_segreduce.h -- policy-templated segmented reduction over sorted (row, lookup)
// pairs: the second half of the deterministic embedding backward.
//
// A Policy says what one lookup contributes and how a finished run is written:
//   struct Policy {
//     struct Args {...};                                   // by-value kernel argument
//     static constexpr bool kHasCount;                     // summaries carry the extra scalar
//     template <class F> static __device__ void contribute(const Args&, const RedField&, unsigned local,
//                                                          int lane_g, F& frag, float& cnt);
//     template <class F> static __device__ void prefetch(const Args&, const RedField&, unsigned row,
//                                                        int lane_g, F& pre);   // loads flush() needs
//     template <class F> static __device__ void flush(const Args&, const RedField&, unsigned row,
//                                                     const F& acc, float cnt, const F& pre, int lane_g);
//   };
// `frag` is the lane's slice of a D-vector, `cnt` one extra scalar per run (unused by the
// generic policy, sum of upstream grads for the fused FM policy).  Chunk summaries are
// `sum_stride = max_dim + extra` floats; cnt lives at offset max_dim.
#pragma once
#include "rbx_bwd_common.h"

namespace rbx {

template <class F>
__device__ __forceinline__ void frag_add(F& a, const F& b) {
#pragma unroll
  for (int q = 0; q < static_cast<int>(sizeof(a.a) / sizeof(float)); ++q) a.a[q] += b.a[q];
}

constexpr unsigned kLongSlots = 512;  // long chains that may be shared between workgroups (one arrival counter + partials each)
constexpr unsigned kLongSplit = 16;   // workgroups per long chain at most
constexpr int kFinList = 3;      // fin[0] work-list length, fin[1] long-list length, fin[2] arrival counter, then the lists
// Interior runs are written directly; a run that crosses the chunk border leaves a head /
// tail summary and a flag, and the chunk where such a run ENDS is queued for the fix-up.
// waves per SIMD the register budget is sized for: at 4 (128 VGPRs) the FM policy spilled 28 bytes per lane to scratch
#define RBX_REDUCE_WAVES 3
template <class Policy, int G, int NV, bool VEC>
__global__ __launch_bounds__(256, RBX_REDUCE_WAVES) void segment_reduce_kernel(const RedPack P, const int n_cat,
                                                             const typename Policy::Args args,
                                                             const unsigned* __restrict__ keys,
                                                             const unsigned* __restrict__ vals, const unsigned n,
                                                             const unsigned sentinel, float* __restrict__ head,
                                                             float* __restrict__ tail, int* __restrict__ flags,
                                                             unsigned* __restrict__ fin, const int max_dim,
                                                             const int sum_stride, const unsigned n_chunks,
                                                             const unsigned chunk) {
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
  const unsigned s = c * chunk;
  const unsigned e = (s + chunk < n) ? s + chunk : n;
  const unsigned key_before = (s > 0) ? keys[s - 1] : sentinel;
  const unsigned key_after = (e < n) ? keys[e] : sentinel;
  unsigned cur = keys[s];
  const bool open_in = (s > 0) && (cur == key_before) && (cur != sentinel);
  bool seen_boundary = false;
  unsigned cur_val = vals[s];
  F acc, pre_last;
  acc.zero();
  pre_last.zero();
  float cnt = 0.f;
  bool head_done = false;
#define RBX_REDUCE_U 8
  // lookups in flight per lane group: 8 for rows of up to 32 floats (the FM tables; 16 measured slower), 4 for wider rows
  // (D = 64 / 128: cfg 4 5.32 -> 5.29 ms, cfg 3 1.93 -> 1.82 ms with 4 instead of 8; with 16 cfg 3 took 2.87 ms)
  constexpr int U = (G * NV * F::W <= 32) ? RBX_REDUCE_U : 4;
  for (unsigned i0 = s; i0 < e; i0 += U) {
    unsigned kk[U + 1], vv[U];
    {
      // the G lanes of a group split the batch's key/value loads and exchange them by shuffle
      constexpr int PER = (U + G - 1) / G;
      unsigned mk[PER], mv[PER];
#pragma unroll
      for (int k = 0; k < PER; ++k) {
        const unsigned i = i0 + k * G + lane_g;
        const bool ok = (k * G + lane_g < U) && (i < e);
        mk[k] = ok ? __builtin_nontemporal_load(keys + i) : sentinel;
        mv[k] = ok ? __builtin_nontemporal_load(vals + i) : 0u;
      }
      const int gbase = (threadIdx.x & 63) & ~(G - 1);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        kk[u] = __shfl(mk[u / G], gbase + (u % G), 64);
        vv[u] = __shfl(mv[u / G], gbase + (u % G), 64);
      }
    }
    kk[U] = (i0 + U < e) ? keys[i0 + U] : key_after;          // key that follows the batch
    F rows[U], pre[U];
    float rc[U];
#define RBX_REDUCE_BATCHED 1
#if RBX_REDUCE_BATCHED
    // Every load of the batch first, with NO use in between, then the arithmetic.  Written as "if (valid) rows[u] += w * load"
    // the batch was U dependent round trips: the compiler waits for a load inside the block that uses it (the ISA of the first
    // form: load, s_waitcnt vmcnt(0), eight times over).  The loads stay under `valid` -- a padding entry of the sharded
    // stores' exchange buffers carries no lookup whose addresses could be read -- which costs a branch but no wait.
    float wu[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      rows[u].zero();
      wu[u] = 0.f;
      if (kk[u] != sentinel && i0 + u < e)
        Policy::fetch(args, sf[vv[u] >> kLocalBits], vv[u] & kLocalMask, lane_g, rows[u], wu[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      pre[u].zero();
      if (kk[u] != sentinel && i0 + u < e) {
        // a run ends after this lookup: fetch what flush() will need NOW (raw: nothing here uses it), so that the random
        // row read overlaps the other loads of the batch instead of serialising the walk
        const unsigned nxt = (i0 + u + 1 < e) ? kk[u + 1] : key_after;
        if (nxt != kk[u]) Policy::prefetch_raw(args, sf[vv[u] >> kLocalBits], kk[u] - sf[vv[u] >> kLocalBits].row_base, lane_g, pre[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool valid = kk[u] != sentinel && i0 + u < e;
      const float w = Policy::weight(args, wu[u]);
      if (valid) rows[u].scale(w); else rows[u].zero();
      rc[u] = (valid && Policy::kHasCount) ? w : 0.f;
    }
#else
#pragma unroll
    for (int u = 0; u < U; ++u) {
      rows[u].zero();
      pre[u].zero();
      rc[u] = 0.f;
      if (kk[u] != sentinel && i0 + u < e) {
        const RedField& fd = sf[vv[u] >> kLocalBits];
        Policy::contribute(args, fd, vv[u] & kLocalMask, lane_g, rows[u], rc[u]);
        // a run ends after this lookup: fetch what flush() will need NOW, so that the random
        // row read overlaps the other loads of the batch instead of serialising the walk
        const unsigned nxt = (i0 + u + 1 < e) ? kk[u + 1] : key_after;
        if (nxt != kk[u]) Policy::prefetch(args, fd, kk[u] - fd.row_base, lane_g, pre[u]);
      }
    }
#endif
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (i0 + u >= e) break;
      if (kk[u] != cur) {                               // first lookup of a new run
        seen_boundary = true;
        acc.zero();
        cnt = 0.f;
        cur = kk[u];
        cur_val = vv[u];
      }
      frag_add(acc, rows[u]);
      cnt += rc[u];
      const unsigned nxt = (i0 + u + 1 < e) ? kk[u + 1] : key_after;
      const bool last_in_chunk = (i0 + u + 1 >= e);
      if (cur != sentinel && nxt != cur && !last_in_chunk) {   // the run ends here, inside the chunk
        const RedField& fd = sf[cur_val >> kLocalBits];
        if (!seen_boundary && open_in) {
          float* dst = head + static_cast<size_t>(c) * sum_stride;
          acc.store(dst, fd.dim, lane_g);
          if (Policy::kHasCount && lane_g == 0) dst[max_dim] = cnt;
          head_done = true;
        } else {
          Policy::flush(args, fd, cur - fd.row_base, acc, cnt, pre[u], lane_g);
        }
      }
      if (last_in_chunk) pre_last = pre[u];
    }
  }
  // the run that is open at the end of the chunk
  int flag = 0;
  if (cur != sentinel) {
    const RedField& fd = sf[cur_val >> kLocalBits];
    const bool open_out = (e < n) && (key_after == cur);
    const bool is_head = !seen_boundary && open_in;
    if (open_out) {
      float* dst = tail + static_cast<size_t>(c) * sum_stride;
      acc.store(dst, fd.dim, lane_g);
      if (Policy::kHasCount && lane_g == 0) dst[max_dim] = cnt;
      if (is_head) flag |= kFlagPass;
      // A tail that is exactly zero (every contribution had a zero upstream gradient: masked positions) is marked, so
      // that the fix-up of a run of hundreds of thousands of such lookups -- SASRec's pad id -- walks flags instead of
      // summing zeros.  Exact: NaN / inf products are not zero and are still read.
      bool zero = (cnt == 0.f);
#pragma unroll
      for (int q = 0; q < static_cast<int>(sizeof(acc.a) / sizeof(float)); ++q) zero = zero && (acc.a[q] == 0.f);
#pragma unroll
      for (int o = 1; o < G; o <<= 1) zero = zero && (__shfl_xor(static_cast<int>(zero), o, 64) != 0);
      if (zero) flag |= kFlagZero;
    } else if (is_head) {
      float* dst = head + static_cast<size_t>(c) * sum_stride;
      acc.store(dst, fd.dim, lane_g);
      if (Policy::kHasCount && lane_g == 0) dst[max_dim] = cnt;
      flag |= kFlagFin;
    } else {
      Policy::flush(args, fd, cur - fd.row_base, acc, cnt, pre_last, lane_g);
    }
  }
  if (head_done) flag |= kFlagFin;
  if (lane_g == 0) {
    flags[c] = flag;
    if (flag & kFlagFin) fin[kFinList + atomicAdd(fin, 1u)] = c;     // fix-up work list (order is irrelevant)
  }
}

// Runs that cross chunk borders, pass 1: one lane group per finalising chunk walks back at
// most kShortHops chunks (the common case: a run of 33..300 lookups); longer chains are
// queued for the cooperative pass below.
constexpr int kShortHops = 8;

template <class Policy, int G, int NV, bool VEC>
__global__ __launch_bounds__(256) void segment_fixup_short_kernel(const RedPack P, const typename Policy::Args args,
                                                                  const unsigned* __restrict__ keys,
                                                                  const unsigned* __restrict__ vals,
                                                                  const float* __restrict__ head,
                                                                  const float* __restrict__ tail,
                                                                  const int* __restrict__ flags,
                                                                  unsigned* __restrict__ fin, const int max_dim,
                                                                  const int sum_stride, const unsigned n_chunks,
                                                                  const unsigned chunk) {
  using F = Frag<G, NV, VEC>;
  const int lane_g = threadIdx.x % G;
  const unsigned ngroups = gridDim.x * (blockDim.x / G);
  const unsigned count = fin[0];
  for (unsigned idx = blockIdx.x * (blockDim.x / G) + threadIdx.x / G; idx < count; idx += ngroups) {
    const unsigned c = fin[kFinList + idx];
    // how long is the chain?  (flags only: cheap, L2-resident)
    int hops = 0;
    bool closed = false;
    for (long long j = static_cast<long long>(c) - 1; j >= 0 && hops < kShortHops; --j) {
      ++hops;
      if (!(flags[j] & kFlagPass)) { closed = true; break; }
    }
    if (!closed && c > 0) {
      if (lane_g == 0) {
        const unsigned pos = atomicAdd(fin + 1, 1u);
        fin[kFinList + n_chunks + pos] = c;
        if (pos < kLongSlots) fin[kFinList + 2 * n_chunks + pos] = 0;      // arrival counter of a chain shared by several workgroups
      }
      continue;
    }
    const unsigned s = c * chunk;
    const unsigned key = keys[s];
    const RedField fd = P.f[vals[s] >> kLocalBits];
    F acc, pre;
    acc.zero();
    pre.zero();
    Policy::prefetch(args, fd, key - fd.row_base, lane_g, pre);
    const float* src = head + static_cast<size_t>(c) * sum_stride;
    acc.add_from(src, fd.dim, lane_g);
    float cnt = Policy::kHasCount ? src[max_dim] : 0.f;
    for (int h = 1; h <= hops; ++h) {                       // independent loads, fixed summation order
      if (flags[c - h] & kFlagZero) continue;               // (adds exact zeros)
      const float* t = tail + static_cast<size_t>(c - h) * sum_stride;
      acc.add_from(t, fd.dim, lane_g);
      if (Policy::kHasCount) cnt += t[max_dim];
    }
    Policy::flush(args, fd, key - fd.row_base, acc, cnt, pre, lane_g);
  }
}

// pass 2: long chains, workgroup-cooperative: the 256/G lane groups of a workgroup walk the pass-through chunks
// backwards 256/G tails per step (flags first, a ballot + LDS min finds where the run
// started); partial sums are combined by a fixed xor butterfly inside each wave and a fixed
// wave order across waves, so the result is order-deterministic.  A 21 845-lookup run
// (V=3 at B=65 536) is 683 chunks = 2 window steps of 512 chunks at D=16 instead of 683 dependent loads;
// a 370 000-lookup run (the pad id of SASRec's [4096, 200] id blocks) is 90 steps.
// When the launch has more workgroups than long chains (a FEW hot rows: one item with 10^5 .. 10^6 non-zero gradients),
// KW = min(16, workgroups / chains) of them share a chain: workgroup jw takes every KW-th window (a window belongs to the
// chain iff the last sorted pair of its nearest chunk carries the chain's key: one load), leaves its partial in a slot of
// its own, and the LAST one to arrive (one counter per chain, zeroed when the chain was listed) adds the KW partials in
// slot order and flushes the row.  KW depends on the number of chains and the launch only, every partial on (chain, jw, KW)
// only: the same bits from run to run, whichever workgroup finishes.
template <class Policy, int G, int NV, bool VEC>
__global__ __launch_bounds__(256) void segment_fixup_long_kernel(const RedPack P, const typename Policy::Args args,
                                                                 const unsigned* __restrict__ keys,
                                                                 const unsigned* __restrict__ vals,
                                                                 const float* __restrict__ head,
                                                                 const float* __restrict__ tail,
                                                                 const int* __restrict__ flags,
                                                                 unsigned* __restrict__ fin, const int max_dim,
                                                                 const int sum_stride, const unsigned n_chunks,
                                                                 const unsigned chunk, float* __restrict__ lpart) {
  using F = Frag<G, NV, VEC>;
  constexpr int NGB = 256 / G;          // lane groups per workgroup
  // chunks per lane group and step: a window of ~512 chunks whatever G is (independent loads, few barriers)
  constexpr int R = (2 * G < 8) ? 8 : ((2 * G > 32) ? 32 : 2 * G);
  constexpr int NA = NV * F::W;
  constexpr int kNone = 1 << 30;
  __shared__ int s_stop[4];
  __shared__ float s_part[4][G * NA + 1];
  __shared__ unsigned s_last;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int gi = threadIdx.x / G, lane_g = threadIdx.x % G;
  const unsigned count = fin[1];
  unsigned KW = 1;
  if (count > 0 && count <= kLongSlots && count * 2 <= gridDim.x && lpart != nullptr) {
    KW = gridDim.x / count;
    KW = KW > kLongSplit ? kLongSplit : KW;
    if (count * KW > kLongSlots) KW = 1;                   // (one partial slot per work item)
  }
  unsigned* lcnt = fin + kFinList + 2 * static_cast<size_t>(n_chunks);
  const unsigned items = count * KW;
  // Workgroups without an item leave at once and are not waited for (round 6): the launch is sized for a batch WITH long
  // chains (Zipf-like ids: ~300 of them at the Criteo shape, which 32 workgroups walked ten in a row -- 56 us at the end of
  // the step's critical chain) and costs a batch without any what one workgroup costs.  (A workgroup that reads the counters
  // after the reset below sees no chains and leaves: the reset happens only after every workgroup WITH an item has arrived.)
  const unsigned needed = items == 0 ? 1u : (items < gridDim.x ? items : gridDim.x);
  if (blockIdx.x >= needed) return;
  for (unsigned item = blockIdx.x; item < items; item += gridDim.x) {
    const unsigned idx = item / KW, jw = item % KW;
    const unsigned c = fin[kFinList + n_chunks + idx];
    const unsigned s = c * chunk;
    const unsigned key = keys[s];
    // how many of the KW workgroups this chain is worth: one per two windows (a chain of a few hundred chunks -- a table
    // of three rows at B = 65 536 -- stays with one workgroup: partial slots, fences and the arrival counter cost more
    // than its one or two window steps).  Its first chunk: lower bound over the chunks' last keys.
    unsigned kw = 1;
    if (KW > 1) {
      if (threadIdx.x == 0) {
        unsigned lo = 0, hi = c;
        while (lo < hi) {
          const unsigned mid = (lo + hi) >> 1;
          if (keys[(static_cast<size_t>(mid) + 1) * chunk - 1] >= key) hi = mid;
          else lo = mid + 1;
        }
        unsigned w = ((c - lo + NGB * R - 1) / (NGB * R)) / 2;
        s_last = w < 1 ? 1u : (w > KW ? KW : w);
      }
      __syncthreads();
      kw = s_last;
      __syncthreads();
      if (jw >= kw) continue;
    }
    const RedField fd = P.f[vals[s] >> kLocalBits];
    F acc, pre;
    acc.zero();
    pre.zero();
    float cnt = 0.f;
    if (gi == 0 && jw == 0) {
      if (kw == 1) Policy::prefetch(args, fd, key - fd.row_base, lane_g, pre);
      const float* src = head + static_cast<size_t>(c) * sum_stride;
      acc.add_from(src, fd.dim, lane_g);
      if (Policy::kHasCount) cnt = src[max_dim];
    }
    long long jbase = static_cast<long long>(c) - 1 - static_cast<long long>(jw) * (NGB * R);
    bool adjoining = jw == 0;                 // the chain's own first window needs no test
    while (jbase >= 0) {
      // (with several workgroups per chain a window does not follow from the previous one's flags: it belongs to this
      //  chain iff its nearest chunk ends in the chain's key)
      if (!adjoining && keys[(static_cast<size_t>(jbase) + 1) * chunk - 1] != key) break;
      adjoining = kw == 1;
      // lane group gi looks at the R chunks jbase - gi*R - r (r = 0..R-1): a window of NGB * R chunks per step
      int fl[R];
      int first_stop = kNone;                              // window distance of the first chunk that ends the walk
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const long long j = jbase - (static_cast<long long>(gi) * R + r);
        fl[r] = (j >= 0) ? flags[j] : 0;
      }
#pragma unroll
      for (int r = R - 1; r >= 0; --r) {
        const long long j = jbase - (static_cast<long long>(gi) * R + r);
        if (j < 0 || !(fl[r] & kFlagPass)) first_stop = gi * R + r;
      }
      int wmin = first_stop;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int other = __shfl_xor(wmin, o, 64);
        wmin = other < wmin ? other : wmin;
      }
      if (lane == 0) s_stop[wid] = wmin;
      __syncthreads();
      int t = s_stop[0];
#pragma unroll
      for (int w = 1; w < 4; ++w) t = (s_stop[w] < t) ? s_stop[w] : t;   // distance of the chunk where the run started
      F part;
      part.zero();
      float pc = 0.f;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int d = gi * R + r;
        const long long j = jbase - d;
        if (j >= 0 && d <= t && !(fl[r] & kFlagZero)) {
          const float* src = tail + static_cast<size_t>(j) * sum_stride;
          part.add_from(src, fd.dim, lane_g);
          if (Policy::kHasCount) pc += src[max_dim];
        }
      }
#pragma unroll
      for (int o = G; o < 64; o <<= 1) {
#pragma unroll
        for (int q = 0; q < NA; ++q) part.a[q] += __shfl_xor(part.a[q], o, 64);
        pc += __shfl_xor(pc, o, 64);
      }
      if (lane < G) {
#pragma unroll
        for (int q = 0; q < NA; ++q) s_part[wid][lane * NA + q] = part.a[q];
        if (lane == 0) s_part[wid][G * NA] = pc;
      }
      __syncthreads();
      if (gi == 0) {
#pragma unroll
        for (int w = 0; w < 4; ++w) {
#pragma unroll
          for (int q = 0; q < NA; ++q) acc.a[q] += s_part[w][lane_g * NA + q];
          cnt += s_part[w][G * NA];
        }
      }
      __syncthreads();
      if (t != kNone) break;
      jbase -= static_cast<long long>(kw) * (NGB * R);
    }
    if (kw == 1) {
      if (gi == 0) Policy::flush(args, fd, key - fd.row_base, acc, cnt, pre, lane_g);
      continue;
    }
    // several workgroups share this chain: leave the partial, the last one to arrive finishes the row
    if (gi == 0) {
      float* slot = lpart + static_cast<size_t>(item) * sum_stride;
      acc.store(slot, fd.dim, lane_g);
      if (Policy::kHasCount && lane_g == 0) slot[max_dim] = cnt;
    }
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) s_last = (atomicAdd(lcnt + idx, 1u) == kw - 1) ? 1u : 0u;
    __syncthreads();
    if (s_last) {
      __threadfence();
      if (gi == 0) {
        F tot;
        tot.zero();
        float tc = 0.f;
        Policy::prefetch(args, fd, key - fd.row_base, lane_g, pre);
        for (unsigned w = 0; w < kw; ++w) {                 // slot order: fixed
          const float* slot = lpart + (static_cast<size_t>(idx) * KW + w) * sum_stride;
          tot.add_from(slot, fd.dim, lane_g);
          if (Policy::kHasCount) tc += slot[max_dim];
        }
        Policy::flush(args, fd, key - fd.row_base, tot, tc, pre, lane_g);
      }
    }
    __syncthreads();
  }
  // the last workgroup to get here clears the counters: the workspace is ready for another backward on the same
  // sorted ids (build_keys clears them for a new sort), and no memset node sits between the sort and the reduce
  // (no fence: every workgroup has consumed the counters -- its loop bounds -- before it arrives here, and nothing but the
  //  counters themselves is handed between workgroups; a __threadfence() per workgroup, an L2 write-back each on this
  //  8-XCD part, made the EMPTY launch of a step without long runs cost 12-17 us)
  __syncthreads();
  if (threadIdx.x == 0) {
    if (atomicAdd(fin + 2, 1u) == needed - 1) {
      fin[0] = 0;
      fin[1] = 0;
      fin[2] = 0;
    }
  }
}

template <class Policy, int G, int NV, bool VEC>
static int launch_reduce(const BwdPlan& p, const typename Policy::Args& args, const unsigned* keys,
                         const unsigned* vals, char* ws, hipStream_t s) {
  const int groups_per_block = 256 / G;
  const unsigned blocks = (p.n_chunks + groups_per_block - 1) / groups_per_block;
  float* head = reinterpret_cast<float*>(ws + p.off_head);
  float* tail = reinterpret_cast<float*>(ws + p.off_tail);
  int* flags = reinterpret_cast<int*>(ws + p.off_flags);
  unsigned* fin = reinterpret_cast<unsigned*>(ws + p.off_fin);
  // fin[0..2] are zero here: build_keys clears them for a new sort, the long fix-up kernel when it is done
  hipLaunchKernelGGL((segment_reduce_kernel<Policy, G, NV, VEC>), dim3(blocks), dim3(256), 0, s, p.red, p.n_cat, args,
                     keys, vals, p.n_lookups, p.total_rows, head, tail, flags, fin, p.max_dim, p.sum_stride,
                     p.n_chunks, static_cast<unsigned>(p.chunk));
  int rc = check_launch("segment_reduce_kernel");
  if (rc != RBX_OK) return rc;
  unsigned short_blocks = blocks;                           // one lane group per finalising chunk, grid-stride
  if (short_blocks > static_cast<unsigned>(kCUs * 8)) short_blocks = kCUs * 8;
  hipLaunchKernelGGL((segment_fixup_short_kernel<Policy, G, NV, VEC>), dim3(short_blocks), dim3(256), 0, s, p.red, args,
                     keys, vals, head, tail, flags, fin, p.max_dim, p.sum_stride, p.n_chunks,
                     static_cast<unsigned>(p.chunk));
  unsigned long_blocks = p.n_chunks;                        // one workgroup per long chain, grid-stride
  if (long_blocks > p.long_cap) long_blocks = p.long_cap;
  hipLaunchKernelGGL((segment_fixup_long_kernel<Policy, G, NV, VEC>), dim3(long_blocks), dim3(256), 0, s, p.red, args,
                     keys, vals, head, tail, flags, fin, p.max_dim, p.sum_stride, p.n_chunks,
                     static_cast<unsigned>(p.chunk), reinterpret_cast<float*>(ws + p.off_long));
  return check_launch("segment_fixup kernels");
}

template <class Policy, bool VEC>
static int dispatch_reduce(const BwdPlan& p, const typename Policy::Args& args, const unsigned* keys,
                           const unsigned* vals, char* ws, hipStream_t s) {
  const int units = VEC ? p.max_dim / 4 : p.max_dim;
  switch (pow2_ceil(units)) {
    case 1: return launch_reduce<Policy, 1, 1, VEC>(p, args, keys, vals, ws, s);
    case 2: return launch_reduce<Policy, 2, 1, VEC>(p, args, keys, vals, ws, s);
    case 4: return launch_reduce<Policy, 4, 1, VEC>(p, args, keys, vals, ws, s);
    case 8: return launch_reduce<Policy, 8, 1, VEC>(p, args, keys, vals, ws, s);
    case 16: return launch_reduce<Policy, 16, 1, VEC>(p, args, keys, vals, ws, s);
    case 32: return launch_reduce<Policy, 32, 1, VEC>(p, args, keys, vals, ws, s);
    case 64: return launch_reduce<Policy, 64, 1, VEC>(p, args, keys, vals, ws, s);
    case 128: return launch_reduce<Policy, 64, 2, VEC>(p, args, keys, vals, ws, s);
    case 256: return launch_reduce<Policy, 64, 4, VEC>(p, args, keys, vals, ws, s);
    default: return fail(RBX_ERR_UNSUPPORTED, "embedding dim too large for one lane group");
  }
}

// ---- re-zero of the gradient rows a previous step wrote ----------------------------------------------
// The sorted (key, val) pairs of the PREVIOUS sort on this workspace name every gradient row that step's
// backward stored to: one lane group per pair, the head of each run of equal keys clears its row (dim floats of
// dW, one float of the LR gradient).  36 MB of stores at the Criteo shape instead of a 379 MB fill of the dense grads.
template <bool VEC>
__global__ __launch_bounds__(256) void rezero_rows_kernel(const RedPack P, const int n_cat, const unsigned* __restrict__ keys,
                                                        const unsigned* __restrict__ vals, const unsigned n,
                                                        const unsigned sentinel, const int lanes) {
  __shared__ RedField sf[RBX_MAX_FIELDS];
  {
    const int words = n_cat * static_cast<int>(sizeof(RedField) / 4);
    const int* src = reinterpret_cast<const int*>(&P);
    int* dst = reinterpret_cast<int*>(sf);
    for (int i = threadIdx.x; i < words; i += blockDim.x) dst[i] = src[i];
  }
  __syncthreads();
  typedef float v4f __attribute__((ext_vector_type(4)));
  constexpr int W = VEC ? 4 : 1;
  // Phase 1, a thread per sorted pair: three independent coalesced loads decide whether the pair starts a run and
  // where its row lives.  Phase 2, a lane group per pair: the wave walks its 64 pairs `64 / lanes` at a time and a
  // row is cleared by ONE coalesced store of its group.  (One lane group per pair from the start, with the loads
  // chained behind each other, was latency-bound at 41 us; a thread per pair storing 4 x 16 B took 83 us.)
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  float* row_ptr = nullptr;
  int dim = 0;
  if (i < n) {
    const unsigned key = keys[i];
    const unsigned prev = (i > 0) ? keys[i - 1] : sentinel;
    const unsigned val = vals[i];
    if (key < sentinel && key != prev) {
      const RedField& fd = sf[val >> kLocalBits];
      const size_t row = key - fd.row_base;
      if (fd.grad != nullptr) {
        row_ptr = fd.grad + row * fd.dim;
        dim = fd.dim;
      }
      if (fd.grad2 != nullptr) fd.grad2[row] = 0.f;
    }
  }
  const int lane = threadIdx.x & 63;
  const int lane_g = lane % lanes, group = lane / lanes, per_step = 64 / lanes;
  for (int base = 0; base < 64; base += per_step) {
    const int src = base + group;
    float* dst = reinterpret_cast<float*>(__shfl(reinterpret_cast<unsigned long long>(row_ptr), src, 64));
    const int d = __shfl(dim, src, 64);
    for (int e = lane_g * W; e < d; e += lanes * W) {
      if constexpr (VEC) {
        const v4f z = {0.f, 0.f, 0.f, 0.f};
        __builtin_nontemporal_store(z, reinterpret_cast<v4f*>(dst + e));
      } else {
        dst[e] = 0.f;
      }
    }
  }
}


// launch helper of rbx_fm_rezero / rbx_embed_rezero (any plan built by make_plan / fm_plan: the kernel reads the RedPack only)
static inline int launch_rezero(const BwdPlan& p, const char* ws, hipStream_t s) {
  const int cur = p.passes & 1;
  const unsigned* keys = reinterpret_cast<const unsigned*>(ws + p.off_keys[cur]);
  const unsigned* vals = reinterpret_cast<const unsigned*>(ws + p.off_vals[cur]);
  const int width = p.vec ? (p.max_dim + 3) / 4 : p.max_dim;
  int lanes = 1;
  while (lanes < width && lanes < 64) lanes *= 2;
  const unsigned blocks = (p.n_lookups + 255) / 256;
  if (p.vec)
    hipLaunchKernelGGL(rezero_rows_kernel<true>, dim3(blocks), dim3(256), 0, s, p.red, p.n_cat, keys, vals, p.n_lookups,
                       p.total_rows, lanes);
  else
    hipLaunchKernelGGL(rezero_rows_kernel<false>, dim3(blocks), dim3(256), 0, s, p.red, p.n_cat, keys, vals, p.n_lookups,
                       p.total_rows, lanes);
  return check_launch("rezero_rows_kernel");
}

}  // namespace rbx
