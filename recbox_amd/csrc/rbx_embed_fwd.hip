// rbx_embed_fwd.hip -- K1/K2: multi-table embedding gather with fused sequence
// pooling, one launch over all features of a layer (gfx950).
//
// Reference behaviour replaced (paths relative to /root/reference/recbox):
//   core/pytorch/layers/embedding.py:116-138, ranking/pytorch/layers/embeddings/
//   feature_embedding.py:188-214 + dict2tensor :169-186, third_party/rechub/basic/
//   layers.py:66-116 with the pooling variants of core/.../sequence.py:8-20,
//   ranking/.../pooling.py:26-40 and rechub/basic/layers.py:187-230.
//
// Mapping.  A (sample, field) pair is served by a lane group of G = dim/4 lanes
// (float4 per lane; G lanes x 1 float on the scalar path), so a 64-wide wavefront
// keeps 64/G independent row reads in flight per load instruction.  Pairs are
// numbered p = b*F + f, field fastest: consecutive groups write consecutive
// slots of the [B, F*D] output row -> fully coalesced stores; the random traffic
// is the table rows only.  Descriptors arrive through the kernarg segment and are
// mirrored into LDS once per workgroup because `f` diverges between lane groups.
// The simple (one id per sample) path is unrolled x4 so that four dependent
// id->row chains per lane are in flight; sequence pools unroll over L instead.
// HBM-bound: no LDS tiling of rows (each row is used once), no MFMA.
#include "rbx_internal.h"

namespace rbx {

template <bool VEC>
struct Acc;
template <>
struct Acc<true> {
  float4 v;
  __device__ __forceinline__ void zero() { v = make_float4(0.f, 0.f, 0.f, 0.f); }
  __device__ __forceinline__ void add(const Acc& o) { v.x += o.v.x; v.y += o.v.y; v.z += o.v.z; v.w += o.v.w; }
  __device__ __forceinline__ void scale(float s) { v.x *= s; v.y *= s; v.z *= s; v.w *= s; }
  __device__ __forceinline__ float hsum() const { return (v.x + v.y) + (v.z + v.w); }
  __device__ __forceinline__ void load(const float* p) { v = *reinterpret_cast<const float4*>(p); }
  __device__ __forceinline__ void store(float* p) const { *reinterpret_cast<float4*>(p) = v; }
  __device__ __forceinline__ void xor_add(int o) {
    v.x += __shfl_xor(v.x, o, 64); v.y += __shfl_xor(v.y, o, 64);
    v.z += __shfl_xor(v.z, o, 64); v.w += __shfl_xor(v.w, o, 64);
  }
};
template <>
struct Acc<false> {
  float v;
  __device__ __forceinline__ void zero() { v = 0.f; }
  __device__ __forceinline__ void add(const Acc& o) { v += o.v; }
  __device__ __forceinline__ void scale(float s) { v *= s; }
  __device__ __forceinline__ float hsum() const { return v; }
  __device__ __forceinline__ void load(const float* p) { v = *p; }
  __device__ __forceinline__ void store(float* p) const { *p = v; }
  __device__ __forceinline__ void xor_add(int o) { v += __shfl_xor(v, o, 64); }
};

// One lane's share of a row: NV units, unit u covers elements [(lane_g + u*G)*W, +W).
template <int G, int NV, bool VEC>
struct RowFrag {
  static constexpr int W = VEC ? 4 : 1;
  Acc<VEC> a[NV];
  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int u = 0; u < NV; ++u) a[u].zero();
  }
  __device__ __forceinline__ void load(const float* row, int dim, int lane_g) {
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const int e = (lane_g + u * G) * W;
      if (e < dim) a[u].load(row + e); else a[u].zero();
    }
  }
  __device__ __forceinline__ void store(float* row, int dim, int lane_g) const {
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const int e = (lane_g + u * G) * W;
      if (e < dim) a[u].store(row + e);
    }
  }
  __device__ __forceinline__ void add(const RowFrag& o) {
#pragma unroll
    for (int u = 0; u < NV; ++u) a[u].add(o.a[u]);
  }
  __device__ __forceinline__ void scale(float s) {
#pragma unroll
    for (int u = 0; u < NV; ++u) a[u].scale(s);
  }
  __device__ __forceinline__ float hsum() const {
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < NV; ++u) s += a[u].hsum();
    return s;
  }
  __device__ __forceinline__ void xor_add(int o) {
#pragma unroll
    for (int u = 0; u < NV; ++u) a[u].xor_add(o);
  }
};

template <int G, int NV, bool VEC>
__global__ __launch_bounds__(256) void embed_fwd_kernel(const FieldPack P, const int F, const long long B,
                                                        float* __restrict__ out, const long long stride_b,
                                                        float* __restrict__ row_scale,
                                                        int* __restrict__ status) {
  __shared__ FieldK sf[RBX_MAX_FIELDS];
  {
    const int words = F * static_cast<int>(sizeof(FieldK) / 4);
    const int* src = reinterpret_cast<const int*>(&P);
    int* dst = reinterpret_cast<int*>(sf);
    for (int i = threadIdx.x; i < words; i += blockDim.x) dst[i] = src[i];
  }
  __syncthreads();

  using Frag = RowFrag<G, NV, VEC>;
  const int lane_g = threadIdx.x % G;
  const long long ngroups = static_cast<long long>(gridDim.x) * (blockDim.x / G);
  const long long npairs = B * F;
  constexpr int U = 4;  // independent pairs in flight per lane group
  const bool small = npairs < (1ll << 32);

  for (long long p0 = static_cast<long long>(blockIdx.x) * (blockDim.x / G) + threadIdx.x / G; p0 < npairs;
       p0 += ngroups * U) {
    // ---- phase 1: ids / values of up to U pairs -------------------------------
    long long bb[U];
    int ff[U];
    long long id0[U];
    bool simple[U], live[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long p = p0 + u * ngroups;
      live[u] = p < npairs;
      const long long pc = live[u] ? p : 0;
      if (small) {   // 32-bit divide: the 64-bit one costs ~100 VALU ops per pair
        const unsigned q = static_cast<unsigned>(pc) / static_cast<unsigned>(F);
        bb[u] = q;
        ff[u] = static_cast<int>(static_cast<unsigned>(pc) - q * static_cast<unsigned>(F));
      } else {
        bb[u] = pc / F;
        ff[u] = static_cast<int>(pc - bb[u] * F);
      }
      const FieldK& fd = sf[ff[u]];
      simple[u] = live[u] && fd.kind == RBX_FIELD_CATEGORICAL && fd.pool == RBX_POOL_NONE;
      id0[u] = simple[u] ? load_id(fd.ids, bb[u] * fd.ids_stride_b, fd.ids_dtype) : 0;
    }
    // ---- phase 2: row reads of the simple pairs -------------------------------
    Frag row[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      row[u].zero();
      if (simple[u]) {
        const FieldK& fd = sf[ff[u]];
        if (id0[u] >= 0 && id0[u] < fd.vocab) {
          row[u].load(fd.table + id0[u] * fd.dim, fd.dim, lane_g);
        } else if (status != nullptr) {
          atomicOr(status, 1);
        }
      }
    }
    // ---- phase 3: stores; everything that is not a plain lookup ---------------
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (!live[u]) continue;
      const FieldK& fd = sf[ff[u]];
      float* dst = out + bb[u] * stride_b + fd.out_off;
      const int dim = fd.dim;
      if (simple[u]) {
        row[u].store(dst, dim, lane_g);
        continue;
      }
      if (fd.kind == RBX_FIELD_DENSE) {
        if (lane_g == 0) dst[0] = load_value(fd.ids, bb[u] * fd.ids_stride_b, fd.ids_dtype);
        continue;
      }
      if (fd.kind == RBX_FIELD_NUMERIC) {
        const float x = load_value(fd.ids, bb[u] * fd.ids_stride_b, fd.ids_dtype);
        Frag w;
        w.load(fd.table, dim, lane_g);
        w.scale(x);
        w.store(dst, dim, lane_g);
        continue;
      }
      // sequence features never reach this kernel (embed_seq_kernel serves them)
    }
  }
}

// Sequence features (multi-hot): one lane group serves ONE (sample, feature) pair and pools in
// registers, so [B, L, D] never reaches HBM.  Kept apart from the one-hot kernel so that neither pays
// the other's registers.  Padded histories are half padding on average, so a chunk of the sequence is
// handled in two steps: (1) the G lanes load their ids in one coalesced sweep, classify them (pad /
// masked / out of range) and COMPACT the surviving (id, position) pairs into a per-group LDS list with a
// ballot + popcount prefix; (2) the list is walked U rows at a time, every load slot doing useful work
// and no id latency between row batches.
// Narrow rows (dim/4 < 16 lanes): the group is still 16 lanes wide and splits into R sub-groups of
// W = G/R lanes, sub-group r pooling list entries r, r+R, ...; the partial pools meet in a shuffle
// butterfly at the end.  This shortens the serial chain per sample R-fold, makes the id sweep read 64+
// contiguous bytes per sample and evens out ragged lengths inside a wave.  Summation order is fixed by
// (R, list order) -> deterministic, and equal to sequence order when R == 1.
template <int G, int R, int NV, bool VEC>
#ifndef RBX_SEQ_WAVES
#define RBX_SEQ_WAVES 4
#endif
#define RBX_SEQ_SG16 32
#ifndef RBX_SEQ_SG32
#define RBX_SEQ_SG32 64
#endif
#ifndef RBX_SEQ_U
#define RBX_SEQ_U 4
#endif
__global__ __launch_bounds__(256, RBX_SEQ_WAVES) void embed_seq_kernel(const FieldPack P, const int F, const long long B,
                                                        float* __restrict__ out, const long long stride_b,
                                                        float* __restrict__ row_scale,
                                                        int* __restrict__ status) {
  constexpr int W = G / R;                                // lanes that hold one row
  using Frag = RowFrag<W, NV, VEC>;
  constexpr int U = (NV * (VEC ? 4 : 1) <= 4) ? RBX_SEQ_U : 4;   // rows in flight per lane
#define RBX_SEQ_IPL16 4
  constexpr int IPL = (G >= 32) ? 4 : RBX_SEQ_IPL16;                  // ids per lane per chunk (chunk = 128..256 lookups)
  constexpr int C = G * IPL;                              // lookups per chunk
  constexpr int GPB = 256 / G;
  __shared__ int s_id[GPB][C];
  __shared__ int s_pos[GPB][C];
  const int lane_w = threadIdx.x % W;
  const int sub = (threadIdx.x % G) / W;
  const int lane_g = threadIdx.x % G;
  const int gidx = threadIdx.x / G;
  const int gshift = (threadIdx.x & 63) & ~(G - 1);       // first lane of the group inside its wave
  const unsigned long long gmask = (G == 64) ? ~0ull : ((1ull << G) - 1ull);
  const unsigned long long below = (1ull << lane_g) - 1ull;
  volatile int* my_id = s_id[gidx];
  volatile int* my_pos = s_pos[gidx];
  // A wavefront works on ONE feature: wave task t = (feature, block of 64/G samples).  The descriptor is
  // then wave-uniform -- it lives in SGPRs, straight from the kernarg segment, instead of 14 VGPRs.
  constexpr int GPW = 64 / G;
  const long long tasks_per_field = (B + GPW - 1) / GPW;
  const long long ntasks = tasks_per_field * F;
  const long long nwaves = static_cast<long long>(gridDim.x) * 4;
  for (long long t = static_cast<long long>(blockIdx.x) * 4 + threadIdx.x / 64; t < ntasks; t += nwaves) {
    const int f = __builtin_amdgcn_readfirstlane(static_cast<int>(t / tasks_per_field));
    const long long b = (t - f * tasks_per_field) * GPW + (threadIdx.x & 63) / G;
    const bool alive = b < B;
    const FieldK& fd = P.f[f];
    const int L = alive ? fd.seq_len : 0, pool = fd.pool, dim = fd.dim, dt = fd.ids_dtype;
    const bool id_pool = (pool == RBX_POOL_MEAN_ID || pool == RBX_POOL_SUM_ID);
    float* dst = out + b * stride_b + fd.out_off;
    const long long base = b * fd.ids_stride_b;
    Frag acc;
    acc.zero();
    float count = 0.f;
    // wave-uniform number of chunks
    int Lmax = L;
#pragma unroll
    for (int o = 32; o >= G && o > 0; o >>= 1) {
      const int other = __shfl_xor(Lmax, o, 64);
      Lmax = other > Lmax ? other : Lmax;
    }
    for (int c0 = 0; c0 < Lmax; c0 += C) {
      long long raw[IPL];
#pragma unroll
      for (int i = 0; i < IPL; ++i) {                     // (1a) one coalesced sweep of id loads
        const int l = c0 + i * G + lane_g;
        raw[i] = (l < L) ? load_raw(fd.ids, base + static_cast<long long>(l) * fd.ids_stride_l, dt) : 0;
      }
      int nvalid = 0;
#pragma unroll
      for (int i = 0; i < IPL; ++i) {                     // (1b) classify + compact
        const int l = c0 + i * G + lane_g;
        const long long id = decode_id(raw[i], dt);
        const bool live = l < L;
        const bool in_range = id >= 0 && id < fd.vocab;
        if (live && !in_range && status != nullptr) atomicOr(status, 1);
        // id-masked rows get weight 0 in the reference's bmm: skip their traffic.  concat keeps every slot.
        const bool use = live && (pool == RBX_POOL_CONCAT || (in_range && !(id_pool && id == fd.mask_id)));
        const unsigned long long m = (__ballot(use) >> gshift) & gmask;
        if (use) {
          const int k = nvalid + __popcll(m & below);
          my_id[k] = in_range ? static_cast<int>(id) : -1;
          my_pos[k] = l;
        }
        nvalid += __popcll(m);
      }
      __builtin_amdgcn_wave_barrier();
      int nmax = nvalid;                                   // wave-uniform batch count
#pragma unroll
      for (int o = 32; o >= G && o > 0; o >>= 1) {
        const int other = __shfl_xor(nmax, o, 64);
        nmax = other > nmax ? other : nmax;
      }
      // (2) dense row batches.  U = 4 rows per lane in flight: deeper batches (8) or software pipelining
      // cost registers, i.e. resident waves, and measured slower at every row width (profiles/r01_kernels_bench.txt)
      for (int k0 = 0; k0 < nmax; k0 += U * R) {
        int idu[U], lu[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int k = k0 + u * R + sub;
          idu[u] = (k < nvalid) ? my_id[k] : -1;
          lu[u] = (k < nvalid) ? my_pos[k] : -1;
        }
        Frag r[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          r[u].zero();
          if (idu[u] >= 0) r[u].load(fd.table + static_cast<long long>(idu[u]) * dim, dim, lane_w);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (pool == RBX_POOL_CONCAT) {
            if (lu[u] >= 0) r[u].store(dst + static_cast<long long>(lu[u]) * dim, dim, lane_w);
          } else {
            acc.add(r[u]);
            if (pool == RBX_POOL_MEAN_VALUE) {
              const float sm = group_sum<W>(r[u].hsum());   // value mask: row sum != 0
              count += (sm != 0.f) ? 1.f : 0.f;
            } else if (pool == RBX_POOL_MEAN_ID) {
              count += (lu[u] >= 0) ? 1.f : 0.f;
            }
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    if (pool != RBX_POOL_CONCAT) {                         // wave-uniform: every lane joins the butterfly
#pragma unroll
      for (int o = W; o < G; o <<= 1) {
        acc.xor_add(o);
        count += __shfl_xor(count, o, 64);
      }
    }
    if (!alive || pool == RBX_POOL_CONCAT || sub != 0) continue;
    if (pool == RBX_POOL_MEAN_VALUE || pool == RBX_POOL_MEAN_ID) {
      const float inv = 1.0f / (count + fd.eps);
      // the reference divides; x * (1/(c+eps)) differs from x / (c+eps) by <= 1 ulp
      acc.scale(inv);
      if (row_scale != nullptr && lane_w == 0) row_scale[static_cast<long long>(fd.slot) * B + b] = inv;
    }
    acc.store(dst, dim, lane_w);
  }
}

template <int G, int NV, bool VEC>
static int launch_fwd(bool seq, const FieldPack& pack, int F, int64_t B, float* out, int64_t stride_b,
                      float* row_scale, int* status, hipStream_t s) {
  const long long npairs = B * F;
  constexpr int SG = (G <= 8) ? 16 : ((G == 16) ? RBX_SEQ_SG16 : ((G == 32) ? RBX_SEQ_SG32 : 64));   // lanes per (sample, sequence feature)
  const int groups_per_block = 256 / (seq ? SG : G);
  long long blocks = (npairs + groups_per_block - 1) / groups_per_block;
  if (seq) blocks = ((B + 64 / SG - 1) / (64 / SG) * F + 3) / 4;          // 4 wave tasks per workgroup
#define RBX_FWD_BLOCKS_PER_CU 32   // 8 left the one-id gather latency-bound: [B,39,16] at B=65536 took 130 us, 100 us with 32 (64: 105)
  const long long cap = static_cast<long long>(kCUs) * (seq ? 64 : RBX_FWD_BLOCKS_PER_CU);   // sequences: one pair per group, many waves
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  if (seq)
    hipLaunchKernelGGL((embed_seq_kernel<SG, SG / G, NV, VEC>), dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s, pack, F,
                       static_cast<long long>(B), out, static_cast<long long>(stride_b), row_scale, status);
  else
    hipLaunchKernelGGL((embed_fwd_kernel<G, NV, VEC>), dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s, pack, F,
                       static_cast<long long>(B), out, static_cast<long long>(stride_b), row_scale, status);
  return check_launch("embed_fwd_kernel");
}

template <bool VEC>
static int dispatch_fwd(bool seq, int units, const FieldPack& pack, int F, int64_t B, float* out, int64_t stride_b,
                        float* row_scale, int* status, hipStream_t s) {
  const int g = pow2_ceil(units);
  switch (g) {
    case 1: return launch_fwd<1, 1, VEC>(seq, pack, F, B, out, stride_b, row_scale, status, s);
    case 2: return launch_fwd<2, 1, VEC>(seq, pack, F, B, out, stride_b, row_scale, status, s);
    case 4: return launch_fwd<4, 1, VEC>(seq, pack, F, B, out, stride_b, row_scale, status, s);
    case 8: return launch_fwd<8, 1, VEC>(seq, pack, F, B, out, stride_b, row_scale, status, s);
    case 16: return launch_fwd<16, 1, VEC>(seq, pack, F, B, out, stride_b, row_scale, status, s);
    case 32: return launch_fwd<32, 1, VEC>(seq, pack, F, B, out, stride_b, row_scale, status, s);
    case 64: return launch_fwd<64, 1, VEC>(seq, pack, F, B, out, stride_b, row_scale, status, s);
    case 128: return launch_fwd<64, 2, VEC>(seq, pack, F, B, out, stride_b, row_scale, status, s);
    case 256: return launch_fwd<64, 4, VEC>(seq, pack, F, B, out, stride_b, row_scale, status, s);
    default: return fail(RBX_ERR_UNSUPPORTED, "embedding dim too large for one lane group (units=%d)", units);
  }
}

static bool field_vec_ok(const rbx_field_t& f, const float* out, int64_t stride_b) {
  if (f.kind == RBX_FIELD_DENSE) return false;
  if (f.dim % 4 != 0 || f.out_off % 4 != 0 || stride_b % 4 != 0) return false;
  if ((reinterpret_cast<uintptr_t>(f.table) & 15) != 0) return false;
  if ((reinterpret_cast<uintptr_t>(out) & 15) != 0) return false;
  return true;
}

}  // namespace rbx

extern "C" int rbx_embed_fwd(const rbx_field_t* fields, int32_t n_fields, int64_t batch, float* d_out,
                             int64_t out_stride_b, float* d_row_scale, int32_t* d_status, void* stream) {
  using namespace rbx;
  if (n_fields <= 0 || n_fields > RBX_MAX_FIELDS) return fail(RBX_ERR_INVALID, "n_fields=%d out of range", n_fields);
  if (batch < 0) return fail(RBX_ERR_INVALID, "negative batch");
  if (batch == 0) return RBX_OK;
  if (d_out == nullptr) return fail(RBX_ERR_INVALID, "d_out is NULL");
  // split into (float4 | scalar) x (one id per sample | sequence) launches; slot = position in the caller's array
  static thread_local rbx_field_t part[4][RBX_MAX_FIELDS];
  int slot[4][RBX_MAX_FIELDS];
  int cnt[4] = {0, 0, 0, 0};
  int units[4] = {1, 1, 1, 1};
  for (int i = 0; i < n_fields; ++i) {
    const bool vec = field_vec_ok(fields[i], d_out, out_stride_b);
    const bool seq = fields[i].kind == RBX_FIELD_CATEGORICAL && (fields[i].seq_len > 1 || fields[i].pool != RBX_POOL_NONE);
    const int k = (vec ? 0 : 1) + (seq ? 2 : 0);
    part[k][cnt[k]] = fields[i];
    slot[k][cnt[k]] = i;
    const int u = vec ? fields[i].dim / 4 : fields[i].dim;
    if (u > units[k]) units[k] = u;
    ++cnt[k];
  }
  for (int k = 0; k < 4; ++k) {
    if (cnt[k] == 0) continue;
    FieldPack pack;
    int rc = pack_fields(part[k], cnt[k], batch, false, &pack);
    if (rc != RBX_OK) return rc;
    for (int i = 0; i < cnt[k]; ++i) pack.f[i].slot = static_cast<unsigned char>(slot[k][i]);
    const bool seq = k >= 2;
    rc = (k % 2 == 0) ? dispatch_fwd<true>(seq, units[k], pack, cnt[k], batch, d_out, out_stride_b, d_row_scale, d_status,
                                           as_stream(stream))
                      : dispatch_fwd<false>(seq, units[k], pack, cnt[k], batch, d_out, out_stride_b, d_row_scale,
                                            d_status, as_stream(stream));
    if (rc != RBX_OK) return rc;
  }
  return RBX_OK;
}
