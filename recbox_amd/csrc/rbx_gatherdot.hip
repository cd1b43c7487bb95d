// rbx_gatherdot.hip -- K7: sampled-softmax / pairwise logits without materialising the
// candidate embeddings (gfx950).
//
// Reference op sequence replaced (paths relative to /root/reference/recbox):
//   pos_embed, neg_embed = item_emb(x, [pos, neg])         third_party/rechub/models/matching/sasrec.py:98-100
//   pos_logits = (seq_output * pos_embed).sum(-1)          sasrec.py:104
//   neg_logits = (seq_output * neg_embed).sum(-1)          sasrec.py:105
// and the same shape in the first-party sampled-softmax scoring (one positive + n negatives per row,
// core/pytorch/losses/softmax_crossentropy_loss.py:14-22 consumes the [rows, 1 + n] logits).
//
//   out[r, c] = scale * < x[r, :], W_c[ids_c[r], :] >
//
// The reference gathers [rows, n_out, D] candidate vectors (cfg 5: 2 x 4096 x 200 x 256 B = 420 MB
// written, read again by the product, and the same again for their gradient); here a lane group of
// D/4 lanes reads the table row and the x row, reduces over d with shuffles and writes 4 bytes.
// HBM-bound on the random row reads: algorithmic bytes per lookup = D*4 (row) + id + 4 (logit);
// x rows are shared by the n_out lookups of a row and stay in L2.
//
// Backward:  dW[id] += scale * g[r, c] * x[r, :]    -- the sorted segmented scatter-add of
//            rbx_segreduce.h with a policy whose contribution is g * x (no [rows, n_out, D] dE tensor);
//            dx[r, :] = scale * sum_c g[r, c] * W_c[ids_c[r], :]   -- a second gather pass.
#include "rbx_segreduce.h"

namespace rbx {

constexpr int kDotMaxOut = 256;

template <int G, int NV, bool VEC>
struct DotFrag {
  static constexpr int W = VEC ? 4 : 1;
  float a[NV * W];
  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int i = 0; i < NV * W; ++i) a[i] = 0.f;
  }
  __device__ __forceinline__ void load(const float* row, int dim, int lane_g) {
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const int e = (lane_g + u * G) * W;
      if (e < dim) {
        if constexpr (VEC) {
          const float4 t = *reinterpret_cast<const float4*>(row + e);
          a[u * 4] = t.x; a[u * 4 + 1] = t.y; a[u * 4 + 2] = t.z; a[u * 4 + 3] = t.w;
        } else {
          a[u] = row[e];
        }
      } else {
#pragma unroll
        for (int k = 0; k < W; ++k) a[u * W + k] = 0.f;
      }
    }
  }
  __device__ __forceinline__ float dot(const DotFrag& o) const {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV * W; ++i) s += a[i] * o.a[i];
    return s;
  }
  __device__ __forceinline__ void fma(const DotFrag& o, float w) {
#pragma unroll
    for (int i = 0; i < NV * W; ++i) a[i] += w * o.a[i];
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
};

struct DotCols {             // output column -> (candidate set, position inside it)
  unsigned char set[kDotMaxOut];
  unsigned char pos[kDotMaxOut];
};

// one lane group per row r: the x row is read once, the candidate rows of the row 4 at a time
template <int G, int NV, bool VEC>
__global__ __launch_bounds__(256) void gatherdot_fwd_kernel(const FieldPack P, const int F, const DotCols C, const long long R,
                                                            const int n_out, const float* __restrict__ x,
                                                            const long long xs, const float scale, const int D,
                                                            float* __restrict__ out, int* __restrict__ status) {
  __shared__ FieldK sf[RBX_MAX_FIELDS];
  {
    const int words = F * static_cast<int>(sizeof(FieldK) / 4);
    const int* src = reinterpret_cast<const int*>(&P);
    int* dst = reinterpret_cast<int*>(sf);
    for (int i = threadIdx.x; i < words; i += blockDim.x) dst[i] = src[i];
  }
  __syncthreads();
  using Frag = DotFrag<G, NV, VEC>;
  constexpr int U = (NV * (VEC ? 4 : 1) <= 4) ? 4 : 2;        // candidate rows in flight per lane group
  const int lane_g = threadIdx.x % G;
  const long long ngroups = static_cast<long long>(gridDim.x) * (blockDim.x / G);
  for (long long r = static_cast<long long>(blockIdx.x) * (blockDim.x / G) + threadIdx.x / G; r < R; r += ngroups) {
    Frag v;
    v.load(x + r * xs, D, lane_g);
    for (int c0 = 0; c0 < n_out; c0 += U) {
      long long id[U];
      const float* tab[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int c = c0 + u;
        id[u] = -1;
        tab[u] = nullptr;
        if (c < n_out) {
          const FieldK& fd = sf[C.set[c]];
          id[u] = load_id(fd.ids, r * fd.ids_stride_b + static_cast<long long>(C.pos[c]) * fd.ids_stride_l, fd.ids_dtype);
          if (id[u] < 0 || id[u] >= fd.vocab) {
            if (status != nullptr) atomicOr(status, 1);
            id[u] = -1;                                          // out-of-range lookups score 0
          }
          tab[u] = fd.table;
        }
      }
      Frag w[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        w[u].zero();
        if (id[u] >= 0) w[u].load(tab[u] + id[u] * D, D, lane_g);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const float s = group_sum<G>(w[u].dot(v));
        if (lane_g == 0 && c0 + u < n_out) out[r * n_out + c0 + u] = s * scale;
      }
    }
  }
}

// dx[r, :] = scale * sum_c g[r, c] * W_c[id_c[r], :]; one lane group per row, 4 candidate rows in flight
template <int G, int NV, bool VEC>
__global__ __launch_bounds__(256) void gatherdot_dx_kernel(const FieldPack P, const int F, const DotCols C, const long long R,
                                                           const int n_out, const float* __restrict__ g,
                                                           const float scale, const int D, float* __restrict__ dx,
                                                           const long long dxs) {
  __shared__ FieldK sf[RBX_MAX_FIELDS];
  {
    const int words = F * static_cast<int>(sizeof(FieldK) / 4);
    const int* src = reinterpret_cast<const int*>(&P);
    int* dst = reinterpret_cast<int*>(sf);
    for (int i = threadIdx.x; i < words; i += blockDim.x) dst[i] = src[i];
  }
  __syncthreads();
  using Frag = DotFrag<G, NV, VEC>;
  constexpr int U = (NV * (VEC ? 4 : 1) <= 4) ? 4 : 2;
  const int lane_g = threadIdx.x % G;
  const long long ngroups = static_cast<long long>(gridDim.x) * (blockDim.x / G);
  for (long long r = static_cast<long long>(blockIdx.x) * (blockDim.x / G) + threadIdx.x / G; r < R; r += ngroups) {
    Frag acc;
    acc.zero();
    for (int c0 = 0; c0 < n_out; c0 += U) {
      long long id[U];
      float wgt[U];
      const float* tab[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int c = c0 + u;
        id[u] = -1;
        wgt[u] = 0.f;
        tab[u] = nullptr;
        if (c < n_out) {
          const FieldK& fd = sf[C.set[c]];
          id[u] = load_id(fd.ids, r * fd.ids_stride_b + static_cast<long long>(C.pos[c]) * fd.ids_stride_l, fd.ids_dtype);
          if (id[u] < 0 || id[u] >= fd.vocab) id[u] = -1;
          wgt[u] = g[r * n_out + c];
          tab[u] = fd.table;
        }
      }
      Frag w[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        w[u].zero();
        if (id[u] >= 0) w[u].load(tab[u] + id[u] * D, D, lane_g);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) acc.fma(w[u], wgt[u] * scale);
    }
    acc.store(dx + r * dxs, D, lane_g);
  }
}

// ---- backward policy: a lookup (row r, column c) contributes scale * g[r, c] * x[r, :] ------------------
struct DotPolicy {
  static constexpr bool kHasCount = false;
  struct Args {
    const float* x;
    long long xs;
    const float* g;
    int n_out;
    float scale;
    int accumulate;
    int col0[RBX_MAX_FIELDS];      // first output column of candidate set i (indexed by RedField::slot)
  };
  template <class F>
  static __device__ __forceinline__ void contribute(const Args& a, const RedField& fd, unsigned local, int lane_g,
                                                    F& frag, float& cnt) {
    const unsigned L = static_cast<unsigned>(fd.seq_len);
    const unsigned r = local / L;
    const unsigned l = local - r * L;
    const float w = a.scale * a.g[static_cast<long long>(r) * a.n_out + a.col0[fd.slot] + l];
    frag.fma_from(a.x + static_cast<long long>(r) * a.xs, fd.dim, lane_g, w);
    (void)cnt;
  }
  template <class F>
  static __device__ __forceinline__ void prefetch(const Args& a, const RedField& fd, unsigned row, int lane_g, F& pre) {
    if (a.accumulate) pre.add_from(fd.grad + static_cast<size_t>(row) * fd.dim, fd.dim, lane_g);
  }
  // the two-phase form (segment_reduce_kernel): loads only, then frag *= weight(w)
  template <class F>
  static __device__ __forceinline__ void fetch(const Args& a, const RedField& fd, unsigned local, int lane_g, F& frag, float& w) {
    const unsigned L = static_cast<unsigned>(fd.seq_len);
    const unsigned r = local / L;
    const unsigned l = local - r * L;
    w = a.g[static_cast<long long>(r) * a.n_out + a.col0[fd.slot] + l];
    frag.load_from(a.x + static_cast<long long>(r) * a.xs, fd.dim, lane_g);
  }
  static __device__ __forceinline__ float weight(const Args& a, float w) { return a.scale * w; }
  template <class F>
  static __device__ __forceinline__ void prefetch_raw(const Args& a, const RedField& fd, unsigned row, int lane_g, F& pre) {
    if (a.accumulate) pre.load_from(fd.grad + static_cast<size_t>(row) * fd.dim, fd.dim, lane_g);
  }
  template <class F>
  static __device__ __forceinline__ void flush(const Args&, const RedField& fd, unsigned row, const F& acc, float,
                                               const F& pre, int lane_g) {
    F out = acc;
    frag_add(out, pre);
    out.store_nt(fd.grad + static_cast<size_t>(row) * fd.dim, fd.dim, lane_g);
  }
};

struct DotHost {
  FieldPack pack;
  DotCols cols;
  rbx_field_t plan_fields[RBX_MAX_FIELDS];     // copies with out_off = 0 (the column offset lives in DotPolicy::Args)
  int col0[RBX_MAX_FIELDS];
  int n_out = 0, D = 0;
  bool vec = true;
};

static int dot_validate(const rbx_field_t* cands, int n, int64_t R, DotHost* h) {
  if (cands == nullptr || n <= 0 || n > RBX_MAX_FIELDS) return fail(RBX_ERR_INVALID, "gatherdot: bad candidate array");
  if (R < 0) return fail(RBX_ERR_INVALID, "gatherdot: negative rows");
  bool used[kDotMaxOut] = {false};
  h->D = cands[0].dim;
  int total = 0;
  for (int i = 0; i < n; ++i) {
    const rbx_field_t& f = cands[i];
    if (f.kind != RBX_FIELD_CATEGORICAL) return fail(RBX_ERR_UNSUPPORTED, "gatherdot: set %d is not categorical", i);
    if (f.dim != h->D) return fail(RBX_ERR_UNSUPPORTED, "gatherdot: all candidate tables must have the same dim");
    if (f.seq_len <= 0 || f.out_off < 0 || f.out_off + f.seq_len > kDotMaxOut)
      return fail(RBX_ERR_UNSUPPORTED, "gatherdot: set %d: columns [%lld, +%d) outside [0, %d)", i,
                  static_cast<long long>(f.out_off), f.seq_len, kDotMaxOut);
    for (int l = 0; l < f.seq_len; ++l) {
      const int c = static_cast<int>(f.out_off) + l;
      if (used[c]) return fail(RBX_ERR_INVALID, "gatherdot: output column %d is written twice", c);
      used[c] = true;
      h->cols.set[c] = static_cast<unsigned char>(i);
      h->cols.pos[c] = static_cast<unsigned char>(l);
    }
    if (f.seq_len > 255) return fail(RBX_ERR_UNSUPPORTED, "gatherdot: more than 255 candidates in one set");
    total += f.seq_len;
    if (f.dim % 4 != 0 || (reinterpret_cast<uintptr_t>(f.table) & 15) != 0) h->vec = false;
    h->plan_fields[i] = f;
    h->plan_fields[i].out_off = 0;
    h->plan_fields[i].pool = RBX_POOL_CONCAT;          // every position of the set is its own lookup
    h->col0[i] = static_cast<int>(f.out_off);
  }
  for (int c = 0; c < total; ++c)
    if (!used[c]) return fail(RBX_ERR_INVALID, "gatherdot: output columns must cover [0, %d) without holes", total);
  h->n_out = total;
  return pack_fields(h->plan_fields, n, R, false, &h->pack);
}

template <int G, int NV, bool VEC>
static int launch_dot_fwd(const DotHost& h, int n, int64_t R, const float* x, int64_t xs, float scale, float* out,
                          int* status, hipStream_t s) {
  const int gpb = 256 / G;
  long long blocks = (R + gpb - 1) / gpb;
  if (blocks > kCUs * 16) blocks = kCUs * 16;
  hipLaunchKernelGGL((gatherdot_fwd_kernel<G, NV, VEC>), dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s, h.pack, n,
                     h.cols, static_cast<long long>(R), h.n_out, x, static_cast<long long>(xs), scale, h.D, out, status);
  return check_launch("gatherdot_fwd_kernel");
}

template <int G, int NV, bool VEC>
static int launch_dot_dx(const DotHost& h, int n, int64_t R, const float* g, float scale, float* dx, int64_t dxs,
                         hipStream_t s) {
  const int gpb = 256 / G;
  long long blocks = (R + gpb - 1) / gpb;
  if (blocks > kCUs * 16) blocks = kCUs * 16;
  hipLaunchKernelGGL((gatherdot_dx_kernel<G, NV, VEC>), dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s, h.pack, n,
                     h.cols, static_cast<long long>(R), h.n_out, g, scale, h.D, dx, static_cast<long long>(dxs));
  return check_launch("gatherdot_dx_kernel");
}

#define RBX_DOT_DISPATCH(FN, VEC, ...)                                                    \
  switch (pow2_ceil((VEC) ? h.D / 4 : h.D)) {                                             \
    case 1: return FN<1, 1, VEC>(__VA_ARGS__);                                            \
    case 2: return FN<2, 1, VEC>(__VA_ARGS__);                                            \
    case 4: return FN<4, 1, VEC>(__VA_ARGS__);                                            \
    case 8: return FN<8, 1, VEC>(__VA_ARGS__);                                            \
    case 16: return FN<16, 1, VEC>(__VA_ARGS__);                                          \
    case 32: return FN<32, 1, VEC>(__VA_ARGS__);                                          \
    case 64: return FN<64, 1, VEC>(__VA_ARGS__);                                          \
    case 128: return FN<64, 2, VEC>(__VA_ARGS__);                                         \
    case 256: return FN<64, 4, VEC>(__VA_ARGS__);                                         \
    default: return fail(RBX_ERR_UNSUPPORTED, "gatherdot: dim %d too large", h.D);        \
  }

static int dispatch_dot_fwd(const DotHost& h, int n, int64_t R, const float* x, int64_t xs, float scale, float* out,
                            int* status, hipStream_t s) {
  if (h.vec) { RBX_DOT_DISPATCH(launch_dot_fwd, true, h, n, R, x, xs, scale, out, status, s) }
  RBX_DOT_DISPATCH(launch_dot_fwd, false, h, n, R, x, xs, scale, out, status, s)
}

static int dispatch_dot_dx(const DotHost& h, int n, int64_t R, const float* g, float scale, float* dx, int64_t dxs,
                           hipStream_t s) {
  if (h.vec) { RBX_DOT_DISPATCH(launch_dot_dx, true, h, n, R, g, scale, dx, dxs, s) }
  RBX_DOT_DISPATCH(launch_dot_dx, false, h, n, R, g, scale, dx, dxs, s)
}

}  // namespace rbx

extern "C" int rbx_gatherdot_fwd(const rbx_field_t* cands, int32_t n_cands, int64_t rows, const float* d_x,
                                 int64_t x_stride, float scale, float* d_out, int32_t* d_status, void* stream) {
  using namespace rbx;
  if (rows == 0) return RBX_OK;                            // empty batch: nothing to read, pointers may be NULL
  static thread_local DotHost h;
  h = DotHost();
  int rc = dot_validate(cands, n_cands, rows, &h);
  if (rc != RBX_OK) return rc;
  if (d_x == nullptr || d_out == nullptr) return fail(RBX_ERR_INVALID, "gatherdot: NULL tensor");
  if (x_stride < h.D) return fail(RBX_ERR_INVALID, "gatherdot: x_stride %lld < dim %d", static_cast<long long>(x_stride), h.D);
  if (x_stride % 4 != 0 || (reinterpret_cast<uintptr_t>(d_x) & 15) != 0) h.vec = false;
  return dispatch_dot_fwd(h, n_cands, rows, d_x, x_stride, scale, d_out, d_status, as_stream(stream));
}

extern "C" size_t rbx_gatherdot_bwd_workspace_size(const rbx_field_t* cands, int32_t n_cands, int64_t rows) {
  using namespace rbx;
  if (rows <= 0) return 0;
  static thread_local DotHost h;
  h = DotHost();
  if (dot_validate(cands, n_cands, rows, &h) != RBX_OK) return 0;
  BwdPlan p;
  if (make_plan(h.plan_fields, n_cands, rows, nullptr, 0, &p) != RBX_OK) return 0;
  return p.bytes;
}

extern "C" int rbx_gatherdot_sort(const rbx_field_t* cands, int32_t n_cands, int64_t rows, void* d_workspace,
                                  size_t workspace_bytes, int32_t* d_status, void* stream) {
  using namespace rbx;
  if (rows == 0) return RBX_OK;
  static thread_local DotHost h;
  h = DotHost();
  int rc = dot_validate(cands, n_cands, rows, &h);
  if (rc != RBX_OK) return rc;
  BwdPlan p;
  rc = make_plan(h.plan_fields, n_cands, rows, nullptr, 0, &p);
  if (rc != RBX_OK) return rc;
  if (p.n_lookups == 0) return RBX_OK;
  if (d_workspace == nullptr || workspace_bytes < p.bytes)
    return fail(RBX_ERR_WORKSPACE, "workspace %zu B < required %zu B", workspace_bytes, p.bytes);
  return run_sort(p, static_cast<char*>(d_workspace), d_status, as_stream(stream));
}

extern "C" int rbx_gatherdot_bwd(const rbx_field_t* cands, int32_t n_cands, int64_t rows, const float* d_x,
                                 int64_t x_stride, const float* d_dout, float scale, float* d_dx, int64_t dx_stride,
                                 int32_t accumulate, void* d_workspace, size_t workspace_bytes, void* stream) {
  using namespace rbx;
  if (rows == 0) return RBX_OK;
  static thread_local DotHost h;
  h = DotHost();
  int rc = dot_validate(cands, n_cands, rows, &h);
  if (rc != RBX_OK) return rc;
  if (d_x == nullptr || d_dout == nullptr) return fail(RBX_ERR_INVALID, "gatherdot_bwd: NULL tensor");
  hipStream_t s = as_stream(stream);
  if (d_dx != nullptr) {
    DotHost& hx = h;
    const bool keep_vec = hx.vec;
    if (dx_stride % 4 != 0 || (reinterpret_cast<uintptr_t>(d_dx) & 15) != 0) hx.vec = false;
    rc = dispatch_dot_dx(hx, n_cands, rows, d_dout, scale, d_dx, dx_stride, s);
    hx.vec = keep_vec;
    if (rc != RBX_OK) return rc;
  }
  BwdPlan p;
  rc = make_plan(h.plan_fields, n_cands, rows, d_x, x_stride, &p);     // vec check against the x rows
  if (rc != RBX_OK) return rc;
  if (p.n_lookups == 0) return RBX_OK;                                  // every table frozen
  if (d_workspace == nullptr || workspace_bytes < p.bytes)
    return fail(RBX_ERR_WORKSPACE, "workspace %zu B < required %zu B", workspace_bytes, p.bytes);
  char* ws = static_cast<char*>(d_workspace);
  const int cur = p.passes & 1;
  const unsigned* keys = reinterpret_cast<const unsigned*>(ws + p.off_keys[cur]);
  const unsigned* vals = reinterpret_cast<const unsigned*>(ws + p.off_vals[cur]);
  DotPolicy::Args args;
  args.x = d_x;
  args.xs = x_stride;
  args.g = d_dout;
  args.n_out = h.n_out;
  args.scale = scale;
  args.accumulate = accumulate;
  for (int i = 0; i < RBX_MAX_FIELDS; ++i) args.col0[i] = (i < n_cands) ? h.col0[i] : 0;
  return p.vec ? dispatch_reduce<DotPolicy, true>(p, args, keys, vals, ws, s)
               : dispatch_reduce<DotPolicy, false>(p, args, keys, vals, ws, s);
}
