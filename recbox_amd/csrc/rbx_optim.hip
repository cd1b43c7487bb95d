// rbx_optim.hip -- sparse-row optimiser step for embedding tables (gfx950): only the rows a backward touched are
// read and written.
//
// The reference trains every table with a dense torch.optim step over a dense [V, D] gradient
// (ranking/pytorch/models/ranking_model.py:191-197: clip_grad_norm_ + optimizer.step(); matching/pytorch/models/
// match_model.py:194-199): at cfg 3 (10 M x 128 rows) that is 5 GB of gradient read and 10-20 GB of optimiser traffic per
// step around a 1.8 ms forward + backward, for ~2 M touched rows.  SURVEY.md 8(b) / 2.2 K3 asks for the opt-in sparse-row
// path: the sorted (row, lookup) pairs that drove the backward's segmented reduce are still in its workspace and name
// every touched row exactly once (the head of each run of equal keys); the summed gradient of such a row sits in the
// dense gradient buffer the backward stored into.  One launch walks the pairs; a lane group per run head applies
//   SGD      w -= lr (g + wd w)
//   Adagrad  s += g^2;  w -= lr g / (sqrt(s) + eps)                         (torch.optim.Adagrad, sparse branch)
//   Adam     m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g^2;  w -= lr m / (sqrt(v) + eps)
//            with lr = lr0 sqrt(1 - b2^t) / (1 - b1^t) folded in by the caller   (torch.optim.SparseAdam: "lazy", the
//            moments of untouched rows do not decay)
// to that row of the table and of its state tensors, and to nothing else.  The tables of the fused FM body that take the
// sort-free path (rbx_tiera.h: every row written by every backward) are walked by row; a row counts as touched when any
// block's presence bitmap has it.
#include <stdlib.h>
#include "rbx_segreduce.h"
#include "rbx_tiera.h"

namespace rbx {

struct OptArgs {
  int kind;
  float lr, beta1, beta2, eps, wd;
  const float* d_lr;         // device step size overriding lr (a captured step: rbx_opt_advance keeps it current), or NULL
};

__device__ __forceinline__ void opt_apply(const OptArgs& o, float g, float& w, float& s1, float& s2) {
  const float lr = o.d_lr != nullptr ? *o.d_lr : o.lr;          // (a uniform address: one scalar load)
  g += o.wd * w;
  if (o.kind == RBX_OPT_SGD) {
    w -= lr * g;
  } else if (o.kind == RBX_OPT_ADAGRAD) {
    s1 += g * g;
    w -= lr * g / (sqrtf(s1) + o.eps);
  } else {
    s1 = o.beta1 * s1 + (1.f - o.beta1) * g;
    s2 = o.beta2 * s2 + (1.f - o.beta2) * g * g;
    w -= lr * s1 / (sqrtf(s2) + o.eps);
  }
}

__global__ void opt_advance_kernel(int kind, float lr, float b1, float b2, float lr_decay, float* t_ptr, float* out) {
  const float t = *t_ptr + 1.f;
  *t_ptr = t;
  float s = lr;
  if (kind == RBX_OPT_ADAGRAD) s = lr / (1.f + (t - 1.f) * lr_decay);
  else if (kind == RBX_OPT_ADAM) s = lr * sqrtf(1.f - powf(b2, t)) / (1.f - powf(b1, t));
  *out = s;
}

struct UpdField {            // 80 B
  float* w;                  // table rows [*, w_stride]
  const float* g;            // dense gradient [*, dim]
  float* w2;                 // fused FM: the dim-1 LR table (or NULL)
  const float* g2;
  float* s1;                 // state of w (same layout as g), NULL where the rule has none
  float* s2;
  float* t1;                 // state of w2
  float* t2;
  unsigned row_base;
  int dim;
  int w_stride;
  int reserved;
};
constexpr int kUpdFields = 32;                       // plan slots one launch serves (kernarg budget)
struct UpdPack { UpdField f[kUpdFields]; };

// one row: the G lanes of a group sweep its dim floats (float4 when VEC)
template <bool VEC>
__device__ __forceinline__ void update_row(const OptArgs& o, const UpdField& fd, size_t row, int lane_g, int G) {
  constexpr int W = VEC ? 4 : 1;
  if (fd.w != nullptr && fd.g != nullptr) {
    float* wrow = fd.w + row * fd.w_stride;
    const float* grow = fd.g + row * fd.dim;
    float* s1row = fd.s1 != nullptr ? fd.s1 + row * fd.dim : nullptr;
    float* s2row = fd.s2 != nullptr ? fd.s2 + row * fd.dim : nullptr;
    for (int e = lane_g * W; e < fd.dim; e += G * W) {
      if constexpr (VEC) {
        const float4 g = *reinterpret_cast<const float4*>(grow + e);
        float4 w = *reinterpret_cast<const float4*>(wrow + e);
        float4 a = s1row ? *reinterpret_cast<const float4*>(s1row + e) : make_float4(0.f, 0.f, 0.f, 0.f);
        float4 b = s2row ? *reinterpret_cast<const float4*>(s2row + e) : make_float4(0.f, 0.f, 0.f, 0.f);
        opt_apply(o, g.x, w.x, a.x, b.x);
        opt_apply(o, g.y, w.y, a.y, b.y);
        opt_apply(o, g.z, w.z, a.z, b.z);
        opt_apply(o, g.w, w.w, a.w, b.w);
        *reinterpret_cast<float4*>(wrow + e) = w;
        if (s1row) *reinterpret_cast<float4*>(s1row + e) = a;
        if (s2row) *reinterpret_cast<float4*>(s2row + e) = b;
      } else {
        float w = wrow[e], a = s1row ? s1row[e] : 0.f, b = s2row ? s2row[e] : 0.f;
        opt_apply(o, grow[e], w, a, b);
        wrow[e] = w;
        if (s1row) s1row[e] = a;
        if (s2row) s2row[e] = b;
      }
    }
  }
  if (fd.w2 != nullptr && fd.g2 != nullptr && lane_g == 0) {
    float w = fd.w2[row], a = fd.t1 ? fd.t1[row] : 0.f, b = fd.t2 ? fd.t2[row] : 0.f;
    opt_apply(o, fd.g2[row], w, a, b);
    fd.w2[row] = w;
    if (fd.t1) fd.t1[row] = a;
    if (fd.t2) fd.t2[row] = b;
  }
}

// sorted pairs -> run heads -> rows.  One lane group per pair; plan slots [slot_lo, slot_lo + n_slots) only.
template <bool VEC>
__global__ __launch_bounds__(256) void sparse_update_kernel(const UpdPack P, const int slot_lo, const int n_slots,
                                                            const OptArgs o, const unsigned* __restrict__ keys,
                                                            const unsigned* __restrict__ vals, const unsigned n,
                                                            const unsigned sentinel, const int G) {
  const int lane_g = threadIdx.x % G;
  const unsigned long long groups = static_cast<unsigned long long>(gridDim.x) * (256 / G);
  for (unsigned long long i = static_cast<unsigned long long>(blockIdx.x) * (256 / G) + threadIdx.x / G; i < n; i += groups) {
    const unsigned key = keys[i];
    if (key >= sentinel) continue;
    if (i > 0 && keys[i - 1] == key) continue;                  // not the head of its run
    const int slot = static_cast<int>(vals[i] >> kLocalBits) - slot_lo;
    if (slot < 0 || slot >= n_slots) continue;
    const UpdField& fd = P.f[slot];
    update_row<VEC>(o, fd, static_cast<size_t>(key - fd.row_base), lane_g, G);
  }
}

// tier A of the fused FM body: one lane group per table row, touched iff some block's bitmap names it
struct TaUpd {               // 88 B
  UpdField u;
  int f_begin, f_count;
};
constexpr int kTaUpdTables = 32;
struct TaUpdPack { TaUpd t[kTaUpdTables]; };

template <bool VEC>
__global__ __launch_bounds__(256) void ta_sparse_update_kernel(const TaUpdPack T, const TaFieldRefPack P, const int n_tab,
                                                               const OptArgs o, const unsigned NB,
                                                               const unsigned* __restrict__ bitmap, const int G) {
  const int lane_g = threadIdx.x % G;
  const unsigned R0 = blockIdx.x * (256 / G) + threadIdx.x / G;
  // tables are few: walk them (row0 ascending) to find the one this row index falls into
  unsigned R = R0;
  int t = 0;
  for (; t < n_tab; ++t) {
    const unsigned V = static_cast<unsigned>(T.t[t].u.reserved);      // (reserved carries the table's row count here)
    if (R < V) break;
    R -= V;
  }
  if (t >= n_tab) return;
  const TaUpd& tb = T.t[t];
  const unsigned V = static_cast<unsigned>(tb.u.reserved);
  if (static_cast<int>(R) == static_cast<int>(tb.u.row_base)) return;  // (row_base carries padding_idx here: never updated)
  const unsigned words = (V + 31u) >> 5;
  bool any = false;
  for (int f = tb.f_begin; f < tb.f_begin + tb.f_count && !any; ++f) {
    const unsigned* bw = bitmap + static_cast<size_t>(P.f[f].fword0) * NB + (R >> 5);
    for (unsigned k = 0; k < NB && !any; ++k) any = ((bw[static_cast<size_t>(k) * words] >> (R & 31u)) & 1u) != 0u;
  }
  if (!any) return;
  UpdField fd = tb.u;
  fd.row_base = 0;
  update_row<VEC>(o, fd, R, lane_g, G);
}

static int opt_validate(const rbx_opt_t* opt, OptArgs* o) {
  if (opt == nullptr) return fail(RBX_ERR_INVALID, "sparse_update: opt is NULL");
  if (opt->kind < RBX_OPT_SGD || opt->kind > RBX_OPT_ADAM) return fail(RBX_ERR_INVALID, "sparse_update: unknown rule %d", opt->kind);
  o->kind = opt->kind;
  o->lr = opt->lr;
  o->beta1 = opt->beta1;
  o->beta2 = opt->beta2;
  o->eps = opt->eps;
  o->wd = opt->weight_decay;
  o->d_lr = opt->d_step_size;
  return RBX_OK;
}

extern "C" int rbx_opt_advance(int32_t kind, float lr, float beta1, float beta2, float lr_decay, float* d_t,
                               float* d_step_size, void* stream) {
  if (kind < RBX_OPT_SGD || kind > RBX_OPT_ADAM) return fail(RBX_ERR_INVALID, "opt_advance: unknown rule %d", kind);
  if (d_t == nullptr || d_step_size == nullptr) return fail(RBX_ERR_INVALID, "opt_advance: NULL counter / step size");
  opt_advance_kernel<<<1, 1, 0, as_stream(stream)>>>(kind, lr, beta1, beta2, lr_decay, d_t, d_step_size);
  return check_launch("opt_advance_kernel");
}

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

static bool upd_vec_ok(const UpdField& u) {
  if (u.dim % 4 != 0 || u.w_stride % 4 != 0) return false;
  return aligned16(u.w) && aligned16(u.g) && (u.s1 == nullptr || aligned16(u.s1)) && (u.s2 == nullptr || aligned16(u.s2));
}

static int lanes_for(int dim, bool vec) {
  const int units = vec ? (dim + 3) / 4 : dim;
  int g = 1;
  while (g < units && g < 64) g *= 2;
  return g;
}

// launch the pair walk over the plan's sorted keys, kUpdFields plan slots at a time
static int launch_pairs(const BwdPlan& p, const UpdField* upd, const OptArgs& o, const char* ws, hipStream_t s) {
  if (p.n_lookups == 0) return RBX_OK;
  const int cur = p.passes & 1;
  const unsigned* keys = reinterpret_cast<const unsigned*>(ws + p.off_keys[cur]);
  const unsigned* vals = reinterpret_cast<const unsigned*>(ws + p.off_vals[cur]);
  for (int lo = 0; lo < p.n_cat; lo += kUpdFields) {
    const int cnt = (p.n_cat - lo < kUpdFields) ? p.n_cat - lo : kUpdFields;
    UpdPack pack;
    bool vec = true;
    int dim = 1;
    for (int c = 0; c < cnt; ++c) {
      pack.f[c] = upd[lo + c];
      if (pack.f[c].w != nullptr) {
        vec = vec && upd_vec_ok(pack.f[c]);
        if (pack.f[c].dim > dim) dim = pack.f[c].dim;
      }
    }
    const int G = lanes_for(dim, vec);
    unsigned long long blocks = (static_cast<unsigned long long>(p.n_lookups) + (256 / G) - 1) / (256 / G);
    if (blocks > static_cast<unsigned long long>(kCUs) * 64) blocks = kCUs * 64;
    if (vec)
      hipLaunchKernelGGL(sparse_update_kernel<true>, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s, pack, lo, cnt, o,
                         keys, vals, p.n_lookups, p.total_rows, G);
    else
      hipLaunchKernelGGL(sparse_update_kernel<false>, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s, pack, lo, cnt, o,
                         keys, vals, p.n_lookups, p.total_rows, G);
    const int rc = check_launch("sparse_update_kernel");
    if (rc != RBX_OK) return rc;
  }
  return RBX_OK;
}

}  // namespace rbx

extern "C" int rbx_embed_sparse_update(const rbx_field_t* fields, int32_t n_fields, int64_t batch, const void* d_workspace,
                                       size_t workspace_bytes, const rbx_opt_t* opt, float* const* d_state1,
                                       float* const* d_state2, void* stream) {
  using namespace rbx;
  OptArgs o;
  int rc = opt_validate(opt, &o);
  if (rc != RBX_OK) return rc;
  if (batch <= 0) return RBX_OK;
  BwdPlan p;
  rc = make_plan(fields, n_fields, batch, nullptr, 0, &p);
  if (rc != RBX_OK) return rc;
  if (p.n_lookups == 0) return RBX_OK;
  if (d_workspace == nullptr || workspace_bytes < p.bytes) return fail(RBX_ERR_WORKSPACE, "sparse_update: workspace too small");
  UpdField upd[RBX_MAX_FIELDS];
  for (int c = 0; c < p.n_cat; ++c) {
    const RedField& rf = p.red.f[c];
    const int i = rf.slot;                              // index of the feature in `fields`
    UpdField& u = upd[c];
    u.w = const_cast<float*>(fields[i].table);
    u.g = fields[i].grad;
    u.w2 = nullptr; u.g2 = nullptr; u.t1 = nullptr; u.t2 = nullptr;
    u.s1 = d_state1 != nullptr ? d_state1[i] : nullptr;
    u.s2 = d_state2 != nullptr ? d_state2[i] : nullptr;
    u.row_base = rf.row_base;
    u.dim = rf.dim;
    u.w_stride = rf.dim;
    u.reserved = 0;
    if (o.kind != RBX_OPT_SGD && u.s1 == nullptr) return fail(RBX_ERR_INVALID, "sparse_update: feature %d has no state tensor", i);
    if (o.kind == RBX_OPT_ADAM && u.s2 == nullptr) return fail(RBX_ERR_INVALID, "sparse_update: feature %d has no second moment", i);
  }
  return launch_pairs(p, upd, o, static_cast<const char*>(d_workspace), as_stream(stream));
}

// defined in rbx_fm_fused.hip: the plan of a fused FM call (tier B sort plan + tier A description) and where its pieces live
namespace rbx {
int fm_update_plan(const rbx_field_t* emb, const rbx_field_t* lr, int n, int64_t B, BwdPlan* p, TaPlan* ta, size_t* off_ta,
                   size_t* bytes, int* src_of_key, int* src_of_tab);
}

extern "C" int rbx_fm_sparse_update(const rbx_field_t* emb, const rbx_field_t* lr, int32_t n_fields, int64_t batch,
                                    const void* d_workspace, size_t workspace_bytes, const rbx_opt_t* opt,
                                    float* const* d_emb_state1, float* const* d_emb_state2, float* const* d_lr_state1,
                                    float* const* d_lr_state2, void* stream) {
  using namespace rbx;
  OptArgs o;
  int rc = opt_validate(opt, &o);
  if (rc != RBX_OK) return rc;
  if (batch <= 0) return RBX_OK;
  BwdPlan p;
  TaPlan ta;
  size_t off_ta = 0, bytes = 0;
  int src_of_key[RBX_MAX_FIELDS], src_of_tab[RBX_MAX_FIELDS];
  rc = fm_update_plan(emb, lr, n_fields, batch, &p, &ta, &off_ta, &bytes, src_of_key, src_of_tab);
  if (rc != RBX_OK) return rc;
  if (d_workspace == nullptr || workspace_bytes < bytes) return fail(RBX_ERR_WORKSPACE, "fm_sparse_update: workspace too small");
  const char* ws = static_cast<const char*>(d_workspace);
  hipStream_t s = as_stream(stream);
  const int D = emb ? emb[0].dim : 1;
  auto fill = [&](UpdField& u, int i) -> int {
    u.w = (emb && emb[i].grad) ? const_cast<float*>(emb[i].table) : nullptr;
    u.g = emb ? emb[i].grad : nullptr;
    u.w2 = (lr && lr[i].grad) ? const_cast<float*>(lr[i].table) : nullptr;
    u.g2 = lr ? lr[i].grad : nullptr;
    u.s1 = (u.w && d_emb_state1) ? d_emb_state1[i] : nullptr;
    u.s2 = (u.w && d_emb_state2) ? d_emb_state2[i] : nullptr;
    u.t1 = (u.w2 && d_lr_state1) ? d_lr_state1[i] : nullptr;
    u.t2 = (u.w2 && d_lr_state2) ? d_lr_state2[i] : nullptr;
    u.dim = D;
    u.w_stride = (emb && emb[i].table_stride != 0) ? static_cast<int>(emb[i].table_stride) : D;
    if (lr && lr[i].table_stride != 0 && lr[i].table_stride != 1)
      return fail(RBX_ERR_UNSUPPORTED, "fm_sparse_update: packed LR tables are not supported");
    u.reserved = 0;
    if (o.kind != RBX_OPT_SGD && ((u.w && !u.s1) || (u.w2 && !u.t1)))
      return fail(RBX_ERR_INVALID, "fm_sparse_update: feature %d has no state tensor", i);
    if (o.kind == RBX_OPT_ADAM && ((u.w && !u.s2) || (u.w2 && !u.t2)))
      return fail(RBX_ERR_INVALID, "fm_sparse_update: feature %d has no second moment", i);
    return RBX_OK;
  };
  // ---- tier B: run heads of the sorted pairs ----
  if (p.n_lookups > 0) {
    UpdField upd[RBX_MAX_FIELDS];
    for (int c = 0; c < p.n_cat; ++c) {
      rc = fill(upd[c], src_of_key[c]);
      if (rc != RBX_OK) return rc;
      upd[c].row_base = p.red.f[c].row_base;
    }
    rc = launch_pairs(p, upd, o, ws, s);
    if (rc != RBX_OK) return rc;
  }
  // ---- tier A: every table row whose bit is set in some block's bitmap ----
  for (int lo = 0; lo < ta.n_tab; lo += kTaUpdTables) {
    const int cnt = (ta.n_tab - lo < kTaUpdTables) ? ta.n_tab - lo : kTaUpdTables;
    TaUpdPack pack;
    TaFieldRefPack refs;
    for (int i = 0; i < ta.n_fld; ++i) refs.f[i] = {ta.fld.f[i].frow0, ta.fld.f[i].fword0};
    bool vec = true;
    unsigned rows = 0;
    for (int c = 0; c < cnt; ++c) {
      const TaTable& tb = ta.tab.t[lo + c];
      TaUpd& tu = pack.t[c];
      rc = fill(tu.u, src_of_tab[lo + c]);
      if (rc != RBX_OK) return rc;
      tu.u.row_base = static_cast<unsigned>(tb.pad);            // (see the kernel: padding_idx travels here)
      tu.u.reserved = tb.vocab;                                 // (... and the row count here)
      tu.f_begin = tb.f_begin;
      tu.f_count = tb.f_count;
      if (tu.u.w != nullptr) vec = vec && upd_vec_ok(tu.u);
      rows += static_cast<unsigned>(tb.vocab);
    }
    const int G = lanes_for(D, vec);
    const unsigned blocks = (rows + (256 / G) - 1) / (256 / G);
    const unsigned* bitmap = reinterpret_cast<const unsigned*>(ws + off_ta + ta.off_bitmap);
    if (vec)
      hipLaunchKernelGGL(ta_sparse_update_kernel<true>, dim3(blocks), dim3(256), 0, s, pack, refs, cnt, o, ta.NB, bitmap, G);
    else
      hipLaunchKernelGGL(ta_sparse_update_kernel<false>, dim3(blocks), dim3(256), 0, s, pack, refs, cnt, o, ta.NB, bitmap, G);
    rc = check_launch("ta_sparse_update_kernel");
    if (rc != RBX_OK) return rc;
  }
  return RBX_OK;
}
