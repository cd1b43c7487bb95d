// rbx_common.hip -- error plumbing, descriptor validation, version (host only).
#include <stdarg.h>
#include <stdio.h>
#include "rbx_internal.h"

namespace rbx {

static thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

int check_launch(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(RBX_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
  return RBX_OK;
}

static int compact_id(int64_t v) {
  if (v == RBX_NO_ID || v < INT_MIN || v > INT_MAX) return kNoId;
  return static_cast<int>(v);
}

int pack_fields(const rbx_field_t* fields, int n, int64_t batch, bool need_grad, FieldPack* out) {
  if (fields == nullptr) return fail(RBX_ERR_INVALID, "fields is NULL");
  if (n <= 0 || n > RBX_MAX_FIELDS) return fail(RBX_ERR_INVALID, "n_fields=%d not in [1,%d]", n, RBX_MAX_FIELDS);
  for (int i = 0; i < n; ++i) {
    const rbx_field_t& f = fields[i];
    FieldK& k = out->f[i];
    if (f.ids == nullptr) return fail(RBX_ERR_INVALID, "field %d: ids is NULL", i);
    if (f.ids_dtype < RBX_I32 || f.ids_dtype > RBX_F64) return fail(RBX_ERR_INVALID, "field %d: bad ids_dtype", i);
    if (f.kind < RBX_FIELD_CATEGORICAL || f.kind > RBX_FIELD_DENSE)
      return fail(RBX_ERR_UNSUPPORTED, "field %d: feature kind %d is not supported", i, f.kind);
    if (f.pool < RBX_POOL_NONE || f.pool > RBX_POOL_CONCAT)
      return fail(RBX_ERR_INVALID, "field %d: bad pool mode %d", i, f.pool);
    if (f.dim <= 0 || f.dim > 1024) return fail(RBX_ERR_UNSUPPORTED, "field %d: dim=%d not in [1,1024]", i, f.dim);
    if (f.seq_len <= 0 || f.seq_len > 32767) return fail(RBX_ERR_INVALID, "field %d: seq_len=%d", i, f.seq_len);
    if (f.kind != RBX_FIELD_DENSE && f.table == nullptr) return fail(RBX_ERR_INVALID, "field %d: table is NULL", i);
    if (f.kind == RBX_FIELD_DENSE && f.dim != 1) return fail(RBX_ERR_INVALID, "field %d: dense dim must be 1", i);
    if (f.kind == RBX_FIELD_CATEGORICAL) {
      if (f.vocab <= 0 || f.vocab > INT_MAX) return fail(RBX_ERR_INVALID, "field %d: vocab=%lld", i, (long long)f.vocab);
      if (f.seq_len == 1 && f.pool != RBX_POOL_NONE && f.pool != RBX_POOL_CONCAT) {
        /* a length-1 sequence is legal; nothing to check */
      }
      if (f.seq_len > 1 && f.pool == RBX_POOL_NONE)
        return fail(RBX_ERR_INVALID, "field %d: seq_len>1 needs a pool mode", i);
    } else if (f.seq_len != 1 || f.pool != RBX_POOL_NONE) {
      return fail(RBX_ERR_INVALID, "field %d: numeric/dense features are scalar per sample", i);
    }
    if (f.table_stride != 0 && f.table_stride != f.dim && f.kind == RBX_FIELD_CATEGORICAL)
      return fail(RBX_ERR_UNSUPPORTED, "field %d: a table row stride of %lld floats (dim %d) is only supported by the fused FM "
                  "entry points", i, (long long)f.table_stride, f.dim);
    if (f.out_off < 0 || f.out_off > INT_MAX) return fail(RBX_ERR_INVALID, "field %d: out_off", i);
    if (f.ids_stride_l < INT_MIN || f.ids_stride_l > INT_MAX) return fail(RBX_ERR_INVALID, "field %d: stride_l", i);
    if (need_grad && f.kind != RBX_FIELD_DENSE && f.grad == nullptr) {
      /* grad == NULL means "frozen": allowed, the field is skipped in backward */
    }
    k.ids = f.ids;
    k.table = f.table;
    k.ids_stride_b = f.ids_stride_b;
    k.ids_stride_l = static_cast<int>(f.ids_stride_l);
    k.vocab = static_cast<int>(f.vocab);
    k.mask_id = compact_id(f.mask_id);
    k.pad_id = compact_id(f.padding_idx);
    k.out_off = static_cast<int>(f.out_off);
    k.dim = static_cast<short>(f.dim);
    k.seq_len = static_cast<short>(f.seq_len);
    k.ids_dtype = static_cast<unsigned char>(f.ids_dtype);
    k.kind = static_cast<unsigned char>(f.kind);
    k.pool = static_cast<unsigned char>(f.pool);
    k.slot = static_cast<unsigned char>(i);
    k.eps = f.eps;
  }
  (void)batch;
  return RBX_OK;
}

}  // namespace rbx

extern "C" const char* rbx_last_error(void) { return rbx::g_err; }
extern "C" int rbx_version(void) { return RBX_VERSION; }
