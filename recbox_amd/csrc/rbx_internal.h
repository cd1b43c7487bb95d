// rbx_internal.h -- shared host/device helpers of librecbox_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <limits.h>
#include <stdint.h>
#include "recbox_hip.h"

namespace rbx {

// ---- error plumbing (thread-local message, C ABI returns the code) ----------
int fail(int code, const char* fmt, ...);
int check_launch(const char* what);

constexpr int kWave = 64;          // CDNA wavefront
constexpr int kCUs = 256;          // MI355X
constexpr int kNoId = INT_MIN;     // compact form of RBX_NO_ID

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// ---- compact per-field descriptor that travels in the kernarg segment --------
// 56 bytes; RBX_MAX_FIELDS of them (3584 B) fit under the 4 KiB kernarg limit.
struct FieldK {
  const void* ids;
  const float* table;
  long long ids_stride_b;
  int ids_stride_l;
  int vocab;
  int mask_id;      // kNoId when unset
  int pad_id;       // kNoId when unset
  int out_off;
  short dim;
  short seq_len;
  unsigned char ids_dtype, kind, pool, slot;
  float eps;
};
static_assert(sizeof(FieldK) == 56, "FieldK must stay 56 bytes");

struct FieldPack {
  FieldK f[RBX_MAX_FIELDS];
};

// Validate a public descriptor array and convert it; returns RBX_OK or fails.
int pack_fields(const rbx_field_t* fields, int n, int64_t batch, bool need_grad, FieldPack* out);

// ---- device helpers ----------------------------------------------------------
__device__ __forceinline__ long long load_id(const void* p, long long idx, int dt) {
  switch (dt) {
    case RBX_I32: return static_cast<const int*>(p)[idx];
    case RBX_I64: return static_cast<const long long*>(p)[idx];
    case RBX_F32: return static_cast<long long>(static_cast<const float*>(p)[idx]);   // .long() truncates
    default:      return static_cast<long long>(static_cast<const double*>(p)[idx]);
  }
}

__device__ __forceinline__ float load_value(const void* p, long long idx, int dt) {
  switch (dt) {
    case RBX_I32: return static_cast<float>(static_cast<const int*>(p)[idx]);
    case RBX_I64: return static_cast<float>(static_cast<const long long*>(p)[idx]);
    case RBX_F32: return static_cast<const float*>(p)[idx];
    default:      return static_cast<float>(static_cast<const double*>(p)[idx]);      // .float()
  }
}

// Two-phase id access for kernels that keep several lookups in flight: load_raw only issues the load
// (no use of the value, so the loads of a batch stay outstanding together); decode_* do the dtype
// conversion afterwards.  Calling load_id per lookup instead serialises the batch: every range
// check waits for its own load (measured: 64 -> 4x us on the fused FM forward).
__device__ __forceinline__ long long load_raw(const void* p, long long idx, int dt) {
  if (dt & 1) return static_cast<const long long*>(p)[idx];                 // I64 / F64: 8-byte elements
  return static_cast<long long>(static_cast<const int*>(p)[idx]);             // I32 / F32: 4-byte elements
}

__device__ __forceinline__ long long decode_id(long long raw, int dt) {
  switch (dt) {
    case RBX_I32:
    case RBX_I64: return raw;
    case RBX_F32: return static_cast<long long>(__int_as_float(static_cast<int>(raw)));
    default:      return static_cast<long long>(__longlong_as_double(raw));
  }
}

__device__ __forceinline__ float decode_value(long long raw, int dt) {
  switch (dt) {
    case RBX_I32:
    case RBX_I64: return static_cast<float>(raw);
    case RBX_F32: return __int_as_float(static_cast<int>(raw));
    default:      return static_cast<float>(__longlong_as_double(raw));
  }
}

template <int W>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = W / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, W);
  return v;
}

inline int pow2_ceil(int v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}

}  // namespace rbx
