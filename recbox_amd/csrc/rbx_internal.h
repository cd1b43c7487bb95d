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

// Individually rounded f32 operations the compiler may NOT contract into fused multiply-adds (HIP's __fmul_rn / __fadd_rn
// are plain `*` / `+` and contract like any other): two kernels written with them perform the same roundings.
__device__ __forceinline__ float mul_rn(float a, float b) {
#pragma clang fp contract(off)
  return a * b;
}
__device__ __forceinline__ float add_rn(float a, float b) {
#pragma clang fp contract(off)
  return a + b;
}
__device__ __forceinline__ float fma_rn(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

template <int W>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = W / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, W);
  return v;
}

// Philox4x32-10 (Salmon et al., SC'11): counter-based, so a parallel kernel and its replay in a backward pass draw the
// same bits.  Used by the negative sampler (rbx_sampler.hip) and by attention dropout (rbx_attn*.hip).
struct Philox {
  static constexpr unsigned kM0 = 0xD2511F53u, kM1 = 0xCD9E8D57u, kW0 = 0x9E3779B9u, kW1 = 0xBB67AE85u;
  // 10 rounds of Philox4x32; c = counter, k = key; result in c
  static __host__ __device__ __forceinline__ void run(unsigned c[4], unsigned k0, unsigned k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
      const unsigned long long p0 = static_cast<unsigned long long>(kM0) * c[0];
      const unsigned long long p1 = static_cast<unsigned long long>(kM1) * c[2];
      const unsigned n0 = static_cast<unsigned>(p1 >> 32) ^ c[1] ^ k0;
      const unsigned n1 = static_cast<unsigned>(p1);
      const unsigned n2 = static_cast<unsigned>(p0 >> 32) ^ c[3] ^ k1;
      const unsigned n3 = static_cast<unsigned>(p0);
      c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
      k0 += kW0;
      k1 += kW1;
    }
  }
};

// ---- attention dropout: keep(bh, query i, key j) ---------------------------------------------------------------
// One Philox call covers a 2 x 4 block of the (query, key) plane with 16-bit decisions: counter = (i >> 2, j >> 2,
// bh, 2 * (bh >> 32) + ((i & 3) >> 1)), word 2 * (i & 1) + ((j & 3) >> 1), half-word j & 1; keep iff it is >= thr16 =
// round(p * 65536).  Forward (lane = query) and both backward phases (lane = query / lane = key) evaluate the same
// function, so the mask is never stored.  The kept probabilities are scaled by 65536 / (65536 - thr16).
struct DropArgs {
  unsigned thr16;      // 0 = no dropout
  float scale;
  unsigned k0, k1;     // seed
  const unsigned long long* seed_add;   // optional device word added to the seed (bumped between hipGraph replays)
};

__device__ __forceinline__ void drop_block(unsigned i4, unsigned j4, unsigned long long bh, int h, unsigned k0, unsigned k1,
                                           unsigned (&c)[4]) {
  c[0] = i4; c[1] = j4; c[2] = static_cast<unsigned>(bh); c[3] = (static_cast<unsigned>(bh >> 32) << 1) | static_cast<unsigned>(h);
  Philox::run(c, k0, k1);
}

__device__ __forceinline__ bool drop_keep(const unsigned (&c)[4], int i_low /* i & 1 */, int j_low /* j & 3 */, unsigned thr16) {
  const unsigned w = c[2 * i_low + (j_low >> 1)];
  return ((w >> (16 * (j_low & 1))) & 0xFFFFu) >= thr16;
}

__device__ __forceinline__ void drop_seed(const DropArgs& d, unsigned* k0, unsigned* k1) {
  unsigned long long s = (static_cast<unsigned long long>(d.k1) << 32) | d.k0;
  if (d.seed_add != nullptr) s += *d.seed_add;
  *k0 = static_cast<unsigned>(s);
  *k1 = static_cast<unsigned>(s >> 32);
}

inline int pow2_ceil(int v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}

// ---- the fused FM forward's per-feature descriptor and id decoding (rbx_fm_fused.hip, rbx_fm_quad.hip) ----------------
struct FmField {            // 48 B
  const void* ids;
  const float* emb;         // [V, D] table or numeric weight [D]; NULL when there is no second-order part
  const float* lr;          // [V] (dim-1 table) or numeric weight [1]; NULL when there is no first-order part
  long long stride_b;
  int vocab;
  unsigned char dtype, kind, r0, r1;
  int emb_stride;           // floats between rows of emb / lr: D and 1 for contiguous tables; equal (e.g. 32) when both
  int lr_stride;            // live in one packed [V, stride] storage -- then a lookup touches ONE 128-byte line
};
struct FmPack { FmField f[RBX_MAX_FIELDS]; };

// Uniform-dtype fast path of the forward (DT = the one ids dtype every feature of the call has; -1 = mixed: the generic code).
// The generic path converts a float64 id with static_cast<long long>(double) -- ~20 emulated instructions, there is no
// 64-bit convert on gfx950 -- keeps ids in 64-bit registers and branches on the dtype of every feature; with 39 features per
// sample the forward was bound by instruction issue, not by memory (every experiment on its memory side came out flat:
// profiles/r02/sort_variants.txt).  Here: one v_cvt_i32_f64 (an id is < 2^31 once it passes the range check; NaN and
// out-of-range values fail it exactly as in the generic path), 32-bit ids, no dtype switch.
template <int DT>
__device__ __forceinline__ long long fm_load_raw(const void* p, long long idx) {
  if constexpr (DT == RBX_I64 || DT == RBX_F64) return static_cast<const long long*>(p)[idx];
  else return static_cast<long long>(static_cast<const int*>(p)[idx]);
}
template <int DT>
__device__ __forceinline__ bool fm_decode_id(long long raw, int vocab, int* id) {
  if constexpr (DT == RBX_I32) {
    *id = static_cast<int>(raw);
    return static_cast<unsigned>(*id) < static_cast<unsigned>(vocab);
  } else if constexpr (DT == RBX_I64) {
    *id = static_cast<int>(raw);
    return static_cast<unsigned long long>(raw) < static_cast<unsigned long long>(vocab);
  } else if constexpr (DT == RBX_F32) {
    const float f = __int_as_float(static_cast<int>(raw));
    *id = __float2int_rz(f);                                // .long() truncates towards zero; saturates beyond int32
    return (f == f) && static_cast<unsigned>(*id) < static_cast<unsigned>(vocab);
  } else {
    const double d = __longlong_as_double(raw);
    *id = __double2int_rz(d);
    return (d == d) && static_cast<unsigned>(*id) < static_cast<unsigned>(vocab);
  }
}
template <int DT>
__device__ __forceinline__ float fm_decode_value(long long raw) {
  if constexpr (DT == RBX_I32 || DT == RBX_I64) return static_cast<float>(raw);
  else if constexpr (DT == RBX_F32) return __int_as_float(static_cast<int>(raw));
  else return static_cast<float>(__longlong_as_double(raw));
}

// rbx_fm_quad.hip: the forward for ids that are the columns of one batch tensor (dim 16).  RBX_OK = launched, 1 = not a
// call of that shape (launch the general kernel), < 0 = error.
int fm_quad_fwd(const FmPack& pack, int F, int D, bool has_emb, bool has_lr, int uniform_dt, long long B, const float* bias,
                float* logit, float* prob, float* ssum, int* status, hipStream_t s);

}  // namespace rbx
