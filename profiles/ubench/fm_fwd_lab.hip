// fm_fwd_lab.hip -- stand-alone laboratory for the forward of the fused FM body at BASELINE cfg 2 (Criteo-shaped: 13 numeric
// + 26 categorical columns of ONE row-major float64 batch tensor [B, 40], dim 16, B = 65 536): forms of the kernel side by
// side on the same device-generated tables and ids, each checked against a host restatement of the arithmetic, so that a
// form can be chosen (and PMC-profiled: `fm_fwd_lab <variant> <iters>` runs one form only) without the Python stack.
//   hipcc --offload-arch=gfx950 -O3 fm_fwd_lab.hip -o fm_fwd_lab
// Variants:  0  lane group of 4 lanes (float4) per sample, 8 features in flight, ids read in place, 39 strided loads per
//               lane group (the form of rbx_fm_fused.hip through round 4)
//            1  "quad": a wavefront = 4 samples x 16 lanes, lane j owns dimension j; the samples' batch rows are read ONCE,
//               coalesced (lane j holds columns j, 16 + j, 32 + j), the owner lane decodes its ids, a row's byte offset
//               travels to the 16 lanes of its sample by DPP row broadcast, all rows of a 16-column slot in flight
//            2  quad + the tables of at most `lds_rows` rows resident in LDS (one workgroup per CU, persistent)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static const int kNum = 13, kCat = 26, kF = 39, kLd = 40, kD = 16;
static const int kCard[26] = {1460, 583, 1000000, 1000000, 305, 24, 12517, 633, 3, 93145, 5683, 1000000, 3194, 27, 14992,
                              1000000, 10, 5652, 2173, 4, 1000000, 18, 15, 286181, 105, 142572};

__host__ __device__ inline unsigned mix(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
__host__ __device__ inline float tval(int t, unsigned row, int d) {      // table t (0..38: 13 numeric weights, 26 tables)
  if (t >= 13 && row == 0) return 0.f;                                   // padding row
  const unsigned h = mix(mix(static_cast<unsigned>(t) * 0x9E3779B9u + row) + static_cast<unsigned>(d) * 0x85ebca6bu);
  return (static_cast<float>(h >> 8) * (1.f / 16777216.f) - 0.5f) * 0.2f;
}
__host__ __device__ inline float lval(int t, unsigned row) { return tval(t, row, 77); }
__host__ __device__ inline double xval(long long b, int c, const int* card, int zipf) {
  const unsigned h = mix(mix(static_cast<unsigned>(b) * 2654435761u + 12345u) + static_cast<unsigned>(c) * 0x27d4eb2fu);
  if (c < 13) return static_cast<double>(h >> 8) * (1.0 / 16777216.0);
  if (c >= 39) return static_cast<double>(h & 1u);
  const int V = card[c - 13];
  if (!zipf) return static_cast<double>(1u + h % static_cast<unsigned>(V));              // uniform over 1..V
  const double u = static_cast<double>(h >> 8) * (1.0 / 16777216.0);                      // log-uniform: Zipf(1)-like
  double r = exp(u * log(static_cast<double>(V)));
  unsigned id = static_cast<unsigned>(r);
  if (id < 1) id = 1;
  if (id > static_cast<unsigned>(V)) id = V;
  return static_cast<double>(id);
}

__global__ void fill_table(float* w, float* lr, int t, long long rows) {
  const long long n = rows * kD;
  for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += static_cast<long long>(gridDim.x) * blockDim.x) {
    const unsigned row = static_cast<unsigned>(i / kD);
    const int d = static_cast<int>(i % kD);
    w[i] = tval(t, row, d);
    if (d == 0) lr[row] = lval(t, row);
  }
}
struct CardPack { int v[26]; };
__global__ void fill_batch(double* X, long long B, CardPack cp, int zipf, long long b0) {
  const long long n = B * kLd;
  for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += static_cast<long long>(gridDim.x) * blockDim.x)
    X[i] = xval(b0 + i / kLd, static_cast<int>(i % kLd), cp.v, zipf);
}

// ------------------------------------------------------------------------------------------------ variant 0
struct F0 { const double* col; const float* emb; const float* lr; int vocab; int kind; };
struct P0 { F0 f[39]; };

__device__ __forceinline__ float group_sum4(float v) {
  v += __shfl_xor(v, 1);
  v += __shfl_xor(v, 2);
  return v;
}

__global__ __launch_bounds__(256) void fm_v0(const P0 P, const int F, const long long B, const float* __restrict__ bias,
                                             float* __restrict__ logit, float* __restrict__ prob, float* __restrict__ ssum) {
  constexpr int G = 4, U = 8;
  const int lane_g = threadIdx.x % G;
  const long long ngroups = static_cast<long long>(gridDim.x) * (blockDim.x / G);
  for (long long b = static_cast<long long>(blockIdx.x) * (blockDim.x / G) + threadIdx.x / G; b < B; b += ngroups) {
    float s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
    float lr = 0.f;
    for (int f0 = 0; f0 < F; f0 += U) {
      long long raw[U];
      float x[U];
      int id[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const F0& fd = P.f[(f0 + u < F) ? f0 + u : F - 1];
        raw[u] = reinterpret_cast<const long long*>(fd.col)[b * kLd];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const F0& fd = P.f[(f0 + u < F) ? f0 + u : F - 1];
        const double d = __longlong_as_double(raw[u]);
        x[u] = 1.f;
        if (fd.kind == 1) {
          int v = __double2int_rz(d);
          if (!((d == d) && static_cast<unsigned>(v) < static_cast<unsigned>(fd.vocab))) { v = 0; x[u] = 0.f; }
          id[u] = v;
        } else {
          x[u] = static_cast<float>(d);
          id[u] = 0;
        }
      }
      float4 e[U];
      float l1[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        e[u] = make_float4(0, 0, 0, 0);
        l1[u] = 0.f;
        if (f0 + u < F) {
          const F0& fd = P.f[f0 + u];
          e[u] = *reinterpret_cast<const float4*>(fd.emb + static_cast<unsigned long long>(static_cast<unsigned>(id[u])) * 16u + lane_g * 4);
          if (lane_g == ((f0 + u) % G)) l1[u] = fd.lr[static_cast<unsigned>(id[u])];
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const float t0 = e[u].x * x[u], t1 = e[u].y * x[u], t2 = e[u].z * x[u], t3 = e[u].w * x[u];
        s[0] += t0; q[0] += t0 * t0;
        s[1] += t1; q[1] += t1 * t1;
        s[2] += t2; q[2] += t2 * t2;
        s[3] += t3; q[3] += t3 * t3;
        lr += l1[u] * x[u];
      }
    }
    float fm = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) fm += (s[i] * s[i] - q[i]) * 0.5f;
    const float total = group_sum4(fm + lr);
    if (lane_g == 0) {
      const float z = total + bias[0];
      logit[b] = z;
      prob[b] = 1.f / (1.f + expf(-z));
    }
    *reinterpret_cast<float4*>(ssum + b * 16 + lane_g * 4) = make_float4(s[0], s[1], s[2], s[3]);
  }
}

// ------------------------------------------------------------------------------------------------ quad forms
template <int K>
__device__ __forceinline__ unsigned bcast16(unsigned v) {        // lane K of every row of 16 lanes -> the row
  return static_cast<unsigned>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x150 + K, 0xF, 0xF, false));
}
template <int K>
__device__ __forceinline__ float bcast16f(float v) { return __uint_as_float(bcast16<K>(__float_as_uint(v))); }

__device__ __forceinline__ float row_sum16(float v) {            // sum over the 16 lanes of a DPP row, in every lane
  v += __shfl_xor(v, 1);
  v += __shfl_xor(v, 2);
  v += __shfl_xor(v, 4);
  v += __shfl_xor(v, 8);
  return v;
}

// Column metadata, one entry per column c < NS * 16 (device arrays; entries past F describe a harmless lookup of row 0 of a
// zero vector): every byte offset is relative to ONE base pointer (`arena`, the lowest address of all tables; the host checks
// that they span < 4 GiB), so that the feature loop needs NO per-feature scalar state: lane (c & 15) of a sample's row of
// 16 lanes computes the byte offset of the sample's row of table c, a DPP row broadcast hands it to the 16 lanes.
struct Meta {
  const int* vocab;          // 0 = numeric column (the value scales row 0 = the weight vector)
  const unsigned* emb_off;   // byte offset of the table (memory: from arena; LDS-resident: from the LDS image)
  const unsigned* emb_stride;
  const unsigned* lr_off;
  const unsigned* lr_stride;
  unsigned long long lds_mask;   // bit c: column c's table and LR weights are read from LDS
};

template <int K, bool LDS, bool FULL>
__device__ __forceinline__ void issue_rows(const int c0, const int F, const unsigned long long mask, const unsigned o,
                                           const unsigned lane4, const char* __restrict__ arena, const char* lds,
                                           float (&e)[16]) {
  if constexpr (K < 16) {
    e[K] = 0.f;
    if (FULL || c0 + K < F) {                // wave-uniform
      const unsigned off = bcast16<K>(o) + lane4;
      if (LDS && ((mask >> (c0 + K)) & 1ull)) e[K] = *reinterpret_cast<const float*>(lds + off);
      else e[K] = *reinterpret_cast<const float*>(arena + static_cast<size_t>(off));
    }
    issue_rows<K + 1, LDS, FULL>(c0, F, mask, o, lane4, arena, lds, e);
  }
}
template <int K>
__device__ __forceinline__ void use_rows(const float x, const float (&e)[16], float& s, float& q) {
  if constexpr (K < 16) {
    const float t = e[K] * bcast16f<K>(x);
    s += t;
    q += t * t;
    use_rows<K + 1>(x, e, s, q);
  }
}

template <bool LDS, int NS>
__global__ __launch_bounds__(LDS ? 1024 : 256) void fm_quad(const Meta M, const char* __restrict__ arena, const int F,
                                                           const double* __restrict__ X, const int ldx, const long long B,
                                                           const float* __restrict__ bias, float* __restrict__ logit,
                                                           float* __restrict__ prob, float* __restrict__ ssum,
                                                           const float4* __restrict__ lds_image, const int lds_bytes) {
  extern __shared__ char lds[];
  if constexpr (LDS) {
    // every thread issues ALL of its loads of the image, then stores (8 x 16 KiB per sweep of the 1024 threads)
    const int n16 = lds_bytes / 16;
    for (int base = 0; base < n16; base += 8 * 1024) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = base + u * 1024 + threadIdx.x;
        v[u] = (i < n16) ? lds_image[i] : make_float4(0, 0, 0, 0);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = base + u * 1024 + threadIdx.x;
        if (i < n16) reinterpret_cast<float4*>(lds)[i] = v[u];
      }
    }
    __syncthreads();
  }
  const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
  const unsigned lane4 = j * 4;
  int voc[NS], cc[NS];
  unsigned eo[NS], es[NS], lo[NS], ls[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int c = s * 16 + j;
    cc[s] = (c < F) ? c : F - 1;
    voc[s] = M.vocab[c];
    eo[s] = M.emb_off[c]; es[s] = M.emb_stride[c];
    lo[s] = M.lr_off[c]; ls[s] = M.lr_stride[c];
  }
  const long long nq = (B + 3) / 4;
  const long long wave = (static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
  const long long nw = (static_cast<long long>(gridDim.x) * blockDim.x) >> 6;
  for (long long qd = wave; qd < nq; qd += nw) {
    const long long b = qd * 4 + g;
    const bool live = b < B;
    const double* xr = X + (live ? b : B - 1) * ldx;
    double c[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) c[s] = xr[cc[s]];
    unsigned o[NS];
    float x[NS], l1[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int v = __double2int_rz(c[s]);
      const bool ok = (c[s] == c[s]) && static_cast<unsigned>(v) < static_cast<unsigned>(voc[s]);   // false for numeric
      const unsigned id = ok ? static_cast<unsigned>(v) : 0u;
      x[s] = (voc[s] > 0) ? (ok ? 1.f : 0.f) : static_cast<float>(c[s]);
      o[s] = eo[s] + id * es[s];
      const unsigned lro = lo[s] + id * ls[s];
      if (LDS && ((M.lds_mask >> (s * 16 + j)) & 1ull)) l1[s] = *reinterpret_cast<const float*>(lds + lro);
      else l1[s] = *reinterpret_cast<const float*>(arena + static_cast<size_t>(lro));
    }
    float sm = 0.f, qs = 0.f;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      float e[16];
      if ((s + 1) * 16 <= F) issue_rows<0, LDS, true>(s * 16, F, M.lds_mask, o[s], lane4, arena, lds, e);
      else issue_rows<0, LDS, false>(s * 16, F, M.lds_mask, o[s], lane4, arena, lds, e);
      use_rows<0>(x[s], e, sm, qs);
    }
    float lr = 0.f;
#pragma unroll
    for (int s = 0; s < NS; ++s) lr += l1[s] * x[s];
    const float total = row_sum16((sm * sm - qs) * 0.5f + lr);
    if (live) {
      if (j == 0) {
        const float z = total + bias[0];
        logit[b] = z;
        prob[b] = 1.f / (1.f + expf(-z));
      }
      ssum[b * 16 + j] = sm;
    }
  }
}

// ------------------------------------------------------------------------------------------------ g4: lane group of 4, DPP quad broadcast
// A sample = 4 lanes (float4 each: the fewest instructions per sample), 16 samples per wavefront.  Lane j of a group reads
// the sample's columns j, 4 + j, 8 + j, ... (consecutive lanes = consecutive columns: a group's loads of one step cover 32
// contiguous bytes, the NI steps of a sample its whole batch row, fetched from memory once), decodes ITS columns -- row byte
// offset from `arena`, LR byte offset; an invalid id or a column past F points at the zero vector at offset 0 -- and issues
// its columns' LR loads.  The feature loop then needs no per-feature state at all: `quad_perm` broadcasts the owner lane's
// offset to the group (folded into the address add), UB rows in flight per lane, features summed in feature order.
template <int J>
__device__ __forceinline__ unsigned qbcast(unsigned v) {
  return static_cast<unsigned>(__builtin_amdgcn_mov_dpp(static_cast<int>(v), J | (J << 2) | (J << 4) | (J << 6), 0xF, 0xF, true));
}
template <int J>
__device__ __forceinline__ float qbcastf(float v) { return __uint_as_float(qbcast<J>(__float_as_uint(v))); }

template <int F0, int U, int UB>
__device__ __forceinline__ void g4_issue(const unsigned (&o)[16], const unsigned lane16, const char* __restrict__ arena,
                                         float4 (&e)[UB]) {
  if constexpr (U < UB) {
    constexpr int f = F0 + U;
    const unsigned off = qbcast<(f & 3)>(o[f >> 2]) + lane16;
    e[U] = *reinterpret_cast<const float4*>(arena + static_cast<size_t>(off));
    g4_issue<F0, U + 1, UB>(o, lane16, arena, e);
  }
}
template <int F0, int U, int UB>
__device__ __forceinline__ void g4_use(const float (&x)[16], const float4 (&e)[UB], float (&s)[4], float (&q)[4]) {
  if constexpr (U < UB) {
    constexpr int f = F0 + U;
    const float xb = qbcastf<(f & 3)>(x[f >> 2]);
    const float t0 = e[U].x * xb, t1 = e[U].y * xb, t2 = e[U].z * xb, t3 = e[U].w * xb;
    s[0] += t0; q[0] += t0 * t0;
    s[1] += t1; q[1] += t1 * t1;
    s[2] += t2; q[2] += t2 * t2;
    s[3] += t3; q[3] += t3 * t3;
    g4_use<F0, U + 1, UB>(x, e, s, q);
  }
}
template <int F0, int NF, int UB>
__device__ __forceinline__ void g4_batches(const unsigned (&o)[16], const float (&x)[16], const unsigned lane16,
                                           const char* __restrict__ arena, float (&s)[4], float (&q)[4]) {
  if constexpr (F0 < NF) {
    constexpr int N = (NF - F0 < UB) ? NF - F0 : UB;
    float4 e[N];
    g4_issue<F0, 0, N>(o, lane16, arena, e);
    g4_use<F0, 0, N>(x, e, s, q);
    __builtin_amdgcn_sched_barrier(0);
    g4_batches<F0 + N, NF, UB>(o, x, lane16, arena, s, q);
  }
}

template <int NI, int UB, int WPE>
__global__ __launch_bounds__(256, WPE) void fm_g4(const Meta M, const char* __restrict__ arena, const int F,
                                                  const double* __restrict__ X, const int ldx, const long long B,
                                                  const float* __restrict__ bias, float* __restrict__ logit,
                                                  float* __restrict__ prob, float* __restrict__ ssum) {
  __shared__ int s_voc[64];
  __shared__ unsigned s_eo[64], s_es[64], s_lo[64], s_ls[64];
  if (threadIdx.x < 64) {
    s_voc[threadIdx.x] = M.vocab[threadIdx.x];
    s_eo[threadIdx.x] = M.emb_off[threadIdx.x];
    s_es[threadIdx.x] = M.emb_stride[threadIdx.x];
    s_lo[threadIdx.x] = M.lr_off[threadIdx.x];
    s_ls[threadIdx.x] = M.lr_stride[threadIdx.x];
  }
  __syncthreads();
  const int lane_g = threadIdx.x & 3;
  const unsigned lane16 = lane_g * 16;
  const long long ngroups = static_cast<long long>(gridDim.x) * (blockDim.x / 4);
  for (long long b = static_cast<long long>(blockIdx.x) * (blockDim.x / 4) + threadIdx.x / 4; b < B; b += ngroups) {
    const double* xr = X + b * ldx;
    double c[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int col = 4 * i + lane_g;
      c[i] = xr[col < F ? col : F - 1];
    }
    unsigned o[16];
    float x[16], l1[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int col = 4 * i + lane_g;
      const int voc = s_voc[col];
      const int v = __double2int_rz(c[i]);
      const bool ok = (c[i] == c[i]) && static_cast<unsigned>(v) < static_cast<unsigned>(voc);     // false for numeric / padding columns
      const bool num = voc == 0 && col < F;
      const unsigned id = ok ? static_cast<unsigned>(v) : 0u;
      x[i] = num ? static_cast<float>(c[i]) : 1.f;
      o[i] = (ok || num) ? s_eo[col] + id * s_es[col] : 0u;             // invalid id / padding column: the zero vector
      const unsigned lro = (ok || num) ? s_lo[col] + id * s_ls[col] : 0u;
      l1[i] = *reinterpret_cast<const float*>(arena + static_cast<size_t>(lro));
    }
    float s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
    g4_batches<0, 4 * NI, UB>(o, x, lane16, arena, s, q);
    float lr = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) lr += l1[i] * x[i];
    float fm = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) fm += (s[i] * s[i] - q[i]) * 0.5f;
    const float total = group_sum4(fm + lr);
    if (lane_g == 0) {
      const float z = total + bias[0];
      logit[b] = z;
      prob[b] = 1.f / (1.f + expf(-z));
    }
    *reinterpret_cast<float4*>(ssum + b * 16 + lane_g * 4) = make_float4(s[0], s[1], s[2], s[3]);
  }
}

// ------------------------------------------------------------------------------------------------ g4s: g4 + scalar-path prefetch into L2
// The vector memory path of a CU holds ~64 misses; the SCALAR data cache is a second path into the same L2.  After the
// decode, the wavefront walks the columns named by `pmask` (rows) / `lmask` (LR weights) and issues one s_load_dword per
// (sample, column) at the row's address: the line travels HBM -> L2 on the scalar cache's account, and the vector load that
// follows finds it in L2 (~250 cycles) instead of holding a miss slot for the HBM latency (~900).  The loaded dword is
// discarded (s101: a register the compiler never allocates here; lgkmcnt holds 15 in flight per wavefront).
template <int C>
__device__ __forceinline__ void spf_col(const unsigned (&o)[16], const char* __restrict__ arena) {
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const unsigned off = static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(o[C >> 2]), j * 4 + (C & 3)));
    asm volatile("s_load_dword s101, %0, %1" :: "s"(arena), "s"(off) : "s101");
  }
}
template <int C, int NC>
__device__ __forceinline__ void spf_all(const unsigned long long mask, const unsigned (&o)[16], const char* __restrict__ arena) {
  if constexpr (C < NC) {
    if ((mask >> C) & 1ull) spf_col<C>(o, arena);
    spf_all<C + 1, NC>(mask, o, arena);
  }
}

template <int NI, int UB, int WPE>
__global__ __launch_bounds__(256, WPE) void fm_g4s(const Meta M, const char* __restrict__ arena, const int F,
                                                   const double* __restrict__ X, const int ldx, const long long B,
                                                   const float* __restrict__ bias, float* __restrict__ logit,
                                                   float* __restrict__ prob, float* __restrict__ ssum,
                                                   const unsigned long long pmask, const unsigned long long lmask) {
  __shared__ int s_voc[64];
  __shared__ unsigned s_eo[64], s_es[64], s_lo[64], s_ls[64];
  if (threadIdx.x < 64) {
    s_voc[threadIdx.x] = M.vocab[threadIdx.x];
    s_eo[threadIdx.x] = M.emb_off[threadIdx.x];
    s_es[threadIdx.x] = M.emb_stride[threadIdx.x];
    s_lo[threadIdx.x] = M.lr_off[threadIdx.x];
    s_ls[threadIdx.x] = M.lr_stride[threadIdx.x];
  }
  __syncthreads();
  const int lane_g = threadIdx.x & 3;
  const unsigned lane16 = lane_g * 16;
  const long long ngroups = static_cast<long long>(gridDim.x) * (blockDim.x / 4);
  for (long long b = static_cast<long long>(blockIdx.x) * (blockDim.x / 4) + threadIdx.x / 4; b < B; b += ngroups) {
    const double* xr = X + b * ldx;
    double c[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int col = 4 * i + lane_g;
      c[i] = xr[col < F ? col : F - 1];
    }
    unsigned o[16], lo[16];
    float x[16], l1[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int col = 4 * i + lane_g;
      const int voc = s_voc[col];
      const int v = __double2int_rz(c[i]);
      const bool ok = (c[i] == c[i]) && static_cast<unsigned>(v) < static_cast<unsigned>(voc);
      const bool num = voc == 0 && col < F;
      const unsigned id = ok ? static_cast<unsigned>(v) : 0u;
      x[i] = num ? static_cast<float>(c[i]) : 1.f;
      o[i] = (ok || num) ? s_eo[col] + id * s_es[col] : 0u;
      lo[i] = (ok || num) ? s_lo[col] + id * s_ls[col] : 0u;
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)");
    spf_all<0, 4 * NI>(pmask, o, arena);
    spf_all<0, 4 * NI>(lmask, lo, arena);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NI; ++i) l1[i] = *reinterpret_cast<const float*>(arena + static_cast<size_t>(lo[i]));
    float s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
    g4_batches<0, 4 * NI, UB>(o, x, lane16, arena, s, q);
    float lr = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) lr += l1[i] * x[i];
    float fm = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) fm += (s[i] * s[i] - q[i]) * 0.5f;
    const float total = group_sum4(fm + lr);
    if (lane_g == 0) {
      const float z = total + bias[0];
      logit[b] = z;
      prob[b] = 1.f / (1.f + expf(-z));
    }
    *reinterpret_cast<float4*>(ssum + b * 16 + lane_g * 4) = make_float4(s[0], s[1], s[2], s[3]);
  }
  asm volatile("s_waitcnt lgkmcnt(0)");
}

// ------------------------------------------------------------------------------------------------ g4n: numeric features off the memory path
// g4 with the first NNUM columns numeric (compile-time here: what the form can gain): their weight vectors and LR weights sit
// in LDS, their contribution x_f w_f is accumulated first (feature order), no row / LR lookup is issued for them.
template <int F, int NNUM>
__device__ __forceinline__ void g4n_numeric(const float (&x)[16], const float4* s_w, const int lane_g, float (&s)[4], float (&q)[4]) {
  if constexpr (F < NNUM) {
    const float xb = qbcastf<(F & 3)>(x[F >> 2]);
    const float4 w = s_w[F * 4 + lane_g];
    const float t0 = w.x * xb, t1 = w.y * xb, t2 = w.z * xb, t3 = w.w * xb;
    s[0] += t0; q[0] += t0 * t0;
    s[1] += t1; q[1] += t1 * t1;
    s[2] += t2; q[2] += t2 * t2;
    s[3] += t3; q[3] += t3 * t3;
    g4n_numeric<F + 1, NNUM>(x, s_w, lane_g, s, q);
  }
}

template <int NI, int UB, int WPE, int NNUM>
__global__ __launch_bounds__(256, WPE) void fm_g4n(const Meta M, const char* __restrict__ arena, const int F,
                                                   const double* __restrict__ X, const int ldx, const long long B,
                                                   const float* __restrict__ bias, float* __restrict__ logit,
                                                   float* __restrict__ prob, float* __restrict__ ssum) {
  __shared__ int s_voc[64];
  __shared__ unsigned s_eo[64], s_es[64], s_lo[64], s_ls[64];
  __shared__ float4 s_w[16 * 4];
  __shared__ float s_wl[16];
  if (threadIdx.x < 64) {
    s_voc[threadIdx.x] = M.vocab[threadIdx.x];
    s_eo[threadIdx.x] = M.emb_off[threadIdx.x];
    s_es[threadIdx.x] = M.emb_stride[threadIdx.x];
    s_lo[threadIdx.x] = M.lr_off[threadIdx.x];
    s_ls[threadIdx.x] = M.lr_stride[threadIdx.x];
    if (threadIdx.x < NNUM * 4)
      s_w[threadIdx.x] = *reinterpret_cast<const float4*>(arena + M.emb_off[threadIdx.x >> 2] + (threadIdx.x & 3) * 16);
    if (threadIdx.x < NNUM) s_wl[threadIdx.x] = *reinterpret_cast<const float*>(arena + M.lr_off[threadIdx.x]);
  }
  __syncthreads();
  const int lane_g = threadIdx.x & 3;
  const unsigned lane16 = lane_g * 16;
  const long long ngroups = static_cast<long long>(gridDim.x) * (blockDim.x / 4);
  for (long long b = static_cast<long long>(blockIdx.x) * (blockDim.x / 4) + threadIdx.x / 4; b < B; b += ngroups) {
    const double* xr = X + b * ldx;
    double c[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int col = 4 * i + lane_g;
      c[i] = xr[col < F ? col : F - 1];
    }
    unsigned o[16];
    float x[16], l1[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int col = 4 * i + lane_g;
      const int voc = s_voc[col];
      const int v = __double2int_rz(c[i]);
      const bool ok = (c[i] == c[i]) && static_cast<unsigned>(v) < static_cast<unsigned>(voc);
      const bool num = col < NNUM;
      const unsigned id = ok ? static_cast<unsigned>(v) : 0u;
      x[i] = num ? static_cast<float>(c[i]) : 1.f;
      o[i] = ok ? s_eo[col] + id * s_es[col] : 0u;
      const unsigned lro = ok ? s_lo[col] + id * s_ls[col] : 0u;
      if (num) l1[i] = s_wl[col];
      else l1[i] = *reinterpret_cast<const float*>(arena + static_cast<size_t>(lro));
    }
    float s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
    g4n_numeric<0, NNUM>(x, s_w, lane_g, s, q);
    g4_batches<NNUM, 4 * NI, UB>(o, x, lane16, arena, s, q);
    float lr = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) lr += l1[i] * x[i];
    float fm = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) fm += (s[i] * s[i] - q[i]) * 0.5f;
    const float total = group_sum4(fm + lr);
    if (lane_g == 0) {
      const float z = total + bias[0];
      logit[b] = z;
      prob[b] = 1.f / (1.f + expf(-z));
    }
    *reinterpret_cast<float4*>(ssum + b * 16 + lane_g * 4) = make_float4(s[0], s[1], s[2], s[3]);
  }
}

// ------------------------------------------------------------------------------------------------ g4l: g4 + small arrays in LDS
// As g4, with the arrays that cost a cache-line lookup per sample but hold few bytes -- the numeric features' weight vectors
// and LR weights, the rows and LR weights of the smallest tables -- copied into LDS by the workgroup's prologue (a copy list of
// 16-byte units: LDS unit u <- arena + src[u]), so that their lookups leave the vector memory path.  `emb_mask` / `lr_mask`:
// bit c set = column c's rows / LR weights are read from LDS (its offsets in the metadata are then LDS byte offsets).
struct MetaL {
  Meta m;
  unsigned long long emb_mask, lr_mask;
  const unsigned* copy_src;
  int n_units;
};

template <int F0, int U, int UB>
__device__ __forceinline__ void g4l_issue(const unsigned (&o)[16], const unsigned lane16, const char* __restrict__ arena,
                                          const char* img, const unsigned long long emb_mask, float4 (&e)[UB]) {
  if constexpr (U < UB) {
    constexpr int f = F0 + U;
    const unsigned off = qbcast<(f & 3)>(o[f >> 2]) + lane16;
    if ((emb_mask >> f) & 1ull) e[U] = *reinterpret_cast<const float4*>(img + off);
    else e[U] = *reinterpret_cast<const float4*>(arena + static_cast<size_t>(off));
    g4l_issue<F0, U + 1, UB>(o, lane16, arena, img, emb_mask, e);
  }
}
template <int F0, int NF, int UB>
__device__ __forceinline__ void g4l_batches(const unsigned (&o)[16], const float (&x)[16], const unsigned lane16,
                                            const char* __restrict__ arena, const char* img,
                                            const unsigned long long emb_mask, float (&s)[4], float (&q)[4]) {
  if constexpr (F0 < NF) {
    constexpr int N = (NF - F0 < UB) ? NF - F0 : UB;
    float4 e[N];
    g4l_issue<F0, 0, N>(o, lane16, arena, img, emb_mask, e);
    g4_use<F0, 0, N>(x, e, s, q);
    __builtin_amdgcn_sched_barrier(0);
    g4l_batches<F0 + N, NF, UB>(o, x, lane16, arena, img, emb_mask, s, q);
  }
}

template <int NI, int UB, int BS, int WPE>
__global__ __launch_bounds__(BS, WPE) void fm_g4l(const MetaL ML, const char* __restrict__ arena, const int F,
                                                  const double* __restrict__ X, const int ldx, const long long B,
                                                  const float* __restrict__ bias, float* __restrict__ logit,
                                                  float* __restrict__ prob, float* __restrict__ ssum) {
  extern __shared__ char dyn[];
  int* s_voc = reinterpret_cast<int*>(dyn);
  unsigned* s_eo = reinterpret_cast<unsigned*>(dyn) + 64;
  unsigned* s_es = s_eo + 64;
  unsigned* s_lo = s_es + 64;
  unsigned* s_ls = s_lo + 64;
  char* img = dyn + 5 * 64 * 4;
  if (threadIdx.x < 64) {
    s_voc[threadIdx.x] = ML.m.vocab[threadIdx.x];
    s_eo[threadIdx.x] = ML.m.emb_off[threadIdx.x];
    s_es[threadIdx.x] = ML.m.emb_stride[threadIdx.x];
    s_lo[threadIdx.x] = ML.m.lr_off[threadIdx.x];
    s_ls[threadIdx.x] = ML.m.lr_stride[threadIdx.x];
  }
  {
    constexpr int CU_ = 6;                     // units per thread per sweep: index loads, then data loads, then LDS stores
    for (int base = 0; base < ML.n_units; base += CU_ * BS) {
      unsigned so[CU_];
#pragma unroll
      for (int u = 0; u < CU_; ++u) {
        const int i = base + u * BS + threadIdx.x;
        so[u] = ML.copy_src[i < ML.n_units ? i : ML.n_units - 1];
      }
      float4 v[CU_];
#pragma unroll
      for (int u = 0; u < CU_; ++u) v[u] = *reinterpret_cast<const float4*>(arena + static_cast<size_t>(so[u]));
#pragma unroll
      for (int u = 0; u < CU_; ++u) {
        const int i = base + u * BS + threadIdx.x;
        if (i < ML.n_units) reinterpret_cast<float4*>(img)[i] = v[u];
      }
    }
  }
  __syncthreads();
  const int lane_g = threadIdx.x & 3;
  const unsigned lane16 = lane_g * 16;
  const unsigned long long emb_mask = ML.emb_mask, lr_mask = ML.lr_mask;
  const long long ngroups = static_cast<long long>(gridDim.x) * (BS / 4);
  for (long long b = static_cast<long long>(blockIdx.x) * (BS / 4) + threadIdx.x / 4; b < B; b += ngroups) {
    const double* xr = X + b * ldx;
    double c[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int col = 4 * i + lane_g;
      c[i] = xr[col < F ? col : F - 1];
    }
    unsigned o[16];
    float x[16], l1[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int col = 4 * i + lane_g;
      const int voc = s_voc[col];
      const int v = __double2int_rz(c[i]);
      const bool ok = (c[i] == c[i]) && static_cast<unsigned>(v) < static_cast<unsigned>(voc);
      const bool num = voc == 0 && col < F;
      const bool use = ok || num;
      const unsigned id = ok ? static_cast<unsigned>(v) : 0u;
      x[i] = num ? static_cast<float>(c[i]) : 1.f;
      const bool e_lds = (emb_mask >> col) & 1ull, l_lds = (lr_mask >> col) & 1ull;
      // an invalid id / a column past F reads the zero vector: LDS unit 0 and arena offset 0 both hold one
      o[i] = use ? s_eo[col] + id * s_es[col] : 0u;
      const unsigned lro = use ? s_lo[col] + id * s_ls[col] : 0u;
      l1[i] = l_lds ? *reinterpret_cast<const float*>(img + lro) : *reinterpret_cast<const float*>(arena + static_cast<size_t>(lro));
      (void)e_lds;
    }
    float s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
    g4l_batches<0, 4 * NI, UB>(o, x, lane16, arena, img, emb_mask, s, q);
    float lr = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) lr += l1[i] * x[i];
    float fm = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) fm += (s[i] * s[i] - q[i]) * 0.5f;
    const float total = group_sum4(fm + lr);
    if (lane_g == 0) {
      const float z = total + bias[0];
      logit[b] = z;
      prob[b] = 1.f / (1.f + expf(-z));
    }
    *reinterpret_cast<float4*>(ssum + b * 16 + lane_g * 4) = make_float4(s[0], s[1], s[2], s[3]);
  }
}

// ------------------------------------------------------------------------------------------------ host
static const int kRot = 8;            // distinct batches replayed in rotation (as bench.py does)

static void host_ref(long long b, int zipf, float* z_out, float* s_out) {
  float s[16], q[16];
  for (int d = 0; d < 16; ++d) s[d] = q[d] = 0.f;
  float lr = 0.f;
  for (int f = 0; f < kF; ++f) {
    const double xv = xval(b, f, kCard, zipf);
    float x = 1.f;
    unsigned id = 0;
    if (f < 13) x = static_cast<float>(xv); else id = static_cast<unsigned>(xv);
    for (int d = 0; d < 16; ++d) {
      const float t = tval(f, id, d) * x;
      s[d] += t;
      q[d] += t * t;
    }
    lr += lval(f, id) * x;
  }
  float fm = 0.f;
  for (int d = 0; d < 16; ++d) fm += (s[d] * s[d] - q[d]) * 0.5f;
  *z_out = fm + lr + 0.125f;
  for (int d = 0; d < 16; ++d) s_out[d] = s[d];
}

__global__ void fill_packed(float* w, int t, long long rows) {      // [rows, 32]: 16 row floats, the LR weight, 15 unused
  const long long n = rows * 32;
  for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += static_cast<long long>(gridDim.x) * blockDim.x) {
    const unsigned row = static_cast<unsigned>(i / 32);
    const int d = static_cast<int>(i % 32);
    w[i] = d < 16 ? tval(t, row, d) : (d == 16 ? lval(t, row) : 0.f);
  }
}

struct MetaHost {
  std::vector<int> vocab;
  std::vector<unsigned> eo, es, lo, ls;
  unsigned long long mask = 0;
  Meta dev{};
  void upload() {
    int* v; unsigned *a, *b, *c, *d;
    CK(hipMalloc(&v, 256)); CK(hipMalloc(&a, 256)); CK(hipMalloc(&b, 256)); CK(hipMalloc(&c, 256)); CK(hipMalloc(&d, 256));
    CK(hipMemcpy(v, vocab.data(), 256, hipMemcpyHostToDevice));
    CK(hipMemcpy(a, eo.data(), 256, hipMemcpyHostToDevice));
    CK(hipMemcpy(b, es.data(), 256, hipMemcpyHostToDevice));
    CK(hipMemcpy(c, lo.data(), 256, hipMemcpyHostToDevice));
    CK(hipMemcpy(d, ls.data(), 256, hipMemcpyHostToDevice));
    dev = Meta{v, a, b, c, d, mask};
  }
};

int main(int argc, char** argv) {
  const int only = argc > 1 ? atoi(argv[1]) : -1;
  const int iters = argc > 2 ? atoi(argv[2]) : 40;
  const int lds_rows = argc > 3 ? atoi(argv[3]) : 640;
  const long long B = 65536;
  // ONE arena: [zero vector | tables | LR tables | packed tables]
  std::vector<size_t> emb_at(kF), lr_at(kF), pk_at(kF);
  size_t at = 256;
  for (int f = 0; f < kF; ++f) { const size_t rows = f < 13 ? 1 : kCard[f - 13] + 1; emb_at[f] = at; at += (rows * 64 + 255) / 256 * 256; }
  for (int f = 0; f < kF; ++f) { const size_t rows = f < 13 ? 1 : kCard[f - 13] + 1; lr_at[f] = at; at += (rows * 4 + 255) / 256 * 256; }
  for (int f = 0; f < kF; ++f) { const size_t rows = f < 13 ? 1 : kCard[f - 13] + 1; pk_at[f] = at; at += (rows * 128 + 255) / 256 * 256; }
  char* arena;
  CK(hipMalloc(&arena, at));
  CK(hipMemset(arena, 0, 256));
  for (int f = 0; f < kF; ++f) {
    const long long rows = f < 13 ? 1 : kCard[f - 13] + 1;
    fill_table<<<1024, 256>>>(reinterpret_cast<float*>(arena + emb_at[f]), reinterpret_cast<float*>(arena + lr_at[f]), f, rows);
    fill_packed<<<1024, 256>>>(reinterpret_cast<float*>(arena + pk_at[f]), f, rows);
  }
  CardPack cp;
  memcpy(cp.v, kCard, sizeof(kCard));
  double* X[2][kRot];
  for (int z = 0; z < 2; ++z)
    for (int r = 0; r < kRot; ++r) {
      CK(hipMalloc(&X[z][r], B * kLd * 8));
      fill_batch<<<1024, 256>>>(X[z][r], B, cp, z, r * B);
    }
  float *bias, *logit, *prob, *ssum;
  CK(hipMalloc(&bias, 4));
  const float hb = 0.125f;
  CK(hipMemcpy(bias, &hb, 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&logit, B * 4));
  CK(hipMalloc(&prob, B * 4));
  CK(hipMalloc(&ssum, B * 64));
  CK(hipDeviceSynchronize());
  printf("B %lld, arena %.1f MB, %d batches in rotation\n", B, at / 1e6, kRot);

  P0 p0[2][kRot];
  for (int z = 0; z < 2; ++z)
    for (int r = 0; r < kRot; ++r)
      for (int f = 0; f < kF; ++f) {
        F0& d = p0[z][r].f[f];
        d.col = X[z][r] + f;
        d.emb = reinterpret_cast<const float*>(arena + emb_at[f]);
        d.lr = reinterpret_cast<const float*>(arena + lr_at[f]);
        d.vocab = f < 13 ? 0 : kCard[f - 13] + 1;
        d.kind = f < 13 ? 0 : 1;
      }

  // metadata of the quad forms.  flags: 1 packed rows (the 128-byte row carries the LR weight) for tables of > lds_rows rows,
  // 2 small tables in LDS, ablations (they compute WRONG results, timing only): 16 no LR misses (every LR lookup reads row 0),
  // 32 no row misses for tables > 100 000 rows, 64 no row misses at all
  std::vector<float> image;
  auto make = [&](int flags, bool fill_image) {
    MetaHost m;
    m.vocab.assign(64, 0); m.eo.assign(64, 0); m.es.assign(64, 0); m.lo.assign(64, 0); m.ls.assign(64, 0);
    std::vector<float> img;
    for (int f = 0; f < kF; ++f) {
      const int rows = f < 13 ? 1 : kCard[f - 13] + 1;
      m.vocab[f] = f < 13 ? 0 : rows;
      const bool in_lds = (flags & 2) && rows <= lds_rows;
      const bool packed = (flags & 1) && !in_lds && f >= 13;
      if (in_lds) {
        m.mask |= 1ull << f;
        m.eo[f] = static_cast<unsigned>(img.size() * 4); m.es[f] = 64;
        for (int r = 0; r < rows; ++r) for (int d = 0; d < 16; ++d) img.push_back(tval(f, r, d));
      } else if (packed) {
        m.eo[f] = static_cast<unsigned>(pk_at[f]); m.es[f] = 128;
        m.lo[f] = static_cast<unsigned>(pk_at[f] + 64); m.ls[f] = 128;
      } else {
        m.eo[f] = static_cast<unsigned>(emb_at[f]); m.es[f] = 64;
        m.lo[f] = static_cast<unsigned>(lr_at[f]); m.ls[f] = 4;
      }
      if ((flags & 64) || ((flags & 32) && rows > 100000)) m.es[f] = 0;
      if (flags & 16) m.ls[f] = 0;
    }
    for (int f = 0; f < kF; ++f) {
      const int rows = f < 13 ? 1 : kCard[f - 13] + 1;
      if ((flags & 2) && rows <= lds_rows) {
        m.lo[f] = static_cast<unsigned>(img.size() * 4); m.ls[f] = 4;
        for (int r = 0; r < rows; ++r) img.push_back(lval(f, r));
      }
    }
    while (img.size() % 4) img.push_back(0.f);
    if (fill_image) image = img;
    m.upload();
    return m;
  };

  // plan of the g4l forms: arrays (rows of a table / its LR weights) ascending by bytes into an LDS budget; numeric first
  struct LPlan { MetaL dev; int lds_bytes; int n_emb, n_lr; };
  auto make_l = [&](int budget_bytes) {
    MetaHost m;
    m.vocab.assign(64, 0); m.eo.assign(64, 0); m.es.assign(64, 0); m.lo.assign(64, 0); m.ls.assign(64, 0);
    struct Arr { size_t bytes; int f; int lr; };
    std::vector<Arr> arrs;
    for (int f = 0; f < kF; ++f) {
      const size_t rows = f < 13 ? 1 : kCard[f - 13] + 1;
      arrs.push_back({rows * 64, f, 0});
      arrs.push_back({(rows * 4 + 15) / 16 * 16, f, 1});
      m.vocab[f] = f < 13 ? 0 : static_cast<int>(rows);
      m.eo[f] = static_cast<unsigned>(emb_at[f]); m.es[f] = 64;
      m.lo[f] = static_cast<unsigned>(lr_at[f]); m.ls[f] = 4;
    }
    std::sort(arrs.begin(), arrs.end(), [](const Arr& a, const Arr& b) { return a.bytes < b.bytes; });
    std::vector<unsigned> src;
    for (int u = 0; u < 4; ++u) src.push_back(0);          // LDS units 0..3: zeros (arena offset 0..63 is a zero vector)
    unsigned long long em = 0, lm = 0;
    int ne = 0, nl = 0;
    for (const Arr& a : arrs) {
      if (src.size() * 16 + a.bytes > static_cast<size_t>(budget_bytes)) break;
      const unsigned at_lds = static_cast<unsigned>(src.size() * 16);
      const size_t from = a.lr ? lr_at[a.f] : emb_at[a.f];
      for (size_t u = 0; u < a.bytes / 16; ++u) src.push_back(static_cast<unsigned>(from + u * 16));
      if (a.lr) { m.lo[a.f] = at_lds; lm |= 1ull << a.f; ++nl; } else { m.eo[a.f] = at_lds; em |= 1ull << a.f; ++ne; }
    }
    m.upload();
    unsigned* d_src;
    CK(hipMalloc(&d_src, src.size() * 4));
    CK(hipMemcpy(d_src, src.data(), src.size() * 4, hipMemcpyHostToDevice));
    LPlan p;
    p.dev = MetaL{m.dev, em, lm, d_src, static_cast<int>(src.size())};
    p.lds_bytes = static_cast<int>(src.size() * 16) + 5 * 64 * 4;
    p.n_emb = ne; p.n_lr = nl;
    return p;
  };
  struct Var { int kernel; int flags; const char* name; int grid; bool check; };
  std::vector<Var> vars = {
      {0, 0, "v0 lane group of 4, 8 features in flight, ids in place (round 4 form)", 1024, true},
      {1, 0, "v1 quad, 16384 waves (one quad each)", 4096, true},
      {1, 0, "v1 quad, 8192 waves", 2048, true},
      {1, 1, "v1p quad, rows of the larger tables packed with their LR weight (128 B)", 4096, true},
      {1, 1, "v1p quad packed, 8192 waves", 2048, true},
      {2, 2, "v2 quad + small tables in LDS, 256 workgroups x 1024", 256, true},
      {2, 3, "v2p quad + small tables in LDS + the rest packed", 256, true},
      {3, 0, "g4: lane group of 4 + quad broadcast, 10 rows in flight, 1024 workgroups", 1024, true},
      {4, 0, "g4: 14 rows in flight, 1024 workgroups", 1024, true},
      {5, 0, "g4: 20 rows in flight (3 waves/SIMD), 1024 workgroups", 1024, true},
      {6, 0, "g4: 8 rows in flight (5 waves/SIMD), 1024 workgroups", 1024, true},
      {20, 2, "g4l: numeric weights only in LDS (2 KB), 256-thread workgroups x 768", 768, true},
      {20, 48, "g4l: 48 KB of small arrays in LDS, 256-thread workgroups x 768 (3 per CU)", 768, true},
      {21, 48, "g4l: 48 KB in LDS, 768-thread workgroups x 256 (1 per CU)", 256, true},
      {21, 150, "g4l: 150 KB in LDS, 768-thread workgroups x 256 (1 per CU)", 256, true},
      {22, 48, "g4l: 48 KB in LDS, 10 in flight, 512-thread workgroups x 512 (2 per CU)", 512, true},
      {23, 150, "g4l: 150 KB in LDS, 10 in flight, 1024-thread workgroups x 256", 256, true},
      {21, 20, "g4l: 20 KB in LDS, 768-thread workgroups x 256", 256, true},
      {9, 0, "g4n: numeric weights in LDS, no lookups for them; 20 rows in flight, 3 waves/SIMD", 1024, true},
      {10, 0, "g4n: 14 rows in flight, 4 waves/SIMD", 1024, true},
      {11, 0, "g4n: all 27 rows in one batch, 3 waves/SIMD", 1024, true},
      {9, 16 + 64, "abl: g4n without LR and row misses", 1024, false},
      {7, 0, "g4: 10 rows in flight, 3 waves/SIMD register budget", 1024, true},
      {8, 0, "g4: 13 rows in flight, 3 waves/SIMD register budget", 1024, true},
      {3, 0, "g4: 10 rows in flight, 2048 workgroups (8 samples per lane group slot)", 2048, true},
      {3, 16 + 64, "abl: g4 (10 in flight) without LR and row misses", 1024, false},
      {1, 16, "abl: v1 without LR misses", 4096, false},
      {40, 0, "g4: 20 in flight, 2 waves/SIMD, 512 workgroups (every wavefront two passes)", 512, true},
      {41, 0, "g4: 40 in flight, 2 waves/SIMD, 512 workgroups", 512, true},
      {42, 0, "g4: 27 in flight, 2 waves/SIMD, 512 workgroups", 512, true},
      {5, 0, "g4: 20 in flight, 3 waves/SIMD, 768 workgroups", 768, true},
      {5, 0, "g4: 20 in flight, 3 waves/SIMD, 512 workgroups", 512, true},
      {5, 0, "g4: 20 in flight, 3 waves/SIMD, 4096 workgroups", 4096, true},
      {43, 0, "g4: 20 in flight, 2 waves/SIMD, 1024 workgroups", 1024, true},
      {44, 0, "g4: 14 in flight, 4 waves/SIMD, 1024 workgroups of 256 = one pass of 16 waves/CU", 1024, true},
      {30, 0, "g4s: g4 (20 in flight) + scalar prefetch of the 1 M-row tables' rows", 1024, true},
      {31, 0, "g4s: + their LR weights", 1024, true},
      {32, 0, "g4s: rows + LR of the 8 tables of > 90 000 rows", 1024, true},
      {33, 0, "g4s: rows + LR of the 12 tables of > 5 000 rows", 1024, true},
      {34, 0, "g4s: LR only, 12 tables", 1024, true},
      {35, 0, "g4s: rows of 8 tables, LR of 12", 1024, true},
      {36, 0, "g4s (10 in flight, 4 waves/SIMD): rows + LR of 8 tables", 1024, true},
      {1, 16 + 32, "abl: v1 without LR misses and without the five 1 M-row + 2 next tables' row misses", 4096, false},
      {1, 16 + 64, "abl: v1 without LR and row misses (batch stream + S store only)", 4096, false},
      {1, 64, "abl: v1 with LR misses only", 4096, false},
  };
  std::vector<float> h_logit(B), h_s(B * 16);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  int vi = 0;
  for (const Var& v : vars) {
    const int my = vi++;
    if (only >= 0 && my != only) continue;
    MetaHost m = make(v.kernel >= 20 ? 0 : v.flags, v.kernel == 2);
    LPlan lp{};
    if (v.kernel >= 20) {
      lp = make_l(v.flags * 1024);
      printf("     LDS plan: budget %d KB -> %d bytes, rows of %d columns and LR weights of %d columns resident\n", v.flags, lp.lds_bytes, lp.n_emb, lp.n_lr);
      CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&fm_g4l<10, 20, 256, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&fm_g4l<10, 20, 768, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&fm_g4l<10, 10, 512, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&fm_g4l<10, 10, 1024, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    float4* d_image = nullptr;
    const int lds_bytes = static_cast<int>(image.size() * 4);
    if (v.kernel == 2) {
      if (lds_bytes > 160 * 1024) { printf("LDS image too large\n"); return 1; }
      CK(hipMalloc(&d_image, lds_bytes));
      CK(hipMemcpy(d_image, image.data(), lds_bytes, hipMemcpyHostToDevice));
      CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&fm_quad<true, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    }
    for (int z = 0; z < 2; ++z) {
      auto launch = [&](int r) {
        if (v.kernel == 0) fm_v0<<<v.grid, 256>>>(p0[z][r], kF, B, bias, logit, prob, ssum);
        else if (v.kernel == 1) fm_quad<false, 3><<<v.grid, 256>>>(m.dev, arena, kF, X[z][r], kLd, B, bias, logit, prob, ssum, nullptr, 0);
        else if (v.kernel == 3) fm_g4<10, 10, 4><<<v.grid, 256>>>(m.dev, arena, kF, X[z][r], kLd, B, bias, logit, prob, ssum);
        else if (v.kernel == 4) fm_g4<10, 14, 4><<<v.grid, 256>>>(m.dev, arena, kF, X[z][r], kLd, B, bias, logit, prob, ssum);
        else if (v.kernel == 5) fm_g4<10, 20, 3><<<v.grid, 256>>>(m.dev, arena, kF, X[z][r], kLd, B, bias, logit, prob, ssum);
        else if (v.kernel == 7) fm_g4<10, 10, 3><<<v.grid, 256>>>(m.dev, arena, kF, X[z][r], kLd, B, bias, logit, prob, ssum);
        else if (v.kernel == 8) fm_g4<10, 13, 3><<<v.grid, 256>>>(m.dev, arena, kF, X[z][r], kLd, B, bias, logit, prob, ssum);
        else if (v.kernel == 9) fm_g4n<10, 20, 3, 13><<<v.grid, 256>>>(m.dev, arena, kF, X[z][r], kLd, B, bias, logit, prob, ssum);
        else if (v.kernel == 10) fm_g4n<10, 14, 4, 13><<<v.grid, 256>>>(m.dev, arena, kF, X[z][r], kLd, B, bias, logit, prob, ssum);
        else if (v.kernel == 11) fm_g4n<10, 27, 3, 13><<<v.grid, 256>>>(m.dev, arena, kF, X[z][r], kLd, B, bias, logit, prob, ssum);
        else if (v.kernel >= 30 && v.kernel <= 36) {
          auto mk = [&](int thr) { unsigned long long mm = 0; for (int f = 13; f < kF; ++f) if (kCard[f - 13] > thr) mm |= 1ull << f; return mm; };
          const unsigned long long m1 = mk(900000), m8 = mk(90000), m12 = mk(5000);
          unsigned long long pm = 0, lm = 0;
          if (v.kernel == 30) pm = m1;
          if (v.kernel == 31) pm = lm = m1;
          if (v.kernel == 32 || v.kernel == 36) pm = lm = m8;
          if (v.kernel == 33) pm = lm = m12;
          if (v.kernel == 34) lm = m12;
          if (v.kernel == 35) { pm = m8; lm = m12; }
          if (v.kernel == 36) fm_g4s<10, 10, 4><<<v.grid, 256>>>(m.dev, arena, kF, X[z][r], kLd, B, bias, logit, prob, ssum, pm, lm);
          else fm_g4s<10, 20, 3><<<v.grid, 256>>>(m.dev, arena, kF, X[z][r], kLd, B, bias, logit, prob, ssum, pm, lm);
        }
        else if (v.kernel == 40) fm_g4<10, 20, 2><<<v.grid, 256>>>(m.dev, arena, kF, X[z][r], kLd, B, bias, logit, prob, ssum);
        else if (v.kernel == 41) fm_g4<10, 40, 2><<<v.grid, 256>>>(m.dev, arena, kF, X[z][r], kLd, B, bias, logit, prob, ssum);
        else if (v.kernel == 42) fm_g4<10, 27, 2><<<v.grid, 256>>>(m.dev, arena, kF, X[z][r], kLd, B, bias, logit, prob, ssum);
        else if (v.kernel == 43) fm_g4<10, 20, 2><<<v.grid, 256>>>(m.dev, arena, kF, X[z][r], kLd, B, bias, logit, prob, ssum);
        else if (v.kernel == 44) fm_g4<10, 14, 4><<<v.grid, 256>>>(m.dev, arena, kF, X[z][r], kLd, B, bias, logit, prob, ssum);
        else if (v.kernel == 6) fm_g4<10, 8, 5><<<v.grid, 256>>>(m.dev, arena, kF, X[z][r], kLd, B, bias, logit, prob, ssum);
        else if (v.kernel == 20) fm_g4l<10, 20, 256, 3><<<v.grid, 256, lp.lds_bytes>>>(lp.dev, arena, kF, X[z][r], kLd, B, bias, logit, prob, ssum);
        else if (v.kernel == 21) fm_g4l<10, 20, 768, 3><<<v.grid, 768, lp.lds_bytes>>>(lp.dev, arena, kF, X[z][r], kLd, B, bias, logit, prob, ssum);
        else if (v.kernel == 22) fm_g4l<10, 10, 512, 4><<<v.grid, 512, lp.lds_bytes>>>(lp.dev, arena, kF, X[z][r], kLd, B, bias, logit, prob, ssum);
        else if (v.kernel == 23) fm_g4l<10, 10, 1024, 4><<<v.grid, 1024, lp.lds_bytes>>>(lp.dev, arena, kF, X[z][r], kLd, B, bias, logit, prob, ssum);
        else fm_quad<true, 3><<<v.grid, 1024, lds_bytes>>>(m.dev, arena, kF, X[z][r], kLd, B, bias, logit, prob, ssum, d_image, lds_bytes);
      };
      double ez = 0, es = 0;
      if (v.check) {
        CK(hipMemset(logit, 0, B * 4));
        CK(hipMemset(ssum, 0, B * 64));
        launch(1);
        CK(hipGetLastError());
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h_logit.data(), logit, B * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(h_s.data(), ssum, B * 64, hipMemcpyDeviceToHost));
        for (long long b = 0; b < B; b += 37) {
          float zr, sr[16];
          host_ref(B + b, z, &zr, sr);          // batch 1 of the rotation = samples B .. 2B-1
          if (z == 1) {            // host pow/log may round an id differently: compare only where S agrees loosely
            bool same = true;
            for (int d = 0; d < 16; ++d) same = same && fabs(sr[d] - h_s[b * 16 + d]) < 1e-4;
            if (!same) continue;
          }
          ez = fmax(ez, fabs(zr - h_logit[b]));
          for (int d = 0; d < 16; ++d) es = fmax(es, fabs(sr[d] - h_s[b * 16 + d]));
        }
      }
      for (int w = 0; w < 8; ++w) launch(w % kRot);
      CK(hipEventRecord(e0));
      for (int it = 0; it < iters; ++it) launch(it % kRot);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      const double us = ms * 1e3 / iters;
      printf("[%2d] %-84s %-7s %7.2f us  frac %.3f   max|dz| %.2e max|dS| %.2e\n", my, v.name, z ? "zipf" : "uniform", us,
             2032.0 * B / us * 1e-3 / 8000.0, ez, es);
    }
  }
  return 0;
}
