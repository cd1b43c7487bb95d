// rbx_dense.hip -- dense tower contractions on the fp32 matrix cores (gfx950).
//
// Reference behaviour replaced: every nn.Linear of the MLP towers
//   core/pytorch/layers/mlp.py:25-37, ranking/pytorch/layers/blocks/mlp_block.py:42-58,
//   third_party/rechub/basic/layers.py:255-263 (and the 1x1-conv FFN / attention
//   projections of third_party/rechub/models/matching/sasrec.py:81-94,110-124)
// forward  y  = act(x W^T + b)          x[M,K] W[N,K]
// backward dx = dy' W,  dW = dy'^T x,  db = colsum(dy'),  dy' = dy * act'(y)
//
// BASELINE.json asks for fp32 logits within 1e-4, and CDNA4 has no TF32: the kernels use
// v_mfma_f32_32x32x2_f32 (exact fp32 products and accumulation, 256 FLOP/clk/CU) rather
// than bf16 MFMA.  One GEMM kernel serves the three contractions through operand layout
// flags.  Tile 128x128x16 per 256-thread workgroup, 2x2 waves, each wave 2x2 MFMA tiles
// of 32x32 (64 accumulator VGPRs).  Operands are staged k-major in LDS (As[k][m],
// Bs[k][n]) so an MFMA operand read is 32 consecutive floats per half-wave: conflict
// free.  Steady state of the k loop (gemm_steady): the next tile's global loads are issued
// first (inline assembly, no tests), parked in the other LDS buffer halfway through the
// current tile's 32 MFMAs, one barrier per k tile, four wavefronts per SIMD; the last two
// k tiles run the tested loop.  Tiles on the matrix edge take the same loop, skip the
// 32 x 32 blocks that hold no output and share the live ones between their wavefronts; a
// narrow tail of output columns rides in the same launch (narrow_tile).  The
// weight-gradient GEMM has a tiny output and K = batch, so it is split along K across
// workgroups (one flat launch, full tiles first) into a workspace and reduced in a fixed
// order (deterministic, no float atomics).  What was measured on the way:
// profiles/r02/gemm_variants.txt.
#include <stdlib.h>
#include <atomic>
#include <mutex>
#include <type_traits>
#include "rbx_internal.h"

namespace rbx {

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define RBX_GEMM_BK 16
constexpr int BM = 128, BN = 128, BK = RBX_GEMM_BK;
constexpr int kXcds = 8;         // MI355X: 8 accelerator complex dies, 32 CUs and one L2 each
constexpr int LDT = BM + 4;     // LDS row stride (floats): keeps b128 stores aligned, spreads k rows over banks

// Optional tail of the epilogue, applied after bias and activation (all pointers may be NULL):
//   v = mask[row, col] > 0 ? v : 0      ReLU mask taken from ANOTHER tensor (dh = (g W2) o [h > 0]: the activation
//                                       backward of the layer below, without a pass of its own)
//   v += res[row, col]                  residual connection / the second gradient of a tensor with two readers
//   v *= rowscale[row]                  SASRec's timeline mask
// Each of them saves one read-modify-write pass over an [M, N] activation (210 MB at cfg 5).
struct Epi {
  const float* res;
  long long ldres;
  const float* mask;
  long long ldmask;
  const float* rowscale;
  // DeepFM's input block x [M, >= fm_cols] feeds the tower's first GEMM, the FM term and the first-order Linear.  With
  // these set, the dx GEMM of the tower adds the other two readers' gradients to its output columns c < fm_cols:
  //   + fm_g[row] * (fm_s[row, c % fm_dim] - fm_x[row, c])  +  lr_g[row] * lr_w[c]
  // instead of three kernels writing three [M, fm_cols] gradients and a fourth one adding them.
  const float* fm_x;
  long long fm_ldx;
  const float* fm_s;
  const float* fm_g;
  const float* lr_g;
  const float* lr_w;
  int fm_cols;
  int fm_dim;
  int fm_mask;            // fm_dim - 1 when fm_dim is a power of two (col & mask instead of col % dim), else -1
};
__device__ __forceinline__ float epi_fm_term(const Epi& e, int row, int col) {
  if (e.fm_x == nullptr || col >= e.fm_cols) return 0.f;
  const float x = e.fm_x[static_cast<long long>(row) * e.fm_ldx + col];
  const int d = e.fm_mask >= 0 ? (col & e.fm_mask) : (col % e.fm_dim);
  float t = e.fm_g[row] * (e.fm_s[static_cast<long long>(row) * e.fm_dim + d] - x);
  if (e.lr_g != nullptr) t += e.lr_g[row] * e.lr_w[col];
  return t;
}
__device__ __forceinline__ float epi_apply(const Epi& e, float v, int row, int col) {
  if (e.mask != nullptr) v = e.mask[static_cast<long long>(row) * e.ldmask + col] > 0.f ? v : 0.f;
  if (e.res != nullptr) v += e.res[static_cast<long long>(row) * e.ldres + col];
  v += epi_fm_term(e, row, col);
  if (e.rowscale != nullptr) v *= e.rowscale[row];
  return v;
}

// Load one 128 x BK operand tile into registers (BK/2 floats per thread).
//   KCONTIG: element (r, k) at base[r * ld + k]   -> thread reads float4 along k
//   else   : element (r, k) at base[k * ld + r]   -> thread reads float4 along r
constexpr int KT = BK / 4;            // threads along k of a k-contiguous tile
constexpr int NP = BK / 8;            // float4 loads per thread and operand
template <bool KCONTIG>
__device__ __forceinline__ void load_tile(const float* __restrict__ base, long long ld, int r0, int k0, int R, int K,
                                          bool vec_ok, float (&reg)[4 * NP]) {
  const int t = threadIdx.x;
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    if constexpr (KCONTIG) {
      const int r = r0 + t / KT + (256 / KT) * p;
      const int k = k0 + (t % KT) * 4;
      const float* src = base + static_cast<long long>(r) * ld + k;
      if (vec_ok && r < R && k + 3 < K) {
        const float4 v = *reinterpret_cast<const float4*>(src);
        reg[p * 4 + 0] = v.x; reg[p * 4 + 1] = v.y; reg[p * 4 + 2] = v.z; reg[p * 4 + 3] = v.w;
      } else if (r < R && k + 3 < K) {              // unaligned rows (K = 1677): four plain loads, no per-element tests
        reg[p * 4 + 0] = src[0]; reg[p * 4 + 1] = src[1]; reg[p * 4 + 2] = src[2]; reg[p * 4 + 3] = src[3];
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) reg[p * 4 + j] = (r < R && k + j < K) ? src[j] : 0.f;
      }
    } else {
      const int k = k0 + (t >> 5) + 8 * p;
      const int r = r0 + (t & 31) * 4;
      const float* src = base + static_cast<long long>(k) * ld + r;
      if (vec_ok && k < K && r + 3 < R) {
        const float4 v = *reinterpret_cast<const float4*>(src);
        reg[p * 4 + 0] = v.x; reg[p * 4 + 1] = v.y; reg[p * 4 + 2] = v.z; reg[p * 4 + 3] = v.w;
      } else if (k < K && r + 3 < R) {
        reg[p * 4 + 0] = src[0]; reg[p * 4 + 1] = src[1]; reg[p * 4 + 2] = src[2]; reg[p * 4 + 3] = src[3];
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) reg[p * 4 + j] = (k < K && r + j < R) ? src[j] : 0.f;
      }
    }
  }
}

template <bool KCONTIG>
__device__ __forceinline__ void store_tile(float* __restrict__ tile, const float (&reg)[4 * NP]) {
  const int t = threadIdx.x;
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    if constexpr (KCONTIG) {
      const int r = t / KT + (256 / KT) * p;
      const int k = (t % KT) * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) tile[(k + j) * LDT + r] = reg[p * 4 + j];
    } else {
      const int k = (t >> 5) + 8 * p;
      const int r = (t & 31) * 4;
      *reinterpret_cast<float4*>(&tile[k * LDT + r]) = make_float4(reg[p * 4], reg[p * 4 + 1], reg[p * 4 + 2], reg[p * 4 + 3]);
    }
  }
}

// Steady-state loads of a k tile that lies inside [kbeg, kend): per-thread offsets that advance by one k tile per step --
// no tests, no branches, no address arithmetic beyond one 64-bit add, so that the loads, the LDS traffic and the MFMAs of
// one k step are ONE basic block.  Output tiles on the matrix edge run the same loop (workgroups that share a k slice
// through the L2 then keep the same pace; with the edge tiles on the tested loads the dW GEMM of cfg 4 lost 40 %):
// the rows of a k-contiguous operand beyond R are clamped to row R - 1, the columns of the other layout beyond R read on
// into the next row (at most 127 floats; the loop stops two k tiles = 32 rows before kend, so that is allocated memory
// whenever ld >= 8).  What those lanes fetch only reaches C rows / columns that are never stored.
template <bool KCONTIG>
__device__ __forceinline__ void tile_offsets(long long ld, int r0, int k0, int R, long long (&off)[NP]) {
  const int t = threadIdx.x;
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    if constexpr (KCONTIG) {
      int r = r0 + t / KT + (256 / KT) * p;
      r = r < R ? r : R - 1;
      off[p] = static_cast<long long>(r) * ld + k0 + (t % KT) * 4;
    } else {
      off[p] = static_cast<long long>(k0 + (t >> 5) + 8 * p) * ld + r0 + (t & 31) * 4;
    }
  }
}
// The loads are issued as inline assembly: written as C++ the compiler sinks them down to the LDS stores that consume
// them (one basic block, single use), which exposes the whole memory latency; a fence does not hold them.  tile_arrived()
// is the matching wait -- it names the registers as read-write so that no use can be scheduled above it.
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void tile_issue(const float* base, long long (&off)[NP], long long step, f32x4 (&v)[NP]) {
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    // four floats in one request whatever the row pitch: global memory takes dword-aligned dwordx4 loads (K = 1677)
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v[p]) : "v"(base + off[p]));
    off[p] += step;
  }
}
__device__ __forceinline__ void tile_arrived(f32x4 (&a)[NP], f32x4 (&b)[NP]) {
  static_assert(NP == 2 || NP == 4, "operand lists below are written for two / four float4 per thread and operand");
  if constexpr (NP == 2)
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(b[0]), "+v"(b[1]) : : "memory");
  else
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[NP - 2]), "+v"(a[NP - 1]), "+v"(b[0]), "+v"(b[1]),
                 "+v"(b[NP - 2]), "+v"(b[NP - 1]) : : "memory");
}
template <bool KCONTIG>
__device__ __forceinline__ void store_tile_v(float* __restrict__ tile, const f32x4 (&v)[NP]) {
  float reg[4 * NP];
#pragma unroll
  for (int p = 0; p < NP; ++p)
#pragma unroll
    for (int j = 0; j < 4; ++j) reg[p * 4 + j] = v[p][j];
  store_tile<KCONTIG>(tile, reg);
}

#define RBX_GEMM_PIPE 1

// MFMA steps kk in [KLO, KHI) of one staged k tile for a wavefront's 2 x 2 tiles of 32 x 32: only the tiles named in
// LIVE (bit 2 i + j) -- a wavefront whose 32-row / 32-column blocks lie beyond M / N skips
// their products (N = 400 is 12.5 blocks: the weight-gradient GEMM [400, 65536] x [65536, 400] would otherwise run
// 512 x 512 outputs' worth of MFMAs for 400 x 400).
template <int KLO, int KHI, int LIVE>
__device__ __forceinline__ void mfma_steps(const float* __restrict__ as, const float* __restrict__ bs, int wm, int wn, int li,
                                           int lk, f32x16 (&acc)[2][2], int wm1, int wn1) {
  // valid blocks are a prefix in both directions: LIVE is 15 (all four), 5 (one column of two), 3 (one row of two), 1, 0
#pragma unroll
  for (int kk = KLO; kk < KHI; kk += 2) {
    float a0 = 0.f, a1 = 0.f, b0 = 0.f, b1 = 0.f;
    if constexpr ((LIVE & 3) != 0) a0 = as[(kk + lk) * LDT + wm + li];
    if constexpr ((LIVE & 12) != 0) a1 = as[(kk + lk) * LDT + wm1 + li];
    if constexpr ((LIVE & 5) != 0) b0 = bs[(kk + lk) * LDT + wn + li];
    if constexpr ((LIVE & 10) != 0) b1 = bs[(kk + lk) * LDT + wn1 + li];
    if constexpr ((LIVE & 1) != 0) acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
    if constexpr ((LIVE & 2) != 0) acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
    if constexpr ((LIVE & 4) != 0) acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
    if constexpr ((LIVE & 8) != 0) acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
  }
}

// All k tiles but the last two: the tile after the current one is loaded without tests and parked in LDS[cur ^ 1] HALFWAY
// through the current tile's MFMAs (its stores issue in their shadow instead of after them), one barrier per tile.
// Leaves k0 / cur at the first tile the tested loop below has to finish (its operands are staged).
template <bool AK, bool BK_, int LIVE>
__device__ __forceinline__ void gemm_steady(const float* __restrict__ A, long long lda, const float* __restrict__ B,
                                            long long ldb, int m0, int n0, int M, int N, int kend, int& k0, int& cur,
                                            float (&As)[2][BK * LDT], float (&Bs)[2][BK * LDT], int wm, int wn, int li, int lk,
                                            f32x16 (&acc)[2][2]) {
  f32x4 va[NP], vb[NP];
  long long pa[NP], pb[NP];
  tile_offsets<AK>(lda, m0, k0 + BK, M, pa);
  tile_offsets<BK_>(ldb, n0, k0 + BK, N, pb);
  const long long step_a = AK ? BK : BK * lda, step_b = BK_ ? BK : BK * ldb;
  for (; k0 + 3 * BK <= kend; k0 += BK) {
    tile_issue(A, pa, step_a, va);
    tile_issue(B, pb, step_b, vb);
    mfma_steps<0, BK / 2, LIVE>(As[cur], Bs[cur], wm, wn, li, lk, acc, wm + 32, wn + 32);
    __builtin_amdgcn_sched_barrier(0);
    tile_arrived(va, vb);
    store_tile_v<AK>(As[cur ^ 1], va);
    store_tile_v<BK_>(Bs[cur ^ 1], vb);
    mfma_steps<BK / 2, BK, LIVE>(As[cur], Bs[cur], wm, wn, li, lk, acc, wm + 32, wn + 32);
    __syncthreads();
    cur ^= 1;
  }
}

// Narrow companion of gemm_f32_kernel for the last 32 * NT (<= 64) output columns: N = 400 is 3 full 128-column
// tiles plus 16 columns, and a fourth full tile would spend 22% of the MFMA time on padding.  The four wavefronts
// stack along M (32 rows each) and every wavefront computes NT 32x32 MFMA tiles; same operand staging.
template <bool A_KCONTIG, bool B_KCONTIG, int NT>
__device__ __forceinline__ void narrow_tile(const float* __restrict__ A, const long long lda, const float* __restrict__ B,
                                            const long long ldb, float* __restrict__ C, const long long ldc, const int M,
                                            const int N, const int K, const int n0, const float* __restrict__ bias,
                                            const int act, const bool vec_a, const bool vec_b, const Epi& epi, const int m0,
                                            float (&As)[2][BK * LDT], float (&Bs)[2][BK * LDT]) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int wm = wid * 32;
  const int li = lane & 31, lk = lane >> 5;
  const int nlim = (n0 + 32 * NT < N) ? n0 + 32 * NT : N;      // B rows beyond the narrow tile are not fetched
  f32x16 acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  // The epilogue's extra operands are fetched NOW: this kernel runs a handful of k tiles (K = 64 for the SASRec
  // projections), so a load issued after the last MFMA is a full memory round trip that nothing hides (measured: the
  // fused launches took 270 us instead of 126).  They arrive while the operand tiles do.
  f32x16 eres[NT], emask[NT];
  float erow[16];
  const bool has_res = epi.res != nullptr, has_mask = epi.mask != nullptr, has_rs = epi.rowscale != nullptr;
  if (has_res || has_mask) {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int col = n0 + j * 32 + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm + (r & 3) + 8 * (r >> 2) + 4 * lk;
        const bool ok = row < M && col < N;
        eres[j][r] = (has_res && ok) ? epi.res[static_cast<long long>(row) * epi.ldres + col] : 0.f;
        emask[j][r] = (has_mask && ok) ? epi.mask[static_cast<long long>(row) * epi.ldmask + col] : 1.f;
      }
    }
  }
  if (has_rs) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + wm + (r & 3) + 8 * (r >> 2) + 4 * lk;
      erow[r] = row < M ? epi.rowscale[row] : 0.f;
    }
  }
  float ra[4 * NP], rb[4 * NP];
  load_tile<A_KCONTIG>(A, lda, m0, 0, M, K, vec_a, ra);
  load_tile<B_KCONTIG>(B, ldb, n0, 0, nlim, K, vec_b, rb);
  store_tile<A_KCONTIG>(As[0], ra);
  store_tile<B_KCONTIG>(Bs[0], rb);
  __syncthreads();
  int cur = 0;
  for (int k0 = 0; k0 < K; k0 += BK) {
    const bool more = k0 + BK < K;
    if (more) {
      load_tile<A_KCONTIG>(A, lda, m0, k0 + BK, M, K, vec_a, ra);
      load_tile<B_KCONTIG>(B, ldb, n0, k0 + BK, nlim, K, vec_b, rb);
    }
    const float* as = As[cur];
    const float* bs = Bs[cur];
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      const float a0 = as[(kk + lk) * LDT + wm + li];
#pragma unroll
      for (int j = 0; j < NT; ++j)
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bs[(kk + lk) * LDT + j * 32 + li], acc[j], 0, 0, 0);
    }
    if (more) {
      store_tile<A_KCONTIG>(As[cur ^ 1], ra);
      store_tile<B_KCONTIG>(Bs[cur ^ 1], rb);
    }
    __syncthreads();
    cur ^= 1;
  }
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int col = n0 + j * 32 + li;
    if (col >= N) continue;
    const float bv = bias != nullptr ? bias[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + wm + (r & 3) + 8 * (r >> 2) + 4 * lk;
      if (row < M) {
        float v = acc[j][r] + bv;
        if (act == 1) v = v > 0.f ? v : 0.f;
        if (has_mask) v = emask[j][r] > 0.f ? v : 0.f;
        if (has_res) v += eres[j][r];
        v += epi_fm_term(epi, row, col);
        if (has_rs) v *= erow[r];
        C[static_cast<long long>(row) * ldc + col] = v;
      }
    }
  }
}

template <bool A_KCONTIG, bool B_KCONTIG, int NT>
__global__ __launch_bounds__(256) void gemm_f32_narrow_kernel(const float* __restrict__ A, const long long lda,
                                                              const float* __restrict__ B, const long long ldb,
                                                              float* __restrict__ C, const long long ldc, const int M,
                                                              const int N, const int K, const int n0,
                                                              const float* __restrict__ bias, const int act,
                                                              const bool vec_a, const bool vec_b, const Epi epi) {
  __shared__ float As[2][BK * LDT];
  __shared__ float Bs[2][BK * LDT];
  narrow_tile<A_KCONTIG, B_KCONTIG, NT>(A, lda, B, ldb, C, ldc, M, N, K, n0, bias, act, vec_a, vec_b, epi,
                                        static_cast<int>(blockIdx.x) * BM, As, Bs);
}

#define RBX_EPI_CH 4   // outputs whose epilogue operands are fetched in one run of loads

// Epilogue of a 128 x 128 tile whose wavefronts hold 2 x 2 MFMA tiles of 32 x 32 (C/D layout: col = lane & 31,
// row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) -- the same for the f32 and the bf16 MFMAs): shared by the kernels below.
__device__ __forceinline__ void gemm_epilogue(const f32x16 (&acc)[2][2], const int m0, const int n0, const int wm, const int wn,
                                              const int li, const int lk, const int live, const int M, const int N,
                                              float* __restrict__ C, const long long ldc, const float* __restrict__ bias,
                                              const int act, const int splits, const Epi& epi, const int tile_rows = BM) {
  // epilogue: C/D layout of 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
  const bool has_mask = epi.mask != nullptr, has_res = epi.res != nullptr, has_fm = epi.fm_x != nullptr,
             has_lr = epi.lr_g != nullptr, has_rs = epi.rowscale != nullptr;
  if (m0 + tile_rows <= M && n0 + BN <= N && splits == 1 && !(has_fm && (has_mask || has_res || has_rs))) {
    // Interior tile: no row / column tests, and the optional operands of the epilogue are fetched for four outputs at a
    // time in one straight run of loads.  (With a test per output every element was its own basic block -- load, wait,
    // store, 64 times per lane: the DeepFM dx GEMM took 330 us longer than the same GEMM without its epilogue.)
    constexpr int CH = RBX_EPI_CH;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int col = n0 + wn + j * 32 + li;
        const int row0 = m0 + wm + i * 32 + 4 * lk;
        const float bv = bias != nullptr ? bias[col] : 0.f;
#pragma unroll
        for (int h = 0; h < 16; h += CH) {
          float add[CH];
#pragma unroll
          for (int q = 0; q < CH; ++q) add[q] = 0.f;
          if (has_fm) {
            if (col < epi.fm_cols) {
              const int d = epi.fm_mask >= 0 ? (col & epi.fm_mask) : (col % epi.fm_dim);
              const float lw = has_lr ? epi.lr_w[col] : 0.f;
              float x[CH], sm[CH], g[CH], gl[CH];
#pragma unroll
              for (int q = 0; q < CH; ++q) {
                const long long row = row0 + ((h + q) & 3) + 8 * ((h + q) >> 2);
                x[q] = epi.fm_x[row * epi.fm_ldx + col];
                sm[q] = epi.fm_s[row * epi.fm_dim + d];
                g[q] = epi.fm_g[row];
                gl[q] = has_lr ? epi.lr_g[row] : 0.f;
              }
#pragma unroll
              for (int q = 0; q < CH; ++q) add[q] = g[q] * (sm[q] - x[q]) + gl[q] * lw;
            }
#pragma unroll
            for (int q = 0; q < CH; ++q) {
              float v = acc[i][j][h + q] + bv;
              if (act == 1) v = v > 0.f ? v : 0.f;
              C[static_cast<long long>(row0 + ((h + q) & 3) + 8 * ((h + q) >> 2)) * ldc + col] = v + add[q];
            }
          } else {
            float keep[CH], sc[CH];
#pragma unroll
            for (int q = 0; q < CH; ++q) { keep[q] = 1.f; sc[q] = 1.f; }
            if (has_mask) {
#pragma unroll
              for (int q = 0; q < CH; ++q)
                keep[q] = epi.mask[static_cast<long long>(row0 + ((h + q) & 3) + 8 * ((h + q) >> 2)) * epi.ldmask + col];
            }
            if (has_res) {
#pragma unroll
              for (int q = 0; q < CH; ++q)
                add[q] = epi.res[static_cast<long long>(row0 + ((h + q) & 3) + 8 * ((h + q) >> 2)) * epi.ldres + col];
            }
            if (has_rs) {
#pragma unroll
              for (int q = 0; q < CH; ++q) sc[q] = epi.rowscale[row0 + ((h + q) & 3) + 8 * ((h + q) >> 2)];
            }
#pragma unroll
            for (int q = 0; q < CH; ++q) {
              float v = acc[i][j][h + q] + bv;
              if (act == 1) v = v > 0.f ? v : 0.f;
              if (has_mask) v = keep[q] > 0.f ? v : 0.f;
              v += add[q];
              if (has_rs) v *= sc[q];
              C[static_cast<long long>(row0 + ((h + q) & 3) + 8 * ((h + q) >> 2)) * ldc + col] = v;
            }
          }
        }
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn + j * 32 + li;
      if (col >= N || ((live >> (2 * i + j)) & 1) == 0) continue;          // (a tile that is not live may lie over a neighbour's)
      const float bv = (bias != nullptr && splits == 1) ? bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        if (row < M) {
          float v = acc[i][j][r] + bv;
          if (act == 1 && splits == 1) v = v > 0.f ? v : 0.f;
          if (splits == 1) v = epi_apply(epi, v, row, col);
          C[static_cast<long long>(row) * ldc + col] = v;
        }
      }
    }
  }
}

// C[M,N] (+bias, act) = A(M,K) * B(K,N); with splits > 1 a workgroup computes one K slice of its tile
// (then C points at the slice's private [M,N] buffer: C + z * M * N, no epilogue math).
template <bool A_KCONTIG, bool B_KCONTIG>
__global__ __launch_bounds__(256, BK == 16 ? 4 : 2) void gemm_f32_kernel(const float* __restrict__ A, const long long lda,
                                                       const float* __restrict__ B, const long long ldb,
                                                       float* __restrict__ C, const long long ldc, const int M,
                                                       const int N, const int K, const int k_per_split,
                                                       const float* __restrict__ bias, const int act,
                                                       const bool vec_a, const bool vec_b, const int tiles_m,
                                                       const int tiles_n, const int splits, const int narrow_from,
                                                       const int narrow_nt, const Epi epi) {
  __shared__ float As[2][BK * LDT];
  __shared__ float Bs[2][BK * LDT];
  // The first `narrow_from` workgroups compute the narrow tail (the last <= 64 columns behind tiles_n full column tiles) of
  // row block blockIdx.x with the narrow kernel's body instead of a launch of their own: their k loop is bound by memory
  // latency (one 32 x 32 tile per wavefront), so they start first and run BESIDE the full tiles, which keep the MFMA pipes
  // busy meanwhile.  (Launched last they began in the final, half-empty round and outlived it: 766 vs 786 us only.)
  if (static_cast<int>(blockIdx.x) < narrow_from) {
    const int m0n = static_cast<int>(blockIdx.x) * BM;
    if (narrow_nt == 1) narrow_tile<A_KCONTIG, B_KCONTIG, 1>(A, lda, B, ldb, C, ldc, M, N, K, tiles_n * BN, bias, act, vec_a, vec_b, epi, m0n, As, Bs);
    else narrow_tile<A_KCONTIG, B_KCONTIG, 2>(A, lda, B, ldb, C, ldc, M, N, K, tiles_n * BN, bias, act, vec_a, vec_b, epi, m0n, As, Bs);
    return;
  }
  // XCD-aware tile order.  Workgroups are dealt round-robin to the 8 XCDs (each with its own 4 MB L2), so launch
  // index L runs on XCD L % 8.  Tiles are numbered n-fastest and XCD x works through ONE contiguous range of them:
  // the workgroups that share an L2 then share the A row block (all n tiles of an m tile back to back) and walk B in
  // the same order, instead of every XCD fetching every A tile.
  int tm_i, tn_j, z = 0;
  if (splits == 1) {
    const int total = tiles_m * tiles_n, L = static_cast<int>(blockIdx.x) - narrow_from;
    const int xcd = L % kXcds, slot = L / kXcds;
    const int q = total / kXcds, rem = total % kXcds;
    const int tile = xcd * q + (xcd < rem ? xcd : rem) + slot;
    tm_i = tile / tiles_n;
    tn_j = tile % tiles_n;
  } else {
    // K split over `splits` workgroups per tile, one flat launch: the tiles with 128 x 128 real outputs first (every K slice
    // of them), the tiles on the matrix edge after them.  All workgroups are resident at once and the dispatcher deals
    // them out in launch order, so the full tiles spread evenly (the host sizes `splits` for two of them per CU) and the
    // edge tiles -- a fraction of the MFMA work -- land on top as third workgroups instead of displacing full ones.
    const int tm_f = M / BM, tn_f = N / BN, n_full = tm_f * tn_f, n_edge = tiles_m * tiles_n - n_full;
    const int L = static_cast<int>(blockIdx.x) - narrow_from;
    if (L < n_full * splits) {
      z = L / n_full;
      const int f = L % n_full;
      tm_i = f / tn_f;
      tn_j = f % tn_f;
    } else {
      const int e = L - n_full * splits;
      z = e / n_edge;
      const int q = e % n_edge, right = (tiles_n > tn_f) ? tm_f : 0;     // the right-hand column strip, then the bottom row
      if (q < right) { tm_i = q; tn_j = tn_f; }
      else { tm_i = tm_f; tn_j = q - right; }
    }
  }
  const int m0 = tm_i * BM, n0 = tn_j * BN;
  const int kbeg = z * k_per_split;
  const int kend = (kbeg + k_per_split < K) ? kbeg + k_per_split : K;
  if (splits > 1) C += static_cast<long long>(z) * M * ldc;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int li = lane & 31, lk = lane >> 5;
  // The wavefront's corner inside the tile and which of its four 32 x 32 tiles hold any output (bit 2 i + j).  Interior
  // tiles: 2 x 2 wavefronts of 64 x 64.  An edge tile with only one or two 32-row (32-column) blocks of real output deals
  // those blocks out over all four wavefronts instead of leaving them to one or two of them (M = 400: the last row of
  // tiles has 16 rows -- its wavefronts take one 32 x 32 tile each, a quarter of an interior tile's MFMA time, not a half).
  int wm = (wid >> 1) * 64, wn = (wid & 1) * 64, live;
  {
    const int rb = (M - m0 + 31) / 32, cb = (N - n0 + 31) / 32;            // blocks with real rows / columns (>= 1)
    int rows, cols;
    if (rb == 1 && cb > 1) { wm = 0; wn = 32 * wid; rows = 1; cols = wid < cb ? 1 : 0; }
    else if (cb == 1 && rb > 1) { wn = 0; wm = 32 * wid; cols = 1; rows = wid < rb ? 1 : 0; }
    else if (rb == 2 && cb > 2) { wm = 32 * (wid & 1); wn = 64 * (wid >> 1); rows = 1; cols = cb - 2 * (wid >> 1); }
    else if (cb == 2 && rb > 2) { wn = 32 * (wid & 1); wm = 64 * (wid >> 1); cols = 1; rows = rb - 2 * (wid >> 1); }
    else { rows = rb - wm / 32; cols = cb - wn / 32; }
    rows = rows > 2 ? 2 : rows;
    cols = cols > 2 ? 2 : cols;
    live = (rows <= 0 || cols <= 0) ? 0 : (rows == 2 && cols == 2) ? 15 : (rows == 2) ? 5 : (cols == 2) ? 3 : 1;
    live = __builtin_amdgcn_readfirstlane(live);
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float ra[4 * NP], rb[4 * NP];
  load_tile<A_KCONTIG>(A, lda, m0, kbeg, M, kend, vec_a, ra);
  load_tile<B_KCONTIG>(B, ldb, n0, kbeg, N, kend, vec_b, rb);
  store_tile<A_KCONTIG>(As[0], ra);
  store_tile<B_KCONTIG>(Bs[0], rb);
  __syncthreads();
  int cur = 0;
  int k0 = kbeg;
#if RBX_GEMM_PIPE
  if ((A_KCONTIG || lda >= 8) && (B_KCONTIG || ldb >= 8)) {     // (see tile_offsets: how far an edge tile reads on)
#define RBX_STEADY(L) gemm_steady<A_KCONTIG, B_KCONTIG, L>(A, lda, B, ldb, m0, n0, M, N, kend, k0, cur, As, Bs, wm, wn, li, lk, acc)
    if (live == 15) RBX_STEADY(15);
    else if (live == 5) RBX_STEADY(5);
    else if (live == 3) RBX_STEADY(3);
    else if (live == 1) RBX_STEADY(1);
    else RBX_STEADY(0);
#undef RBX_STEADY
  }
#endif
  for (; k0 < kend; k0 += BK) {
    const bool more = k0 + BK < kend;
    if (more) {                                    // next tile's HBM reads fly under this tile's MFMAs
      load_tile<A_KCONTIG>(A, lda, m0, k0 + BK, M, kend, vec_a, ra);
      load_tile<B_KCONTIG>(B, ldb, n0, k0 + BK, N, kend, vec_b, rb);
    }
    // (the last two k tiles run all four products: tiles that are not live read a clamped block and are never stored)
    mfma_steps<0, BK, 15>(As[cur], Bs[cur], wm, wn, li, lk, acc, wm < 96 ? wm + 32 : 96, wn < 96 ? wn + 32 : 96);
    if (more) {
      store_tile<A_KCONTIG>(As[cur ^ 1], ra);
      store_tile<B_KCONTIG>(Bs[cur ^ 1], rb);
    }
    __syncthreads();
    cur ^= 1;
  }
  gemm_epilogue(acc, m0, n0, wm, wn, li, lk, live, M, N, C, ldc, bias, act, splits, epi);
}

// ---- f32 GEMM on the bf16 matrix cores: operands split three ways, six products ---------------------------------------------
// CDNA4 runs v_mfma_f32_32x32x2_f32 at the f32 VECTOR rate (157 TF); its bf16 MFMAs are 16x that (2.5 PF) and accumulate
// in f32.  Every f32 x = h + m + l with bf16 h = rn(x), m = rn(x - h), l = rn(x - h - m) (3 x 8 significant bits:
// |x - h - m - l| <= 2^-24 |x|), so
//     a b = ah bh + (ah bm + am bh) + (ah bl + al bh + am bm) + O(2^-24 |a b|):
// six v_mfma_f32_32x32x16_bf16 per 16 k (192 cycles) instead of eight f32 MFMAs (512 cycles), with an error per product of
// the size of ONE f32 rounding -- the sums carry the same ~sqrt(K) 2^-24 as the f32 kernel's (tests: the same tolerances
// against float64).  bf16 has f32's exponent range: nothing overflows that f32 would not; non-finite inputs come out as
// NaN (inf - inf in the split), f32 denormals lose their low parts.
// Form: y = x W^T and dx = dy W, i.e. A [M, K] row-major activations against weights.  The WEIGHTS are split once per call
// by rbx_split_bf16 into three k-major bf16 planes (transposed for dx), which the caller registers for the duration of the
// GEMM call (rbx_split_register); the activations are split on their way from registers to LDS (v_cvt_pk_bf16_f32, 4.5 VALU
// ops per element beside the MFMAs).  LDS: three bf16 planes per operand, rows k-major in 80-byte pitch (conflict-free
// b128 reads).  The weight-gradient GEMM (both operands batch-major activations): gemm_bxt_kernel further down.
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef float f32x4u_t __attribute__((ext_vector_type(4), aligned(4)));      // a dwordx4 load needs dword alignment only
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
constexpr int SBK = 32;                 // k per staged tile: two MFMA steps of 16
constexpr int SLD = SBK + 8;            // LDS row pitch, bf16 elements
constexpr int SPLANE = BM * SLD;        // one plane of one operand

__device__ __forceinline__ void split2(f32x2_t x, unsigned& h, unsigned& m, unsigned& l) {
  const bf16x2_t hb = __builtin_convertvector(x, bf16x2_t);
  x -= __builtin_convertvector(hb, f32x2_t);
  const bf16x2_t mb = __builtin_convertvector(x, bf16x2_t);
  x -= __builtin_convertvector(mb, f32x2_t);
  const bf16x2_t lb = __builtin_convertvector(x, bf16x2_t);
  h = __builtin_bit_cast(unsigned, hb);
  m = __builtin_bit_cast(unsigned, mb);
  l = __builtin_bit_cast(unsigned, lb);
}
// A tile [128, SBK] of f32 activations, global -> registers: thread t takes k = 4 (t % 8) .. + 3 of rows t / 8 + 32 p.
// k beyond K reads as zero, rows beyond M are clamped (their products only reach outputs that are never stored).
__device__ __forceinline__ void bx6_load_a(const float* __restrict__ A, long long lda, int m0, int k0, int M, int K,
                                           f32x4u_t (&v)[4]) {
  const int t = threadIdx.x;
  const int k = k0 + (t & 7) * 4;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    int r = m0 + (t >> 3) + 32 * p;
    r = r < M ? r : M - 1;
    const float* src = A + static_cast<long long>(r) * lda + k;
    if (k + 3 < K) {
      v[p] = *reinterpret_cast<const f32x4u_t*>(src);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[p][j] = (k + j < K) ? src[j] : 0.f;
    }
  }
}
__device__ __forceinline__ void bx6_store_a(unsigned short* __restrict__ tile, const f32x4u_t (&v)[4]) {
  const int t = threadIdx.x;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    unsigned h0, m0, l0, h1, m1, l1;
    split2(f32x2_t{v[p][0], v[p][1]}, h0, m0, l0);
    split2(f32x2_t{v[p][2], v[p][3]}, h1, m1, l1);
    unsigned short* dst = tile + ((t >> 3) + 32 * p) * SLD + (t & 7) * 4;
    *reinterpret_cast<uint2*>(dst) = make_uint2(h0, h1);
    *reinterpret_cast<uint2*>(dst + SPLANE) = make_uint2(m0, m1);
    *reinterpret_cast<uint2*>(dst + 2 * SPLANE) = make_uint2(l0, l1);
  }
}
// B tile [128 rows (output columns), SBK] of the pre-split weights.  Layout of the planes (rbx_split_bf16): per row, per group
// of 8 k, the three planes' 16 bytes side by side -- [row][kp / 8][3][8] bf16, kp a multiple of SBK (zero-filled) -- so that a
// row's share of a k tile is 192 contiguous bytes (with one [rows][kp] array per plane it was three 64-byte pieces: three
// times the requests of the f32 original, and the kernel ran at 100 TF instead of 167).  Thread t takes the 16-byte chunks
// t + 256 i, i < 6: chunk j = row j / 12, piece j % 12 = 3 (k group) + plane.
__device__ __forceinline__ void bx6_load_b(const unsigned short* __restrict__ Bp, int kp, int n0, int k0, int N, u32x4_t (&v)[6]) {
  const int t = threadIdx.x;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int j = t + 256 * i;
    int r = n0 + j / 12;
    r = r < N ? r : N - 1;
    v[i] = *reinterpret_cast<const u32x4_t*>(Bp + static_cast<long long>(r) * 3 * kp + (k0 >> 3) * 24 + (j % 12) * 8);
  }
}
__device__ __forceinline__ void bx6_store_b(unsigned short* __restrict__ tile, const u32x4_t (&v)[6]) {
  const int t = threadIdx.x;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int j = t + 256 * i;
    const int c = j % 12;
    *reinterpret_cast<u32x4_t*>(tile + (c % 3) * SPLANE + (j / 12) * SLD + (c / 3) * 8) = v[i];
  }
}

template <int LIVE>
__device__ __forceinline__ void bx6_loop(const float* __restrict__ A, const long long lda, const unsigned short* __restrict__ Bp,
                                         const int kp, const int m0, const int n0, const int M, const int N, const int K,
                                         unsigned short* __restrict__ As, unsigned short* __restrict__ Bs, const int wm,
                                         const int wn, const int li, const int lk, f32x16 (&acc)[2][2]) {
  f32x4u_t ra[4];
  u32x4_t rb[6];
  bx6_load_a(A, lda, m0, 0, M, K, ra);
  bx6_load_b(Bp, kp, n0, 0, N, rb);
  const unsigned short* ap = As + (wm + li) * SLD + 8 * lk;
  const unsigned short* bp = Bs + (wn + li) * SLD + 8 * lk;
  for (int k0 = 0; k0 < K; k0 += SBK) {
    bx6_store_a(As, ra);
    bx6_store_b(Bs, rb);
    if (k0 + SBK < K) {                           // the next tile's reads fly under the barrier and this tile's MFMAs
      bx6_load_a(A, lda, m0, k0 + SBK, M, K, ra);
      bx6_load_b(Bp, kp, n0, k0 + SBK, N, rb);
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < SBK / 16; ++ks) {
      bf16x8_t a[2][3], b[2][3];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 3; ++q) {                 // (only the fragments some live tile needs: the others lie outside the tile)
          if ((LIVE >> (2 * i)) & 3) a[i][q] = *reinterpret_cast<const bf16x8_t*>(ap + q * SPLANE + i * 32 * SLD + ks * 16);
          if ((LIVE >> i) & 5) b[i][q] = *reinterpret_cast<const bf16x8_t*>(bp + q * SPLANE + i * 32 * SLD + ks * 16);
        }
      // six products per output tile, the four tiles' chains interleaved (a dependent MFMA waits for its predecessor);
      // terms in ascending size
#define RBX_BX6_TERM(QA, QB)                                                                                        \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)                       \
    if ((LIVE >> (2 * i + j)) & 1) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][QA], b[j][QB], acc[i][j], 0, 0, 0)
      RBX_BX6_TERM(2, 0);
      RBX_BX6_TERM(0, 2);
      RBX_BX6_TERM(1, 1);
      RBX_BX6_TERM(1, 0);
      RBX_BX6_TERM(0, 1);
      RBX_BX6_TERM(0, 0);
#undef RBX_BX6_TERM
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256, 2) void gemm_bx6_kernel(const float* __restrict__ A, const long long lda,
                                                          const unsigned short* __restrict__ Bp, const int kp,
                                                          float* __restrict__ C, const long long ldc, const int M, const int N,
                                                          const int K, const float* __restrict__ bias, const int act,
                                                          const int tiles_m, const int tiles_n, const Epi epi) {
  __shared__ __attribute__((aligned(16))) unsigned short As[3 * SPLANE];
  __shared__ __attribute__((aligned(16))) unsigned short Bs[3 * SPLANE];
  // XCD-aware tile order, as gemm_f32_kernel
  int tm_i, tn_j;
  {
    const int total = tiles_m * tiles_n, L = static_cast<int>(blockIdx.x);
    const int xcd = L % kXcds, slot = L / kXcds;
    const int q = total / kXcds, rem = total % kXcds;
    const int tile = xcd * q + (xcd < rem ? xcd : rem) + slot;
    tm_i = tile / tiles_n;
    tn_j = tile % tiles_n;
  }
  const int m0 = tm_i * BM, n0 = tn_j * BN;
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  const int li = lane & 31, lk = lane >> 5;
  // a wavefront's corner in the tile and its live 32 x 32 output tiles (bit 2 i + j), as gemm_f32_kernel: an edge tile with
  // one or two 32-blocks of real rows (columns) deals them out over all four wavefronts (N = 400: the fourth column tile
  // holds 16 columns)
  int wm = (wid >> 1) * 64, wn = (wid & 1) * 64, live;
  {
    const int rb = (M - m0 + 31) / 32, cb = (N - n0 + 31) / 32;
    int rows, cols;
    if (rb == 1 && cb > 1) { wm = 0; wn = 32 * wid; rows = 1; cols = wid < cb ? 1 : 0; }
    else if (cb == 1 && rb > 1) { wn = 0; wm = 32 * wid; cols = 1; rows = wid < rb ? 1 : 0; }
    else if (rb == 2 && cb > 2) { wm = 32 * (wid & 1); wn = 64 * (wid >> 1); rows = 1; cols = cb - 2 * (wid >> 1); }
    else if (cb == 2 && rb > 2) { wn = 32 * (wid & 1); wm = 64 * (wid >> 1); cols = 1; rows = rb - 2 * (wid >> 1); }
    else { rows = rb - wm / 32; cols = cb - wn / 32; }
    rows = rows > 2 ? 2 : rows;
    cols = cols > 2 ? 2 : cols;
    live = (rows <= 0 || cols <= 0) ? 0 : (rows == 2 && cols == 2) ? 15 : (rows == 2) ? 5 : (cols == 2) ? 3 : 1;
    live = __builtin_amdgcn_readfirstlane(live);
  }
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // (one copy of the k loop per set of live output tiles: a test per MFMA is two scalar instructions beside each of them)
  if (live == 15) bx6_loop<15>(A, lda, Bp, kp, m0, n0, M, N, K, As, Bs, wm, wn, li, lk, acc);
  else if (live == 5) bx6_loop<5>(A, lda, Bp, kp, m0, n0, M, N, K, As, Bs, wm, wn, li, lk, acc);
  else if (live == 3) bx6_loop<3>(A, lda, Bp, kp, m0, n0, M, N, K, As, Bs, wm, wn, li, lk, acc);
  else if (live == 1) bx6_loop<1>(A, lda, Bp, kp, m0, n0, M, N, K, As, Bs, wm, wn, li, lk, acc);
  else bx6_loop<0>(A, lda, Bp, kp, m0, n0, M, N, K, As, Bs, wm, wn, li, lk, acc);
  gemm_epilogue(acc, m0, n0, wm, wn, li, lk, live, M, N, C, ldc, bias, act, 1, epi);
}

// ---- the same GEMM with a 256 x 128 tile, pipelined ----------------------------------------------------------------------------
// gemm_bx6_kernel above runs at 0.34-0.38 of the bf16 pipes whatever its loop looks like (a one-barrier, double-buffered
// form of the same 128 x 128 tile measured 157 vs 160 TF): at six MFMAs per 16 k a 128 x 128 tile asks the L2 for 20 KB
// (8 KB of f32 activations + 12 KB of weight planes) per 768 MFMA cycles -- 16 TB/s over the chip at full rate, more than
// the L2s deliver; it is the plain-bf16 ladder of the guide again (128^2 tiles: 0.36 of peak).  Here a workgroup of EIGHT
// wavefronts owns 256 rows x 128 columns (the weight tile amortised over twice the rows: 28 KB per 2 x the products), k tiles
// of 16 in two LDS buffers, ONE barrier per tile: the tile after the current one is split and parked in the other buffer
// between the two halves of the current tile's MFMAs, the loads of the tile after that issued right behind.
#define RBX_BXP_STAGES 2
#define RBX_BXP_SCHED 1
constexpr int PBK = 16;                 // k per tile: one MFMA step
constexpr int PLD = PBK + 8;            // LDS row pitch, bf16 elements (48 bytes: conflict-free b128 reads of 16 rows)
constexpr int PBM = 256;                // rows of the workgroup's tile
constexpr int PTHREADS = 512;
constexpr int PPLANE_A = PBM * PLD, PPLANE_B = BN * PLD;
constexpr int PBUF_A = 3 * PPLANE_A, PBUF_B = 3 * PPLANE_B;

// A tile [256, 16]: thread t takes k = 4 (t % 4) .. + 3 of rows t / 4 and 128 + t / 4
template <bool GUARD>
__device__ __forceinline__ void bxp_load_a(const float* __restrict__ A, long long lda, int m0, int k0, int M, int K,
                                           f32x4u_t (&v)[2]) {
  const int t = threadIdx.x;
  const int k = k0 + (t & 3) * 4;
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    int r = m0 + (t >> 2) + 128 * p;
    r = r < M ? r : M - 1;
    const float* src = A + static_cast<long long>(r) * lda + k;
    if (!GUARD || k + 3 < K) {
      v[p] = *reinterpret_cast<const f32x4u_t*>(src);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[p][j] = (k + j < K) ? src[j] : 0.f;
    }
  }
}
__device__ __forceinline__ void bxp_store_a(unsigned short* __restrict__ buf, const f32x4u_t (&v)[2]) {
  const int t = threadIdx.x;
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    unsigned h0, m0, l0, h1, m1, l1;
    split2(f32x2_t{v[p][0], v[p][1]}, h0, m0, l0);
    split2(f32x2_t{v[p][2], v[p][3]}, h1, m1, l1);
    unsigned short* dst = buf + ((t >> 2) + 128 * p) * PLD + (t & 3) * 4;
    *reinterpret_cast<uint2*>(dst) = make_uint2(h0, h1);
    *reinterpret_cast<uint2*>(dst + PPLANE_A) = make_uint2(m0, m1);
    *reinterpret_cast<uint2*>(dst + 2 * PPLANE_A) = make_uint2(l0, l1);
  }
}
// B tile [128 rows, 16 k] of the interleaved planes: 96 contiguous bytes per row = 768 chunks of 16 bytes; thread t takes
// chunk t and, the first 256 threads, chunk 512 + t
__device__ __forceinline__ void bxp_load_b(const unsigned short* __restrict__ Bp, int kp, int n0, int k0, int N, u32x4_t (&v)[2]) {
  const int t = threadIdx.x;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int j = t + PTHREADS * i;
    j = j < 768 ? j : t;                   // (the upper half of the second round repeats its first chunk: no branch)
    int r = n0 + j / 6;
    r = r < N ? r : N - 1;
    v[i] = *reinterpret_cast<const u32x4_t*>(Bp + static_cast<long long>(r) * 3 * kp + (k0 >> 3) * 24 + (j % 6) * 8);
  }
}
__device__ __forceinline__ void bxp_store_b(unsigned short* __restrict__ buf, const u32x4_t (&v)[2]) {
  const int t = threadIdx.x;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int j = t + PTHREADS * i;
    j = j < 768 ? j : t;
    const int c = j % 6;
    *reinterpret_cast<u32x4_t*>(buf + (c % 3) * PPLANE_B + (j / 6) * PLD + (c / 3) * 8) = v[i];
  }
}

#define RBX_BXP_TERM(QA, QB)                                                                                        \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)                       \
    if ((LIVE >> (2 * i + j)) & 1) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][QA], b[j][QB], acc[i][j], 0, 0, 0)

template <int LIVE>
__device__ __forceinline__ void bxp_loop(const float* __restrict__ A, const long long lda, const unsigned short* __restrict__ Bp,
                                         const int kp, const int m0, const int n0, const int M, const int N, const int K,
                                         unsigned short* __restrict__ As, unsigned short* __restrict__ Bs, const int wm,
                                         const int wn, const int li, const int lk, f32x16 (&acc)[2][2]) {
  // NS register sets: the loads of a tile are issued NS iterations ahead of its split (one ahead: 46 % of the wavefront
  // cycles parked (PMC), 158 TF at 8192^3; two: 184) -- set (t + 1) % NS holds tile t + 1 when iteration t starts
  constexpr int NS = RBX_BXP_STAGES;
  f32x4u_t ra[NS][2];
  u32x4_t rb[NS][2];
  const int kt = (K + PBK - 1) / PBK;              // tiles; the weight planes are zero-filled up to a multiple of 32
  auto fetch = [&](int tile, auto set_c) {
    constexpr int set = decltype(set_c)::value;
    if ((tile + 1) * PBK <= K) bxp_load_a<false>(A, lda, m0, tile * PBK, M, K, ra[set]);
    else bxp_load_a<true>(A, lda, m0, tile * PBK, M, K, ra[set]);
    bxp_load_b(Bp, kp, n0, tile * PBK, N, rb[set]);
  };
  fetch(0, std::integral_constant<int, 0>{});
  bxp_store_a(As, ra[0]);
  bxp_store_b(Bs, rb[0]);
  if (kt > 1) fetch(1, std::integral_constant<int, 1 % NS>{});
  if (NS > 1 && kt > 2) fetch(2, std::integral_constant<int, 2 % NS>{});
  if (NS > 2 && kt > 3) fetch(3, std::integral_constant<int, 3 % NS>{});
  __syncthreads();
  const int aoff = (wm + li) * PLD + 8 * lk, boff = (wn + li) * PLD + 8 * lk;
  auto step = [&](int t, int cur, auto set_c) {
    constexpr int set = decltype(set_c)::value;    // the set that holds tile t + 1
    const unsigned short* ap = As + cur * PBUF_A + aoff;
    const unsigned short* bp = Bs + cur * PBUF_B + boff;
    bf16x8_t a[2][3], b[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        if ((LIVE >> (2 * i)) & 3) a[i][q] = *reinterpret_cast<const bf16x8_t*>(ap + q * PPLANE_A + i * 32 * PLD);
        if ((LIVE >> i) & 5) b[i][q] = *reinterpret_cast<const bf16x8_t*>(bp + q * PPLANE_B + i * 32 * PLD);
      }
    RBX_BXP_TERM(2, 0);
    RBX_BXP_TERM(0, 2);
    RBX_BXP_TERM(1, 1);
    if (t + 1 < kt) {                              // tile t + 1 -> the other buffer, in the shadow of this tile's MFMAs
      bxp_store_a(As + (cur ^ 1) * PBUF_A, ra[set]);
      bxp_store_b(Bs + (cur ^ 1) * PBUF_B, rb[set]);
    }
    if (t + 1 + NS < kt) fetch(t + 1 + NS, set_c); // the tile NS iterations ahead into the set just emptied
    RBX_BXP_TERM(1, 0);
    RBX_BXP_TERM(0, 1);
    RBX_BXP_TERM(0, 0);
    __syncthreads();
  };
  // Steady state (every tile up to t + 1 + NS lies inside K: no tests): the same step as ONE basic block, with the order
  // the instructions should issue in spelled out -- the twelve LDS reads first, then an MFMA with four of the split's VALU
  // ops / one LDS store / one global load in each of its shadows (RBX_BXP_SCHED=0: the compiler's own order, which puts
  // the whole split behind the MFMAs).
  auto steady = [&](int t, int cur, auto set_c) {
    constexpr int set = decltype(set_c)::value;
    const unsigned short* ap = As + cur * PBUF_A + aoff;
    const unsigned short* bp = Bs + cur * PBUF_B + boff;
    bf16x8_t a[2][3], b[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        a[i][q] = *reinterpret_cast<const bf16x8_t*>(ap + q * PPLANE_A + i * 32 * PLD);
        b[i][q] = *reinterpret_cast<const bf16x8_t*>(bp + q * PPLANE_B + i * 32 * PLD);
      }
    RBX_BXP_TERM(2, 0);
    RBX_BXP_TERM(0, 2);
    RBX_BXP_TERM(1, 1);
    bxp_store_a(As + (cur ^ 1) * PBUF_A, ra[set]);
    bxp_store_b(Bs + (cur ^ 1) * PBUF_B, rb[set]);
    bxp_load_a<false>(A, lda, m0, (t + 1 + NS) * PBK, M, K, ra[set]);
    bxp_load_b(Bp, kp, n0, (t + 1 + NS) * PBK, N, rb[set]);
    RBX_BXP_TERM(1, 0);
    RBX_BXP_TERM(0, 1);
    RBX_BXP_TERM(0, 0);
#if RBX_BXP_SCHED
    __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);                    // DS reads
#pragma unroll
    for (int g = 0; g < 12; ++g) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                   // MFMA
      __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);                   // VALU
    }
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                   // DS write
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                   // VMEM read
    }
#endif
    __syncthreads();
  };
  int t = 0;
  if constexpr (LIVE == 15 && NS == 2) {
    while ((t + 3 + NS) * PBK <= K) {              // two steps per round: tiles t + 1 + NS and t + 2 + NS are read without tests
      steady(t, 0, std::integral_constant<int, 1>{});
      steady(t + 1, 1, std::integral_constant<int, 0>{});
      t += 2;
    }
  }
  // the rest (and edge tiles): the tested step; t is even here, so LDS buffer and register set line up with U = 0
  while (t < kt) {
#define RBX_BXP_STEP(U)                                                            \
    if (t < kt) { step(t, (U) & 1, std::integral_constant<int, ((U) + 1) % NS>{}); ++t; }
    RBX_BXP_STEP(0) RBX_BXP_STEP(1) RBX_BXP_STEP(2) RBX_BXP_STEP(3) RBX_BXP_STEP(4) RBX_BXP_STEP(5)
#undef RBX_BXP_STEP
  }
}
#undef RBX_BXP_TERM

__global__ __launch_bounds__(PTHREADS, 1) void gemm_bxp_kernel(const float* __restrict__ A, const long long lda,
                                                               const unsigned short* __restrict__ Bp, const int kp,
                                                               float* __restrict__ C, const long long ldc, const int M,
                                                               const int N, const int K, const float* __restrict__ bias,
                                                               const int act, const int tiles_m, const int tiles_n,
                                                               const Epi epi) {
  extern __shared__ __attribute__((aligned(16))) unsigned short bxp_lds[];          // 2 x (A 36 KB + B 18 KB) = 108 KB
  unsigned short* As = bxp_lds;
  unsigned short* Bs = bxp_lds + 2 * PBUF_A;
  int tm_i, tn_j;
  {
    const int total = tiles_m * tiles_n, L = static_cast<int>(blockIdx.x);
    const int xcd = L % kXcds, slot = L / kXcds;
    const int q = total / kXcds, rem = total % kXcds;
    const int tile = xcd * q + (xcd < rem ? xcd : rem) + slot;
    tm_i = tile / tiles_n;
    tn_j = tile % tiles_n;
  }
  const int m0 = tm_i * PBM, n0 = tn_j * BN;
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  const int li = lane & 31, lk = lane >> 5;
  // eight wavefronts as 4 x 2, each 64 x 64 = 2 x 2 MFMA tiles; the live ones of an edge tile (bit 2 i + j)
  const int wm = (wid >> 1) * 64, wn = (wid & 1) * 64;
  int live;
  {
    int rows = (M - m0 - wm + 31) / 32, cols = (N - n0 - wn + 31) / 32;
    rows = rows > 2 ? 2 : rows;
    cols = cols > 2 ? 2 : cols;
    live = (rows <= 0 || cols <= 0) ? 0 : (rows == 2 && cols == 2) ? 15 : (rows == 2) ? 5 : (cols == 2) ? 3 : 1;
    live = __builtin_amdgcn_readfirstlane(live);
  }
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  if (live == 15) bxp_loop<15>(A, lda, Bp, kp, m0, n0, M, N, K, As, Bs, wm, wn, li, lk, acc);
  else if (live == 5) bxp_loop<5>(A, lda, Bp, kp, m0, n0, M, N, K, As, Bs, wm, wn, li, lk, acc);
  else if (live == 3) bxp_loop<3>(A, lda, Bp, kp, m0, n0, M, N, K, As, Bs, wm, wn, li, lk, acc);
  else if (live == 1) bxp_loop<1>(A, lda, Bp, kp, m0, n0, M, N, K, As, Bs, wm, wn, li, lk, acc);
  else bxp_loop<0>(A, lda, Bp, kp, m0, n0, M, N, K, As, Bs, wm, wn, li, lk, acc);
  gemm_epilogue(acc, m0, n0, wm, wn, li, lk, live, M, N, C, ldc, bias, act, 1, epi, PBM);
}

// ---- the weight-gradient GEMM dW = dy^T x on the same pipes --------------------------------------------------------------------
// Both operands are batch-major activations: A(i, kk) = dy[kk, i], B(kk, col) = x[kk, col] with the reduction index kk = the
// sample.  Same 256 x 128 x 16 tiles, LDS layout, MFMA phase and prefetch depth as gemm_bxp_kernel; what differs is the
// staging -- a lane reads ONE output row / column (dword loads: 64 consecutive floats of a sample's row per wavefront) for
// pairs of consecutive samples, so that a pair is one packed bf16x2 word of a k-major LDS row (b32 stores) -- both operands
// split in the kernel, and the K (batch) range split over workgroups into a workspace (splitk_reduce_kernel: fixed order).
// A tile [256 i, 16 kk]: thread t takes i = t % 256 and the four sample pairs of kk in [8 (t / 256), + 8)
template <bool GUARD>
__device__ __forceinline__ void bxt_load_a(const float* __restrict__ A, long long lda, int m0, int k0, int M, int kend,
                                           float (&v)[8]) {
  const int t = threadIdx.x;
  int i = m0 + (t & 255);
  i = i < M ? i : M - 1;
  const int kk = k0 + 8 * (t >> 8);
  const float* src = A + static_cast<long long>(kk) * lda + i;
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = (!GUARD || kk + e < kend) ? src[static_cast<long long>(e) * lda] : 0.f;
}
__device__ __forceinline__ void bxt_store_a(unsigned short* __restrict__ buf, const float (&v)[8]) {
  const int t = threadIdx.x;
  unsigned* dst = reinterpret_cast<unsigned*>(buf + (t & 255) * PLD + 8 * (t >> 8));
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    unsigned h, m, l;
    split2(f32x2_t{v[2 * j], v[2 * j + 1]}, h, m, l);
    dst[j] = h;
    dst[j + PPLANE_A / 2] = m;
    dst[j + PPLANE_A] = l;
  }
}
// B tile [128 col, 16 kk]: thread t takes col = t % 128 and the two sample pairs of kk in [4 (t / 128), + 4)
template <bool GUARD>
__device__ __forceinline__ void bxt_load_b(const float* __restrict__ B, long long ldb, int n0, int k0, int N, int kend,
                                           float (&v)[4]) {
  const int t = threadIdx.x;
  int c = n0 + (t & 127);
  c = c < N ? c : N - 1;
  const int kk = k0 + 4 * (t >> 7);
  const float* src = B + static_cast<long long>(kk) * ldb + c;
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = (!GUARD || kk + e < kend) ? src[static_cast<long long>(e) * ldb] : 0.f;
}
__device__ __forceinline__ void bxt_store_b(unsigned short* __restrict__ buf, const float (&v)[4]) {
  const int t = threadIdx.x;
  unsigned* dst = reinterpret_cast<unsigned*>(buf + (t & 127) * PLD + 4 * (t >> 7));
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    unsigned h, m, l;
    split2(f32x2_t{v[2 * j], v[2 * j + 1]}, h, m, l);
    dst[j] = h;
    dst[j + PPLANE_B / 2] = m;
    dst[j + PPLANE_B] = l;
  }
}

#define RBX_BXT_TERM(QA, QB)                                                                                        \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)                       \
    if ((LIVE >> (2 * i + j)) & 1) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][QA], b[j][QB], acc[i][j], 0, 0, 0)

template <int LIVE>
__device__ __forceinline__ void bxt_loop(const float* __restrict__ A, const long long lda, const float* __restrict__ B,
                                         const long long ldb, const int m0, const int n0, const int M, const int N,
                                         const int kbeg, const int kend, unsigned short* __restrict__ As,
                                         unsigned short* __restrict__ Bs, const int wm, const int wn, const int li, const int lk,
                                         f32x16 (&acc)[2][2]) {
  constexpr int NS = 2;
  float ra[NS][8], rb[NS][4];
  const int kt = (kend - kbeg + PBK - 1) / PBK;
  auto fetch = [&](int tile, auto set_c) {
    constexpr int set = decltype(set_c)::value;
    bxt_load_a<true>(A, lda, m0, kbeg + tile * PBK, M, kend, ra[set]);
    bxt_load_b<true>(B, ldb, n0, kbeg + tile * PBK, N, kend, rb[set]);
  };
  fetch(0, std::integral_constant<int, 0>{});
  bxt_store_a(As, ra[0]);
  bxt_store_b(Bs, rb[0]);
  if (kt > 1) fetch(1, std::integral_constant<int, 1>{});
  if (kt > 2) fetch(2, std::integral_constant<int, 0>{});
  __syncthreads();
  const int aoff = (wm + li) * PLD + 8 * lk, boff = (wn + li) * PLD + 8 * lk;
  auto step = [&](int t, int cur, auto set_c) {
    constexpr int set = decltype(set_c)::value;    // the set that holds tile t + 1
    const unsigned short* ap = As + cur * PBUF_A + aoff;
    const unsigned short* bp = Bs + cur * PBUF_B + boff;
    bf16x8_t a[2][3], b[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        if ((LIVE >> (2 * i)) & 3) a[i][q] = *reinterpret_cast<const bf16x8_t*>(ap + q * PPLANE_A + i * 32 * PLD);
        if ((LIVE >> i) & 5) b[i][q] = *reinterpret_cast<const bf16x8_t*>(bp + q * PPLANE_B + i * 32 * PLD);
      }
    RBX_BXT_TERM(2, 0);
    RBX_BXT_TERM(0, 2);
    RBX_BXT_TERM(1, 1);
    if (t + 1 < kt) {
      bxt_store_a(As + (cur ^ 1) * PBUF_A, ra[set]);
      bxt_store_b(Bs + (cur ^ 1) * PBUF_B, rb[set]);
    }
    if (t + 1 + NS < kt) fetch(t + 1 + NS, set_c);
    RBX_BXT_TERM(1, 0);
    RBX_BXT_TERM(0, 1);
    RBX_BXT_TERM(0, 0);
    __syncthreads();
  };
  // steady state, as in bxp_loop: no tests, the issue order spelled out (18 LDS stores and 12 dword loads here)
  auto steady = [&](int t, int cur, auto set_c) {
    constexpr int set = decltype(set_c)::value;
    const unsigned short* ap = As + cur * PBUF_A + aoff;
    const unsigned short* bp = Bs + cur * PBUF_B + boff;
    bf16x8_t a[2][3], b[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        a[i][q] = *reinterpret_cast<const bf16x8_t*>(ap + q * PPLANE_A + i * 32 * PLD);
        b[i][q] = *reinterpret_cast<const bf16x8_t*>(bp + q * PPLANE_B + i * 32 * PLD);
      }
    RBX_BXT_TERM(2, 0);
    RBX_BXT_TERM(0, 2);
    RBX_BXT_TERM(1, 1);
    bxt_store_a(As + (cur ^ 1) * PBUF_A, ra[set]);
    bxt_store_b(Bs + (cur ^ 1) * PBUF_B, rb[set]);
    bxt_load_a<false>(A, lda, m0, kbeg + (t + 1 + NS) * PBK, M, kend, ra[set]);
    bxt_load_b<false>(B, ldb, n0, kbeg + (t + 1 + NS) * PBK, N, kend, rb[set]);
    RBX_BXT_TERM(1, 0);
    RBX_BXT_TERM(0, 1);
    RBX_BXT_TERM(0, 0);
#if RBX_BXP_SCHED
    __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);
#pragma unroll
    for (int g = 0; g < 12; ++g) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
      __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
    }
#pragma unroll
    for (int g = 0; g < 12; ++g) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    }
#endif
    __syncthreads();
  };
  int t = 0;
  if constexpr (LIVE == 15) {
    while (kbeg + (t + 3 + NS) * PBK <= kend) {
      steady(t, 0, std::integral_constant<int, 1>{});
      steady(t + 1, 1, std::integral_constant<int, 0>{});
      t += 2;
    }
  }
  for (; t < kt; t += 2) {
    step(t, 0, std::integral_constant<int, 1>{});
    if (t + 1 < kt) step(t + 1, 1, std::integral_constant<int, 0>{});
  }
}
#undef RBX_BXT_TERM

__global__ __launch_bounds__(PTHREADS, 1) void gemm_bxt_kernel(const float* __restrict__ A, const long long lda,
                                                               const float* __restrict__ B, const long long ldb,
                                                               float* __restrict__ C, const int M, const int N, const int K,
                                                               const int k_per_split, const int tiles_n, const int n_tiles) {
  extern __shared__ __attribute__((aligned(16))) unsigned short bxp_lds[];
  unsigned short* As = bxp_lds;
  unsigned short* Bs = bxp_lds + 2 * PBUF_A;
  const int tile = static_cast<int>(blockIdx.x) % n_tiles, z = static_cast<int>(blockIdx.x) / n_tiles;
  const int m0 = (tile / tiles_n) * PBM, n0 = (tile % tiles_n) * BN;
  const int kbeg = z * k_per_split;
  const int kend = (kbeg + k_per_split < K) ? kbeg + k_per_split : K;
  C += static_cast<long long>(z) * M * N;
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  const int li = lane & 31, lk = lane >> 5;
  const int wm = (wid >> 1) * 64, wn = (wid & 1) * 64;
  int live;
  {
    int rows = (M - m0 - wm + 31) / 32, cols = (N - n0 - wn + 31) / 32;
    rows = rows > 2 ? 2 : rows;
    cols = cols > 2 ? 2 : cols;
    live = (rows <= 0 || cols <= 0) ? 0 : (rows == 2 && cols == 2) ? 15 : (rows == 2) ? 5 : (cols == 2) ? 3 : 1;
    live = __builtin_amdgcn_readfirstlane(live);
  }
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  if (live == 15) bxt_loop<15>(A, lda, B, ldb, m0, n0, M, N, kbeg, kend, As, Bs, wm, wn, li, lk, acc);
  else if (live == 5) bxt_loop<5>(A, lda, B, ldb, m0, n0, M, N, kbeg, kend, As, Bs, wm, wn, li, lk, acc);
  else if (live == 3) bxt_loop<3>(A, lda, B, ldb, m0, n0, M, N, kbeg, kend, As, Bs, wm, wn, li, lk, acc);
  else if (live == 1) bxt_loop<1>(A, lda, B, ldb, m0, n0, M, N, kbeg, kend, As, Bs, wm, wn, li, lk, acc);
  else bxt_loop<0>(A, lda, B, ldb, m0, n0, M, N, kbeg, kend, As, Bs, wm, wn, li, lk, acc);
  // partial [M, N] of this K slice: plain stores (splits = 2 selects the epilogue's no-bias, no-activation path)
  gemm_epilogue(acc, m0, n0, wm, wn, li, lk, live, M, N, C, static_cast<long long>(N), nullptr, 0, 2, Epi{}, PBM);
}

// src [rows, cols] f32 (row pitch ld) -> bf16 planes h, m, l in the layout bx6_load_b reads: out[r][c / 8][q][c % 8],
// c < cp = cols rounded up to a multiple of SBK (zero-filled); transpose: out row r is src COLUMN r.
__global__ __launch_bounds__(256) void split_bf16_kernel(const float* __restrict__ src, const long long ld, const int rows,
                                                         const int cols, const int transpose,
                                                         unsigned short* __restrict__ out) {
  const int orows = transpose ? cols : rows, ocols = transpose ? rows : cols;
  const int cp = (ocols + SBK - 1) / SBK * SBK;
  const long long total = static_cast<long long>(orows) * (cp / 2);          // pairs of output elements
  for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < total;
       i += static_cast<long long>(gridDim.x) * blockDim.x) {
    const int r = static_cast<int>(i / (cp / 2)), c = static_cast<int>(i % (cp / 2)) * 2;
    f32x2_t x = {0.f, 0.f};
    if (transpose) {
      if (c < ocols) x[0] = src[static_cast<long long>(c) * ld + r];
      if (c + 1 < ocols) x[1] = src[static_cast<long long>(c + 1) * ld + r];
    } else {
      if (c < ocols) x[0] = src[static_cast<long long>(r) * ld + c];
      if (c + 1 < ocols) x[1] = src[static_cast<long long>(r) * ld + c + 1];
    }
    unsigned h, m, l;
    split2(x, h, m, l);
    unsigned* dst = reinterpret_cast<unsigned*>(out + static_cast<long long>(r) * 3 * cp + (c >> 3) * 24 + (c & 7));
    dst[0] = h;
    dst[4] = m;
    dst[8] = l;
  }
}

static bool vec_ok(const float* p, long long ld);

// C[i] = sum_z part[z][i] in a fixed order.  A workgroup owns 64 outputs; its 4 wavefronts take every 4th slice
// (4 independent partial sums each, so the loads overlap) and meet in LDS.  The earlier one-thread-per-output
// loop ran 148 us for 512 slices of a [64, 64] weight gradient: 16 workgroups of dependent loads.
constexpr int kRedZ = 16;                  // slices summed side by side per output (wavefronts of the reduce workgroup)
__global__ __launch_bounds__(64 * kRedZ) void splitk_reduce_kernel(const float* __restrict__ part, const long long n,
                                                                    const int splits, float* __restrict__ out,
                                                                    const float* __restrict__ part2, const long long n2,
                                                                    float* __restrict__ out2, const int blocks1) {
  // (part2, n2, out2): a second, smaller reduction over the same number of slices rides in the same launch -- the bias
  // partials beside the weight partials -- in the workgroups from blocks1 on
  __shared__ float red[kRedZ][64];
  const int col = threadIdx.x & 63, zl = threadIdx.x >> 6;
  const bool second = static_cast<int>(blockIdx.x) >= blocks1;
  const float* src = second ? part2 : part;
  float* dst = second ? out2 : out;
  const long long nn = second ? n2 : n;
  const long long b0 = second ? blockIdx.x - blocks1 : blockIdx.x;
  const long long nb = second ? gridDim.x - blocks1 : blocks1;
  for (long long i0 = b0 * 64; i0 < nn; i0 += nb * 64) {
    const long long i = i0 + col;
    float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
    if (i < nn) {
      int z = zl;
      for (; z + 3 * kRedZ < splits; z += 4 * kRedZ) {
        t0 += src[static_cast<long long>(z) * nn + i];
        t1 += src[static_cast<long long>(z + kRedZ) * nn + i];
        t2 += src[static_cast<long long>(z + 2 * kRedZ) * nn + i];
        t3 += src[static_cast<long long>(z + 3 * kRedZ) * nn + i];
      }
      for (; z < splits; z += kRedZ) t0 += src[static_cast<long long>(z) * nn + i];
    }
    red[zl][col] = (t0 + t1) + (t2 + t3);
    __syncthreads();
    if (zl == 0 && i < nn) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < kRedZ; ++w) t += red[w][col];
      dst[i] = t;
    }
    __syncthreads();
  }
}

static void launch_splitk_reduce(hipStream_t s, unsigned blocks, const float* part, long long n, int splits, float* out,
                                 const float* part2 = nullptr, long long n2 = 0, float* out2 = nullptr) {
  const unsigned blocks2 = part2 != nullptr ? static_cast<unsigned>((n2 + 63) / 64) : 0u;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks + blocks2), dim3(64 * kRedZ), 0, s, part, n, splits, out, part2, n2,
                     out2, static_cast<int>(blocks));
}

// dy' = dy * (y > 0)   (ReLU backward, in a scratch buffer so dy stays intact)
__global__ __launch_bounds__(256) void relu_mask_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                        const long long n, float* __restrict__ out) {
  const long long stride = static_cast<long long>(gridDim.x) * blockDim.x;
  for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride)
    out[i] = y[i] > 0.f ? dy[i] : 0.f;
}

// column sums of dy[M,N]: grid (ceil(N/64), row_blocks); partial[rb][n]
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ dy, const int M, const int N,
                                                             const int rows_per_block, float* __restrict__ partial) {
  __shared__ float red[4][64];
  const int n = blockIdx.x * 64 + (threadIdx.x & 63);
  const int r0 = blockIdx.y * rows_per_block;
  const int r1 = (r0 + rows_per_block < M) ? r0 + rows_per_block : M;
  float t = 0.f;
  if (n < N) {
    int r = r0 + (threadIdx.x >> 6);
    for (; r + 28 < r1; r += 32) {                           // 8 rows in flight, added in the same (ascending) order
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = dy[static_cast<long long>(r + 4 * u) * N + n];
#pragma unroll
      for (int u = 0; u < 8; ++u) t += v[u];
    }
    for (; r < r1; r += 4) t += dy[static_cast<long long>(r) * N + n];
  }
  red[threadIdx.x >> 6][threadIdx.x & 63] = t;
  __syncthreads();
  if (threadIdx.x < 64 && n < N)
    partial[static_cast<long long>(blockIdx.y) * N + n] = (red[0][threadIdx.x] + red[1][threadIdx.x]) +
                                                          (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// ---- n == 1: the logit heads (Linear(400, 1) of every tower, rechub LR's Linear(F*D, 1)) -----------------------------
// A [M, K] x [K] product is a streaming read of x; on the 128 x 32 narrow GEMM tile it ran at 2.6 TB/s ([65 536, 1664]:
// 168 us forward, 480 us backward).  Here: a wavefront per row with a fixed xor butterfly (forward), an outer product
// whose lanes keep their columns of w in registers (dx), and g-weighted column sums with fixed-order partials (dW, db).
__global__ __launch_bounds__(256) void gemv_fwd_kernel(const float* __restrict__ x, const long long ldx,
                                                       const float* __restrict__ w, const float* __restrict__ bias,
                                                       const int M, const int K, const int act, const int vec,
                                                       float* __restrict__ y) {
  const int lane = threadIdx.x & 63;
  const int nwaves = gridDim.x * 4;
  const int k4 = vec ? (K & ~3) : 0;
  for (int r = blockIdx.x * 4 + (threadIdx.x >> 6); r < M; r += nwaves) {
    const float* __restrict__ xr = x + static_cast<long long>(r) * ldx;
    float acc = 0.f;
#pragma unroll 4
    for (int c = lane * 4; c < k4; c += 256) {
      const float4 a = *reinterpret_cast<const float4*>(xr + c);
      const float4 b = *reinterpret_cast<const float4*>(w + c);
      acc += (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w);
    }
#pragma unroll 4
    for (int c = k4 + lane; c < K; c += 64) acc += xr[c] * w[c];
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) acc += __shfl_xor(acc, o, 64);
    if (lane == 0) {
      float v = acc + (bias != nullptr ? bias[0] : 0.f);
      if (act == 1 && v < 0.f) v = 0.f;
      y[r] = v;
    }
  }
}

// dx[r, c] = g[r] * w[c].  VEC: K % 4 == 0, 16-byte aligned rows.  When the grid stride is a multiple of the row length
// (outer_grid) a lane keeps its columns; otherwise it recomputes (row, column) per element.
template <bool VEC>
__global__ __launch_bounds__(256) void outer_kernel(const float* __restrict__ g, const float* __restrict__ w, const int M,
                                                    const int K, float* __restrict__ dx, const long long lddx) {
  constexpr int W = VEC ? 4 : 1;
  const unsigned per_row = static_cast<unsigned>(K / W);
  const long long total = static_cast<long long>(M) * per_row;
  const long long step = static_cast<long long>(gridDim.x) * blockDim.x;
  long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= total) return;
  if (step % per_row == 0) {
    long long r = i / per_row;
    const int c = static_cast<int>(i - r * per_row) * W;
    const long long dr = step / per_row;
    float wv[W];
#pragma unroll
    for (int q = 0; q < W; ++q) wv[q] = w[c + q];
#pragma unroll 4
    for (; r < M; r += dr) {
      const float gr = g[r];
      float* dst = dx + r * lddx + c;
      if constexpr (VEC) *reinterpret_cast<float4*>(dst) = make_float4(gr * wv[0], gr * wv[1], gr * wv[2], gr * wv[3]);
      else dst[0] = gr * wv[0];
    }
    return;
  }
  for (; i < total; i += step) {
    const long long r = i / per_row;
    const int c = static_cast<int>(i - r * per_row) * W;
    const float gr = g[r];
    float* dst = dx + r * lddx + c;
    if constexpr (VEC) *reinterpret_cast<float4*>(dst) = make_float4(gr * w[c], gr * w[c + 1], gr * w[c + 2], gr * w[c + 3]);
    else dst[0] = gr * w[c];
  }
}

static unsigned outer_grid(long long total_vecs, long long per_row) {
  long long blocks = (total_vecs + 255) / 256;
  const long long cap = kCUs * 16;
  if (blocks > cap) blocks = cap;
  long long a = per_row, b = 256;
  while (b != 0) { const long long t = a % b; a = b; b = t; }
  const long long unit = per_row / a;
  if (unit <= blocks) blocks = blocks / unit * unit;
  return static_cast<unsigned>(blocks < 1 ? 1 : blocks);
}

// dw_part[rb][k] = sum over the row block of g[r] * x[r, k]; db_part[rb] = sum of g[r]: grid (ceil(K/64), row_blocks)
__global__ __launch_bounds__(256) void wcolsum_partial_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                              const long long ldx, const int M, const int K,
                                                              const int rows_per_block, float* __restrict__ dw_part,
                                                              float* __restrict__ db_part) {
  __shared__ float red[4][64];
  __shared__ float redg[4];
  const int n = blockIdx.x * 64 + (threadIdx.x & 63);
  const int r0 = blockIdx.y * rows_per_block;
  const int r1 = (r0 + rows_per_block < M) ? r0 + rows_per_block : M;
  const bool ok = n < K;
  float t = 0.f, gs = 0.f;
  int r = r0 + (threadIdx.x >> 6);
  for (; r + 28 < r1; r += 32) {                             // 8 rows in flight, added in ascending order
    float v[8], gg[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      gg[u] = g[r + 4 * u];
      v[u] = ok ? x[static_cast<long long>(r + 4 * u) * ldx + n] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      t += gg[u] * v[u];
      gs += gg[u];
    }
  }
  for (; r < r1; r += 4) {
    const float gr = g[r];
    t += gr * (ok ? x[static_cast<long long>(r) * ldx + n] : 0.f);
    gs += gr;
  }
  red[threadIdx.x >> 6][threadIdx.x & 63] = t;
  if ((threadIdx.x & 63) == 0) redg[threadIdx.x >> 6] = gs;
  __syncthreads();
  if (threadIdx.x < 64 && ok)
    dw_part[static_cast<long long>(blockIdx.y) * K + n] = (red[0][threadIdx.x] + red[1][threadIdx.x]) +
                                                          (red[2][threadIdx.x] + red[3][threadIdx.x]);
  if (threadIdx.x == 0 && blockIdx.x == 0 && db_part != nullptr)
    db_part[blockIdx.y] = (redg[0] + redg[1]) + (redg[2] + redg[3]);
}

// dW[n,k] = g^T x (and db = column sums of g) for a TALL, NARROW layer -- n <= 256, k <= 64 with hundreds of thousands of rows
// (SASRec's [B*L, 64] x [64, 64] and fused [64 -> 192] projections): a streaming reduction over the rows, bound by reading g and x once.
// The four wavefronts of a workgroup own the four 32 x 32 quadrants of dW; a lane feeds v_mfma_f32_32x32x2_f32 straight
// from global memory -- A[i][kk] = g[r + kk][c0 + i], B[kk][j] = x[r + kk][d0 + j]: lanes 0..31 read 128 contiguous bytes
// of row r, lanes 32..63 of row r + 1 -- with 8 row pairs in flight, no LDS staging.  Every workgroup leaves a partial
// [n, k] (and [n]) that splitk_reduce_kernel sums in a fixed order.  (The general split-K tile kernel spent 285 us on
// [819200, 64]^T x [819200, 64]: a quarter-filled 128 x 128 tile per workgroup; its column-sum companion 107 us.)
template <int NQ>
__global__ __launch_bounds__(256) void tall_dw_kernel(const float* __restrict__ g, const long long ldg,
                                                      const float* __restrict__ x, const long long ldx, const int M,
                                                      const int n, const int k, const int rows_per_wg,
                                                      float* __restrict__ dw_part, float* __restrict__ db_part) {
  // wavefront w: x columns d0 = (w & 1) * 32 .. + 32 against the g column blocks (w >> 1) + 2 q, q < NQ (n <= 64 NQ)
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int li = lane & 31, lk = lane >> 5;
  const int d0 = (wid & 1) * 32;
  const bool d_ok = d0 + li < k;
  const int r_beg = blockIdx.x * rows_per_wg;
  const int r_end = (r_beg + rows_per_wg < M) ? r_beg + rows_per_wg : M;
  f32x16 acc[NQ];
  float colsum[NQ];
  bool c_ok[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    colsum[q] = 0.f;
    c_ok[q] = ((wid >> 1) + 2 * q) * 32 + li < n;
  }
  constexpr int U = (NQ == 1) ? 8 : 4;     // row pairs in flight (a wavefront feeding both x halves from one g load was slower)
  const float* gp = g + (wid >> 1) * 32 + li;
  const float* xp = x + d0 + li;
  for (int r0 = r_beg; r0 < r_end; r0 += 2 * U) {
    float a[NQ][U], b[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int r = r0 + 2 * u + lk;
      const bool in = r < r_end;
      b[u] = (in && d_ok) ? xp[static_cast<long long>(r) * ldx] : 0.f;
#pragma unroll
      for (int q = 0; q < NQ; ++q) a[q][u] = (in && c_ok[q]) ? gp[static_cast<long long>(r) * ldg + q * 64] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][u], b[u], acc[q], 0, 0, 0);
        colsum[q] += a[q][u];
      }
    }
  }
  // C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
  float* out = dw_part + static_cast<long long>(blockIdx.x) * n * k;
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int c0 = ((wid >> 1) + 2 * q) * 32;
    if (d_ok) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = c0 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        if (row < n) out[row * k + d0 + li] = acc[q][r];
      }
    }
    if (db_part != nullptr && (wid & 1) == 0) {            // the wavefronts of x-quadrant 0 also own the column sums
      const float t = colsum[q] + __shfl_xor(colsum[q], 32, 64);
      if (lk == 0 && c_ok[q]) db_part[static_cast<long long>(blockIdx.x) * n + c0 + li] = t;
    }
  }
}

static bool vec_ok(const float* p, long long ld) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0 && (ld % 4) == 0; }

// ---- K = 64, N = 64 over hundreds of thousands of rows: the weights live in REGISTERS ------------------------------------
// SASRec's projections and FFN convolutions (sasrec.py:81-94,110-124) are [B*L, 64] x [64, 64]: 6.7 GFLOP against 420 MB of
// activations, a streaming pass.  On the staged tile kernel above they ran at 3.3-3.6 TB/s (four k tiles of 16, a barrier
// each, W re-staged per 128 rows).  Here a wavefront owns 32-row slabs and never meets the others.  The k index a lane
// feeds to v_mfma_f32_32x32x2_f32 may be ANY pairing the two operands agree on: lane (m, h) = (lane % 32, lane / 32) holds
// floats [32 h, 32 h + 32) of row m and MFMA step j takes k = 32 h + j against W(k, n), which sits in 2 x 32 registers per
// lane for the whole kernel.  A slab is fetched as eight fully coalesced 1 KB requests (four rows each; per-lane 128-byte
// reads of 64 different lines thrashed the L1: 8x the L2 traffic, slower than the tile kernel), turned into that layout
// through a wavefront-private 8 KB of LDS (no barrier), and the next slab's requests are in flight under the current
// slab's 64 MFMAs.  MFMA-bound rate: 64 x 64 cycles per slab and SIMD = 9.6 TB/s of traffic, above what HBM delivers.
#define RBX_NT_STORE 0   /* measured on SASRec (profiles/r03): plain stores 10.27 ms, streamed stores 10.33 */
#define RBX_NT_EPI 1
constexpr int kSlabWaves = 4;              // wavefronts per workgroup (independent of each other)
constexpr int kSlabLd = 64 + 4;            // LDS row pitch of a slab (floats): b128 reads of 32 rows spread over the banks
// A slab = 32 rows of 64 floats, fetched as eight 1 KB requests: request p covers rows 4 p .. 4 p + 3, lane l takes floats
// [4 (l % 16), + 4) of row 4 p + l / 16.  The per-lane byte offsets below are the same for every full slab (computed once);
// the slab's first row comes in as a wave-uniform base (SGPR pair), so a request costs no vector arithmetic at all -- the
// first version spent ~600 integer instructions per slab on 64-bit row addresses and per-row bounds tests, as long as
// the 64 MFMAs themselves (profiles/r03: 70 % of the wavefront cycles were issue stalls, the MFMA pipes 0.43 busy).
__device__ __forceinline__ void slab_offsets(long long ld, int rows_left, int lane, unsigned (&off)[8]) {
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    int row = 4 * p + (lane >> 4);
    row = row < rows_left ? row : rows_left - 1;         // (the last, partial slab re-reads its last row)
    off[p] = static_cast<unsigned>((static_cast<long long>(row) * ld + 4 * (lane & 15)) * 4);
  }
}
__device__ __forceinline__ void slab_issue(const float* base, const unsigned (&off)[8], f32x4 (&v)[8]) {
#pragma unroll
  for (int p = 0; p < 8; ++p) asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=v"(v[p]) : "v"(off[p]), "s"(base));
}
__device__ __forceinline__ void slab_arrived(f32x4 (&v)[8]) {
  asm volatile("s_waitcnt vmcnt(0)"
               : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7])
               :
               : "memory");
}
// registers of slab_issue -> the lane's half row, through the wavefront's LDS slab
__device__ __forceinline__ void slab_turn(float* __restrict__ lds, int lane, const f32x4 (&v)[8], float (&a)[32]) {
  float* dst = lds + (lane >> 4) * kSlabLd + 4 * (lane & 15);
#pragma unroll
  for (int p = 0; p < 8; ++p) *reinterpret_cast<f32x4*>(dst + 4 * p * kSlabLd) = v[p];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const float* src = lds + (lane & 31) * kSlabLd + 32 * (lane >> 5);
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const f32x4 u = *reinterpret_cast<const f32x4*>(src + 4 * q);
    a[4 * q] = u[0]; a[4 * q + 1] = u[1]; a[4 * q + 2] = u[2]; a[4 * q + 3] = u[3];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();         // (the next slab_turn overwrites what these reads fetch)
}
// element (i, t) of a wavefront's two 32 x 32 output tiles sits in row (i & 3) + 8 (i >> 2) + 4 h, column 32 t + m
__device__ __forceinline__ constexpr int slab_row(int i) { return (i & 3) + 8 * (i >> 2); }

// EPI: 1 residual, 2 mask, 4 row scale, 8 ReLU -- compile-time, so that a launch carries only its own epilogue
template <bool B_KCONTIG, int EPI>
__global__ __launch_bounds__(64 * kSlabWaves, 2) void gemm_f32_k64n64_kernel(
    const float* __restrict__ A, const long long lda, const float* __restrict__ B, const long long ldb, float* __restrict__ C,
    const long long ldc, const int M, const float* __restrict__ bias, const Epi epi) {
  __shared__ float slab[kSlabWaves][32 * kSlabLd];
  const int lane = threadIdx.x & 63, m = lane & 31, h = lane >> 5;
  const int wid = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));      // wave-uniform, and known to be
  const int nw = static_cast<int>(gridDim.x) * kSlabWaves;
  const int slabs = (M + 31) >> 5;
  int s = static_cast<int>(blockIdx.x) * kSlabWaves + wid;
  if (s >= slabs) return;
  float* lds = slab[wid];
  unsigned off_full[8], off[8];
  slab_offsets(lda, 32, lane, off_full);
  f32x4 nx[8];
  {
    const int left = M - s * 32;
#pragma unroll
    for (int p = 0; p < 8; ++p) off[p] = off_full[p];
    if (left < 32) slab_offsets(lda, left, lane, off);
    slab_issue(A + static_cast<long long>(s) * 32 * lda, off, nx);
  }
  // W(k = 32 h + j, n = 32 t + m), t = 0, 1
  float w0[32], w1[32];
  if constexpr (B_KCONTIG) {               // B(k, n) = B[n * ldb + k]: 32 consecutive floats of rows m and 32 + m
    const float4* p0 = reinterpret_cast<const float4*>(B + static_cast<long long>(m) * ldb + 32 * h);
    const float4* p1 = reinterpret_cast<const float4*>(B + static_cast<long long>(32 + m) * ldb + 32 * h);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const float4 u = p0[q], v = p1[q];
      w0[4 * q] = u.x; w0[4 * q + 1] = u.y; w0[4 * q + 2] = u.z; w0[4 * q + 3] = u.w;
      w1[4 * q] = v.x; w1[4 * q + 1] = v.y; w1[4 * q + 2] = v.z; w1[4 * q + 3] = v.w;
    }
  } else {                                 // B(k, n) = B[k * ldb + n]
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      const float* r = B + static_cast<long long>(32 * h + j) * ldb + m;
      w0[j] = r[0];
      w1[j] = r[32];
    }
  }
  const float b0 = bias != nullptr ? bias[m] : 0.f, b1 = bias != nullptr ? bias[32 + m] : 0.f;
  constexpr bool has_res = (EPI & 1) != 0, has_mask = (EPI & 2) != 0, has_rs = (EPI & 4) != 0, relu = (EPI & 8) != 0;
  // per-lane parts of the epilogue's addresses (bytes): row 4 h of the slab, column m
  const long long c_lane = (4LL * h * ldc + m) * 4;
  const long long res_lane = has_res ? (4LL * h * epi.ldres + m) * 4 : 0;
  const long long msk_lane = has_mask ? (4LL * h * epi.ldmask + m) * 4 : 0;
  float a[32];
  slab_arrived(nx);
  slab_turn(lds, lane, nx, a);
  for (;;) {
    const int r0 = s * 32;
    int sn = s + nw;
    const bool more = sn < slabs;
    sn = more ? sn : s;                    // (the last round re-requests its own slab: no branch around the asm)
    {
      const int left = M - sn * 32;
#pragma unroll
      for (int p = 0; p < 8; ++p) off[p] = off_full[p];
      if (left < 32) slab_offsets(lda, left, lane, off);
      slab_issue(A + static_cast<long long>(sn) * 32 * lda, off, nx);
    }
    const int left = M - r0;               // rows of this slab that exist (wave-uniform)
    const bool full = left >= 32;
    // the epilogue's operands are fetched now, under the MFMAs
    f32x16 res0, res1, msk0, msk1;
    float rs[16];
    if constexpr (has_res) {
      const char* base = reinterpret_cast<const char*>(epi.res + static_cast<long long>(r0) * epi.ldres) + res_lane;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int k = (full || slab_row(i) + 4 * h < left) ? slab_row(i) : 0;
        const float* q = reinterpret_cast<const float*>(base + static_cast<long long>(k) * epi.ldres * 4);
#if RBX_NT_EPI
        res0[i] = __builtin_nontemporal_load(q);
        res1[i] = __builtin_nontemporal_load(q + 32);
#else
        res0[i] = q[0];
        res1[i] = q[32];
#endif
      }
    }
    if constexpr (has_mask) {
      const char* base = reinterpret_cast<const char*>(epi.mask + static_cast<long long>(r0) * epi.ldmask) + msk_lane;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int k = (full || slab_row(i) + 4 * h < left) ? slab_row(i) : 0;
        const float* q = reinterpret_cast<const float*>(base + static_cast<long long>(k) * epi.ldmask * 4);
#if RBX_NT_EPI
        msk0[i] = __builtin_nontemporal_load(q);
        msk1[i] = __builtin_nontemporal_load(q + 32);
#else
        msk0[i] = q[0];
        msk1[i] = q[32];
#endif
      }
    }
    if constexpr (has_rs) {
      // the slab's 32 row scales as ONE request (lane l: row l), dealt to the rows a lane finishes through the LDS crossbar:
      // sixteen loads of two distinct words each per slab made the kernel 40 us slower (147 vs 107 us at 819 200 rows) --
      // these kernels are bound by the number of memory requests, not by bytes
      const int rr = r0 + (lane & 31);
      const float mine = epi.rowscale[rr < M ? rr : M - 1];
#pragma unroll
      for (int i = 0; i < 16; ++i) rs[i] = __shfl(mine, slab_row(i) + 4 * h, 64);
    }
    f32x16 acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], w0[j], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], w1[j], acc1, 0, 0, 0);
    }
    char* cbase = reinterpret_cast<char*>(C + static_cast<long long>(r0) * ldc) + c_lane;
    auto finish = [&](auto guarded) {        // two copies of the epilogue: full slabs store without a test per row
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        float v0 = acc0[i] + b0, v1 = acc1[i] + b1;
        if constexpr (relu) { v0 = v0 > 0.f ? v0 : 0.f; v1 = v1 > 0.f ? v1 : 0.f; }
        if constexpr (has_mask) { v0 = msk0[i] > 0.f ? v0 : 0.f; v1 = msk1[i] > 0.f ? v1 : 0.f; }
        if constexpr (has_res) { v0 += res0[i]; v1 += res1[i]; }
        if constexpr (has_rs) { v0 *= rs[i]; v1 *= rs[i]; }
        if (!decltype(guarded)::value || slab_row(i) + 4 * h < left) {
          float* q = reinterpret_cast<float*>(cbase + static_cast<long long>(slab_row(i)) * ldc * 4);
#if RBX_NT_STORE
          __builtin_nontemporal_store(v0, q);
          __builtin_nontemporal_store(v1, q + 32);
#else
          q[0] = v0;
          q[32] = v1;
#endif
        }
      }
    };
    if (full) finish(std::false_type{});
    else finish(std::true_type{});
    slab_arrived(nx);
    if (!more) break;
    slab_turn(lds, lane, nx, a);
    s = sn;
  }
}

// The same slab form for [M, 64] x [64 -> 128] (NT = 4: SASRec's fused K | V projection, sasrec.py:81-87 via
// nn.MultiheadAttention's in_proj) and [M, 128] x [128 -> 64] (KH = 2: its dx): W no longer fits the registers beside the
// slab, so it sits in LDS once per workgroup ([k][n], 33-35 KB) and an MFMA step reads its B operand from there (one b32
// per lane: 32 consecutive floats per half-wave, conflict-free).  A 128-wide row is fetched as two 64-wide halves (each
// request still covers whole 256-byte runs) and turned one after the other through the same 8.5 KB of LDS, their products
// landing in the same accumulators.  Epilogue: bias, ReLU, residual.
template <int KH, int NT, bool B_KCONTIG, bool HAS_RES>
__global__ __launch_bounds__(64 * kSlabWaves, 2) void gemm_f32_slabw_kernel(
    const float* __restrict__ A, const long long lda, const float* __restrict__ B, const long long ldb, float* __restrict__ C,
    const long long ldc, const int M, const float* __restrict__ bias, const int act, const Epi epi) {
  constexpr int K = 64 * KH, N = 32 * NT, WLD = N + 4;
  __shared__ float slab[kSlabWaves][32 * kSlabLd];
  __shared__ float wl[K * WLD];
  for (int e = threadIdx.x; e < K * N; e += 64 * kSlabWaves) {
    int k, n;
    if constexpr (B_KCONTIG) { n = e / K; k = e % K; }       // B(k, n) = B[n * ldb + k]
    else { k = e / N; n = e % N; }                           // B(k, n) = B[k * ldb + n]
    wl[k * WLD + n] = B_KCONTIG ? B[static_cast<long long>(n) * ldb + k] : B[static_cast<long long>(k) * ldb + n];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, m = lane & 31, h = lane >> 5;
  const int wid = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  const int nw = static_cast<int>(gridDim.x) * kSlabWaves;
  const int slabs = (M + 31) >> 5;
  int s = static_cast<int>(blockIdx.x) * kSlabWaves + wid;
  if (s >= slabs) return;
  float* lds = slab[wid];
  unsigned off_full[8], off[8];
  slab_offsets(lda, 32, lane, off_full);
  f32x4 nx[KH][8];
  auto issue = [&](int sl) {
    const int left = M - sl * 32;
#pragma unroll
    for (int p = 0; p < 8; ++p) off[p] = off_full[p];
    if (left < 32) slab_offsets(lda, left, lane, off);
#pragma unroll
    for (int hf = 0; hf < KH; ++hf) slab_issue(A + static_cast<long long>(sl) * 32 * lda + 64 * hf, off, nx[hf]);
  };
  auto arrived = [&]() {
#pragma unroll
    for (int hf = 0; hf < KH; ++hf) slab_arrived(nx[hf]);
  };
  issue(s);
  float bv[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) bv[t] = bias != nullptr ? bias[32 * t + m] : 0.f;
  const long long c_lane = (4LL * h * ldc + m) * 4;
  const long long res_lane = HAS_RES ? (4LL * h * epi.ldres + m) * 4 : 0;
  const float* wp = wl + 32 * h * WLD + m;
  float a[KH][32];
  arrived();
#pragma unroll
  for (int hf = 0; hf < KH; ++hf) slab_turn(lds, lane, nx[hf], a[hf]);
  for (;;) {
    const int r0 = s * 32;
    int sn = s + nw;
    const bool more = sn < slabs;
    sn = more ? sn : s;
    issue(sn);
    const int left = M - r0;
    const bool full = left >= 32;
    f32x16 res[NT];
    if constexpr (HAS_RES) {
      const char* base = reinterpret_cast<const char*>(epi.res + static_cast<long long>(r0) * epi.ldres) + res_lane;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int k = (full || slab_row(i) + 4 * h < left) ? slab_row(i) : 0;
        const float* q = reinterpret_cast<const float*>(base + static_cast<long long>(k) * epi.ldres * 4);
#pragma unroll
        for (int t = 0; t < NT; ++t) res[t][i] = __builtin_nontemporal_load(q + 32 * t);
      }
    }
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
#pragma unroll
    for (int hf = 0; hf < KH; ++hf)
#pragma unroll
      for (int j = 0; j < 32; ++j)
#pragma unroll
        for (int t = 0; t < NT; ++t)
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[hf][j], wp[(64 * hf + j) * WLD + 32 * t], acc[t], 0, 0, 0);
    char* cbase = reinterpret_cast<char*>(C + static_cast<long long>(r0) * ldc) + c_lane;
    auto finish = [&](auto guarded) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (!decltype(guarded)::value || slab_row(i) + 4 * h < left) {
          float* q = reinterpret_cast<float*>(cbase + static_cast<long long>(slab_row(i)) * ldc * 4);
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            float v = acc[t][i] + bv[t];
            if (act == 1) v = v > 0.f ? v : 0.f;
            if constexpr (HAS_RES) v += res[t][i];
            q[32 * t] = v;
          }
        }
      }
    };
    if (full) finish(std::false_type{});
    else finish(std::true_type{});
    arrived();
    if (!more) break;
#pragma unroll
    for (int hf = 0; hf < KH; ++hf) slab_turn(lds, lane, nx[hf], a[hf]);
    s = sn;
  }
}

// dW[64, 64] = g^T x and db = column sums of g over hundreds of thousands of rows, in the slab form of the kernel above:
// a wavefront fetches 32-row slabs of g and x as fully coalesced 1 KB requests (the next slab's are in flight under the
// current one's MFMAs), parks them in its own 2 x 8.5 KB of LDS and feeds v_mfma_f32_32x32x2_f32 from there --
// A[i][kk] = g[r + kk][32 qi + i], B[kk][j] = x[r + kk][32 qj + j], 16 steps x 4 quadrants per slab -- so every byte of g
// and x is requested from memory exactly once (tall_dw_kernel's four quadrant wavefronts each read a half of both: twice
// the L1 traffic, dword requests; 3.5 TB/s).  The four wavefronts' sums meet in LDS in a fixed order; one [64, 64] (+ [64])
// partial per workgroup goes to splitk_reduce_kernel.
// SCALED: row r of g counts row_scale[r] times (dW = (diag(s) g)^T x, db likewise): the `* ~timeline_mask` of a SASRec block
// in the backward, without a pass that writes the scaled gradient (rbx_linear_dwdb_scaled).
template <bool SCALED>
__global__ __launch_bounds__(64 * kSlabWaves, 2) void tall_dw64_kernel(const float* __restrict__ g, const long long ldg,
                                                                      const float* __restrict__ x, const long long ldx,
                                                                      const int M, float* __restrict__ dw_part,
                                                                      float* __restrict__ db_part, const int abl,
                                                                      const float* __restrict__ row_scale) {
  __shared__ float lds[kSlabWaves * 2 * 32 * kSlabLd];
  __shared__ float cs_lds[kSlabWaves][64];
  static_assert(kSlabWaves * 2 * 32 * kSlabLd >= kSlabWaves * 64 * 64, "the slabs' LDS also holds the wavefronts' [64, 64] sums");
  const int lane = threadIdx.x & 63, m = lane & 31, h = lane >> 5;
  const int wid = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  const int nw = static_cast<int>(gridDim.x) * kSlabWaves;
  const int slabs = (M + 31) >> 5;
  float* sg = lds + wid * 2 * 32 * kSlabLd;
  float* sx = sg + 32 * kSlabLd;
  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
  f32x4 cs = {0.f, 0.f, 0.f, 0.f};
  float* pg = sg + (lane >> 4) * kSlabLd + 4 * (lane & 15);
  float* px = sx + (lane >> 4) * kSlabLd + 4 * (lane & 15);
  // coalesced registers -> LDS (rows beyond M carry zeros in g: their products and column sums vanish)
  auto park = [&](int left, const f32x4 (&vg)[8], const f32x4 (&vx)[8], const float (&sc)[8]) {
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      f32x4 u = vg[p];
      if constexpr (SCALED) u *= sc[p];
      if (left < 32 && 4 * p + (lane >> 4) >= left) u = f32x4{0.f, 0.f, 0.f, 0.f};
      cs += u;
      *reinterpret_cast<f32x4*>(pg + 4 * p * kSlabLd) = u;
      *reinterpret_cast<f32x4*>(px + 4 * p * kSlabLd) = vx[p];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  int s = static_cast<int>(blockIdx.x) * kSlabWaves + wid;
  if (s < slabs) {
    unsigned og_full[8], ox_full[8], og[8], ox[8];
    slab_offsets(ldg, 32, lane, og_full);
    slab_offsets(ldx, 32, lane, ox_full);
    auto issue = [&](int sl, f32x4 (&vg)[8], f32x4 (&vx)[8], float (&sc)[8]) {
      const int left = M - sl * 32;
      if constexpr (SCALED) {                 // (plain loads in front of the slab's: they are back long before the wait below)
#pragma unroll
        for (int p = 0; p < 8; ++p) {
          const int row = sl * 32 + 4 * p + (lane >> 4);
          sc[p] = row_scale[row < M ? row : M - 1];
        }
      }
#pragma unroll
      for (int p = 0; p < 8; ++p) { og[p] = og_full[p]; ox[p] = ox_full[p]; }
      if (left < 32) {
        slab_offsets(ldg, left, lane, og);
        slab_offsets(ldx, left, lane, ox);
      }
      slab_issue(g + static_cast<long long>(sl) * 32 * ldg, og, vg);
      slab_issue(x + static_cast<long long>(sl) * 32 * ldx, ox, vx);
    };
    f32x4 ng[8], nx[8];
    float nsc[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
    issue(s, ng, nx, nsc);
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(ng[0]), "+v"(ng[1]), "+v"(ng[2]), "+v"(ng[3]), "+v"(ng[4]), "+v"(ng[5]),
                 "+v"(ng[6]), "+v"(ng[7]) : : "memory");
    slab_arrived(nx);
    park(M - s * 32, ng, nx, nsc);
    const float* rg = sg + h * kSlabLd + m;
    const float* rx = sx + h * kSlabLd + m;
    for (;;) {
      int sn = s + nw;
      const bool more = sn < slabs;
      sn = more ? sn : s;
      issue(sn, ng, nx, nsc);
      if (abl != 1)
#pragma unroll
      for (int jj = 0; jj < 16; ++jj) {
        const float a0 = rg[2 * jj * kSlabLd], a1 = rg[2 * jj * kSlabLd + 32];
        const float b0 = rx[2 * jj * kSlabLd], b1 = rx[2 * jj * kSlabLd + 32];
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(ng[0]), "+v"(ng[1]), "+v"(ng[2]), "+v"(ng[3]), "+v"(ng[4]), "+v"(ng[5]),
                   "+v"(ng[6]), "+v"(ng[7]) : : "memory");
      slab_arrived(nx);
      if (!more) break;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();       // every lane has read the current slab
      s = sn;
      if (abl != 2) park(M - s * 32, ng, nx, nsc);
    }
  }
  __syncthreads();                            // all slabs consumed: the LDS now takes the four [64, 64] sums
  float* mine = lds + wid * 64 * 64;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) mine[(32 * a + slab_row(i) + 4 * h) * 64 + 32 * b + m] = acc[a][b][i];
  // column sums: lanes l, l ^ 16, l ^ 32, l ^ 48 hold the same four columns of different rows
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    float t = cs[c];
    t += __shfl_xor(t, 16, 64);
    t += __shfl_xor(t, 32, 64);
    if (lane < 16) cs_lds[wid][4 * lane + c] = t;
  }
  __syncthreads();
  float* out = dw_part + static_cast<long long>(blockIdx.x) * 64 * 64;
  for (int e = threadIdx.x; e < 64 * 64; e += 64 * kSlabWaves) {
    float t = lds[e];
#pragma unroll
    for (int w = 1; w < kSlabWaves; ++w) t += lds[w * 64 * 64 + e];
    out[e] = t;
  }
  if (db_part != nullptr && threadIdx.x < 64) {
    float t = cs_lds[0][threadIdx.x];
#pragma unroll
    for (int w = 1; w < kSlabWaves; ++w) t += cs_lds[w][threadIdx.x];
    db_part[static_cast<long long>(blockIdx.x) * 64 + threadIdx.x] = t;
  }
}

// RBX_GEMM_BX6=0: every GEMM on v_mfma_f32_32x32x2_f32 (exact f32 products; A/B measurements, debugging); default 1: weights
// with registered bf16 planes run the split-operand kernels on the bf16 matrix cores
static int bx6_mode() {
  static const int v = [] { const char* e = getenv("RBX_GEMM_BX6"); return (e != nullptr && atoi(e) == 0) ? 0 : 1; }();
  return v;
}
// weights whose bf16 planes the caller has made for the GEMM calls it is about to issue (rbx_split_register)
struct SplitEntry {
  const float* w;
  const unsigned short* planes;
  int rows, cols, transposed;
};
constexpr int kSplitSlots = 256;   // registrations live for the duration of ONE call (a tower layer); 256 concurrent ones (threads x towers) before a GEMM falls back to the f32 kernel
static SplitEntry g_split[kSplitSlots];
static std::mutex g_split_mu;
static std::atomic<unsigned long long> g_bx6_launches{0};     // observability: GEMMs that ran on the split-operand kernel
static bool split_find(const float* w, int transposed, int rows, int cols, SplitEntry* out) {
  std::lock_guard<std::mutex> lock(g_split_mu);
  for (int i = 0; i < kSplitSlots; ++i)
    if (g_split[i].w == w && g_split[i].planes != nullptr && g_split[i].transposed == transposed && g_split[i].rows == rows &&
        g_split[i].cols == cols) {
      *out = g_split[i];
      return true;
    }
  return false;
}
// generic driver: C[M,N] = op(A) op(B)
template <bool AK, bool BK_>
static int run_gemm(const float* A, long long lda, const float* B, long long ldb, float* C, int M, int N, int K,
                    const float* bias, int act, float* ws, size_t ws_floats, hipStream_t s, long long ldc = 0,
                    const Epi epi = Epi{}) {
  if (ldc == 0) ldc = N;
  const bool has_epi = epi.res != nullptr || epi.mask != nullptr || epi.rowscale != nullptr || epi.fm_x != nullptr;
  const int tm = (M + BM - 1) / BM, tn = (N + BN - 1) / BN;
  int splits = 1;
  const long long tiles = static_cast<long long>(tm) * tn;
  if (tiles < kCUs && K >= 4096 && ws != nullptr && ldc == N && !has_epi) {   // tiny output, long reduction: split K
    // every workgroup of the launch is resident at once (33.8 KB of LDS each), so the kernel lasts as long as the CU
    // with the most workgroups: pick the split count whose tiles x splits fills whole rounds of the 256 CUs best
    // (k = 1677: 56 tiles x 10 splits = 560 workgroups left a third of the chip idle during the last round; x 9 = 504 fits)
    // Sized on the tiles with 128 x 128 real outputs: the edge tiles are launched after them and cost a fraction
    // (M = 400 is three full rows of tiles and one with 16 rows -- counted as full, the 56 tiles of cfg 4's layer-1 dW
    // got 9 slices (455 k steps each) where the 42 full ones fill the chip with 12 (341 steps)).
    const long long n_full = static_cast<long long>(M / BM) * (N / BN);
    const long long sized = n_full > 0 ? n_full : tiles;
    // (slices of at least 512 reduction rows; a SHORT reduction -- K < 32 768: the weight gradient of a tower at the per-GPU
    //  batch of an 8-GPU strong-scaling run -- may be cut into slices of 128, or its handful of workgroups walk the whole
    //  batch on a few CUs: [128, 256] x 8 192 rows took 54 us in 32 workgroups, profiles/r06/small_batches.txt)
    const int max_splits = K >= 32768 ? K / 512 : K / 128;
    const long long fit = static_cast<long long>(ws_floats / (static_cast<size_t>(M) * N));
    int best = 1;
    double best_eff = 0.0;
    for (int sp = 1; sp <= max_splits && sp <= fit && tiles * sp <= 4 * kCUs; ++sp) {
      const long long wgs = sized * sp;
      const long long rounds = (wgs + kCUs - 1) / kCUs;
      double eff = static_cast<double>(wgs) / static_cast<double>(rounds * kCUs);
      if (rounds < 2) eff *= 0.9;                      // one workgroup per CU hides less latency than two
      if (eff > best_eff + 1e-9) { best_eff = eff; best = sp; }
    }
    splits = best;
  }
  // 64 -> 64 over many rows: the streaming kernel with the weights in registers
  if (AK && K == 64 && N == 64 && splits == 1 && epi.fm_x == nullptr && M >= 2048 && vec_ok(A, lda) &&
      (!BK_ || vec_ok(B, ldb)) && true) {
    const int slabs = (M + 31) / 32;
    int wgs = (slabs + kSlabWaves - 1) / kSlabWaves;
    if (wgs > 2 * kCUs) wgs = 2 * kCUs;              // two workgroups of four wavefronts per CU: two wavefronts per SIMD
    const dim3 grid(wgs), block(64 * kSlabWaves);
    const int code = (epi.res != nullptr ? 1 : 0) | (epi.mask != nullptr ? 2 : 0) | (epi.rowscale != nullptr ? 4 : 0) | (act == 1 ? 8 : 0);
#define RBX_K64(E) case E: hipLaunchKernelGGL((gemm_f32_k64n64_kernel<BK_, E>), grid, block, 0, s, A, lda, B, ldb, C, ldc, M, bias, epi); break
    switch (code) {
      RBX_K64(0); RBX_K64(1); RBX_K64(2); RBX_K64(3); RBX_K64(4); RBX_K64(5); RBX_K64(6); RBX_K64(7);
      RBX_K64(8); RBX_K64(9); RBX_K64(10); RBX_K64(11); RBX_K64(12); RBX_K64(13); RBX_K64(14); RBX_K64(15);
    }
#undef RBX_K64
    return check_launch("gemm_f32_k64n64_kernel");
  }
  // dW = dy^T x of a compute-bound tower layer: the split-operand kernel with transposed staging, K (the batch) split
  if (!AK && !BK_ && bx6_mode() == 1 && ws != nullptr && !has_epi && ldc == N && K >= 8192 && M >= 128 &&
      N >= 128) {
    const int tm2 = (M + PBM - 1) / PBM;
    const int n_tiles = tm2 * tn;
    int sp = kCUs / n_tiles;                           // one workgroup per CU (108 KB of LDS each): one round of the chip
    if (4 * M < 3 * tm2 * PBM) sp = 0;                 // 256-row tiles less than 3/4 full (M = 128): the f32 kernel's 128-row tiles
    const long long fit6 = static_cast<long long>(ws_floats / (static_cast<size_t>(M) * N));
    if (sp > fit6) sp = static_cast<int>(fit6);
    if (sp > K / (K >= 32768 ? 1024 : 256)) sp = K / (K >= 32768 ? 1024 : 256);      // (short reductions: see the f32 kernel's rule above)
    if (sp >= 1) {
      int kps6 = (K + sp - 1) / sp;
      kps6 = (kps6 + PBK - 1) / PBK * PBK;
      sp = (K + kps6 - 1) / kps6;
      static const bool attr_set = [] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bxt_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   2 * (PBUF_A + PBUF_B) * 2) == hipSuccess;
      }();
      (void)attr_set;
      hipLaunchKernelGGL(gemm_bxt_kernel, dim3(n_tiles * sp), dim3(PTHREADS), 2 * (PBUF_A + PBUF_B) * 2, s, A, lda, B, ldb, ws, M,
                         N, K, kps6, tn, n_tiles);
      int rc6 = check_launch("gemm_bxt_kernel");
      if (rc6 != RBX_OK) return rc6;
      g_bx6_launches.fetch_add(1, std::memory_order_relaxed);
      const long long n = static_cast<long long>(M) * N;
      long long blocks = (n + 63) / 64;
      if (blocks > kCUs * 8) blocks = kCUs * 8;
      launch_splitk_reduce(s, static_cast<unsigned>(blocks), ws, n, sp, C);
      return check_launch("splitk_reduce_kernel");
    }
  }
  // 64 -> 128 and 128 -> 64 over many rows: the slab kernel with the weights in LDS
  if (AK && splits == 1 && epi.fm_x == nullptr && epi.mask == nullptr && epi.rowscale == nullptr &&
      M >= 2048 && vec_ok(A, lda) && true && ((K == 64 && N == 128) || (K == 128 && N == 64))) {
    const int slabs = (M + 31) / 32;
    int wgs = (slabs + kSlabWaves - 1) / kSlabWaves;
    if (wgs > 2 * kCUs) wgs = 2 * kCUs;
    const dim3 grid(wgs), block(64 * kSlabWaves);
#define RBX_SLABW(KH_, NT_, R) hipLaunchKernelGGL((gemm_f32_slabw_kernel<KH_, NT_, BK_, R>), grid, block, 0, s, A, lda, B, ldb, C, ldc, M, bias, act, epi)
    if (K == 64) { if (epi.res != nullptr) RBX_SLABW(1, 4, true); else RBX_SLABW(1, 4, false); }
    else { if (epi.res != nullptr) RBX_SLABW(2, 2, true); else RBX_SLABW(2, 2, false); }
#undef RBX_SLABW
    return check_launch("gemm_f32_slabw_kernel");
  }
  // weights with registered bf16 planes: the split-operand kernel on the bf16 matrix cores
  if (AK && splits == 1 && bx6_mode() > 0) {
    SplitEntry e;
    if (split_find(B, BK_ ? 0 : 1, BK_ ? N : K, BK_ ? K : N, &e)) {
      const int kp = (K + SBK - 1) / SBK * SBK;
      if (M < 2 * PBM) {                               // few rows: the 128 x 128 form
        hipLaunchKernelGGL(gemm_bx6_kernel, dim3(tn * tm), dim3(256), 0, s, A, lda, e.planes, kp, C, ldc, M, N, K, bias, act, tm,
                           tn, epi);
      } else {
        static const bool attr_set = [] {
          return hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bxp_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     2 * (PBUF_A + PBUF_B) * 2) == hipSuccess;
        }();
        (void)attr_set;
        const int tm2 = (M + PBM - 1) / PBM;
        hipLaunchKernelGGL(gemm_bxp_kernel, dim3(tn * tm2), dim3(PTHREADS), 2 * (PBUF_A + PBUF_B) * 2, s, A, lda, e.planes, kp,
                           C, ldc, M, N, K, bias, act, tm2, tn, epi);
      }
      g_bx6_launches.fetch_add(1, std::memory_order_relaxed);
      return check_launch("gemm_bx6_kernel");
    }
  }
  int kps = (K + splits - 1) / splits;
  kps = (kps + BK - 1) / BK * BK;
  splits = (K + kps - 1) / kps;
  float* dst = (splits > 1) ? ws : C;
  // the last partial column tile: when it is at most 64 columns wide (and K is not split) it goes to the narrow kernel
  const int tail = N % BN;
#define RBX_GEMM_NARROW_TAIL 1
  // RBX_GEMM_NARROW_TAIL=1: a tail of at most 64 columns behind full column tiles goes to the narrow kernel (a second
  // launch that re-reads A); 0: the main kernel's edge-tile path takes it in the same launch (A comes from the L2).
  // Measured at cfg 4 (N = 400 = 3 x 128 + 16, profiles/r02/gemm_variants.txt): layer-1 forward 820 us with the narrow
  // launch, 856 in one launch (a fourth workgroup slot per row block for 4 % of the columns); 400 x 400: 214 vs 230.
  const int tn_full = (splits == 1 && tail > 0 && tail <= 64 && (RBX_GEMM_NARROW_TAIL || N < BN)) ? N / BN : tn;
#define RBX_GEMM_NARROW_INSIDE 1
  const bool inside = RBX_GEMM_NARROW_INSIDE && tn_full > 0 && tn_full < tn;       // the narrow tail rides in the main launch
  if (tn_full > 0)
    hipLaunchKernelGGL((gemm_f32_kernel<AK, BK_>), dim3(tn_full * tm * splits + (inside ? tm : 0)), dim3(256),
                       0, s, A, lda, B, ldb, dst,
                       (splits > 1) ? static_cast<long long>(N) : ldc, M, N, K, kps, bias, act, vec_ok(A, lda),
                       vec_ok(B, ldb), tm, tn_full, splits, inside ? tm : 0, tail <= 32 ? 1 : 2, epi);
  if (tn_full < tn && !inside) {
    const int n0 = tn_full * BN;
    if (tail <= 32)
      hipLaunchKernelGGL((gemm_f32_narrow_kernel<AK, BK_, 1>), dim3(tm), dim3(256), 0, s, A, lda, B, ldb, C, ldc, M, N, K, n0,
                         bias, act, vec_ok(A, lda), vec_ok(B, ldb), epi);
    else
      hipLaunchKernelGGL((gemm_f32_narrow_kernel<AK, BK_, 2>), dim3(tm), dim3(256), 0, s, A, lda, B, ldb, C, ldc, M, N, K, n0,
                         bias, act, vec_ok(A, lda), vec_ok(B, ldb), epi);
  }
  int rc = check_launch("gemm_f32_kernel");
  if (rc != RBX_OK) return rc;
  if (splits > 1) {
    const long long n = static_cast<long long>(M) * N;
    long long blocks = (n + 63) / 64;
    if (blocks > kCUs * 8) blocks = kCUs * 8;
    launch_splitk_reduce(s, static_cast<unsigned>(blocks), ws, n, splits, C);
    rc = check_launch("splitk_reduce_kernel");
  }
  return rc;
}

}  // namespace rbx

extern "C" int rbx_linear_fwd(const float* d_x, int64_t x_stride, const float* d_w, const float* d_bias, int64_t m,
                              int32_t n, int32_t k, int32_t act, float* d_y, void* stream) {
  if (m == 0) return RBX_OK;   // empty batch: nothing to do, pointers may be NULL
  using namespace rbx;
  if (d_x == nullptr || d_w == nullptr || d_y == nullptr) return fail(RBX_ERR_INVALID, "linear: NULL tensor");
  if (m < 0 || n <= 0 || k <= 0 || m > INT_MAX) return fail(RBX_ERR_INVALID, "linear: bad shape");
  if (x_stride < k) return fail(RBX_ERR_INVALID, "linear: x_stride %lld < k %d", static_cast<long long>(x_stride), k);
  if (act != 0 && act != 1) return fail(RBX_ERR_UNSUPPORTED, "linear: activation code %d", act);
  if (m == 0) return RBX_OK;
  if (n == 1) {                                             // a logit head: one streaming pass, a wavefront per row
    long long blocks = (m + 3) / 4;
    if (blocks > kCUs * 16) blocks = kCUs * 16;
    const int vec = (x_stride % 4 == 0) && ((reinterpret_cast<uintptr_t>(d_x) | reinterpret_cast<uintptr_t>(d_w)) & 15) == 0;
    hipLaunchKernelGGL(gemv_fwd_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, as_stream(stream), d_x,
                       static_cast<long long>(x_stride), d_w, d_bias, static_cast<int>(m), k, act, vec, d_y);
    return check_launch("gemv_fwd_kernel");
  }
  // y[m,n] = x[m,k] * W[n,k]^T : A = x (k contiguous), B(k,n) = W[n*k + k] (k contiguous)
  return run_gemm<true, true>(d_x, x_stride, d_w, k, d_y, static_cast<int>(m), n, k, d_bias, act, nullptr, 0,
                              as_stream(stream));
}

extern "C" int rbx_linear_fwd_fused(const float* d_x, int64_t x_stride, const float* d_w, const float* d_bias, int64_t m,
                                    int32_t n, int32_t k, int32_t act, const float* d_residual, int64_t residual_stride,
                                    const float* d_row_scale, float* d_y, int64_t y_stride, void* stream) {
  if (m == 0) return RBX_OK;
  using namespace rbx;
  if (d_x == nullptr || d_w == nullptr || d_y == nullptr) return fail(RBX_ERR_INVALID, "linear_fused: NULL tensor");
  if (m < 0 || n <= 1 || k <= 0 || m > INT_MAX) return fail(RBX_ERR_INVALID, "linear_fused: bad shape (n must be > 1)");
  if (x_stride < k || y_stride < n || (d_residual != nullptr && residual_stride < n))
    return fail(RBX_ERR_INVALID, "linear_fused: a row stride is shorter than its row");
  if (act != 0 && act != 1) return fail(RBX_ERR_UNSUPPORTED, "linear_fused: activation code %d", act);
  Epi epi{};
  epi.res = d_residual;
  epi.ldres = static_cast<long long>(residual_stride);
  epi.rowscale = d_row_scale;
  return run_gemm<true, true>(d_x, x_stride, d_w, k, d_y, static_cast<int>(m), n, k, d_bias, act, nullptr, 0,
                              as_stream(stream), y_stride, epi);
}

extern "C" int rbx_linear_dx_fused(const float* d_dy, int64_t dy_stride, const float* d_w, int64_t m, int32_t n, int32_t k,
                                   const float* d_mask, int64_t mask_stride, const float* d_residual,
                                   int64_t residual_stride, float* d_dx, int64_t dx_stride, void* stream) {
  if (m == 0) return RBX_OK;
  using namespace rbx;
  if (d_dy == nullptr || d_w == nullptr || d_dx == nullptr) return fail(RBX_ERR_INVALID, "linear_dx_fused: NULL tensor");
  if (m < 0 || n <= 0 || k <= 1 || m > INT_MAX) return fail(RBX_ERR_INVALID, "linear_dx_fused: bad shape (k must be > 1)");
  if (dy_stride < n || dx_stride < k || (d_mask != nullptr && mask_stride < k) ||
      (d_residual != nullptr && residual_stride < k))
    return fail(RBX_ERR_INVALID, "linear_dx_fused: a row stride is shorter than its row");
  Epi epi{};
  epi.res = d_residual;
  epi.ldres = static_cast<long long>(residual_stride);
  epi.mask = d_mask;
  epi.ldmask = static_cast<long long>(mask_stride);
  // dx[m,k] = dy[m,n] * W[n,k]: A = dy (n contiguous = its K), B(kk=n, col=k) = W[n*k + k] (col contiguous)
  return run_gemm<true, false>(d_dy, dy_stride, d_w, k, d_dx, static_cast<int>(m), k, n, nullptr, 0, nullptr, 0,
                               as_stream(stream), dx_stride, epi);
}

extern "C" int rbx_linear_dx_deepfm(const float* d_dy, int64_t dy_stride, const float* d_w, int64_t m, int32_t n, int32_t k,
                                    const float* d_x, int64_t x_stride, const float* d_fm_sum, int32_t fm_dim,
                                    int32_t fm_cols, const float* d_fm_g, const float* d_lr_g, const float* d_lr_w,
                                    float* d_dx, int64_t dx_stride, void* stream) {
  if (m == 0) return RBX_OK;
  using namespace rbx;
  if (!d_dy || !d_w || !d_dx || !d_x || !d_fm_sum || !d_fm_g) return fail(RBX_ERR_INVALID, "linear_dx_deepfm: NULL tensor");
  if ((d_lr_g == nullptr) != (d_lr_w == nullptr)) return fail(RBX_ERR_INVALID, "linear_dx_deepfm: lr_g and lr_w come together");
  if (m < 0 || n <= 0 || k <= 1 || m > INT_MAX || fm_dim <= 0 || fm_cols <= 0 || fm_cols > k || fm_cols % fm_dim != 0)
    return fail(RBX_ERR_INVALID, "linear_dx_deepfm: bad shape (fm_cols %d of k %d, fm_dim %d)", fm_cols, k, fm_dim);
  if (dy_stride < n || dx_stride < k || x_stride < fm_cols) return fail(RBX_ERR_INVALID, "linear_dx_deepfm: row stride too short");
  Epi epi{};
  epi.fm_x = d_x;
  epi.fm_ldx = static_cast<long long>(x_stride);
  epi.fm_s = d_fm_sum;
  epi.fm_g = d_fm_g;
  epi.lr_g = d_lr_g;
  epi.lr_w = d_lr_w;
  epi.fm_cols = fm_cols;
  epi.fm_dim = fm_dim;
  epi.fm_mask = (fm_dim & (fm_dim - 1)) == 0 ? fm_dim - 1 : -1;
  return run_gemm<true, false>(d_dy, dy_stride, d_w, k, d_dx, static_cast<int>(m), k, n, nullptr, 0, nullptr, 0,
                               as_stream(stream), dx_stride, epi);
}

// split-K scratch of the weight gradient: room for 2 x CUs slices of [n, k], at most 64 MiB
// workgroups of tall_dw_kernel at most (each leaves an [n, k] partial): 4 per CU -- measured on
// SASRec's [819 200, 64] x [819 200, 64] weight gradients: 158 us with 512 workgroups, 118 with 1024, 124 with 2048 (and the
// reduce over the partials grows with them).
static int tall_wgs_max() { return 4 * rbx::kCUs; }

static size_t dw_ws_floats(int32_t n, int32_t k) {
  const size_t slices = static_cast<size_t>(tall_wgs_max() > 2 * rbx::kCUs ? tall_wgs_max() : 2 * rbx::kCUs);
  const size_t want = static_cast<size_t>(n) * k * slices;
  const size_t cap = size_t(1) << 24;
  const size_t one = static_cast<size_t>(n) * k;
  return want < cap ? want : (cap > one ? cap : one);
}

extern "C" size_t rbx_linear_bwd_workspace_size(int64_t m, int32_t n, int32_t k, int32_t act) {
  // relu-masked dy copy + split-K slices of dW (at most 2*CUs tiles worth) + bias partials
  const size_t masked = (act == 1) ? static_cast<size_t>(m) * n : 0;
  const size_t splits = 2 * rbx::kCUs;
  const size_t dw = dw_ws_floats(n, k);
  // bias partials: one row of n floats per 1024-row block -- or per workgroup of the tall-and-narrow kernel, whichever is more
  const size_t db_rows = static_cast<size_t>((m + 1023) / 1024) > static_cast<size_t>(tall_wgs_max())
                             ? static_cast<size_t>((m + 1023) / 1024) : static_cast<size_t>(tall_wgs_max());
  const size_t db = db_rows * n;
  (void)splits;
  return (masked + dw + db + 1024) * sizeof(float);
}

extern "C" int rbx_linear_bwd(const float* d_x, int64_t x_stride, const float* d_w, const float* d_y, const float* d_dy,
                              int64_t m, int32_t n, int32_t k, int32_t act, float* d_dx, int64_t dx_stride, float* d_dw,
                              float* d_db, void* d_workspace, size_t workspace_bytes, void* stream) {
  if (m == 0) return RBX_OK;   // empty batch: nothing to do, pointers may be NULL
  using namespace rbx;
  if (d_x == nullptr || d_w == nullptr || d_dy == nullptr) return fail(RBX_ERR_INVALID, "linear_bwd: NULL tensor");
  if (x_stride < k || (d_dx != nullptr && dx_stride < k)) return fail(RBX_ERR_INVALID, "linear_bwd: row stride < k");
  if (act == 1 && d_y == nullptr) return fail(RBX_ERR_INVALID, "linear_bwd: y is needed for the ReLU mask");
  if (m <= 0 || m > INT_MAX) return (m == 0) ? RBX_OK : fail(RBX_ERR_INVALID, "linear_bwd: bad m");
  const size_t need = rbx_linear_bwd_workspace_size(m, n, k, act);
  if (d_workspace == nullptr || workspace_bytes < need) return fail(RBX_ERR_WORKSPACE, "linear_bwd: workspace too small");
  hipStream_t s = as_stream(stream);
  float* ws = static_cast<float*>(d_workspace);
  const int M = static_cast<int>(m);
  const float* g = d_dy;
  if (act == 1) {
    const long long cnt = static_cast<long long>(m) * n;
    long long blocks = (cnt + 255) / 256;
    if (blocks > kCUs * 8) blocks = kCUs * 8;
    hipLaunchKernelGGL(relu_mask_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s, d_dy, d_y, cnt, ws);
    g = ws;
    ws += cnt;
  }
  const size_t dw_floats = dw_ws_floats(n, k);
  int rc = RBX_OK;
  if (n == 1) {
    // logit head: dx = g (x) w as a streaming store, dW / db as g-weighted column sums of x (fixed-order partials)
    if (d_dx != nullptr) {
      const bool vec = (k % 4 == 0) && (dx_stride % 4 == 0) &&
                       ((reinterpret_cast<uintptr_t>(d_dx) | reinterpret_cast<uintptr_t>(d_w)) & 15) == 0;
      const long long per_row = vec ? k / 4 : k;
      const unsigned blocks = outer_grid(static_cast<long long>(m) * per_row, per_row);
      if (vec)
        hipLaunchKernelGGL(outer_kernel<true>, dim3(blocks), dim3(256), 0, s, g, d_w, M, k, d_dx, static_cast<long long>(dx_stride));
      else
        hipLaunchKernelGGL(outer_kernel<false>, dim3(blocks), dim3(256), 0, s, g, d_w, M, k, d_dx, static_cast<long long>(dx_stride));
    }
    if (d_dw != nullptr || d_db != nullptr) {
      // row blocks: 1024 rows (the db partial area is sized for that), more when the dW partials would not fit
      long long rpb = 1024;
      while (((m + rpb - 1) / rpb) * static_cast<long long>(k) > static_cast<long long>(dw_floats)) rpb *= 2;
      const int rb = static_cast<int>((m + rpb - 1) / rpb);
      float* part = ws + dw_floats;
      hipLaunchKernelGGL(wcolsum_partial_kernel, dim3((k + 63) / 64, rb), dim3(256), 0, s, g, d_x,
                         static_cast<long long>(x_stride), M, k, static_cast<int>(rpb), ws, d_db != nullptr ? part : nullptr);
      if (d_dw != nullptr)
        launch_splitk_reduce(s, static_cast<unsigned>((k + 63) / 64), ws, static_cast<long long>(k), rb, d_dw,
                             d_db != nullptr ? part : nullptr, 1LL, d_db);
      else if (d_db != nullptr)
        launch_splitk_reduce(s, 1u, part, 1LL, rb, d_db);
    }
    return check_launch("logit head backward kernels");
  }
  if (d_dx != nullptr) {
    // dx[m,k] = g[m,n] * W[n,k]: A = g (n contiguous = its K), B(kk=n, col=k) = W[n*k + k] (col contiguous)
    rc = run_gemm<true, false>(g, n, d_w, k, d_dx, M, k, n, nullptr, 0, nullptr, 0, s, dx_stride);
    if (rc != RBX_OK) return rc;
  }
  if (d_dw != nullptr && n == 64 && k == 64 && m >= 8192 && vec_ok(g, n) && vec_ok(d_x, x_stride) && true &&
      dw_floats >= static_cast<size_t>(2 * kCUs) * 64 * 64) {
    const int slabs = (M + 31) / 32;
    int n_wg = (slabs + kSlabWaves - 1) / kSlabWaves;
    if (n_wg > 2 * kCUs) n_wg = 2 * kCUs;
    float* part = ws + dw_floats;
    hipLaunchKernelGGL(tall_dw64_kernel<false>, dim3(n_wg), dim3(64 * kSlabWaves), 0, s, g, static_cast<long long>(n), d_x,
                       static_cast<long long>(x_stride), M, ws, d_db != nullptr ? part : nullptr, 0, nullptr);
    launch_splitk_reduce(s, 64u, ws, 64LL * 64, n_wg, d_dw, d_db != nullptr ? part : nullptr, 64LL, d_db);
    return check_launch("tall dW / db kernels (slab form)");
  }
  if (d_dw != nullptr && n == 128 && k == 64 && m >= 8192 && vec_ok(g, n) && vec_ok(d_x, x_stride) && true &&
      dw_floats >= static_cast<size_t>(2 * kCUs) * 64 * 64) {
    // [m, 128]^T x [m, 64] (the fused K | V projection): the slab kernel once per 64-column half of g (x read twice: 840 MB
    // of coalesced 1 KB requests against tall_dw_kernel<2>'s 630 MB of dword requests, 150 vs 199 us)
    const int slabs = (M + 31) / 32;
    int n_wg = (slabs + kSlabWaves - 1) / kSlabWaves;
    if (n_wg > 2 * kCUs) n_wg = 2 * kCUs;
    float* part = ws + dw_floats;
    for (int half = 0; half < 2; ++half) {
      hipLaunchKernelGGL(tall_dw64_kernel<false>, dim3(n_wg), dim3(64 * kSlabWaves), 0, s, g + 64 * half,
                         static_cast<long long>(n), d_x, static_cast<long long>(x_stride), M, ws,
                         d_db != nullptr ? part : nullptr, 0, nullptr);
      launch_splitk_reduce(s, 64u, ws, 64LL * 64, n_wg, d_dw + static_cast<long long>(half) * 64 * k,
                           d_db != nullptr ? part : nullptr, 64LL, d_db != nullptr ? d_db + 64 * half : nullptr);
    }
    return check_launch("tall dW / db kernels (slab form, two halves)");
  }
  if (d_dw != nullptr && n <= 256 && k <= 64 && m >= 8192) {
    // tall and narrow: one streaming pass over g and x leaves dW and db partials per workgroup (tall_dw_kernel)
    long long n_wg = static_cast<long long>(dw_floats / (static_cast<size_t>(n) * k));
    if (n_wg > tall_wgs_max()) n_wg = tall_wgs_max();
    if (n_wg > (m + 63) / 64) n_wg = (m + 63) / 64;
    int rows_per_wg = static_cast<int>((m + n_wg - 1) / n_wg);
    rows_per_wg = (rows_per_wg + 15) / 16 * 16;
    n_wg = (m + rows_per_wg - 1) / rows_per_wg;
    float* part = ws + dw_floats;
    float* dbp = d_db != nullptr ? part : nullptr;
    const dim3 grid(static_cast<unsigned>(n_wg));
    const long long ldg = n, ldx = x_stride;
    switch ((n + 63) / 64) {
      case 1: hipLaunchKernelGGL(tall_dw_kernel<1>, grid, dim3(256), 0, s, g, ldg, d_x, ldx, M, n, k, rows_per_wg, ws, dbp); break;
      case 2: hipLaunchKernelGGL(tall_dw_kernel<2>, grid, dim3(256), 0, s, g, ldg, d_x, ldx, M, n, k, rows_per_wg, ws, dbp); break;
      case 3: hipLaunchKernelGGL(tall_dw_kernel<3>, grid, dim3(256), 0, s, g, ldg, d_x, ldx, M, n, k, rows_per_wg, ws, dbp); break;
      default: hipLaunchKernelGGL(tall_dw_kernel<4>, grid, dim3(256), 0, s, g, ldg, d_x, ldx, M, n, k, rows_per_wg, ws, dbp); break;
    }
    const long long nk = static_cast<long long>(n) * k;
    long long blocks = (nk + 63) / 64;
    launch_splitk_reduce(s, static_cast<unsigned>(blocks), ws, nk, static_cast<int>(n_wg), d_dw,
                         d_db != nullptr ? part : nullptr, static_cast<long long>(n), d_db);       // dW and db in one launch
    return check_launch("tall dW / db kernels");
  }
  if (d_dw != nullptr) {
    // dW[n,k] = g^T[n,m] * x[m,k]: A(i=n, kk=m) = g[m*n + n] (row contiguous), B(kk=m, col=k) = x[m*k + k]
    rc = run_gemm<false, false>(g, n, d_x, x_stride, d_dw, n, k, M, nullptr, 0, ws, dw_floats, s);
    if (rc != RBX_OK) return rc;
  }
  if (d_db != nullptr) {
    float* part = ws + dw_floats;
    const int rb = static_cast<int>((m + 1023) / 1024);
    hipLaunchKernelGGL(colsum_partial_kernel, dim3((n + 63) / 64, rb), dim3(256), 0, s, g, M, n, 1024, part);
    long long blocks = (n + 63) / 64;
    launch_splitk_reduce(s, static_cast<unsigned>(blocks), part, static_cast<long long>(n), rb, d_db);
    rc = check_launch("bias grad kernels");
  }
  return rc;
}

// dW[64, 64] = (diag(row_scale) dy)^T x and db = its column sums in the slab kernel, the scaled gradient never written
// (the `seqs *= ~timeline_mask` of a SASRec block, sasrec.py:92, in the backward of the Linear in front of it)
extern "C" int rbx_linear_dwdb_scaled(const float* d_x, int64_t x_stride, const float* d_dy, int64_t dy_stride,
                                      const float* d_row_scale, int64_t m, int32_t n, int32_t k, float* d_dw, float* d_db,
                                      void* d_workspace, size_t workspace_bytes, void* stream) {
  if (m == 0) return RBX_OK;
  using namespace rbx;
  if (!d_x || !d_dy || !d_row_scale || !d_dw) return fail(RBX_ERR_INVALID, "linear_dwdb_scaled: NULL tensor");
  if (m < 0 || m > INT_MAX || x_stride < k || dy_stride < n) return fail(RBX_ERR_INVALID, "linear_dwdb_scaled: bad shape");
  const size_t need = rbx_linear_bwd_workspace_size(m, n, k, 0);
  if (d_workspace == nullptr || workspace_bytes < need) return fail(RBX_ERR_WORKSPACE, "linear_dwdb_scaled: workspace too small");
  const size_t dw_floats = dw_ws_floats(n, k);
  if (!(n == 64 && k == 64 && m >= 8192 && vec_ok(d_dy, dy_stride) && vec_ok(d_x, x_stride) &&
        dw_floats >= static_cast<size_t>(2 * kCUs) * 64 * 64))
    return fail(RBX_ERR_UNSUPPORTED, "linear_dwdb_scaled: only [m >= 8192, 64]^T x [m, 64] with 16-byte aligned rows");
  hipStream_t s = as_stream(stream);
  float* ws = static_cast<float*>(d_workspace);
  const int M = static_cast<int>(m);
  const int slabs = (M + 31) / 32;
  int n_wg = (slabs + kSlabWaves - 1) / kSlabWaves;
  if (n_wg > 2 * kCUs) n_wg = 2 * kCUs;
  float* part = ws + dw_floats;
  hipLaunchKernelGGL(tall_dw64_kernel<true>, dim3(n_wg), dim3(64 * kSlabWaves), 0, s, d_dy, static_cast<long long>(dy_stride),
                     d_x, static_cast<long long>(x_stride), M, ws, d_db != nullptr ? part : nullptr, 0, d_row_scale);
  launch_splitk_reduce(s, 64u, ws, 64LL * 64, n_wg, d_dw, d_db != nullptr ? part : nullptr, 64LL, d_db);
  return check_launch("tall dW / db kernels (slab form, scaled rows)");
}

extern "C" int rbx_linear_dx_scaled(const float* d_dy, int64_t dy_stride, const float* d_w, int64_t m, int32_t n, int32_t k,
                                    const float* d_mask, int64_t mask_stride, const float* d_residual,
                                    int64_t residual_stride, const float* d_row_scale, float* d_dx, int64_t dx_stride,
                                    void* stream) {
  if (m == 0) return RBX_OK;
  using namespace rbx;
  if (d_dy == nullptr || d_w == nullptr || d_dx == nullptr || d_row_scale == nullptr)
    return fail(RBX_ERR_INVALID, "linear_dx_scaled: NULL tensor");
  if (m < 0 || n <= 0 || k <= 1 || m > INT_MAX) return fail(RBX_ERR_INVALID, "linear_dx_scaled: bad shape (k must be > 1)");
  if (dy_stride < n || dx_stride < k || (d_mask != nullptr && mask_stride < k) ||
      (d_residual != nullptr && residual_stride < k))
    return fail(RBX_ERR_INVALID, "linear_dx_scaled: a row stride is shorter than its row");
  Epi epi{};
  epi.res = d_residual;
  epi.ldres = static_cast<long long>(residual_stride);
  epi.mask = d_mask;
  epi.ldmask = static_cast<long long>(mask_stride);
  epi.rowscale = d_row_scale;
  return run_gemm<true, false>(d_dy, dy_stride, d_w, k, d_dx, static_cast<int>(m), k, n, nullptr, 0, nullptr, 0,
                               as_stream(stream), dx_stride, epi);
}

// ---- bf16 planes of a weight matrix for the split-operand GEMM (see gemm_bx6_kernel) -------------------------------------
extern "C" size_t rbx_split_bf16_size(int32_t rows, int32_t cols, int32_t transpose) {
  if (rows <= 0 || cols <= 0) return 0;
  const long long orows = transpose ? cols : rows, ocols = transpose ? rows : cols;
  const long long cp = (ocols + rbx::SBK - 1) / rbx::SBK * rbx::SBK;
  return static_cast<size_t>(3 * orows * cp * 2);
}

extern "C" int rbx_split_bf16(const float* d_src, int64_t ld, int32_t rows, int32_t cols, int32_t transpose, void* d_out,
                              void* stream) {
  using namespace rbx;
  if (rows <= 0 || cols <= 0) return RBX_OK;
  if (d_src == nullptr || d_out == nullptr || ld < cols) return fail(RBX_ERR_INVALID, "split_bf16: bad arguments");
  if ((reinterpret_cast<uintptr_t>(d_out) & 15) != 0) return fail(RBX_ERR_INVALID, "split_bf16: output must be 16-byte aligned");
  const long long pairs = static_cast<long long>(rbx_split_bf16_size(rows, cols, transpose) / 12);
  long long blocks = (pairs + 255) / 256;
  if (blocks > kCUs * 8) blocks = kCUs * 8;
  hipLaunchKernelGGL(split_bf16_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, as_stream(stream), d_src,
                     static_cast<long long>(ld), rows, cols, transpose, static_cast<unsigned short*>(d_out));
  return check_launch("split_bf16_kernel");
}

extern "C" int rbx_split_register(const float* d_w, const void* d_planes, int32_t rows, int32_t cols, int32_t transposed) {
  using namespace rbx;
  if (d_w == nullptr || d_planes == nullptr || rows <= 0 || cols <= 0) return fail(RBX_ERR_INVALID, "split_register: bad arguments");
  std::lock_guard<std::mutex> lock(g_split_mu);
  int slot = -1;
  for (int i = 0; i < kSplitSlots; ++i)
    if (g_split[i].w == d_w && g_split[i].transposed == (transposed ? 1 : 0)) slot = i;
  for (int i = 0; i < kSplitSlots && slot < 0; ++i)
    if (g_split[i].planes == nullptr) slot = i;
  if (slot < 0) return fail(RBX_ERR_UNSUPPORTED, "split_register: all %d slots are taken", kSplitSlots);
  g_split[slot] = SplitEntry{d_w, static_cast<const unsigned short*>(d_planes), rows, cols, transposed ? 1 : 0};
  return RBX_OK;
}

extern "C" uint64_t rbx_gemm_bx6_count(void) { return rbx::g_bx6_launches.load(std::memory_order_relaxed); }

extern "C" int rbx_split_unregister(const float* d_w) {
  using namespace rbx;
  std::lock_guard<std::mutex> lock(g_split_mu);
  for (int i = 0; i < kSplitSlots; ++i)
    if (g_split[i].w == d_w) g_split[i] = SplitEntry{nullptr, nullptr, 0, 0, 0};
  return RBX_OK;
}
