// rbx_attn_stream.h -- the causal attention kernels of rbx_attn_mfma.hip with K and V STREAMED through LDS in tiles of 32 keys
// (head_dim 64, 3..7 tiles: 64 < L <= 224; included by rbx_attn_mfma.hip, same reference behaviour:
// third_party/rechub/models/matching/sasrec.py:79-94).
//
// The resident form keeps a whole sequence's K and V in LDS (104 KB at L = 200): one workgroup per CU, so a CU's memory phase
// (the next sequence's 102 KB) is hidden only by registers, and the eight wavefronts meet at a barrier per sequence.  Here a
// workgroup holds TWO tiles' worth of ring (2 stages x {K, V} x {sequence A, sequence B} x 8 KB = 64 KB): two workgroups per
// CU, each one's loads under the other's MFMAs.
//
//   * A workgroup serves a PAIR of sequences and has nT compute wavefronts + one loader wavefront.  Iteration i of its
//     nT + 1 iterations holds key tile i of sequence A and key tile nT - i of sequence B.  Compute wavefront t owns query
//     tile t of A (needs key tiles 0..t: iterations 0..t) and then query tile nT-1-t of B (key tiles nT-1-t..0: iterations
//     t+1..nT): EXACTLY one tile step per wavefront per iteration, nT + 1 steps each -- the causal triangle of A and the
//     mirrored one of B make a rectangle, so the per-iteration barrier costs no balance (a single sequence per workgroup in
//     lockstep would run 10 step-times for 7 steps of work per wavefront).  One set of accumulators and one Q tile live at
//     a time.
//   * The loader wavefront moves the tiles by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write pass):
//     at iteration i it issues the 32 x 1 KB pieces of iteration i + 1 into the other stage and waits for them in front of
//     the barrier that ends iteration i.  Compute wavefronts never wait for memory except for their own Q tile.
//   * LDS-DMA writes lane-linear (16 bytes per lane at base + 16 lane), so a tile is row-major [32][64] WITHOUT padding and
//     the conflict-free layout comes from the source side: physical 16-byte chunk p of row r holds logical chunk
//     p ^ (r & 15).  K rows (lane = row) are read with ds_read_b128 -- the 16 lanes of a b128 group hit 16 different
//     chunks = all 64 banks; V columns (lane = column, one row per half-wave) with ds_read_b32 -- 8 chunks x 4 words = 32 banks.
//   * Rows beyond L in the last tile are fetched from row L - 1 (finite values; their keys are masked / their p is 0).
#pragma once

namespace rbx {

constexpr int kStreamTile = kT * 64;               // floats of one [32][64] tile (8 KB)
constexpr int kStreamStage = 4 * kStreamTile;      // K_A, V_A, K_B, V_B
constexpr size_t kStreamLds = 2 * kStreamStage * sizeof(float);

__device__ __forceinline__ void stream_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// one [32][64] tile: rows row0.. of g -> `tile`, swizzled on the source side (see above); 8 pieces of 4 rows
__device__ __forceinline__ void dma_tile(const float* __restrict__ g, const long long ld, const int row0, const int rows,
                                         float* tile) {
  const int lane = threadIdx.x & 63;
  const int rsub = lane >> 4, p = lane & 15;
#pragma unroll
  for (int n = 0; n < 8; ++n) {
    const int r = 4 * n + rsub;
    int gr = row0 + r;
    gr = gr < rows ? gr : rows - 1;
    const float* src = g + static_cast<long long>(gr) * ld + ((p ^ (r & 15)) << 2);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(tile + n * 256), 16, 0, 0);
  }
}

// acc[row = key li of the tile][col = lane's query] = sum_d Kt[li][d] * reg[d]      (column pairing as load_tile_regs)
template <bool X6>
__device__ __forceinline__ f32x16 tile_dot_sw(const float* __restrict__ Kt, const TileOp<64, X6>& op) {
  const int lane = threadIdx.x & 63;
  const int li = lane & 31, half = lane >> 5;
  const int b = (half * 8) ^ (li & 15);
  const float* row = Kt + li * 64;
  f32x16 acc, acc1;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = acc1[r] = 0.f;
  if constexpr (X6) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const float4 a0 = *reinterpret_cast<const float4*>(row + ((b ^ (2 * s)) << 2));
      const float4 a1 = *reinterpret_cast<const float4*>(row + ((b ^ (2 * s + 1)) << 2));
      const float x[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
      mfma6(acc, acc1, split8(x), op.p[s]);
    }
  } else {
#pragma unroll
    for (int cc = 0; cc < 8; ++cc) {
      const float4 a = *reinterpret_cast<const float4*>(row + ((b ^ cc) << 2));
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, op.v[4 * cc], acc, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, op.v[4 * cc + 1], acc1, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, op.v[4 * cc + 2], acc, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, op.v[4 * cc + 3], acc1, 0, 0, 0);
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] += acc1[r];
  return acc;
}

// out[dt][row = d][col = lane] += sum_r Vt[tile_row(r)][dt * 32 + li] * w[r]
template <bool X6>
__device__ __forceinline__ void tile_accumulate_sw(const float* __restrict__ Vt, const f32x16& w, f32x16 (&out)[2]) {
  const int lane = threadIdx.x & 63;
  const int li = lane & 31, half = lane >> 5;
  const int cl = (li >> 2) ^ (4 * half);
  const float* base = Vt + half * 4 * 64 + (li & 3);
  if constexpr (X6) {
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    float x[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = w[8 * s + e];
    const Split8 b = split8(x);
    Split8 a[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int r = 8 * s + e;
        x[e] = base[((r & 3) + 8 * (r >> 2)) * 64 + ((cl ^ (r & 3)) << 2) + ((dt ^ ((r >> 2) & 1)) << 5)];
      }
      a[dt] = split8(x);
    }
    mfma6x2(out[0], out[1], a[0], a[1], b);
  }
  } else {
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float* a = base + ((r & 3) + 8 * (r >> 2)) * 64 + ((cl ^ (r & 3)) << 2);
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
      out[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(dt ^ ((r >> 2) & 1)) << 5], w[r], out[dt], 0, 0, 0);
  }
  }
}

// the wavefront's own tile as B operands, unscaled and with NO arithmetic on the loaded values: reg[s] = g[row0 + li][half * 32 + s],
// rows beyond `rows` read row rows - 1 (never stored).  Nothing uses the registers here, so the compiler's wait sits at the
// first MFMA that reads them -- behind the barrier -- and not behind the loads (load_tile_regs scales in place: waited at once).
__device__ __forceinline__ void load_tile_raw(const float* __restrict__ g, const long long ld, int row0, int rows, float (&reg)[32]) {
  const int lane = threadIdx.x & 63;
  int row = row0 + (lane & 31);
  row = row < rows ? row : rows - 1;
  const float4* src = reinterpret_cast<const float4*>(g + static_cast<long long>(row) * ld + (lane >> 5) * 32);
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const float4 v = src[q];
    reg[4 * q] = v.x; reg[4 * q + 1] = v.y; reg[4 * q + 2] = v.z; reg[4 * q + 3] = v.w;
  }
}

template <bool X6>
__device__ __forceinline__ void stream_load_q(const float* __restrict__ g, const long long ld, int row0, int rows, TileOp<64, X6>& op,
                                              float (&tmp)[32]) {
  load_tile_raw(g, ld, row0, rows, tmp);
  if constexpr (!X6) op.v = tmp;
}
template <bool X6>
__device__ __forceinline__ void stream_split_q(const float (&tmp)[32], TileOp<64, X6>& op) {
  if constexpr (X6) make_op<64, X6>(tmp, op);
}

template <bool DROP>
__global__ __launch_bounds__(512, 4) void attn_stream_fwd_kernel(const float* __restrict__ Q0, const float* __restrict__ K0,
                                                                  const float* __restrict__ V0, const int L, const float scale,
                                                                  float* __restrict__ O0, float* __restrict__ LSE,
                                                                  const DropArgs drop, const AttnLd ld, const long long BH) {
  constexpr int HD = 64;
  extern __shared__ float lds[];
  const int nT = (L + kT - 1) / kT;
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), li = lane & 31, half = lane >> 5;
  const long long bhA = 2 * static_cast<long long>(blockIdx.x), bhB = bhA + 1;
  const bool hasB = bhB < BH;
  if (wid == nT) {                                           // the loader wavefront
    const float* KA = K0 + attn_base(bhA, ld.heads, L, ld.k, HD);
    const float* VA = V0 + attn_base(bhA, ld.heads, L, ld.v, HD);
    const float* KB = K0 + attn_base(hasB ? bhB : bhA, ld.heads, L, ld.k, HD);
    const float* VB = V0 + attn_base(hasB ? bhB : bhA, ld.heads, L, ld.v, HD);
    for (int it = 0; it <= nT + 1; ++it) {                   // (it = 0: the prologue's barrier)
      if (it <= nT) {
        float* st = lds + (it & 1) * kStreamStage;
        if (it < nT) {
          dma_tile(KA, ld.k, it * kT, L, st);
          dma_tile(VA, ld.v, it * kT, L, st + kStreamTile);
        }
        if (it >= 1 && hasB) {
          dma_tile(KB, ld.k, (nT - it) * kT, L, st + 2 * kStreamTile);
          dma_tile(VB, ld.v, (nT - it) * kT, L, st + 3 * kStreamTile);
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      stream_barrier();
    }
    return;
  }
  const int t = wid;
  unsigned dk0 = 0, dk1 = 0;
  if (DROP) drop_seed(drop, &dk0, &dk1);
  constexpr bool X6 = (RBX_ATTN_BF16X6 & 1) != 0, XA = (RBX_ATTN_BF16X6 & 2) != 0;
  TileOp<HD, X6> qop;
  float qsplit[HD / 2];                                      // (f32 form: these registers ARE the operand)
  bool fresh = true;                                         // qsplit holds a tile that qop does not yet
  auto load_q = [&](const long long bh, const int row0) {
    stream_load_q<X6>(Q0 + attn_base(bh, ld.heads, L, ld.q, HD), ld.q, row0, L, qop, qsplit);
    fresh = true;
  };
  load_q(bhA, t * kT);
  f32x16 oacc[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;
  float m = -INFINITY, lsum = 0.f;
  stream_barrier();                                          // stage 0 holds iteration 0's tiles
  for (int it = 0; it <= nT; ++it) {
    const bool isA = it <= t;
    if (isA || hasB) {
      const long long bh = isA ? bhA : bhB;
      const int qt = isA ? t : nT - 1 - t, kt = isA ? it : nT - it;
      const int i0 = qt * kT, qi = i0 + li, j0 = kt * kT;
      const float* Kt = lds + (it & 1) * kStreamStage + (isA ? 0 : 2 * kStreamTile);
      const float* Vt = Kt + kStreamTile;
      if (X6 && fresh) {
        stream_split_q<X6>(qsplit, qop);
        fresh = false;
      }
      f32x16 s = tile_dot_sw<X6>(Kt, qop);                     // S^T[key][query]
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] *= scale;            // (the Q tile is unscaled: load_tile_raw)
      if (kt == qt) {                                        // only the diagonal tile has masked keys (and the tail beyond L)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int kj = j0 + tile_row(r, half);
          if (kj >= L || kj > qi) s[r] = -INFINITY;
        }
      }
      float mx = -INFINITY;
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[r]);
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float mn = fmaxf(m, mx);
      const float alpha = (mn == -INFINITY) ? 1.f : __expf(m - mn);
      float ps = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s[r] = (s[r] == -INFINITY) ? 0.f : __expf(s[r] - mn);
        ps += s[r];
      }
      ps += __shfl_xor(ps, 32, 64);
      lsum = lsum * alpha + ps;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[dt][r] *= alpha;
      m = mn;
      if (DROP) {                                            // the same mask words as the resident kernel (and the backward)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          unsigned c[4];
          drop_block(static_cast<unsigned>(qi) >> 2, static_cast<unsigned>(j0 + 8 * g + 4 * half) >> 2,
                     static_cast<unsigned long long>(bh), (qi & 3) >> 1, dk0, dk1, c);
#pragma unroll
          for (int q = 0; q < 4; ++q) s[4 * g + q] = drop_keep(c, qi & 1, q, drop.thr16) ? s[4 * g + q] * drop.scale : 0.f;
        }
      }
      tile_accumulate_sw<XA>(Vt, s, oacc);                      // O^T[d][query] += V^T P^T
      if (it == t || it == nT) {                             // the tile is complete
        float* O = O0 + attn_base(bh, ld.heads, L, ld.o, HD);
        store_transposed<HD>(O, ld.o, i0, L, 1.0f / lsum, oacc);
        if (half == 0 && qi < L) LSE[bh * L + qi] = m + __logf(lsum);
        if (it == t && hasB) {                               // on to sequence B's tile: its Q rows arrive under the barrier
          load_q(bhB, (nT - 1 - t) * kT);
#pragma unroll
          for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;
          m = -INFINITY;
          lsum = 0.f;
        }
      }
    }
    stream_barrier();
  }
}

}  // namespace rbx
