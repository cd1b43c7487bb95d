// rbx_attn_planes.h -- the streamed causal attention forward of rbx_attn_stream.h with the K / V tiles kept in LDS as bf16
// PLANES (x = h + m + l, three bf16 values per f32 element), split ONCE per tile by the workgroup instead of once per
// wavefront and tile product (rbx_attn_mfma.hip: `split8`, ~10 VALU operations per element pair and wavefront).  Opt-in
// (RBX_ATTN_STREAM=2), d = 64, causal, 7 key tiles (192 < L <= 224); included by rbx_attn_mfma.hip.
//
//   * Schedule: the pair schedule of rbx_attn_stream.h (sequence A's key tiles upwards, B's downwards, one tile step per
//     wavefront per iteration), but PERSISTENT: a workgroup per CU walks pairs, the ring runs across pair boundaries.
//   * No loader wavefront and no f32 landing zone: every thread fetches one float4 of each of the next iteration's four
//     tiles (K_A, V_A, K_B, V_B: 4 x 512 float4) into registers while it computes, splits them after its tile step and
//     writes the three planes of the OTHER stage; one barrier per iteration.
//   * A tile's plane is [2 column halves][32 rows][32 columns] bf16, 64-byte rows, the four 16-byte chunks of a row
//     XOR-swizzled with (row >> 2) & 3.  Row-wise fragments (S^T = K Q^T: lane = key row, 8 consecutive d) are one
//     ds_read_b128 per plane, conflict-free by the swizzle; column-wise fragments (O^T += V^T P^T: lane = d, 8 keys) come
//     out of the SAME planes through ds_read_b64_tr_b16 (profiles/r04/tr16_probe.txt: within 16 lanes, lane i receives
//     element i % 4 of the 8 bytes addressed by lanes i / 4, i / 4 + 4, i / 4 + 8, i / 4 + 12): four 64-byte rows of a
//     read cover all 64 banks.
//   * LDS: 2 stages x 4 tiles x 12 KB = 96 KB, one workgroup of 8 wavefronts per CU.
#pragma once

namespace rbx {

#define RBX_PL_ABL 0   // profiles/ubench/attn_stream.hip: 1 = no tile streaming inside the loop, 2 = no tile steps, 4 = no wait before commit
constexpr int kPlBytes = 4096;                     // one plane of one tile
constexpr int kPlTile = 3 * kPlBytes;              // h, m, l
constexpr int kPlStage = 4 * kPlTile;              // K_A, V_A, K_B, V_B
constexpr size_t kPlanesLds = 2 * kPlStage;

// every wavefront's plane writes are in LDS before any wavefront passes (global loads and stores stay in flight)
__device__ __forceinline__ void stream_barrier_lds() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

typedef short ash4 __attribute__((ext_vector_type(4)));
typedef short ash8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ abf16x8_t pl_tr8(const char* p1, const char* p2) {
  const ash4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) ash4*)(p1));
  const ash4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) ash4*)(p2));
  const ash8 c = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(abf16x8_t, c);
}

// acc[row = key li][col = lane's query] = sum_d K[li][d] q[d], K from the tile's planes
__device__ __forceinline__ f32x16 tile_dot_pl(const char* __restrict__ Kt, const TileOp<64, true>& op) {
  const int lane = threadIdx.x & 63;
  const int li = lane & 31, half = lane >> 5;
  const char* base = Kt + half * 2048 + li * 64;
  const int sw = (li >> 2) & 3;
  f32x16 acc, acc1;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = acc1[r] = 0.f;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int off = (s ^ sw) << 4;
    Split8 a;
    a.h = *reinterpret_cast<const abf16x8_t*>(base + off);
    a.m = *reinterpret_cast<const abf16x8_t*>(base + kPlBytes + off);
    a.l = *reinterpret_cast<const abf16x8_t*>(base + 2 * kPlBytes + off);
    mfma6(acc, acc1, a, op.p[s]);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] += acc1[r];
  return acc;
}

// out[dt][row = d][col = lane] += sum_r V[tile_row(r)][dt * 32 + li] * w[r], V from the tile's planes
__device__ __forceinline__ void tile_accumulate_pl(const char* __restrict__ Vt, const f32x16& w, f32x16 (&out)[2]) {
  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, i = lane & 15, half = lane >> 5;
  const int cidx = ((g & 1) << 1) | ((i & 3) >> 1);
  const int rowoff = (i >> 2) * 64 + (i & 1) * 8 + half * 4 * 64;
  const char* p1 = Vt + rowoff + ((cidx ^ half) << 4);                // rows 16 s + 4 half + j:      (row >> 2) & 3 = half
  const char* p2 = Vt + rowoff + 8 * 64 + ((cidx ^ (half + 2)) << 4);  // rows 16 s + 8 + 4 half + j:  half + 2
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    float x[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = w[8 * s + e];
    const Split8 b = split8(x);
    Split8 a[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      const int o = dt * 2048 + s * 16 * 64;
      a[dt].h = pl_tr8(p1 + o, p2 + o);
      a[dt].m = pl_tr8(p1 + o + kPlBytes, p2 + o + kPlBytes);
      a[dt].l = pl_tr8(p1 + o + 2 * kPlBytes, p2 + o + 2 * kPlBytes);
    }
    mfma6x2(out[0], out[1], a[0], a[1], b);
  }
}

// Every load inside the persistent loop is an asm statement and the ONE wait of an iteration (`pl_arrived`, after the tile
// step, in front of the plane writes) names all their registers: with loads the compiler counts, its wait for the next
// tile's Q rows sat at the loop header as vmcnt(0) and drained the tile prefetch issued just before the barrier.
__device__ __forceinline__ void pl_load(attn_f4& v, const float* src) {
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(src));
}
__device__ __forceinline__ void pl_arrived(attn_f4 (&pf)[4], attn_f4 (&qv)[8]) {
  asm volatile("s_waitcnt vmcnt(0)"
               : "+v"(pf[0]), "+v"(pf[1]), "+v"(pf[2]), "+v"(pf[3]), "+v"(qv[0]), "+v"(qv[1]), "+v"(qv[2]), "+v"(qv[3]),
                 "+v"(qv[4]), "+v"(qv[5]), "+v"(qv[6]), "+v"(qv[7])
               :
               : "memory");
}
// the wavefront's own tile, unscaled: qv[q] = g[row0 + li][half * 32 + 4 q .. + 3] (rows beyond `rows`: row rows - 1).  Issued
// under a (wave-uniform) condition: the destinations are read-write operands, so the registers are the same on both paths.
__device__ __forceinline__ void pl_load_q(const float* __restrict__ g, const long long ld, int row0, int rows, attn_f4 (&qv)[8]) {
  const int lane = threadIdx.x & 63;
  int row = row0 + (lane & 31);
  row = row < rows ? row : rows - 1;
  const float* src = g + static_cast<long long>(row) * ld + (lane >> 5) * 32;
#pragma unroll
  for (int q = 0; q < 8; ++q) asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(qv[q]) : "v"(src + 4 * q));
}

// four floats of row r, columns c4 .. c4 + 3 -> the three planes of `tile`
__device__ __forceinline__ void planes_store(char* tile, const int r, const int c4, const attn_f4 v) {
  unsigned h0, m0, l0, h1, m1, l1;
  attn_split2(af32x2_t{v[0], v[1]}, h0, m0, l0);
  attn_split2(af32x2_t{v[2], v[3]}, h1, m1, l1);
  char* dst = tile + (c4 >> 5) * 2048 + r * 64 + ((((c4 & 31) >> 3) ^ ((r >> 2) & 3)) << 4) + (c4 & 7) * 2;
  *reinterpret_cast<uint2*>(dst) = make_uint2(h0, h1);
  *reinterpret_cast<uint2*>(dst + kPlBytes) = make_uint2(m0, m1);
  *reinterpret_cast<uint2*>(dst + 2 * kPlBytes) = make_uint2(l0, l1);
}

template <bool DROP>
__global__ __launch_bounds__(512) void attn_planes_fwd_kernel(const float* __restrict__ Q0, const float* __restrict__ K0,
                                                              const float* __restrict__ V0, const int L, const float scale,
                                                              float* __restrict__ O0, float* __restrict__ LSE,
                                                              const DropArgs drop, const AttnLd ld, const long long BH) {
  constexpr int HD = 64, nT = 7;
  extern __shared__ float lds_f[];
  char* lds = reinterpret_cast<char*>(lds_f);
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6), li = lane & 31, half = lane >> 5;
  const long long pairs = (BH + 1) / 2;
  const int pr = tid >> 4, pc4 = (tid & 15) * 4;             // this thread's float4 of every tile: row, first column
  attn_f4 pf[4];
  // the four tiles of iteration `it` of pair `p` -> registers (rows beyond L: row L - 1)
  // (UNCONDITIONAL: an asm load inside a branch makes the join copy its destination registers while the data is still in
  //  flight -- the landing load then overwrites whatever the allocator put there next, e.g. an address.  Slots without a
  //  tile -- A at it = nT, B at it = 0 or an odd count's last pair -- fetch a neighbouring tile that commit() skips.)
  auto fetch = [&](const long long p, const int it) {
    const long long bhA = 2 * p, bhB = (bhA + 1 < BH) ? bhA + 1 : bhA;
    const int ta = it < nT ? it : nT - 1, tb = it >= 1 ? nT - it : nT - 1;
    int ra = ta * kT + pr, rb = tb * kT + pr;
    ra = ra < L ? ra : L - 1;
    rb = rb < L ? rb : L - 1;
    pl_load(pf[0], K0 + attn_base(bhA, ld.heads, L, ld.k, HD) + static_cast<long long>(ra) * ld.k + pc4);
    pl_load(pf[1], V0 + attn_base(bhA, ld.heads, L, ld.v, HD) + static_cast<long long>(ra) * ld.v + pc4);
    pl_load(pf[2], K0 + attn_base(bhB, ld.heads, L, ld.k, HD) + static_cast<long long>(rb) * ld.k + pc4);
    pl_load(pf[3], V0 + attn_base(bhB, ld.heads, L, ld.v, HD) + static_cast<long long>(rb) * ld.v + pc4);
  };
  auto commit = [&](const long long p, const int it, char* stage) {
    if (it < nT) {
      planes_store(stage, pr, pc4, pf[0]);
      planes_store(stage + kPlTile, pr, pc4, pf[1]);
    }
    if (it >= 1 && 2 * p + 1 < BH) {
      planes_store(stage + 2 * kPlTile, pr, pc4, pf[2]);
      planes_store(stage + 3 * kPlTile, pr, pc4, pf[3]);
    }
  };
  const int t = wid;
  const bool computes = wid < nT;
  unsigned dk0 = 0, dk1 = 0;
  if (DROP) drop_seed(drop, &dk0, &dk1);
  TileOp<HD, true> qop;
  attn_f4 qv[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) qv[q] = attn_f4{0.f, 0.f, 0.f, 0.f};
  bool fresh = false;
  f32x16 oacc[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;
  float m = -INFINITY, lsum = 0.f;
  long long p = blockIdx.x;
  if (p >= pairs) return;
  fetch(p, 0);
  if (computes) {
    pl_load_q(Q0 + attn_base(2 * p, ld.heads, L, ld.q, HD), ld.q, t * kT, L, qv);
    fresh = true;
  }
  pl_arrived(pf, qv);
  commit(p, 0, lds);
  fetch(p, 1);
  stream_barrier_lds();
  int stage = 0;
  for (;;) {
    const long long bhA = 2 * p, bhB = bhA + 1, pn = p + gridDim.x;
    const bool hasB = bhB < BH;
    for (int it = 0; it <= nT; ++it) {
      const char* st = lds + stage * kPlStage;
      const bool isA = it <= t;
      const bool active = computes && (isA || hasB);
      const bool last = active && (isA ? it == t : it == nT);   // this step completes the wavefront's tile
      const long long bh = isA ? bhA : bhB;
      const int qt_ = isA ? t : nT - 1 - t;
      if (active && !(RBX_PL_ABL & 2)) {
        const int qt = qt_, kt = isA ? it : nT - it;
        const int i0 = qt * kT, qi = i0 + li, j0 = kt * kT;
        const char* Kt = st + (isA ? 0 : 2 * kPlTile);
        const char* Vt = Kt + kPlTile;
        if (fresh) {
          float qraw[HD / 2];
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            qraw[4 * q] = qv[q][0]; qraw[4 * q + 1] = qv[q][1]; qraw[4 * q + 2] = qv[q][2]; qraw[4 * q + 3] = qv[q][3];
          }
          make_op<HD, true>(qraw, qop);
          fresh = false;
        }
        // the next tile's Q rows are requested now and split at its first step (ONE statement site: two sites under
        // different conditions got different destination registers and a copy at the join, before the data had landed)
        const bool to_b = isA && hasB;
        if (last && (to_b || pn < pairs))
          pl_load_q(Q0 + attn_base(to_b ? bhB : 2 * pn, ld.heads, L, ld.q, HD), ld.q, (to_b ? nT - 1 - t : t) * kT, L, qv);
        f32x16 s = tile_dot_pl(Kt, qop);                     // S^T[key][query]
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] *= scale;
        if (kt == qt) {
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
        if (DROP) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            unsigned c[4];
            drop_block(static_cast<unsigned>(qi) >> 2, static_cast<unsigned>(j0 + 8 * g + 4 * half) >> 2,
                       static_cast<unsigned long long>(bh), (qi & 3) >> 1, dk0, dk1, c);
#pragma unroll
            for (int q = 0; q < 4; ++q) s[4 * g + q] = drop_keep(c, qi & 1, q, drop.thr16) ? s[4 * g + q] * drop.scale : 0.f;
          }
        }
        tile_accumulate_pl(Vt, s, oacc);                     // O^T[d][query] += V^T P^T
      }
      // the next iteration's tiles: registers -> the other stage; then the fetch of the one after it
      if (!(RBX_PL_ABL & 4)) pl_arrived(pf, qv);
      const bool more_here = it < nT;
      const long long p1 = more_here ? p : pn;               // the next iteration's slot (pair, it) and the one after it
      const int it1 = more_here ? it + 1 : 0;
      const long long p2 = it1 < nT ? p1 : p1 + gridDim.x;
      const int it2 = it1 < nT ? it1 + 1 : 0;
      if (!(RBX_PL_ABL & 1)) {
        if (p1 < pairs) commit(p1, it1, lds + (stage ^ 1) * kPlStage);
        fetch(p2 < pairs ? p2 : p, it2);                     // (one call site, no branch around the loads)
      }
      if (last) {                                            // (the stores behind the wait: nothing waits for them)
        const int i0 = qt_ * kT, qi = i0 + li;
        float* O = O0 + attn_base(bh, ld.heads, L, ld.o, HD);
        store_transposed<HD>(O, ld.o, i0, L, 1.0f / lsum, oacc);
        if (half == 0 && qi < L) LSE[bh * L + qi] = m + __logf(lsum);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;
        m = -INFINITY;
        lsum = 0.f;
        fresh = true;
      }
      stream_barrier_lds();
      stage ^= 1;
    }
    if (pn >= pairs) break;
    p = pn;
  }
}

}  // namespace rbx
