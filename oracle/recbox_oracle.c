/* recbox_oracle.c -- plain-C CPU restatement of the hot path's index/gather/scatter
 * arithmetic.  TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it; the product (recbox_amd/) never does.
 *
 * It speaks the SAME descriptor struct as the C ABI (include/recbox_hip.h, rbx_field_t) but
 * every pointer is a HOST pointer, and each function restates the reference ops it stands for
 * (paths relative to /root/reference/recbox):
 *   orc_embed_fwd / orc_embed_bwd
 *       nn.Embedding / nn.Linear(1,D) lookups + pooling + stack/cat and their autograd backward:
 *       core/pytorch/layers/embedding.py:116-138, core/pytorch/layers/sequence.py:8-20,
 *       ranking/pytorch/layers/embeddings/feature_embedding.py:188-214, ranking/pytorch/layers/
 *       pooling.py:26-40, third_party/rechub/basic/layers.py:66-116,135-148,187-230
 *   orc_fm_fwd / orc_fm_bwd
 *       LogisticRegression + InnerProductInteraction("product_sum") + FactorizationMachine:
 *       ranking/pytorch/layers/blocks/logistic_regression.py:30-35,
 *       ranking/pytorch/layers/interactions/inner_product.py:41-48, blocks/factorization_machine.py:30-34
 *   orc_interaction_fwd      inner_product.py:40-56 (all four modes), rechub/basic/layers.py:286-292
 *   orc_l2norm_fwd / orc_pairdot_fwd   third_party/rechub/models/matching/dssm.py:48,57,65
 * Pinned by tests/test_c_oracle.py against the fixtures generated from the live reference
 * (tests/golden/, oracle/gen_golden.py).  Sums run left to right in fp32 like a scalar loop.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "recbox_hip.h"

static int64_t id_at(const rbx_field_t* f, int64_t idx) {
  switch (f->ids_dtype) {
    case RBX_I32: return ((const int32_t*)f->ids)[idx];
    case RBX_I64: return ((const int64_t*)f->ids)[idx];
    case RBX_F32: return (int64_t)((const float*)f->ids)[idx];   /* .long() truncates */
    default: return (int64_t)((const double*)f->ids)[idx];
  }
}

static float value_at(const rbx_field_t* f, int64_t idx) {
  switch (f->ids_dtype) {
    case RBX_I32: return (float)((const int32_t*)f->ids)[idx];
    case RBX_I64: return (float)((const int64_t*)f->ids)[idx];
    case RBX_F32: return ((const float*)f->ids)[idx];
    default: return (float)((const double*)f->ids)[idx];           /* .float() */
  }
}

/* returns 0, or 1 when an id was out of range (the reference raises IndexError) */
int orc_embed_fwd(const rbx_field_t* fields, int32_t n, int64_t B, float* out, int64_t stride_b, float* row_scale) {
  int bad = 0;
  for (int32_t fi = 0; fi < n; ++fi) {
    const rbx_field_t* f = &fields[fi];
    const int D = f->dim, L = f->seq_len;
#pragma omp parallel for reduction(| : bad)
    for (int64_t b = 0; b < B; ++b) {
      float* dst = out + b * stride_b + f->out_off;
      if (f->kind == RBX_FIELD_DENSE) { dst[0] = value_at(f, b * f->ids_stride_b); continue; }
      if (f->kind == RBX_FIELD_NUMERIC) {
        const float x = value_at(f, b * f->ids_stride_b);
        for (int d = 0; d < D; ++d) dst[d] = x * f->table[d];
        continue;
      }
      if (f->pool != RBX_POOL_CONCAT) for (int d = 0; d < D; ++d) dst[d] = 0.f;
      float count = 0.f;
      for (int l = 0; l < L; ++l) {
        const int64_t id = id_at(f, b * f->ids_stride_b + l * f->ids_stride_l);
        const int ok = id >= 0 && id < f->vocab;
        if (!ok) bad |= 1;
        const float* row = ok ? f->table + id * D : NULL;
        if (f->pool == RBX_POOL_NONE) {
          for (int d = 0; d < D; ++d) dst[d] = row ? row[d] : 0.f;
        } else if (f->pool == RBX_POOL_CONCAT) {
          for (int d = 0; d < D; ++d) dst[(int64_t)l * D + d] = row ? row[d] : 0.f;
        } else {
          const int id_pool = (f->pool == RBX_POOL_MEAN_ID || f->pool == RBX_POOL_SUM_ID);
          const int masked = id_pool && id == f->mask_id;         /* InputMask: weight 0 */
          if (row && !masked) for (int d = 0; d < D; ++d) dst[d] += row[d];
          if (f->pool == RBX_POOL_MEAN_VALUE) {                   /* value mask: sum_d row != 0 */
            float s = 0.f;
            if (row) for (int d = 0; d < D; ++d) s += row[d];
            count += (s != 0.f) ? 1.f : 0.f;
          } else if (f->pool == RBX_POOL_MEAN_ID) {
            count += masked ? 0.f : 1.f;
          }
        }
      }
      if (f->pool == RBX_POOL_MEAN_VALUE || f->pool == RBX_POOL_MEAN_ID) {
        const float den = count + f->eps;
        for (int d = 0; d < D; ++d) dst[d] = dst[d] / den;
        if (row_scale) row_scale[(int64_t)fi * B + b] = 1.0f / den;
      }
    }
  }
  return bad;
}

/* dense gradients accumulated into fields[f].grad (caller zero-fills), sample order */
void orc_embed_bwd(const rbx_field_t* fields, int32_t n, int64_t B, const float* dout, int64_t stride_b,
                   const float* row_scale) {
  for (int32_t fi = 0; fi < n; ++fi) {
    const rbx_field_t* f = &fields[fi];
    if (f->kind == RBX_FIELD_DENSE || f->grad == NULL) continue;
    const int D = f->dim, L = f->seq_len;
    for (int64_t b = 0; b < B; ++b) {
      const float* g = dout + b * stride_b + f->out_off;
      if (f->kind == RBX_FIELD_NUMERIC) {
        const float x = value_at(f, b * f->ids_stride_b);
        for (int d = 0; d < D; ++d) f->grad[d] += x * g[d];
        continue;
      }
      for (int l = 0; l < L; ++l) {
        const int64_t id = id_at(f, b * f->ids_stride_b + l * f->ids_stride_l);
        if (id < 0 || id >= f->vocab) continue;
        if (id == f->padding_idx) continue;                        /* nn.Embedding(padding_idx): zero grad row */
        const int id_pool = (f->pool == RBX_POOL_MEAN_ID || f->pool == RBX_POOL_SUM_ID);
        if (id_pool && id == f->mask_id) continue;
        float w = 1.f;
        if (f->pool == RBX_POOL_MEAN_VALUE || f->pool == RBX_POOL_MEAN_ID) w = row_scale[(int64_t)fi * B + b];
        const float* src = (f->pool == RBX_POOL_CONCAT) ? g + (int64_t)l * D : g;
        float* dst = f->grad + id * D;
        for (int d = 0; d < D; ++d) dst[d] += w * src[d];
      }
    }
  }
}

void orc_fm_fwd(const rbx_field_t* emb, const rbx_field_t* lr, int32_t n, int64_t B, const float* bias,
                float* logit, float* ssum) {
  const int D = emb ? emb[0].dim : 1;
#pragma omp parallel for
  for (int64_t b = 0; b < B; ++b) {
    float s[1024], q[1024];
    for (int d = 0; d < D; ++d) s[d] = q[d] = 0.f;
    float first = 0.f;
    for (int32_t fi = 0; fi < n; ++fi) {
      const rbx_field_t* lead = emb ? &emb[fi] : &lr[fi];
      float x = 1.f;
      int64_t id = 0;
      if (lead->kind == RBX_FIELD_NUMERIC) x = value_at(lead, b * lead->ids_stride_b);
      else {
        id = id_at(lead, b * lead->ids_stride_b);
        if (id < 0 || id >= lead->vocab) { id = 0; x = 0.f; }
      }
      if (emb) {
        const float* row = emb[fi].table + id * D;
        for (int d = 0; d < D; ++d) { const float e = row[d] * x; s[d] += e; q[d] += e * e; }
      }
      if (lr) first += lr[fi].table[id] * x;
    }
    float fm = 0.f;
    if (emb) for (int d = 0; d < D; ++d) fm += (s[d] * s[d] - q[d]) * 0.5f;
    logit[b] = fm + first + (bias ? bias[0] : 0.f);
    if (emb && ssum) for (int d = 0; d < D; ++d) ssum[b * D + d] = s[d];
  }
}

/* dE_f = g (S - e_f); dLR_f = g; accumulated into .grad of emb / lr (dense), dbias += sum g */
void orc_fm_bwd(const rbx_field_t* emb, const rbx_field_t* lr, int32_t n, int64_t B, const float* g,
                const float* ssum, float* dbias) {
  const int D = emb ? emb[0].dim : 1;
  for (int64_t b = 0; b < B; ++b) {
    for (int32_t fi = 0; fi < n; ++fi) {
      const rbx_field_t* lead = emb ? &emb[fi] : &lr[fi];
      if (lead->kind == RBX_FIELD_NUMERIC) {
        const float x = value_at(lead, b * lead->ids_stride_b);
        if (emb && emb[fi].grad)
          for (int d = 0; d < D; ++d) emb[fi].grad[d] += g[b] * x * (ssum[b * D + d] - x * emb[fi].table[d]);
        if (lr && lr[fi].grad) lr[fi].grad[0] += g[b] * x;
        continue;
      }
      const int64_t id = id_at(lead, b * lead->ids_stride_b);
      if (id < 0 || id >= lead->vocab) continue;
      if (emb && emb[fi].grad && id != emb[fi].padding_idx) {
        const float* row = emb[fi].table + id * D;
        float* dst = emb[fi].grad + id * D;
        for (int d = 0; d < D; ++d) dst[d] += g[b] * (ssum[b * D + d] - row[d]);
      }
      if (lr && lr[fi].grad && id != lr[fi].padding_idx) lr[fi].grad[id] += g[b];
    }
    if (dbias) dbias[0] += g[b];
  }
}

void orc_interaction_fwd(const float* emb, int64_t B, int32_t F, int32_t D, int32_t mode, float* out) {
  const int64_t P = (int64_t)F * (F - 1) / 2;
#pragma omp parallel for
  for (int64_t b = 0; b < B; ++b) {
    const float* e = emb + b * F * D;
    if (mode <= 1) {
      float tot = 0.f;
      for (int d = 0; d < D; ++d) {
        float s = 0.f, q = 0.f;
        for (int f = 0; f < F; ++f) { s += e[f * D + d]; q += e[f * D + d] * e[f * D + d]; }
        const float v = (s * s - q) * 0.5f;
        if (mode == 1) out[b * D + d] = v; else tot += v;
      }
      if (mode == 0) out[b] = tot;
    } else {
      int64_t p = 0;
      for (int i = 0; i < F; ++i)
        for (int j = i + 1; j < F; ++j, ++p) {
          if (mode == 2) {
            float acc = 0.f;
            for (int d = 0; d < D; ++d) acc += e[i * D + d] * e[j * D + d];
            out[b * P + p] = acc;
          } else {
            for (int d = 0; d < D; ++d) out[(b * P + p) * D + d] = e[i * D + d] * e[j * D + d];
          }
        }
    }
  }
}

void orc_l2norm_fwd(const float* x, int64_t rows, int32_t D, float eps, float* y) {
  for (int64_t r = 0; r < rows; ++r) {
    float ss = 0.f;
    for (int d = 0; d < D; ++d) ss += x[r * D + d] * x[r * D + d];
    float nrm = sqrtf(ss);
    if (nrm < eps) nrm = eps;
    for (int d = 0; d < D; ++d) y[r * D + d] = x[r * D + d] / nrm;
  }
}

void orc_pairdot_fwd(const float* u, const float* v, int64_t B, int32_t N, int32_t D, float scale, float* out) {
  for (int64_t b = 0; b < B; ++b)
    for (int n = 0; n < N; ++n) {
      float acc = 0.f;
      for (int d = 0; d < D; ++d) acc += u[b * D + d] * v[(b * N + n) * D + d];
      out[b * N + n] = acc * scale;
    }
}

/* ---- negative sampling + item-corpus gather --------------------------------------------------
 * Distribution and layout follow matching/pytorch/dataloaders/h5_generator.py:61-84 (np.random.choice(num_items,
 * size=(n, num_negs), replace=True); ignore_pos_items: uniform over the complement of the query's items),
 * :144-181 (hstack([pos, negs])), :23-28 + :49-58 (item_corpus[k][item_indexes], flattened).
 * The random STREAM is the build's own (numpy's MT19937 cannot be drawn in parallel): Philox4x32-10 as published
 * (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11; Random123 v1.14 known-answer
 * vectors are checked in tests/test_c_oracle.py), key = seed, counter = (element, attempt, 0). */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static int64_t orc_draw(uint64_t element, uint32_t attempt, uint64_t seed, uint64_t num_items) {
  const uint32_t ctr[4] = {(uint32_t)element, (uint32_t)(element >> 32), attempt, 0u};
  const uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
  uint32_t o[4];
  orc_philox4x32_10(ctr, key, o);
  const uint64_t r = ((uint64_t)o[1] << 32) | o[0];
  return (int64_t)(((unsigned __int128)r * num_items) >> 64);
}

void orc_negsample(int64_t num_items, int64_t rows, int32_t num_negs, uint64_t seed, uint64_t offset, const int64_t* pos,
                   const int64_t* query, const int64_t* excl_off, const int64_t* excl_items, int64_t* out) {
  const int width = num_negs + (pos ? 1 : 0);
  for (int64_t r = 0; r < rows; ++r) {
    if (pos) out[r * width] = pos[r];
    for (int j = 0; j < num_negs; ++j) {
      const uint64_t element = offset + (uint64_t)r * num_negs + j;
      int64_t lo = 0, hi = 0, item = 0;
      if (excl_off) { lo = excl_off[query[r]]; hi = excl_off[query[r] + 1]; }
      int found = 0;
      for (uint32_t a = 0; a < 64 && !found; ++a) {
        item = orc_draw(element, a, seed, (uint64_t)num_items);
        int hit = 0;
        for (int64_t k = lo; k < hi && !hit; ++k) hit = excl_items[k] == item;     /* membership, linear on purpose */
        found = !hit;
      }
      if (!found && hi - lo < num_items) {      /* 64 rejections: the k-th item of the complement, exactly */
        item = orc_draw(element, 64, seed, (uint64_t)(num_items - (hi - lo)));
        for (int64_t k = lo; k < hi && excl_items[k] <= item; ++k) ++item;
      }
      out[r * width + (pos ? 1 : 0) + j] = item;
    }
  }
}

/* dst[q, :] = src[index[q], :] for one column (v[item_indexes]); returns 1 when an index was out of range */
int orc_gather_rows(const void* src, void* dst, int64_t row_bytes, const int64_t* index, int64_t n_index, int64_t n_rows) {
  int bad = 0;
  for (int64_t q = 0; q < n_index; ++q) {
    int64_t row = index[q];
    if (row < 0 || row >= n_rows) { bad = 1; row = 0; }
    memcpy((char*)dst + q * row_bytes, (const char*)src + row * row_bytes, (size_t)row_bytes);
  }
  return bad;
}
