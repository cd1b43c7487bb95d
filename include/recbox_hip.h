/* recbox_hip.h -- C ABI of librecbox_hip.so (gfx950 / MI355X only).
 *
 * Drop-in boundary for the embedding-lookup + feature-interaction hot path of
 * reczoo/RecBox.  The reference has no FFI of its own (it is pure Python on
 * PyTorch): each entry point below replaces the ATen op
 * sequence that one reference nn.Module dispatches, and cites that module
 * (paths relative to /root/reference/recbox).  INTEGRATION.md shows the ctypes
 * stub a maintainer would add to call these from the reference's own layers.
 *
 * Conventions
 *   - plain C: pointers and sizes only, no C++ or torch types;
 *   - every pointer named d_* / inside rbx_field_t is a DEVICE pointer, every
 *     other pointer is a HOST pointer;
 *   - the caller owns all memory including workspaces (query *_workspace_size);
 *     the library allocates nothing persistent and keeps no global state but a
 *     thread-local error string;
 *   - kernels are enqueued on `stream` (a hipStream_t passed as void*; NULL is
 *     the default stream) and never synchronise; calls are re-entrant across
 *     streams;
 *   - return value: 0 on success, a negative rbx_status_t otherwise, message in
 *     rbx_last_error();
 *   - all floating point is IEEE fp32; ids are exact (bit-identical row choice).
 */
#ifndef RECBOX_HIP_H
#define RECBOX_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RBX_VERSION 124          /* 0.1.24: rbx_sort_chained (the id sort in 1 + passes launches); 0.1.23: rbx_fm_tier_c / rbx_fm_rezero_fusable, rbx_linear_fwd_bnstats / rbx_linear_dx_bnsums / rbx_batchnorm_*_from_partials removed (variants that lost their A/B); 0.1.22: rbx_cin_outer_*; 0.1.21: rbx_prelu_* / rbx_dropout / rbx_dice_* (csrc/rbx_act.hip); 0.1.20: rbx_fm_quad (rbx_fm_fwd's kernel for ids that are columns of one batch tensor); 0.1.19: rbx_seqblock_* (the row-local chains of a SASRec block as single passes); 0.1.18: rbx_fm_tier_c (the fused FM backward's sort-free tier C), rbx_opt_advance, rbx_opt_t.d_step_size, rbx_comm_bind_collectives takes ncclCommUserRank; 0.1.17: rbx_all_reduce / rbx_all_gather / rbx_comm_bind_collectives (every collective of the sharded step on the caller's stream); 0.1.16: rbx_linear_dx_scaled / rbx_linear_dwdb_scaled, rbx_rowscale_seq / rbx_seq_colsum; 0.1.15: rbx_linear_fwd_bnstats / rbx_linear_dx_bnsums / rbx_batchnorm_*_from_partials; 0.1.14: rbx_split_bf16 / _register / _unregister (f32 GEMM on the bf16 matrix cores); 0.1.13: rbx_fm_sort_phases, rbx_fm_bwd phases bits 3 / 4 (round 3: the fused FM backward in
                                  * two tiers); 0.1.11: rbx_fm_fwd grew d_prob; 0.1.10: rbx_field_t grew table_stride (round 2) */
#define RBX_MAX_FIELDS 64        /* fields per call */
#define RBX_NO_ID INT64_MIN      /* "no such id" for padding_idx / mask_id */

typedef enum {
  RBX_OK = 0,
  RBX_ERR_INVALID = -1,          /* bad argument (ValueError on the Python side) */
  RBX_ERR_LAUNCH = -2,           /* HIP launch / runtime failure (RuntimeError) */
  RBX_ERR_WORKSPACE = -3,        /* workspace too small */
  RBX_ERR_UNSUPPORTED = -4       /* NotImplementedError on the Python side */
} rbx_status_t;

/* dtype of an id / value column as it reaches the layer (int64,
 * int32, or float64 when the ranking loader hstacks the split; the kernels do
 * the reference's `.long()` / `.float()` casts in registers). */
typedef enum { RBX_I32 = 0, RBX_I64 = 1, RBX_F32 = 2, RBX_F64 = 3 } rbx_dtype_t;

typedef enum {
  RBX_FIELD_CATEGORICAL = 0,     /* nn.Embedding(vocab, dim)(ids.long()) */
  RBX_FIELD_NUMERIC = 1,         /* nn.Linear(1, dim, bias=False)(x.float().view(-1,1)) */
  RBX_FIELD_DENSE = 2            /* rechub DenseFeature: x.float() copied, dim == 1 */
} rbx_field_kind_t;

typedef enum {
  RBX_POOL_NONE = 0,             /* one id per sample */
  RBX_POOL_SUM = 1,              /* recbox MaskedSumPooling: plain sum over L */
  RBX_POOL_MEAN_VALUE = 2,       /* recbox MaskedAveragePooling: sum / (#rows with sum_d != 0 + eps) */
  RBX_POOL_MEAN_ID = 3,          /* rechub AveragePooling: sum_{id != mask_id} / (count + eps) */
  RBX_POOL_SUM_ID = 4,           /* rechub SumPooling with InputMask */
  RBX_POOL_CONCAT = 5            /* rechub ConcatPooling: keep [L, dim] */
} rbx_pool_t;

/* One feature of a multi-table lookup.  Output slot of sample b:
 *   d_out + b * out_stride_b + out_off   (dim floats; L*dim for POOL_CONCAT). */
typedef struct rbx_field {
  const void*  ids;              /* [B] or [B, seq_len] ids; numeric/dense: values */
  const float* table;            /* [vocab, dim] table; numeric: Linear weight as [dim]; dense: NULL */
  float*       grad;             /* backward only: dense grad, same shape as table (accumulated into) */
  int64_t      ids_stride_b;     /* element stride between samples */
  int64_t      ids_stride_l;     /* element stride inside a sequence */
  int64_t      vocab;
  int64_t      padding_idx;      /* nn.Embedding.padding_idx: row gets zero grad; RBX_NO_ID if unset */
  int64_t      mask_id;          /* *_ID pools: lookups equal to it get weight 0; RBX_NO_ID if unset */
  int64_t      out_off;          /* float offset of the slot inside an output row */
  int32_t      dim;
  int32_t      seq_len;          /* 1 unless a sequence feature */
  int32_t      ids_dtype;        /* rbx_dtype_t */
  int32_t      kind;             /* rbx_field_kind_t */
  int32_t      pool;             /* rbx_pool_t */
  float        eps;              /* MEAN pools: 1e-12 (recbox), 1e-16 (rechub), 1e-8 (RecBole) */
  int64_t      table_stride;     /* floats between consecutive table rows; 0 = dim (a contiguous [vocab, dim] table).
                                  * Only rbx_fm_fwd / rbx_fm_sort / rbx_fm_bwd / rbx_fm_rezero honour other values: the
                                  * embedding row and the dim-1 LR weight of an id may then share ONE padded row of a
                                  * packed [vocab, stride] storage (emb.table = base, lr.table = base + D, both with
                                  * table_stride = stride: one 128-byte line per lookup instead of two); every other
                                  * entry point rejects a stride != dim.  Gradients stay contiguous [vocab, dim]. */
} rbx_field_t;

const char* rbx_last_error(void);
int rbx_version(void);

/* ---- K1/K2: multi-table gather (+ fused sequence pooling) ------------------
 * Replaces, in ONE launch over all fields, the per-feature Python loop of
 *   core/pytorch/layers/embedding.py:116-138  (EmbeddingDictLayer.forward + callbacks),
 *   ranking/pytorch/layers/embeddings/feature_embedding.py:188-214 (+ dict2tensor :169-186),
 *   third_party/rechub/basic/layers.py:66-116 (EmbeddingLayer.forward, InputMask, pooling :176-230),
 * i.e. nn.Embedding / nn.Linear(1,D) / masked mean|sum pooling / stack|cat.
 * d_row_scale: [F, B] floats, written for MEAN pools (1/(count+eps)), read by the
 * backward; may be NULL when no MEAN pool is present.
 * d_status: optional int32 flag word, set non-zero when an id is out of range
 * (the reference raises IndexError); out-of-range lookups read as zero rows. */
int rbx_embed_fwd(const rbx_field_t* fields, int32_t n_fields, int64_t batch,
                  float* d_out, int64_t out_stride_b, float* d_row_scale,
                  int32_t* d_status, void* stream);

/* ---- K3: embedding backward = sorted, segmented, deterministic scatter-add ---
 * Replaces autograd's embedding_dense_backward (+ pooling / stack backward) of the
 * modules above.  Two phases so that the id sort can run early (ids are known in
 * the forward) on another stream:
 *   rbx_embed_sort   : builds (global row, lookup) pairs for every categorical
 *                      lookup, drops padding_idx / masked ids, radix-sorts them.
 *   rbx_embed_bwd    : segment-reduces d_out rows in sorted order and adds each
 *                      touched row ONCE into fields[f].grad (dense [vocab,dim]).
 *                      accumulate == 0: the caller pre-zeroed the grads, touched rows
 *                      are stored (no read); accumulate != 0: read-modify-write into
 *                      existing grads.  Numeric fields get
 *                      grad[d] += sum_b x_b * d_out[b, off+d].  Features that
 *                      share a table (same `table` pointer) are merged.
 * Results are run-to-run deterministic (no float atomics). */
size_t rbx_embed_bwd_workspace_size(const rbx_field_t* fields, int32_t n_fields, int64_t batch);
int rbx_embed_sort(const rbx_field_t* fields, int32_t n_fields, int64_t batch,
                   void* d_workspace, size_t workspace_bytes, int32_t* d_status, void* stream);
int rbx_embed_bwd(const rbx_field_t* fields, int32_t n_fields, int64_t batch,
                  const float* d_dout, int64_t out_stride_b, const float* d_row_scale,
                  int32_t accumulate, void* d_workspace, size_t workspace_bytes, void* stream);
/* Same, with one indirection: sample b reads its upstream gradient from row d_dout_index[b] of d_dout (int32, [batch])
 * instead of row b.  The owner side of the sharded exchange (rbx_shard_serve below) uses it: many received lookups
 * share one gradient row (the pooled history of a sample), which is then never expanded in memory. */
int rbx_embed_bwd_indexed(const rbx_field_t* fields, int32_t n_fields, int64_t batch,
                          const float* d_dout, int64_t out_stride_b, const int32_t* d_dout_index,
                          const float* d_row_scale, int32_t accumulate, void* d_workspace, size_t workspace_bytes,
                          void* stream);

/* ---- K4: InnerProductInteraction / rechub FM on a materialised [B,F,D] tensor ---
 * ranking/pytorch/layers/interactions/inner_product.py:40-56,
 * third_party/rechub/basic/layers.py:286-292.
 * mode 0: product_sum -> [B,1]; 1: bi_interaction -> [B,D];
 * mode 2: inner_product -> [B,F(F-1)/2]; 3: elementwise_product -> [B,F(F-1)/2,D].
 * emb_stride_b / demb_stride_b: floats between consecutive samples (>= F*D): the [F, D] block of a sample may
 * be the leading columns of a wider activation row (the flattened embedding layout of rechub's DeepFM). */
int rbx_interaction_fwd(const float* d_emb, int64_t emb_stride_b, int64_t batch, int32_t n_fields, int32_t dim,
                        int32_t mode, float* d_out, void* stream);
int rbx_interaction_bwd(const float* d_emb, int64_t emb_stride_b, const float* d_dout, int64_t batch, int32_t n_fields,
                        int32_t dim, int32_t mode, float* d_demb, int64_t demb_stride_b, void* stream);

/* ---- the pairing step of BilinearInteraction / BilinearInteractionV2
 * (ranking/pytorch/layers/interactions/bilinear_interaction.py:24-90).  out[b, p(i,j), :] = left * right[b, j, :] over
 * the F(F-1)/2 pairs i < j in triu order; per_pair == 0: left[B, F, D] = hidden (e_i W or e_i W_i: field_all /
 * field_each), indexed by i; per_pair == 1: left[B, P, D] = e_i W_p (field_interaction), indexed by the pair.  The
 * products with W run on rbx_linear_fwd.  Backward writes d_dleft (same shape as left) and d_dright[B, F, D]. */
int rbx_pairmul_fwd(const float* d_left, const float* d_right, int64_t batch, int32_t n_fields, int32_t dim,
                    int32_t per_pair, float* d_out, void* stream);
int rbx_pairmul_bwd(const float* d_left, const float* d_right, const float* d_dout, int64_t batch, int32_t n_fields,
                    int32_t dim, int32_t per_pair, float* d_dleft, float* d_dright, void* stream);

/* ---- fused FM model body: gather + LR + second-order interaction, [B,F,D] never stored ----
 * Replaces the op sequence feature_embedding.py:188-214 -> logistic_regression.py:30-35 ->
 * inner_product.py:41-48 -> factorization_machine.py:30-34 (forward and autograd backward).
 * emb[i] (dim D, all equal) and lr[i] (dim 1) describe the SAME feature i (same ids); only
 * one-id-per-sample categorical and numeric features can be fused.  Either array may be NULL
 * (lr == NULL: interaction only; emb == NULL: LogisticRegression only).
 *   d_logit[B] = sum_f lr_f + bias + 0.5 * sum_d[(sum_f e_fd)^2 - sum_f e_fd^2]
 *   d_prob[B]  = sigmoid(d_logit)  (optional, NULL to skip: the model's y_pred -- ranking_model.py output_activation --
 *                out of the same pass)
 *   d_sum[B,D] = sum_f e_f   (kept for the backward; may be NULL for inference)
 * Backward: row r of table f gets  dW[r] += sum_b g_b S_b - w_r * sum_b g_b  and
 * dW_lr[r] += sum_b g_b  over the samples b that looked r up (sorted, segmented,
 * deterministic); numeric weights and the bias are batch reductions.  Grads go into
 * emb[i].grad / lr[i].grad (dense; stored when accumulate == 0, added otherwise) and d_dbias[1].
 * phases: bit 0 = categorical tables (needs the sort), bit 1 = numeric weights + bias (does
 * not): a caller that sorts on another stream runs phase 2 first and phase 1 after the join; bit 2 = the
 * numeric-feature gradients and d_dbias are uninitialised memory: STORE them (no zero fill by the caller) instead of adding.
 * Two tiers (round 3).  Tables of up to 16 384 rows ("tier A", admitted smallest first while their gradients stay under
 * 4 MB; a rule over the tables alone, never the batch) skip the global sort: rbx_fm_sort leaves, per (feature, block of
 * 2048 samples), the block's (row, sample) pairs sorted in LDS and a bitmap of the rows present; rbx_fm_bwd sums
 * g_b S_b per (block, row), then ONE lane group per table row adds that row's block partials in ascending order and
 * WRITES the row -- every row of a tier-A table is written by every backward (zeros where nothing was looked up), so
 * these tables need no zero fill and rbx_fm_rezero skips them.  The other tables ("tier B") keep the sorted, segmented
 * path.  All sums have a fixed order: gradients are bit-identical run to run.  With bit 0 set, bit 3 (8) leaves tier A
 * out and bit 4 (16) leaves tier B out of this call: a caller with two streams runs the two tiers side by side (tier A
 * needs rbx_fm_sort_phases bits 0 and 2 done, tier B bits 0 and 1).  Calls with emb == NULL are one tier (B). */
int rbx_fm_fwd(const rbx_field_t* emb, const rbx_field_t* lr, int32_t n_fields, int64_t batch,
               const float* d_lr_bias, const float* d_extra, int32_t n_extra, int32_t extra_stride,
               int32_t extra_lr_off, const int32_t* d_extra_index, int64_t extra_rows, float* d_logit,
               float* d_prob, float* d_sum, int32_t* d_status, void* stream);
/* Rows of row-sharded tables arrive from their owners instead of being gathered locally:
 * d_extra[B, n_extra, extra_stride] holds, per (sample, table), the embedding row in floats [0, D)
 * and the dim-1 LR weight at float extra_lr_off (-1: none); they take part in S, Q and the LR sum
 * exactly like local features.  rbx_fm_extra_bwd writes their gradient block in the same packed
 * layout ([g (S - e) | g at the LR slot | 0]); it travels back to the owners (recbox_amd/sharded.py).
 * dim == 0 there means "LR weights only".
 * d_extra_index (optional, [B, n_extra] int32): d_extra is then the exchange buffer itself,
 * [extra_rows, extra_stride], and row (b, t) sits at wire slot d_extra_index[b, t] (rbx_route); a slot
 * outside [0, extra_rows) is a lookup that found no room on the wire: zero row, no gradient.  The
 * indexed rbx_fm_extra_bwd writes only the slots that are referenced: the caller zero-fills d_dextra. */
int rbx_fm_extra_bwd(const float* d_dlogit, const float* d_sum, const float* d_extra, int64_t batch,
                     int32_t n_extra, int32_t dim, int32_t extra_stride, int32_t extra_lr_off,
                     const int32_t* d_extra_index, int64_t extra_rows, float* d_dextra, void* stream);

/* ---- C1: routing of the padded, sync-free exchange of row-sharded tables (no reference precedent:
 * SURVEY.md 2.1 / 8e; layout in recbox_amd/sharded.py).  tables[t] describes the id column of sharded
 * table t (only ids / ids_stride_b / ids_dtype are read: the same strided, typed columns rbx_fm_fwd
 * takes); lookup i = b * n_tables + t.  owner = id mod world; d_base[world, n_tables] = first row of table t inside
 * the owner's packed weight; the row number base[owner][t] + id div world is written to wire slot
 * d_slot[i] = owner * capacity + (count of earlier lookups with that owner) of d_send[world * capacity]
 * (empty slots = -1).  Lookups that do not fit get slot world * capacity and set *d_overflow = 1
 * (never cleared here).  Stable and deterministic; no host sync. */
size_t rbx_route_workspace_size(int64_t n_lookups, int32_t world);
int rbx_route(const rbx_field_t* tables, int32_t n_tables, int64_t batch, int32_t world, int64_t capacity,
              const int64_t* d_base, int64_t* d_send, int32_t* d_slot, uint8_t* d_overflow, void* d_workspace,
              size_t workspace_bytes, void* stream);
/* The same with 32-bit row numbers on the wire (d_send [world * capacity] int32, empty slots = -1): the row numbers of one
 * shard fit 31 bits whenever rbx_embed_fwd can address the shard at all, and the id exchange is half as long. */
int rbx_route32(const rbx_field_t* tables, int32_t n_tables, int64_t batch, int32_t world, int64_t capacity,
                const int64_t* d_base, int32_t* d_send, int32_t* d_slot, uint8_t* d_overflow, void* d_workspace,
                size_t workspace_bytes, void* stream);

/* ---- C1 (second generation): ONE exchange each way for row-sharded tables, pooled lookups reduced at the owner
 * (csrc/rbx_shard.hip; no reference precedent -- the reference's only parallelism is nn.DataParallel / DDP, SURVEY.md
 * 2.1 -- this is SURVEY.md 5.8 / 8e and BASELINE.json configs 3 and 4).  Per sample: n_rows single-row lookups
 * (one-hot features and the columns of a pooling='concat' sequence: third_party/rechub/basic/layers.py:72,98-107) and
 * n_pool <= 1 id-masked mean / sum pooled sequence (layers.py:135-148,187-210; YoutubeDNN's history,
 * models/matching/youtube_dnn.py:46-56).  owner = id mod world, local row = d_base[owner][lookup] + id div world.
 * Static wire format per (requester, owner) pair:
 *   int32 chunk of rbx_shard_int_chunk() values: [cap_pool row numbers grouped by sample | n_pool * (batch + 1)
 *     offsets | cap_rows row numbers (-1 = empty)],
 *   fp32 chunk of rbx_shard_float_rows() rows x dim: [n_pool * batch partial sums (gradients) | cap_rows rows].
 * The all-to-alls between the calls are the caller's (recbox_amd/comm.py over RCCL, or rbx_all_to_all).
 *   rbx_shard_route        requester: ids -> d_send[world][int chunk]; d_slot[batch][n_rows] = wire slot of each
 *                          single-row lookup (owner * cap_rows + rank; world * cap_rows = did not fit / out of range);
 *                          d_inv[batch] = 1 / (valid ids + eps) (mean) or 1 (sum).  row_fields[t] / pool_field: ids,
 *                          strides, ids_dtype, vocab (+ seq_len, mask_id, pool, eps of the pooled one) are read.
 *                          Lookups beyond a capacity set *d_overflow; ids outside [0, vocab) set bit 0 of *d_status
 *                          (the reference raises IndexError) and are treated as absent.
 *   rbx_shard_serve        owner: d_recv[world][int chunk] -> d_back[world][float rows][dim] (one partial sum per
 *                          (source, sample), one row per occupied single-row slot), plus for the backward
 *                          d_keys[world * (cap_pool + cap_rows)] (local row of every received lookup, -1 = none) and
 *                          d_src (row of the gradient buffer [world * float rows, dim] holding its upstream gradient):
 *                          rbx_embed_sort over d_keys + rbx_embed_bwd_indexed(d_dout_index = d_src) is the owner's
 *                          deterministic scatter-add.  A row number >= n_local_rows sets bit 1 of *d_status.
 *   rbx_shard_combine_fwd  requester: places row (b, t) and the scaled sum of the world partial sums of sample b at
 *                          d_out + b * out_stride_b + col_off[lookup] (HOST array, n_rows + n_pool entries; the
 *                          pooled lookup is the last one): the slots of the layer's [batch, width] output block.
 *   rbx_shard_combine_bwd  requester: the upstream gradient of those slots -> d_gsend[world][float rows][dim]. */
typedef struct rbx_shard_geom {
  int32_t world;
  int32_t dim;               /* floats per row, multiple of 4 */
  int32_t n_rows;            /* single-row lookups per sample */
  int32_t n_pool;            /* pooled lookups per sample: 0 or 1 */
  int64_t batch;             /* samples per rank (equal on every rank: static wire sizes) */
  int64_t cap_rows;          /* wire slots per (requester, owner) pair for single-row lookups */
  int64_t cap_pool;          /* ... for the ids of the pooled lookup */
} rbx_shard_geom_t;
size_t rbx_shard_int_chunk(const rbx_shard_geom_t* geom);
size_t rbx_shard_float_rows(const rbx_shard_geom_t* geom);
size_t rbx_shard_route_workspace_size(const rbx_shard_geom_t* geom, int32_t pool_seq_len);
int rbx_shard_route(const rbx_shard_geom_t* geom, const rbx_field_t* row_fields, const rbx_field_t* pool_field,
                    const int64_t* d_base, int32_t* d_send, int32_t* d_slot, float* d_inv, uint8_t* d_overflow,
                    int32_t* d_status, void* d_workspace, size_t workspace_bytes, void* stream);
int rbx_shard_serve(const rbx_shard_geom_t* geom, const int32_t* d_recv, const float* d_weight, int64_t n_local_rows,
                    float* d_back, int32_t* d_keys, int32_t* d_src, int32_t* d_status, void* stream);
int rbx_shard_combine_fwd(const rbx_shard_geom_t* geom, const float* d_back, const int32_t* d_slot, const float* d_inv,
                          float* d_out, int64_t out_stride_b, const int64_t* col_off, void* stream);
int rbx_shard_combine_bwd(const rbx_shard_geom_t* geom, const float* d_dout, int64_t dout_stride_b,
                          const int64_t* col_off, const int32_t* d_slot, const float* d_inv, float* d_gsend,
                          void* stream);
size_t rbx_fm_bwd_workspace_size(const rbx_field_t* emb, const rbx_field_t* lr, int32_t n_fields, int64_t batch);
int rbx_fm_sort(const rbx_field_t* emb, const rbx_field_t* lr, int32_t n_fields, int64_t batch,
                void* d_workspace, size_t workspace_bytes, int32_t* d_status, void* stream);
/* rbx_fm_sort in pieces, for a caller that wants to order other work between them.  phases: bit 0 = id columns of any
 * dtype / stride -> the workspace's int32 [feature][batch] matrix, range-checked (d_status) -- both tiers read it; bit 1 =
 * tier B: (row, sample) pairs + segmented radix sort; bit 2 = tier A: the per-block sorts.  Bits 1 and 2 need bit 0 done
 * on the same workspace (same call or an earlier one on the same stream).  rbx_fm_sort == all three. */
int rbx_fm_sort_phases(const rbx_field_t* emb, const rbx_field_t* lr, int32_t n_fields, int64_t batch,
                       void* d_workspace, size_t workspace_bytes, int32_t* d_status, int32_t phases, void* stream);
int rbx_fm_bwd(const rbx_field_t* emb, const rbx_field_t* lr, int32_t n_fields, int64_t batch,
               const float* d_dlogit, const float* d_sum, float* d_dbias, int32_t accumulate,
               int32_t phases, void* d_workspace, size_t workspace_bytes, void* stream);
/* For a caller that keeps PERSISTENT dense gradient buffers (the reference's autograd allocates and zero-fills a new
 * [V, D] gradient per table and step: torch/nn/functional.py embedding backward; 379 MB per step at the Criteo
 * shape, of which a batch touches 36 MB): with the workspace of the PREVIOUS rbx_fm_sort (same fields, same batch)
 * and emb[i].grad / lr[i].grad = the buffers that step's rbx_fm_bwd stored into, write zeros to exactly the rows it
 * touched.  Call it before the next rbx_fm_sort reuses the workspace. */
/* (Round 4's sort-free third tier -- rbx_fm_tier_c, rbx_fm_rezero_fusable, rbx_fm_sort_phases bit 3 -- measured a draw on uniform
 * ids and a loss on skewed ones in two rounds running and was removed in round 5: docs/history/.) */
/* Round 5: rbx_fm_fwd has a second kernel (csrc/rbx_fm_quad.hip) for the wire format of the reference's ranking loader --
 * every feature's ids / values the columns, in feature order, of ONE row-major batch tensor of one dtype
 * (ranking/pytorch/dataloaders/h5_dataloader.py:36-47, ranking_model.py:106-116), dim 16, both field arrays given, the
 * tables within 4 GiB of each other, no extra rows.  Same results as the general kernel, operation for operation.  Taken
 * automatically; rbx_fm_quad(0) forces the general kernel (A/B measurements, tests), a negative value only reads; returns
 * the previous setting (RBX_FM_QUAD=0 in the environment: off from the start). */
int rbx_fm_quad(int32_t enable);
/* Round 5: the id sort behind every backward (rbx_embed_sort, rbx_fm_sort, ...) runs as 1 + passes launches when no table
 * group of the call has more than 64 sort tiles (131 072 lookups): the pairs kernel counts every pass's digit per tile, and a
 * scatter workgroup takes the start of its runs from those counts plus what the tiles in front of it publish
 * (csrc/rbx_embed_bwd.hip, BwdPlan::chained).  rbx_sort_chained(0) forces the histogram / scan / scatter launches per pass
 * for every call (A/B measurements, tests; the results are the same pairs in the same order); a negative value only reads;
 * returns the previous setting.  The workspace sizes depend on the setting: size and call with the same one. */
int rbx_sort_chained(int32_t enable);
int rbx_fm_rezero(const rbx_field_t* emb, const rbx_field_t* lr, int32_t n_fields, int64_t batch,
                  void* d_workspace, size_t workspace_bytes, void* stream);
/* Two lookups over the SAME id tensors with the same table layout (the embedding tables of FeatureEmbedding and the dim-1
 * tables of LogisticRegression over one batch: feature_embedding.py + logistic_regression.py:30-35) sort identical
 * (row, sample) pairs.  Given the descriptors of a sort that is already in d_src_workspace (src_is_fm = 0: src_a is an
 * rbx_embed_sort field array, src_b unused; 1: src_a / src_b are rbx_fm_sort's emb / lr arrays) and those of the lookup
 * that needs one (dst_*), copy the sorted pairs instead of sorting again.  Returns RBX_ERR_UNSUPPORTED -- nothing was
 * launched, sort as usual -- when the two do not sort the same pairs. */
int rbx_sort_share(const rbx_field_t* src_a, const rbx_field_t* src_b, int32_t src_n, int32_t src_is_fm,
                   const void* d_src_workspace, const rbx_field_t* dst_a, const rbx_field_t* dst_b, int32_t dst_n,
                   int32_t dst_is_fm, void* d_dst_workspace, size_t dst_workspace_bytes, int64_t batch, void* stream);
/* The same for the generic lookup: the rows named by the previous rbx_embed_sort on this workspace (fields[i].grad =
 * the persistent buffers its rbx_embed_bwd stored into).  Only for tables that get their gradient from that lookup alone. */
int rbx_embed_rezero(const rbx_field_t* fields, int32_t n_fields, int64_t batch, void* d_workspace,
                     size_t workspace_bytes, void* stream);

/* ---- opt-in sparse-row optimiser step (SURVEY.md 8b "or, opt-in, a sparse-row update path"; 2.2 K3).  The reference
 * runs a dense torch.optim step over dense [V, D] gradients (ranking/pytorch/models/ranking_model.py:191-197,
 * matching/pytorch/models/match_model.py:194-199).  After rbx_embed_sort + rbx_embed_bwd (rbx_fm_sort + rbx_fm_bwd) of a
 * step, the workspace still holds the sorted (row, lookup) pairs: the head of every run of equal rows names one touched row,
 * whose summed gradient sits in fields[i].grad.  These calls apply ONE optimiser step to exactly those rows of
 * fields[i].table (written in place) and of the state tensors -- same shape as the gradient, d_state1[i] / d_state2[i] per
 * feature i (HOST arrays of DEVICE pointers; features that share a table pass the same pointers):
 *   RBX_OPT_SGD      w -= lr (g + weight_decay w)
 *   RBX_OPT_ADAGRAD  state1 += g^2; w -= lr g / (sqrt(state1) + eps)          (torch.optim.Adagrad's sparse branch; the
 *                    caller folds lr_decay into lr)
 *   RBX_OPT_ADAM     state1 = beta1 state1 + (1 - beta1) g; state2 = beta2 state2 + (1 - beta2) g^2;
 *                    w -= lr state1 / (sqrt(state2) + eps)   with lr = lr0 sqrt(1 - beta2^t) / (1 - beta1^t) folded in by
 *                    the caller (torch.optim.SparseAdam: the moments of untouched rows do not decay)
 * Rows the batch did not touch, padding_idx rows and masked lookups are neither read nor written.  The tier-A tables of
 * the fused FM body (every row written by every backward) count a row as touched when some block's presence bitmap has it.
 * Call between the backward and the next sort on that workspace. */
typedef enum { RBX_OPT_SGD = 0, RBX_OPT_ADAGRAD = 1, RBX_OPT_ADAM = 2 } rbx_opt_kind_t;
typedef struct rbx_opt {
  int32_t kind;              /* rbx_opt_kind_t */
  float   lr;                /* effective step size (bias correction / decay folded in) */
  float   beta1, beta2;      /* Adam */
  float   eps;
  float   weight_decay;      /* L2 on the touched rows: g += weight_decay * w (0 for torch's sparse rules) */
  const float* d_step_size;  /* NULL, or a DEVICE float that overrides lr: a step captured into a hipGraph bakes every
                              * by-value argument in, so a rule whose step size depends on the step count (Adam's bias
                              * correction, Adagrad's lr_decay) reads it from memory that rbx_opt_advance updates in-graph */
} rbx_opt_t;
/* t = *d_t + 1 -> *d_t; *d_step_size = the rule's effective step size at step t: SGD lr; Adagrad lr / (1 + (t - 1) lr_decay);
 * Adam lr sqrt(1 - beta2^t) / (1 - beta1^t).  One tiny launch per optimiser step; d_t is a float counter (exact to 2^24). */
int rbx_opt_advance(int32_t kind, float lr, float beta1, float beta2, float lr_decay, float* d_t, float* d_step_size,
                    void* stream);
int rbx_embed_sparse_update(const rbx_field_t* fields, int32_t n_fields, int64_t batch, const void* d_workspace,
                            size_t workspace_bytes, const rbx_opt_t* opt, float* const* d_state1, float* const* d_state2,
                            void* stream);
int rbx_fm_sparse_update(const rbx_field_t* emb, const rbx_field_t* lr, int32_t n_fields, int64_t batch,
                         const void* d_workspace, size_t workspace_bytes, const rbx_opt_t* opt,
                         float* const* d_emb_state1, float* const* d_emb_state2, float* const* d_lr_state1,
                         float* const* d_lr_state2, void* stream);

/* ---- C2: the exchange itself on the caller's stream.  The reference has no sharded exchange (nn.DataParallel / DDP
 * only); recbox_amd/comm.py uses torch.distributed's all_to_all_single by default, which runs on RCCL's
 * own stream behind two event joins.  rbx_all_to_all is the same grouped ncclSend / ncclRecv sequence enqueued on
 * `stream`: block p of d_send (bytes_per_peer bytes) goes to rank p, block p of d_recv comes from rank p.  `comm` is
 * the ncclComm_t of the process group (ProcessGroupNCCL._comm_ptr()); rbx_comm_bind hands over the RCCL entry points
 * of the library the process has already loaded (ncclGroupStart, ncclGroupEnd, ncclSend, ncclRecv, ncclGetErrorString). */
int rbx_comm_bind(void* fn_group_start, void* fn_group_end, void* fn_send, void* fn_recv, void* fn_error_string);
int rbx_all_to_all(void* comm, const void* d_send, void* d_recv, size_t bytes_per_peer, int32_t world, void* stream);
/* The other two collectives of the sharded step the same way (round 4): the flat all-reduce of the replicated /
 * data-parallel gradients (the reference's counterpart is nn.DataParallel's gradient gather,
 * third_party/rechub/trainers/ctr_trainer.py:41-43) and the all-gather of a synchronised BatchNorm's statistics, as
 * ncclAllReduce / ncclAllGather on `stream` -- with all three on the step's own stream no collective crosses to RCCL's
 * stream, and the whole step (collectives included) is one hipGraph capture.  dtype: RBX_I32 / RBX_I64 / RBX_F32 /
 * RBX_F64; op: RBX_REDUCE_*; in place when d_send == d_recv.  rbx_comm_bind_collectives hands over ncclAllReduce and
 * (optional, NULL) ncclAllGather and ncclCommUserRank of the loaded RCCL; with the latter rbx_all_to_all moves the block
 * a rank keeps for itself by a device-to-device copy instead of RCCL's self send / recv. */
#define RBX_REDUCE_SUM 0
#define RBX_REDUCE_MAX 2
#define RBX_REDUCE_MIN 3
int rbx_comm_bind_collectives(void* fn_all_reduce, void* fn_all_gather, void* fn_comm_user_rank);
int rbx_all_reduce(void* comm, const void* d_send, void* d_recv, size_t count, int32_t dtype, int32_t op, void* stream);
int rbx_all_gather(void* comm, const void* d_send, void* d_recv, size_t bytes_per_rank, void* stream);

/* ---- K5: two-tower scoring (third_party/rechub/models/matching/dssm.py:48,57,65,
 * youtube_dnn.py:47-48,56,65,70).  l2norm = F.normalize(x, p=2, dim=-1, eps): y = x / max(||x||, eps);
 * d_inv[rows] keeps 1/max(||x||,eps) (negative when the clamp was active) for the backward.
 * pairdot: out[b,n] = scale * <u[b,:], v[b,n,:]>  (DSSM: n_cand = 1; YoutubeDNN: 1 + n_neg, scale = 1/T). */
int rbx_l2norm_fwd(const float* d_x, int64_t rows, int32_t dim, float eps, float* d_y, float* d_inv, void* stream);
/* The same over rows that sit inside a wider block: row r = (r / inner, r % inner) at d_x + (r / inner) * outer_stride +
 * (r % inner) * dim -- the [B, 1 + n_neg, D] item rows of YoutubeDNN read where the gather left them, behind the user
 * columns of one [B, width] block (youtube_dnn.py:52-70), without a contiguous copy.  d_y [rows, dim] is contiguous. */
int rbx_l2norm_fwd_strided(const float* d_x, int64_t inner, int64_t outer_stride, int64_t rows, int32_t dim, float eps,
                           float* d_y, float* d_inv, void* stream);
/* F.normalize of the candidate rows and their inner product with the (already normalised) user vector in ONE pass
 * (youtube_dnn.py:56,65,70): out[b, n] = scale * <u[b], v[b, n]> / max(|v[b, n]|, eps); d_inv[b, n] keeps 1 / max(|v|, eps)
 * (negative when the clamp was active).  The normalised rows are never materialised.  v rows at d_v + b * v_outer_stride +
 * n * dim (read where the gather left them); backward: d_du[b] = sum_n scale dout v_hat, d_dv rows (the projection of
 * scale dout u / |v| orthogonal to v_hat; plain when clamped) at d_dv + b * dv_outer_stride + n * dim. */
int rbx_cosdot_fwd(const float* d_u, const float* d_v, int64_t v_outer_stride, int64_t batch, int32_t n_cand, int32_t dim,
                   float eps, float scale, float* d_out, float* d_inv, void* stream);
int rbx_cosdot_bwd(const float* d_u, const float* d_v, int64_t v_outer_stride, const float* d_inv, const float* d_dout,
                   int64_t batch, int32_t n_cand, int32_t dim, float scale, float* d_du, float* d_dv,
                   int64_t dv_outer_stride, void* stream);
int rbx_l2norm_bwd(const float* d_y, const float* d_inv, const float* d_dy, int64_t rows, int32_t dim,
                   float* d_dx, void* stream);
int rbx_pairdot_fwd(const float* d_u, const float* d_v, int64_t batch, int32_t n_cand, int32_t dim, float scale,
                    float* d_out, void* stream);
int rbx_pairdot_bwd(const float* d_u, const float* d_v, const float* d_dout, int64_t batch, int32_t n_cand,
                    int32_t dim, float scale, float* d_du, float* d_dv, void* stream);

/* ---- BatchNorm1d of the dense towers (third_party/rechub/basic/layers.py:255-263: Linear -> BatchNorm1d ->
 * activation -> Dropout after every layer; optional in core/pytorch/layers/mlp.py:25-37 and
 * ranking/pytorch/layers/blocks/mlp_block.py:42-58).  x [rows, cols] row-major, statistics per column.
 * training != 0: batch statistics (biased variance normalises, the running statistics receive
 * (1 - momentum) * old + momentum * {mean, unbiased variance} when the pointers are given); training == 0:
 * running statistics.  d_mean / d_rstd [cols] are outputs kept for the backward.  relu != 0 applies ReLU to y;
 * the backward then takes that y as d_y_relu (mask y > 0) -- NULL when no activation was fused.
 * Backward: d_dgamma = sum dy * xhat, d_dbeta = sum dy (always written, also used as scratch),
 * d_dx = gamma * rstd * (dy - dbeta/M - xhat * dgamma/M) (training) or gamma * rstd * dy (eval); NULL skips it.
 * Reductions are two-stage in a fixed order (Welford partials merged with Chan's formula): deterministic. */
size_t rbx_batchnorm_workspace_size(int64_t rows, int32_t cols);
int rbx_batchnorm_fwd(const float* d_x, int64_t rows, int32_t cols, const float* d_gamma, const float* d_beta, float eps,
                      int32_t training, float momentum, float* d_running_mean, float* d_running_var, int32_t relu,
                      float* d_mean, float* d_rstd, float* d_y, void* d_workspace, size_t workspace_bytes, void* stream);
int rbx_batchnorm_bwd(const float* d_x, const float* d_dy, const float* d_y_relu, int64_t rows, int32_t cols,
                      const float* d_gamma, const float* d_mean, const float* d_rstd, int32_t training, float* d_dx,
                      float* d_dgamma, float* d_dbeta, void* d_workspace, size_t workspace_bytes, void* stream);

/* nn.PReLU behind the BatchNorm of a tower layer (rechub MLP with activation="prelu": Linear -> BatchNorm1d -> PReLU ->
 * Dropout, third_party/rechub/basic/layers.py:255-263, activation.py:44-45; DSSM's towers in BASELINE cfg 1) in the same
 * passes: y = z > 0 ? z : a z on the normalised value z, a = d_slope[0] (slope_n == 1: nn.PReLU()) or one per column
 * (slope_n == cols).  Backward rebuilds z from x (its sign is the mask -- y's is not when a <= 0):
 * dz = z > 0 ? dy : a dy, d_dslope_cols[c] = sum dy z over z <= 0 (sum it over c for the single-parameter form). */
int rbx_batchnorm_prelu_fwd(const float* d_x, int64_t rows, int32_t cols, const float* d_gamma, const float* d_beta,
                            const float* d_slope, int32_t slope_n, float eps, int32_t training, float momentum,
                            float* d_running_mean, float* d_running_var, float* d_mean, float* d_rstd, float* d_y,
                            void* d_workspace, size_t workspace_bytes, void* stream);
int rbx_batchnorm_prelu_bwd(const float* d_x, const float* d_dy, int64_t rows, int32_t cols, const float* d_gamma,
                            const float* d_beta, const float* d_slope, int32_t slope_n, const float* d_mean,
                            const float* d_rstd, int32_t training, float* d_dx, float* d_dgamma, float* d_dbeta,
                            float* d_dslope_cols, void* d_workspace, size_t workspace_bytes, void* stream);
/* The same in pieces, for a SYNCHRONISED BatchNorm over the ranks of a data-parallel job (torch.nn.SyncBatchNorm: RecBole's
 * DDP path converts every BatchNorm of the model, third_party/recbole/trainer/trainer.py:60-64; rechub's nn.DataParallel
 * keeps per-replica statistics, ctr_trainer.py:43 -- the default here).  The caller's collectives sit between the calls:
 *   rbx_batchnorm_stats       d_raw[3, cols] = this rank's (count, mean, M2 = sum (x - mean)^2) per column;
 *                             all-gather, merge (Chan), derive mean / rstd and the running statistics;
 *   rbx_batchnorm_apply       y = (x - mean) rstd gamma + beta (+ ReLU) with the GLOBAL d_mean / d_rstd;
 *   rbx_batchnorm_bwd_reduce  this rank's d_dgamma = sum dy xhat, d_dbeta = sum dy (the parameter gradients: the job's
 *                             gradient all-reduce sums them over the ranks like every other replicated parameter);
 *                             all-reduce a COPY of the two for the next call;
 *   rbx_batchnorm_bwd_dx      dx = gamma rstd (dy - dbeta / N - xhat dgamma / N) with the GLOBAL sums and
 *                             N = total_rows = rows of all ranks together. */
int rbx_batchnorm_stats(const float* d_x, int64_t rows, int32_t cols, float* d_raw, void* d_workspace,
                        size_t workspace_bytes, void* stream);
int rbx_batchnorm_apply(const float* d_x, int64_t rows, int32_t cols, const float* d_gamma, const float* d_beta,
                        const float* d_mean, const float* d_rstd, int32_t relu, float* d_y, void* stream);
int rbx_batchnorm_bwd_reduce(const float* d_x, const float* d_dy, const float* d_y_relu, int64_t rows, int32_t cols,
                             const float* d_mean, const float* d_rstd, float* d_dgamma, float* d_dbeta, void* d_workspace,
                             size_t workspace_bytes, void* stream);
int rbx_batchnorm_bwd_dx(const float* d_x, const float* d_dy, const float* d_y_relu, int64_t rows, int32_t cols,
                         const float* d_gamma, const float* d_mean, const float* d_rstd, const float* d_dgamma,
                         const float* d_dbeta, int64_t total_rows, float* d_dx, void* stream);

/* ---- xDeepFM's CIN outer product (ranking/pytorch/layers/interactions/compressed_interaction_net.py:35-48; csrc/rbx_cin.hip):
 * d_z[(b, d), h * m + j] = x0[b, h, d] * xk[(b, d), j], rows = batch * dim, in the A layout of the GEMM that runs the 1x1
 * Conv1d over the channel axis.  d_x0 [batch, n_fields, dim] (the embedding layer's output as it is); d_xk [batch * dim, m]
 * (the previous layer's GEMM output as it is) or NULL for the first layer (X_k = X_0, m == n_fields).  Backward: d_dx0
 * [batch, n_fields, dim] (both uses of X_0 when d_xk is NULL) and d_dxk [batch * dim, m]; NULL skips either. */
int rbx_cin_outer_fwd(const float* d_x0, const float* d_xk, int64_t batch, int32_t n_fields, int32_t m, int32_t dim,
                      float* d_z, void* stream);
int rbx_cin_outer_bwd(const float* d_x0, const float* d_xk, const float* d_dz, int64_t batch, int32_t n_fields, int32_t m,
                      int32_t dim, float* d_dx0, float* d_dxk, void* stream);

/* ---- the towers' activations that are not fused into a GEMM epilogue or a BatchNorm pass (csrc/rbx_act.hip; round 5).
 * x, y, dy, dx: [rows, cols] row-major f32.  rbx_act_workspace_size: bytes for the column reductions of the calls that take a
 * workspace (partials per block of 256 rows, merged in block order: results repeat bit for bit).
 * nn.PReLU standing alone (core/pytorch/layers/mlp.py:25-37, ranking/pytorch/layers/blocks/mlp_block.py:42-58 with
 *   hidden_activations="PReLU" and no BatchNorm in front): n_slope = 1 or cols; backward dx = dy (x > 0 ? 1 : a) and
 *   d_dslope[n_slope] = sum dy x [x <= 0] (NULL skips either).
 * nn.Dropout(p), training (the same towers' dropout_rates, third_party/rechub/basic/layers.py:255-263): y = x keep / (1 - p)
 *   with keep(i) a counter-based function of (seed + *d_seed_add, i) (Philox4x32-10) -- the SAME call on dy is the backward,
 *   nothing is stored; not torch's random stream: equal in distribution to the reference, not bit for bit.
 * Dice (core/pytorch/layers/activations.py:23-33): p = sigmoid((x - mean) rstd) with the statistics of a non-affine
 *   BatchNorm1d (training: batch statistics, biased variance, running statistics updated with `momentum` when given;
 *   evaluation: the running statistics), y = p x + alpha (1 - p) x, alpha [cols].  d_mean / d_rstd [cols] are outputs kept for
 *   the backward, which returns dx (through the statistics too, in training) and d_dalpha[cols] (NULL skips either). */
size_t rbx_act_workspace_size(int64_t rows, int32_t cols);
int rbx_prelu_fwd(const float* d_x, int64_t rows, int32_t cols, const float* d_slope, int32_t n_slope, float* d_y,
                  void* stream);
int rbx_prelu_bwd(const float* d_x, const float* d_dy, int64_t rows, int32_t cols, const float* d_slope, int32_t n_slope,
                  float* d_dx, float* d_dslope, void* d_workspace, size_t workspace_bytes, void* stream);
int rbx_dropout(const float* d_x, int64_t n, float p, uint64_t seed, const uint64_t* d_seed_add, float* d_y, void* stream);
int rbx_dice_fwd(const float* d_x, int64_t rows, int32_t cols, const float* d_alpha, float eps, int32_t training,
                 float momentum, float* d_running_mean, float* d_running_var, float* d_mean, float* d_rstd, float* d_y,
                 void* d_workspace, size_t workspace_bytes, void* stream);
int rbx_dice_bwd(const float* d_x, const float* d_dy, int64_t rows, int32_t cols, const float* d_alpha, const float* d_mean,
                 const float* d_rstd, int32_t training, float* d_dx, float* d_dalpha, void* d_workspace,
                 size_t workspace_bytes, void* stream);

/* ---- LayerNorm over the last dimension (the five nn.LayerNorm(D, eps=1e-8) of a SASRec block stack,
 * third_party/rechub/models/matching/sasrec.py:52-63,81-94).  x [rows, dim] contiguous; biased variance, eps inside
 * the square root (torch semantics); d_mean / d_rstd [rows] are kept for the backward.  Backward:
 * dx = rstd * (g - mean_d(g) - xhat * mean_d(g * xhat)) with g = dy * gamma (NULL skips it);
 * d_dgamma = sum_rows dy * xhat, d_dbeta = sum_rows dy (both or neither; two-stage fixed-order reduction). */
int rbx_layernorm_fwd(const float* d_x, int64_t rows, int32_t dim, const float* d_gamma, const float* d_beta, float eps,
                      float* d_mean, float* d_rstd, float* d_y, void* stream);
size_t rbx_layernorm_bwd_workspace_size(int64_t rows, int32_t dim);
int rbx_layernorm_bwd(const float* d_x, const float* d_dy, int64_t rows, int32_t dim, const float* d_gamma,
                      const float* d_mean, const float* d_rstd, float* d_dx, float* d_dgamma, float* d_dbeta,
                      void* d_workspace, size_t workspace_bytes, void* stream);

/* ---- K7: candidate scoring without materialising the candidate embeddings
 * (third_party/rechub/models/matching/sasrec.py:98-105: pos/neg logits = (seq_output * item_emb(ids)).sum(-1);
 * the [rows, 1 + n] sampled-softmax logits consumed by core/pytorch/losses/softmax_crossentropy_loss.py:14-22).
 *   d_out[r, c] = scale * < d_x[r, :], table_c[ids_c[r], :] >
 * cands[i] is one candidate set: ids ([rows] or [rows, seq_len] through ids_stride_b / ids_stride_l, any id
 * dtype), table/vocab/dim (all sets share dim), seq_len = candidates per row, out_off = first output column;
 * the sets together must cover columns [0, n_out) exactly once (n_out <= 256).  Out-of-range ids score 0 and
 * raise d_status.  Backward: dense dW into cands[i].grad through the sorted segmented scatter-add
 * (contribution scale * g[r,c] * x[r,:]; rows equal to padding_idx receive no gradient; grad == NULL =
 * frozen), and d_dx[r, :] = scale * sum_c g[r,c] * row (skipped when d_dx is NULL).  rbx_gatherdot_sort
 * depends on the ids only and may run on another stream during the forward. */
int rbx_gatherdot_fwd(const rbx_field_t* cands, int32_t n_cands, int64_t rows, const float* d_x, int64_t x_stride,
                      float scale, float* d_out, int32_t* d_status, void* stream);
size_t rbx_gatherdot_bwd_workspace_size(const rbx_field_t* cands, int32_t n_cands, int64_t rows);
int rbx_gatherdot_sort(const rbx_field_t* cands, int32_t n_cands, int64_t rows, void* d_workspace,
                       size_t workspace_bytes, int32_t* d_status, void* stream);
int rbx_gatherdot_bwd(const rbx_field_t* cands, int32_t n_cands, int64_t rows, const float* d_x, int64_t x_stride,
                      const float* d_dout, float scale, float* d_dx, int64_t dx_stride, int32_t accumulate,
                      void* d_workspace, size_t workspace_bytes, void* stream);

/* ---- the two-tower loader's per-epoch negative sampling and per-batch item-corpus gather
 * (matching/pytorch/dataloaders/h5_generator.py:61-84 sampling_block, :144-181 negative_sampling,
 * :23-28 TrainDataset.__getitem__, :49-58 collate_fn).
 * rbx_negsample: d_out[rows, (d_pos ? 1 : 0) + num_negs]; column 0 = d_pos[r] when given (hstack([pos, negs]));
 * every other entry is uniform over [0, num_items) with replacement (np.random.choice(..., replace=True)).
 * With d_excl_offsets/d_excl_items (CSR over queries, items sorted ascending inside a query) and d_query[rows]
 * (query of each row), items the query interacted with are never drawn (ignore_pos_items=True: uniform over
 * the complement: up to 64 redraws, then ONE exact draw of the k-th non-excluded item, so a query that interacted with
 * almost the whole corpus never receives one of its own items).  Generator: Philox4x32-10, key = seed, counter =
 * (offset + r * num_negs + j, attempt); item = high 64 bits of (low 64 random bits) x num_items.  Same (seed, offset) ->
 * same draws.  rbx_negsample_checked also bounds-checks d_query[r] against n_queries (rows of the CSR): an index outside
 * [0, n_queries) sets *d_status (numpy raises IndexError) and draws without exclusion. */
int rbx_negsample(int64_t num_items, int64_t rows, int32_t num_negs, uint64_t seed, uint64_t offset,
                  const int64_t* d_pos, const int64_t* d_query, const int64_t* d_excl_offsets,
                  const int64_t* d_excl_items, int64_t* d_out, void* stream);
int rbx_negsample_checked(int64_t num_items, int64_t rows, int32_t num_negs, uint64_t seed, uint64_t offset,
                          const int64_t* d_pos, const int64_t* d_query, const int64_t* d_excl_offsets,
                          const int64_t* d_excl_items, int64_t n_queries, int32_t* d_status, int64_t* d_out, void* stream);
/* rbx_gather_rows: for every column c and index q: dst_c[q, :] = src_c[d_index[q], :] (row_bytes bytes, any
 * element type: ids, sequences [n_items, L], float features) -- the byte-exact equivalent of
 * dict((k, v[item_indexes]) for k, v in item_corpus.items()) followed by flatten(end_dim=1).  An index outside
 * [0, n_src_rows) raises d_status (numpy raises IndexError) and reads row 0. */
typedef struct rbx_rowcopy {
  const void* src;     /* [n_src_rows, row_bytes] */
  void* dst;           /* [n_index, row_bytes] */
  int64_t row_bytes;
} rbx_rowcopy_t;
int rbx_gather_rows(const rbx_rowcopy_t* cols, int32_t n_cols, const int64_t* d_index, int64_t n_index,
                    int64_t n_src_rows, int32_t* d_status, void* stream);

/* ---- exact top-k retrieval for evaluation (core/metrics.py:54-68 evaluate_block,
 * utils/ann/faiss.py:3-15 IndexFlatIP.search).  The score matrix U I^T comes from rbx_linear_fwd (fp32 MFMA).
 * rbx_topk: for every row the k (<= 1024) largest of d_scores[r, 0..n) sorted by (score descending, column
 * ascending -- ties are deterministic, the reference leaves them to faiss/numpy); d_index (optional, same
 * layout as d_scores, any int64 values) supplies the id reported for each column, otherwise the column itself.
 * n < 2^32 - 1.  Rows shorter than k are padded with (-FLT_MAX, -1) like faiss.  Rows longer than 16 384 scores
 * (workspace): a threshold estimated from 8 192 samples of the row, one filtering sweep, exact selection among the
 * candidates; rows the estimate cannot serve fall back, on the device, to an exact selection in several levels.
 * rbx_penalize_members: scores[r, j] = (float)((double)scores[r, j] + penalty) where candidates[r, j] is in the
 * sorted CSR list offsets/items of query d_query[r]  (mask train items: "scores += -1e9 * mask").
 * rbx_membership: flags[r, j] = candidates[r, j] in the list of d_query[r]  (hits against valid_user2items). */
size_t rbx_topk_workspace_size(int64_t rows, int64_t n, int32_t k);
int rbx_topk(const float* d_scores, const int64_t* d_index, int64_t rows, int64_t n, int64_t row_stride, int32_t k,
             float* d_out_scores, int64_t* d_out_index, void* d_workspace, size_t workspace_bytes, void* stream);
int rbx_penalize_members(const int64_t* d_candidates, int64_t rows, int32_t k, const int64_t* d_query,
                         const int64_t* d_offsets, const int64_t* d_items, double penalty, float* d_scores,
                         void* stream);
int rbx_membership(const int64_t* d_candidates, int64_t rows, int32_t k, const int64_t* d_query,
                   const int64_t* d_offsets, const int64_t* d_items, uint8_t* d_flags, void* stream);

/* ---- cross layers (ranking/pytorch/layers/interactions/cross_net.py:22-59).
 * out = xi + x0 * h + bias:  CrossNetV2: h = Linear_i(xi) [rows, dim] (h_cols == dim, bias NULL);
 * CrossNet: h = xi w_i [rows, 1] (h_cols == 1) and bias[dim].  The Linear itself is rbx_linear_fwd.
 * Backward of the element-wise tail: d_dx0 = dout * h (NULL skips it), d_dh = dout * x0 (summed over the row when
 * h_cols == 1); d(xi) is dout itself and d(bias) its column sum. */
int rbx_cross_fwd(const float* d_x0, const float* d_xi, const float* d_h, const float* d_bias, int64_t rows, int32_t dim,
                  int32_t h_cols, float* d_out, void* stream);
int rbx_cross_bwd(const float* d_x0, const float* d_h, const float* d_dout, int64_t rows, int32_t dim, int32_t h_cols,
                  float* d_dx0, float* d_dh, void* stream);
/* out[r, :] = (alpha * x[r, :] + add[r, :]) * scale[r]   (add optional): SASRec's embedding prologue `e *= sqrt(D);
 * e += position_emb(...); e *= ~timeline_mask.unsqueeze(-1)` and its per-block mask (third_party/rechub/models/matching/
 * sasrec.py:68-77, 92) in one pass; its backward is the same call on the incoming gradient. */
int rbx_rowscale(const float* d_x, const float* d_add, const float* d_scale, int64_t rows, int32_t dim, float alpha,
                 float* d_out, void* stream);
/* The same prologue with the position rows read in place: sasrec.py:68-77 looks up positions = tile(arange(L), [B, 1]), i.e.
 * the table's first L rows for every sequence.  rbx_rowscale_seq: d_add holds add_rows rows of `dim` floats that repeat every
 * add_rows rows of x (rows % add_rows == 0).  rbx_seq_colsum: the gradient of those rows,
 *   out[l, :] = sum_b scale[b * seq_len + l] * g[b, l, :]        (g [batch, seq_len, dim], out [seq_len, dim], overwritten),
 * one streaming pass + a fixed-order sum of per-32-sequence partials (deterministic); the reference's embedding backward
 * (index_add over B * L position ids) gives the same sums. */
int rbx_rowscale_seq(const float* d_x, const float* d_add, int64_t add_rows, const float* d_scale, int64_t rows, int32_t dim,
                     float alpha, float* d_out, void* stream);
size_t rbx_seq_colsum_workspace_size(int64_t batch, int32_t seq_len, int32_t dim);
int rbx_seq_colsum(const float* d_g, const float* d_scale, int64_t batch, int32_t seq_len, int32_t dim, float* d_out,
                   void* d_workspace, size_t workspace_bytes, void* stream);
/* out[r, c] = base[r, c] + (c < prefix_cols ? a[r, c] + b[r, c] : 0) for c < cols; base, a, b optional (NULL = zeros),
 * strides in floats.  The input gradient of a row block that one consumer reads whole and two more read through its
 * leading columns: DeepFM feeds the same embeddings to the tower (embeddings | dense values), to FM and to the
 * first-order Linear (third_party/rechub/models/ranking/deepfm.py:34-39; autograd's glue for it is a fill, a strided
 * copy and two adds). */
int rbx_sum_prefix(const float* d_base, int64_t base_stride, const float* d_a, int64_t a_stride, const float* d_b,
                   int64_t b_stride, int64_t rows, int32_t cols, int32_t prefix_cols, float* d_out, int64_t out_stride,
                   void* stream);

/* ---- the ranking harness's loss: F.binary_cross_entropy(y_pred, y_true, reduction='mean') on sigmoid outputs
 * (ranking/pytorch/models/ranking_model.py:69, ranking/pytorch/torch_utils.py:54-65).  torch semantics: both log terms
 * clamped at -100; d_loss[1] = mean; backward d_dprob = d_dloss[0] / n * (p - y) / max(p (1 - p), 1e-12).
 * Block partials + fixed-order final sum: deterministic. */
size_t rbx_bce_workspace_size(int64_t n);
int rbx_bce_mean_fwd(const float* d_prob, const float* d_target, int64_t n, float* d_loss, void* d_workspace,
                     size_t workspace_bytes, void* stream);
int rbx_bce_mean_bwd(const float* d_prob, const float* d_target, const float* d_dloss, int64_t n, float* d_dprob,
                     void* stream);
/* The same loss on LOGITS with both backward steps folded in (a step that owns its loss, recbox_amd.graph.ShardedFMStep):
 * p = sigmoid(x) (optional output d_prob = the model's y_pred, ranking_model.py forward), d_loss[0] = the mean BCE as above,
 * d_dlogit = grad_scale * dL/dx = grad_scale / n * (p - y) / max(p (1 - p), 1e-12) * (1 - p) p (optional). */
int rbx_sigmoid_bce_mean(const float* d_logit, const float* d_target, int64_t n, float grad_scale, float* d_prob,
                         float* d_loss, float* d_dlogit, void* d_workspace, size_t workspace_bytes, void* stream);
/* The same in ONE launch: the workgroup that finishes last adds the block partials (same fixed order, so the same bits as the
 * two-launch form).  d_counter[1] is caller-owned state that must be 0 before the first call; every call leaves it at 0.
 * Calls that share a counter must not overlap (one counter per stream). */
int rbx_sigmoid_bce_mean_onepass(const float* d_logit, const float* d_target, int64_t n, float grad_scale, float* d_prob,
                                 float* d_loss, float* d_dlogit, void* d_workspace, size_t workspace_bytes,
                                 uint32_t* d_counter, void* stream);
/* y[i] = scalar[0] * x[i], the scalar read on the device: the backward of a loss whose dL/dlogit (for an upstream gradient
 * of 1) was already written by rbx_sigmoid_bce_mean in the forward -- autograd's upstream scalar is applied without a
 * host read (ops.binary_cross_entropy on the output of ops.sigmoid_output). */
int rbx_scale_by_scalar(const float* d_x, const float* d_scalar, int64_t n, float* d_y, void* stream);

/* ---- dense tower: y = act(x W^T + b) on the fp32 matrix cores (v_mfma_f32_32x32x2_f32) -------
 * core/pytorch/layers/mlp.py:25-37, ranking/pytorch/layers/blocks/mlp_block.py:42-58,
 * third_party/rechub/basic/layers.py:255-263.  x[m,k] with row stride x_stride >= k floats (a column block of
 * a wider activation, e.g. the embedding part of a padded [B, width] gather output, is read in place),
 * W[n,k] (nn.Linear layout), bias[n] or NULL, act: 0 none, 1 ReLU.  Backward: dx[m,k] (row stride dx_stride;
 * NULL to skip), dW[n,k], db[n] (NULL to skip) are OVERWRITTEN; d_y (forward output) is only read when
 * act == 1.  Rows that are 16-byte aligned (pointer and stride) are loaded as float4. */
int rbx_linear_fwd(const float* d_x, int64_t x_stride, const float* d_w, const float* d_bias, int64_t m, int32_t n,
                   int32_t k, int32_t act, float* d_y, void* stream);
size_t rbx_linear_bwd_workspace_size(int64_t m, int32_t n, int32_t k, int32_t act);
int rbx_linear_bwd(const float* d_x, int64_t x_stride, const float* d_w, const float* d_y, const float* d_dy, int64_t m,
                   int32_t n, int32_t k, int32_t act, float* d_dx, int64_t dx_stride, float* d_dw, float* d_db,
                   void* d_workspace, size_t workspace_bytes, void* stream);
/* The same contraction with the element-wise neighbours of a transformer sub-layer folded into the epilogue, so that none
 * of them is a pass of its own over an [m, n] activation (third_party/rechub/models/matching/sasrec.py:81-94: the residual
 * `Q + mha_outputs`, `seqs *= ~timeline_mask`; in the backward, the sum autograd forms for a tensor with two readers and
 * the ReLU backward of the FFN's hidden layer):
 *   rbx_linear_fwd_fused:  y[m,n]  = (act(x W^T + b) + residual) * row_scale[row]       residual, row_scale optional (NULL)
 *   rbx_linear_dx_fused:   dx[m,k] = ((dy W) o [mask > 0]) + residual                   mask [m,k], residual [m,k] optional
 * Every operand has its own row stride (floats).  n > 1 (k > 1 for dx): the logit-head kernels have no such tail. */
int rbx_linear_fwd_fused(const float* d_x, int64_t x_stride, const float* d_w, const float* d_bias, int64_t m, int32_t n,
                         int32_t k, int32_t act, const float* d_residual, int64_t residual_stride,
                         const float* d_row_scale, float* d_y, int64_t y_stride, void* stream);
int rbx_linear_dx_fused(const float* d_dy, int64_t dy_stride, const float* d_w, int64_t m, int32_t n, int32_t k,
                        const float* d_mask, int64_t mask_stride, const float* d_residual, int64_t residual_stride,
                        float* d_dx, int64_t dx_stride, void* stream);
/* The `seqs *= ~timeline_mask` of a SASRec block (sasrec.py:92) in the backward of the Linear in front of it, without a pass
 * that writes the scaled gradient g = diag(row_scale) dy:
 *   rbx_linear_dx_scaled:   dx[m,k] = (((dy W) o [mask > 0]) + residual) * row_scale[row]
 *   rbx_linear_dwdb_scaled: dW[n,k] = g^T x, db[n] = column sums of g (d_db may be NULL); workspace as rbx_linear_bwd's with
 *                           act = 0.  RBX_ERR_UNSUPPORTED unless n = k = 64, m >= 8192 and the rows are 16-byte aligned (the
 *                           slab kernel, tall_dw64_kernel): the caller then forms g itself (rbx_rowscale) and calls rbx_linear_bwd. */
int rbx_linear_dx_scaled(const float* d_dy, int64_t dy_stride, const float* d_w, int64_t m, int32_t n, int32_t k,
                         const float* d_mask, int64_t mask_stride, const float* d_residual, int64_t residual_stride,
                         const float* d_row_scale, float* d_dx, int64_t dx_stride, void* stream);
int rbx_linear_dwdb_scaled(const float* d_x, int64_t x_stride, const float* d_dy, int64_t dy_stride, const float* d_row_scale,
                           int64_t m, int32_t n, int32_t k, float* d_dw, float* d_db, void* d_workspace,
                           size_t workspace_bytes, void* stream);
/* DeepFM (third_party/rechub/models/ranking/deepfm.py:34-42): one gathered block x [m, k] = [F * D embeddings | dense values] feeds
 * the tower's first Linear, the FM term over its leading fm_cols = F * D columns and the first-order Linear over the same
 * columns.  rbx_fm_sum_fwd: y_fm[b] = 0.5 sum_d (S_d^2 - sum_f e_fd^2) and S[b, d] = sum_f e[b, f, d] in one pass.
 * rbx_linear_dx_deepfm: the block's WHOLE gradient out of the tower's dx GEMM --
 *   dx[b, c] = sum_j dy[b, j] W[j, c]  +  [c < fm_cols] (g_fm[b] (S[b, c % fm_dim] - x[b, c]) + g_lr[b] lr_w[c])
 * instead of three kernels writing three [m, fm_cols] gradients and a fourth adding them (d_lr_g / d_lr_w may be NULL). */
int rbx_fm_sum_fwd(const float* d_emb, int64_t emb_stride_b, int64_t batch, int32_t n_fields, int32_t dim, float* d_out,
                   float* d_sum, void* stream);
/* ... with the first-order Linear over the same n_fields * dim columns in the same pass (third_party/rechub/models/ranking/
 * deepfm.py:37: LR reads the block FM reads): d_lr_out[b] = <x[b, :n_fields * dim], d_lr_w> + d_lr_b[0] (d_lr_b may be NULL). */
int rbx_fm_sum_lr_fwd(const float* d_emb, int64_t emb_stride_b, int64_t batch, int32_t n_fields, int32_t dim, float* d_out,
                      float* d_sum, const float* d_lr_w, const float* d_lr_b, float* d_lr_out, void* stream);
int rbx_linear_dx_deepfm(const float* d_dy, int64_t dy_stride, const float* d_w, int64_t m, int32_t n, int32_t k,
                         const float* d_x, int64_t x_stride, const float* d_fm_sum, int32_t fm_dim, int32_t fm_cols,
                         const float* d_fm_g, const float* d_lr_g, const float* d_lr_w, float* d_dx, int64_t dx_stride,
                         void* stream);

/* f32 GEMM on the bf16 matrix cores (csrc/rbx_dense.hip, gemm_bx6_kernel): a weight matrix split ONCE per call into three
 * bf16 planes w = h + m + l (|w - h - m - l| <= 2^-24 |w|), the activations split inside the kernel, six bf16 MFMA products
 * per f32 product with f32 accumulation -- results at f32 rounding level (the same test tolerances), ~2.7x fewer matrix-core
 * cycles than v_mfma_f32_32x32x2_f32.  (rbx_linear_bwd's weight gradient dW = dy^T x takes the same route on its own when the
 * batch has >= 8192 rows: both operands are activations, split inside gemm_bxt_kernel; RBX_GEMM_BX6_DW=0 keeps it on f32 MFMAs.)
 * No counterpart in the reference (its nn.Linear is one torch call):
 *   rbx_split_bf16_size(rows, cols, transpose): bytes of the planes of a [rows, cols] matrix;
 *   rbx_split_bf16: d_out[r][c / 8][q][c % 8] (q = 0..2 = h, m, l; c < cols rounded up to 32, zero-filled; transpose != 0:
 *                   out row r is COLUMN r of d_src) -- an opaque layout for the kernel's tile loads; transpose = 0 serves
 *                   y = x W^T (rbx_linear_fwd*), 1 serves dx = dy W (rbx_linear_dx*);
 *   rbx_split_register(d_w, d_planes, rows, cols, transposed): from now on a rbx_linear_fwd* / rbx_linear_dx* call whose
 *                   weight pointer is d_w (shape [rows, cols]) runs on the planes; rbx_split_unregister(d_w) ends that (the
 *                   planes must stay valid until the GEMMs issued in between have run).  Host-side table, 256 slots,
 *                   thread-safe; RBX_GEMM_BX6=0 in the environment ignores every registration. */
size_t rbx_split_bf16_size(int32_t rows, int32_t cols, int32_t transpose);
int rbx_split_bf16(const float* d_src, int64_t ld, int32_t rows, int32_t cols, int32_t transpose, void* d_out, void* stream);
int rbx_split_register(const float* d_w, const void* d_planes, int32_t rows, int32_t cols, int32_t transposed);
int rbx_split_unregister(const float* d_w);
uint64_t rbx_gemm_bx6_count(void);      /* GEMM calls that ran on the split-operand kernel so far (tests, logs) */

/* ---- K6: fused masked-softmax attention, any sequence length (head_dim in {4, 8, 16, 32, 64}): no explicit mask,
 * lq == lk <= 256 and head_dim 32 / 64 run on the matrix cores, everything else on the VALU kernels that stream the keys
 * through LDS in chunks (round 5: the LDS-residency limit of rounds 1-4 is gone) ----
 * ranking/pytorch/layers/attentions/dot_product_attention.py:31-43 (ScaledDotProductAttention) and the
 * attention core of nn.MultiheadAttention in third_party/rechub/models/matching/sasrec.py:81-87.
 * q[bh, lq, hd], k/v[bh, lk, hd] contiguous; score = scale * <q, k>; causal != 0 hides keys j > i;
 * d_mask (optional float [bh, lq, lk]): entries == 0 set the score to mask_fill (-1e9 for the
 * first-party layer, -inf for a boolean attn_mask).  d_lse[bh, lq] is kept for the backward; d_p
 * (optional [bh, lq, lk]) receives the attention probabilities.  d_scratch: bh*lq floats. */
int rbx_attn_fwd(const float* d_q, const float* d_k, const float* d_v, const float* d_mask, int64_t bh,
                 int32_t lq, int32_t lk, int32_t head_dim, float scale, int32_t causal, float mask_fill,
                 float* d_o, float* d_lse, float* d_p, void* stream);
int rbx_attn_bwd(const float* d_q, const float* d_k, const float* d_v, const float* d_mask, const float* d_o,
                 const float* d_do, const float* d_lse, int64_t bh, int32_t lq, int32_t lk, int32_t head_dim,
                 float scale, int32_t causal, float mask_fill, float* d_dq, float* d_dk, float* d_dv,
                 float* d_scratch, void* stream);

/* The same with DROPOUT ON THE ATTENTION PROBABILITIES, as both reference attentions apply it in training
 * (nn.MultiheadAttention(embed_dim, heads, dropout) inside rechub SASRec, sasrec.py:29,56,81-87 -- default rate 0.5;
 * first-party ScaledDotProductAttention: `attention = self.dropout(attention)`, dot_product_attention.py:40-41):
 * out = (keep o softmax(s) / (1 - p)) V, the normaliser being the undropped sum; d_p, when asked for, receives the
 * dropped probabilities (what the reference returns).  keep(bh, i, j) is a counter-based function of (seed +
 * *d_seed_add, bh, i, j) -- Philox4x32-10, one call per 2 x 4 block of the (query, key) plane, 16-bit decisions,
 * p rounded to a multiple of 2^-16 -- evaluated again by the backward: no mask is stored.  d_seed_add (optional device
 * word): added to `seed` in the kernel, so that a hipGraph replay can draw a new mask (bump it between replays).
 * rbx_attn_dropout_mask writes keep as [bh, lq, lk] bytes (tests; callers that want to inspect the mask).
 * p_drop == 0 is exactly rbx_attn_fwd / rbx_attn_bwd. */
int rbx_attn_dropout_fwd(const float* d_q, const float* d_k, const float* d_v, const float* d_mask, int64_t bh,
                         int32_t lq, int32_t lk, int32_t head_dim, float scale, int32_t causal, float mask_fill,
                         float p_drop, uint64_t seed, const uint64_t* d_seed_add, float* d_o, float* d_lse, float* d_p,
                         void* stream);
int rbx_attn_dropout_bwd(const float* d_q, const float* d_k, const float* d_v, const float* d_mask, const float* d_o,
                         const float* d_do, const float* d_lse, int64_t bh, int32_t lq, int32_t lk, int32_t head_dim,
                         float scale, int32_t causal, float mask_fill, float p_drop, uint64_t seed,
                         const uint64_t* d_seed_add, float* d_dq, float* d_dk, float* d_dv, float* d_scratch,
                         void* stream);
int rbx_attn_dropout_mask(int64_t bh, int32_t lq, int32_t lk, float p_drop, uint64_t seed, const uint64_t* d_seed_add,
                          uint8_t* d_keep, void* stream);

/* The same attention on PACKED operands (MFMA path only: seq_len <= 256, head_dim in {32, 64}, no explicit mask; anything
 * else returns RBX_ERR_UNSUPPORTED): element (b, l, h, d) of a tensor sits at ptr + (b * seq_len + l) * ld + h * head_dim + d.
 * nn.MultiheadAttention projects Q, K, V with Linear layers and then transposes / splits them into [B * H, L, hd] copies
 * (sasrec.py:81-87 through torch/nn/functional.py multi_head_attention_forward); here the kernels read the projections'
 * outputs where they are -- Q [B, L, E], K and V as the two halves of ONE fused [B, L, 2 E] projection (d_k = kv,
 * d_v = kv + E, ldk = ldv = 2 E) -- and write O [B, L, E] for the output projection, dQ, and dK | dV into one
 * [B, L, 2 E] gradient, so no transpose, split or concatenation kernel runs.  d_lse / d_scratch: [batch * heads, seq_len]. */
int rbx_attn_packed_fwd(const float* d_q, int64_t ldq, const float* d_k, int64_t ldk, const float* d_v, int64_t ldv,
                        int64_t batch, int32_t heads, int32_t seq_len, int32_t head_dim, float scale, int32_t causal,
                        float p_drop, uint64_t seed, const uint64_t* d_seed_add, float* d_o, int64_t ldo, float* d_lse,
                        void* stream);
int rbx_attn_packed_bwd(const float* d_q, int64_t ldq, const float* d_k, int64_t ldk, const float* d_v, int64_t ldv,
                        const float* d_o, int64_t ldo, const float* d_do, int64_t lddo, const float* d_lse, int64_t batch,
                        int32_t heads, int32_t seq_len, int32_t head_dim, float scale, int32_t causal, float p_drop,
                        uint64_t seed, const uint64_t* d_seed_add, float* d_dq, int64_t lddq, float* d_dk, int64_t lddk,
                        float* d_dv, int64_t lddv, float* d_scratch, void* stream);

/* ---- K7, second half: the loss epilogue over the sampled logits (one forward pass + fixed-order final sum, one
 * backward pass; ATen runs log_softmax / nll_loss / log_sigmoid / mul / sum and their backward as 4-8 kernels each).
 *   rbx_softmax_ce_*        mean_r [ logsumexp(x[r, :]) - x[r, t_r] ] over logits [rows, n_classes] (row stride in floats):
 *                           core/pytorch/losses/softmax_crossentropy_loss.py:14-22 (t = 0: -log softmax(y_pred)[:, 0]) and
 *                           the CrossEntropyLoss of third_party/rechub/trainers/match_trainer.py:59-60.  d_target NULL = 0;
 *                           a target outside [0, n_classes) sets *d_status (torch raises).  d_lse[rows] is kept for the
 *                           backward, which writes d_dlogits [rows, n_classes] contiguous.
 *   rbx_pair_logsigmoid_*   scale * sum_i -w_i (log sigmoid(pos_i) + log sigmoid(-neg_i)): the pos/neg objective over
 *                           the [B, L] logit blocks of third_party/rechub/models/matching/sasrec.py:100-107 (w = 1 on
 *                           real positions, 0 on padding; NULL = all ones). */
size_t rbx_loss_workspace_size(int64_t n);
int rbx_softmax_ce_fwd(const float* d_logits, int64_t stride, int64_t rows, int32_t n_classes, const int64_t* d_target,
                       float* d_loss, float* d_lse, int32_t* d_status, void* d_workspace, size_t workspace_bytes,
                       void* stream);
int rbx_softmax_ce_bwd(const float* d_logits, int64_t stride, int64_t rows, int32_t n_classes, const int64_t* d_target,
                       const float* d_lse, const float* d_dloss, float* d_dlogits, void* stream);
int rbx_pair_logsigmoid_fwd(const float* d_pos, const float* d_neg, const float* d_weight, int64_t n, float scale,
                            float* d_loss, void* d_workspace, size_t workspace_bytes, void* stream);
int rbx_pair_logsigmoid_bwd(const float* d_pos, const float* d_neg, const float* d_weight, const float* d_dloss, int64_t n,
                            float scale, float* d_dpos, float* d_dneg, void* stream);

/* ---- pooling of a materialised [B,L,D] tensor (standalone pooling modules) ------
 * core/pytorch/layers/sequence.py:4-20, ranking/pytorch/layers/pooling.py:22-40,
 * third_party/rechub/basic/layers.py:176-230.
 * numer_masked: numerator = sum_l mask[b,l]*E[b,l,:] instead of the plain sum.
 * denom: 0 none | 1 #rows whose sum_d != 0 (recbox value mask) | 2 sum_l mask | 3 L.
 * d_inv[B] receives 1/(denom+eps) for the backward. */
int rbx_pool_fwd(const float* d_emb, const float* d_mask, int64_t batch, int32_t seq_len, int32_t dim,
                 int32_t numer_masked, int32_t denom, float eps, float* d_out, float* d_inv, void* stream);
int rbx_pool_bwd(const float* d_dout, const float* d_mask, const float* d_inv, int64_t batch, int32_t seq_len,
                 int32_t dim, int32_t numer_masked, float* d_demb, void* stream);

/* ---- the row-local chains of a SASRec block as single passes over [m, 64] ------------------------------------------
 * third_party/rechub/models/matching/sasrec.py:81-94 (one block of seq_forward) and :110-124 (PointWiseFeedForward).
 * Everything of a block except the attention is local to a row of the [B L, 64] activation; these entry points carry a
 * 32-row slab through the whole chain in registers (csrc/rbx_seqblock.hip) instead of one pass per LayerNorm /
 * projection / residual.  embed_dim is 64 (cfg 5); all activations contiguous with 16-byte aligned bases; biases and
 * LayerNorm parameters may be NULL (0 / 1).
 *   rbx_seqblock_qkv_fwd:  q = LayerNorm(x) (mean, rstd written; q itself only when d_q != NULL); Q = q Wq^T + bq; KV[:, :64] = x Wk^T + bk;
 *                          KV[:, 64:] = x Wv^T + bv, in_w [192, 64] / in_b [192] = nn.MultiheadAttention's in_proj.
 *   rbx_seqblock_ffn_fwd:  with d_attn != NULL first x = res + attn Wo^T + bo (WRITTEN to d_x: `Q + mha_outputs`; with
 *                          d_res_mean / d_res_rstd != NULL d_res is the block input e and res = LayerNorm(e) is rebuilt
 *                          from them and d_res_ln_w / _b, so that rbx_seqblock_qkv_fwd need not store q),
 *                          otherwise d_x is the input; then n = LayerNorm(x) (written to d_n unless NULL),
 *                          h = relu(n W1^T + b1), out = (n + h W2^T + b2) * keep[row] (keep NULL: 1).
 *   rbx_seqblock_ffn_bwd:  the backward of the second chain from its LayerNorm on, one pass: with g = dout * keep[row],
 *                          dW2 = g^T h, db2 = colsum g, dh = (g W2) o [h > 0], dW1 = dh^T n, db1 = colsum dh, dn = dh W1 + g,
 *                          and the LayerNorm backward of dn (dgamma, dbeta, d_dx); n is rebuilt from x, mean, rstd, gamma,
 *                          beta.  Parameter gradients (any may be NULL) are OVERWRITTEN, summed in a fixed order.
 *   rbx_seqblock_attn_in_bwd: behind the attention's backward, one pass: dq = dQ Wq + g (g = the gradient arriving over
 *                          the residual), dgamma / dbeta and the LayerNorm backward of dq, de = that + dK Wk + dV Wv
 *                          (d_dKV [m, 128] = dK | dV), stored as de * row_scale[row] * alpha when d_row_scale != NULL (the backward of SASRec's
 *                          input stage `(alpha e + position) * keep`, sasrec.py:68-77, folded into the first block).  The
 *                          three in-projection weight gradients are not part of it. */
/*   rbx_seqblock_inproj_dw: the three in-projection weight gradients in one pass: d_dw [192, 64] = dQ^T q | dK^T x | dV^T x
 *                          (nn.MultiheadAttention.in_proj_weight's layout), d_db [192] = their column sums; q = LayerNorm(x)
 *                          is rebuilt from x, mean, rstd, gamma, beta.  Either output may be NULL.
 *   rbx_seqblock_attn_out_bwd: the out-projection's backward, one pass: dO = g Wo, dWo = g^T O, dbo = colsum g. */
int rbx_seqblock_qkv_fwd(const float* d_x, int64_t m, const float* d_ln_w, const float* d_ln_b, float eps,
                         const float* d_in_w, const float* d_in_b, float* d_mean, float* d_rstd, float* d_q, float* d_Q,
                         float* d_KV, void* stream);
int rbx_seqblock_ffn_fwd(const float* d_attn, const float* d_res, const float* d_wo, const float* d_bo, float* d_x, int64_t m,
                         const float* d_ln_w, const float* d_ln_b, float eps, const float* d_w1, const float* d_b1,
                         const float* d_w2, const float* d_b2, const float* d_keep, float* d_mean, float* d_rstd, float* d_n,
                         float* d_h, float* d_out, const float* d_res_mean, const float* d_res_rstd, const float* d_res_ln_w,
                         const float* d_res_ln_b, void* stream);
size_t rbx_seqblock_ffn_bwd_workspace_size(int64_t m);
int rbx_seqblock_ffn_bwd(const float* d_dout, const float* d_keep, const float* d_h, const float* d_x, const float* d_mean,
                         const float* d_rstd, int64_t m, const float* d_ln_w, const float* d_ln_b, const float* d_w1,
                         const float* d_w2, float* d_dx, float* d_dw1, float* d_db1, float* d_dw2, float* d_db2,
                         float* d_dgamma, float* d_dbeta, void* d_workspace, size_t workspace_bytes, void* stream);
size_t rbx_seqblock_attn_in_bwd_workspace_size(int64_t m);
int rbx_seqblock_attn_in_bwd(const float* d_dQ, const float* d_dKV, const float* d_g, const float* d_x, const float* d_mean,
                             const float* d_rstd, int64_t m, const float* d_ln_w, const float* d_in_w,
                             const float* d_row_scale, float alpha, float* d_de, float* d_dgamma, float* d_dbeta,
                             void* d_workspace, size_t workspace_bytes, void* stream);
size_t rbx_seqblock_attn_out_bwd_workspace_size(int64_t m);
int rbx_seqblock_attn_out_bwd(const float* d_g, const float* d_O, int64_t m, const float* d_wo, float* d_dO, float* d_dwo,
                              float* d_dbo, void* d_workspace, size_t workspace_bytes, void* stream);
size_t rbx_seqblock_inproj_dw_workspace_size(int64_t m);
int rbx_seqblock_inproj_dw(const float* d_dQ, const float* d_dKV, const float* d_x, const float* d_mean, const float* d_rstd,
                           int64_t m, const float* d_ln_w, const float* d_ln_b, float* d_dw, float* d_db, void* d_workspace,
                           size_t workspace_bytes, void* stream);
#ifdef __cplusplus
}
#endif
#endif /* RECBOX_HIP_H */
