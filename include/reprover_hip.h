/*
 * reprover_hip.h — C ABI of libreprover_hip.so, the MI355X (gfx950) premise-retrieval engine.
 *
 * Drop-in boundary for the hot path of lean-dojo/ReProver named by BASELINE.json:north_star:
 *   retrieval/model.py::PremiseRetriever (_encode :92-114, reindex_corpus :183-210,
 *   retrieve :338-375, predict hooks :274-336), common.py::Corpus.get_nearest_premises :299-326,
 *   retrieval/index.py :13-41.
 * The reference has no native code and no FFI of its own: it calls HuggingFace `transformers`
 * (T5EncoderModel) and torch ops from Python.  These entry points are what a ctypes/cffi stub
 * on the reference side binds instead (INTEGRATION.md shows the stub); the Python package
 * `reprover_amd` is exactly such a stub plus the host-side mirror of the reference classes.
 *
 * Conventions
 *   - plain C: pointers, sizes, POD structs; no torch / HIP C++ types.  `stream` is a
 *     hipStream_t passed as void* (NULL = the legacy default stream).
 *   - every pointer documented "device" must be a HIP device pointer valid on the current device.
 *   - every call returns an RpStatus (0 = OK, < 0 = error); rp_last_error() gives the message
 *     of the last failing call on the calling thread.
 *   - launches are asynchronous on `stream`; no call synchronises the device unless documented.
 *   - no hidden allocations after rp_encoder_create(); scratch comes from caller workspaces sized
 *     by the matching *_workspace_bytes() function.
 */
#ifndef REPROVER_HIP_H
#define REPROVER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int32_t RpStatus;
enum {
  RP_OK = 0,
  RP_E_INVALID = -1,      /* bad argument / unsupported shape                                   */
  RP_E_HIP = -2,          /* a HIP runtime call failed                                          */
  RP_E_WORKSPACE = -3,    /* caller workspace too small                                         */
  RP_E_UNSUPPORTED = -4,  /* model configuration outside what the kernels implement             */
  RP_E_COMM = -5          /* RCCL could not be bound, or one of its calls failed                */
};

enum { RP_DT_F32 = 0, RP_DT_BF16 = 1 };

/* ABI / build identification; bumps when a signature changes. */
int32_t rp_abi_version(void);   /* 6 */
/* Message of the last error returned on this thread ("" if none). */
const char* rp_last_error(void);

/* ---------------------------------------------------------------------------------------------
 * Encoder: T5EncoderModel forward + masked mean-pool + L2 normalise
 *   replaces retrieval/model.py:92-114 (_encode) and, beneath it, transformers
 *   models/t5/modeling_t5.py T5Stack.forward :663-750 (what model.py:44-45,101-105 invoke).
 * ------------------------------------------------------------------------------------------- */
typedef struct RpT5Config {
  int32_t vocab_size;      /* rows of the embedding table (384 for ByT5)                        */
  int32_t d_model;         /* 1472 (small) / 1536 (base); must be a multiple of 32              */
  int32_t d_kv;            /* per-head width; kernels implement 64                              */
  int32_t num_heads;       /* 6 / 12                                                            */
  int32_t d_ff;            /* 3584 / 3968; multiple of 32; feed_forward_proj = gated-gelu       */
  int32_t num_layers;      /* 12 / 18                                                           */
  int32_t rel_num_buckets; /* 32                                                                */
  int32_t rel_max_distance;/* 128                                                               */
  float   layer_norm_eps;  /* 1e-6                                                              */
} RpT5Config;

/* Weights in HuggingFace layout (nn.Linear = [out, in] row-major), all device pointers of dtype
 * `weight_dtype` (RP_DT_F32 or RP_DT_BF16); key names: SURVEY.md App. B.5.  The engine packs its
 * own bf16 copies (fused QKV, gate/up-interleaved wi, padded) at create time; the caller may free
 * the originals afterwards. */
typedef struct RpT5LayerWeights {
  const void* ln_attn;   /* encoder.block.i.layer.0.layer_norm.weight          [d_model]         */
  const void* q;         /* ...layer.0.SelfAttention.q.weight                  [H*d_kv, d_model] */
  const void* k;         /* ...k.weight                                        [H*d_kv, d_model] */
  const void* v;         /* ...v.weight                                        [H*d_kv, d_model] */
  const void* o;         /* ...o.weight                                        [d_model, H*d_kv] */
  const void* ln_ff;     /* ...layer.1.layer_norm.weight                       [d_model]         */
  const void* wi_0;      /* ...layer.1.DenseReluDense.wi_0.weight              [d_ff, d_model]   */
  const void* wi_1;      /* ...wi_1.weight                                     [d_ff, d_model]   */
  const void* wo;        /* ...wo.weight                                       [d_model, d_ff]   */
} RpT5LayerWeights;

typedef struct RpT5Weights {
  const void* embed;            /* shared.weight                               [vocab, d_model]  */
  const void* rel_bias;         /* block.0 relative_attention_bias.weight      [buckets, H]      */
  const void* final_ln;         /* encoder.final_layer_norm.weight             [d_model]         */
  const RpT5LayerWeights* layers; /* host array of num_layers entries                           */
} RpT5Weights;

typedef struct RpEncoder RpEncoder;

/* Allocates and packs device weights (synchronises the device once). */
RpStatus rp_encoder_create(const RpT5Config* cfg, const RpT5Weights* weights, int32_t weight_dtype,
                           RpEncoder** out);
void     rp_encoder_destroy(RpEncoder* enc);

/* Scratch needed by rp_encode_varlen for `total_tokens` packed tokens in `batch` sequences. */
size_t   rp_encoder_workspace_bytes(const RpEncoder* enc, int32_t total_tokens, int32_t batch);

/* Encode `batch` sequences given as packed token ids (varlen, no padding):
 *   ids        device int32 [total_tokens]  — ByT5 ids (byte+3, EOS=1 last), sequence b occupies
 *                                             [cu_seqlens[b], cu_seqlens[b+1])
 *   cu_seqlens device int32 [batch+1], cu_seqlens[0] = 0, cu_seqlens[batch] = total_tokens
 *   max_len    longest sequence (host value; sizes the attention grid)
 *   out        device [batch, d_model] of out_dtype (RP_DT_F32 / RP_DT_BF16): unit-norm rows,
 *              = F.normalize(masked_mean(last_hidden_state))  (model.py:108-114)
 * Equivalent to PremiseRetriever._encode(input_ids, attention_mask) with right-padded inputs:
 * padding never influences a row, so it is not materialised. */
RpStatus rp_encode_varlen(RpEncoder* enc, const int32_t* ids, const int32_t* cu_seqlens,
                          int32_t batch, int32_t total_tokens, int32_t max_len,
                          void* out, int32_t out_dtype,
                          void* workspace, size_t workspace_bytes, void* stream);

/* The reference's own call form: PremiseRetriever._encode(input_ids, attention_mask) (retrieval/model.py:92-114)
 * on a right-padded batch as the tokenizer / datamodule.py:130-144 collate produce it.
 *   input_ids       device int64 [batch, padded_len]
 *   attention_mask  device int64 [batch, padded_len]   1 on real tokens, then 0
 *   out             device [batch, d_model] of out_dtype, as rp_encode_varlen
 *   meta            device int32 [4], written asynchronously: meta[0] = total real tokens, meta[1] = longest
 *                   sequence, meta[2] != 0 when some row is NOT right-padded (ones after a zero) or is empty -
 *                   `out` is then meaningless and the caller must raise (read it at its next synchronisation).
 * Lengths, cu_seqlens and the packed ids are derived on the device (no host round trip, no torch kernels); the
 * encoder pass is launched for the upper bound batch * padded_len tokens and skips, on the device, every tile
 * beyond the real count.  Results are bit-identical to rp_encode_varlen on the packed form of the same batch. */
size_t   rp_encode_padded_workspace_bytes(const RpEncoder* enc, int32_t batch, int32_t padded_len);
RpStatus rp_encode_padded(RpEncoder* enc, const int64_t* input_ids, const int64_t* attention_mask,
                          int32_t batch, int32_t padded_len, void* out, int32_t out_dtype, int32_t* meta,
                          void* workspace, size_t workspace_bytes, void* stream);

/* Host-only helper, exact restatement of T5Attention._relative_position_bucket (bidirectional)
 * — modeling_t5.py:216-262; exported so the bucket table can be checked without a GPU. */
int32_t  rp_relative_position_bucket(int32_t relative_position, int32_t num_buckets,
                                     int32_t max_distance);

/* ---------------------------------------------------------------------------------------------
 * Retrieval: similarity GEMM + accessibility mask + exact top-k
 *   replaces common.py:299-326 (Corpus.get_nearest_premises): `Q @ E.T`, full argsort, and the
 *   per-query Python walk over accessible premises.
 *
 * Accessibility in array form (common.py:280-289; derivation SURVEY.md §8 a6'):
 *   premise i is accessible to query j  iff
 *       bit j of file_bits_t[file_of[i]]                       (file imported, transitively)
 *    or (file_of[i] == own_file[j] and end_key[i] <= q_key[j]) (earlier in the same file)
 *   with key = (line_nb << 20) | column_nb.  Passing file_of == NULL disables the mask.
 * Ordering: (score descending, id ascending) — the reference's argsort leaves ties unspecified.
 * ------------------------------------------------------------------------------------------- */
enum { RP_TOPK_AUTO = 0, RP_TOPK_DENSE = 1 /* force the single-pass dense path */ };

/* D = the embedding width for BOTH entry points (the e4m3 one plans with half the 2-byte units per row; its plan can
 * only differ by taking the first-generation filter kernel, which needs no more workspace than this returns). */
size_t   rp_sim_topk_workspace_bytes(int32_t B, int32_t N, int32_t D, int32_t k, int32_t flags);

/*   Q            device bf16 [B, D]   query embeddings (unit norm)
 *   E            device bf16 [N, D]   this rank's rows of the premise-embedding matrix
 *   file_of      device int32 [N]     file index of each premise           (NULL = no mask)
 *   end_key      device int64 [N]     end position key of each premise
 *   file_bits_t  device uint32 [F, ceil(B/32)]  bit j of row f = query j may use file f
 *   own_file     device int32 [B];  q_key device int64 [B]
 *   id_offset    added to local row numbers to form the ids written (row-sharded corpus)
 *   out_scores   device f32 [B, k];  out_ids device int32 [B, k]  (rows sorted best-first;
 *                entries past out_count[j] are -inf / -1)
 *   out_count    device int32 [B]: min(k, #accessible premises on this rank); the caller maps
 *                a global count < k to the reference's ValueError (common.py:323-324).
 *                -1 = "internal candidate overflow: call again with RP_TOPK_DENSE".  The two-pass plan keeps at most
 *                max(8192, 8 k stride) + k candidate keys per query (about k * stride lie above the sampled bound); a
 *                query that would need more - adversarial score distributions, or a sample with fewer than k accessible
 *                rows, which yields no bound - reports -1 and every caller repeats the search with the dense plan.
 *   k <= 1024 per call (the final selection sorts in LDS).  The reference accepts any k (common.py:299-326):
 *   rp_sim_topk_after continues the same ranking page by page, which is how the host shim serves k > 1024.    */
RpStatus rp_sim_topk(const void* Q, const void* E, int32_t B, int32_t N, int32_t D,
                     const int32_t* file_of, const int64_t* end_key,
                     const uint32_t* file_bits_t, int32_t F,
                     const int32_t* own_file, const int64_t* q_key,
                     int32_t id_offset, int32_t k, int32_t flags,
                     float* out_scores, int32_t* out_ids, int32_t* out_count,
                     void* workspace, size_t workspace_bytes, void* stream);

/* The next page of rp_sim_topk's ranking: identical arguments plus, per query, the LAST entry of what the caller already
 * holds - after_score device f32 [B], after_id device int32 [B] (the id as written by the previous call, id_offset
 * included; after_id[j] < 0: no bound for query j).  Only premises that come strictly after (after_score[j], after_id[j])
 * in the (score descending, id ascending) order qualify; out_count counts those (min(k, remaining accessible)).  Pages
 * concatenate to exactly what one call with a larger k would return: the order is total (ids break ties).          */
RpStatus rp_sim_topk_after(const void* Q, const void* E, int32_t B, int32_t N, int32_t D,
                           const int32_t* file_of, const int64_t* end_key,
                           const uint32_t* file_bits_t, int32_t F,
                           const int32_t* own_file, const int64_t* q_key,
                           int32_t id_offset, const float* after_score, const int32_t* after_id,
                           int32_t k, int32_t flags,
                           float* out_scores, int32_t* out_ids, int32_t* out_count,
                           void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * e4m3 index (BASELINE.json configs[4]: fp8 similarity).  No reference counterpart: the reference
 * keeps the index in the model dtype (retrieval/model.py:190-194).  Rows are quantised one by one,
 *   scale[r] = max|x[r,:]| / 448 (1.0 for an all-zero row),
 *   q[r,c]   = round-to-nearest-even to OCP e4m3fn of x[r,c] * (448 / max|x[r,:]|),
 * and the score of (query j, premise i) is  (sum_c q8[j,c] * e8[i,c]) * q_scale[j] * e_scale[i]
 * with the sum accumulated in fp32 by the block-scaled MFMA (all block scales 2^0).  Mask, ordering,
 * outputs, workspace (rp_sim_topk_workspace_bytes) and flags are exactly rp_sim_topk's; D % 64 == 0.
 * ------------------------------------------------------------------------------------------- */
RpStatus rp_quantize_rows_e4m3(const void* X, int32_t x_dtype /* RP_DT_F32 | RP_DT_BF16 */, int64_t rows,
                               int32_t D, void* out_fp8 /* u8 [rows, D] */, float* out_scale /* [rows] */,
                               void* stream);
RpStatus rp_sim_topk_fp8(const void* Q8, const float* q_scale, const void* E8, const float* e_scale,
                         int32_t B, int32_t N, int32_t D,
                         const int32_t* file_of, const int64_t* end_key,
                         const uint32_t* file_bits_t, int32_t F,
                         const int32_t* own_file, const int64_t* q_key,
                         int32_t id_offset, int32_t k, int32_t flags,
                         float* out_scores, int32_t* out_ids, int32_t* out_count,
                         void* workspace, size_t workspace_bytes, void* stream);

RpStatus rp_sim_topk_fp8_after(const void* Q8, const float* q_scale, const void* E8, const float* e_scale,
                               int32_t B, int32_t N, int32_t D,
                               const int32_t* file_of, const int64_t* end_key,
                               const uint32_t* file_bits_t, int32_t F,
                               const int32_t* own_file, const int64_t* q_key,
                               int32_t id_offset, const float* after_score, const int32_t* after_id,
                               int32_t k, int32_t flags,
                               float* out_scores, int32_t* out_ids, int32_t* out_count,
                               void* workspace, size_t workspace_bytes, void* stream);

/* Merge R per-rank results (as gathered by an RCCL all-gather) into the global top-k.
 *   scores device f32 [R, B, k], ids device int32 [R, B, k], counts device int32 [R, B]
 *   workspace: rp_topk_merge_workspace_bytes(R, B, k) bytes.                                   */
size_t   rp_topk_merge_workspace_bytes(int32_t R, int32_t B, int32_t k);
RpStatus rp_topk_merge(const float* scores, const int32_t* ids, const int32_t* counts,
                       int32_t R, int32_t B, int32_t k,
                       float* out_scores, int32_t* out_ids, int32_t* out_count,
                       void* workspace, size_t workspace_bytes, void* stream);

/* The same merge reading an all-gather's receive buffer as it lies: every rank sent ONE packed block
 *   [ scores f32 [Bt, k] | ids int32 [Bt, k] | counts int32 [Bt] ]      (Bt = all queries of the step)
 * so rank r's element [q, i] sits at base + r * rank_stride + q * k + i (4-byte units; rank_stride = Bt * (2 k + 1)) and
 * a rank merges only its own queries by passing pointers advanced to its first query (scores + q0 * k, ids + q0 * k,
 * counts + q0) with B = its number of queries.  No copy, no re-layout between the collective and the merge. */
RpStatus rp_topk_merge_strided(const float* scores, const int32_t* ids, const int32_t* counts, int64_t rank_stride,
                               int32_t R, int32_t B, int32_t k,
                               float* out_scores, int32_t* out_ids, int32_t* out_count,
                               void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * The collective of the sharded retrieve step (one all-gather of the per-rank lists, then the
 * per-query merge).  The reference has no counterpart - it replicates the whole index per process
 * (retrieval/model.py:274-279, prover/proof_search.py:438-447); this is the exchange its row-sharded replacement needs,
 * behind the ABI so that a caller without torch.distributed can run it.
 *
 * RCCL is bound at run time (dlopen: a copy the process already holds, else librccl.so.1 on the search path, else
 * /opt/rocm/lib; RP_RCCL_LIB overrides); without it these calls return RP_E_COMM and everything else keeps working.
 * One communicator per (rank, GPU): rank 0 calls rp_comm_unique_id, the caller carries the 128 bytes to the other ranks
 * by whatever channel it has (a file, a socket, torch.distributed's store), every rank calls rp_comm_init with its
 * device current.  Calls are enqueued on `stream`; nothing synchronises; no allocation after rp_comm_init.
 * ------------------------------------------------------------------------------------------- */
#define RP_COMM_ID_BYTES 128
typedef struct RpComm RpComm;
RpStatus rp_comm_unique_id(void* id_out /* host, RP_COMM_ID_BYTES */);
RpStatus rp_comm_init(const void* unique_id, int32_t rank, int32_t world, RpComm** out);
RpStatus rp_comm_destroy(RpComm* comm);
int32_t  rp_comm_world(const RpComm* comm);
int32_t  rp_comm_rank(const RpComm* comm);
/* ncclAllGather of bytes_per_rank bytes (a multiple of 4) from every rank: recv = [world, bytes_per_rank], rank order.
 * Used as it stands for the query embeddings ([B, D] bf16 per rank) and for persisting a sharded index. */
RpStatus rp_comm_allgather(RpComm* comm, const void* send, void* recv, size_t bytes_per_rank, void* stream);
/* The step's exchange in one call: all-gather of this rank's packed block
 *   send_block  [ scores f32 [Bt, k] | ids int32 [Bt, k] | counts int32 [Bt] ]   (rp_sim_topk wrote into it in place)
 *   recv_blocks [world, Bt (2 k + 1)] 4-byte units
 * followed by rp_topk_merge_strided over queries [q0, q0 + B) read from recv_blocks as they lie -> out_* [B, k] / [B].
 * A rank whose count is -1 (candidate overflow, see rp_sim_topk) contributes nothing to the merge: the caller reads the
 * gathered counts (recv_blocks[r, 2 Bt k + q], identical on every rank) and, if any is negative, all ranks redo the step
 * with RP_TOPK_DENSE - what dist.PendingShardedSearch.finish does.
 * workspace: rp_topk_merge_workspace_bytes(world, B, k). */
RpStatus rp_allgather_topk(RpComm* comm, const void* send_block, void* recv_blocks, int32_t Bt, int32_t k,
                           int32_t q0, int32_t B, float* out_scores, int32_t* out_ids, int32_t* out_count,
                           void* workspace, size_t workspace_bytes, void* stream);

/* Per-batch accessibility operand built on the device (replaces the host-side bit transposition of
 * common.py:280-289's closure for a batch): file_bits_t [F, ceil(B/32)] (as rp_sim_topk takes it) from
 *   reach     device uint64 [F, ceil(F/64)]   bit g of row f = file f imports file g (transitive closure, resident)
 *   own_file  device int32 [B]                the file each query's theorem lives in
 * so a search uploads 12 bytes per query (own_file, q_key). */
RpStatus rp_build_file_bits(const uint64_t* reach, int32_t F, const int32_t* own_file, int32_t B,
                            uint32_t* file_bits_t, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Training forward, loss part: replaces retrieval/model.py:133-139
 *   similarity = torch.mm(context_emb, all_premise_embs.t());  loss = F.mse_loss(similarity, label)
 * (the embeddings come from rp_encode_padded / rp_encode_varlen with out_dtype RP_DT_F32).
 *   context_emb    device f32 [B, D];  premise_embs device f32 [P, D]  (P = B * (1 + num_negatives))
 *   label          device f32 [B, P]   (datamodule.py:160-175)
 *   out_loss       device f32 [1]: mean over the B x P entries of (similarity - label)^2, summed in index order
 *   out_similarity device f32 [B, P] or NULL
 * The encoder's backward is not part of this library; the two ends of the training step around it are:
 * ------------------------------------------------------------------------------------------- */
size_t   rp_contrastive_mse_workspace_bytes(int32_t B, int32_t P);
RpStatus rp_contrastive_mse(const float* context_emb, const float* premise_embs, const float* label,
                            int32_t B, int32_t P, int32_t D, float* out_loss, float* out_similarity,
                            void* workspace, size_t workspace_bytes, void* stream);
/* Backward of that loss (what autograd does behind retrieval/model.py:137-139): with S = out_similarity of the forward,
 * dS = 2 (S - label) / (B P);  d_context_emb [B, D] = dS premise_embs;  d_premise_embs [P, D] = dS^T context_emb. */
RpStatus rp_contrastive_mse_backward(const float* context_emb, const float* premise_embs, const float* similarity,
                                     const float* label, int32_t B, int32_t P, int32_t D, float* d_context_emb,
                                     float* d_premise_embs, void* stream);
/* One torch.optim.AdamW update (/root/reference/common.py:395: `torch.optim.AdamW(parameters, lr=lr)`; decoupled
 * weight decay, bias-corrected moments) of n fp32 parameters, in place:  param, exp_avg, exp_avg_sq device f32 [n],
 * 16-byte aligned;  step = 1, 2, ...;  lr = base rate x the schedule's factor for this step
 * (get_constant_schedule_with_warmup: min(1, (step - 1) / warmup_steps)).  torch defaults: betas (0.9, 0.999),
 * eps 1e-8, weight_decay 1e-2. */
RpStatus rp_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, int32_t step,
                       float lr, float beta1, float beta2, float eps, float weight_decay, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Training step: replaces what autograd + Lightning do behind retrieval/model.py:155-181 (`training_step` differentiating
 * `forward` :116-140 through `_encode` :92-114 and transformers' T5Stack) and common.py:381-405 (`get_optimizers`).
 * Parameters, gradients and optimizer moments are FLAT fp32 device buffers in one canonical layout:
 *   embed [V, D] | rel_bias [buckets, H] | final_ln [D] | per layer: ln_attn [D], q, k, v [H*d_kv, D], o [D, H*d_kv],
 *   ln_ff [D], wi_0, wi_1 [d_ff, D], wo [D, d_ff]      (HF shapes; every tensor starts at a multiple of 64 elements)
 * rp_train_param_layout writes the rp_train_param_tensors(cfg) + 1 element offsets (last = total length incl. padding).
 * One step = rp_train_forward (all sequences of the batch - contexts, positives, negatives - packed as ONE varlen pass)
 *  -> rp_contrastive_mse (+ _backward) on the [batch, D] embeddings -> rp_train_backward -> rp_grad_norm (optional
 * clipping) -> rp_adamw_step over the flat buffers -> rp_trainer_load_params.  Dropout: rp_trainer_set_dropout (off by
 * default).  d_model and d_ff must be multiples of 64.
 * ------------------------------------------------------------------------------------------- */
typedef struct RpTrainer RpTrainer;
int32_t  rp_train_param_tensors(const RpT5Config* cfg);                 /* 3 + 9 * num_layers */
RpStatus rp_train_param_layout(const RpT5Config* cfg, int64_t* offsets /* [tensors + 1] */);
/* `params`: device f32 flat buffer in the layout above.  Allocates the bf16 compute copies (synchronises once). */
RpStatus rp_trainer_create(const RpT5Config* cfg, const float* params, RpTrainer** out);
void     rp_trainer_destroy(RpTrainer* tr);
/* The inference engine on the trainer's current weights (for rp_encode_varlen / rp_encode_padded: validation re-indexes
 * with the model being trained, retrieval/model.py:212-225); owned by the trainer. */
RpEncoder* rp_trainer_encoder(RpTrainer* tr);
/* Dropout of the following rp_train_forward / rp_train_backward pairs (T5's dropout_rate, 0.1 in the reference's training:
 * transformers modeling_t5.py :725 embeddings, :168/:360 attention probabilities, :400 / :140 the two residual branches, :110
 * inside the gated FFN, :745 after the final norm).  Counter-based: element (site, row, column) is kept iff the 16-bit field
 * (column & 1) of hash(seed, site, row, column >> 1) is >= round(p * 2^16), and scaled by 1 / (1 - p); the backward
 * regenerates the masks of the forward that ran with the same seed, nothing is stored.  p = 0 (the default) switches it off;
 * pass a fresh seed per step. */
RpStatus rp_trainer_set_dropout(RpTrainer* tr, float p, uint32_t seed);
/* Refresh every compute copy from the fp32 masters (call after each optimizer step); launch-only. */
RpStatus rp_trainer_load_params(RpTrainer* tr, const float* params, void* stream);
size_t   rp_train_workspace_bytes(const RpTrainer* tr, int32_t total_tokens, int32_t batch);
/* Forward over `batch` packed sequences (ids / cu_seqlens as rp_encode_varlen); out_emb device f32 [batch, d_model] =
 * _encode's unit-norm rows; the activations the backward needs stay in `workspace`, which must be handed unchanged to
 * rp_train_backward. */
RpStatus rp_train_forward(RpTrainer* tr, const int32_t* ids, const int32_t* cu_seqlens, int32_t batch,
                          int32_t total_tokens, float* out_emb, void* workspace, size_t workspace_bytes, void* stream);
/* d_emb device f32 [batch, d_model] = d loss / d out_emb;  grads: flat f32 buffer, every element overwritten with
 * d loss / d parameter (padding gaps untouched: zero them once).  Deterministic: no floating-point atomics on HBM. */
RpStatus rp_train_backward(RpTrainer* tr, const float* params, const int32_t* ids, const int32_t* cu_seqlens,
                           int32_t batch, int32_t total_tokens, const float* d_emb, float* grads,
                           void* workspace, size_t workspace_bytes, void* stream);
/* out_norm[0] (device) = ||grads||_2 over n floats (Lightning's gradient_clip_val, confs/cli_lean4_random.yaml:19, clips
 * on it); scratch: 1024 device floats.  Pair with rp_adamw_step_clipped. */
RpStatus rp_grad_norm(const float* grads, int64_t n, float* out_norm, float* scratch, void* stream);
/* rp_adamw_step with the gradient scaled by min(1, max_norm / (total_norm[0] + 1e-6)) (torch.nn.utils.clip_grad_norm_);
 * total_norm: device f32 [1] from rp_grad_norm, or NULL for no clipping. */
RpStatus rp_adamw_step_clipped(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, int32_t step,
                               float lr, float beta1, float beta2, float eps, float weight_decay,
                               const float* total_norm, float max_norm, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Per-kernel timing with HIP events on the launch stream (used by bench.py for the roofline
 * object).  While enabled, every kernel launch of the engine is bracketed by an event pair.
 * ------------------------------------------------------------------------------------------- */
enum {
  RP_K_EMBED = 0, RP_K_RMSNORM = 1, RP_K_GEMM_QKV = 2, RP_K_ATTENTION = 3, RP_K_GEMM_O = 4,
  RP_K_GEMM_WI = 5, RP_K_GEMM_WO = 6, RP_K_POOL = 7, RP_K_SCAN = 8 /* dense / filter pass */, RP_K_SELECT = 9,
  RP_K_SCAN_SAMPLE = 10 /* sample pass of the two-pass plan */,
  /* training step (rp_train_*) */
  RP_K_BWD_DGRAD = 11 /* dY W GEMMs (+ fused GELU / RMSNorm backward) */, RP_K_BWD_WGRAD = 12 /* dY^T X GEMMs */,
  RP_K_BWD_ATTENTION = 13, RP_K_BWD_OTHER = 14 /* pooling, embedding, row statistics, gradient finishing */,
  RP_K_OPTIMIZER = 15 /* AdamW, gradient norm, weight re-packing */,
  RP_K_COLLECTIVE = 16 /* the all-gather of rp_comm_allgather / rp_allgather_topk */, RP_K_COUNT = 17
};
RpStatus rp_profile_enable(int32_t on);   /* on != 0: start collecting (clears previous records) */
/* Synchronises the recorded events; total_ms = sum of launch durations, launches = their number. */
RpStatus rp_profile_read(int32_t kernel_class, double* total_ms, int64_t* launches);

/* ---------------------------------------------------------------------------------------------
 * Kernel-level entry points used by the parity tests (tests/test_kernels_gpu.py) to check each
 * HIP kernel against the oracle in isolation.  Same conventions as above.
 * ------------------------------------------------------------------------------------------- */
enum { RP_EPI_STORE_BF16 = 0, RP_EPI_RESID = 1, RP_EPI_GEGLU_BF16 = 2,
       RP_EPI_RESID8 = 3 /* the inference pass's residual stream: out = bf16 plane [M, n_valid], then one-byte extension plane [M, n_valid] */ };
/* C = A[M,K] (bf16) x W[N,K]^T (bf16); M, N multiples of 128 (N may exceed n_valid: only the
 * first n_valid columns are written), K multiple of 32.
 *   STORE_BF16: out bf16 [M, n_valid]
 *   RESID:      out = the residual stream's two bf16 planes [2, M, n_valid] (hi = bf16(x), lo = bf16(x - hi));
 *               x += C, re-split (n_valid multiple of 8) - the training step's form
 *   RESID8:     out = bf16 plane [M, n_valid] followed by a uint8 plane [M, n_valid]: x = float(((hi << 16) | (ext << 8)) -
 *               0x8000), the fp32 word of x rounded to its top 24 bits (hi = that word rounded to 16 bits, half away from
 *               zero; ext = the signed remainder stored biased by 128 = bits 8..15 of (the rounded word + 0x8000); ABI 6);
 *               x += C, re-split - the inference pass's form
 *   GEGLU_BF16: W rows interleaved 32 gate / 32 up; out bf16 [M, n_valid/2] = gelu_new(g)*u  */
RpStatus rp_dbg_gemm(const void* A, const void* W, void* out, int32_t M, int32_t N, int32_t K,
                     int32_t n_valid, int32_t epilogue, void* stream);
/* The same GEMM with the fused T5-RMSNorm pieces (rp_encoder.hip "RMSNorm folded into the GEMMs"):
 *   STORE / GEGLU: accumulators are scaled by rs[row] = rsqrt(sum_p ssp_in[p, row] * inv_d + eps), ssp slot-major [np, M] (ssp_in NULL = 1);
 *   RESID: additionally writes ssp_out[p, row] = sum of x^2 over features [64p, 64p+64) (x as stored: hi + lo); xb_out is
 *          unused (the hi plane IS the next projection's operand) and may be NULL. */
RpStatus rp_dbg_gemm_fused(const void* A, const void* W, void* out, int32_t M, int32_t N, int32_t K,
                           int32_t n_valid, int32_t epilogue, const float* ssp_in, int32_t np_in, float inv_d,
                           float eps, void* xb_out, float* ssp_out, int32_t np_out, void* stream);
/* The T5 RMSNorm statistic as the product computes it (there is no separate normalisation pass): rs[row] =
 * rsqrt(sum_p ssp[p, row] * inv_d + eps) from the slot-major partial sums of squares the residual epilogues emit. */
/* Box calibration for the bench line: `waves_per_cu` x 256 CUs waves each issue `mfmas` v_mfma_f32_32x32x16_bf16 from
 * registers (pseudo-random bf16 operands, eight accumulators per wave, no memory traffic): 32768 FLOP per instruction.
 * The achieved rate is what THIS box's matrix pipes sustain under load (the chip clocks to its power budget: boxes of one
 * pool differ by several per cent), against which a step time can be read.  `sink` (>= 4 bytes) keeps the result live. */
RpStatus rp_dbg_mfma_probe(int32_t waves_per_cu, int32_t mfmas, float* sink, void* stream);
RpStatus rp_dbg_rowscale(const float* ssp /* [np, rows] */, float* rs /* [rows] */, int32_t rows, int32_t np,
                         float inv_d, float eps, void* stream);
RpStatus rp_dbg_attention(const void* qkv_bf16, const int32_t* cu_seqlens, const float* bias_tab,
                          void* out_bf16, int32_t batch, int32_t max_len, int32_t num_heads,
                          int32_t rows_total, void* stream);
/* Training kernels in isolation (tests/test_train_kernels_gpu.py).
 *   rp_dbg_wgrad: out f32 [|splits|, ny, nx], partial s = Y[rows of split s]^T X;  Y bf16 [T, ny], X bf16 [T, nx], T % 64 == 0;
 *     splits > 0: 256 x 256 tiles, splits < 0: 128 x 128 tiles.
 *   rp_dbg_attention_bwd: runs the forward (att_out, lse_out [H, rows_total]) and both backward kernels on packed qkv;
 *     att = the O the backward uses for delta (NULL: att_out); dqkv bf16 [rows_total, 3*H*64] (rows of real tokens written);
 *     dtab f32 [2*128+1, H]: gradient of the [H, 257] bias table, transposed.  Synchronises.
 *   rp_dbg_dgrad: mode 0 = gated-GELU backward epilogue (A = dx [M, K], W = Wo2^T [N = d_ff, K]; aux0 = gu bf16 [M, 2N],
 *     aux1 = rs [M]; out0 = dzs bf16 [M, 2N], out1 = row dots f32 [M]); mode 1 = RMSNorm-backward residual epilogue
 *     (A = dzs [M, K], W [N, K]; aux0 = x bf16 [M, N], aux1 = rcoef [M]; out0 = planes [2, M, N] updated in place).
 *     variant: tile configuration (0 = 128x128x32, 26 = 256x256x64 pipelined). */
RpStatus rp_dbg_wgrad(const void* Y, const void* X, float* out, int32_t T, int32_t ny, int32_t nx, int32_t splits,
                      void* stream);
/* two products over the same T token rows in ONE launch of 256 x 256 tiles with a common split count (the training step's
 * weight-gradient pairs): out0 f32 [splits, ny0, nx0], out1 f32 [splits, ny1, nx1] as rp_dbg_wgrad writes them. */
RpStatus rp_dbg_wgrad_pair(const void* Y0, const void* X0, float* out0, int32_t ny0, int32_t nx0, const void* Y1,
                           const void* X1, float* out1, int32_t ny1, int32_t nx1, int32_t T, int32_t splits, void* stream);
RpStatus rp_dbg_attention_bwd(const void* qkv, const void* att, const void* datt, const int32_t* cu_seqlens,
                              const float* bias_tab, int32_t batch, int32_t num_heads, int32_t rows_total, void* lse_out,
                              void* att_out, void* dqkv, float* dtab, void* stream);
/* out u8 [rows, cols]: 1 where element (site, row0 + r, col0 + c) is kept.  Sites: 0 embeddings, 1 final norm output,
 * 16 + 8 i + {0 attention probabilities (row = the query's packed token index, col = (head << 20) | key offset in its
 * sequence), 1 attention residual branch, 2 gated product inside the FFN, 3 FFN residual branch} (row = packed token
 * index, col = feature).  Element (site, row, col) is kept iff the 16-bit field (col & 1) of hash(seed, site, row, col >> 1)
 * is >= round(p * 65536): one hash per column pair (rp_encoder_kernels.h::drop_mul2). */
RpStatus rp_dbg_dropout_mask(float p, uint32_t seed, uint32_t site, uint32_t row0, uint32_t col0, int32_t rows,
                             int32_t cols, uint8_t* out, void* stream);
RpStatus rp_dbg_dgrad(const void* A, const void* W, int32_t M, int32_t N, int32_t K, int32_t mode, const void* aux0,
                      const float* aux1, void* out0, float* out1, int32_t variant, void* stream);
/* Tuning knobs (integers), e.g. "gemm_variant"; returns RP_E_INVALID for unknown names. */
RpStatus rp_set_option(const char* name, int32_t value);

#ifdef __cplusplus
}
#endif
#endif /* REPROVER_HIP_H */
