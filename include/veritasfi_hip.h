/*
 * veritasfi_hip.h -- C ABI of libveritasfi_hip.so (MI355X / gfx950, hand-written HIP).
 *
 * Drop-in boundary for VeritasFi's dense-retrieval hot path.  The
 * reference is pure Python and delegates this path to third-party objects; each entry point
 * below names the reference call it stands behind (paths relative to the reference root).
 * Plain pointers and sizes only -- no torch / numpy types cross this boundary.  The Python
 * classes in veritasfi_amd/ bind it with ctypes (INTEGRATION.md shows the reference-side stub).
 *
 * Conventions
 *   - every function returns VF_OK (0) or a negative VF_E* code; nothing aborts or throws across
 *     the ABI; vf_last_error() gives a thread-local message for the last failure.
 *   - "host" pointers are caller-owned host memory; "device" pointers (suffix _device / d_*) are
 *     HBM pointers on the index's GPU (e.g. torch tensor .data_ptr()); the library never frees them.
 *   - ranking contract: descending cosine, ties broken by LOWER id first; k > N pads ids with -1
 *     and scores with -FLT_MAX (faiss contract, src/utils/faissRetriever.py:37).
 *   - scores are the CANONICAL fp32 cosine (DESIGN.md "Canonical score"): bit-identical to the CPU
 *     oracle in oracle/vf_oracle.c, independent of sharding, batch size and kernel path.
 *   - zero-norm rows / queries score 0 (sklearn divides by 1: experiments/retriever/step3_mul.py:275).
 *   - a handle may be used from several threads (per-handle mutex); handles are per process.
 */
#ifndef VERITASFI_HIP_H
#define VERITASFI_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VF_VERSION 100 /* 0.1.0 */

enum {
    VF_OK = 0,
    VF_EINVAL = -1,       /* bad argument */
    VF_ENOMEM = -2,       /* host or device allocation failed */
    VF_EHIP = -3,         /* a HIP runtime call failed (message has the HIP error string) */
    VF_EUNSUPPORTED = -4, /* valid request this build does not implement (e.g. > 2^32-1 rows per shard) */
    VF_EINTERNAL = -5
};

enum { VF_DTYPE_F32 = 0, VF_DTYPE_F16 = 1, VF_DTYPE_FP8_E4M3 = 2 };

typedef struct vf_index vf_index;

/* counters of the last search on a handle (for tests / benches; not part of the reference surface) */
typedef struct vf_search_stats {
    int64_t path;            /* 0 = small-N exact dense, 1 = fused MFMA scan, 2 = chunked exact */
    int64_t n_queries;
    int64_t candidates;      /* total candidates appended by the fused scan (all queries) */
    int64_t max_candidates;  /* largest per-query candidate count */
    int64_t uncertified;     /* queries whose fused result failed the exactness certificate */
    int64_t overflowed;      /* queries whose candidate buffer overflowed */
    int64_t exact_reruns;    /* queries recomputed by the chunked exact path */
    int64_t wide_launches;   /* k_scan_wide main passes of the call (0: the 64-query HBM-bound scan served it) */
    int64_t wide_queries;    /* queries those passes served (up to 1024 per pass) */
    int64_t aux_cus;         /* CUs the main scan left to the small kernels of the other slots (0 = no CU split) */
    int64_t scans_overlap;   /* 1 = main scans of consecutive slots were not ordered against each other */
    int64_t scan_kernel;     /* main-scan kernel of the call: 1 k_scan (register loads), 2 k_scan2 (whole-line LDS-DMA), 3 k_scan_wide, 4 k_scan_wide8 (fp8 matrix instruction), 5 k_scan2r (k_scan2, half the query image in registers) */
    int64_t reserved[4];
} vf_search_stats;

int vf_version(void);
const char* vf_last_error(void);
int vf_device_count(int32_t* out);

/* ---- dense index: replaces faiss.IndexFlatIP + normalize_L2 ---------------------------------
 * src/utils/faissRetriever.py:11-26 (FaissRetriever.__init__): rows [n,d] row-major of `dtype`,
 * copied to device `device_id`.  Rows are stored as given (fp16 rows are the corpus; fp32 rows keep
 * an fp32 copy for exact scoring plus an fp16 scan copy); per-row norms are computed on the GPU.
 * VF_DTYPE_FP8_E4M3 rows (one byte per element, OCP e4m3: bias 7, no infinities, 0x7F / 0xFF = NaN) STAY fp8 in
 * HBM: the fused scan reads the bytes (half the traffic of fp16) and converts them to fp16 in registers -- exactly,
 * every e4m3 value is an fp16 value -- and the exact re-score decodes the same bytes.  A per-row scale of a
 * scaled-fp8 store cancels in the cosine, so none is taken.
 * `id_offset` is added to every returned id (row-sharding across ranks). */
int vf_index_create(vf_index** out, const void* rows, int64_t n, int32_t d, int32_t dtype,
                    int32_t device_id, int64_t id_offset);

/* Same, rows already resident in HBM on `device_id`.  The index BORROWS d_rows (no copy for fp16
 * with d % 128 == 0); the caller keeps it alive until vf_index_destroy. */
int vf_index_create_device(vf_index** out, const void* d_rows, int64_t n, int32_t d, int32_t dtype,
                           int32_t device_id, int64_t id_offset);

/* ---- ONE handle over several devices (single-process serving) -------------------------------------------------
 * The reference serves from one process (RAGManager singleton, src/utils/ragManager.py:17-30; it builds
 * EnsembleRetriever -> FaissRetriever(embeddings, fn) at src/utils/ensembleRetriever.py:39-43 and searches at :66,:139).
 * vf_index_create_sharded splits the rows into contiguous blocks of ceil(n / n_dev), one per listed
 * device (a device may be listed more than once), and returns a handle every vf_index_* entry point accepts: a search
 * peer-copies the queries from the home device (device_ids[0]) to every shard, runs the shard searches concurrently,
 * peer-copies each packed per-shard top-k back and merges them on the home device -- the result is bit-identical to
 * the single-device one (scores are canonical).  Device-buffer entry points take / return buffers on the HOME device.
 * vf_index_group adopts indexes the caller built itself (e.g. over device-resident shards); they must be contiguous
 * row blocks in ascending order (id_offset of each = id_offset + n of the previous) and the group then owns them. */
int vf_index_create_sharded(vf_index** out, const void* rows, int64_t n, int32_t d, int32_t dtype,
                            const int32_t* device_ids, int32_t n_dev);
int vf_index_create_sharded_from_file(vf_index** out, const char* path, const int32_t* device_ids, int32_t n_dev);
int vf_index_group(vf_index** out, vf_index** shards, int32_t n_shards);
/* number of shards of a handle (0 for a plain single-device index) and their devices (first `cap` of them) */
int vf_index_shards(vf_index* idx, int32_t* n_shards, int32_t* device_ids, int32_t cap);
/* peer (xGMI) access between the home device and every shard's device, as enabled when the group was formed:
 * ok[g] = 1 when both directions are on (or shard g lives on the home device), 0 when a direction could not be enabled
 * and the query / result copies of that shard are staged through the host; *n_missing counts the zeros.  The reference
 * has no counterpart (faiss-CPU index in one process, src/utils/faissRetriever.py:17-19 shows the abandoned GPU try). */
int vf_index_peer_access(vf_index* idx, int32_t* ok, int32_t cap, int32_t* n_missing);

/* Corpus file (.vfc): 64-byte header {char magic[8]="VFCORPUS"; u32 version=1; u32 dtype; u64 n; u32 d; u32 flags;
 * u8 reserved[32]} + n*d row-major elements (+ int64[n] external ids when flags bit 0 is set).  Stands where the
 * reference pulls every embedding out of Chroma into Python lists at start-up (src/utils/ensembleRetriever.py:39-43,
 * src/utils/faissRetriever.py:14): rows [row_lo, row_hi) are streamed from disk through pinned staging straight
 * into HBM (a rank loads only its shard).  veritasfi_amd/corpus_file.py writes the format. */
int vf_corpus_file_info(const char* path, int64_t* n, int32_t* d, int32_t* dtype, int32_t* has_ids);
int vf_index_create_from_file(vf_index** out, const char* path, int64_t row_lo, int64_t row_hi, int32_t device_id,
                              int64_t id_offset);

/* src/utils/faissRetriever.py:34-38 (FaissRetriever.invoke after embed_query): queries [nq,d] fp32,
 * NOT normalised (the library normalises, like faiss.normalize_L2 at :35); out_ids [nq,k] int64,
 * out_scores [nq,k] fp32, best first.  Synchronous; host buffers. */
int vf_index_search(vf_index* idx, const float* queries, int32_t nq, int32_t k, int64_t* out_ids,
                    float* out_scores);

/* Same with device buffers; work is enqueued on `stream` (a hipStream_t, NULL = default stream) and
 * the call returns after the results are final in d_ids / d_scores (it synchronises the stream once,
 * to read the 64-byte certificate word; see DESIGN.md "Exactness certificate"). */
int vf_index_search_device(vf_index* idx, const float* d_queries, int32_t nq, int32_t k,
                           int64_t* d_ids, float* d_scores, void* stream);

/* Pipelined form for throughput: _begin enqueues batch work into slot (0 <= slot < vf_index_slots)
 * and returns immediately; _end waits for that slot, certifies, repairs if needed.  Outputs are valid
 * after _end.  Slots run on internal streams ordered after `stream` at _begin time. */
int vf_index_slots(vf_index* idx, int32_t* out);
int vf_index_search_begin(vf_index* idx, int32_t slot, const float* d_queries, int32_t nq, int32_t k,
                          int64_t* d_ids, float* d_scores, void* stream);
int vf_index_search_end(vf_index* idx, int32_t slot);

int vf_index_info(vf_index* idx, int64_t* n, int32_t* d, int32_t* dtype, int32_t* device_id);
int vf_index_stats(vf_index* idx, vf_search_stats* out);
/* tuning knobs, by name ("force_path", "sample_rows", "margin", "cap", "waves" ...); tests use
 * force_path to exercise every path on the same data.  Unknown name -> VF_EINVAL.
 * "wide_mfma": matrix instruction of the wide scan (batches of >= 129 queries) over e4m3 rows: -1 auto (= 1), 1 the fp8
 *   instruction on the row bytes as stored (k_scan_wide8: the query goes in as a hi + lo pair of e4m3 codes and its exactly
 *   known residual is the query's certificate bound), 0 the fp16 instruction on converted rows (k_scan_wide).  Results are
 *   identical bit for bit; vf_search_stats.scan_kernel says which one ran.
 * "sample_rows": rows per wave the sample pass scores to seed the thresholds: -1 auto (8 for shards of up to 1.5M rows while the
 *   sample still holds 16 k' rows, else 16), or 1..64.  A speed setting: a looser seed admits more candidates, results do not change.
 * "scan_impl": the narrow scan's kernel: 1 k_scan (register loads); 2 (default) k_scan2 (whole-line LDS-DMA loads) for fp16 rows, and
 *   its register-image form k_scan2r where it measured faster (fp16 rows of 384 / 512 / 768 / 1024 elements and e4m3 rows of 768 / 1024, above 1.1M rows); 3 k_scan2
 *   wherever it fits (e4m3 rows converted in registers); 4 k_scan2, never k_scan2r; 5 k_scan2r wherever a shape of it exists, at any
 *   row count.  Same results from each.
 * "sample_impl": the sample pass's kernel: -1 auto (k_scan2r's operand path where it exists and the CU split is on), 0 k_scan, 1 k_scan2r
 *   wherever it fits.  Same sample rows and slots either way; results do not change. */
int vf_index_set_option(vf_index* idx, const char* name, int64_t value);
/* Live kernel timing with HIP events on the stream the kernels run on (bench.py roofline):
 * after vf_index_set_option(idx, "profile", 1) every fused search records events around its main
 * k_scan launch and around the whole per-batch pipeline; this returns the accumulated totals
 * (milliseconds, launches) since the option was set.  bytes_per_launch = algorithmic bytes the main
 * k_scan launch reads (rows scanned x (d*2 + 4)). */
int vf_index_profile(vf_index* idx, double* scan_ms_total, int64_t* scan_launches, double* pipeline_ms_total,
                     int64_t* scan_bytes_per_launch);
/* Makespan of the timed main scans since "profile" was set: the first launch's begin to the last launch's end (HIP events
 * on the streams the scans run on) and the number of launches.  Small shards run their scans OVERLAPPED (option
 * overlap_scans, vf_search_stats.scans_overlap): span / launches is then the launch interval the roofline is computed from. */
int vf_index_profile_span(vf_index* idx, double* span_ms, int64_t* launches);
/* Debug aid (not part of the reference surface): wall-clock stamps of the last main scan of `slot`
 * when option "debug" has bit 7 set; returns the number of 64-bit words copied (>= 0) or VF_E*. */
int vf_index_debug_read(vf_index* idx, int32_t slot, unsigned long long* out, int64_t n_words);
int vf_index_destroy(vf_index* idx);

/* ---- small dense cosine ------------------------------------------------------------------------
 * src/utils/ensembleRetriever.py:275-279 (compute_similarity_mtx after the embed loop):
 * x [n,d] fp32 host -> out [n,n] fp32 host, canonical cosine of every row pair. */
int vf_cosine_matrix(const float* x, int32_t n, int32_t d, float* out, int32_t device_id);
/* The same matrix for n rows ALREADY in the index, picked by global id (out [n,n] fp32 host, n <= 4096): what vf_cosine_matrix
 * returns for those rows' values, without re-embedding the chunk texts (ensembleRetriever.py:275 embeds each of the n
 * retrieved chunks again; their embeddings are rows of the HBM-resident corpus).  Valid when the embedder treats documents and
 * queries alike (the reference embeds both with embed_query, :275 and faissRetriever.py:33).  Single-device handles. */
int vf_cosine_matrix_rows(vf_index* idx, const int64_t* ids, int32_t n, float* out);
/* Mixed form: ids[i] >= 0 names a corpus row, ids[i] == -1 takes the NEXT vector of extra [n_extra, d] fp32 host (in order of
 * appearance; the count of -1 entries must equal n_extra) -- a chunk text the corpus does not hold, embedded by the caller.  This is
 * what serves the reference's own call, compute_similarity_mtx(chunk_texts) (src/utils/vllmManager.py:462 ->
 * src/utils/ensembleRetriever.py:265-281: texts only), from the rows in HBM: known texts by row, unknown ones embedded.  An
 * all-extra call returns vf_cosine_matrix's bits, an all-row call vf_cosine_matrix_rows's.  Single-device and sharded handles. */
int vf_cosine_matrix_rows_mixed(vf_index* idx, const int64_t* ids, int32_t n, const float* extra, int32_t n_extra, float* out);

/* experiments/retriever/step3_mul.py:275 (similarities_matrix = cosine_similarity(E, C)):
 * a [na,d], b [nb,d] fp32 host -> out [na,nb] fp32 host. */
int vf_cosine_scores(const float* a, int32_t na, const float* b, int64_t nb, int32_t d, float* out,
                     int32_t device_id);

/* ---- multi-GPU merge: after the RCCL all-gather of per-shard top-k -----------------
 * d_ids_parts / d_score_parts [nparts, nq, k] (global ids, -1 padded) on the current device ->
 * d_ids / d_scores [nq, k].  Asynchronous on `stream`. */
int vf_merge_topk_device(const int64_t* d_ids_parts, const float* d_score_parts, int32_t nparts,
                         int32_t nq, int32_t k, int64_t* d_ids, float* d_scores, int32_t device_id,
                         void* stream);

/* Same, from ONE all-gathered buffer: part g is the blob [ids nq*k int64][scores nq*k fp32][pad] at byte offset
 * g * S, S = nq*k*12 rounded up to a multiple of 16 (so every part's int64 ids stay aligned; a rank packs its result
 * that way so that a batch needs a single collective). */
int vf_merge_topk_packed_device(const void* d_parts, int32_t nparts, int32_t nq, int32_t k, int64_t* d_ids,
                                float* d_scores, int32_t device_id, void* stream);

/* src/utils/vllmManager.py:443-457 (rank_chunk score fusion): scores = rerank + time_score,
 * order = argsort descending (ties lower index first).  Host buffers, n <= 4096. */
int vf_fuse_rank(const float* rerank_scores, const float* time_scores, int32_t n, float* out_scores,
                 int64_t* out_order, int32_t device_id);

/* ---- encoder forward: embedding model and cross-encoder re-ranker -----------------------------
 * Stands where the reference calls third-party model objects:
 *   HuggingFaceEmbeddings(...).embed_query / embed_documents   src/utils/ragManager.py:50,
 *       src/utils/faissRetriever.py:33, src/load_data.py:99,124 (the embed loop)
 *   get_embeddings' model(**inputs) + pooling                   experiments/retriever/step3_mul.py:191-209,
 *       continuous_retrieval.py:127-152
 *   reranker.compute_score(pairs, batch_size=8)                  src/utils/vllmManager.py:450-452
 * BERT-family post-LN encoder (BERT / XLM-R), fp16 weights, fp32 accumulate.  Tokenisation stays in
 * Python (the reference's tokenizers are third-party too); this boundary takes token ids. */
typedef struct vf_encoder vf_encoder;
typedef struct vf_encoder_config {
    int32_t vocab, hidden, layers, heads, ffn, max_pos, type_vocab;
    int32_t roberta_pad_idx; /* -1: BERT positions 0..t-1; >= 0: cumsum(mask)*mask + pad_idx (RoBERTa / XLM-R) */
    int32_t pooling;         /* 0 CLS, 1 unmasked mean (continuous_retrieval.py:148), 2 last token (step3_mul.py:181-188),
                              * 3 masked mean (sentence-transformers pooling_mode_mean_tokens: src/utils/ragManager.py:50's
                              * HuggingFaceEmbeddings on a model whose 1_Pooling/config.json selects it) */
    int32_t normalize;       /* 1: L2-normalise the pooled vector (bge models) */
    int32_t head;            /* 0: embeddings [b, hidden]; 1: RobertaClassificationHead -> one logit per sequence */
    float ln_eps;
} vf_encoder_config;

/* Weight blobs (host), in this order.  fp16: word[vocab,H] pos[max_pos,H] type[type_vocab,H], then per
 * layer Wqkv[3H,H] (q rows, k rows, v rows) Wo[H,H] W1[F,H] W2[H,F], then (head==1) dense[H,H] out_proj[H].
 * fp32: emb_ln_gamma[H] emb_ln_beta[H], then per layer bqkv[3H] bo[H] ln1_gamma[H] ln1_beta[H] b1[F] b2[H]
 * ln2_gamma[H] ln2_beta[H], then (head==1) dense_bias[H] out_proj_bias[1].  nn.Linear layout [out,in]. */
int vf_encoder_weight_sizes(const vf_encoder_config* cfg, int64_t* n_fp16, int64_t* n_fp32);
int vf_encoder_create(vf_encoder** out, const vf_encoder_config* cfg, const void* w_fp16, int64_t n_fp16,
                      const float* w_fp32, int64_t n_fp32, int32_t device_id);
/* ids / mask / type_ids (may be NULL) [b, t] int32 host, t % 32 == 0, t <= 8192 and within the position table (pad with
 * mask 0; up to 512 tokens K / V stay resident in LDS, longer sequences -- bge-m3's 8192 -- stream them);
 * t_valid = columns the tokenizer produced (<= t; the rest is alignment padding the caller added: the
 * unmasked-mean and last-token poolings count only the first t_valid columns);
 * out [b, hidden] (head 0) or [b] (head 1) fp32 host. */
int vf_encoder_forward(vf_encoder* enc, const int32_t* ids, const int32_t* mask, const int32_t* type_ids,
                       int32_t b, int32_t t, int32_t t_valid, float* out);
/* Same, with pooling / normalize chosen for this call (-1 = the handle's setting); embedding handles only.
 * get_embeddings picks the pooling per call site on one loaded model (step3_mul.py:203-207 last token,
 * continuous_retrieval.py:146-149 unmasked mean). */
int vf_encoder_forward_pooled(vf_encoder* enc, const int32_t* ids, const int32_t* mask, const int32_t* type_ids,
                              int32_t b, int32_t t, int32_t t_valid, int32_t pooling, int32_t normalize, float* out);
/* last_hidden_state [b, t, hidden] fp32 host (callers that pool themselves: step3_mul.py:203-207) */
int vf_encoder_forward_hidden(vf_encoder* enc, const int32_t* ids, const int32_t* mask, const int32_t* type_ids,
                              int32_t b, int32_t t, float* out_hidden);
int vf_encoder_info(vf_encoder* enc, vf_encoder_config* out);
int vf_encoder_destroy(vf_encoder* enc);
/* ---- decoder-only models: the embedder / re-ranker family the reference configures by default ----------------
 * Qwen3-Embedding with last_token_pool -- get_embeddings in experiments/retriever/step3_mul.py:181-209 (model :384),
 * continuous_retrieval.py:127-152 -- and "Yes"-logit LLM re-rankers (experiments/profile/stress_test.py:197,212-225:
 * score = logits[:, -1, yes_loc]).  Pre-norm layer: x += Wo attn(rope(qnorm(q)), rope(knorm(k)), v) ; x += Wdown
 * (silu(Wgate n) * Wup n), n = RMSNorm(x); causal grouped-query attention; final RMSNorm.  fp16 weights and GEMM
 * operands; the RESIDUAL STREAM x is fp32 (the reference runs these models in the checkpoint's wider dtype,
 * step3_mul.py:62-64: a residual beyond the fp16 range must not overflow), fp32 accumulation / norms / softmax.
 * head_dim 64, 128 or 256 (gemma), t <= 4096 (the reference's truncation length, step3_mul.py:200). */
typedef struct vf_decoder vf_decoder;
typedef struct vf_decoder_config {
    int32_t vocab, hidden, layers, heads, kv_heads, head_dim, ffn;
    float rope_theta, rms_eps;
    int32_t qk_norm;    /* 1: RMSNorm over head_dim on q and k before RoPE (Qwen3) */
    int32_t pooling;    /* 0 first token, 1 unmasked mean, 2 last token (step3_mul.py:181-188) */
    int32_t normalize;  /* 1: L2-normalise the pooled vector */
    int32_t head;       /* 0: embeddings [b, hidden]; 2: one vocabulary token's logit at the last position -> [b] */
    int32_t act;            /* gated MLP activation: 0 SiLU (Qwen3), 1 tanh-GELU (gemma's gelu_pytorch_tanh) */
    int32_t norm_plus_one;  /* 1: RMSNorm gains are stored zero-centred, y = x_hat * (1 + w) (gemma) */
    float embed_scale;      /* token embeddings are multiplied by this (gemma: sqrt(hidden)); 0 or 1 = none */
} vf_decoder_config;
/* Weight blobs (host).  fp16: embed[vocab,H], then per layer Wqkv[(heads+2*kv_heads)*head_dim, H] (q rows, k rows,
 * v rows) Wo[H, heads*head_dim] Wgate_up[2*ffn, H] (gate rows, up rows) Wdown[H, ffn], then (head == 2) the lm_head
 * row of the scored token [H].  fp32: per layer input_norm[H] post_attention_norm[H] q_norm[head_dim] k_norm[head_dim],
 * then final_norm[H].  nn.Linear layout [out, in]. */
int vf_decoder_weight_sizes(const vf_decoder_config* cfg, int64_t* n_fp16, int64_t* n_fp32);
int vf_decoder_create(vf_decoder** out, const vf_decoder_config* cfg, const void* w_fp16, int64_t n_fp16,
                      const float* w_fp32, int64_t n_fp32, int32_t device_id);
/* ids / mask [b, t] int32 host, t % 32 == 0, t <= 4096 (left- or right-padded, mask 0); positions are column indices
 * (what HF does when no position_ids are passed); out [b, hidden] (head 0) or [b] (head 2) fp32 host. */
int vf_decoder_forward(vf_decoder* dec, const int32_t* ids, const int32_t* mask, int32_t b, int32_t t, int32_t t_valid,
                       float* out);
/* vf_decoder_forward with the L2 normalisation of the pooled row chosen per call (normalize: -1 the handle's, 0, 1; ignored by the
 * token-logit head): HipDecoderEmbeddings -- last_token_pool + F.normalize (experiments/retriever/continuous_retrieval.py:55-60,
 * step3_mul.py:181-209) -- takes unit vectors from the pooling kernel whatever the handle was created with. */
int vf_decoder_forward_pooled(vf_decoder* dec, const int32_t* ids, const int32_t* mask, int32_t b, int32_t t, int32_t t_valid,
                              int32_t normalize, float* out);
/* last_hidden_state [b, t, hidden] fp32 host (after the final RMSNorm), for callers that pool themselves -- the
 * reference's generic route outputs.last_hidden_state -> last_token_pool (experiments/retriever/step3_mul.py:203-207). */
int vf_decoder_forward_hidden(vf_decoder* dec, const int32_t* ids, const int32_t* mask, int32_t b, int32_t t,
                              float* out_hidden);
int vf_decoder_destroy(vf_decoder* dec);

/* re-ranker = encoder with head == 1 */
int vf_reranker_create(vf_encoder** out, const vf_encoder_config* cfg, const void* w_fp16, int64_t n_fp16,
                       const float* w_fp32, int64_t n_fp32, int32_t device_id);
int vf_reranker_score(vf_encoder* rr, const int32_t* ids, const int32_t* mask, const int32_t* type_ids, int32_t b,
                      int32_t t, float* out_scores);
int vf_reranker_destroy(vf_encoder* rr);

/* ---- vision tower (CLIP-style ViT): the "figure encoder" of BASELINE configs[3] -------------------------------------------
 * The reference holds no image model; the contract is transformers' CLIPVisionModelWithProjection (the model family the
 * config names): image_embeds = visual_projection(post_layernorm(last_hidden_state[:, 0])).  Patch embedding without bias,
 * class token, learned positions, pre_layrnorm, PRE-LayerNorm layers, head dim 64 (ViT-B/32, B/16, L/14), at most 511
 * patches per image.  The caller's image processor has already resized and normalised the pixels. */
typedef struct vf_vit vf_vit;
typedef struct vf_vit_config {
    int32_t image, patch, channels; /* 224, 14, 3 */
    int32_t hidden, layers, heads, ffn;
    int32_t proj_dim;               /* width of the joint space (768 for ViT-L/14) */
    int32_t act;                    /* 0 erf-GELU, 1 quick-GELU x * sigmoid(1.702 x) (CLIP's default) */
    int32_t normalize;              /* 1: L2-normalise the projected vector */
    float ln_eps;
} vf_vit_config;
/* Weight blobs (host), in this order.  Kp = channels*patch*patch rounded up to 64.  fp16: patch[H,Kp] (Conv2d weight
 * flattened [H, c*p*p], zero-padded columns) class[H] pos[(P+1),H], then per layer Wqkv[3H,H] (q rows, k rows, v rows)
 * Wo[H,H] W1[F,H] W2[H,F], then proj[proj_dim,H].  fp32: pre_ln_gamma[H] pre_ln_beta[H], then per layer ln1_gamma[H]
 * ln1_beta[H] bqkv[3H] bo[H] ln2_gamma[H] ln2_beta[H] b1[F] b2[H], then post_ln_gamma[H] post_ln_beta[H]. */
int vf_vit_weight_sizes(const vf_vit_config* cfg, int64_t* n_fp16, int64_t* n_fp32);
int vf_vit_create(vf_vit** out, const vf_vit_config* cfg, const void* w_fp16, int64_t n_fp16, const float* w_fp32,
                  int64_t n_fp32, int32_t device_id);
/* pixels [b, channels, image, image] fp32 host -> out [b, proj_dim] fp32 host */
int vf_vit_forward(vf_vit* vit, const float* pixels, int32_t b, float* out);
/* the same on raw bytes: pixels [b, 3, image, image] uint8 host (resized / cropped by the caller), normalised on the device as
 * (x / 255 - mean[c]) / std[c] -- the image processor's rescale + normalize; a quarter of the PCIe bytes */
int vf_vit_forward_u8(vf_vit* vit, const unsigned char* pixels, const float* mean3, const float* std3, int32_t b, float* out);
int vf_vit_destroy(vf_vit* vit);

/* ---- CLIP text tower: the QUERY side of the figure leg (BASELINE configs[3]) ------------------------------------------------
 * Figure rows are CLIP image embeddings (vf_vit_*); a text query can only be compared with them after going through the same
 * model's text tower -- transformers' CLIPTextModelWithProjection: text_embeds = text_projection(final_layer_norm(h)[eos]).
 * Token + learned position embeddings, PRE-LayerNorm layers with CAUSAL attention plus the key-padding mask, head dim 64,
 * at most 512 positions (CLIP: 77).  The reference has no counterpart (its ingest embeds text only, src/load_data.py:98-128);
 * the embedder surface it would be injected through is the one of src/utils/ragManager.py:50 (embed_query / embed_documents). */
typedef struct vf_clip_text vf_clip_text;
typedef struct vf_clip_text_config {
    int32_t vocab, max_pos;         /* 49408, 77 */
    int32_t hidden, layers, heads, ffn; /* ViT-L/14's text tower: 768, 12, 12, 3072 */
    int32_t proj_dim;               /* width of the joint space (768 for ViT-L/14) */
    int32_t act;                    /* 0 erf-GELU, 1 quick-GELU */
    int32_t eos_token_id;           /* 2 (legacy configs of the published checkpoints): pool at argmax(ids); else at the first such id */
    int32_t normalize;              /* 1: L2-normalise the projected vector */
    float ln_eps;
} vf_clip_text_config;
/* Weight blobs (host), in this order.  fp16: token[vocab,H] pos[max_pos,H], then per layer Wqkv[3H,H] (q rows, k rows, v rows)
 * Wo[H,H] W1[F,H] W2[H,F], then proj[proj_dim,H].  fp32: per layer ln1_gamma[H] ln1_beta[H] bqkv[3H] bo[H] ln2_gamma[H]
 * ln2_beta[H] b1[F] b2[H], then final_ln_gamma[H] final_ln_beta[H]. */
int vf_clip_text_weight_sizes(const vf_clip_text_config* cfg, int64_t* n_fp16, int64_t* n_fp32);
int vf_clip_text_create(vf_clip_text** out, const vf_clip_text_config* cfg, const void* w_fp16, int64_t n_fp16, const float* w_fp32,
                        int64_t n_fp32, int32_t device_id);
/* ids [b, t] int32 host, right-padded, t <= max_pos; mask [b, t] (1 = token) or NULL (every position valid: what the HF
 * pipeline passes) -> out [b, proj_dim] fp32 host */
int vf_clip_text_forward(vf_clip_text* ct, const int32_t* ids, const int32_t* mask, int32_t b, int32_t t, float* out);
int vf_clip_text_destroy(vf_clip_text* ct);

#ifdef __cplusplus
}
#endif
#endif /* VERITASFI_HIP_H */
