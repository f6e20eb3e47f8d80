/*
 * mi355rag.h — C-ABI of libmi355rag.so: the MI355X (gfx950) dense-retrieval hot path.
 *
 * This is the drop-in boundary for the ONE data-parallel path of alxd/rag's
 * advanced_rag.py: chunk embedding + exact (flat) top-k search behind the
 * vector-store retriever interface.  The reference has no FFI of its own (it is
 * pure Python that calls pip packages), so every entry point cites the reference
 * call site whose third-party arithmetic it replaces:
 *
 *   advanced_rag.py:104-107  HuggingFaceEmbeddings(all-MiniLM-L6-v2, cpu)  -> mi_encoder_*
 *   advanced_rag.py:295      FAISS.from_documents(split_data, embed_func)  -> mi_index_create / mi_index_add*
 *   advanced_rag.py:296,303  vector_store.as_retriever(k).invoke(query)    -> mi_index_search*
 *
 * Conventions
 *   - every function returns MI_OK (0) or a negative mi_status; no exceptions, no aborts;
 *     mi_last_error() returns a thread-local human-readable message for the last failure
 *     on the calling thread.
 *   - handles are opaque; each is freed only by its *_destroy.
 *   - the caller owns every buffer it passes in, and every output buffer.
 *   - one mi_ctx per GPU, one process per GPU (multi-GPU = one process per rank; the
 *     exchange step is a torch.distributed/RCCL all-gather in the host layer followed by
 *     mi_topk_merge_dev).  Every entry point selects its ctx's device itself
 *     (hipSetDevice) and takes the handle's mutex, so it may be called from arbitrary
 *     host threads (LangChain RunnableParallel workers, Gradio handlers).
 *   - "*_dev" variants take raw device pointers (e.g. torch.Tensor.data_ptr()) and a
 *     hipStream_t passed as void* (NULL = the ctx's own stream, MI_STREAM_LEGACY = the null stream); they enqueue work and
 *     return without a host sync unless noted.  Host-pointer variants are synchronous.
 *   - the library has NO CPU fallback: without a gfx950 device mi_init fails with
 *     MI_ERR_NO_DEVICE.
 *
 * Score / order semantics (identical in oracle/oracle_search.c, the CPU restatement):
 *   dot(q,x)   = fmaf chain over k = 0..d-1 starting from +0.0f, in fp32
 *                (for a BF16 index both q and x are first rounded to bf16, RNE).
 *   IP metric  : score = dot(q,x), best = largest.
 *   L2 metric  : score = max(0, fmaf(-2, dot(q,x), dot(q,q) + dot(x,x))), best = smallest.
 *   NaN scores rank as the worst possible value; ties rank the LOWER id first.
 *   Results are best-first; when fewer than k rows exist the tail is id = -1 and
 *   score = -inf (IP) / +inf (L2)  [faiss IndexFlat convention].
 */
#ifndef MI355RAG_H
#define MI355RAG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mi_ctx     mi_ctx;
typedef struct mi_index   mi_index;
typedef struct mi_encoder mi_encoder;

typedef enum {
    MI_OK               =  0,
    MI_ERR_INVALID_ARG  = -1,
    MI_ERR_NO_DEVICE    = -2,   /* no gfx950 GPU / HIP runtime failure at init */
    MI_ERR_HIP          = -3,   /* a HIP call failed; see mi_last_error() */
    MI_ERR_OOM          = -4,
    MI_ERR_UNSUPPORTED  = -5,   /* e.g. k > MI_MAX_K */
    MI_ERR_INTERNAL     = -6
} mi_status;

enum { MI_METRIC_IP = 0, MI_METRIC_L2 = 1 };
enum { MI_DTYPE_F32 = 0, MI_DTYPE_BF16 = 1 };

/* search path selection (mi_index_set_mode) */
enum {
    MI_MODE_AUTO  = 0,  /* MFMA scan + certificate where available, exact scan otherwise */
    MI_MODE_EXACT = 1,  /* always the canonical-order fp32 scan (also the fallback path) */
    MI_MODE_FAST  = 2   /* MFMA scan + certificate; error if the index cannot use it */
};

#define MI_MAX_K 2048
/* `stream` argument of the *_dev entry points: the legacy default ("null") stream.  NULL selects the ctx's own
 * (non-blocking) stream, so a caller whose tensors live on the null stream -- PyTorch's default stream has
 * handle 0 -- passes this instead.  Same value as hipStreamLegacy. */
#define MI_STREAM_LEGACY ((void*)1)

/* ------------------------------------------------------------------ context */

/* Opens device `device_id` (must be gfx950), creates the ctx stream.  No counterpart in
 * the reference (it is CPU-only, advanced_rag.py:106 "device": "cpu"). */
int  mi_init(int device_id, mi_ctx** out);
void mi_ctx_destroy(mi_ctx* ctx);
/* Blocks until all work enqueued on the ctx stream has finished. */
int  mi_ctx_synchronize(mi_ctx* ctx);
const char* mi_last_error(void);
const char* mi_version(void);

/* ------------------------------------------------------------------ flat index
 * Replaces faiss.IndexFlatL2 / IndexFlatIP as created by FAISS.from_documents
 * (advanced_rag.py:295; langchain-community 0.0.19 -> faiss-cpu 1.7.3, requirements.txt:3,6). */

int  mi_index_create(mi_ctx* ctx, int dim, int metric, int dtype, mi_index** out);
void mi_index_destroy(mi_index* idx);

/* Optional: pre-size device storage for n_rows rows in total (avoids regrowth copies). */
int  mi_index_reserve(mi_index* idx, int64_t n_rows);

/* Appends n rows (host, fp32, row-major [n][dim]); copies, caller keeps ownership.
 * index.add(np.array(embeddings, float32)) in the reference's FAISS wrapper. */
int  mi_index_add(mi_index* idx, int64_t n, const float* x_host);
/* Same from device memory; src_dtype MI_DTYPE_F32 or MI_DTYPE_BF16, row pitch = dim elements. */
int  mi_index_add_dev(mi_index* idx, int64_t n, const void* x_dev, int src_dtype, void* stream);
/* Appends n synthetic rows generated on device by the counter-based generator
 * (seed, global row number = id_base + current ntotal + i); rows are L2-normalised when
 * `normalize` != 0.  Bench/test input only. */
int  mi_index_add_synthetic(mi_index* idx, int64_t n, uint64_t seed, int normalize);

int64_t mi_index_ntotal(const mi_index* idx);
int  mi_index_dim(const mi_index* idx);
/* ids reported by search = id_base + local row number (row shards of one corpus). */
int  mi_index_set_id_base(mi_index* idx, int64_t id_base);
int  mi_index_set_mode(mi_index* idx, int mode);
/* Removes all rows, keeps storage (reset_app_updated, advanced_rag.py:435-437). */
int  mi_index_reset(mi_index* idx);

/* Copies rows [row0, row0+n) back as fp32 (the stored values, i.e. bf16-rounded for a BF16
 * index) — faiss reconstruct_n; used by the parity tests. */
int  mi_index_get_rows(mi_index* idx, int64_t row0, int64_t n, float* out_host);

/* Shard files (the reference never persists, advanced_rag.py:435-437): rows in the index's
 * own storage type -- fp32, or bf16 bit patterns as uint16 -- row pitch = dim elements, no widening on the way
 * out; and the matching add from host rows of either type (src_dtype MI_DTYPE_F32 / MI_DTYPE_BF16). */
int  mi_index_get_rows_raw(mi_index* idx, int64_t row0, int64_t n, void* out_host);
int  mi_index_add_raw(mi_index* idx, int64_t n, const void* x_host, int src_dtype);
int  mi_index_dtype(const mi_index* idx);

/* Exact top-k.  index.search(np.array([v], float32), k) in the reference
 * (advanced_rag.py:296 -> similarity_search -> faiss IndexFlat::search).
 * q_host [nq][dim] fp32; out_scores [nq][k] fp32; out_ids [nq][k] int64 (-1 padded). */
int  mi_index_search(mi_index* idx, int64_t nq, const float* q_host, int k,
                     float* out_scores, int64_t* out_ids);
/* Device-pointer variant: all three buffers in device memory.  Synchronises the stream once
 * internally (to read the certificate-failure count) before returning. */
int  mi_index_search_dev(mi_index* idx, int64_t nq, const float* q_dev, int k,
                         float* out_scores_dev, int64_t* out_ids_dev, void* stream);

/* Counters since creation: out[0] queries searched, out[1] queries answered by the MFMA
 * path with a passing certificate, out[2] queries re-run on the exact path, out[3] corpus
 * rows scanned by the last search. */
int  mi_index_stats(const mi_index* idx, int64_t out[4]);

/* Timing of the last search's dominant scan kernel, measured with HIP events on the stream
 * the kernel ran on: out_ms = total milliseconds, out_launches = launches it covers,
 * out_kind = 0 exact fp32 scan, 1 bf16 MFMA scan. */
int  mi_index_last_scan_time(const mi_index* idx, float* out_ms, int* out_launches, int* out_kind);

/* Merges `nlists` best-first lists per query into one best-first list of length k
 * (the step after the all-gather of per-GPU top-k).
 * in_scores [nq][nlists*k_in] fp32, in_ids same shape int64 (id < 0 = padding), all device.
 * Any nlists*k_in < 2^24 and k <= MI_MAX_K: more than 4096 entries per query are merged in passes
 * through scratch buffers owned by the ctx (allocated on first use). */
int  mi_topk_merge_dev(mi_ctx* ctx, int metric, int64_t nq, int nlists, int k_in,
                       const float* in_scores_dev, const int64_t* in_ids_dev,
                       int k, float* out_scores_dev, int64_t* out_ids_dev, void* stream);

/* The same for lists still packed as the all-gather leaves them (ONE collective).  A rank's block is
 * [nq][k] fp32 scores followed, at ids_offset, by [nq][k] int64 ids; block_bytes is the all-gather's per-rank
 * message.  A rank points mi_index_search_dev's outputs into its block, gathers the `nlists` blocks rank-major
 * into packed_dev (8-byte aligned) and merges them in place: no transpose, no second collective. */
int  mi_topk_packed_layout(int64_t nq, int k, size_t* ids_offset, size_t* block_bytes);
int  mi_topk_merge_packed_dev(mi_ctx* ctx, int metric, int64_t nq, int nlists, int k_in, const void* packed_dev,
                              int k, float* out_scores_dev, int64_t* out_ids_dev, void* stream);

/* ------------------------------------------------------------------ communicator (multi-GPU hosts without torch.distributed)
 * The reference has no multi-GPU path; rag_amd/distributed.py runs the sharded search over
 * torch.distributed.  These entry points give a C / Go / Java host the same three steps over RCCL (loaded with dlopen at
 * the first call; MI355RAG_RCCL overrides the library path).  One process per GPU: rank 0 calls mi_comm_unique_id and
 * hands the 128 bytes to the other ranks by its own means, every rank calls mi_comm_create (collective), then
 * mi_index_search_sharded_dev = mi_index_search_dev into the rank's packed block -> ONE ncclAllGather of block_bytes ->
 * mi_topk_merge_packed_dev; every rank returns the identical, exact answer.  Rank r's index holds the contiguous rows
 * [r * ceil(N / G), ...) with mi_index_set_id_base.  q_dev must be identical on every rank. */
typedef struct mi_comm mi_comm;
int  mi_comm_unique_id(void* out_id128);
int  mi_comm_create(mi_ctx* ctx, const void* id128, int nranks, int rank, mi_comm** out);
void mi_comm_destroy(mi_comm* comm);
int  mi_comm_size(const mi_comm* comm);
int  mi_comm_rank(const mi_comm* comm);
int  mi_comm_all_gather_dev(mi_comm* comm, const void* send_dev, void* recv_dev, size_t bytes_per_rank, void* stream);
int  mi_index_search_sharded_dev(mi_index* idx, mi_comm* comm, int64_t nq, const float* q_dev, int k,
                                 float* out_scores_dev, int64_t* out_ids_dev, void* stream);

/* ------------------------------------------------------------------ one index over several GPUs, one host process
 * advanced_rag.py is ONE Python process (module-global chain, :336-337) that builds one store (FAISS.from_documents, :295)
 * and searches it from whatever thread LangChain picks (:296, :409): a drop-in for it that uses the whole node cannot be
 * one-process-per-rank.  A mi_multi_index owns one ctx + one flat index per listed device (SURVEY.md section 8b sketched
 * this as mi_init(device_ids, n_devices)).  Rows are placed round-robin by blocks of MI_MULTI_BLOCK rows: global row g
 * lives on shard (g / block) % G, so appends keep working and shards stay balanced; ids reported are global row numbers.
 * A search broadcasts the queries (peer copies, or one host-to-device copy per GPU), runs mi_index_search_dev on every
 * shard from its own host thread, copies the G packed blocks peer-to-peer to the first device and merges them there with
 * mi_topk_merge_packed_dev -- no RCCL inside one process -- and returns the single-index answer bit for bit.
 * A device may be listed more than once (one shard per entry).  Every function takes the handle's mutex. */
#define MI_MULTI_BLOCK 16384
typedef struct mi_multi_index mi_multi_index;
int  mi_multi_index_create(const int* device_ids, int n_devices, int dim, int metric, int dtype, mi_multi_index** out);
void mi_multi_index_destroy(mi_multi_index* m);
/* Rows per placement block (default MI_MULTI_BLOCK); only while the index is empty. */
int  mi_multi_index_set_block_rows(mi_multi_index* m, int64_t rows);
int  mi_multi_index_devices(const mi_multi_index* m);
int64_t mi_multi_index_ntotal(const mi_multi_index* m);
int64_t mi_multi_index_shard_ntotal(const mi_multi_index* m, int shard);
int  mi_multi_index_set_mode(mi_multi_index* m, int mode);
int  mi_multi_index_reserve(mi_multi_index* m, int64_t n_rows);
/* index.add (advanced_rag.py:295): host fp32 rows [n][dim]; the shards receive their blocks concurrently. */
int  mi_multi_index_add(mi_multi_index* m, int64_t n, const float* x_host);
int  mi_multi_index_add_raw(mi_multi_index* m, int64_t n, const void* x_host, int src_dtype);
int  mi_multi_index_add_synthetic(mi_multi_index* m, int64_t n, uint64_t seed, int normalize);
int  mi_multi_index_reset(mi_multi_index* m);
/* Rows [row0, row0+n) in global order; raw != 0: the storage type (fp32, or bf16 bit patterns), else fp32. */
int  mi_multi_index_get_rows(mi_multi_index* m, int64_t row0, int64_t n, void* out_host, int raw);
/* index.search (advanced_rag.py:296): host buffers in and out. */
int  mi_multi_index_search(mi_multi_index* m, int64_t nq, const float* q_host, int k, float* out_scores, int64_t* out_ids);
/* q_dev and both outputs on the FIRST listed device; `stream` is a stream of that device.  Complete on return. */
int  mi_multi_index_search_dev(mi_multi_index* m, int64_t nq, const float* q_dev, int k,
                               float* out_scores_dev, int64_t* out_ids_dev, void* stream);
/* out[0] queries searched, out[1] (query, shard) searches certified by the MFMA path, out[2] re-run on the exact
 * path, out[3] rows. */
int  mi_multi_index_stats(const mi_multi_index* m, int64_t out[4]);

/* ------------------------------------------------------------------ encoder
 * Replaces HuggingFaceEmbeddings(...).embed_documents / embed_query
 * (advanced_rag.py:104-107; sentence-transformers -> transformers BertModel,
 * modeling_bert.py) from token ids onward: embeddings + LN, L x [QKV, softmax(QK^T/sqrt(dh)
 * + mask) V, out-proj + residual + LN, FFN1 + erf-GELU, FFN2 + residual + LN],
 * masked mean-pool, L2-normalise. */

typedef struct {
    int32_t vocab_size;               /* 30522 */
    int32_t hidden_size;              /* 384; must be a multiple of 128 and <= 1024 */
    int32_t num_hidden_layers;        /* 6 */
    int32_t num_attention_heads;      /* 12; hidden_size / heads must be 32 */
    int32_t intermediate_size;        /* 1536; multiple of 128 */
    int32_t max_position_embeddings;  /* 512 */
    int32_t type_vocab_size;          /* 2 */
    float   layer_norm_eps;           /* 1e-12 */
    int32_t normalize;                /* 1 = L2-normalise the pooled vector */
} mi_bert_config;

/* Number of fp32 elements mi_encoder_create expects in weights_blob for cfg (layout in
 * DESIGN.md "encoder weights blob"; rag_amd/encoder_weights.py packs it from a
 * safetensors / state_dict with the HuggingFace BERT parameter names). */
int64_t mi_encoder_blob_elems(const mi_bert_config* cfg);

int  mi_encoder_create(mi_ctx* ctx, const mi_bert_config* cfg,
                       const float* weights_blob_host, size_t n_elems, mi_encoder** out);
void mi_encoder_destroy(mi_encoder* enc);

/* ids, mask: host int32 [B][L] (mask 1 = token, 0 = padding); out_emb host fp32 [B][hidden]. */
int  mi_encoder_encode(mi_encoder* enc, int64_t B, int L, const int32_t* ids,
                       const int32_t* mask, float* out_emb);
int  mi_encoder_encode_dev(mi_encoder* enc, int64_t B, int L, const int32_t* ids_dev,
                           const int32_t* mask_dev, float* out_emb_dev, void* stream);
/* Ragged bulk embedding, the call FAISS.from_documents makes through embed_documents
 * (advanced_rag.py:295): n sequences as one flat token array; sequence i is
 * flat_ids[offsets[i] .. offsets[i+1]) (host int32 / int64 [n+1]); out_emb host fp32 [n][hidden]
 * in INPUT order.  The library does what sentence-transformers' encode loop does above the
 * boundary -- sort by length, cut into batches -- except that a batch is cut by a token budget
 * (64K tokens) instead of 32 rows, so bulk calls run the fused bulk kernels whatever the length
 * mix, and that the tokens of a batch are packed back to back instead of padded to the longest
 * sequence: no padding token is computed.  max_len > 0 truncates sequences (the tokenizer's
 * max_seq_length, 256 for all-MiniLM-L6-v2); max_batch > 0 caps the rows per batch
 * (encode_kwargs["batch_size"] when the caller sets it).  An empty sequence yields a zero vector. */
int  mi_encoder_encode_ragged(mi_encoder* enc, int64_t n, const int32_t* flat_ids, const int64_t* offsets,
                              int max_len, int max_batch, float* out_emb);

/* Cross-encoder scoring of (query, passage) pairs: the rerank batch BASELINE config 5 feeds
 * (README.md:27 and requirements.txt:24 name a reranker; advanced_rag.py has none).  Same encoder
 * forward with token type ids, then transformers' BertPooler + BertForSequenceClassification head:
 * logits = cls_w tanh(pooler_w h[CLS] + pooler_b) + cls_b.
 * mi_encoder_set_head: host fp32 pooler_w [H][H], pooler_b [H], cls_w [num_labels][H], cls_b [num_labels].
 * mi_encoder_classify: ids, type_ids (NULL = all 0), mask host int32 [B][L]; out_logits host fp32 [B][num_labels]. */
int  mi_encoder_set_head(mi_encoder* enc, const float* pooler_w, const float* pooler_b,
                         const float* cls_w, const float* cls_b, int num_labels);
int  mi_encoder_classify(mi_encoder* enc, int64_t B, int L, const int32_t* ids, const int32_t* type_ids,
                         const int32_t* mask, float* out_logits);
int  mi_encoder_classify_dev(mi_encoder* enc, int64_t B, int L, const int32_t* ids_dev, const int32_t* type_ids_dev,
                             const int32_t* mask_dev, float* out_logits_dev, void* stream);
/* The same for n pairs of different lengths as one flat token array (sequence i = flat_ids[offsets[i] .. offsets[i+1]),
 * token types alongside): passes are cut by the token budget and packed without padding, as in
 * mi_encoder_encode_ragged; out_logits host fp32 [n][num_labels] in INPUT order.  Pairs are never truncated here
 * (1 .. max_position_embeddings tokens each: the tokenizer above the boundary truncates and keeps the closing [SEP]). */
int  mi_encoder_classify_ragged(mi_encoder* enc, int64_t n, const int32_t* flat_ids, const int32_t* flat_type_ids,
                                const int64_t* offsets, int max_batch, float* out_logits);

/* Counters since creation: out[0] sequences encoded, out[1] tokens processed (padding included),
 * out[2] passes that ran the bulk kernel family (fused per-tile kernels, >= 12,288 tokens),
 * out[3] passes that ran the latency family (general GEMMs). */
int  mi_encoder_stats(const mi_encoder* enc, int64_t out[4]);
/* Allocation-free variant of mi_encoder_encode_dev.  No counterpart in the reference; it is for callers that
 * cannot allocate: code under HIP graph capture, tight serving loops that embed one query per call, harnesses
 * that forbid hipMalloc.  Same kernels and results; the activation workspace comes from a 16-MiB file-scope
 * device array of the library (one per GPU, held until the process ends) instead of hipMalloc, and the call
 * makes no hipMalloc / hipFree.  Inputs whose workspace does not fit (MiniLM-L6: more than ~2,000 tokens)
 * fail with MI_ERR_UNSUPPORTED.  All calls on one GPU share that array: calls that overlap on one GPU race. */
int  mi_encoder_encode_dev_noalloc(mi_encoder* enc, int64_t B, int L, const int32_t* ids_dev,
                                   const int32_t* mask_dev, float* out_emb_dev, void* stream);
/* Timing of the last encode pass (all kernels), HIP events on its stream. */
int  mi_encoder_last_time(const mi_encoder* enc, float* out_ms);

/* ---- host text path: WordPiece tokenizer (pure host code, no GPU call)
 * Replaces the tokenizer advanced_rag.py:104-107 reaches through HuggingFaceEmbeddings ->
 * sentence-transformers -> `tokenizers` (BertNormalizer, BertPreTokenizer, WordPiece, [CLS] .. [SEP], truncation):
 * embed_documents (advanced_rag.py:295) from List[str] was bound by it.  vocab: the bytes of vocab.txt (one entry per
 * line, line number = id).  encode_batch: text i = text[text_offsets[i] .. text_offsets[i + 1]); out_ids holds n rows of
 * max_len ids, row i filled with out_lens[i] of them ([CLS] pieces [SEP], cut to max_len); out_lens[i] = -1 marks a text
 * that is left to the reference tokenizer: one with a spelled-out special token ("[" A-Z+ "]"), and non-ASCII text while no
 * Unicode table is set, or with a character outside the table / of class "decline", or malformed UTF-8.  n_threads <= 0:
 * OMP_NUM_THREADS, else min(16, cores).
 * set_unicode: per-code-point tables for UTF-8 text, code points [0, n_codepoints).  map[cp] says what the normalizer
 * (clean-up, CJK spacing, accent stripping, lower-casing: all per character in BertNormalizer) turns an INPUT character
 * into: 0xFFFFFFFF = itself, else (offset into pool << 8) | number of bytes of UTF-8 there (0 bytes = the character is
 * dropped).  cls[cp] says what an OUTPUT character is to BertPreTokenizer: 0 word character, 1 space, 2 punctuation (a
 * word of its own), 3 decline (the whole text is handed back with -1).  The tables are copied; n_codepoints = 0 removes
 * them.  Not to be called while encode_batch / encode_pairs run on the same tokenizer.
 * encode_pairs: the cross-encoder's input (no counterpart in advanced_rag.py, which has no reranker: README.md:27 and
 * requirements.txt:24 name one; the framing is `tokenizers`' for a BERT pair): row i = [CLS] a_i [SEP] b_i [SEP], out_types
 * the segment ids (0 up to and including the first [SEP], 1 after), cut to max_len (>= 3) by the library's default
 * "longest_first" rule: the longer side gives way first, both get half of the room when both are long.  out_ids and
 * out_types hold n rows of max_len; out_lens[i] = -1 when either side is declined as above.  Equal neighbouring a_i
 * (one query against many passages) are tokenised once. */
typedef struct mi_wordpiece mi_wordpiece;
int  mi_wordpiece_create(const char* vocab, int64_t vocab_bytes, int lowercase, mi_wordpiece** out);
int  mi_wordpiece_set_unicode(mi_wordpiece* wp, int64_t n_codepoints, const uint8_t* cls, const uint32_t* map,
                              const char* pool, int64_t pool_bytes);
void mi_wordpiece_destroy(mi_wordpiece* wp);
int  mi_wordpiece_encode_batch(const mi_wordpiece* wp, const char* text, const int64_t* text_offsets, int64_t n,
                               int max_len, int n_threads, int32_t* out_ids, int64_t* out_lens);
int  mi_wordpiece_encode_pairs(const mi_wordpiece* wp, const char* text_a, const int64_t* offsets_a, const char* text_b,
                               const int64_t* offsets_b, int64_t n, int max_len, int n_threads, int32_t* out_ids,
                               int32_t* out_types, int64_t* out_lens);

#ifdef __cplusplus
}
#endif
#endif /* MI355RAG_H */
