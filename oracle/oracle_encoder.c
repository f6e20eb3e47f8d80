/*
 * oracle_encoder.c — CPU fp32 restatement of the bi-encoder forward the reference runs.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under rag_amd/ (the product) may import, link or call
 * this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
 *
 * What it restates
 *   advanced_rag.py:104-107  HuggingFaceEmbeddings("sentence-transformers/all-MiniLM-L6-v2", cpu)
 *   advanced_rag.py:295      embed_documents (inside FAISS.from_documents)
 *   advanced_rag.py:296/303  embed_query (inside the retriever)
 * The arithmetic lives in sentence-transformers>=2.4.0 -> transformers>=4.34.0 BertModel
 * (requirements.txt:25,49), which the reference does not vendor.  The reference holds no
 * fixtures for it: PARITY UNPINNED by the reference.  It is pinned instead against the
 * transformers BertModel installed in the build container (5.15.0; modeling_bert.py), built
 * from a local BertConfig with seeded weights: tests/golden/make_golden.py writes the
 * fixtures, tests/test_oracle_encoder.py checks this file against them.
 *
 * Follows transformers/models/bert/modeling_bert.py:
 *   :98-107   embeddings = word + token_type(0) + position, LayerNorm
 *   :111-136  softmax(Q K^T * dh^-0.5 + mask) V        (mask additive 0 / finfo.min)
 *   :175-177  per-head split of the Q/K/V linears
 *   :289-293  attention output dense + residual + LayerNorm
 *   :334-337  intermediate dense + erf-GELU (activations.py GELUActivation)
 *   :347-351  output dense + residual + LayerNorm
 *   :96-100   token type ids: zeros for a bi-encoder, the segment ids of a cross-encoder pair
 *   :451-463  BertPooler: tanh(W_p h[:, 0] + b_p)
 *   :1118-1119 BertForSequenceClassification: logits = W_c pooled + b_c (dropout = identity in eval)
 * and sentence-transformers' Pooling(mean) + Normalize:
 *   sum_t h_t m_t / max(sum_t m_t, 1e-9);  x / max(||x||_2, 1e-12).
 *
 * Weights blob (fp32, this order; linears are [out][in] as torch.nn.Linear stores them):
 *   word_emb[V][H] pos_emb[P][H] type_emb[T][H] emb_ln_g[H] emb_ln_b[H]
 *   per layer: Wq[H][H] bq[H] Wk[H][H] bk[H] Wv[H][H] bv[H] Wo[H][H] bo[H] ln1_g[H] ln1_b[H]
 *              Wi[I][H] bi[I] Wd[H][I] bd[H] ln2_g[H] ln2_b[H]
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
    int32_t vocab_size, hidden_size, num_hidden_layers, num_attention_heads, intermediate_size,
            max_position_embeddings, type_vocab_size;
    float layer_norm_eps;
    int32_t normalize;
} oracle_bert_config;

int64_t oracle_encoder_blob_elems(const oracle_bert_config* c) {
    const int64_t H = c->hidden_size, I = c->intermediate_size;
    int64_t n = (int64_t)c->vocab_size * H + (int64_t)c->max_position_embeddings * H +
                (int64_t)c->type_vocab_size * H + 2 * H;
    n += (int64_t)c->num_hidden_layers * (4 * (H * H + H) + 2 * H + (I * H + I) + (H * I + H) + 2 * H);
    return n;
}

/* y[M][N] = x[M][K] * W[N][K]^T + b ; rows blocked by 4 so each W row is reused from L1 */
static void linear(const float* x, const float* W, const float* b, float* y, int M, int K, int N) {
    for (int m0 = 0; m0 < M; m0 += 4) {
        const int mr = (M - m0 < 4) ? (M - m0) : 4;
        for (int n = 0; n < N; ++n) {
            const float* w = W + (size_t)n * K;
            float acc[4] = {0, 0, 0, 0};
            for (int i = 0; i < mr; ++i) {
                const float* xr = x + (size_t)(m0 + i) * K;
                float s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0, s5 = 0, s6 = 0, s7 = 0;
                int k = 0;
                for (; k + 8 <= K; k += 8) {
                    s0 += xr[k] * w[k];         s1 += xr[k + 1] * w[k + 1];
                    s2 += xr[k + 2] * w[k + 2]; s3 += xr[k + 3] * w[k + 3];
                    s4 += xr[k + 4] * w[k + 4]; s5 += xr[k + 5] * w[k + 5];
                    s6 += xr[k + 6] * w[k + 6]; s7 += xr[k + 7] * w[k + 7];
                }
                float s = ((s0 + s1) + (s2 + s3)) + ((s4 + s5) + (s6 + s7));
                for (; k < K; ++k) s += xr[k] * w[k];
                acc[i] = s;
            }
            for (int i = 0; i < mr; ++i) y[(size_t)(m0 + i) * N + n] = acc[i] + b[n];
        }
    }
}

static void layer_norm(float* x, const float* g, const float* b, int M, int H, float eps) {
    for (int m = 0; m < M; ++m) {
        float* r = x + (size_t)m * H;
        double mean = 0, var = 0;
        for (int i = 0; i < H; ++i) mean += r[i];
        mean /= H;
        for (int i = 0; i < H; ++i) { const double dlt = r[i] - mean; var += dlt * dlt; }
        var /= H;
        const float inv = (float)(1.0 / sqrt(var + (double)eps));
        for (int i = 0; i < H; ++i) r[i] = (float)(r[i] - mean) * inv * g[i] + b[i];
    }
}

/* One chunk (sequence) forward.  ids/mask [L]; out [H].  Optional taps (may be NULL):
 * tap_emb [L][H] after the embedding LayerNorm, tap_layers [nl][L][H] after each layer. */
static void encode_one(const oracle_bert_config* c, const float* blob, const int32_t* ids, const int32_t* types,
                       const int32_t* mask, int L, float* out, float* tap_emb, float* tap_layers, float* cls_out)
{
    const int H = c->hidden_size, I = c->intermediate_size, NH = c->num_attention_heads;
    const int DH = H / NH;
    const float* word = blob;
    const float* pos = word + (size_t)c->vocab_size * H;
    const float* type = pos + (size_t)c->max_position_embeddings * H;
    const float* eg = type + (size_t)c->type_vocab_size * H;
    const float* eb = eg + H;
    const float* p = eb + H;

    float* h = (float*)malloc((size_t)L * H * sizeof(float));
    float* q = (float*)malloc((size_t)L * H * sizeof(float));
    float* k = (float*)malloc((size_t)L * H * sizeof(float));
    float* v = (float*)malloc((size_t)L * H * sizeof(float));
    float* ctx = (float*)malloc((size_t)L * H * sizeof(float));
    float* t = (float*)malloc((size_t)L * H * sizeof(float));
    float* mid = (float*)malloc((size_t)L * I * sizeof(float));
    float* prob = (float*)malloc((size_t)L * sizeof(float));

    for (int i = 0; i < L; ++i)
        for (int j = 0; j < H; ++j)
            h[(size_t)i * H + j] = (word[(size_t)ids[i] * H + j] + type[(size_t)(types ? types[i] : 0) * H + j]) + pos[(size_t)i * H + j];
    layer_norm(h, eg, eb, L, H, c->layer_norm_eps);
    if (tap_emb) memcpy(tap_emb, h, (size_t)L * H * sizeof(float));

    const float scaling = 1.0f / sqrtf((float)DH);
    for (int l = 0; l < c->num_hidden_layers; ++l) {
        const float *Wq = p, *bq = Wq + (size_t)H * H, *Wk = bq + H, *bk = Wk + (size_t)H * H,
                    *Wv = bk + H, *bv = Wv + (size_t)H * H, *Wo = bv + H, *bo = Wo + (size_t)H * H,
                    *g1 = bo + H, *b1 = g1 + H, *Wi = b1 + H, *bi = Wi + (size_t)I * H,
                    *Wd = bi + I, *bd = Wd + (size_t)H * I, *g2 = bd + H, *b2 = g2 + H;
        p = b2 + H;
        linear(h, Wq, bq, q, L, H, H);
        linear(h, Wk, bk, k, L, H, H);
        linear(h, Wv, bv, v, L, H, H);
        for (int hd = 0; hd < NH; ++hd) {
            for (int i = 0; i < L; ++i) {
                float mx = -INFINITY;
                for (int j = 0; j < L; ++j) {
                    float s = 0;
                    for (int e = 0; e < DH; ++e) s += q[(size_t)i * H + hd * DH + e] * k[(size_t)j * H + hd * DH + e];
                    s = s * scaling + (mask[j] ? 0.0f : -FLT_MAX);
                    prob[j] = s;
                    if (s > mx) mx = s;
                }
                float den = 0;
                for (int j = 0; j < L; ++j) { prob[j] = expf(prob[j] - mx); den += prob[j]; }
                for (int e = 0; e < DH; ++e) {
                    float s = 0;
                    for (int j = 0; j < L; ++j) s += prob[j] * v[(size_t)j * H + hd * DH + e];
                    ctx[(size_t)i * H + hd * DH + e] = s / den;
                }
            }
        }
        linear(ctx, Wo, bo, t, L, H, H);
        for (size_t i = 0; i < (size_t)L * H; ++i) h[i] = t[i] + h[i];
        layer_norm(h, g1, b1, L, H, c->layer_norm_eps);
        linear(h, Wi, bi, mid, L, H, I);
        for (size_t i = 0; i < (size_t)L * I; ++i) {
            const float x = mid[i];
            mid[i] = x * 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
        }
        linear(mid, Wd, bd, t, L, I, H);
        for (size_t i = 0; i < (size_t)L * H; ++i) h[i] = t[i] + h[i];
        layer_norm(h, g2, b2, L, H, c->layer_norm_eps);
        if (tap_layers) memcpy(tap_layers + (size_t)l * L * H, h, (size_t)L * H * sizeof(float));
    }

    if (cls_out) memcpy(cls_out, h, (size_t)H * sizeof(float));     /* final hidden state of the [CLS] token (position 0) */
    if (!out) { free(h); free(q); free(k); free(v); free(ctx); free(t); free(mid); free(prob); return; }

    /* masked mean pool + L2 normalise */
    float msum = 0;
    for (int i = 0; i < L; ++i) msum += (float)(mask[i] != 0);
    if (msum < 1e-9f) msum = 1e-9f;
    double nrm = 0;
    for (int j = 0; j < H; ++j) {
        float s = 0;
        for (int i = 0; i < L; ++i) if (mask[i]) s += h[(size_t)i * H + j];
        out[j] = s / msum;
        nrm += (double)out[j] * out[j];
    }
    if (c->normalize) {
        float d = (float)sqrt(nrm);
        if (d < 1e-12f) d = 1e-12f;
        for (int j = 0; j < H; ++j) out[j] /= d;
    }
    free(h); free(q); free(k); free(v); free(ctx); free(t); free(mid); free(prob);
}

/* ids, mask [B][L] int32; out [B][H].  Returns 0 or -1 (bad token id / L too long). */
int oracle_encode(const oracle_bert_config* c, const float* blob, int64_t B, int L,
                  const int32_t* ids, const int32_t* mask, float* out, int nthreads)
{
    if (L <= 0 || L > c->max_position_embeddings || c->hidden_size % c->num_attention_heads) return -1;
    for (int64_t i = 0; i < B * L; ++i) if (ids[i] < 0 || ids[i] >= c->vocab_size) return -1;
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#else
    (void)nthreads;
#endif
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
    for (int64_t b = 0; b < B; ++b)
        encode_one(c, blob, ids + b * L, NULL, mask + b * L, L, out + b * c->hidden_size, NULL, NULL, NULL);
    return 0;
}

/* Cross-encoder logits: encoder forward with token type ids, BertPooler, classifier.
 * head = pooler_w[H][H] pooler_b[H] cls_w[nl][H] cls_b[nl]; types may be NULL (all 0); out [B][nl]. */
int oracle_classify(const oracle_bert_config* c, const float* blob, const float* head, int nlabels, int64_t B, int L,
                    const int32_t* ids, const int32_t* types, const int32_t* mask, float* out, int nthreads)
{
    const int H = c->hidden_size;
    if (L <= 0 || L > c->max_position_embeddings || H % c->num_attention_heads || nlabels < 1) return -1;
    for (int64_t i = 0; i < B * L; ++i) {
        if (ids[i] < 0 || ids[i] >= c->vocab_size) return -1;
        if (types && (types[i] < 0 || types[i] >= c->type_vocab_size)) return -1;
    }
    const float *Wp = head, *bp = Wp + (size_t)H * H, *Wc = bp + H, *bc = Wc + (size_t)nlabels * H;
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#else
    (void)nthreads;
#endif
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
    for (int64_t b = 0; b < B; ++b) {
        float* cls = (float*)malloc((size_t)2 * H * sizeof(float));
        float* pooled = cls + H;
        encode_one(c, blob, ids + b * L, types ? types + b * L : NULL, mask + b * L, L, NULL, NULL, NULL, cls);
        for (int o = 0; o < H; ++o) {
            float s = 0;
            for (int j = 0; j < H; ++j) s += Wp[(size_t)o * H + j] * cls[j];
            pooled[o] = tanhf(s + bp[o]);
        }
        for (int o = 0; o < nlabels; ++o) {
            float s = 0;
            for (int j = 0; j < H; ++j) s += Wc[(size_t)o * H + j] * pooled[j];
            out[b * nlabels + o] = s + bc[o];
        }
        free(cls);
    }
    return 0;
}

/* Single sequence with per-stage taps, for stage-by-stage comparison with the HIP kernels. */
int oracle_encode_taps(const oracle_bert_config* c, const float* blob, int L, const int32_t* ids,
                       const int32_t* mask, float* out, float* tap_emb, float* tap_layers)
{
    if (L <= 0 || L > c->max_position_embeddings) return -1;
    for (int i = 0; i < L; ++i) if (ids[i] < 0 || ids[i] >= c->vocab_size) return -1;
    encode_one(c, blob, ids, NULL, mask, L, out, tap_emb, tap_layers, NULL);
    return 0;
}
