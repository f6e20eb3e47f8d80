// encoder.hip — MiniLM/BERT-class bi-encoder forward (HuggingFaceEmbeddings, advanced_rag.py:104-107).
// PLACEHOLDER: the kernels land in the next milestone; until then the entry points fail loudly.
#include "kernels.h"

struct mi_encoder { int unused; };

extern "C" int64_t mi_encoder_blob_elems(const mi_bert_config* c)
{
    if (!c) return -1;
    const int64_t H = c->hidden_size, I = c->intermediate_size;
    int64_t n = (int64_t)c->vocab_size * H + (int64_t)c->max_position_embeddings * H + (int64_t)c->type_vocab_size * H + 2 * H;
    n += (int64_t)c->num_hidden_layers * (4 * (H * H + H) + 2 * H + (I * H + I) + (H * I + H) + 2 * H);
    return n;
}
extern "C" int mi_encoder_create(mi_ctx*, const mi_bert_config*, const float*, size_t, mi_encoder** out)
{
    if (out) *out = nullptr;
    mi_set_error("mi_encoder_create: encoder kernels not built yet");
    return MI_ERR_UNSUPPORTED;
}
extern "C" void mi_encoder_destroy(mi_encoder*) {}
extern "C" int mi_encoder_encode(mi_encoder*, int64_t, int, const int32_t*, const int32_t*, float*)
{
    mi_set_error("mi_encoder_encode: encoder kernels not built yet");
    return MI_ERR_UNSUPPORTED;
}
extern "C" int mi_encoder_encode_dev(mi_encoder*, int64_t, int, const int32_t*, const int32_t*, float*, void*)
{
    mi_set_error("mi_encoder_encode_dev: encoder kernels not built yet");
    return MI_ERR_UNSUPPORTED;
}
extern "C" int mi_encoder_last_time(const mi_encoder*, float* out_ms)
{
    if (out_ms) *out_ms = 0.0f;
    return MI_ERR_UNSUPPORTED;
}
