// Host-side WordPiece tokenizer for the text path in front of the encoder (pure host code; no kernel in this file).
//
// Replaces, for ASCII input, what advanced_rag.py:104-107 reaches through HuggingFaceEmbeddings -> sentence-transformers
// -> `tokenizers` (BertNormalizer + BertPreTokenizer + WordPiece + [CLS] .. [SEP] + truncation): the bulk embedding of
// List[str] was bound by that library (29-31k texts/s of 256 tokens on 16 host cores against 140k chunks/s on the GPU).
// The algorithm is the published one (BERT tokenization.py / tokenizers' bert.rs + wordpiece.rs), restated:
//   clean      drop NUL and control characters (all of 0x00-0x1F and 0x7F but \t \n \r); \t \n \r and space separate words
//   lowercase  A-Z -> a-z when the model is uncased
//   split      every ASCII punctuation character (33-47, 58-64, 91-96, 123-126) is a word of its own
//   WordPiece  per word, greedy longest match first; pieces after the first carry the prefix "##"; a word of more than 100
//              characters, or one with a stretch no vocabulary entry matches, is ONE [UNK]
//   frame      [CLS] pieces[: max_len - 2] [SEP]
// A text with any byte >= 0x80 is NOT handled here (accent stripping, CJK spacing and Unicode categories stay with the
// reference library), nor is one that spells a special token ("[SEP]", "[MASK]": any "[" A-Z+ "]"), which the reference
// matches in the raw string: its length comes back as -1 and the caller tokenises it there.  Parity with `tokenizers` on ASCII
// text is held by tests/test_wordpiece.py.
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/mi355rag.h"
#include "common.h"

struct mi_wordpiece {
    std::string pool;                    // vocabulary entries back to back
    std::vector<uint32_t> off, len;      // per id
    std::vector<int32_t> table;          // open addressing, id or -1
    uint32_t mask = 0;
    int32_t unk = -1, cls = -1, sep = -1;
    int lowercase = 1;
    int max_chars = 100;
};

namespace {
constexpr uint64_t FNV0 = 1469598103934665603ull, FNVP = 1099511628211ull;
inline uint64_t fnv(uint64_t h, unsigned char c) { return (h ^ c) * FNVP; }
inline uint64_t fnv_str(const char* s, size_t n, uint64_t h = FNV0) { for (size_t i = 0; i < n; ++i) h = fnv(h, (unsigned char)s[i]); return h; }

inline bool is_punct(unsigned char c) { return (c >= 33 && c <= 47) || (c >= 58 && c <= 64) || (c >= 91 && c <= 96) || (c >= 123 && c <= 126); }

// id of the entry equal to (prefix ## if cont) + s[0:n], or -1
inline int32_t find(const mi_wordpiece* w, uint64_t h, const char* s, size_t n, bool cont)
{
    const size_t want = n + (cont ? 2 : 0);
    for (uint32_t p = (uint32_t)(h ^ (h >> 32)) & w->mask;; p = (p + 1) & w->mask) {
        const int32_t id = w->table[p];
        if (id < 0) return -1;
        if (w->len[(size_t)id] != want) continue;
        const char* e = w->pool.data() + w->off[(size_t)id];
        if (cont) { if (e[0] == '#' && e[1] == '#' && memcmp(e + 2, s, n) == 0) return id; }
        else if (memcmp(e, s, n) == 0) return id;
    }
}

// one word (already cleaned and lower-cased, no space, no punctuation inside unless it is a single punctuation character)
inline void word_pieces(const mi_wordpiece* w, const char* s, size_t L, std::vector<int32_t>& out, uint64_t hcont0)
{
    if ((int)L > w->max_chars) { out.push_back(w->unk); return; }
    const size_t mark = out.size();
    uint64_t hs[128];
    size_t start = 0;
    while (start < L) {
        uint64_t h = start ? hcont0 : FNV0;                    // hash of "##" when the piece continues a word
        for (size_t e = start; e < L; ++e) { h = fnv(h, (unsigned char)s[e]); hs[e - start] = h; }
        int32_t got = -1;
        size_t end = L;
        for (; end > start; --end) {
            got = find(w, hs[end - start - 1], s + start, end - start, start > 0);
            if (got >= 0) break;
        }
        if (got < 0) { out.resize(mark); out.push_back(w->unk); return; }
        out.push_back(got);
        start = end;
    }
}

// -1: the text holds a byte >= 0x80; else the number of ids written to out (<= max_len)
int64_t encode_one(const mi_wordpiece* w, const char* t, size_t n, int max_len, std::vector<int32_t>& tmp, std::string& word, int32_t* out, uint64_t hcont0)
{
    for (size_t i = 0; i < n; ++i) {
        const unsigned char c = (unsigned char)t[i];
        if (c >= 0x80) return -1;
        if (c == '[') {                                        // "[SEP]", "[MASK]", ... in the text: the reference matches its special
            size_t j = i + 1;                                  // tokens in the raw string before anything else -- leave the text to it
            while (j < n && t[j] >= 'A' && t[j] <= 'Z') ++j;
            if (j > i + 1 && j < n && t[j] == ']') return -1;
        }
    }
    tmp.clear();
    word.clear();
    const size_t budget = max_len >= 2 ? (size_t)max_len - 2 : 0;
    auto flush = [&]() { if (!word.empty()) { word_pieces(w, word.data(), word.size(), tmp, hcont0); word.clear(); } };
    for (size_t i = 0; i < n && tmp.size() < budget; ++i) {      // (pieces past the budget are cut anyway: stop early)
        unsigned char c = (unsigned char)t[i];
        if (c == ' ' || c == '\t' || c == '\n' || c == '\r') { flush(); continue; }
        if (c < 32 || c == 127) continue;                      // control characters vanish (they do not separate words)
        if (is_punct(c)) { flush(); const char pc = (char)c; word_pieces(w, &pc, 1, tmp, hcont0); continue; }
        if (w->lowercase && c >= 'A' && c <= 'Z') c = (unsigned char)(c + 32);
        word.push_back((char)c);
    }
    if (tmp.size() < budget) flush();
    const size_t m = std::min(tmp.size(), budget);
    int64_t k = 0;
    if (max_len >= 1) out[k++] = w->cls;
    for (size_t i = 0; i < m; ++i) out[k++] = tmp[i];
    if (max_len >= 2) out[k++] = w->sep;
    return k;
}
}  // namespace

extern "C" int mi_wordpiece_create(const char* vocab, int64_t vocab_bytes, int lowercase, mi_wordpiece** out)
{
    MI_REQUIRE(vocab && vocab_bytes > 0 && out, "mi_wordpiece_create: NULL or empty vocabulary");
    mi_wordpiece* w = new mi_wordpiece();
    w->lowercase = lowercase ? 1 : 0;
    w->pool.assign(vocab, (size_t)vocab_bytes);
    size_t b = 0;
    for (size_t i = 0; i <= (size_t)vocab_bytes; ++i) {
        if (i == (size_t)vocab_bytes || vocab[i] == '\n') {
            if (i == (size_t)vocab_bytes && b == i) break;       // no entry after a trailing newline
            size_t e = i;
            if (e > b && vocab[e - 1] == '\r') --e;
            w->off.push_back((uint32_t)b);
            w->len.push_back((uint32_t)(e - b));
            b = i + 1;
        }
    }
    const size_t V = w->off.size();
    size_t cap = 16;
    while (cap < 2 * V + 2) cap <<= 1;
    w->table.assign(cap, -1);
    w->mask = (uint32_t)(cap - 1);
    for (size_t id = 0; id < V; ++id) {
        if (w->len[id] == 0) continue;                           // a gap in the id space
        const char* s = w->pool.data() + w->off[id];
        const uint64_t h = fnv_str(s, w->len[id]);
        uint32_t p = (uint32_t)(h ^ (h >> 32)) & w->mask;
        bool dup = false;
        while (w->table[p] >= 0) {                               // a repeated entry keeps its first id (as a dict built in order would not: reject)
            const int32_t o = w->table[p];
            if (w->len[(size_t)o] == w->len[id] && memcmp(w->pool.data() + w->off[(size_t)o], s, w->len[id]) == 0) { dup = true; break; }
            p = (p + 1) & w->mask;
        }
        if (dup) { delete w; mi_set_error("mi_wordpiece_create: vocabulary entry %zu repeats an earlier one", id); return MI_ERR_INVALID_ARG; }
        w->table[p] = (int32_t)id;
    }
    w->unk = find(w, fnv_str("[UNK]", 5), "[UNK]", 5, false);
    w->cls = find(w, fnv_str("[CLS]", 5), "[CLS]", 5, false);
    w->sep = find(w, fnv_str("[SEP]", 5), "[SEP]", 5, false);
    if (w->unk < 0 || w->cls < 0 || w->sep < 0) { delete w; mi_set_error("mi_wordpiece_create: the vocabulary lacks [UNK], [CLS] or [SEP]"); return MI_ERR_INVALID_ARG; }
    *out = w;
    return MI_OK;
}

extern "C" void mi_wordpiece_destroy(mi_wordpiece* w) { delete w; }

extern "C" int mi_wordpiece_encode_batch(const mi_wordpiece* w, const char* text, const int64_t* text_offsets, int64_t n,
                                         int max_len, int n_threads, int32_t* out_ids, int64_t* out_lens)
{
    MI_REQUIRE(w && (n == 0 || (text_offsets && out_ids && out_lens)), "mi_wordpiece_encode_batch: NULL argument");
    MI_REQUIRE(n >= 0 && max_len >= 2 && max_len <= 65536, "mi_wordpiece_encode_batch: n / max_len out of range");
    MI_REQUIRE(n == 0 || text || text_offsets[n] == text_offsets[0], "mi_wordpiece_encode_batch: NULL text");
    for (int64_t i = 0; i < n; ++i) MI_REQUIRE(text_offsets[i + 1] >= text_offsets[i], "mi_wordpiece_encode_batch: offsets must not decrease");
    if (n == 0) return MI_OK;
    int T = n_threads;
    if (T <= 0) {
        const char* e = getenv("OMP_NUM_THREADS");
        T = e ? atoi(e) : 0;
        if (T <= 0) { T = (int)std::thread::hardware_concurrency(); if (T > 16) T = 16; }
    }
    if (T < 1) T = 1;
    if ((int64_t)T > (n + 15) / 16) T = (int)((n + 15) / 16);
    const uint64_t hcont0 = fnv_str("##", 2);
    std::atomic<int64_t> next{0};
    auto work = [&]() {
        std::vector<int32_t> tmp;
        std::string word;
        tmp.reserve((size_t)max_len + 64);
        for (;;) {
            const int64_t i0 = next.fetch_add(16);
            if (i0 >= n) break;
            const int64_t i1 = std::min(n, i0 + 16);
            for (int64_t i = i0; i < i1; ++i)
                out_lens[i] = encode_one(w, text + text_offsets[i], (size_t)(text_offsets[i + 1] - text_offsets[i]), max_len, tmp, word,
                                         out_ids + i * (int64_t)max_len, hcont0);
        }
    };
    if (T == 1) { work(); return MI_OK; }
    std::vector<std::thread> pool;
    pool.reserve((size_t)T - 1);
    for (int t = 1; t < T; ++t) pool.emplace_back(work);
    work();
    for (auto& th : pool) th.join();
    return MI_OK;
}
