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
// Text with bytes >= 0x80 (UTF-8) goes the same way once mi_wordpiece_set_unicode has handed over per-code-point tables:
// `map` (what one input character becomes after clean-up, CJK spacing, accent stripping and lower-casing: itself, nothing,
// or up to a few characters) and `cls` (what an OUTPUT character is to the splitter: word character, space, punctuation, or
// "decline").  Both steps of the reference are per character, so tables reproduce them exactly; the caller fills them from
// the reference library's own normalizer and pre-tokenizer (rag_amd/embeddings.py), which keeps Unicode versions out of
// this file.  Words are then matched at character boundaries and the 100-character rule counts characters.
// Left to the reference library (the text's length comes back as -1 and the caller tokenises it there): non-ASCII text while
// no table is set, a character beyond the table or of class "decline", malformed UTF-8, and any text that spells a special
// token ("[SEP]", "[MASK]": any "[" A-Z+ "]"), which the reference matches in the raw string.  Parity with `tokenizers` is
// held by tests/test_wordpiece.py.
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
    // Unicode tables (mi_wordpiece_set_unicode); empty = ASCII only
    std::vector<uint8_t> ucls;           // per OUTPUT code point: 0 word character, 1 space, 2 punctuation, 3 decline
    std::vector<uint32_t> umap;          // per INPUT code point: 0xFFFFFFFF itself, else (offset into upool << 8) | bytes
    std::string upool;
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

// the same for a word of UTF-8 text: ends[i] = byte offset just past character i (nc characters)
inline void word_pieces_u(const mi_wordpiece* w, const char* s, const uint32_t* ends, size_t nc, std::vector<int32_t>& out, uint64_t hcont0)
{
    if ((int)nc > w->max_chars) { out.push_back(w->unk); return; }
    const size_t mark = out.size();
    uint64_t hs[128];
    size_t start = 0;                                          // in characters
    while (start < nc) {
        const size_t b0 = start ? ends[start - 1] : 0;
        uint64_t h = start ? hcont0 : FNV0;
        size_t b = b0;
        for (size_t e = start; e < nc; ++e) { for (; b < ends[e]; ++b) h = fnv(h, (unsigned char)s[b]); hs[e - start] = h; }
        int32_t got = -1;
        size_t end = nc;
        for (; end > start; --end) {
            got = find(w, hs[end - start - 1], s + b0, ends[end - 1] - b0, start > 0);
            if (got >= 0) break;
        }
        if (got < 0) { out.resize(mark); out.push_back(w->unk); return; }
        out.push_back(got);
        start = end;
    }
}

// one UTF-8 character at t[i..n): its code point and length, or 0 length if malformed (overlong, surrogate, > 0x10FFFF, cut short)
inline uint32_t utf8_at(const unsigned char* t, size_t i, size_t n, uint32_t* len)
{
    const uint32_t c = t[i];
    if (c < 0x80) { *len = 1; return c; }
    uint32_t need, cp, lo;
    if ((c & 0xE0) == 0xC0) { need = 1; cp = c & 0x1F; lo = 0x80; }
    else if ((c & 0xF0) == 0xE0) { need = 2; cp = c & 0x0F; lo = 0x800; }
    else if ((c & 0xF8) == 0xF0) { need = 3; cp = c & 0x07; lo = 0x10000; }
    else { *len = 0; return 0; }
    if (i + need >= n) { *len = 0; return 0; }                 // cut short
    for (uint32_t k = 1; k <= need; ++k) {
        const uint32_t x = t[i + k];
        if ((x & 0xC0) != 0x80) { *len = 0; return 0; }
        cp = (cp << 6) | (x & 0x3F);
    }
    if (cp < lo || cp > 0x10FFFF || (cp >= 0xD800 && cp <= 0xDFFF)) { *len = 0; return 0; }
    *len = need + 1;
    return cp;
}

struct Scratch { std::vector<int32_t> tmp, tmp_a; std::string word; std::vector<uint32_t> ends; bool a_ok = false; };

// UTF-8 text through the tables: its pieces (at least `budget` of them if it has that many) into tmp; false = declined
bool pieces_u(const mi_wordpiece* w, const char* t, size_t n, size_t budget, Scratch& sc, std::vector<int32_t>& tmp, uint64_t hcont0)
{
    const unsigned char* u = (const unsigned char*)t;
    const size_t ncp = w->umap.size();
    std::string& word = sc.word;
    std::vector<uint32_t>& ends = sc.ends;
    tmp.clear(); word.clear(); ends.clear();
    auto flush = [&]() { if (!ends.empty()) { word_pieces_u(w, word.data(), ends.data(), ends.size(), tmp, hcont0); word.clear(); ends.clear(); } };
    // one OUTPUT character (bytes p[0:len), code point cp) into the splitter; false = decline
    auto feed = [&](uint32_t cp, const char* p, uint32_t len) -> bool {
        if (cp >= ncp) return false;
        switch (w->ucls[cp]) {
        case 0: word.append(p, len); ends.push_back((uint32_t)word.size()); return true;
        case 1: flush(); return true;
        case 2: { flush(); const uint32_t e = len; word_pieces_u(w, p, &e, 1, tmp, hcont0); return true; }
        default: return false;
        }
    };
    for (size_t i = 0; i < n && tmp.size() < budget;) {
        uint32_t len;
        const uint32_t cp = utf8_at(u, i, n, &len);
        if (len == 0 || cp >= ncp) return false;
        const uint32_t m = w->umap[cp];
        if (m == 0xFFFFFFFFu) { if (!feed(cp, t + i, len)) return false; }
        else {
            const char* p = w->upool.data() + (m >> 8);
            const size_t pn = m & 0xFF;
            for (size_t k = 0; k < pn;) {                        // (validated when the tables were set)
                uint32_t l2;
                const uint32_t c2 = utf8_at((const unsigned char*)p, k, pn, &l2);
                if (!feed(c2, p + k, l2)) return false;
                k += l2;
            }
        }
        i += len;
    }
    if (tmp.size() < budget) flush();
    return true;
}

// the pieces of one text (no [CLS] / [SEP]) into tmp, stopping once `budget` of them exist (pieces past a cut are never
// needed); false: declined (see the header of this file)
bool pieces(const mi_wordpiece* w, const char* t, size_t n, size_t budget, Scratch& sc, std::vector<int32_t>& tmp, uint64_t hcont0)
{
    std::string& word = sc.word;
    bool high = false;
    for (size_t i = 0; i < n; ++i) {
        const unsigned char c = (unsigned char)t[i];
        if (c >= 0x80) { high = true; continue; }
        if (c == '[') {                                        // "[SEP]", "[MASK]", ... in the text: the reference matches its special
            size_t j = i + 1;                                  // tokens in the raw string before anything else -- leave the text to it
            while (j < n && t[j] >= 'A' && t[j] <= 'Z') ++j;
            if (j > i + 1 && j < n && t[j] == ']') return false;
        }
    }
    if (high) return !w->umap.empty() && pieces_u(w, t, n, budget, sc, tmp, hcont0);
    tmp.clear();
    word.clear();
    auto flush = [&]() { if (!word.empty()) { word_pieces(w, word.data(), word.size(), tmp, hcont0); word.clear(); } };
    for (size_t i = 0; i < n && tmp.size() < budget; ++i) {
        unsigned char c = (unsigned char)t[i];
        if (c == ' ' || c == '\t' || c == '\n' || c == '\r') { flush(); continue; }
        if (c < 32 || c == 127) continue;                      // control characters vanish (they do not separate words)
        if (is_punct(c)) { flush(); const char pc = (char)c; word_pieces(w, &pc, 1, tmp, hcont0); continue; }
        if (w->lowercase && c >= 'A' && c <= 'Z') c = (unsigned char)(c + 32);
        word.push_back((char)c);
    }
    if (tmp.size() < budget) flush();
    return true;
}

// [CLS] pieces[: max_len - 2] [SEP] into out; -1: declined, else the number of ids written (<= max_len)
int64_t encode_one(const mi_wordpiece* w, const char* t, size_t n, int max_len, Scratch& sc, int32_t* out, uint64_t hcont0)
{
    const size_t budget = max_len >= 2 ? (size_t)max_len - 2 : 0;
    if (!pieces(w, t, n, budget, sc, sc.tmp, hcont0)) return -1;
    const size_t m = std::min(sc.tmp.size(), budget);
    int64_t k = 0;
    if (max_len >= 1) out[k++] = w->cls;
    for (size_t i = 0; i < m; ++i) out[k++] = sc.tmp[i];
    if (max_len >= 2) out[k++] = w->sep;
    return k;
}

// [CLS] A [SEP] B [SEP] with segment ids 0 0 0 1 1, cut to max_len the way `tokenizers` cuts a pair ("longest_first",
// truncation.rs): with room = max_len - 3 and n1 <= n2 the two piece counts, nothing is cut if they fit; else the longer
// one alone gives way while the shorter fits in half the room, else both are cut to room / 2 (the longer keeps the odd one).
// The rule needs the TRUE counts of both sides (which of two over-long sides is "longer" decides who gets the odd piece),
// so neither side stops early.  -1: one side was declined.
int64_t encode_pair(const mi_wordpiece* w, const char* ta, size_t na, const char* tb, size_t nb, int max_len, Scratch& sc,
                    bool reuse_a, int32_t* out, int32_t* types, uint64_t hcont0)
{
    const size_t all = (size_t)-1;
    if (!reuse_a) sc.a_ok = pieces(w, ta, na, all, sc, sc.tmp_a, hcont0);
    if (!sc.a_ok) return -1;
    if (!pieces(w, tb, nb, all, sc, sc.tmp, hcont0)) return -1;
    size_t n1 = sc.tmp_a.size(), n2 = sc.tmp.size();
    const size_t room = (size_t)max_len - 3;
    if (n1 + n2 > room) {
        const bool swap = n1 > n2;
        if (swap) std::swap(n1, n2);
        n2 = n1 > room ? n1 : std::max(n1, room - n1);
        if (n1 + n2 > room) { n1 = room / 2; n2 = n1 + room % 2; }
        if (swap) std::swap(n1, n2);
        n1 = std::min(n1, sc.tmp_a.size());
        n2 = std::min(n2, sc.tmp.size());
    }
    int64_t k = 0;
    out[k] = w->cls; types[k++] = 0;
    for (size_t i = 0; i < n1; ++i) { out[k] = sc.tmp_a[i]; types[k++] = 0; }
    out[k] = w->sep; types[k++] = 0;
    for (size_t i = 0; i < n2; ++i) { out[k] = sc.tmp[i]; types[k++] = 1; }
    out[k] = w->sep; types[k++] = 1;
    return k;
}

int pool_threads(int n_threads, int64_t n)
{
    int T = n_threads;
    if (T <= 0) {
        const char* e = getenv("OMP_NUM_THREADS");
        T = e ? atoi(e) : 0;
        if (T <= 0) { T = (int)std::thread::hardware_concurrency(); if (T > 16) T = 16; }
    }
    if (T < 1) T = 1;
    if ((int64_t)T > (n + 15) / 16) T = (int)((n + 15) / 16);
    return T;
}

template <class F> void run_pool(int T, F&& work)
{
    if (T == 1) { work(); return; }
    std::vector<std::thread> pool;
    pool.reserve((size_t)T - 1);
    for (int t = 1; t < T; ++t) pool.emplace_back(work);
    work();
    for (auto& th : pool) th.join();
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

extern "C" int mi_wordpiece_set_unicode(mi_wordpiece* w, int64_t n_codepoints, const uint8_t* cls, const uint32_t* map,
                                        const char* pool, int64_t pool_bytes)
{
    MI_REQUIRE(w, "mi_wordpiece_set_unicode: NULL tokenizer");
    if (n_codepoints == 0) { w->ucls.clear(); w->umap.clear(); w->upool.clear(); return MI_OK; }
    MI_REQUIRE(cls && map && n_codepoints >= 128 && n_codepoints <= 0x110000 && pool_bytes >= 0 && pool_bytes < (1 << 24) && (pool || pool_bytes == 0),
               "mi_wordpiece_set_unicode: tables out of range (128 <= n_codepoints <= 0x110000, pool < 16 MiB)");
    for (int64_t cp = 0; cp < n_codepoints; ++cp) {
        MI_REQUIRE(cls[cp] <= 3, "mi_wordpiece_set_unicode: cls[%lld] = %d is no class", (long long)cp, (int)cls[cp]);
        const uint32_t m = map[cp];
        if (m == 0xFFFFFFFFu) continue;
        const size_t off = m >> 8, len = m & 0xFF;
        MI_REQUIRE((int64_t)(off + len) <= pool_bytes, "mi_wordpiece_set_unicode: map[%lld] points past the pool", (long long)cp);
        for (size_t k = 0; k < len;) {
            uint32_t l2;
            const uint32_t c2 = utf8_at((const unsigned char*)pool + off, k, len, &l2);
            MI_REQUIRE(l2 != 0 && (int64_t)c2 < n_codepoints, "mi_wordpiece_set_unicode: map[%lld] is not UTF-8 of characters inside the table", (long long)cp);
            k += l2;
        }
    }
    w->ucls.assign(cls, cls + n_codepoints);
    w->umap.assign(map, map + n_codepoints);
    w->upool.assign(pool ? pool : "", (size_t)pool_bytes);
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
    const int T = pool_threads(n_threads, n);
    const uint64_t hcont0 = fnv_str("##", 2);
    std::atomic<int64_t> next{0};
    run_pool(T, [&]() {
        Scratch sc;
        sc.tmp.reserve((size_t)max_len + 64);
        for (;;) {
            const int64_t i0 = next.fetch_add(16);
            if (i0 >= n) break;
            const int64_t i1 = std::min(n, i0 + 16);
            for (int64_t i = i0; i < i1; ++i)
                out_lens[i] = encode_one(w, text + text_offsets[i], (size_t)(text_offsets[i + 1] - text_offsets[i]), max_len, sc,
                                         out_ids + i * (int64_t)max_len, hcont0);
        }
    });
    return MI_OK;
}

extern "C" int mi_wordpiece_encode_pairs(const mi_wordpiece* w, const char* text_a, const int64_t* offsets_a, const char* text_b,
                                         const int64_t* offsets_b, int64_t n, int max_len, int n_threads, int32_t* out_ids,
                                         int32_t* out_types, int64_t* out_lens)
{
    MI_REQUIRE(w && (n == 0 || (offsets_a && offsets_b && out_ids && out_types && out_lens)), "mi_wordpiece_encode_pairs: NULL argument");
    MI_REQUIRE(n >= 0 && max_len >= 3 && max_len <= 65536, "mi_wordpiece_encode_pairs: n / max_len out of range (max_len >= 3)");
    MI_REQUIRE(n == 0 || ((text_a || offsets_a[n] == offsets_a[0]) && (text_b || offsets_b[n] == offsets_b[0])), "mi_wordpiece_encode_pairs: NULL text");
    for (int64_t i = 0; i < n; ++i)
        MI_REQUIRE(offsets_a[i + 1] >= offsets_a[i] && offsets_b[i + 1] >= offsets_b[i], "mi_wordpiece_encode_pairs: offsets must not decrease");
    if (n == 0) return MI_OK;
    const int T = pool_threads(n_threads, n);
    const uint64_t hcont0 = fnv_str("##", 2);
    std::atomic<int64_t> next{0};
    run_pool(T, [&]() {
        Scratch sc;
        for (;;) {
            const int64_t i0 = next.fetch_add(16);
            if (i0 >= n) break;
            const int64_t i1 = std::min(n, i0 + 16);
            for (int64_t i = i0; i < i1; ++i) {
                const char* ta = text_a + offsets_a[i];
                const size_t na = (size_t)(offsets_a[i + 1] - offsets_a[i]);
                // one query against many passages: the first text of a pair is tokenised once per run of equal texts
                const bool same = i > i0 && na == (size_t)(offsets_a[i] - offsets_a[i - 1]) && memcmp(ta, text_a + offsets_a[i - 1], na) == 0;
                out_lens[i] = encode_pair(w, ta, na, text_b + offsets_b[i], (size_t)(offsets_b[i + 1] - offsets_b[i]), max_len, sc, same,
                                          out_ids + i * (int64_t)max_len, out_types + i * (int64_t)max_len, hcont0);
            }
        }
    });
    return MI_OK;
}
