"""CPU: the native WordPiece tokenizer (mi_wordpiece_*, rag_amd/csrc/wordpiece.hip; host code, no GPU call) against the
`tokenizers` library -- what advanced_rag.py:104-107 runs through sentence-transformers -- token for token.  No
vocabulary file exists offline, so the vocabulary is built here: letters, digits, punctuation, common English stems,
suffix pieces and the five specials.  Parity is pinned to `tokenizers` 's own output (the reference's dependency,
installed in this image), not to a restatement."""
import os
import random
import string

import numpy as np
import pytest

STEMS = ["the", "a", "an", "of", "and", "to", "in", "is", "it", "that", "retriev", "augment", "generat", "index", "vector",
         "search", "query", "token", "embed", "model", "chunk", "doc", "text", "data", "graph", "flat", "exact", "top", "score",
         "hello", "world", "fox", "quick", "brown", "jump", "over", "lazy", "dog", "cat", "sat", "mat", "red", "un", "re", "pre",
         "trans", "form", "er", "s", "x", "y", "z", "ab", "abc", "abcd", "12", "123", "2024", "0", "1", "2", "3", "4", "5", "6", "7", "8", "9"]
SUFFIXES = ["s", "ed", "ing", "er", "al", "ion", "ation", "ly", "e", "or", "ment", "able", "0", "1", "2", "3", "5", "9", "00", "a", "b", "c",
            "d", "x", "y", "z", "t", "n", "r", "ve", "ive"]


def _vocab():
    v = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"]
    v += list(string.ascii_lowercase) + list(string.punctuation.replace("[", "").replace("]", "")) + ["[", "]"]
    v += [s for s in STEMS if s not in v]
    v += ["##" + s for s in SUFFIXES] + ["##" + c for c in string.ascii_lowercase if c not in SUFFIXES]
    assert len(set(v)) == len(v)
    return v


def _write_vocab(tmp_path):
    p = os.path.join(str(tmp_path), "vocab.txt")
    with open(p, "w") as f:
        f.write("\n".join(_vocab()) + "\n")
    return p


def _random_texts(n, seed):
    rnd = random.Random(seed)
    words = STEMS + [s + t for s in STEMS[:30] for t in SUFFIXES[:8]]
    out = []
    for _ in range(n):
        parts = []
        for _ in range(rnd.randint(0, 60)):
            r = rnd.random()
            if r < 0.55:
                w = rnd.choice(words)
            elif r < 0.70:
                w = "".join(rnd.choice(string.ascii_letters + string.digits) for _ in range(rnd.randint(1, 14)))
            elif r < 0.80:
                w = rnd.choice(string.punctuation) * rnd.randint(1, 3)
            elif r < 0.85:
                w = "".join(rnd.choice(string.ascii_lowercase) for _ in range(rnd.randint(95, 105)))          # around the 100-character rule
            elif r < 0.90:
                w = rnd.choice(words) + rnd.choice(["\x00", "\x01", "\x0b", "\x0c", "\x1f", "\x7f"]) + rnd.choice(words)   # control characters vanish
            elif r < 0.95:
                w = rnd.choice(words).upper() + rnd.choice([",", ".", "'s", "-", "_", "(", ")", "?!"]) + rnd.choice(words).capitalize()
            else:
                w = rnd.choice(["[SEP]", "[MASK]", "[x]", "[AB", "AB]", "[]", "[A1]", "#", "##", "##ing"])
            parts.append(w)
        sep = rnd.choice([" ", " ", "  ", "\t", "\n", " \r\n "])
        out.append(sep.join(parts) + rnd.choice(["", " ", "\n"]))
    return out


@pytest.mark.parametrize("max_len", [16, 64, 256])
def test_native_wordpiece_equals_tokenizers(tmp_path, max_len):
    from tokenizers import BertWordPieceTokenizer
    from rag_amd.embeddings import NativeWordPiece
    tok = BertWordPieceTokenizer(_write_vocab(tmp_path), lowercase=True)._tokenizer
    tok.no_padding()
    tok.enable_truncation(max_length=max_len)
    nat = NativeWordPiece(tok, max_len)
    texts = ["", " ", "a", "Hello, World!", "x" * 100, "x" * 101, "the " * 400, "unretrievable transformers", "\x00\x01", "a\x0bb c\x7fd",
             "café naïve", "中文 text", "tab\tsep\nnl\r\ncr", "[SEP] inside", "[UNK]", "###", "don't stop-me_now (ok)"]
    texts += _random_texts(1500, seed=max_len)
    flat, off = nat.encode_csr(texts)
    assert flat.dtype == np.int32 and off.dtype == np.int64 and off[0] == 0 and off[-1] == flat.size
    assert 0.5 * len(texts) < nat.last_native_texts < len(texts)      # most texts take the native path, some the reference one
    ref = tok.encode_batch([t.replace("\n", " ") for t in texts])
    bad = [i for i, e in enumerate(ref) if flat[off[i]:off[i + 1]].tolist() != e.ids]
    assert not bad, (len(bad), repr(texts[bad[0]]), flat[off[bad[0]]:off[bad[0] + 1]].tolist(), ref[bad[0]].ids)
    assert max(len(e.ids) for e in ref) == max_len                     # the truncation path was exercised
    # threads do not change the answer
    nat1 = NativeWordPiece(tok, max_len, n_threads=1)
    f1, o1 = nat1.encode_csr(texts)
    assert np.array_equal(f1, flat) and np.array_equal(o1, off)


def test_cased_pipeline_and_refusals(tmp_path):
    from tokenizers import BertWordPieceTokenizer, Tokenizer
    from tokenizers.models import BPE
    from rag_amd.embeddings import NativeWordPiece, _is_standard_bert_pipeline
    vp = _write_vocab(tmp_path)
    cased = BertWordPieceTokenizer(vp, lowercase=False)._tokenizer
    cased.no_padding()
    cased.enable_truncation(max_length=32)
    assert _is_standard_bert_pipeline(cased) is False
    nat = NativeWordPiece(cased, 32)
    texts = ["Hello World", "the The THE", "ab Ab"] + _random_texts(200, seed=9)
    flat, off = nat.encode_csr(texts)
    for i, e in enumerate(cased.encode_batch([t.replace("\n", " ") for t in texts])):
        assert flat[off[i]:off[i + 1]].tolist() == e.ids, repr(texts[i])
    # another pipeline is refused, and so is a vocabulary without the specials
    assert _is_standard_bert_pipeline(Tokenizer(BPE())) is None
    with pytest.raises(ValueError):
        NativeWordPiece(Tokenizer(BPE()), 32)
    from rag_amd import _capi, MI355Error
    import ctypes
    h = ctypes.c_void_p()
    blob = b"a\nb\nc"
    with pytest.raises(MI355Error):
        _capi.check(_capi.load().mi_wordpiece_create(blob, len(blob), 1, ctypes.byref(h)))


def test_embeddings_object_uses_it_only_when_it_may(tmp_path, monkeypatch):
    """MI355Embeddings builds the native tokenizer for a stock BERT directory (no GPU is touched before the encoder is
    created, so only the tokenizer half is exercised here)."""
    from rag_amd.embeddings import NativeWordPiece, _load_tokenizer
    with open(os.path.join(str(tmp_path), "vocab.txt"), "w") as f:
        f.write("\n".join(_vocab()) + "\n")
    tok = _load_tokenizer(str(tmp_path), 48)
    nat = NativeWordPiece(tok, 48)
    texts = _random_texts(300, seed=4)
    flat, off = nat.encode_csr(texts)
    ref = [e.ids for e in tok.encode_batch([t.replace("\n", " ") for t in texts])]
    assert [flat[off[i]:off[i + 1]].tolist() for i in range(len(texts))] == ref


# ---- UTF-8 text through the per-code-point tables (mi_wordpiece_set_unicode) ----

CYR = "абвгдежзиклмнопрстуфхцчшщыьэюя"
GRK = "αβγδεζηθικλμνξοπρστυφχψω"
ACC = "éèêëáàâäíìîïóòôöúùûüñçÅØßœæĲİıŁđ"
UPUNCT = "“”‘’—–…«»¿¡§•·、。「」％"
USPACE = [" ", " ", "　", " ", " "]
UCTRL = ["​", "­", "﻿", "\u0085", "�", "‍", "؜"]
MARKS = ["́", "̈", "̧", "ा", "்", "⃝"]          # Mn, Mn, Mn, Mc, Mn (virama), Me
CJK = "中文字符検索データ語"
HANGUL = "한국어검색문서"
EMOJI = ["\U0001F600", "\U0001F680", "\U0001D400", "\U00020000", "\U0002A6D6", "\U000E0001"]   # plane 1 (native), planes 2 / 14 (declined)


def _unicode_vocab():
    import unicodedata
    v = _vocab()
    extra = list(CYR) + ["##" + c for c in CYR] + ["при", "вет", "##вет", "мир", "поиск"]
    extra += list(GRK) + ["##" + c for c in GRK] + ["δοκιμη", "##ιμη"]                              # (accent-stripped forms)
    extra += list(ACC.lower()) + ["##" + c for c in "éèïöüñç"] + ["café", "caf", "##é", "naïve", "İ", "Å"]
    extra += list(UPUNCT) + list(CJK) + ["##" + c for c in CJK[:3]]
    extra += list(HANGUL) + ["##" + c for c in HANGUL[:3]]
    jamo = sorted({c for s in HANGUL for c in unicodedata.normalize("NFD", s)})
    extra += jamo + ["##" + c for c in jamo]
    extra += [EMOJI[0], EMOJI[2], "##" + EMOJI[0], "́", "##́", "é", "##ा"]
    for e in extra:
        if e not in v:
            v.append(e)
    assert len(set(v)) == len(v)
    return v


def _unicode_texts(n, seed):
    rnd = random.Random(seed)
    pools = [CYR, GRK, ACC, string.ascii_letters, CJK, HANGUL]
    fixed = ["при", "вет", "привет", "мир", "поиск", "Привет", "δοκιμή", "Δοκιμή", "café", "Café", "naïve", "CAFÉ", "İstanbul", "Ångström"]
    out = []
    for _ in range(n):
        parts = []
        tame = rnd.random() < 0.7                                  # nothing in it that makes the native path hand the text back
        for _ in range(rnd.randint(0, 50)):
            r = rnd.random()
            if tame and r >= 0.79 and not 0.85 <= r < 0.90:
                r *= 0.79
            if r < 0.30:
                w = rnd.choice(fixed + STEMS[:20])
            elif r < 0.55:
                p = rnd.choice(pools)
                w = "".join(rnd.choice(p) for _ in range(rnd.randint(1, 9)))
            elif r < 0.65:
                w = rnd.choice(STEMS) + rnd.choice(UPUNCT) + rnd.choice(fixed)
            elif r < 0.72:
                w = rnd.choice(fixed) + rnd.choice(UCTRL) + rnd.choice(STEMS)                     # format / control characters vanish
            elif r < 0.79:
                w = rnd.choice(fixed) + rnd.choice(USPACE) + rnd.choice(fixed)                    # Unicode spaces separate
            elif r < 0.85:
                w = rnd.choice("aeoux" + CYR[:5]) + "".join(rnd.choice(MARKS) for _ in range(rnd.randint(1, 2))) + rnd.choice(STEMS)
            elif r < 0.90:
                p = rnd.choice([CYR, ACC, GRK, HANGUL])
                w = "".join(rnd.choice(p) for _ in range(rnd.randint(97, 103)))                   # around the 100-CHARACTER rule
            elif r < 0.95:
                w = rnd.choice(EMOJI) * rnd.randint(1, 2) + rnd.choice(["", "x", "я"])
            else:
                w = rnd.choice(["[SEP]", "[MASK]é", "№", "½", "ﬁ", "Ⅳ", "㎏", "ǆ", "ß", "ẞ", "ſ", "Σ", "ς"])
            parts.append(w)
        out.append(rnd.choice([" ", " ", "\n", " ", " \t"]).join(parts))
    return out


@pytest.mark.parametrize("kw", [dict(lowercase=True), dict(lowercase=False), dict(lowercase=True, strip_accents=False),
                                dict(lowercase=False, strip_accents=True), dict(lowercase=True, handle_chinese_chars=False)],
                         ids=["uncased", "cased", "uncased-keep-accents", "cased-strip-accents", "uncased-no-cjk-spacing"])
def test_unicode_text_equals_tokenizers(tmp_path, kw):
    from tokenizers import BertWordPieceTokenizer
    from rag_amd.embeddings import NativeWordPiece
    p = os.path.join(str(tmp_path), "vocab.txt")
    with open(p, "w", encoding="utf-8") as f:
        f.write("\n".join(_unicode_vocab()) + "\n")
    tok = BertWordPieceTokenizer(p, **kw)._tokenizer
    tok.no_padding()
    tok.enable_truncation(max_length=48)
    nat = NativeWordPiece(tok, 48)
    assert nat.unicode is None                                  # tables are built at the first non-ASCII text, not before
    texts = ["café naïve", "中文 text", "“hello” — world…", "Ångström İstanbul", "x́y", "한국어 abc", "a​b", "nb sp",
             "\U0001F600 emoji", "\U00020000 ext b", "Привет мир", "ẹ́", "é" * 100, "é" * 101, "я" * 300, "при вет " * 100,
             "aा்b"]
    texts += _unicode_texts(1500, seed=len(kw) * 7 + int(kw["lowercase"]))
    flat, off = nat.encode_csr(texts)
    assert nat.unicode is True
    ref = tok.encode_batch([t.replace("\n", " ") for t in texts])
    bad = [i for i, e in enumerate(ref) if flat[off[i]:off[i + 1]].tolist() != e.ids]
    assert not bad, (len(bad), repr(texts[bad[0]]), flat[off[bad[0]]:off[bad[0] + 1]].tolist(), ref[bad[0]].ids, ref[bad[0]].tokens)
    n_ascii = sum(t.isascii() for t in texts)
    # most non-ASCII texts went native (fewer where lower-casing leaves a combining dot behind: "İ" -> "i̇")
    share = 0.25 if kw.get("strip_accents") is False and kw["lowercase"] else 0.5
    assert nat.last_native_texts > n_ascii + share * (len(texts) - n_ascii), (nat.last_native_texts, n_ascii)
    assert nat.last_native_texts < len(texts)                   # ... and some were handed back (planes 2+, surviving marks, "[SEP]")
    unk = tok.token_to_id("[UNK]")
    assert np.mean(flat == unk) < 0.5                           # the vocabulary matched real pieces: not a test of [UNK] == [UNK]
    # one thread gives the same answer
    f1, o1 = NativeWordPiece(tok, 48, n_threads=1).encode_csr(texts)
    assert np.array_equal(f1, flat) and np.array_equal(o1, off)


def test_unicode_tables_are_validated_and_bad_utf8_is_declined(tmp_path):
    import ctypes
    from rag_amd import _capi, MI355Error
    lib = _capi.load()
    blob = "\n".join(_unicode_vocab()).encode("utf-8")
    h = ctypes.c_void_p()
    _capi.check(lib.mi_wordpiece_create(blob, len(blob), 1, ctypes.byref(h)))
    N = 0x800

    def set_tables(cls, mp, pool):
        return lib.mi_wordpiece_set_unicode(h, cls.size, cls.ctypes.data_as(ctypes.c_void_p), mp.ctypes.data_as(ctypes.c_void_p), pool, len(pool))

    def encode(bs):
        off = np.array([0, len(bs)], dtype=np.int64)
        ids = np.zeros((1, 16), dtype=np.int32)
        ln = np.zeros(1, dtype=np.int64)
        _capi.check(lib.mi_wordpiece_encode_batch(h, bs, off.ctypes.data_as(ctypes.c_void_p), 1, 16, 1, ids.ctypes.data_as(ctypes.c_void_p),
                                                  ln.ctypes.data_as(ctypes.c_void_p)))
        return int(ln[0]), ids[0, :max(int(ln[0]), 0)].tolist()

    try:
        assert encode("é".encode())[0] == -1                                       # no table yet: declined
        cls = np.zeros(N, dtype=np.uint8)
        cls[0x20] = 1
        mp = np.full(N, 0xFFFFFFFF, dtype=np.uint32)
        bad_cls = cls.copy(); bad_cls[5] = 4
        with pytest.raises(MI355Error):
            _capi.check(set_tables(bad_cls, mp, b""))
        bad_mp = mp.copy(); bad_mp[0xE9] = (3 << 8) | 2                            # past a 4-byte pool
        with pytest.raises(MI355Error):
            _capi.check(set_tables(cls, bad_mp, b"abcd"))
        bad_mp[0xE9] = (0 << 8) | 2
        with pytest.raises(MI355Error):
            _capi.check(set_tables(cls, bad_mp, b"\xc3\x28zz"))                    # not UTF-8
        with pytest.raises(MI355Error):
            _capi.check(set_tables(cls, bad_mp, "ࠀ".encode()[:2] + b"zz"))    # a character cut short
        with pytest.raises(MI355Error):
            _capi.check(set_tables(cls[:64], mp[:64], b""))                        # a table that does not even hold ASCII
        mp[0xE9] = (0 << 8) | 1                                                    # é -> e
        _capi.check(set_tables(cls, mp, b"e"))
        v = _unicode_vocab()
        n, ids = encode("café x".encode())
        assert n == 5 and [v[i] for i in ids] == ["[CLS]", "caf", "##e", "x", "[SEP]"]   # through the table: é -> e
        n, ids = encode("é".encode())
        assert [v[i] for i in ids] == ["[CLS]", "e", "[SEP]"]
        for bs in [b"a\xc3", b"a\xc3\x28", b"\xed\xa0\x80", b"\xc0\xaf", b"\xf4\x90\x80\x80", b"\xff", "ࠀ".encode()]:   # cut short, bad
            assert encode(bs)[0] == -1, bs                                         # continuation, surrogate, overlong, > 10FFFF, beyond the table
        _capi.check(lib.mi_wordpiece_set_unicode(h, 0, None, None, None, 0))
        assert encode("é".encode())[0] == -1
    finally:
        lib.mi_wordpiece_destroy(h)


# ---- (query, passage) pairs for the cross-encoder (mi_wordpiece_encode_pairs) ----

@pytest.mark.parametrize("max_len", [3, 4, 5, 16, 17, 64])
def test_pairs_equal_tokenizers(tmp_path, max_len):
    """[CLS] a [SEP] b [SEP], segment ids, and the library's "longest_first" cut (which side gives way, who keeps the odd
    piece) against tokenizers' own encode(a, b), for even and odd room and both orders of long / short."""
    from tokenizers import BertWordPieceTokenizer
    from rag_amd.embeddings import NativeWordPiece
    p = os.path.join(str(tmp_path), "vocab.txt")
    with open(p, "w", encoding="utf-8") as f:
        f.write("\n".join(_unicode_vocab()) + "\n")
    tok = BertWordPieceTokenizer(p, lowercase=True)._tokenizer
    tok.no_padding()
    tok.enable_truncation(max_length=max_len)
    nat = NativeWordPiece(tok, max_len)
    assert nat.pairs_supported()
    rnd = random.Random(max_len)
    words = STEMS + ["café", "привет", "—", "retrievers", "x" * 101]

    def txt(k):
        return " ".join(rnd.choice(words) for _ in range(k))

    pairs = [(txt(rnd.randint(0, max_len)), txt(rnd.randint(0, 2 * max_len))) for _ in range(1500)]
    pairs += [(txt(rnd.randint(0, 2 * max_len)), txt(rnd.randint(0, max_len))) for _ in range(1500)]
    q = txt(5)
    pairs += [(q, txt(rnd.randint(0, 30))) for _ in range(100)]                    # one query, many passages
    pairs += [("", ""), ("a", ""), ("", "a"), ("[SEP] x", "y"), ("x", "[MASK]"), ("\U00020000", "y"), ("y", "\U00020000 z")]
    flat, types, off = nat.encode_pairs_csr(pairs)
    assert flat.dtype == np.int32 and types.dtype == np.int32 and off[-1] == flat.size == types.size
    assert nat.last_native_texts == len(pairs) - 4                                 # the last four were handed back
    for i, (a, b) in enumerate(pairs):
        e = tok.encode(a, b)
        assert flat[off[i]:off[i + 1]].tolist() == e.ids and types[off[i]:off[i + 1]].tolist() == e.type_ids, (max_len, a, b)
    assert int(np.diff(off).max()) == max_len
    f1, t1, o1 = NativeWordPiece(tok, max_len, n_threads=1).encode_pairs_csr(pairs)
    assert np.array_equal(f1, flat) and np.array_equal(t1, types) and np.array_equal(o1, off)
    # a tokenizer that cuts pairs another way is not restated
    tok.enable_truncation(max_length=max_len, strategy="only_first")
    assert not NativeWordPiece(tok, max_len).pairs_supported()
    tok.enable_truncation(max_length=max_len, strategy="only_second")
    with pytest.raises(ValueError):
        NativeWordPiece(tok, max_len)
