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
