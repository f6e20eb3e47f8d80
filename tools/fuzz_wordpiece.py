"""CPU soak of the native WordPiece tokenizer's UTF-8 path (mi_wordpiece_*, rag_amd/csrc/wordpiece.hip) against `tokenizers`:
random strings over ASSIGNED code points of planes 0 and 1 (all scripts, marks, format characters, symbols), with a
vocabulary that holds the normalised form of three quarters of the alphabet so that most words match real pieces.
    python tools/fuzz_wordpiece.py [seeds=12] [texts per seed=20000]    -> one JSON line
Needs no GPU (the library is loaded for its host entry points only)."""
import json
import os
import random
import sys
import tempfile
import time
import unicodedata

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import test_wordpiece as tw  # noqa: E402
from tokenizers import BertWordPieceTokenizer  # noqa: E402
from rag_amd.embeddings import NativeWordPiece  # noqa: E402

CONFIGS = [dict(lowercase=True), dict(lowercase=False), dict(lowercase=True, strip_accents=False), dict(lowercase=False, strip_accents=True)]


def one_seed(seed, n_texts):
    rnd = random.Random(seed)
    kw = CONFIGS[seed % len(CONFIGS)]

    def rc():
        while True:
            r = rnd.random()
            cp = (rnd.randrange(0x20, 0x250) if r < 0.3 else rnd.randrange(0x250, 0x3000) if r < 0.6 else
                  rnd.randrange(0x3000, 0x10000) if r < 0.85 else rnd.randrange(0x10000, 0x20000))
            if not 0xD800 <= cp <= 0xDFFF and unicodedata.category(chr(cp)) != "Cn":
                return chr(cp)

    d = tempfile.mkdtemp()
    base = tw._unicode_vocab()
    nz = BertWordPieceTokenizer(tw._write_vocab(d), **kw)._tokenizer.normalizer
    alphabet = [rc() for _ in range(4000)]
    extra = set()
    for c in alphabet[:3000]:
        for ch in nz.normalize_str(c):
            if not ch.isspace():
                extra.update((ch, "##" + ch))
    have = set(base)
    vocab = base + [e for e in sorted(extra) if e not in have]
    p = os.path.join(d, "v.txt")
    with open(p, "w", encoding="utf-8") as f:
        f.write("\n".join(vocab) + "\n")
    tok = BertWordPieceTokenizer(p, **kw)._tokenizer
    tok.no_padding()
    tok.enable_truncation(max_length=64)
    nat = NativeWordPiece(tok, 64)
    texts = ["".join(rnd.choice(alphabet) if rnd.random() < 0.8 else rnd.choice(" \t ab,.") for _ in range(rnd.randint(1, 80)))
             for _ in range(n_texts)]
    t = time.time()
    flat, off = nat.encode_csr(texts)
    dt = time.time() - t
    ref = tok.encode_batch(texts)
    bad = [i for i, e in enumerate(ref) if flat[off[i]:off[i + 1]].tolist() != e.ids]
    if bad:
        i = bad[0]
        print("MISMATCH", kw, repr(texts[i]), flat[off[i]:off[i + 1]].tolist(), ref[i].ids, file=sys.stderr)
    unk = tok.token_to_id("[UNK]")
    return len(bad), nat.last_native_texts, float(np.mean(flat == unk)), dt


def main():
    seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    n_texts = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
    bad = native = 0
    unk = []
    for s in range(seeds):
        b, n, u, _ = one_seed(s, n_texts)
        bad += b
        native += n
        unk.append(u)
    print(json.dumps({"what": "wordpiece_utf8_fuzz", "seeds": seeds, "texts": seeds * n_texts, "native_texts": native, "mismatches": bad,
                      "unk_share": round(float(np.mean(unk)), 3), "configs": CONFIGS}))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
