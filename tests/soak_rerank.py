"""Randomised parity soak of the cross-encoder path: random numbers of (query, passage) pairs of random lengths through
mi_encoder_classify_ragged (packed passes, both kernel families, random pass caps, 1 .. 3 labels) against the CPU oracle,
allowance as in tests/test_rerank_gpu.py (20 x 2^-9 x mean |pooler row| x max |classifier row|).  MiniLM-shaped layers, two
of them.  Runs for FUZZ_SECONDS (default 120); prints progress and one summary line; status 1 on a violation.
Test infrastructure (imports oracle/): not part of the product."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from rag_amd._capi import BertConfig
from rag_amd.embeddings import Encoder
from rag_amd.encoder_weights import random_blob, random_head

budget = float(os.environ.get("FUZZ_SECONDS", "120"))
seed0 = int(os.environ.get("FUZZ_SEED", "1"))
vals = (3000, 384, 2, 12, 1536, 512, 2, 1e-12, 0)
cfg, ocfg = BertConfig(*vals), oracle.BertConfig(*vals)
blob = random_blob(cfg, seed=seed0, init_range=0.05)
t_end = time.time() + budget
cases = fails = pairs = 0
worst = 0.0                                                  # worst error as a fraction of the allowance
it = 0
encs = {}
while time.time() < t_end:
    seed = seed0 * 100003 + it
    it += 1
    rng = np.random.default_rng(seed)
    nl = int(rng.integers(1, 4))
    if nl not in encs:
        e = Encoder(cfg, blob)
        h = random_head(cfg, num_labels=nl, seed=seed0 + nl, init_range=0.1)
        e.set_head(*h)
        encs[nl] = (e, h)
    enc, head = encs[nl]
    tol = 20.0 * 2.0 ** -9 * float(np.linalg.norm(head[0], axis=1).mean()) * float(np.linalg.norm(head[2], axis=1).max())
    n = int(rng.integers(1, 70))
    hi = int(rng.choice([6, 40, 130, 300, 512]))
    lens = rng.integers(1, hi + 1, size=n)
    L = int(lens.max())
    ids = np.zeros((n, L), dtype=np.int32); types = np.zeros((n, L), dtype=np.int32); mask = np.zeros((n, L), dtype=np.int32)
    for b in range(n):
        ids[b, :lens[b]] = rng.integers(4, 3000, size=lens[b])
        types[b, int(rng.integers(0, lens[b] + 1)):lens[b]] = 1
        mask[b, :lens[b]] = 1
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    flat_ids = np.concatenate([ids[b, :lens[b]] for b in range(n)])
    flat_types = np.concatenate([types[b, :lens[b]] for b in range(n)])
    os.environ["MI_ENC_LATENCY_TOKENS"] = str(rng.choice([0, 12288, 1 << 30]))
    os.environ["MI_QKV_FUSED_MIN_MEAN"] = str(rng.choice([0, 0, 128]))          # qkv_attn_kernel for every bulk pass up to 256 tokens / default
    got = enc.classify_ragged(flat_ids, flat_types, off, max_batch=int(rng.choice([0, 0, 3, 10])))
    want = oracle.classify(ocfg, blob, head, ids, types, mask)
    err = float(np.abs(got - want).max()) / tol
    worst = max(worst, err)
    cases += 1; pairs += n
    if cases % 10 == 0:
        print(json.dumps({"progress": cases, "pairs": pairs, "worst_error_over_allowance": round(worst, 3)}), flush=True)
    if not (err < 1.0) or got.shape != (n, nl) or not np.all(np.isfinite(got)):
        print(json.dumps({"seed": seed, "n": n, "hi": hi, "labels": nl, "error_over_allowance": err, "latency_tokens": os.environ["MI_ENC_LATENCY_TOKENS"]}), flush=True)
        fails += 1
        break
print(json.dumps({"cases": cases, "failures": fails, "pairs": pairs, "worst_error_over_allowance": round(worst, 3),
                  "seconds": round(budget - (t_end - time.time()), 1), "seed0": seed0}), flush=True)
sys.exit(1 if fails else 0)
