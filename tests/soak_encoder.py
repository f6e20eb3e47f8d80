"""Randomised parity soak of the encoder: ragged batches (lengths 0 .. beyond max_len, shuffled), the padded entry point,
both kernel families (MI_ENC_LATENCY_TOKENS flipped per call), random pass caps -- the HIP path (C-ABI) against the CPU
oracle, 1 - cos <= 1e-3 per sequence (north_star's tolerance).  MiniLM-shaped layers (hidden 384, 12 heads, FFN 1536: the
shapes the fused kernels are built for), two of them so the oracle keeps up.  Runs for FUZZ_SECONDS (default 240), prints
progress and one summary line; a violation is printed with its seed and ends the run with status 1.
Test infrastructure (imports oracle/): not part of the product."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from rag_amd._capi import BertConfig
from rag_amd.embeddings import Encoder
from rag_amd.encoder_weights import random_blob

budget = float(os.environ.get("FUZZ_SECONDS", "240"))
seed0 = int(os.environ.get("FUZZ_SEED", "1"))
vals = (3000, 384, 2, 12, 1536, 512, 2, 1e-12, 1)
cfg, ocfg = BertConfig(*vals), oracle.BertConfig(*vals)
blob = random_blob(cfg, seed=seed0, init_range=0.05)
enc = Encoder(cfg, blob)
t_end = time.time() + budget
cases = fails = seqs = 0
worst = 0.0
it = 0
while time.time() < t_end:
    seed = seed0 * 100003 + it
    it += 1
    rng = np.random.default_rng(seed)
    n = int(rng.integers(1, 90))
    hi = int(rng.choice([8, 40, 130, 257, 300]))
    lens = rng.integers(0, hi + 1, size=n)
    if rng.random() < 0.3:
        lens[:] = int(rng.integers(1, hi + 1))                 # uniform batch
    max_len = int(rng.choice([0, 64, 256]))
    cap = cfg.max_position_embeddings if max_len == 0 else max_len
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    flat = rng.integers(0, 3000, size=int(offsets[-1])).astype(np.int32)
    os.environ["MI_ENC_LATENCY_TOKENS"] = str(rng.choice([0, 12288, 1 << 30]))     # always bulk / default hand-over / never bulk
    os.environ["MI_QKV_FUSED_MIN_MEAN"] = str(rng.choice([0, 0, 128]))          # qkv_attn_kernel for every bulk pass up to 256 tokens / default
    max_batch = int(rng.choice([0, 0, 5, 33]))
    E = enc.encode_ragged(flat, offsets, max_len=max_len, max_batch=max_batch)
    # oracle: padded batch of the truncated sequences
    tl = np.minimum(lens, cap)
    L = max(1, int(tl.max()))
    ids = np.zeros((n, L), dtype=np.int32); mask = np.zeros((n, L), dtype=np.int32)
    for i in range(n):
        ids[i, :tl[i]] = flat[offsets[i]:offsets[i] + tl[i]]; mask[i, :tl[i]] = 1
    rows = np.flatnonzero(tl > 0)
    ok = bool(np.all(E[tl == 0] == 0))
    err = 0.0
    if rows.size:
        Eo = oracle.encode(ocfg, blob, ids[rows], mask[rows])
        cos = (E[rows] * Eo).sum(1) / (np.linalg.norm(E[rows], axis=1) * np.linalg.norm(Eo, axis=1))
        err = float(np.max(1.0 - cos))
        # the padded entry point on the same batch
        Ep = enc.encode_ids(ids[rows], mask[rows])
        cosp = (Ep * Eo).sum(1) / (np.linalg.norm(Ep, axis=1) * np.linalg.norm(Eo, axis=1))
        err = max(err, float(np.max(1.0 - cosp)))
    worst = max(worst, err)
    cases += 1; seqs += n
    if cases % 10 == 0:
        print(json.dumps({"progress": cases, "sequences": seqs, "worst_1_minus_cos": worst}), flush=True)
    if not ok or not (err < 1e-3) or not np.all(np.isfinite(E)):
        print(json.dumps({"seed": seed, "n": n, "hi": hi, "max_len": max_len, "max_batch": max_batch, "zero_rows_ok": ok, "err": err,
                          "latency_tokens": os.environ["MI_ENC_LATENCY_TOKENS"]}), flush=True)
        fails += 1
        break
st = enc.stats()
print(json.dumps({"cases": cases, "failures": fails, "sequences": seqs, "worst_1_minus_cos": worst, "bulk_passes": st["bulk_passes"],
                  "latency_passes": st["latency_passes"], "seconds": round(budget - (t_end - time.time()), 1), "seed0": seed0}), flush=True)
sys.exit(1 if fails else 0)
