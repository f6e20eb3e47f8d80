"""Latency of embedding ONE query (advanced_rag.py: embed_query before every search) through the host-buffer
C-ABI entry point, MiniLM-L6 with seeded random weights.  Not a bench line; evidence for DESIGN.md."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rag_amd.embeddings import Encoder
from rag_amd.encoder_weights import minilm_l6_config, random_blob
cfg = minilm_l6_config()
enc = Encoder(cfg, random_blob(cfg, seed=0, init_range=0.02))
rng = np.random.default_rng(0)
for B, L in ((1, 16), (1, 64), (1, 256), (8, 64), (32, 256)):
    ids = rng.integers(1000, 30000, (B, L), dtype=np.int32); mask = np.ones_like(ids)
    for _ in range(5): enc.encode_ids(ids, mask)
    ts = []; ks = []
    for _ in range(30):
        t0 = time.perf_counter(); enc.encode_ids(ids, mask); ts.append(time.perf_counter() - t0); ks.append(enc.last_time_ms())
    print(json.dumps({"chunks": B, "tokens": L, "host_call_us_median": round(float(np.median(ts)) * 1e6, 1),
                      "device_us_median": round(float(np.median(ks)) * 1e3, 1)}), flush=True)
