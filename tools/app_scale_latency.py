"""Latency of the app's own case (advanced_rag.py: a few thousand chunks, fp32 L2 store, one query, top-5) through
the host-buffer C-ABI entry points.  Not a bench line; evidence for DESIGN.md."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rag_amd.index import FlatIndex
rng = np.random.default_rng(0)
for n in (2000, 20000, 200000):
    X = rng.standard_normal((n, 384)).astype(np.float32); X /= np.linalg.norm(X, axis=1, keepdims=True)
    q = rng.standard_normal((1, 384)).astype(np.float32); q /= np.linalg.norm(q)
    idx = FlatIndex(384, metric="l2", dtype="f32"); idx.add(X)
    for _ in range(5): idx.search(q, 5)
    ts = []
    for _ in range(50):
        t0 = time.perf_counter(); idx.search(q, 5); ts.append(time.perf_counter() - t0)
    print(json.dumps({"rows": n, "dim": 384, "dtype": "f32", "metric": "l2", "queries": 1, "k": 5,
                      "host_call_us_median": round(float(np.median(ts)) * 1e6, 1), "host_call_us_min": round(min(ts) * 1e6, 1),
                      "path": idx.last_scan_time()["kind"]}))
