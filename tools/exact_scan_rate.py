"""Rate of the exact (canonical-order) scan, the fallback path: 1M x 384 rows, 1,000 queries, MODE_EXACT, scan
kernel time from the library's HIP events -> TFLOP/s of 2 Q N d.  With MI_EXACT_MFMA=0 the fp32 VALU kernel runs
instead of the fp32 MFMA one (one process each: the switch is read once)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import sys, json, numpy as np, torch
sys.path.insert(0, %r)
from rag_amd.index import FlatIndex, MODE_EXACT
import os
for dtype in ("f32", "bf16"):
    for n, d, nq in ((1_000_000, 384, 1000), (30_000, 768, 4096), (1_000_000, 1024, 200)):
        idx = FlatIndex(d, metric="ip", dtype=dtype); idx.set_mode(MODE_EXACT); idx.add_synthetic(n, seed=3)
        g = FlatIndex(d, metric="ip", dtype="f32"); g.add_synthetic(nq, seed=4)
        q = torch.from_numpy(g.reconstruct_n(0, nq)).cuda()
        idx.search_torch(q, 10); torch.cuda.synchronize()
        ms = []
        for _ in range(3):
            idx.search_torch(q, 10); torch.cuda.synchronize(); ms.append(idx.last_scan_time()["ms"])
        m = float(np.median(ms))
        print(json.dumps({"kernel": "valu" if os.environ.get("MI_EXACT_MFMA") == "0" else "mfma_f32", "dtype": dtype, "rows": n, "dim": d,
                          "queries": nq, "scan_ms": round(m, 3), "TFLOPs": round(2.0 * nq * n * d / m / 1e9, 1)}), flush=True)
""" % ROOT
for env in ({"MI_EXACT_MFMA": "0"}, {}):
    e = dict(os.environ); e.update(env)
    out = subprocess.run([sys.executable, "-c", CHILD], env=e, capture_output=True, text=True, timeout=280)
    sys.stdout.write(out.stdout)
    if out.returncode != 0:
        sys.stderr.write(out.stderr[-3000:]); sys.exit(1)
