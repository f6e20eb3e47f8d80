"""BASELINE config 2 (1M x 384 fp32 rows, 1,000 queries, k = 10): wall and scan time against the slice count of the MFMA
scan (MI_MFMA_NS) and the variant (MI_MFMA_PD / MI_MFMA_M16).  One JSON line per setting."""
import os, sys, json, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rag_amd.index import FlatIndex
N = int(float(os.environ.get("N", "1e6"))); d = 384; k = 10; Q = int(os.environ.get("Q", "1000"))
idx = FlatIndex(d, metric="ip", dtype=os.environ.get("DT", "f32")); idx.add_synthetic(N, seed=0)
g = FlatIndex(d, metric="ip", dtype="f32"); g.add_synthetic(Q, seed=1)
q = torch.from_numpy(g.reconstruct_n(0, Q)).cuda()
ref = None
for setting in os.environ.get("SETTINGS", "0;64;96;128;160;192;256;320").split(";"):
    for kv in ("MI_MFMA_NS", "MI_MFMA_PD", "MI_MFMA_M16"): os.environ.pop(kv, None)
    parts = setting.split(",")
    if parts[0] != "0": os.environ["MI_MFMA_NS"] = parts[0]
    if len(parts) > 1: os.environ["MI_MFMA_PD"] = parts[1]
    if len(parts) > 2: os.environ["MI_MFMA_M16"] = parts[2]
    for _ in range(3): D, I = idx.search_torch(q, k)
    torch.cuda.synchronize()
    wall, scan = [], []
    for _ in range(9):
        t0 = time.perf_counter(); D, I = idx.search_torch(q, k); torch.cuda.synchronize()
        wall.append((time.perf_counter() - t0) * 1e3); scan.append(idx.last_scan_time()["ms"])
    if ref is None: ref = (D.clone(), I.clone())
    same = bool(torch.equal(ref[1], I) and torch.equal(ref[0].view(torch.int32), D.view(torch.int32)))
    print(json.dumps({"setting": setting, "wall_ms": round(float(np.median(wall)), 3), "scan_ms": round(float(np.median(scan)), 3),
                      "kind": idx.last_scan_time()["kind"], "same_answer": same, "exact": idx.stats()["exact"]}), flush=True)
