"""Reads the cycle stamps the MF_TRACE build of the scan leaves (tools/_trace/libtrace.so, loaded through
MI355RAG_LIB) and prints where a workgroup's waves spend a unit.  Debug tool, not part of the product."""
import ctypes, sys, os, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rag_amd import _capi
from rag_amd.index import FlatIndex, MODE_FAST

n = int(float(os.environ.get("N", "100e6"))); nq = int(os.environ.get("Q", "10000")); D = int(os.environ.get("D", "384")); K = int(os.environ.get("K", "10"))
idx = FlatIndex(D, metric="ip", dtype="bf16"); idx.set_mode(MODE_FAST)
idx.add_synthetic(n, seed=1)
g = torch.Generator(device="cpu").manual_seed(5)
q = torch.nn.functional.normalize(torch.randn(nq, D, generator=g), dim=1).cuda()
for _ in range(2):
    Dd, I = idx.search_torch(q, K)
torch.cuda.synchronize()
print("scan ms", idx.last_scan_time())
lib = _capi.load()
buf = np.zeros((4, 8, 64, 8), dtype=np.uint32)
lib.mi_debug_read_trace.argtypes = [ctypes.c_void_p]
rc = lib.mi_debug_read_trace(buf.ctypes.data_as(ctypes.c_void_p)); assert rc == 0
np.save("gpurun_out/trace.npy", buf)
for b in range(4):
    t = buf[b].astype(np.int64)                       # [wave][unit][k]
    print(f"--- workgroup slot {b}")
    per = (t[:, 1:, 0] - t[:, :-1, 0]) & 0xffffffff
    print("unit period per wave (median):", np.median(per, axis=1))
    NKB = 6 if D > 384 else (3 if D > 128 else 1); kb = (np.arange(64) + int(os.environ.get("U0", "3000"))) % NKB
    def seg(name, a, b_, sel=None):
        d = (t[:, :, b_] - t[:, :, a]) & 0xffffffff
        if sel is not None: d = d[:, sel]
        print(f"  {name:34s} w0-3 mean {d[:4].mean():7.0f}  w4-7 mean {d[4:].mean():7.0f}  median {np.median(d):7.0f} p90 {np.percentile(d, 90):7.0f}")
    seg("chain", 0, 1)
    seg("issue DMA", 1, 6)
    seg("filter slot kb<last (stamp only)", 6, 2, kb != NKB - 1)
    seg("filter slot kb=last", 6, 2, kb == NKB - 1)
    seg("wait vmcnt", 2, 3)
    seg("barrier", 3, 4)
    d = (t[:, 1:, 0] - t[:, :-1, 4]) & 0xffffffff
    print("  after barrier -> next top: mean", d.mean())
    s3 = t[:, :, 3]
    print("  arrival skew at barrier (max-min over waves): median", np.median(s3.max(0) - s3.min(0)))
