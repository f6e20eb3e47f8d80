"""Start-up trace of the MFMA scan: where one workgroup's first units go (200,000 rows x 1 query, L2, F32 index).
Needs the stamp build of the library, loaded through MI355RAG_LIB:
  hipcc <FLAGS of rag_amd/_build.py> -DMF_TRACE -DMF_TRACE_U0=0 -c rag_amd/csrc/search_mfma.hip -o tools/_trace/search_mfma.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_trace/libtrace.so build/obj/{capi,ingest,search_exact,encoder}.o tools/_trace/search_mfma.o
  MI_MFMA_NS=192 MI355RAG_LIB=tools/_trace/libtrace.so python3 tools/trace_scan_startup.py
This is how the padding-query compactions (latency section) were found.  Debug tool, not part of the product."""
import ctypes, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rag_amd import _capi
from rag_amd.index import FlatIndex, MODE_FAST
rng = np.random.default_rng(0)
n = 200000
X = rng.standard_normal((n, 384)).astype(np.float32); X /= np.linalg.norm(X, axis=1, keepdims=True)
q = rng.standard_normal((1, 384)).astype(np.float32); q /= np.linalg.norm(q)
idx = FlatIndex(384, metric="l2", dtype="f32"); idx.set_mode(MODE_FAST); idx.add(X)
for _ in range(6): idx.search(q, 5)
print("scan", idx.last_scan_time())
lib = _capi.load()
buf = np.zeros((4, 8, 64, 8), dtype=np.uint32)
lib.mi_debug_read_trace.argtypes = [ctypes.c_void_p]
assert lib.mi_debug_read_trace(buf.ctypes.data_as(ctypes.c_void_p)) == 0
t = buf[0].astype(np.int64)      # [wave][unit][k]
for w in range(8):
    e, l, x = t[w, 63, 0], t[w, 63, 1], t[w, 63, 2]
    print(f"wave {w}: entry->loop {(l - e) & 0xffffffff}  loop {(x - l) & 0xffffffff}  total {(x - e) & 0xffffffff} (x10ns @100MHz)")
w = 0
nun = int(os.environ.get("NUN", "30"))
for u in range(nun):
    r = t[w, u]
    if r[0] == 0: break
    print(u, "top->chain_end", (r[1]-r[0]) & 0xffffffff, "filter", (r[2]-r[6]) & 0xffffffff, "wait", (r[3]-r[2]) & 0xffffffff, "barrier", (r[4]-r[3]) & 0xffffffff, " since loop", (r[0]-t[w,63,1]) & 0xffffffff)
print("wave 5:")
w = 5
for u in range(nun):
    r = t[w, u]
    if r[0] == 0: break
    print(u, "top->chain_end", (r[1]-r[0]) & 0xffffffff, "filter", (r[2]-r[6]) & 0xffffffff, "wait", (r[3]-r[2]) & 0xffffffff, "barrier", (r[4]-r[3]) & 0xffffffff, " since loop", (r[0]-t[w,63,1]) & 0xffffffff)
