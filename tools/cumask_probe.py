"""Two encoder passes side by side on two plain streams (CU-masked streams are refused on this pool: hipExtStreamCreateWithCUMask
returns 2 for any mask but all-ones): does overlapping one pass's memory-bound kernels (QKV, attention) with the other's
MFMA-bound layer tail beat two passes back to back?  MI_FFN3_GRID=<n> leaves 256 - n CUs free of ffn3 workgroups."""
import ctypes, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rag_amd import _capi
from rag_amd.embeddings import Encoder
from rag_amd.encoder_weights import minilm_l6_config, random_blob
hip = ctypes.CDLL("libamdhip64.so")
cfg = minilm_l6_config(); blob = random_blob(cfg, seed=0, init_range=0.02)
encs = [Encoder(cfg, blob), Encoder(cfg, blob)]
lib = _capi.load()
B, L = 256, 256
ids = [torch.randint(1000, 30000, (B, L), dtype=torch.int32, device="cuda") for _ in range(2)]
mask = torch.ones((B, L), dtype=torch.int32, device="cuda")
outs = [torch.empty((B, cfg.hidden_size), dtype=torch.float32, device="cuda") for _ in range(2)]
def mk():
    s = ctypes.c_void_p(); assert hip.hipStreamCreateWithFlags(ctypes.byref(s), 1) == 0; return s
def run(e, k, s):
    _capi.check(lib.mi_encoder_encode_dev(encs[e]._h, B, L, ctypes.c_void_p(ids[k].data_ptr()), ctypes.c_void_p(mask.data_ptr()),
                                          ctypes.c_void_p(outs[k].data_ptr()), s))
def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
sa, sb = mk(), mk()
t_seq = timeit(lambda: (run(0, 0, sa), run(0, 1, sa)))
t_par = timeit(lambda: (run(0, 0, sa), run(1, 1, sb)))
print(f"MI_FFN3_GRID={os.environ.get('MI_FFN3_GRID', '256')}: two passes back to back on one stream {t_seq:.3f} ms; side by side on two streams {t_par:.3f} ms "
      f"({2 * B / t_seq * 1e3:.0f} vs {2 * B / t_par * 1e3:.0f} chunks/s)", flush=True)
