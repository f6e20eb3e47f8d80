"""Reads the phase stamps the FF_TRACE build of ffn_ln_kernel leaves (tools/_trace/libtrace.so through MI355RAG_LIB)
and prints where a workgroup spends a 128-row tile.  Debug tool, not part of the product."""
import ctypes, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rag_amd import _capi
from rag_amd._capi import BertConfig
from rag_amd.embeddings import Encoder
from rag_amd.encoder_weights import random_blob
cfg = BertConfig(vocab_size=30522, hidden_size=384, num_hidden_layers=6, num_attention_heads=12, intermediate_size=1536,
                 max_position_embeddings=512, type_vocab_size=2, layer_norm_eps=1e-12, normalize=1)
blob = random_blob(cfg, seed=0, init_range=0.02)
rng = np.random.default_rng(0)
NSEQ = int(os.environ.get("NSEQ", "256"))                     # sequences of 256 tokens (256 = 64K tokens = 512 row tiles: two per CU)
ids = rng.integers(0, 30522, size=(NSEQ, 256)).astype(np.int32); mask = np.ones((NSEQ, 256), np.int32)
os.environ["MI_ENC_LATENCY_TOKENS"] = "0"
enc = Encoder(cfg, blob)
for _ in range(3): enc.encode_ids(ids, mask)
print("encode ms (64K tokens):", enc.last_time_ms())
lib = _capi.load()
buf = np.zeros((8, 8, 48), np.uint32)
lib.mi_debug_read_ff_trace.argtypes = [ctypes.c_void_p]
assert lib.mi_debug_read_ff_trace(buf.ctypes.data_as(ctypes.c_void_p)) == 0
t = buf.astype(np.int64)
names = ["front GEMM (out-proj, 12 stages)", "front bias+resid+LN -> LDS", "h fragments from LDS", "weight prologue (5 DMA blocks, wait)",
         "step 1 of stages 0, 1 (no overlap)", "main loop (48 stages)", "b2/gamma/beta + residual -> epilogue layout", "final bias+resid+LN, rows out"]
tot = (t[:, :, 8] - t[:, :, 0]) & 0xffffffff
real = (t[:, :, 14] - t[:, :, 15]) & 0xffffffff
print("tile total cycles (median over slots/waves):", np.median(tot), " real 100MHz ticks:", np.median(real), " -> clock GHz %.2f" % (np.median(tot) / np.median(real) / 10.0))
for k in range(8):
    d = (t[:, :, k + 1] - t[:, :, k]) & 0xffffffff
    print(f"{names[k]:46s} median {np.median(d):8.0f} cyc  ({100 * np.median(d) / np.median(tot):5.1f} %)  min {d.min():8d} max {d.max():8d}")

# inside one stage (FF_TRACE_KT): ticks 0 top, 13 at barrier 1, 1 past it, 2 fragments of group 0 there / at barrier 2, 3 past it, 4..12 behind groups 0..8
tk = t[:, :, 16:30]
order = [0, 13, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12]
lab = ["top", "vmcnt+lgkm waited, at barrier 1", "past barrier 1", "af + W2 group 0 read, at barrier 2", "past barrier 2"] + [f"behind group {g}" for g in range(9)]
print("one stage, cycles since the stage top, per wave of slot 0 and median over all:")
for j, k in enumerate(order):
    d = (tk[:, :, k] - tk[:, :, 0]) & 0xffffffff
    print(f"{lab[j]:40s} median {np.median(d):7.0f}   slot0 waves: " + " ".join(f"{x:5d}" for x in d[0]))
print("barrier 1 hold (past - at) per slot x wave:")
print(((tk[:, :, 1] - tk[:, :, 13]) & 0xffffffff))
print("barrier 2 hold per slot x wave:")
print(((tk[:, :, 3] - tk[:, :, 2]) & 0xffffffff))
print("at-barrier-1 relative to the earliest wave of the workgroup:")
a = tk[:, :, 13]; print(a - a.min(axis=1, keepdims=True))
print("past-barrier-1 relative to the earliest:")
a = tk[:, :, 1]; print(a - a.min(axis=1, keepdims=True))
print("past-barrier-2 relative to the earliest:")
a = tk[:, :, 3]; print(a - a.min(axis=1, keepdims=True))
print("timeline of one stage relative to the release of barrier 1 (rows: events, columns: waves 0..7), slots 0 and 3:")
ev = [13, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 0]
evn = ["at B1", "past B1", "at B2", "past B2"] + [f"g{g} done" for g in range(9)] + ["next top"]
for sl in (0, 3):
    base = tk[sl, :, 1].min()
    for j, k in enumerate(ev):
        print(f"  {evn[j]:9s} " + " ".join(f"{int(x - base):6d}" for x in tk[sl, :, k]))
if True:   # (the round-1 kernel this used to switch to is retired)
    print("ffn2 stage timeline relative to the release of barrier alpha (rows: events, columns: waves 0..7), slots 0 and 3:")
    ev2 = [0, 1, 2, 3, 4, 5, 6, 9, 10, 11, 12, 7, 8]
    evn2 = ["top", "at alpha", "past alpha", "g1 done", "g2 done", "at beta", "past beta", "g3 mfma'd", "+2 DMA", "g4 mfma'd", "+2 DMA", "g4 done", "g5 done"]
    for sl in (0, 3):
        base = tk[sl, :, 2].min()
        for j, k in enumerate(ev2):
            print(f"  {evn2[j]:10s} " + " ".join(f"{int(x - base):6d}" for x in tk[sl, :, k]))

if True:   # (the round-1 kernel this used to switch to is retired)
    print("front GEMM: top of stage kt relative to the kernel's first stamp (slot 0, waves 0..7):")
    fr = t[0, :, 32:44]
    for kt in range(12):
        print(f"  stage {kt:2d} " + " ".join(f"{int((x - t[0, w, 0]) & 0xffffffff):6d}" for w, x in enumerate(fr[:, kt])))
    print("  end      " + " ".join(f"{int((t[0, w, 1] - t[0, w, 0]) & 0xffffffff):6d}" for w in range(8)))
