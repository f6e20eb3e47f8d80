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
ids = rng.integers(0, 30522, size=(256, 256)).astype(np.int32); mask = np.ones((256, 256), np.int32)
enc = Encoder(cfg, blob)
for _ in range(3): enc.encode_ids(ids, mask)
print("encode ms (64K tokens):", enc.last_time_ms())
lib = _capi.load()
buf = np.zeros((8, 8, 16), np.uint32)
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
