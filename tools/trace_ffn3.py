"""Cycle stamps of ffn3_kernel (debug build: FILE=encoder_ffn3 tools/build_ab.sh f3trace "-DF3_TRACE -mllvm -amdgpu-spill-vgpr-to-agpr=0";
run with MI355RAG_LIB=tools/_ab/f3trace.so).  Prints cycles per phase of the first tile of eight workgroups."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rag_amd import _capi
from rag_amd.embeddings import Encoder
from rag_amd.encoder_weights import minilm_l6_config, random_blob
cfg = minilm_l6_config()
enc = Encoder(cfg, random_blob(cfg, seed=0))
ids = torch.randint(1000, 30000, (256, 256), dtype=torch.int32, device="cuda")
mask = torch.ones_like(ids)
for _ in range(3):
    enc.encode_ids_torch(ids, mask)
torch.cuda.synchronize()
lib = _capi.load()
buf = (ctypes.c_ulonglong * (8 * 4 * 8))()
assert lib.mi_ffn3_trace_read(buf) == 0
t = np.array(buf, dtype=np.uint64).reshape(8, 4, 8).astype(np.int64)
names = ["token loads + acc init", "front GEMM (12 blocks)", "LayerNorm 1", "FFN (96 blocks)", "LayerNorm 2 + stores"]
d = np.diff(t[:, :, :6], axis=2)
print("cycles per phase, median over 8 workgroups x 4 waves (first tile of each):")
for i, n in enumerate(names):
    print(f"  {n:28s} {int(np.median(d[:, :, i])):8d}   (min {int(d[:, :, i].min())}, max {int(d[:, :, i].max())})")
print(f"  {'tile':28s} {int(np.median(t[:, :, 5] - t[:, :, 0])):8d};  per FFN block {np.median(d[:, :, 3]) / 96:.0f}, per front block {np.median(d[:, :, 1]) / 12:.0f}")
print("enc.last_time_ms", enc.last_time_ms())
