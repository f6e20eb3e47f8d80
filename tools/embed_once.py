"""One warm-up and ONE bulk-embedding call of 4096 chunks x 256 tokens (16 passes of 64K tokens) on resident inputs:
the workload of bench.py's embed `value`, for profiler passes (every kernel of the call appears 16 x layers times)."""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rag_amd.embeddings import Encoder
from rag_amd.encoder_weights import minilm_l6_config, random_blob
cfg = minilm_l6_config()
enc = Encoder(cfg, random_blob(cfg, seed=0, init_range=0.02))
g = torch.Generator(device="cpu").manual_seed(1234)
ids = torch.randint(1000, 30000, (4096, 256), generator=g, dtype=torch.int32).cuda()
mask = torch.ones((4096, 256), dtype=torch.int32, device="cuda")
enc.encode_ids_torch(ids[:256], mask[:256]); torch.cuda.synchronize()
E = enc.encode_ids_torch(ids, mask); torch.cuda.synchronize()
print(json.dumps({"chunks": 4096, "tokens": 256, "last_pass_ms": enc.last_time_ms(), "stats": enc.stats()}))
