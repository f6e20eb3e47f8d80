"""Bulk-embedding rate (4096 chunks x 256 tokens resident, best of 5 calls) with the layer tail the environment selects
(MI_FFN3=0: ffn2_kernel, otherwise ffn3_kernel): run twice in one GPU call for a same-box A/B."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rag_amd.embeddings import Encoder
from rag_amd.encoder_weights import minilm_l6_config, random_blob
cfg = minilm_l6_config()
enc = Encoder(cfg, random_blob(cfg, seed=0, init_range=0.02))
g = torch.Generator(device="cpu").manual_seed(1234)
ids = torch.randint(1000, 30000, (4096, 256), generator=g, dtype=torch.int32).cuda()
mask = torch.ones((4096, 256), dtype=torch.int32, device="cuda")
enc.encode_ids_torch(ids, mask); torch.cuda.synchronize()
best = 1e9
for _ in range(5):
    t0 = time.perf_counter()
    enc.encode_ids_torch(ids, mask); torch.cuda.synchronize()
    best = min(best, time.perf_counter() - t0)
print(f"MI_FFN3={os.environ.get('MI_FFN3', '(default)')}  {4096 / best:9.1f} chunks/s   {best * 1e3:.2f} ms per 4096 chunks   {best * 1e6 / 16 / cfg.num_hidden_layers:.1f} us per layer pass")
