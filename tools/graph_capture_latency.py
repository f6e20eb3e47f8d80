"""HIP-graph capture of the allocation-free encoder entry point (mi_encoder_encode_dev_noalloc) through torch.cuda.CUDAGraph:
replays must equal direct calls bit for bit, also after the static input tensor is overwritten; per-call latency of both.
Not a bench line; evidence for DESIGN.md section 4.6."""
import os, sys, time, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rag_amd.embeddings import Encoder
from rag_amd.encoder_weights import minilm_l6_config, random_blob
cfg = minilm_l6_config()
enc = Encoder(cfg, random_blob(cfg, seed=0, init_range=0.02))
rng = np.random.default_rng(0)
for B, L in ((1, 16), (1, 64), (4, 128)):
    ids = torch.from_numpy(rng.integers(1000, 30000, (B, L), dtype=np.int32)).cuda(); mask = torch.ones_like(ids)
    out = torch.empty((B, cfg.hidden_size), dtype=torch.float32, device="cuda")
    for _ in range(3): enc.encode_ids_torch_noalloc(ids, mask, out)
    torch.cuda.synchronize()
    want = out.clone()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        enc.encode_ids_torch_noalloc(ids, mask, out)       # warm on the side stream
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        enc.encode_ids_torch_noalloc(ids, mask, out)
    out.zero_(); g.replay(); torch.cuda.synchronize()
    same = bool(torch.equal(out.view(torch.int32), want.view(torch.int32)))
    # new input through the same graph
    ids2 = torch.from_numpy(rng.integers(1000, 30000, (B, L), dtype=np.int32)).cuda()
    ids.copy_(ids2); g.replay(); torch.cuda.synchronize(); got2 = out.clone()
    enc.encode_ids_torch_noalloc(ids2, mask, out); torch.cuda.synchronize()
    same2 = bool(torch.equal(out.view(torch.int32), got2.view(torch.int32)))
    def timeit(f, n=200):
        for _ in range(10): f()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): f(); torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e6
    t_direct = timeit(lambda: enc.encode_ids_torch_noalloc(ids, mask, out))
    t_graph = timeit(lambda: g.replay())
    print(json.dumps({"B": B, "L": L, "replay_equals_direct": same, "new_input_equals_direct": same2, "direct_us": round(t_direct, 1), "graph_us": round(t_graph, 1)}), flush=True)
