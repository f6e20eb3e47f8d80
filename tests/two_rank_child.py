"""Child of tests/test_sharded_two_ranks_gpu.py: one of two ranks (gloo rendezvous, both on GPU 0 -- RCCL refuses two ranks on
one device, gloo moves CUDA tensors) running the PRODUCT's sharded path: HIP engine per shard, the packed block, ONE
all_gather_into_tensor, the HIP merge reading the gathered blocks in place; and the data-parallel embedding with its gather.
Prints OK lines; any mismatch raises."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rag_amd._capi import BertConfig                      # noqa: E402
from rag_amd.distributed import ShardedFlatIndex, embed_sharded, shard_range   # noqa: E402
from rag_amd.embeddings import Encoder                    # noqa: E402
from rag_amd.encoder_weights import random_blob           # noqa: E402
from rag_amd.index import FlatIndex                       # noqa: E402

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("gloo")

# ---- search: every rank is handed the same corpus, keeps its rows; duplicates straddle the shard boundary
for metric, dtype, N, d, Q, k in (("ip", "bf16", 90_000, 384, 300, 10), ("l2", "f32", 5_001, 100, 40, 7), ("ip", "f32", 9, 64, 3, 8)):
    rng = np.random.default_rng(N + d)
    X = rng.standard_normal((N, d)).astype(np.float32)
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    lo, hi = shard_range(N, world, 0)
    if N > 100:
        X[hi - 3:hi + 3] = X[7]                           # six copies of one row across the boundary: ties by id
    Qm = rng.standard_normal((Q, d)).astype(np.float32)
    Qm[0] = X[7]
    sh = ShardedFlatIndex(d, metric=metric, dtype=dtype, device=0)
    sh.add_global(X)
    q = torch.from_numpy(Qm).to(dev)
    D, I = sh.search_torch(q, k)
    torch.cuda.synchronize()
    one = FlatIndex(d, metric=metric, dtype=dtype, device=0)
    one.add(X)
    D1, I1 = one.search_torch(q, k)
    torch.cuda.synchronize()
    assert torch.equal(I, I1), (metric, dtype, N)
    assert torch.equal(D.view(torch.int32), D1.view(torch.int32)), (metric, dtype, N)
    # identical on every rank
    mine = torch.cat([I.reshape(-1).double(), D.reshape(-1).double()]).cpu()
    both = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(both, mine)
    assert all(torch.equal(both[0], b) or bool(torch.isnan(both[0]).any()) for b in both)
    print(f"rank {rank} OK search {metric} {dtype} N={N} k={k}", flush=True)

# ---- embedding: contiguous chunk ranges per rank, no collective on the path, one gather of the result
vals = (500, 128, 2, 4, 256, 64, 2, 1e-12, 1)
cfg = BertConfig(*vals)
enc = Encoder(cfg, random_blob(cfg, seed=4, init_range=0.1))
rng = np.random.default_rng(11)
n = 37
lens = rng.integers(1, 60, size=n)
offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
flat = rng.integers(0, 500, size=int(offsets[-1])).astype(np.int32)
E = embed_sharded(enc, flat, offsets, gather=True)
want = enc.encode_ragged(flat, offsets)
assert E.shape == want.shape and np.array_equal(E.view(np.uint32), want.view(np.uint32))
part = embed_sharded(enc, flat, offsets, gather=False)
lo, hi = shard_range(n, world, rank)
assert np.array_equal(part.view(np.uint32), want[lo:hi].view(np.uint32))
print(f"rank {rank} OK embed_sharded", flush=True)
dist.barrier()
dist.destroy_process_group()
