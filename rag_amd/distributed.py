"""Row-sharded flat search over the GPUs of one node.

One process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI).  Rank r owns the
contiguous row block [r*ceil(N/G), min(N, (r+1)*ceil(N/G))) and reports GLOBAL ids
(mi_index_set_id_base).  A search is: local exact top-k on every rank -> ONE all-gather of the
per-rank (score, id) lists (Q*k*12 bytes per rank: latency-bound, not link-bound) -> the same
deterministic merge on every rank, so every rank returns the identical answer, equal to the
single-index answer.  There is no other collective on the path.

The local engine and the merge are injectable so the sharding / gather logic can be covered
by world_size-2 gloo tests on CPU (tests/test_distributed_cpu.py injects the oracle there);
the defaults are the HIP engine and the HIP merge, and nothing else ships.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import numpy as np


def shard_range(n_total: int, world: int, rank: int) -> Tuple[int, int]:
    per = (n_total + world - 1) // world
    lo = min(n_total, rank * per)
    hi = min(n_total, lo + per)
    return lo, hi


class ShardedFlatIndex:
    """Exact top-k over a corpus row-sharded across the ranks of a process group."""

    def __init__(self, d: int, metric: str = "l2", dtype: str = "f32", device: Optional[int] = None,
                 group=None, local_index=None, merge_fn: Optional[Callable] = None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.d, self.metric, self.dtype = int(d), metric, dtype
        self.n_total = 0
        if local_index is None:
            from .index import FlatIndex
            import torch
            dev = torch.cuda.current_device() if device is None else device
            local_index = FlatIndex(d, metric=metric, dtype=dtype, device=dev)
        self.local = local_index
        self._merge = merge_fn
        self._on_gpu = hasattr(self.local, "search_torch")

    # ---- build
    def add_global(self, X: np.ndarray) -> None:
        """Every rank is handed the same [N][d] block; each keeps only its rows."""
        X = np.asarray(X, dtype=np.float32)
        if self.n_total != 0:
            raise ValueError("add_global: contiguous row shards are fixed at the first add; "
                             "use add_local on each rank to grow a shard")
        lo, hi = shard_range(X.shape[0], self.world, self.rank)
        self.local.set_id_base(lo)
        if hi > lo:
            self.local.add(X[lo:hi])
        self.n_total = X.shape[0]

    def add_synthetic_global(self, n_total: int, seed: int = 0) -> None:
        lo, hi = shard_range(n_total, self.world, self.rank)
        self.local.set_id_base(lo)
        self.local.add_synthetic(hi - lo, seed=seed)
        self.n_total = n_total

    # ---- search
    def search(self, Q, k: int):
        """Q: [nq][d] fp32 (NumPy, or a CUDA tensor on the GPU path), identical on every rank.
        Returns (scores [nq][k], ids [nq][k]) as NumPy arrays, identical on every rank."""
        import torch
        dist = self.dist
        if self._on_gpu:
            q = Q if torch.is_tensor(Q) else torch.from_numpy(np.ascontiguousarray(Q, dtype=np.float32)).cuda()
            D, I = self.local.search_torch(q.contiguous(), k)
        else:
            Dn, In = self.local.search(np.asarray(Q, dtype=np.float32), k)
            D, I = torch.from_numpy(Dn), torch.from_numpy(In)
        nq = D.shape[0]
        if self.world > 1:
            Dg = torch.empty((self.world * nq, k), dtype=D.dtype, device=D.device)   # rank-major concat
            Ig = torch.empty((self.world * nq, k), dtype=I.dtype, device=I.device)
            dist.all_gather_into_tensor(Dg, D.contiguous(), group=self.group)
            dist.all_gather_into_tensor(Ig, I.contiguous(), group=self.group)
            Dm = Dg.view(self.world, nq, k).permute(1, 0, 2).reshape(nq, self.world * k).contiguous()
            Im = Ig.view(self.world, nq, k).permute(1, 0, 2).reshape(nq, self.world * k).contiguous()
            D, I = self._merge_lists(Dm, Im, k)
        return D.cpu().numpy(), I.cpu().numpy()

    def _merge_lists(self, Dm, Im, k):
        if self._merge is not None:
            return self._merge(Dm, Im, k, self.metric)
        from .index import topk_merge_torch
        return topk_merge_torch(Dm, Im, k, self.metric)
