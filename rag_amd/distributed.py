"""Row-sharded flat search and data-parallel embedding over the GPUs of one node.

One process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI).

Search.  Rank r owns the contiguous row block [r*ceil(N/G), min(N, (r+1)*ceil(N/G))) and reports GLOBAL ids
(mi_index_set_id_base).  A search is: local exact top-k on every rank, written straight into the rank's
packed block ([nq][k] fp32 scores | [nq][k] int64 ids, mi_topk_packed_layout) -> ONE all-gather of that block
(Q*k*12 bytes per rank: latency-bound, not link-bound) -> the same deterministic merge on every rank, reading
the rank-major blocks in place (mi_topk_merge_packed_dev).  Every rank returns the identical answer, equal to
the single-index answer.  There is no other collective, and no transpose or copy between the two steps.

Embedding (advanced_rag.py:295 at BASELINE config 3's scale) is pure data parallel: rank r encodes the
contiguous chunk range shard_range(n, G, r); no collective on the path, an optional all-gather of the
[n][hidden] result when every rank needs all of it.

The local engine, the merge and the encoder are injectable so the sharding / gather logic can be covered by
world_size-2 gloo tests on CPU (tests/test_distributed_cpu.py injects the oracle there); the defaults are the
HIP engine and the HIP merge, and nothing else ships.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import numpy as np


def shard_range(n_total: int, world: int, rank: int) -> Tuple[int, int]:
    per = (n_total + world - 1) // world
    lo = min(n_total, rank * per)
    hi = min(n_total, lo + per)
    return lo, hi


def packed_layout(nq: int, k: int) -> Tuple[int, int]:
    """(ids_offset, block_bytes) of one rank's packed result block; mirrors mi_topk_packed_layout."""
    off = (nq * k * 4 + 255) & ~255
    return off, (off + nq * k * 8 + 255) & ~255


class ShardedFlatIndex:
    """Exact top-k over a corpus row-sharded across the ranks of a process group."""

    def __init__(self, d: int, metric: str = "l2", dtype: str = "f32", device: Optional[int] = None,
                 group=None, local_index=None, merge_fn: Optional[Callable] = None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.d, self.metric, self.dtype = int(d), metric, str(dtype).lower()
        self.n_total = 0
        if local_index is None:
            from .index import FlatIndex
            import torch
            dev = torch.cuda.current_device() if device is None else device
            local_index = FlatIndex(d, metric=metric, dtype=dtype, device=dev)
        self.local = local_index
        self._merge = merge_fn
        self._on_gpu = hasattr(self.local, "search_packed")

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
    def search_torch(self, q, k: int):
        """GPU path: q is a contiguous fp32 CUDA tensor [nq][d], identical on every rank.  Returns CUDA tensors
        (scores [nq][k], ids [nq][k]), identical on every rank."""
        import torch
        nq = q.shape[0]
        _, block = packed_layout(nq, k)
        gathered = torch.empty((self.world, block), dtype=torch.uint8, device=q.device)
        if self.world > 1:
            mine = torch.empty((block,), dtype=torch.uint8, device=q.device)
            self.local.search_packed(q, k, mine)
            self.dist.all_gather_into_tensor(gathered.view(-1), mine, group=self.group)     # THE collective
        else:
            self.local.search_packed(q, k, gathered[0])
        from .index import topk_merge_packed_torch
        return topk_merge_packed_torch(gathered, nq, self.world, k, k, self.metric)

    def search(self, Q, k: int):
        """Q: [nq][d] fp32 (NumPy, or a CUDA tensor on the GPU path), identical on every rank.
        Returns (scores [nq][k], ids [nq][k]) as NumPy arrays, identical on every rank."""
        import torch
        if self._on_gpu:
            q = Q if torch.is_tensor(Q) else torch.from_numpy(np.ascontiguousarray(Q, dtype=np.float32)).cuda()
            D, I = self.search_torch(q.contiguous(), k)
            return D.cpu().numpy(), I.cpu().numpy()
        # host engines (the CPU tests): same block layout, same single collective
        Dn, In = self.local.search(np.asarray(Q, dtype=np.float32), k)
        nq = Dn.shape[0]
        off, block = packed_layout(nq, k)
        mine = np.zeros(block, dtype=np.uint8)
        mine[:nq * k * 4] = np.ascontiguousarray(Dn, dtype=np.float32).view(np.uint8).reshape(-1)
        mine[off:off + nq * k * 8] = np.ascontiguousarray(In, dtype=np.int64).view(np.uint8).reshape(-1)
        gathered = torch.empty((self.world, block), dtype=torch.uint8)
        if self.world > 1:
            self.dist.all_gather_into_tensor(gathered.view(-1), torch.from_numpy(mine), group=self.group)
        else:
            gathered[0] = torch.from_numpy(mine)
        g = gathered.numpy()
        Dm = np.concatenate([g[r, :nq * k * 4].view(np.float32).reshape(nq, k) for r in range(self.world)], axis=1)
        Im = np.concatenate([g[r, off:off + nq * k * 8].view(np.int64).reshape(nq, k) for r in range(self.world)], axis=1)
        if self._merge is None:
            if self.world == 1:
                return Dn, In                      # one rank: the local lists are the answer
            raise RuntimeError("a host engine needs a merge_fn (the HIP merge runs on device tensors only)")
        D, I = self._merge(torch.from_numpy(Dm), torch.from_numpy(Im), k, self.metric)
        return D.numpy(), I.numpy()


    # ---- persistence: one row-major shard file per rank
    def save_local(self, folder_path: str, index_name: str = "index") -> None:
        """Rank r writes `<name>.shard<r>.rows.npy` -- its rows in the storage type (bf16 bit patterns as uint16, or
        fp32), row-major, streamed in slabs -- and `<name>.shard<r>.json` (its global row range); rank 0 adds
        `<name>.shards.json`.  No collective besides a closing barrier; nothing is unpickled on load."""
        import json
        import os
        os.makedirs(folder_path, exist_ok=True)
        n = self.local.ntotal
        lo = int(getattr(self.local, "id_base", None) if getattr(self.local, "id_base", None) is not None
                 else shard_range(self.n_total, self.world, self.rank)[0])
        # the file holds what reconstruct_raw returns (bf16 bit patterns or fp32 rows): its dtype comes from the data
        npdt = np.asarray(self.local.reconstruct_raw(0, min(n, 1))).dtype if n else (np.uint16 if self.dtype in ("bf16", "bfloat16") else np.float32)
        path = os.path.join(folder_path, f"{index_name}.shard{self.rank:03d}.rows.npy")
        mm = np.lib.format.open_memmap(path, mode="w+", dtype=npdt, shape=(n, self.d))
        slab = max(1, (256 << 20) // (self.d * 4))
        for r0 in range(0, n, slab):
            m = min(slab, n - r0)
            mm[r0:r0 + m] = self.local.reconstruct_raw(r0, m)
        mm.flush()
        del mm
        with open(os.path.join(folder_path, f"{index_name}.shard{self.rank:03d}.json"), "w") as f:
            json.dump({"row0": lo, "rows": n}, f)
        if self.rank == 0:
            with open(os.path.join(folder_path, f"{index_name}.shards.json"), "w") as f:
                json.dump({"format": "mi355rag-shards-1", "n_total": self.n_total, "d": self.d, "dtype": self.dtype,
                           "metric": self.metric, "world": self.world}, f)
        if self.world > 1:
            self.dist.barrier(group=self.group)

    @classmethod
    def load_local(cls, folder_path: str, index_name: str = "index", **kwargs) -> "ShardedFlatIndex":
        """Rebuilds the sharded index on the CURRENT process group: rank r takes rows shard_range(n_total, world, r),
        read (memory-mapped) from whichever shard files hold them, so a store saved on 8 ranks loads on 1, 2 or 4."""
        import json
        import os
        with open(os.path.join(folder_path, f"{index_name}.shards.json")) as f:
            meta = json.load(f)
        if meta.get("format") != "mi355rag-shards-1":
            raise ValueError(f"{folder_path}: not a mi355rag shard set")
        sh = cls(meta["d"], metric=meta["metric"], dtype=meta["dtype"], **kwargs)
        lo, hi = shard_range(meta["n_total"], sh.world, sh.rank)
        sh.local.set_id_base(lo)
        slab = max(1, (256 << 20) // (int(meta["d"]) * 4))
        for r in range(meta["world"]):
            with open(os.path.join(folder_path, f"{index_name}.shard{r:03d}.json")) as f:
                part = json.load(f)
            a, b = max(lo, part["row0"]), min(hi, part["row0"] + part["rows"])
            if a >= b:
                continue
            rows = np.load(os.path.join(folder_path, f"{index_name}.shard{r:03d}.rows.npy"), allow_pickle=False, mmap_mode="r")
            for r0 in range(a, b, slab):
                sh.local.add_raw(np.ascontiguousarray(rows[r0 - part["row0"]:min(b, r0 + slab) - part["row0"]]))
        sh.n_total = meta["n_total"]
        return sh


def embed_sharded(encoder, flat_ids: np.ndarray, offsets: np.ndarray, group=None, gather: bool = True,
                  max_len: int = 0) -> np.ndarray:
    """Data-parallel bulk embedding (BASELINE config 3): every rank is handed the same n ragged sequences (CSR),
    encodes the contiguous range shard_range(n, world, rank) with `encoder.encode_ragged` -- no collective -- and,
    when `gather`, the [n][hidden] result is all-gathered so every rank returns all rows; otherwise a rank
    returns only its own rows.  `encoder`: rag_amd.embeddings.Encoder, or anything with its encode_ragged."""
    import torch
    import torch.distributed as dist
    on = dist.is_initialized()
    rank = dist.get_rank(group) if on else 0
    world = dist.get_world_size(group) if on else 1
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    n = offsets.size - 1
    lo, hi = shard_range(n, world, rank)
    sub_off = offsets[lo:hi + 1] - offsets[lo]
    sub_ids = np.ascontiguousarray(flat_ids[offsets[lo]:offsets[hi]], dtype=np.int32)
    E = encoder.encode_ragged(sub_ids, sub_off, max_len=max_len) if hi > lo else np.zeros((0, encoder.hidden_size), np.float32)
    if not gather or world == 1:
        return E
    per = (n + world - 1) // world
    H = E.shape[1]
    backend = dist.get_backend(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    mine = torch.zeros((per, H), dtype=torch.float32, device=dev)
    mine[:hi - lo] = torch.from_numpy(E).to(dev)
    out = torch.empty((world * per, H), dtype=torch.float32, device=dev)
    dist.all_gather_into_tensor(out, mine, group=group)
    return out[:n].cpu().numpy()
