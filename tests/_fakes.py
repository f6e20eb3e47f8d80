"""Test doubles: an oracle-backed stand-in for rag_amd.index.FlatIndex so the HOST logic
(vector store, retriever, sharding) can be covered without a GPU.  Tests only."""
import numpy as np

import oracle


class OracleIndex:
    def __init__(self, d, metric="l2", dtype="f32"):
        self.d = d
        self.metric = {"l2": oracle.METRIC_L2, "ip": oracle.METRIC_IP}[metric]
        self.dtype = {"f32": oracle.DTYPE_F32, "bf16": oracle.DTYPE_BF16}[dtype]
        self.X = np.zeros((0, d), dtype=np.float32)
        self.id_base = 0

    @property
    def ntotal(self):
        return self.X.shape[0]

    def add(self, x):
        x = np.asarray(x, dtype=np.float32)
        assert x.ndim == 2 and x.shape[1] == self.d
        self.X = np.concatenate([self.X, x])

    def add_synthetic(self, n, seed=0, normalize=True):
        self.add(oracle.synth_rows(seed, self.id_base + self.ntotal, n, self.d, normalize))

    def set_id_base(self, b):
        self.id_base = int(b)

    def search(self, q, k):
        return oracle.search(self.X, np.asarray(q, dtype=np.float32), k, metric=self.metric,
                             dtype=self.dtype, id_base=self.id_base)

    def reconstruct_n(self, row0, n):
        x = self.X[row0:row0 + n]
        if self.dtype == oracle.DTYPE_BF16:
            x = oracle.bf16_to_f32(oracle.round_bf16(x))
        return np.ascontiguousarray(x, dtype=np.float32)

    def reconstruct_raw(self, row0, n):
        x = self.X[row0:row0 + n]
        return oracle.round_bf16(x) if self.dtype == oracle.DTYPE_BF16 else np.ascontiguousarray(x, dtype=np.float32)

    def add_raw(self, x):
        self.add(oracle.bf16_to_f32(x) if x.dtype == np.uint16 else x)

    def reset(self):
        self.X = self.X[:0]


class HashEmbeddings:
    """Deterministic bag-of-words embedding (tests only): no model, no GPU."""

    def __init__(self, d=32):
        self.d = d

    def _one(self, text):
        v = np.zeros(self.d, dtype=np.float32)
        for w in text.lower().split():
            h = 0
            for ch in w:
                h = (h * 131 + ord(ch)) % 1000003
            v[h % self.d] += 1.0
        n = np.linalg.norm(v)
        return v / n if n > 0 else v

    def embed_documents(self, texts):
        return [self._one(t).tolist() for t in texts]

    def embed_query(self, text):
        return self._one(text).tolist()


class OracleEncoder:
    """Oracle-backed stand-in for rag_amd.embeddings.Encoder.encode_ragged (one sequence per oracle call, padded
    to its own length), so the data-parallel embedding split can be covered without a GPU.  Tests only."""

    def __init__(self, cfg, blob):
        self.cfg, self.blob = cfg, blob
        self.hidden_size = int(cfg.hidden_size)
        self.calls = 0

    def encode_ragged(self, flat_ids, offsets, max_len=0, max_batch=0):
        n = len(offsets) - 1
        out = np.zeros((n, self.hidden_size), dtype=np.float32)
        for i in range(n):
            ids = np.asarray(flat_ids[offsets[i]:offsets[i + 1]], dtype=np.int32)
            if max_len:
                ids = ids[:max_len]
            if ids.size:
                out[i] = oracle.encode(self.cfg, self.blob, ids[None, :], np.ones((1, ids.size), dtype=np.int32))[0]
        self.calls += 1
        return out
