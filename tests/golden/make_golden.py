#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/ (run in the build container only).

The reference (alxd/rag advanced_rag.py) ships no tests, fixtures or golden vectors, and
cannot be imported here (langchain / faiss / sentence-transformers / gradio are absent and
it loads models by hub name at import time), so the fixtures come from INDEPENDENT
restatements of the two algorithms, not from oracle/ and not from rag_amd/:

  search_*.npz   exact top-k by a pure-Python fmaf chain built on exact rational arithmetic
                 (fractions.Fraction + explicit round-to-nearest-even to binary32), ordered by
                 (score, id).  Includes planted duplicate rows and exact ties.
  cross_encoder_small.npz  transformers.BertForSequenceClassification (the class a sentence-transformers
                 CrossEncoder wraps) from a LOCAL BertConfig with seeded weights, CPU fp32, eager attention,
                 one label: (query, passage) pair batches with token type ids -> logits.
  encoder_small.npz  transformers.BertModel (the library the reference's HuggingFaceEmbeddings
                 runs, requirements.txt:25) built from a LOCAL BertConfig with seeded weights,
                 CPU fp32, eager attention, + sentence-transformers' mean-pool / normalise
                 restated in torch.  Stores the packed weights blob, inputs, per-layer hidden
                 states and final embeddings.

Usage:  python tests/golden/make_golden.py
"""
from __future__ import annotations

import os
import sys
from fractions import Fraction

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


# ----------------------------------------------------------------- exact binary32 arithmetic
def f32_round(x: Fraction) -> np.float32:
    """Round an exact rational to the nearest binary32 (ties to even), normal range only."""
    if x == 0:
        return np.float32(0.0)
    sign = -1 if x < 0 else 1
    ax = abs(x)
    # ax = m * 2^e with 2^23 <= m < 2^24
    e = ax.numerator.bit_length() - ax.denominator.bit_length() - 24
    while ax / Fraction(2) ** e >= 2 ** 24:
        e += 1
    while ax / Fraction(2) ** e < 2 ** 23:
        e -= 1
    assert e >= -149, "subnormal range not supported in the fixture generator"
    q = ax / Fraction(2) ** e
    fl = q.numerator // q.denominator
    rem = q - fl
    if rem > Fraction(1, 2) or (rem == Fraction(1, 2) and (fl & 1)):
        fl += 1
    val = Fraction(fl) * Fraction(2) ** e
    return np.float32(sign * float(val))       # exact: fl has <= 25 significant bits


def fmaf(a: np.float32, b: np.float32, c: np.float32) -> np.float32:
    return f32_round(Fraction(float(a)) * Fraction(float(b)) + Fraction(float(c)))


def canon_dot(q: np.ndarray, x: np.ndarray) -> np.float32:
    acc = np.float32(0.0)
    for a, b in zip(q, x):
        acc = fmaf(a, b, acc)
    return acc


def bf16_round(x: np.ndarray) -> np.ndarray:
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)


def exact_topk(X, Q, k, metric, bf16):
    if bf16:
        X, Q = bf16_round(X), bf16_round(Q)
    n = X.shape[0]
    xx = [canon_dot(x, x) for x in X] if metric == "l2" else None
    scores = np.empty((Q.shape[0], k), dtype=np.float32)
    ids = np.empty((Q.shape[0], k), dtype=np.int64)
    for qi, q in enumerate(Q):
        qq = canon_dot(q, q) if metric == "l2" else None
        vals = []
        for i in range(n):
            dot = canon_dot(q, X[i])
            if metric == "l2":
                dist = fmaf(np.float32(-2.0), dot, np.float32(qq + xx[i]))   # qq + xx: one fp32 add
                if dist < 0:
                    dist = np.float32(0.0)
                vals.append((float(dist), i))
            else:
                vals.append((-float(dot), i))
        vals.sort()                              # (score best-first, id ascending)
        for j in range(k):
            if j < n:
                s, i = vals[j]
                scores[qi, j] = s if metric == "l2" else -s
                ids[qi, j] = i
            else:
                scores[qi, j] = np.inf if metric == "l2" else -np.inf
                ids[qi, j] = -1
    return scores, ids


def make_search():
    rng = np.random.default_rng(20250824)
    n, d, nq = 96, 24, 6
    X = rng.standard_normal((n, d)).astype(np.float32)
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    X[40:47] = X[3]            # duplicate rows -> exact ties, broken by id
    X[90] = X[3]
    X[60] = 0.0                # a zero row
    Q = rng.standard_normal((nq, d)).astype(np.float32)
    Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    Q[0] = X[3]
    Q[1] = -X[3]
    out = {"X": X, "Q": Q}
    for metric in ("ip", "l2"):
        for bf16 in (False, True):
            for k in (1, 4, 5, 10, 100):
                s, i = exact_topk(X, Q, k, metric, bf16)
                tag = f"{metric}_{'bf16' if bf16 else 'f32'}_k{k}"
                out[f"scores_{tag}"] = s
                out[f"ids_{tag}"] = i
                print("search", tag, "done", flush=True)
    np.savez_compressed(os.path.join(HERE, "search_small.npz"), **out)


# ----------------------------------------------------------------- encoder (transformers BertModel)
def make_encoder():
    import torch
    from transformers import BertConfig, BertModel
    from rag_amd._capi import BertConfig as MiCfg
    from rag_amd.encoder_weights import pack_state_dict

    torch.manual_seed(1234)
    hf = BertConfig(vocab_size=300, hidden_size=128, num_hidden_layers=2, num_attention_heads=4,
                    intermediate_size=256, max_position_embeddings=64, type_vocab_size=2,
                    layer_norm_eps=1e-12, hidden_act="gelu", hidden_dropout_prob=0.0,
                    attention_probs_dropout_prob=0.0)
    hf._attn_implementation = "eager"
    model = BertModel(hf, add_pooling_layer=False).eval()
    with torch.no_grad():                         # make biases / LN params non-trivial
        for name, p in model.named_parameters():
            if name.endswith("bias"):
                p.normal_(0.0, 0.05)
            elif "LayerNorm.weight" in name:
                p.normal_(1.0, 0.05)
            else:
                p.normal_(0.0, 0.08)
    B, L = 5, 40
    g = torch.Generator().manual_seed(99)
    ids = torch.randint(0, hf.vocab_size, (B, L), generator=g)
    lens = [40, 33, 17, 2, 1]
    mask = torch.zeros((B, L), dtype=torch.long)
    for b, n in enumerate(lens):
        mask[b, :n] = 1
    with torch.no_grad():
        out = model(input_ids=ids, attention_mask=mask, output_hidden_states=True)
        h = out.last_hidden_state
        m = mask.unsqueeze(-1).float()
        pooled = (h * m).sum(1) / torch.clamp(m.sum(1), min=1e-9)
        emb = torch.nn.functional.normalize(pooled, p=2, dim=1)
    cfg = MiCfg(vocab_size=hf.vocab_size, hidden_size=hf.hidden_size, num_hidden_layers=hf.num_hidden_layers,
                num_attention_heads=hf.num_attention_heads, intermediate_size=hf.intermediate_size,
                max_position_embeddings=hf.max_position_embeddings, type_vocab_size=hf.type_vocab_size,
                layer_norm_eps=hf.layer_norm_eps, normalize=1)
    blob = pack_state_dict(model.state_dict(), cfg)
    np.savez_compressed(
        os.path.join(HERE, "encoder_small.npz"),
        config=np.array([hf.vocab_size, hf.hidden_size, hf.num_hidden_layers, hf.num_attention_heads,
                         hf.intermediate_size, hf.max_position_embeddings, hf.type_vocab_size], dtype=np.int64),
        layer_norm_eps=np.float64(hf.layer_norm_eps), blob=blob.astype(np.float32),
        ids=ids.numpy().astype(np.int32), mask=mask.numpy().astype(np.int32),
        hidden_states=torch.stack(out.hidden_states).numpy().astype(np.float32),   # [layers+1][B][L][H]
        pooled=pooled.numpy().astype(np.float32), embeddings=emb.numpy().astype(np.float32))
    print("encoder fixture done; blob elems", blob.size, flush=True)


def make_cross_encoder():
    import torch
    from transformers import BertConfig, BertForSequenceClassification
    from rag_amd._capi import BertConfig as MiCfg
    from rag_amd.encoder_weights import head_from_state_dict, pack_state_dict

    torch.manual_seed(4321)
    hf = BertConfig(vocab_size=300, hidden_size=128, num_hidden_layers=2, num_attention_heads=4,
                    intermediate_size=256, max_position_embeddings=64, type_vocab_size=2,
                    layer_norm_eps=1e-12, hidden_act="gelu", hidden_dropout_prob=0.0,
                    attention_probs_dropout_prob=0.0, num_labels=1)
    hf._attn_implementation = "eager"
    model = BertForSequenceClassification(hf).eval()
    with torch.no_grad():
        for name, p in model.named_parameters():
            if name.endswith("bias"):
                p.normal_(0.0, 0.05)
            elif "LayerNorm.weight" in name:
                p.normal_(1.0, 0.05)
            else:
                p.normal_(0.0, 0.08)
    B, L = 12, 48
    g = torch.Generator().manual_seed(7)
    ids = torch.randint(3, hf.vocab_size, (B, L), generator=g)
    qlen = 9                                            # [CLS] query [SEP] | passage [SEP]
    lens = [48, 48, 40, 33, 31, 26, 20, 17, 15, 12, 11, 10]
    mask = torch.zeros((B, L), dtype=torch.long)
    types = torch.zeros((B, L), dtype=torch.long)
    for b, n in enumerate(lens):
        mask[b, :n] = 1
        types[b, qlen:n] = 1
        ids[b, :qlen] = ids[0, :qlen]                   # the same query in every pair
        ids[b, n:] = 0
    with torch.no_grad():
        logits = model(input_ids=ids, attention_mask=mask, token_type_ids=types).logits
    cfg = MiCfg(vocab_size=hf.vocab_size, hidden_size=hf.hidden_size, num_hidden_layers=hf.num_hidden_layers,
                num_attention_heads=hf.num_attention_heads, intermediate_size=hf.intermediate_size,
                max_position_embeddings=hf.max_position_embeddings, type_vocab_size=hf.type_vocab_size,
                layer_norm_eps=hf.layer_norm_eps, normalize=0)
    sd = model.state_dict()
    blob = pack_state_dict(sd, cfg)
    pw, pb, cw, cb = head_from_state_dict(sd, cfg)
    np.savez_compressed(
        os.path.join(HERE, "cross_encoder_small.npz"),
        config=np.array([hf.vocab_size, hf.hidden_size, hf.num_hidden_layers, hf.num_attention_heads,
                         hf.intermediate_size, hf.max_position_embeddings, hf.type_vocab_size], dtype=np.int64),
        layer_norm_eps=np.float64(hf.layer_norm_eps), blob=blob.astype(np.float32),
        pooler_w=pw, pooler_b=pb, cls_w=cw, cls_b=cb,
        ids=ids.numpy().astype(np.int32), types=types.numpy().astype(np.int32), mask=mask.numpy().astype(np.int32),
        logits=logits.numpy().astype(np.float32))
    print("cross-encoder fixture done; logits", logits.numpy().ravel()[:4], flush=True)


if __name__ == "__main__":
    which = sys.argv[1:] or ["encoder", "search", "cross"]
    if "encoder" in which:
        make_encoder()
    if "cross" in which:
        make_cross_encoder()
    if "search" in which:
        make_search()
