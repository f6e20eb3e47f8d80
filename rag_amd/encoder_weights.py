"""Packs HuggingFace BERT parameters into the flat fp32 blob mi_encoder_create expects.

Blob order (all fp32; linears keep torch.nn.Linear's [out][in] layout):
  word_emb[V][H] pos_emb[P][H] type_emb[T][H] emb_ln_g[H] emb_ln_b[H]
  per layer: Wq bq Wk bk Wv bv Wo bo ln1_g ln1_b Wi bi Wd bd ln2_g ln2_b
Parameter names follow transformers' BertModel state_dict (modeling_bert.py); a leading
"bert." / "0.auto_model." prefix (sentence-transformers checkpoints) is stripped.
"""
from __future__ import annotations

import json
import os
from typing import Dict, Mapping

import numpy as np

from ._capi import BertConfig


def config_from_dict(cfg: Mapping, normalize: bool = True) -> BertConfig:
    act = cfg.get("hidden_act", "gelu")
    if act != "gelu":
        raise ValueError(f"only hidden_act='gelu' (erf) is supported, got {act!r}")
    if cfg.get("position_embedding_type", "absolute") != "absolute":
        raise ValueError("only absolute position embeddings are supported")
    return BertConfig(
        vocab_size=int(cfg["vocab_size"]), hidden_size=int(cfg["hidden_size"]),
        num_hidden_layers=int(cfg["num_hidden_layers"]), num_attention_heads=int(cfg["num_attention_heads"]),
        intermediate_size=int(cfg["intermediate_size"]),
        max_position_embeddings=int(cfg["max_position_embeddings"]),
        type_vocab_size=int(cfg.get("type_vocab_size", 2)),
        layer_norm_eps=float(cfg.get("layer_norm_eps", 1e-12)), normalize=int(bool(normalize)))


def minilm_l6_config(normalize: bool = True) -> BertConfig:
    """all-MiniLM-L6-v2 dimensions: hidden 384, 6 layers, 12 heads, FFN 1536."""
    return BertConfig(vocab_size=30522, hidden_size=384, num_hidden_layers=6, num_attention_heads=12,
                      intermediate_size=1536, max_position_embeddings=512, type_vocab_size=2,
                      layer_norm_eps=1e-12, normalize=int(bool(normalize)))


def _strip(name: str) -> str:
    for pre in ("0.auto_model.", "auto_model.", "bert.", "model."):
        if name.startswith(pre):
            return name[len(pre):]
    return name


def pack_state_dict(sd: Mapping[str, "np.ndarray"], cfg: BertConfig) -> np.ndarray:
    """state_dict (name -> array-like; torch tensors accepted) -> flat fp32 blob."""
    def get(name: str, shape) -> np.ndarray:
        t = g[name]
        a = t.detach().cpu().float().numpy() if hasattr(t, "detach") else np.asarray(t, dtype=np.float32)
        if tuple(a.shape) != tuple(shape):
            raise ValueError(f"{name}: expected shape {tuple(shape)}, got {tuple(a.shape)}")
        return np.ascontiguousarray(a, dtype=np.float32).reshape(-1)

    g: Dict[str, object] = {_strip(k): v for k, v in sd.items()}
    H, I = cfg.hidden_size, cfg.intermediate_size
    parts = [
        get("embeddings.word_embeddings.weight", (cfg.vocab_size, H)),
        get("embeddings.position_embeddings.weight", (cfg.max_position_embeddings, H)),
        get("embeddings.token_type_embeddings.weight", (cfg.type_vocab_size, H)),
        get("embeddings.LayerNorm.weight", (H,)),
        get("embeddings.LayerNorm.bias", (H,)),
    ]
    for l in range(cfg.num_hidden_layers):
        p = f"encoder.layer.{l}."
        parts += [
            get(p + "attention.self.query.weight", (H, H)), get(p + "attention.self.query.bias", (H,)),
            get(p + "attention.self.key.weight", (H, H)), get(p + "attention.self.key.bias", (H,)),
            get(p + "attention.self.value.weight", (H, H)), get(p + "attention.self.value.bias", (H,)),
            get(p + "attention.output.dense.weight", (H, H)), get(p + "attention.output.dense.bias", (H,)),
            get(p + "attention.output.LayerNorm.weight", (H,)), get(p + "attention.output.LayerNorm.bias", (H,)),
            get(p + "intermediate.dense.weight", (I, H)), get(p + "intermediate.dense.bias", (I,)),
            get(p + "output.dense.weight", (H, I)), get(p + "output.dense.bias", (H,)),
            get(p + "output.LayerNorm.weight", (H,)), get(p + "output.LayerNorm.bias", (H,)),
        ]
    return np.concatenate(parts)


def random_blob(cfg: BertConfig, seed: int = 0, init_range: float = 0.02) -> np.ndarray:
    """Seeded random weights with BERT's init (normal(0, init_range), LN gamma 1 / beta 0,
    zero biases perturbed slightly so bias paths are exercised) — synthetic-weights bench/test input."""
    rng = np.random.default_rng(seed)
    H, I = cfg.hidden_size, cfg.intermediate_size

    def w(*shape):
        return (rng.standard_normal(shape) * init_range).astype(np.float32).reshape(-1)

    def b(n):
        return (rng.standard_normal(n) * init_range).astype(np.float32)

    def gamma(n):
        return (1.0 + rng.standard_normal(n) * init_range).astype(np.float32)

    parts = [w(cfg.vocab_size, H), w(cfg.max_position_embeddings, H), w(cfg.type_vocab_size, H), gamma(H), b(H)]
    for _ in range(cfg.num_hidden_layers):
        parts += [w(H, H), b(H), w(H, H), b(H), w(H, H), b(H), w(H, H), b(H), gamma(H), b(H),
                  w(I, H), b(I), w(H, I), b(H), gamma(H), b(H)]
    return np.concatenate(parts)


def head_from_state_dict(sd: Mapping[str, "np.ndarray"], cfg: BertConfig):
    """(pooler_w, pooler_b, cls_w, cls_b) of a BertForSequenceClassification state_dict (cross-encoder)."""
    def arr(t):
        return t.detach().cpu().float().numpy() if hasattr(t, "detach") else np.asarray(t, dtype=np.float32)
    g = {_strip(k): v for k, v in sd.items()}
    H = cfg.hidden_size
    pw, pb = arr(g["pooler.dense.weight"]), arr(g["pooler.dense.bias"])
    cw, cb = arr(g["classifier.weight"]), arr(g["classifier.bias"])
    if pw.shape != (H, H) or pb.shape != (H,) or cw.ndim != 2 or cw.shape[1] != H or cb.shape != (cw.shape[0],):
        raise ValueError("classification head has unexpected shapes")
    return pw, pb, cw, cb


def random_head(cfg: BertConfig, num_labels: int = 1, seed: int = 0, init_range: float = 0.02):
    rng = np.random.default_rng(seed)
    H = cfg.hidden_size
    return ((rng.standard_normal((H, H)) * init_range).astype(np.float32), (rng.standard_normal(H) * init_range).astype(np.float32),
            (rng.standard_normal((num_labels, H)) * init_range).astype(np.float32),
            (rng.standard_normal(num_labels) * init_range).astype(np.float32))


def load_local_dir(path: str, normalize: bool = True, with_head: bool = False):
    """(BertConfig, blob) from a LOCAL directory holding config.json + model.safetensors
    (never a hub name: there is no network on the target machines)."""
    if not os.path.isdir(path):
        raise FileNotFoundError(
            f"model_name must be a local directory with config.json and model.safetensors, got {path!r}")
    with open(os.path.join(path, "config.json")) as f:
        cfg = config_from_dict(json.load(f), normalize=normalize)
    from safetensors.numpy import load_file
    st = os.path.join(path, "model.safetensors")
    if not os.path.exists(st):
        sub = os.path.join(path, "0_Transformer", "model.safetensors")
        st = sub if os.path.exists(sub) else st
    sd = load_file(st)
    if with_head:
        return cfg, pack_state_dict(sd, cfg), head_from_state_dict(sd, cfg)
    return cfg, pack_state_dict(sd, cfg)


def unpack_blob(cfg: BertConfig, blob: np.ndarray) -> Dict[str, np.ndarray]:
    """The inverse of pack_state_dict: flat fp32 blob -> transformers' BertModel parameter names."""
    H, I = cfg.hidden_size, cfg.intermediate_size
    blob = np.asarray(blob, dtype=np.float32).reshape(-1)
    pos = 0

    def take(*shape):
        nonlocal pos
        n = int(np.prod(shape))
        a = blob[pos:pos + n].reshape(shape)
        pos += n
        return a

    sd = {"embeddings.word_embeddings.weight": take(cfg.vocab_size, H),
          "embeddings.position_embeddings.weight": take(cfg.max_position_embeddings, H),
          "embeddings.token_type_embeddings.weight": take(cfg.type_vocab_size, H),
          "embeddings.LayerNorm.weight": take(H), "embeddings.LayerNorm.bias": take(H)}
    for l in range(cfg.num_hidden_layers):
        p = f"encoder.layer.{l}."
        for nm, shp in [("attention.self.query", (H, H)), ("attention.self.key", (H, H)), ("attention.self.value", (H, H)),
                        ("attention.output.dense", (H, H))]:
            sd[p + nm + ".weight"] = take(*shp)
            sd[p + nm + ".bias"] = take(shp[0])
        sd[p + "attention.output.LayerNorm.weight"] = take(H)
        sd[p + "attention.output.LayerNorm.bias"] = take(H)
        sd[p + "intermediate.dense.weight"] = take(I, H)
        sd[p + "intermediate.dense.bias"] = take(I)
        sd[p + "output.dense.weight"] = take(H, I)
        sd[p + "output.dense.bias"] = take(H)
        sd[p + "output.LayerNorm.weight"] = take(H)
        sd[p + "output.LayerNorm.bias"] = take(H)
    if pos != blob.size:
        raise ValueError(f"blob has {blob.size} elements, config needs {pos}")
    return sd


def synthetic_vocab(vocab_size: int):
    """A WordPiece vocabulary of `vocab_size` entries for benches and tests (no vocabulary file exists offline): the five
    BERT specials, whole words "w<n>" and continuation pieces "##x<n>", so "w17x4" tokenises to two pieces by the usual
    greedy longest-prefix match."""
    specials = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"]
    n_words = (vocab_size - len(specials)) * 2 // 3
    n_suffix = vocab_size - len(specials) - n_words
    return specials + [f"w{i}" for i in range(n_words)] + [f"##x{i}" for i in range(n_suffix)], n_words, n_suffix


def synthetic_texts(n: int, words_per_text: int, vocab_size: int, seed: int = 0, composite: float = 0.3):
    """n strings over synthetic_vocab(vocab_size): `words_per_text` space-separated words, a fraction `composite` of them a
    word + one continuation piece (two tokens)."""
    _, n_words, n_suffix = synthetic_vocab(vocab_size)
    rng = np.random.default_rng(seed)
    w = rng.integers(0, n_words, size=(n, words_per_text))
    sfx = rng.integers(0, n_suffix, size=(n, words_per_text))
    comp = rng.random((n, words_per_text)) < composite
    return [" ".join(f"w{a}x{b}" if c else f"w{a}" for a, b, c in zip(w[i], sfx[i], comp[i])) for i in range(n)]


def write_synthetic_model_dir(path: str, cfg: BertConfig, blob: np.ndarray) -> str:
    """A local model directory (config.json, vocab.txt, model.safetensors) for `cfg` with the weights of `blob` and
    synthetic_vocab(cfg.vocab_size): what HuggingFaceEmbeddings(model_name=<dir>) loads, written without any download."""
    from safetensors.numpy import save_file
    os.makedirs(path, exist_ok=True)
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump({"vocab_size": cfg.vocab_size, "hidden_size": cfg.hidden_size, "num_hidden_layers": cfg.num_hidden_layers,
                   "num_attention_heads": cfg.num_attention_heads, "intermediate_size": cfg.intermediate_size,
                   "max_position_embeddings": cfg.max_position_embeddings, "type_vocab_size": cfg.type_vocab_size,
                   "layer_norm_eps": cfg.layer_norm_eps, "hidden_act": "gelu"}, f)
    with open(os.path.join(path, "vocab.txt"), "w") as f:
        f.write("\n".join(synthetic_vocab(cfg.vocab_size)[0]) + "\n")
    save_file({k: np.ascontiguousarray(v) for k, v in unpack_blob(cfg, blob).items()}, os.path.join(path, "model.safetensors"))
    return path
