"""CPU: the encoder oracle (oracle/oracle_encoder.c) against the committed transformers
BertModel fixture, and against a live BertModel at MiniLM-L6 dimensions."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden", "encoder_small.npz")


def _cfg(oracle, g):
    c = [int(v) for v in g["config"]]
    return oracle.BertConfig(c[0], c[1], c[2], c[3], c[4], c[5], c[6], float(g["layer_norm_eps"]), 1)


def test_oracle_matches_bertmodel_fixture(oracle_mod):
    g = np.load(GOLD)
    cfg = _cfg(oracle_mod, g)
    assert g["blob"].size == oracle_mod.blob_elems(cfg)
    E = oracle_mod.encode(cfg, g["blob"], g["ids"], g["mask"])
    assert np.abs(E - g["embeddings"]).max() < 2e-5
    cos = (E * g["embeddings"]).sum(1)
    assert np.all(1 - cos < 1e-6)
    # stage by stage on the longest sequence (positions beyond the mask are compared too:
    # padded rows are still computed, only masked as keys)
    out, tap_emb, tap_layers = oracle_mod.encode_taps(cfg, g["blob"], g["ids"][0], g["mask"][0])
    hs = g["hidden_states"]
    assert np.abs(tap_emb - hs[0, 0]).max() < 2e-5
    for l in range(cfg.num_hidden_layers):
        assert np.abs(tap_layers[l] - hs[l + 1, 0]).max() < 5e-5, l
    assert np.abs(out - g["embeddings"][0]).max() < 2e-5


def test_oracle_matches_live_bertmodel_minilm_dims(oracle_mod):
    torch = pytest.importorskip("torch")
    tr = pytest.importorskip("transformers")
    from rag_amd._capi import BertConfig as MiCfg
    from rag_amd.encoder_weights import pack_state_dict
    torch.manual_seed(0)
    hf = tr.BertConfig(vocab_size=2000, hidden_size=384, num_hidden_layers=6, num_attention_heads=12,
                       intermediate_size=1536, max_position_embeddings=512, type_vocab_size=2,
                       layer_norm_eps=1e-12, hidden_act="gelu")
    model = tr.BertModel(hf, add_pooling_layer=False).eval()
    ids = torch.randint(0, 2000, (2, 48))
    mask = torch.ones((2, 48), dtype=torch.long)
    mask[1, 30:] = 0
    with torch.no_grad():
        h = model(input_ids=ids, attention_mask=mask).last_hidden_state
        m = mask.unsqueeze(-1).float()
        want = torch.nn.functional.normalize((h * m).sum(1) / m.sum(1).clamp(min=1e-9), dim=1).numpy()
    cfg = MiCfg(2000, 384, 6, 12, 1536, 512, 2, 1e-12, 1)
    blob = pack_state_dict(model.state_dict(), cfg)
    ocfg = oracle_mod.BertConfig(2000, 384, 6, 12, 1536, 512, 2, 1e-12, 1)
    got = oracle_mod.encode(ocfg, blob, ids.numpy(), mask.numpy())
    assert np.all(1 - (got * want).sum(1) < 1e-6)
    assert np.abs(got - want).max() < 2e-5


def test_all_masked_sequence_gives_zero_vector(oracle_mod):
    g = np.load(GOLD)
    cfg = _cfg(oracle_mod, g)
    ids = g["ids"][:1]
    E = oracle_mod.encode(cfg, g["blob"], ids, np.zeros_like(ids))
    assert np.all(E == 0)


def test_bad_token_id_is_rejected(oracle_mod):
    g = np.load(GOLD)
    cfg = _cfg(oracle_mod, g)
    ids = g["ids"][:1].copy()
    ids[0, 0] = 10**6
    with pytest.raises(ValueError):
        oracle_mod.encode(cfg, g["blob"], ids, g["mask"][:1])


XGOLD = os.path.join(os.path.dirname(__file__), "golden", "cross_encoder_small.npz")


def test_oracle_classify_matches_sequence_classification_fixture(oracle_mod):
    """Cross-encoder head (token type ids + BertPooler + classifier) against the committed
    transformers.BertForSequenceClassification fixture (tests/golden/make_golden.py cross)."""
    g = np.load(XGOLD)
    c = [int(v) for v in g["config"]]
    cfg = oracle_mod.BertConfig(c[0], c[1], c[2], c[3], c[4], c[5], c[6], float(g["layer_norm_eps"]), 0)
    head = (g["pooler_w"], g["pooler_b"], g["cls_w"], g["cls_b"])
    got = oracle_mod.classify(cfg, g["blob"], head, g["ids"], g["types"], g["mask"])
    assert got.shape == g["logits"].shape
    assert np.abs(got - g["logits"]).max() < 5e-5
    # the type ids matter: all-zero types give different logits
    other = oracle_mod.classify(cfg, g["blob"], head, g["ids"], None, g["mask"])
    assert np.abs(other - g["logits"]).max() > 1e-3
    with pytest.raises(ValueError):
        oracle_mod.classify(cfg, g["blob"], head, g["ids"], g["types"] + 5, g["mask"])
