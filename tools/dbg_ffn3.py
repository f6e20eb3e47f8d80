"""Debug: ffn3_kernel against ffn2_kernel (MI_FFN3 is read per call) on one-layer encoders with parts of the layer zeroed."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rag_amd._capi import BertConfig
from rag_amd.embeddings import Encoder
from rag_amd.encoder_weights import random_blob, unpack_blob, pack_state_dict

os.environ["MI_ENC_LATENCY_TOKENS"] = "0"
cfg = BertConfig(2000, 384, 1, 12, 1536, 512, 2, 1e-12, 1)
base = unpack_blob(cfg, random_blob(cfg, seed=3, init_range=0.05))
rng = np.random.default_rng(0)
ids = rng.integers(0, 2000, size=(6, 128)).astype(np.int32)
mask = np.ones_like(ids)
P = "encoder.layer.0."
cases = {
    "full": [],
    "no_ffn1": [P + "intermediate.dense.weight", P + "intermediate.dense.bias"],
    "no_ffn2": [P + "output.dense.weight"],
    "no_wo": [P + "attention.output.dense.weight"],
    "no_wo_no_ffn": [P + "attention.output.dense.weight", P + "intermediate.dense.weight", P + "intermediate.dense.bias", P + "output.dense.weight"],
}
for name, zero in cases.items():
    sd = {k: np.array(v) for k, v in base.items()}
    for k in zero:
        sd[k][...] = 0
    enc = Encoder(cfg, pack_state_dict(sd, cfg))
    os.environ["MI_FFN3"] = "0"
    E2 = enc.encode_ids(ids, mask)
    os.environ["MI_FFN3"] = "1"
    E3 = enc.encode_ids(ids, mask)
    cos = 1 - (E2 * E3).sum(1) / (np.linalg.norm(E2, axis=1) * np.linalg.norm(E3, axis=1))
    print(f"{name:14s} max 1-cos ffn3 vs ffn2: {cos.max():.3e}  finite {np.isfinite(E3).all()}", flush=True)

# ---- channel-level look at the simplest failing case
cfg0 = BertConfig(2000, 384, 1, 12, 1536, 512, 2, 1e-12, 0)
sd = {k: np.array(v) for k, v in base.items()}
for k in cases["no_wo_no_ffn"]:
    sd[k][...] = 0
enc = Encoder(cfg0, pack_state_dict(sd, cfg0))
os.environ["MI_FFN3"] = "0"; E2 = enc.encode_ids(ids, mask)
os.environ["MI_FFN3"] = "1"; E3 = enc.encode_ids(ids, mask)
np.set_printoptions(precision=4, suppress=True, linewidth=200)
print("E2[0,:24]", E2[0, :24]); print("E3[0,:24]", E3[0, :24])
print("sorted equal?", np.abs(np.sort(E2[0]) - np.sort(E3[0])).max(), "abs diff max", np.abs(E2 - E3).max())
d = np.abs(E2[0] - E3[0]); print("channels with diff > 1e-2:", np.nonzero(d > 1e-2)[0][:64], (d > 1e-2).sum())
