"""GPU: the allocation-free entry point (mi_encoder_encode_dev_noalloc) against the entry point it is a variant of
(mi_encoder_encode_dev), through the C-ABI: same kernels, so the outputs are held to bit equality; two calls in a
row, the second with a smaller input (the shared scratch keeps the first call's contents); sizes beyond the
array are refused.  Collected last (test_zz_*): a failure here must not hide the tests before it."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_noalloc_entry_point_matches_encode_dev(oracle_mod):
    import torch
    from rag_amd import MI355Error
    from rag_amd._capi import BertConfig
    from rag_amd.embeddings import Encoder
    from rag_amd.encoder_weights import random_blob
    vals = (2000, 384, 6, 12, 1536, 512, 2, 1e-12, 1)
    cfg, ocfg = BertConfig(*vals), oracle_mod.BertConfig(*vals)
    blob = random_blob(cfg, seed=9, init_range=0.05)
    enc = Encoder(cfg, blob)
    rng = np.random.default_rng(12)

    def batch(B, L):
        ids = rng.integers(0, 2000, size=(B, L)).astype(np.int32)
        mask = np.ones((B, L), dtype=np.int32)
        for b in range(1, B):
            mask[b, rng.integers(1, L + 1):] = 0
        return ids, mask, torch.from_numpy(ids).cuda(), torch.from_numpy(mask).cuda()

    first = batch(7, 256)                                  # 1,792 tokens: close to what the array holds
    second = batch(2, 19)                                  # then a smaller one into the same scratch
    outs = []
    for ids, mask, ids_d, mask_d in (first, second):
        got = enc.encode_ids_torch_noalloc(ids_d, mask_d)
        outs.append(got)
    torch.cuda.synchronize()
    for (ids, mask, ids_d, mask_d), got in zip((first, second), outs):
        want = enc.encode_ids_torch(ids_d, mask_d)
        torch.cuda.synchronize()
        assert torch.equal(got.view(torch.int32), want.view(torch.int32))
        Eo = oracle_mod.encode(ocfg, blob, ids[:2], mask[:2])
        E = got[:2].cpu().numpy()
        cos = (E * Eo).sum(1) / (np.linalg.norm(E, axis=1) * np.linalg.norm(Eo, axis=1))
        assert np.all(1.0 - cos < 1e-3)
    # once more, interleaved with the allocating entry point, which must keep working as it did
    again = enc.encode_ids_torch_noalloc(first[2], first[3])
    torch.cuda.synchronize()
    assert torch.equal(again.view(torch.int32), outs[0].view(torch.int32))
    # beyond the array: a clear error, no allocation, no launch
    ids, mask, ids_d, mask_d = batch(64, 256)
    with pytest.raises(MI355Error, match="workspace"):
        enc.encode_ids_torch_noalloc(ids_d, mask_d)
