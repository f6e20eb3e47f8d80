"""GPU: the allocation-free entry point under HIP graph capture -- the callers it exists for.
mi_encoder_encode_dev_noalloc is captured into a graph on a side stream (torch.cuda.CUDAGraph: any hipMalloc, hipFree or
synchronising call inside the library would abort the capture), replayed, replayed again after the static input tensor
was overwritten, and each replay must equal a direct call on the same input bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_noalloc_entry_point_captures_into_a_graph_and_replays():
    import torch
    from rag_amd._capi import BertConfig
    from rag_amd.embeddings import Encoder
    from rag_amd.encoder_weights import random_blob
    cfg = BertConfig(2000, 384, 6, 12, 1536, 512, 2, 1e-12, 1)
    enc = Encoder(cfg, random_blob(cfg, seed=3, init_range=0.05))
    rng = np.random.default_rng(5)
    B, L = 3, 48
    ids = torch.from_numpy(rng.integers(0, 2000, size=(B, L)).astype(np.int32)).cuda()
    mask = torch.ones_like(ids)
    mask[1, 30:] = 0
    out = torch.empty((B, 384), dtype=torch.float32, device="cuda")
    enc.encode_ids_torch_noalloc(ids, mask, out)
    torch.cuda.synchronize()
    want = out.clone()

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        enc.encode_ids_torch_noalloc(ids, mask, out)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        enc.encode_ids_torch_noalloc(ids, mask, out)
    out.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out.view(torch.int32), want.view(torch.int32))

    ids2 = torch.from_numpy(rng.integers(0, 2000, size=(B, L)).astype(np.int32)).cuda()
    ids.copy_(ids2)                                        # the graph reads the same device addresses
    graph.replay()
    torch.cuda.synchronize()
    replayed = out.clone()
    direct = enc.encode_ids_torch(ids2, mask)              # the allocating entry point, outside any capture
    torch.cuda.synchronize()
    assert torch.equal(replayed.view(torch.int32), direct.view(torch.int32))
    assert float(replayed.abs().sum()) > 0.0
