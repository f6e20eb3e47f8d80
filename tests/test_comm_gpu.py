"""GPU: the communicator behind the C-ABI (mi_comm_*: RCCL through dlopen) and mi_index_search_sharded_dev.
One MI355X per test box and RCCL refuses two ranks on one GPU, so this runs a world of ONE rank: the collective is a
real ncclAllGather on the library's stream, the merge reads the gathered block in place, and the answer must equal
mi_index_search bit for bit.  Worlds of 2 and 8 ranks are covered without the collective by tests/test_sharded_gpu.py
(same packed blocks, same merge) and with gloo by tests/test_distributed_cpu.py."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("metric,dtype,n,nq,k", [("ip", "bf16", 50_000, 33, 10), ("l2", "f32", 3_000, 5, 100)])
def test_world_of_one_equals_plain_search(oracle_mod, metric, dtype, n, nq, k):
    import torch
    from rag_amd import _capi
    from rag_amd.index import FlatIndex
    lib = _capi.load()
    d = 384
    rng = np.random.default_rng(7)
    X = rng.standard_normal((n, d)).astype(np.float32)
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    Q = rng.standard_normal((nq, d)).astype(np.float32)
    idx = FlatIndex(d, metric=metric, dtype=dtype, device=0)
    idx.set_id_base(1000)
    idx.add(X)
    D0, I0 = idx.search(Q, k)

    uid = ctypes.create_string_buffer(128)
    _capi.check(lib.mi_comm_unique_id(uid))
    comm = ctypes.c_void_p()
    _capi.check(lib.mi_comm_create(idx._ctx, uid, 1, 0, ctypes.byref(comm)))
    try:
        assert lib.mi_comm_size(comm) == 1 and lib.mi_comm_rank(comm) == 0
        q = torch.from_numpy(Q).cuda()
        Ds = torch.empty((nq, k), dtype=torch.float32, device="cuda")
        Is = torch.empty((nq, k), dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        for _ in range(2):                      # the second call reuses the communicator's buffers
            _capi.check(lib.mi_index_search_sharded_dev(idx._h, comm, nq, ctypes.c_void_p(q.data_ptr()), k,
                                                        ctypes.c_void_p(Ds.data_ptr()), ctypes.c_void_p(Is.data_ptr()),
                                                        _capi.stream_arg(q.device)))
            torch.cuda.synchronize()
            np.testing.assert_array_equal(Is.cpu().numpy(), I0)
            np.testing.assert_array_equal(Ds.cpu().numpy().view(np.uint32), D0.view(np.uint32))
        # the bare collective: a rank's bytes come back in its slot
        src = torch.arange(4096, dtype=torch.uint8, device="cuda")
        dst = torch.zeros(4096, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        _capi.check(lib.mi_comm_all_gather_dev(comm, ctypes.c_void_p(src.data_ptr()), ctypes.c_void_p(dst.data_ptr()), 4096,
                                               _capi.stream_arg(q.device)))
        torch.cuda.synchronize()
        assert torch.equal(src, dst)
    finally:
        lib.mi_comm_destroy(comm)
    # the oracle agrees (global ids)
    Do, Io = oracle_mod.search(X, Q, k, metric=oracle_mod.METRIC_IP if metric == "ip" else oracle_mod.METRIC_L2,
                               dtype=oracle_mod.DTYPE_BF16 if dtype == "bf16" else oracle_mod.DTYPE_F32, id_base=1000)
    np.testing.assert_array_equal(I0, Io)
    np.testing.assert_array_equal(D0.view(np.uint32), Do.view(np.uint32))
