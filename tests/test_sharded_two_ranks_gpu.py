"""GPU: the sharded search and the data-parallel embedding with TWO ranks on one device (tests/two_rank_child.py under
torch.distributed.run, gloo rendezvous on 127.0.0.1 -- RCCL refuses two ranks on one GPU; gloo carries the CUDA tensors).
The product path is what runs: HIP engine per shard, packed block, one all_gather_into_tensor, HIP merge in place.
Sharded answer == single-index answer, ids and score bits, on both ranks."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_two_ranks_on_one_gpu_equal_the_single_index():
    here = os.path.dirname(os.path.abspath(__file__))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(here, "two_rank_child.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=200)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    for rank in (0, 1):
        assert r.stdout.count(f"rank {rank} OK search") == 3, r.stdout[-2000:]
        assert f"rank {rank} OK embed_sharded" in r.stdout, r.stdout[-2000:]
