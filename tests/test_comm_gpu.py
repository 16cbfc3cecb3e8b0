"""-m gpu: the RCCL communicator behind include/unite_comm.h (libunite_comm.so) on the one GPU of a test box: a one-rank communicator
(RCCL refuses two ranks on one device) -- rendezvous id, init, in-place all-reduce (sum / mean; f32 and bf16) and broadcast are identities
there, asynchronous on the given stream, and the communicator can be destroyed and created again.  Multi-rank runs are the driver's."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_one_rank_communicator_roundtrip():
    from unite_amd import _lib
    lib = _lib.load_comm()
    assert lib.unite_comm_world() == 0 and lib.unite_comm_rank() == -1
    for _ in range(2):                                   # twice: destroy really releases the communicator
        buf = ctypes.create_string_buffer(_lib.COMM_ID_BYTES)
        assert lib.unite_comm_unique_id(buf, _lib.COMM_ID_BYTES) == 0
        assert any(buf.raw)
        torch.cuda.set_device(0)
        assert lib.unite_comm_init(0, 1, buf.raw, _lib.COMM_ID_BYTES) == 0
        assert lib.unite_comm_init(0, 1, buf.raw, _lib.COMM_ID_BYTES) < 0          # one communicator per process
        assert lib.unite_comm_world() == 1 and lib.unite_comm_rank() == 0
        side = torch.cuda.Stream()
        g = torch.randn(1 << 20, device="cuda")
        ref = g.clone()
        h = torch.randn(4096, device="cuda").bfloat16()
        href = h.clone()
        side.wait_stream(torch.cuda.current_stream())
        s = side.cuda_stream
        assert lib.unite_comm_allreduce_bucket(g.data_ptr(), g.numel(), 0, 1, s) == 0     # mean over one rank
        assert lib.unite_comm_allreduce_bucket(g.data_ptr(), g.numel(), 0, 0, s) == 0     # sum
        assert lib.unite_comm_allreduce_bucket(h.data_ptr(), h.numel(), 1, 1, s) == 0
        assert lib.unite_comm_broadcast(g.data_ptr(), g.numel() * 4, 0, s) == 0
        assert lib.unite_comm_broadcast(g.data_ptr(), 16, 1, s) < 0                       # root outside the communicator
        assert lib.unite_comm_allreduce_bucket(g.data_ptr(), g.numel(), 7, 1, s) < 0      # unknown dtype
        side.synchronize()
        assert torch.equal(g, ref) and torch.equal(h, href)
        assert lib.unite_comm_destroy() == 0
        assert lib.unite_comm_world() == 0
    assert lib.unite_comm_allreduce_bucket(g.data_ptr(), g.numel(), 0, 1, 0) < 0          # no communicator
