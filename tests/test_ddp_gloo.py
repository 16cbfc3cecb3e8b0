"""CPU-only, world_size 2 over gloo: the bucketed gradient reducer of unite_amd/ddp.py (the N > 1 path of bench.py)
averages contiguous slices of a flat gradient buffer as layers complete in backward order, and matches a single-process
run on the concatenated batch."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from unite_amd.ddp import GradReducer
    n = 10 * 1024
    g = torch.Generator().manual_seed(100 + rank)
    grad = torch.randn(n, generator=g)
    # 5 "layers" of 2048 elements; backward completes them from the end of the buffer
    tags = [(i, i * 2048, (i + 1) * 2048) for i in reversed(range(5))]
    red = GradReducer(grad, tags, bucket_bytes=3 * 2048 * 4)
    assert [sorted(b["tags"]) for b in red.buckets] == [[2, 3, 4], [0, 1]]
    for step in range(2):
        if step:
            grad.copy_(torch.randn(n, generator=g))
        local = grad.clone()
        for t, _, _ in tags:
            red.layer_done(t)
        red.finish()
        gathered = [torch.empty(n) for _ in range(world)]
        dist.all_gather(gathered, local)
        ref = torch.stack(gathered).mean(0)
        assert torch.allclose(grad, ref, atol=1e-6), (rank, step)
    # a layer that never reports (unused parameters) is still reduced by finish()
    grad.copy_(torch.full((n,), float(rank)))
    red.layer_done(4)
    red.finish()
    assert torch.allclose(grad, torch.full((n,), (world - 1) / 2.0))
    # meters all-reduce (SmoothedValue.synchronize_between_processes)
    from unite_amd.utils import SmoothedValue
    sv = SmoothedValue()
    sv.update(float(rank + 1), n=1)
    sv.synchronize_between_processes()
    assert sv.count == world and sv.total == sum(range(1, world + 1))
    q.put(rank)
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_grad_reducer_world2_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    for p in ps:
        p.join(100)
        assert p.exitcode == 0
    assert sorted(q.get() for _ in range(world)) == [0, 1]
