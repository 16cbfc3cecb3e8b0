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
    # after_bucket (the hook FusedAdamW.step_range hangs on): once per bucket, with the bucket's range, when its slice already holds the mean
    grad.copy_(torch.randn(n, generator=g))
    gathered = [torch.empty(n) for _ in range(world)]
    dist.all_gather(gathered, grad.clone())
    ref = torch.stack(gathered).mean(0)
    seen = []
    red.after_bucket = lambda lo, hi: seen.append((lo, hi, bool(torch.allclose(grad[lo:hi], ref[lo:hi], atol=1e-6))))
    for t, _, _ in tags:
        red.layer_done(t)
    red.finish()
    red.after_bucket = None
    assert seen == [(2 * 2048, 5 * 2048, True), (0, 2 * 2048, True)], seen
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


def test_bucket_layout_keeps_the_last_bucket_small():
    """ViT-B-like layout (a decoder block, 12 layers of 7.08 M parameters, a small patch-embed layer, in backward completion order):
    64-MiB buckets over whole layers, and the layers that finish last split off into a tail bucket of at most 32 MiB -- the one
    all-reduce that cannot overlap any backward work."""
    from unite_amd.ddp import GradReducer
    L, dec, pe = 7_087_872, 2_400_000, 600_000
    sizes = [("clip_decoder", dec)] + [(i, L) for i in reversed(range(12))] + [("patch_embed", pe)]
    total = sum(n for _, n in sizes)
    tags, hi = [], total
    for t, n in sizes:                       # forward order in memory = reverse of completion order: completion walks down
        tags.append((t, hi - n, hi))
        hi -= n
    red = GradReducer(torch.zeros(1), tags, bucket_bytes=64 << 20)       # (layout only: world size 1, nothing is launched)
    layout = [[p[0] for p in b["parts"]] for b in red.buckets]
    assert layout[-1] == [0, "patch_embed"]
    assert all((b["hi"] - b["lo"]) * 4 <= (96 << 20) for b in red.buckets) and (red.buckets[-1]["hi"] - red.buckets[-1]["lo"]) * 4 <= (32 << 20)
    covered = sorted((b["lo"], b["hi"]) for b in red.buckets)             # buckets tile the buffer exactly once
    assert covered[0][0] == 0 and covered[-1][1] == total and all(a[1] == b[0] for a, b in zip(covered, covered[1:]))
    assert sorted([t for b in red.buckets for t in b["tags"]], key=str) == sorted([t for t, _ in sizes], key=str)
    # no tail split when asked not to
    red2 = GradReducer(torch.zeros(1), tags, bucket_bytes=64 << 20, tail_bytes=0)
    assert [[p[0] for p in b["parts"]] for b in red2.buckets][-2:] == [[2, 1, 0], ["patch_embed"]]


def _worker_accum(rank, world, port, q):
    """update_freq = 2 (engine_for_finetuning.py:83-84) and stage 3's unused parameters (clip_decoder.* gets no gradient):
    micro-batch 1 runs under no_sync (adds into the buffer, nothing is reduced, nothing stays pending), micro-batch 2 reduces the
    accumulated sum once per bucket; the result equals the mean over ranks of the accumulated gradient."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from unite_amd.ddp import GradReducer
    from unite_amd.utils import NativeScalerWithGradNormCount
    n = 6 * 1024
    tags = [("dec", 5 * 1024, 6 * 1024)] + [(i, i * 1024, (i + 1) * 1024) for i in reversed(range(5))]
    grad = torch.zeros(n)
    red = GradReducer(grad, tags, bucket_bytes=2 * 1024 * 4, tail_bytes=1024 * 4)
    assert [sorted(map(str, b["tags"])) for b in red.buckets] == [["4", "dec"], ["2", "3"], ["1"], ["0"]]
    g = torch.Generator().manual_seed(7 + rank)
    micro = [torch.randn(n, generator=g) for _ in range(2)]
    for m in micro:
        m[5 * 1024:] = 0.0                    # the unused layer ("dec") never receives a gradient

    class FakeLoss:                           # what the scaler sees: .backward() runs the hand-scheduled backward
        def __init__(self, m):
            self.m = m

        def backward(self, create_graph=False):
            for t, lo, hi in tags[1:]:        # "dec" never reports
                grad[lo:hi] += self.m[lo:hi]
                red.layer_done(t)

    class FakeOpt:
        _flat = None

    scaler = NativeScalerWithGradNormCount()
    for step in range(2):
        grad.zero_()
        before = red.launched
        assert scaler(FakeLoss(micro[0]), FakeOpt(), update_grad=False, reducer=red) is None
        assert red.launched == before and all(len(p) == len(b["tags"]) for p, b in zip(red._pending, red.buckets))
        FakeLoss(micro[1]).backward()
        assert red.launched == before + 3      # three buckets completed by their layers
        red.finish()                           # ... and the bucket holding the unused layer by finish()
        assert red.launched == before + 4
        local = micro[0] + micro[1]
        gathered = [torch.empty(n) for _ in range(world)]
        dist.all_gather(gathered, local)
        assert torch.allclose(grad, torch.stack(gathered).mean(0), atol=1e-6), (rank, step)
    q.put(rank)
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_grad_reducer_world4_accumulation_and_unused_layers_gloo():
    world, port = 4, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker_accum, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    for p in ps:
        p.join(100)
        assert p.exitcode == 0
    assert sorted(q.get() for _ in range(world)) == [0, 1, 2, 3]
