"""-m gpu: the data-parallel path end to end on one MI355X -- two FRESH child processes (subprocess, never a re-exec of the
pytest process) share the GPU and exchange gradients over gloo: stage-1 student under DistributedDataParallel on half of a
batch each equals the single-process step on the whole batch (SURVEY.md section 4's DDP property, through layer_done and the
side-stream completion events); the stage-2 classifier wraps too and accumulates over update_freq = 2 micro-batches without
reducing in between.  (RCCL itself needs one GPU per rank: the driver's multi-GPU bench is its first run.)"""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def rel_l2(a, b):
    return ((a.float() - b.float()).norm() / (b.float().norm() + 1e-12)).item()


@pytest.mark.timeout(600)
def test_two_ranks_one_gpu_equal_single_process(tmp_path):
    world, port = 2, str(_free_port())
    outs = [str(tmp_path / f"r{r}.pt") for r in range(world)]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    ps = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_ddp_gpu_worker.py"), str(r), str(world), port, outs[r]],
                           env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    logs = []
    for p in ps:
        try:
            o, _ = p.communicate(timeout=500)
        except subprocess.TimeoutExpired:
            p.kill()
            o, _ = p.communicate()
        logs.append(o.decode(errors="replace")[-3000:])
    assert all(p.returncode == 0 for p in ps), "\n----\n".join(logs)
    r0, r1 = (torch.load(o, weights_only=True) for o in outs)
    # stage 1: both ranks hold the same reduced gradient = the full-batch gradient; mean of the two half-batch losses = full loss
    assert torch.equal(r0["s1.grad"], r1["s1.grad"])
    assert rel_l2(r0["s1.grad"], r0["s1.full_grad"]) <= 2e-3
    assert abs(0.5 * (r0["s1.loss"] + r1["s1.loss"]) - r0["s1.full_loss"]) <= 2e-5 * abs(r0["s1.full_loss"])
    assert r0["s1.launched"] == r1["s1.launched"] > 0
    # ... and with the teacher one batch ahead (TeacherAhead + the planner's shared-GPU setting: split-K plans may differ, values do not)
    assert torch.equal(r0["s1.ahead_grad"], r1["s1.ahead_grad"])
    assert rel_l2(r0["s1.ahead_grad"], r0["s1.grad"]) <= 1e-5
    assert abs(r0["s1.ahead_loss"] - r0["s1.loss"]) <= 2e-6 * abs(r0["s1.loss"])
    # AdamW of each bucket right behind its all-reduce (opt-in) == the single launch after the backward, bit for bit, over three steps
    assert torch.equal(r0["s1.opt_params"], r1["s1.opt_params"]) and torch.equal(r0["s1.bucket_params"], r1["s1.bucket_params"])
    assert torch.equal(r0["s1.bucket_params"], r0["s1.opt_params"])
    assert r0["s1.bucket_gn"] == r0["s1.opt_gn"] and not torch.equal(r0["s1.opt_params"], r0["s1.init_params"])
    # stage 2: one collective per bucket for the whole accumulation step, gradient = full-batch mean-CE gradient
    assert torch.equal(r0["s2.grad"], r1["s2.grad"])
    assert rel_l2(r0["s2.grad"], r0["s2.full_grad"]) <= 2e-3
    assert r0["s2.launched"] == r1["s2.launched"] > 0
    assert abs(r0["s2.gn"] - r0["s2.full_grad"].norm().item()) <= 2e-3 * r0["s2.full_grad"].norm().item()


@pytest.mark.timeout(600)
def test_rccl_path_one_rank(tmp_path):
    """The reducer's RCCL path on the one GPU of a test box (tests/_ddp_rccl_worker.py): backend 'nccl', one rank, collectives forced.
    Three optimisation steps under the wrapper -- with the AdamW of each bucket behind its all-reduce, and with the single launch -- leave
    bit for bit the parameters, gradients and gradient norms of the unwrapped run, and the collectives were really issued."""
    out = tmp_path / "rccl.pt"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_ddp_rccl_worker.py"), str(_free_port()), str(out)], cwd=ROOT,
                       capture_output=True, text=True, timeout=500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    res = torch.load(out, weights_only=True)
    assert res["backend"] == "nccl"
    p0, g0, n0, _ = res["plain"]
    for key in ("ddp", "ddp_bucket"):
        p, g, n, launched = res[key]
        assert launched >= 3 * 4, launched                      # several buckets per step, three steps
        assert torch.equal(p, p0) and torch.equal(g, g0) and n == n0, key
