#!/usr/bin/env python
"""Stage-1 training throughput (clips/s) of the UNITE hot path on MI355X -- the metric of BASELINE.json.

A "step" is one full stage-1 iteration on a batch of B=32 synthetic 8-frame 224x224 clips per GPU
(BASELINE configs[1]): frozen CLIP-B/16 teacher forward -> attention-guided mask (mask_ratio 0.8) -> targets on the
320 visible tokens -> ViT-B/16 student forward + decoders + UMT loss -> backward -> gradient all-reduce (N>1) ->
global grad-norm -> AdamW.  Inputs are resident in HBM before the timed region; nothing is skipped inside it.

    python bench.py [--gpus N] [--steps K] [--warmup W]          (N > 1 without a launcher: bench.py starts the N ranks itself, see self_start)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
    python bench.py --config 3|4|5 [--steps K] [--warmup W]       (one GPU: BASELINE configs[2..4] -- stage 2, stage 3, ViT-L stage 1)

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline     : the dominant kernels (bf16 MFMA GEMM family + fused teacher kernel): algorithmic FLOPs / launch time,
                 measured with HIP events on the launch stream in a second pass of the same steps (events off during
                 the timed region so that `value` is undisturbed).  step_frac puts the whole step against the 2.5 PF/s
                 bf16 peak (462.2 GF per clip).  student_alone_ms / student_alone_frac: the student step (forward, decoders,
                 loss, backward, grad-norm, AdamW: 179.7 GF per clip, the number north_star's 40 % target is about) MEASURED
                 alone on the GPU in its own loop on pre-computed teacher outputs, planned for a GPU of its own;
                 teacher_ms / teacher_frac: the frozen teacher alone (282.5 GF per clip).  step_minus_teacher_ms is the old
                 subtraction (what the student ADDS to an overlapped step), kept under its own name.
                 traffic: HBM bytes per launch from the PMC counters (profiles/, source quoted); traffic_algorithmic_bytes:
                 the same launches' operands + outputs + residual / aux matrices, each once, per launch.
  host_enqueue_ms_per_step : wall time the host needs to enqueue one step (no sync inside a step).
  cpu_baseline : the CPU oracle (oracle/umt_oracle.py, fp32 torch) timed on this host's cores on a bounded sample.
"""
import argparse
import contextlib
import ctypes as C
import json
import os
import sys
import time


ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)



def self_start(argv):
    """`python bench.py --gpus N` with N > 1 and no launcher around it (RANK unset): start the N ranks ourselves, BEFORE this process has made
    any HIP call, as fresh children -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P
    bench.py <the same arguments>` (the reference's launch line, stage1.sh:15-17) -- relay rank 0's one JSON line on stdout (anything else the
    children write to stdout goes to stderr) and exit with the launcher's return code.  Returns None when there is nothing to start."""
    import socket
    import subprocess
    n = 1
    for i, tok in enumerate(argv):
        if tok == "--gpus" and i + 1 < len(argv):
            n = int(argv[i + 1])
        elif tok.startswith("--gpus="):
            n = int(tok.split("=", 1)[1])
    if n <= 1 or "RANK" in os.environ:
        return None
    with socket.socket() as sk:                      # a free rendezvous port on the loopback interface
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    from unite_amd import hwqueues                 # (host logic only: importing the package makes no HIP call)
    masked = hwqueues.mask_entries(env)
    gpus = masked if masked is not None else hwqueues.kfd_gpu_nodes()
    if gpus and n > gpus:                          # a rehearsal: more ranks than GPUs, so ranks share devices -- tell the hardware-queue rule
        env.setdefault("UNITE_RANKS_SHARE_GPU", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    print(f"[bench] --gpus {n} without a launcher: starting {' '.join(cmd)}", file=sys.stderr, flush=True)
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    record = None
    for line in child.stdout:
        try:
            rec = json.loads(line)
            if isinstance(rec, dict) and "metric" in rec:
                record = line.strip()
                continue
        except ValueError:
            pass
        sys.stderr.write(line)
    rc = child.wait()
    if record is not None:
        print(record, flush=True)
    return rc if rc != 0 or record is not None else 1


if __name__ == "__main__":
    _rc = self_start(sys.argv[1:])
    if _rc is not None:
        raise SystemExit(_rc)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

# Before the first HIP call (`import unite_amd` applies the same rule, but this file touches the GPU before it imports the package): the
# step's four busy streams plus RCCL's do not fit HIP's default four hardware queues -- with a process group alive the teacher's stream shares
# the student's queue and the step takes 24.2 instead of 20.4 ms.  Decided from the environment alone (unite_amd/hwqueues.py: no torch.cuda call).
from unite_amd.hwqueues import apply as _apply_hw_queues  # noqa: E402
_HWQ = _apply_hw_queues()

PEAK_BF16 = 2.5e15            # dense bf16 MFMA peak, MI355X_MICROARCH.md "Chip-level parameters"
GF_STUDENT, GF_TEACHER = 179.7e9, 282.5e9     # algorithmic FLOPs per clip, BASELINE.md section 2


def cpu_baseline(seconds_budget=30.0):
    """oracle stage-1 step (teacher fwd, student fwd+bwd, grad-norm, AdamW), fp32, the box's host cores: BASELINE configs[0] (B = 4, the
    reference's own CPU-runnable case) for a few steps, then ONE step at the benchmark's own batch (B = 32, configs[1]) if the budget allows
    -- `value` is the B = 32 rate when that step ran, the B = 4 rate otherwise; `sample` says which and quotes both."""
    from oracle import umt_oracle as O
    from oracle.filler import fill_state_dict, make_importance, make_videos
    from tests.shapes import student_shapes, teacher_shapes
    # the GPU box gives one GPU's job a 16-CPU share whatever os.cpu_count() says: oversubscribing it stalls for minutes
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    threads = max(1, min(avail, int(os.environ.get("UNITE_CPU_THREADS", 16))))
    torch.set_num_threads(threads)
    scfg, tcfg = O.StudentCfg(), O.TeacherCfg()
    ssd = fill_state_dict(student_shapes(scfg), 12)
    tsd = fill_state_dict(teacher_shapes(tcfg), 11)
    B = 4
    data = {}

    def batch(Bn):
        if Bn not in data:
            data[Bn] = (make_videos(Bn, 8, 224, 224, 13), O.mask_from_importance(make_importance(Bn * 8, 196, 14), 40, Bn))
        return data[Bn]
    m = {k: torch.zeros_like(v) for k, v in ssd.items()}
    v = {k: torch.zeros_like(v) for k, v in ssd.items()}

    def step(i, Bn=4):
        vid, mask = batch(Bn)
        leaf = {k: p.requires_grad_(True) for k, p in ssd.items()}
        loss, *_ = O.stage1_loss(leaf, tsd, vid, mask, scfg, tcfg)
        loss.backward()
        O.grad_norm([p.grad for p in leaf.values()])
        with torch.no_grad():
            for k, p in leaf.items():
                p.requires_grad_(False)
                O.adamw_step(p, p.grad, m[k], v[k], i, 1.5e-4 * Bn / 256, 0.9, 0.95, 1e-8, 0.05 if p.ndim > 1 else 0.0)
                p.grad = None

    t0 = time.time()
    step(1)                                   # warm-up
    warm = time.time() - t0
    print(f"[bench] cpu baseline warm-up step: {warm:.1f} s on {threads} threads", file=sys.stderr, flush=True)
    n = 2
    t0 = time.time()
    for i in range(n):
        step(2 + i)
    dt = (time.time() - t0) / n
    sample = f"{n} timed stage-1 steps at B={B} (8fx224^2, ViT-B/16 + CLIP-B/16, fp32 torch CPU oracle) after 1 warm-up: {dt:.2f} s/step = {B / dt:.2f} clips/s"
    value = B / dt
    spent = warm + n * dt
    if spent + 8.5 * dt <= seconds_budget + 5.0:          # one step at the benchmark's batch: ~8x the B = 4 step
        batch(32)
        t0 = time.time()
        step(2 + n, 32)
        dt32 = time.time() - t0
        value = 32 / dt32
        sample = f"1 timed stage-1 step at B=32 (the benchmark's batch): {dt32:.1f} s = {value:.2f} clips/s; before it " + sample
    return {"value": round(value, 3), "unit": "clips/s", "cores": threads, "kind": "port", "sample": sample}


def other_config(a):
    """--config 3 | 4 | 5: the per-GPU step of BASELINE configs[2..4] on one GPU, in this file's record format (no roofline pass: the
    kernels are those of the headline step, profiled there; mfma_frac = clips/s x algorithmic GFLOP per clip / 2.5 PF/s)"""
    import importlib.util
    if a.gpus != 1:
        raise SystemExit("--config 3 / 4 / 5 time the per-GPU step on ONE GPU")
    sys.stdout.flush()
    record_fd = os.dup(1)
    os.dup2(2, 1)                                   # stdout carries the record only
    spec = importlib.util.spec_from_file_location("bench_configs", os.path.join(ROOT, "tools", "bench_configs.py"))
    bc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bc)
    name = {3: "stage2", 4: "stage3", 5: "vitl"}[a.config]
    steps, warmup = (a.steps, a.warmup)
    r = bc.run(name, None, "clip_l14", steps, warmup)
    unit = "clip pairs/s" if a.config == 4 else "clips/s"
    out = {"metric": {3: "stage-2 train clips/sec (ViT-B/16, 16fx224^2, 8 classes)", 4: "stage-3 train (src, tgt) clip pairs/sec (ViT-B/16 + CLIP-L/14 mask teacher, 8fx224^2)",
                      5: "stage-1 train clips/sec (ViT-L/16 + CLIP-L/14, 16fx224^2)"}[a.config],
           "value": r["clips_per_s"], "unit": unit, "n_gpus": 1, "steps": steps, "warmup": warmup, "ms_per_step": r["ms_per_step"],
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
           "config": {"workload": r["workload"] + f" (BASELINE configs[{a.config - 1}], per-GPU shapes on one GPU)", "global_batch": r["batch"], "parallelism": "dp1"},
           "final_loss": r["loss"], "final_grad_norm": r["grad_norm"], "ranks_seen": 1, "hw_queues": _HWQ,
           "roofline": {"bound": "mfma", "achieved": round(r["mfma_frac"] * PEAK_BF16 / 1e12, 1), "peak": PEAK_BF16 / 1e12, "unit": "TFLOP/s",
                        "frac": r["mfma_frac"], "traffic": None, "gflop_per_unit": r["gflop_per_clip"],
                        "note": "whole step: units/s x algorithmic GFLOP per unit (SURVEY 8d) over the bf16 MFMA peak; per-kernel figures: the headline config's roofline"},
           "hbm_peak_GiB": r["hbm_peak_GiB"]}
    os.write(record_fd, (json.dumps(out) + "\n").encode())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=32, help="clips per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--serial", action="store_true", help="one HIP stream for the whole run (no side-stream weight gradients, no multi-stream "
                                                          "teacher): per-kernel durations are then undisturbed, as in the roofline pass")
    ap.add_argument("--graph", type=int, default=0, help="1: capture the whole step in a HIP graph and replay it (single rank; measured SLOWER than "
                                                        "eager launches on ROCm 7.0: DESIGN.md section 5); 0 (default): eager launches")
    ap.add_argument("--ahead", type=int, default=int(os.environ.get("UNITE_TEACHER_AHEAD", "1")),
                    help="1: the frozen teacher runs one batch ahead of the student on its own stream (engine_stage1.TeacherAhead, what "
                         "train_one_epoch does by default); 0: teacher and student of a step strictly one after the other")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL, the measured path) | gloo (rehearsal of the N>1 flow on one GPU)")
    ap.add_argument("--config", type=int, default=2, choices=[2, 3, 4, 5],
                    help="BASELINE.json configs[] by 1-based position: 2 (default) the headline stage-1 step, B = 32; 3 stage-2 fine-tune step "
                         "(ViT-B/16, 16 f, B = 16); 4 stage-3 collaborative step (ViT-B/16 + CLIP-L/14 mask teacher, 16 src + 16 tgt); "
                         "5 stage-1 ViT-L/16 + CLIP-L/14 (16 f, B = 8).  3-5: one GPU, the per-GPU shapes of those configs (tools/bench_configs.py)")
    a = ap.parse_args()
    if a.config != 2:
        return other_config(a)
    # stdout carries exactly ONE line, the JSON record: whatever libraries print there (RCCL's version banner at communicator creation, gloo's
    # connection messages) is sent to stderr by pointing file descriptor 1 at it for the whole run; the record is written to the saved descriptor
    sys.stdout.flush()
    record_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N bench.py --gpus N ...")
    ndev = torch.cuda.device_count()
    local = local % max(ndev, 1)                                   # rehearsal: several ranks may share the one GPU of a test box
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # UNITE_DDP_FORCE_COLLECTIVES=1 at N = 1: a one-rank RCCL group and the data-parallel wrapper anyway -- the bucket all-reduces, their side
    # stream and the joins run as they do at N > 1 (a rehearsal on a one-GPU box; values are unchanged by a one-rank mean)
    rehearse = world == 1 and os.environ.get("UNITE_DDP_FORCE_COLLECTIVES", "0") == "1"
    if rehearse:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
    if world > 1 or rehearse:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(a.backend, rank=rank, world_size=world)

    import unite_amd
    from unite_amd import _lib, ops
    if (world > 1 or rehearse) and a.backend == "nccl" and os.environ.get("UNITE_COMM_NATIVE", "0") == "1":
        # the native communicator is created NOW, before the step's streams exist, as torch's process group creates its own at init_process_group.
        # (Round 4 traced why the one-rank rehearsal through it takes 27.5 instead of 20.5 ms per step: with it, HIP puts the teacher's stream and
        # the weight-gradient stream on ONE hardware queue -- 189 kernels per step on queue 8, queues 5 and 6 unused -- whatever GPU_MAX_HW_QUEUES
        # says; creating the communicator early does not change that assignment.  The native path stays opt-in and is not the measured one.)
        from unite_amd.ddp import native_comm
        native_comm()
    from unite_amd.ddp import DistributedDataParallel
    from unite_amd.engine_stage1 import StepState, TeacherAhead, stage1_step, student_phase, teacher_phase
    from unite_amd.optim_factory import create_optimizer
    from unite_amd.utils import NativeScalerWithGradNormCount, cosine_scheduler
    from types import SimpleNamespace

    torch.manual_seed(0 + rank)                                   # run_stage1.py:613
    B, T = a.batch, 8
    student = unite_amd.create_model(
        "adaptation_umt_base_patch16_224", pretrained=False, drop_path_rate=0.1, drop_block_rate=None, use_learnable_pos_emb=False,
        use_checkpoint=False, checkpoint_num=0, clip_decoder_embed_dim=768, clip_output_dim=512, clip_norm_type='l2', num_frames=T,
        tubelet_size=1, clip_return_layers=[6, 7, 8, 9, 10, 11], clip_student_return_interval=1, use_cls_token=False).to(dev).train()
    teacher = unite_amd.clip.clip_b16(pretrained=False, return_attn=True, clip_return_layers=[6, 7, 8, 9, 10, 11]).to(dev)
    model = DistributedDataParallel(student) if (world > 1 or rehearse) else student
    total_batch = B * world
    args = SimpleNamespace(opt="adamw", weight_decay=0.05, lr=1.5e-4 * total_batch / 256, opt_eps=1e-8, opt_betas=[0.9, 0.95])
    n_iter = a.warmup + 2 * a.steps + 4
    with contextlib.redirect_stdout(sys.stderr):                  # the reference's factories print their settings; stdout carries the JSON line only
        opt = create_optimizer(args, student, skip_list=student.no_weight_decay())
        lr_sched = cosine_scheduler(args.lr, 1e-5, 1, n_iter, warmup_epochs=0)
    scaler = NativeScalerWithGradNormCount()
    reducer = getattr(model, "reducer", None)
    videos = torch.randn(B, 3, T, 224, 224, device=dev)           # synthetic, ImageNet-normalised-like; resident in HBM
    state = StepState()
    state.seed = 1000 * rank
    if a.serial:
        student.runtime().runner.wgrad_stream = False
        teacher.runtime().two_streams = False
        state.overlap_targets = False
    it = [0]

    use_graph = a.graph == 1
    graphed = None
    if use_graph:
        if world > 1:
            raise SystemExit("--graph 1 is single-rank (the bucket all-reduces go through torch.distributed: eager)")
        from unite_amd.graph_step import GraphedStage1Step
        graphed = GraphedStage1Step(model, teacher, opt, scaler, tuple(videos.shape), 0.8, clip_grad=None, state=state)

    ahead = TeacherAhead(teacher, state, dev, 0.8, 'attention') if (a.ahead == 1 and not a.serial and not use_graph) else None
    # the planner weight every GEMM launch of a step carries (unite_gemm_args.plan_sharing): the shared-GPU setting while the teacher runs
    # ahead; --serial profiles the kernels of that default step one at a time and keeps its setting
    share = [float(os.environ.get("UNITE_GEMM_SHARING", str(TeacherAhead.DEFAULT_SHARING))) if (a.ahead == 1 and not use_graph) else 0.0]
    touts = []
    torch.cuda.synchronize()               # `videos` is complete: the teacher's stream need not wait for the student's at every launch

    # experiment: extra HBM traffic on a stream of its own (UNITE_HBM_LOAD = GB copied per step) to see how memory-bound the step is
    hbm_gb = float(os.environ.get("UNITE_HBM_LOAD", "0"))
    if hbm_gb > 0:
        hbm_src = torch.empty(int(hbm_gb * 2 ** 30), dtype=torch.uint8, device=dev)
        hbm_dst = torch.empty_like(hbm_src)
        hbm_stream = torch.cuda.Stream(device=dev)

    def step():
        if hbm_gb > 0:
            with torch.cuda.stream(hbm_stream):
                hbm_dst.copy_(hbm_src)
        i = it[0]
        for g in opt.param_groups:
            g["lr"] = lr_sched[min(i, len(lr_sched) - 1)] * g["lr_scale"]
        it[0] += 1
        if graphed is not None and not graph_off[0]:
            return graphed(videos)
        with ops.plan(sharing=share[0]):
            if ahead is not None and not graph_off[0]:
                # every step enqueues ONE teacher phase (for the step after it) and ONE student step (on the outputs of the teacher phase
                # enqueued a step earlier): K timed steps = K teacher forwards + K student steps, as in train_one_epoch
                if not touts:
                    touts.append(ahead.launch(videos, inputs_ready=False))
                cur = touts.pop()
                touts.append(ahead.launch(videos, inputs_ready=False))
                loss = student_phase(model, videos, cur, B, 'mixed')
            else:
                loss = stage1_step(model, teacher, videos, B, 0.8, 'attention', None, 'mixed', state)
            opt.zero_grad()
            gn = scaler(loss, opt, clip_grad=None, parameters=None, reducer=reducer)
        return loss, gn
    graph_off = [False]

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # diagnostic (UNITE_CLOCK_PROBE=1): a short kernel at the end of every timed step stamps {shader clock counter, 100-MHz reference counter}
    # per XCD (unite_clock_stamp) -> "shader_clock_mhz": the clock the chip held, averaged over each step.  (A resident probe wave on its own
    # stream -- unite_clock_probe -- is no good here: beside it the multi-stream step takes 42 instead of 20 ms.)
    probe = None
    if os.environ.get("UNITE_CLOCK_PROBE", "0") == "1":
        probe = torch.zeros(a.steps + 1, 8, 2, dtype=torch.int64, device=dev)

    sprio = os.environ.get("UNITE_STUDENT_PRIO")
    sctx = torch.cuda.stream(torch.cuda.Stream(device=dev, priority=int(sprio))) if sprio is not None else contextlib.nullcontext()
    with sctx:
        for _ in range(a.warmup):
            loss, gn = step()
        fence()
        if probe is not None:
            _lib.check(_lib.load().unite_clock_stamp(probe[0].data_ptr(), torch.cuda.current_stream().cuda_stream), "unite_clock_stamp")
        t0 = time.perf_counter()
        for k_step in range(a.steps):
            loss, gn = step()
            if probe is not None:
                _lib.check(_lib.load().unite_clock_stamp(probe[k_step + 1].data_ptr(), torch.cuda.current_stream().cuda_stream), "unite_clock_stamp")
        t_enq = time.perf_counter() - t0          # the host has enqueued every launch of the timed steps (nothing synchronises inside a step)
        fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    clock = None
    if probe is not None:
        sm = probe.cpu().numpy().astype("float64")                         # [step boundary][XCD][{shader, reference}]
        mhz = []
        for x in range(8):
            c, r = sm[:, x, 0], sm[:, x, 1]
            ok = (r[1:] > 0) & (r[:-1] > 0) & (r[1:] > r[:-1])              # both stamps landed on this XCD
            mhz += list(((c[1:] - c[:-1]) / (r[1:] - r[:-1]) * 100.0)[ok])
        if mhz:
            mhz.sort()
            clock = {"mean": round(sum(mhz) / len(mhz), 1), "p10": round(mhz[len(mhz) // 10], 1), "median": round(mhz[len(mhz) // 2], 1),
                     "p90": round(mhz[len(mhz) * 9 // 10], 1), "samples": len(mhz), "averaged_over": "one step, per XCD"}
    loss_v, gn_v = float(loss.item()), float(gn.item())
    # random-init student against a random-init teacher: the cosine loss starts at ~2 and can only move inside [0, 4] (run_stage1.py:431)
    if not (loss_v == loss_v and 0.0 <= loss_v <= 4.0 and gn_v == gn_v and gn_v > 0.0):
        raise SystemExit(f"bench: implausible final loss {loss_v} / grad-norm {gn_v}: the timed steps did not train")
    ms_step = dt / a.steps * 1e3
    if rank == 0:
        print(f"[bench] timed region: {a.steps} steps in {dt:.3f} s ({dt / a.steps * 1e3:.2f} ms/step), loss {loss_v:.4f}", file=sys.stderr, flush=True)
    clips_s = total_batch * a.steps / dt

    def set_concurrency(on: bool):
        """side-stream weight gradients (student) and the teacher's frame ranges on their own streams; off = every kernel alone on the GPU"""
        student.runtime().runner.wgrad_stream = on
        teacher.runtime().two_streams = on
        state.overlap_targets = on

    # the step by phase: the frozen teacher ALONE on the GPU in a steady-state loop (configured as in the timed region), and the student step
    # ALONE on the GPU (forward, decoders, loss, backward with its side-stream weight gradients, grad-norm, AdamW) on the outputs of ONE
    # pre-computed teacher phase, every launch planned for a GPU of its own (sharing 0) -- a measurement, not step - teacher
    teacher_ms, student_alone_ms = None, None
    if not a.no_roofline:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n_t = min(a.steps, 20)
        trt = teacher.runtime()
        keep_streams = trt.n_streams
        if ahead is not None:           # the teacher as the timed region runs it: one stream, tile GEMM kernels, shared-GPU planner weight
            trt.n_streams = ahead.n_streams
        with (ahead.hints() if ahead is not None else ops.plan(sharing=share[0])):
            for _ in range(3):
                teacher.forward_attention(videos)
            torch.cuda.synchronize()
            ev0.record()
            for _ in range(n_t):
                teacher.forward_attention(videos)
            ev1.record()
            torch.cuda.synchronize()
        teacher_ms = ev0.elapsed_time(ev1) / n_t
        trt.n_streams = keep_streams
        if world == 1 and graphed is None:
            with ops.plan(sharing=0.0):
                tout_fixed = teacher_phase(teacher, videos, 0.8, 'attention', None, state, slot=7, inline_targets=True)
                torch.cuda.synchronize()

                def student_step():
                    l_ = student_phase(model, videos, tout_fixed, B, 'mixed')
                    opt.zero_grad()
                    scaler(l_, opt, clip_grad=None, parameters=None, reducer=reducer)
                for _ in range(3):
                    student_step()
                torch.cuda.synchronize()
                ev0.record()
                for _ in range(n_t):
                    student_step()
                ev1.record()
                torch.cuda.synchronize()
            student_alone_ms = ev0.elapsed_time(ev1) / n_t

    roof = None
    if not a.no_roofline:
        graph_off[0] = True             # the per-launch events of the roofline pass need eager launches
        set_concurrency(False)          # kernel durations of the roofline pass must not include time shared with other kernels
    if not a.no_roofline and rank == 0:
        lib = _lib.load()
        n_prof = min(a.steps, 5)
        lib.unite_prof_enable(1, 400 * n_prof)
        torch.cuda.synchronize()
        for _ in range(n_prof):
            step()
        torch.cuda.synchronize()
        ms, cnt, fl, by = C.c_double(), C.c_int64(), C.c_double(), C.c_double()
        lib.unite_prof_summary(C.byref(ms), C.byref(cnt), C.byref(fl), C.byref(by))
        lib.unite_prof_enable(0, 0)
        print(f"[bench] profiled pass: {cnt.value} MFMA-kernel launches, {ms.value:.1f} ms", file=sys.stderr, flush=True)
        ach = fl.value / (ms.value * 1e-3) / 1e12 if ms.value > 0 else 0.0
        # the same pass with every launch planned for a GPU of its own (sharing weight 0): what the kernels reach when they are both
        # configured and measured stand-alone
        ach_alone = None
        if share[0] > 0:
            keep_share, share[0] = share[0], 0.0
            step()
            lib.unite_prof_enable(1, 400 * n_prof)
            torch.cuda.synchronize()
            for _ in range(n_prof):
                step()
            torch.cuda.synchronize()
            ms2, cnt2, fl2 = C.c_double(), C.c_int64(), C.c_double()
            lib.unite_prof_summary(C.byref(ms2), C.byref(cnt2), C.byref(fl2), None)
            lib.unite_prof_enable(0, 0)
            share[0] = keep_share
            ach_alone = fl2.value / (ms2.value * 1e-3) / 1e12 if ms2.value > 0 else None
        # HBM bytes per MFMA-kernel launch: NOT measured by this run (PMC counters need rocprofv3: tools/final_prof.sh collects the
        # FETCH_SIZE / WRITE_SIZE passes of `bench.py --serial` and tools/pmc_traffic.py reduces them); quoted with its source, or null
        traffic, traffic_source = None, None
        for name in ("r04_gemm_traffic.json", "r03_gemm_traffic.json", "r02_gemm_traffic.json"):
            tp = os.path.join(ROOT, "profiles", name)
            if os.path.exists(tp):
                tj = json.load(open(tp))
                traffic = round(tj["traffic_bytes_per_launch"])
                traffic_source = f"profiles/{name} (rocprofv3 --pmc passes of bench.py --serial at commit {tj.get('commit', '?')}; not measured in this run)"
                break
        roof = {"bound": "mfma", "kernel": "bf16 MFMA GEMM kernels: gemm_deep_kernel / gemm_wide_kernel family (all layouts/epilogues) + teacher_qkv_attn_kernel (projection + attention FLOPs)", "achieved": round(ach, 1),
                "peak": PEAK_BF16 / 1e12, "unit": "TFLOP/s", "frac": round(ach * 1e12 / PEAK_BF16, 4), "traffic": traffic,
                "traffic_source": traffic_source,
                "traffic_algorithmic_bytes": round(by.value / max(cnt.value, 1)),
                "frac_planned_alone": None if ach_alone is None else round(ach_alone * 1e12 / PEAK_BF16, 4),
                "launches_per_step": cnt.value // n_prof, "gemm_ms_per_step": round(ms.value / n_prof, 3),
                "gemm_gflop_per_step": round(fl.value / n_prof / 1e9, 1),
                "step_mfma_frac_full": round(clips_s / world * (GF_STUDENT + GF_TEACHER) / PEAK_BF16, 4),
                "step_frac": round(clips_s / world * (GF_STUDENT + GF_TEACHER) / PEAK_BF16, 4),
                "teacher_ms": round(teacher_ms, 3),
                "teacher_frac": round(B * GF_TEACHER / (teacher_ms * 1e-3) / PEAK_BF16, 4),
                "student_alone_ms": None if student_alone_ms is None else round(student_alone_ms, 3),
                "student_alone_frac": None if student_alone_ms is None else round(B * GF_STUDENT / (student_alone_ms * 1e-3) / PEAK_BF16, 4),
                "step_minus_teacher_ms": round(ms_step - teacher_ms, 3),
                "note": "HIP events on the launch stream around every launch of these kernels in a second pass of the same steps, run on ONE stream "
                        "(the timed region overlaps the teacher of the next batch, the student and its weight-gradient GEMMs on separate streams, "
                        "which would charge each launch for time it shares with other kernels); same kernels and planner setting as the timed "
                        "region, same numbers as `bench.py --serial` under rocprofv3.  The planner sizes launches for a SHARED GPU "
                        "(plan_sharing 0.9 in every unite_gemm_args: larger tiles, fewer split-K slices), so these stand-alone durations are longer "
                        "than with UNITE_GEMM_SHARING=0 (frac_planned_alone: the same pass planned and measured stand-alone) while the step is shorter: step_frac is the number that counts the whole step"}
    elif world > 1:
        # keep ranks in lock-step with rank 0's profiled passes (every step all-reduces)
        n_prof = min(a.steps, 5)
        for _ in range(n_prof + ((1 + n_prof) if share[0] > 0 else 0)):
            step()
        torch.cuda.synchronize()

    if not a.no_roofline:
        set_concurrency(not a.serial)
        graph_off[0] = False

    # what one step costs the HOST: the same step at B = 2, where the GPU work (~3 ms) is shorter than the enqueue, so wall time per step is
    # host time per step (t_enq of the timed region mostly measures back-pressure: the launch queue is full while the GPU is ~20 ms behind)
    host_ms = None
    if world == 1 and not a.no_roofline:
        v2, st2 = videos[:2].contiguous(), StepState()
        for k in range(13):
            if k == 3:
                torch.cuda.synchronize()
                th = time.perf_counter()
            with ops.plan(sharing=share[0]):
                l2 = stage1_step(model, teacher, v2, 2, 0.8, 'attention', None, 'mixed', st2)
            opt.zero_grad()
            scaler(l2, opt, clip_grad=None, parameters=None, reducer=reducer)
        torch.cuda.synchronize()
        host_ms = (time.perf_counter() - th) / 10 * 1e3
    if rank == 0:
        out = {"metric": "stage-1 train clips/sec (ViT-B/16, 8fx224^2)", "value": round(clips_s, 2), "unit": "clips/s", "n_gpus": world,
               "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_step, 3), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
               "config": {"workload": "stage1 UMT pretrain, ViT-B/16 student + CLIP-B/16 teacher, synthetic 8fx224^2 clips, "
                                      f"B={B}/GPU, mask_ratio=0.8, bf16 MFMA + fp32 accumulate/master (BASELINE configs[1])",
                          "global_batch": total_batch, "parallelism": f"dp{world}", "drop_path": 0.1, "optimizer": "AdamW(0.9,0.95) wd 0.05",
                          "launch": "hip_graph" if graphed is not None else "eager",
                          "schedule": "teacher one batch ahead of the student (own stream)" if ahead is not None else "teacher then student",
                          "teacher_residual_rows": {False: "f32", True: "bf16", "f16": "f16"}[teacher.runtime().res16]},
               "ranks_seen": dist.get_world_size() if dist.is_initialized() else 1, "hw_queues": _HWQ,
               "final_loss": round(loss_v, 5), "final_grad_norm": round(gn_v, 5),
               "host_enqueue_ms_per_step": round(t_enq / a.steps * 1e3, 3),
               "host_ms_per_step_unblocked": None if host_ms is None else round(host_ms, 3)}
        if clock is not None:
            out["shader_clock_mhz"] = clock
        if roof is not None:
            out["roofline"] = roof
        if not a.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline()
        os.write(record_fd, (json.dumps(out) + "\n").encode())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
