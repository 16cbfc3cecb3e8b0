"""A/B of the tile kernels' main-loop schedules in ONE process (plan_flags bits 2 / 3 of unite_gemm_args; ops.plan(sched=...)):
sched 0 = fragment reads at the head of each phase, sched 1 = software-pipelined reads between the MFMAs.  For every shape: the two
products compared bit for bit, then interleaved rounds of 20 launches each, median and best; next to them the vendor library's plain
product (yardstick only, tools/blaslt_ref.py).  UNITE_GEMM_KERNEL=deep256|deep128 pins the tile size for the whole process.
Usage: python tools/gemm_sched_ab.py [rounds]"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unite_amd import ops  # noqa: E402

# name, M, N, K, layout (nn: x W^T, nt: B stored [K, N] = input gradient, tn: both k-strided = weight gradient), epilogue options
SHAPES = [("teacher c_fc", 50432, 3072, 768, "nn", ""), ("teacher c_fc +bias+qgelu", 50432, 3072, 768, "nn", "bias,qgelu"),
          ("teacher c_proj", 50432, 768, 3072, "nn", ""), ("teacher c_proj f32+res", 50432, 768, 3072, "nn", "f32,bias,res"),
          ("teacher out_proj", 50432, 768, 768, "nn", ""), ("teacher qkv", 50432, 2304, 768, "nn", ""),
          ("student qkv", 10240, 2304, 768, "nn", ""), ("student proj", 10240, 768, 768, "nn", ""),
          ("student fc1", 10240, 3072, 768, "nn", ""), ("student fc2", 10240, 768, 3072, "nn", ""),
          ("dgrad fc1 (NT)", 10240, 768, 3072, "nt", ""), ("dgrad fc2 (NT)", 10240, 3072, 768, "nt", ""), ("dgrad qkv (NT)", 10240, 768, 2304, "nt", ""),
          ("wgrad fc1 (TN)", 3072, 768, 10240, "tn", ""), ("wgrad proj (TN)", 768, 768, 10240, "tn", ""),
          ("square 4096", 4096, 4096, 4096, "nn", ""), ("square 8192", 8192, 8192, 8192, "nn", "")]


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    dev = torch.device("cuda:0")
    pin = os.environ.get("UNITE_GEMM_KERNEL", "planner")
    print(f"# tile kernel: {pin}; persistent kernel off for both arms; {rounds} interleaved rounds x 20 launches, random normal operands", flush=True)
    for name, M, N, K, lay, opt in SHAPES:
        opts = set(opt.split(",")) if opt else set()
        tn, nt = lay == "tn", lay == "nt"
        a = torch.randn((K, M) if tn else (M, K), device=dev).bfloat16()
        w = torch.randn((K, N) if (tn or nt) else (N, K), device=dev).bfloat16()
        f32 = tn or "f32" in opts
        outs = [torch.empty(M, N, dtype=torch.float32 if f32 else torch.bfloat16, device=dev) for _ in range(2)]
        bias = torch.randn(N, device=dev) if "bias" in opts else None
        res = torch.randn(M, N, device=dev) if "res" in opts else None
        act = ops.ACT_QUICKGELU if "qgelu" in opts else ops.ACT_NONE
        ws = torch.empty(220 << 20, dtype=torch.uint8, device=dev) if tn else None

        def run(sched, out):
            with ops.plan(persistent=0, sched=sched):
                ops.gemm(a, w, out, trans_a=tn, trans_b=tn or nt, bias=bias, act=act, residual=res, workspace=ws)
        ref = (lambda: torch.matmul(a.t(), w)) if tn else (lambda: torch.matmul(a, w)) if nt else (lambda: torch.matmul(a, w.t()))
        run(0, outs[0])
        run(1, outs[1])
        torch.cuda.synchronize()
        same = torch.equal(outs[0], outs[1])
        if not opts:      # plain product: also against the vendor library's result (bf16 rounding of an f32 sum: a few ulp apart at most)
            r = ref().float()
            err = ((outs[1].float() - r).abs().max() / r.abs().max()).item()
        else:
            err = float("nan")

        def timed(fn):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / 20 * 1e3
        arms = [lambda: run(0, outs[0]), lambda: run(1, outs[1]), ref]
        for fn in arms:
            for _ in range(3):
                fn()
        ts = [[], [], []]
        for _ in range(rounds):
            for i, fn in enumerate(arms):
                ts[i].append(timed(fn))
        med = [statistics.median(t) for t in ts]
        best = [min(t) for t in ts]
        fl = 2.0 * M * N * K / 1e6
        print(f"{name:26s} M={M:6d} N={N:5d} K={K:5d} | sched0 {med[0]:7.1f} us ({best[0]:7.1f}) {fl / med[0]:7.1f} TF/s | sched1 {med[1]:7.1f} us ({best[1]:7.1f}) "
              f"{fl / med[1]:7.1f} TF/s | s1/s0 {med[1] / med[0]:5.3f} | vendor {med[2]:7.1f} us  s1/vendor {med[1] / med[2]:5.2f} | bit-equal {same} rel-err {err:.1e}", flush=True)


if __name__ == "__main__":
    main()
