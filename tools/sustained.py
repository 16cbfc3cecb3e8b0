"""SUSTAINED timing of single GEMM shapes: each arm runs back to back for `seconds` (default 0.3 s) with the shader-clock probe beside it.
Under a sustained MFMA load the chip lowers its clock until it fits its power limit, so the time per launch of a long loop measures the
ENERGY a launch costs (time = energy / power limit), where a short interleaved burst (tools/gemm_sched_ab.py) measures its cycles.  The
training step is a sustained load (profiles/r04_clock_notes.txt), so this is the figure that predicts it.
Usage: python tools/sustained.py [seconds] "name:M,N,K[,f32][,bias][,qgelu][,gelu (+ saved pre-activation)][,dgelu][,dsave (GELU + saved derivative)][,mulaux][,res][,res16 (f16 residual rows in, f16 rows out)][,nt][,tn][,vendor][,s0][,e0|,e1][,pp][,w8|,w9 (planner sharing weight 0.8 / 0.9, as in the step)]" ...   (UNITE_GEMM_DEBUG_SKIP etc. apply)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unite_amd import ops  # noqa: E402
from tools.clock_under_load import probe_while  # noqa: E402


def main():
    argv = sys.argv[1:]
    seconds = 0.3
    if argv and argv[0].replace(".", "").isdigit():
        seconds = float(argv.pop(0))
    dev = torch.device("cuda:0")
    tag = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("UNITE_GEMM") or k.startswith("UNITE_PP"))
    print(f"# sustained loops of {seconds} s each; {tag or 'default switches'}", flush=True)
    for spec in argv:
        name, _, rest = spec.rpartition(":")
        f = rest.split(",")
        M, N, K = int(f[0]), int(f[1]), int(f[2])
        o = set(f[3:])
        tn, nt = "tn" in o, "nt" in o
        a = torch.randn((K, M) if tn else (M, K), device=dev).bfloat16()
        w = torch.randn((K, N) if (tn or nt) else (N, K), device=dev).bfloat16()
        out = torch.empty(M, N, dtype=torch.float32 if (tn or "f32" in o) else torch.bfloat16, device=dev)
        bias = torch.randn(N, device=dev) if "bias" in o else None
        res = torch.randn(M, N, device=dev) if "res" in o else torch.randn(M, N, device=dev).half() if "res16" in o else None
        if "res16" in o:
            out = torch.empty(M, N, dtype=torch.float16, device=dev)
        act = (ops.ACT_QUICKGELU if "qgelu" in o else ops.ACT_GELU if "gelu" in o else ops.ACT_DGELU if "dgelu" in o else
               ops.ACT_GELU_DSAVE if "dsave" in o else ops.ACT_MULAUX if "mulaux" in o else ops.ACT_NONE)
        aux_out = torch.empty(M, N, dtype=torch.bfloat16, device=dev) if ("gelu" in o or "dsave" in o) else None
        aux_in = torch.randn(M, N, device=dev).bfloat16() if ("dgelu" in o or "mulaux" in o) else None
        ws = torch.empty(220 << 20, dtype=torch.uint8, device=dev) if tn else None
        if "vendor" in o:
            fn = (lambda: torch.matmul(a.t(), w)) if tn else (lambda: torch.matmul(a, w)) if nt else (lambda: torch.matmul(a, w.t()))
        else:
            sched = 0 if "s0" in o else 1
            pp = 1 if "pp" in o else 0
            epi = 1 if "e1" in o else 0 if "e0" in o else None
            share = 0.8 if "w8" in o else 0.9 if "w9" in o else None

            def fn():
                with ops.plan(persistent=2 if pp else 0, sched=sched, epi=epi, sharing=share):
                    ops.gemm(a, w, out, trans_a=tn, trans_b=tn or nt, bias=bias, act=act, residual=res, workspace=ws, aux_out=aux_out, aux_in=aux_in)
        c, us = probe_while(fn, seconds)
        print(f"{name or rest:34s} {rest:40s} {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TF/s  clock {c['mean']:6.0f} MHz", flush=True)


if __name__ == "__main__":
    main()
