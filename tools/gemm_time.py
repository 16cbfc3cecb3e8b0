#!/usr/bin/env python
"""Time GEMM shapes under the library's env switches (one process per setting):
   python tools/gemm_time.py "M,N,K[,f32][,bias][,qgelu|gelu|dgelu][,res][,nt]" ...   -> us per launch (median of 3 x 20 launches)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unite_amd import ops
tag = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("UNITE_GEMM"))
for spec in sys.argv[1:]:
    f = spec.split(",")
    M, N, K = int(f[0]), int(f[1]), int(f[2])
    opts = set(f[3:])
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    b = torch.randn(N, K, device="cuda").to(torch.bfloat16)
    out = torch.empty(M, N, dtype=torch.float32 if "f32" in opts else torch.bfloat16, device="cuda")
    bias = torch.randn(N, device="cuda") if "bias" in opts else None
    res = torch.randn(M, N, device="cuda") if "res" in opts else None
    act = ops.ACT_QUICKGELU if "qgelu" in opts else ops.ACT_GELU if "gelu" in opts else ops.ACT_DGELU if "dgelu" in opts else ops.ACT_NONE
    aux = torch.randn(M, N, device="cuda").to(torch.bfloat16) if ("gelu" in opts or "dgelu" in opts) else None
    if "nt" in opts:                                   # B stored [K, N] (input-gradient layout)
        b = b.t().contiguous()
    run = lambda: ops.gemm(a, b, out, bias=bias, act=act, residual=res, trans_b="nt" in opts,
                           aux_out=aux if act == ops.ACT_GELU else None, aux_in=aux if act == ops.ACT_DGELU else None)
    for _ in range(5):
        run()
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / 20)
    t = sorted(ts)[1]
    print(f"[{tag}] {spec:32s} {t:8.1f} us  {2.0 * M * N * K / t / 1e6:7.1f} TF/s", flush=True)
