#!/usr/bin/env python
"""Weight-gradient GEMM shapes of the stage-1 student (dW = dY^T X, K = 10 240 tokens) through unite_gemm_bf16, with / without the fused bias
row sums and the split-K workspace, planned alone (sharing 0) and for a shared GPU (0.8):  python tools/wgrad_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unite_amd import ops
K = int(os.environ.get("TOKENS", 10240))
ws = torch.empty(32768 + 16 * 4096 * 4 + 16 * 3072 * 1024 * 4, dtype=torch.uint8, device="cuda")
for name, M, N in (("qkv", 2304, 768), ("proj", 768, 768), ("fc1", 3072, 768), ("fc2", 768, 3072), ("dec", 512, 768), ("patch", 768, 768)):
    a = torch.randn(K, M, device="cuda").to(torch.bfloat16)
    b = torch.randn(K, N, device="cuda").to(torch.bfloat16)
    out = torch.empty(M, N, device="cuda")
    rs = torch.empty(M, device="cuda")
    line = f"{name:6s} {M:5d}x{N:5d}x{K}:"
    sweep = os.environ.get("UNITE_GEMM_FORCE_PLAN") is not None
    for share in ((0.0,) if sweep else (0.0, 0.8)):
        for kw, tag in (((dict(workspace=ws, rowsum_out=rs), "forced"),) if sweep else ((dict(), "plain"), (dict(workspace=ws), "splitk"), (dict(workspace=ws, rowsum_out=rs), "splitk+rowsum"))):
            with ops.plan(sharing=share):
                run = lambda: ops.gemm(a, b, out, trans_a=True, trans_b=True, **kw)
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
            line += f"  w={share} {tag} {sorted(ts)[1]:6.1f}us"
    print(line, flush=True)
