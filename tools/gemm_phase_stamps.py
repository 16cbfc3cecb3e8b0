"""Where a 256 x 256 workgroup of the bf16-epilogue GEMM spends its cycles (UNITE_GEMM_DEBUG_SKIP=7 UNITE_GEMM_KERNEL=deep256): shader-clock stamps of
thread 0 at the phase boundaries, median / mean over the launch's workgroups, for teacher c_fc plain and with bias + QuickGELU."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unite_amd import ops
M, N, K = 50432, 3072, 768
a = torch.randn(M, K, device="cuda").bfloat16(); w = torch.randn(N, K, device="cuda").bfloat16()
bias = torch.randn(N, device="cuda")
out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
nwg = ((M + 255) // 256) * ((N + 255) // 256)
ws = torch.zeros(max(nwg * 8 * 8, 1 << 20), dtype=torch.uint8, device="cuda")
for name, kw in (("plain", {}), ("bias+qgelu", dict(bias=bias, act=ops.ACT_QUICKGELU))):
    for _ in range(30):
        with ops.plan(persistent=0, sharing=1.0): ops.gemm(a, w, out, workspace=ws, **kw)
    torch.cuda.synchronize()
    st = ws[: nwg * 64].view(torch.int64).view(nwg, 8).cpu().double()
    d = lambda i, j: (st[:, j] - st[:, i])
    names = [("start -> loop end", 0, 1), ("trailing DMA wait + barrier", 1, 2), ("bias / act / pack / LDS writes", 2, 3), ("barrier", 3, 4), ("LDS reads + store issue", 4, 5), ("stores complete (vmcnt 0)", 5, 6), ("whole workgroup", 0, 6)]
    print(f"== c_fc {name}: {nwg} workgroups, shader-clock cycles (median / mean)")
    for nm, i, j in names:
        x = d(i, j); print(f"   {nm:34s} {x.median().item():9.0f} {x.mean().item():9.0f}")
