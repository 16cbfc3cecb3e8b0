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
res = torch.randn(M, 768, device="cuda")
a2 = torch.randn(M, 3072, device="cuda").bfloat16(); w2 = torch.randn(768, 3072, device="cuda").bfloat16(); out2 = torch.empty(M, 768, device="cuda"); b2 = torch.randn(768, device="cuda")
CASES = (("c_fc plain (bf16 epilogue form 1)", a, w, out, {}), ("c_fc bias + QuickGELU (form 1)", a, w, out, dict(bias=bias, act=ops.ACT_QUICKGELU)),
         ("c_proj f32 + bias + residual (form 0: stamps 3-6 = pass-0 staging writes, barrier, pass-0 chunks, rest)", a2, w2, out2, dict(bias=b2, residual=res)))
for name, a, w, out, kw in CASES:
    nwg = ((out.shape[0] + 255) // 256) * ((out.shape[1] + 255) // 256)
    ws.zero_()
    for _ in range(30):
        with ops.plan(persistent=0, sharing=1.0): ops.gemm(a, w, out, workspace=ws, **kw)
    torch.cuda.synchronize()
    st = ws[: nwg * 64].view(torch.int64).view(nwg, 8).cpu().double()
    d = lambda i, j: (st[:, j] - st[:, i])
    names = [("start -> loop end", 0, 1), ("trailing DMA wait + barrier", 1, 2), ("bias / act / pack / LDS writes", 2, 3), ("barrier", 3, 4), ("LDS reads + store issue", 4, 5), ("stores complete (vmcnt 0)", 5, 6), ("whole workgroup", 0, 6)]
    print(f"== {name}: {nwg} workgroups, shader-clock cycles (median / mean)")
    for nm, i, j in names:
        x = d(i, j); print(f"   {nm:34s} {x.median().item():9.0f} {x.mean().item():9.0f}")
