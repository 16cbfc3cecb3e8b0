#!/usr/bin/env python
"""Weight gradients of one ViT-B block (tokens = 10240): four single launches (split-K planner) vs one grouped launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unite_amd import ops
Mtok = int(sys.argv[1]) if len(sys.argv) > 1 else 10240
D = 768
shapes = [(3072, D), (D, 3072), (3 * D, D), (D, D)]          # (out rows, out cols): dW = dY^T X
probs = []
ws = torch.empty(16 * 3072 * 1024 * 4, dtype=torch.uint8, device="cuda")
for (r, c) in shapes:
    dy = torch.randn(Mtok, r, device="cuda").to(torch.bfloat16)
    x = torch.randn(Mtok, c, device="cuda").to(torch.bfloat16)
    out = torch.zeros(r, c, device="cuda")
    probs.append((dy, x, out, dict(trans_a=True, trans_b=True)))

def single():
    for dy, x, out, kw in probs:
        ops.gemm(dy, x, out, workspace=ws, **kw)

def grouped():
    ops.gemm_grouped(probs)

def timeit(f):
    for _ in range(3):
        f()
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            f()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 100)
    return sorted(ts)[1]
fl = sum(2.0 * Mtok * r * c for r, c in shapes)
for name, f in (("single x4", single), ("grouped", grouped), ("single x4", single), ("grouped", grouped)):
    t = timeit(f)
    print(f"{name:10s} {t:8.1f} us  {fl / t / 1e6:7.1f} TF/s")
for i, (dy, x, out, kw) in enumerate(probs):
    t = timeit(lambda: ops.gemm(dy, x, out, workspace=ws, **kw))
    t2 = timeit(lambda: ops.gemm(dy, x, out, **kw))
    print(f"problem {i} {tuple(out.shape)}: planner {t:7.1f} us   no-split {t2:7.1f} us")
