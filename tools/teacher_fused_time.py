#!/usr/bin/env python
"""Teacher block front half: fused projection + attention kernel vs the qkv GEMM + attention pair (CLIP-B/16 shapes, 256 frames)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unite_amd import ops
BT, L, H = int(os.environ.get("BT", 256)), 197, int(os.environ.get("H", 12))
D = H * 64
h = torch.randn(BT * L, D, device="cuda").to(torch.bfloat16)
w = (torch.randn(3 * D, D, device="cuda") * D ** -0.5).to(torch.bfloat16)
b = torch.randn(3 * D, device="cuda") * 0.1
qkv = torch.empty(BT * L, 3 * D, dtype=torch.bfloat16, device="cuda")
o = torch.empty(BT * L, D, dtype=torch.bfloat16, device="cuda")
lse = torch.empty(BT, H, L, device="cuda")
def unfused():
    ops.gemm(h, w, qkv, bias=b)
    ops.attn_fwd(qkv, o, lse, BT, L, H, 0.125)
def fused():
    ops.teacher_qkv_attn(h, w, b, o, BT, L, H, 0.125)
def t(f):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 50
fl = 2.0 * BT * L * 3 * D * D + 4.0 * BT * H * L * L * 64
for name, f in (("unfused", unfused), ("fused", fused), ("unfused", unfused), ("fused", fused)):
    us = t(f)
    print(f"{name:8s} {us:8.1f} us   {fl / us / 1e6:7.1f} TF/s")
