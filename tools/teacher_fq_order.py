#!/usr/bin/env python
"""The fused teacher projection + attention kernel alone, sustained (0.3 s back to back), under the unit order the environment selects
(UNITE_TEACHER_FQ_ORDER, read once per process); prints a checksum of the output so that two orders can be compared bit for bit."""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unite_amd import ops

BT, L, H = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (256, 197, 12)
D = H * 64
g = torch.Generator(device="cuda").manual_seed(0)
h = torch.randn(BT * L, D, device="cuda", generator=g).to(torch.bfloat16)
w = (torch.randn(3 * D, D, device="cuda", generator=g) * 0.03).to(torch.bfloat16)
b = torch.randn(3 * D, device="cuda", generator=g) * 0.1
o = torch.empty(BT * L, D, dtype=torch.bfloat16, device="cuda")
fn = lambda: ops.teacher_qkv_attn(h, w, b, o, BT, L, H, 0.125)
for _ in range(20): fn()
torch.cuda.synchronize()
n, t0 = 0, time.time()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
while time.time() - t0 < 0.3:
    for _ in range(50): fn()
    n += 50
    torch.cuda.synchronize()
e1.record(); torch.cuda.synchronize()
sha = hashlib.sha256(o.cpu().view(torch.uint8).numpy().tobytes()).hexdigest()[:16]
print(f"[UNITE_TEACHER_FQ_ORDER={os.environ.get('UNITE_TEACHER_FQ_ORDER', 'default')}] BT={BT} L={L} H={H}: {e0.elapsed_time(e1) * 1e3 / n:.1f} us per launch  sha256 {sha}", flush=True)
