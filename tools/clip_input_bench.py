import sys, torch
sys.path.insert(0, '.')
from unite_amd.data import ClipToTensor
x = torch.randint(0, 256, (32, 8, 224, 224, 3), dtype=torch.uint8, device="cuda")
f = (torch.rand(32, device="cuda") < 0.5).to(torch.uint8)
c = ClipToTensor(reuse_output=True)
for _ in range(3): c(x, f)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): c(x, f)
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 20 * 1e3
n = x.numel()
print(f"clip_u8_to_f32 B=32 8x224x224: {t:.1f} us, {n * 5 / t / 1e6:.2f} TB/s (1 B in + 4 B out per element)")
