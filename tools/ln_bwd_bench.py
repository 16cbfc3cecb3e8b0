import sys, torch
sys.path.insert(0, '.')
from unite_amd import ops
M, D = 10240, 768
dy = torch.randn(M, D, device="cuda").to(torch.bfloat16)
x = torch.randn(M, D, device="cuda"); res = torch.randn(M, D, device="cuda")
mean = x.mean(1).contiguous(); rstd = (x.var(1, unbiased=False) + 1e-6).rsqrt().contiguous()
g = torch.randn(D, device="cuda"); dx = torch.empty_like(x); dxb = torch.empty(M, D, dtype=torch.bfloat16, device="cuda")
dg, db, ds = (torch.zeros(D, device="cuda") for _ in range(3))
ws = torch.empty(ops.layernorm_bwd_workspace(M, D), dtype=torch.uint8, device="cuda")
def run(): ops.layernorm_bwd(dy, x, mean, rstd, g, dx_residual=res, dx_out=dx, dx_bf16=dxb, dgamma=dg, dbeta=db, dxsum=ds, workspace=ws)
for _ in range(5): run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): run()
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) * 20
print(f"layernorm_bwd [{M},{D}]: {t:.1f} us ({126e6 * (M/10240) / t / 1e6:.2f} TB/s incl. second stage)")
