"""Host cost of one stage-1 step: the step at B = 2 (GPU work ~3 ms, far below the host's enqueue time, so wall time per step = host time
per step) under cProfile.  Usage: python tools/host_profile.py [n_steps]"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, '.')
os.environ.setdefault("B", "2")
import runpy
ns = runpy.run_path(os.path.join(os.path.dirname(__file__), "host_overhead.py"))
step = ns["step"]
import torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    step()
torch.cuda.synchronize()
print(f"B={os.environ['B']}: {1e3 * (time.perf_counter() - t0) / n:.2f} ms/step wall (host-bound)")
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(35)
