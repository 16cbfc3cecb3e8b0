"""Which shader clock does the chip hold under each kind of load?  unite_clock_probe (one wave, s_memtime against the 100-MHz reference counter,
a sample every 250 us) beside loops of single kernels: idle, LayerNorm (HBM-bound), the teacher's c_fc product through the tile kernel under both
main-loop schedules and through the vendor library (yardstick), a deep square.  Prints mean / p10 / median / p90 MHz and the loop's rate.
Usage: python tools/clock_under_load.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unite_amd import _lib, ops  # noqa: E402


def probe_while(fn, seconds=0.25, interval_us=250):
    dev = torch.device("cuda:0")
    n = int(seconds * 1e6 / interval_us)
    buf = torch.zeros(n, 2, dtype=torch.int64, device=dev)
    side = torch.cuda.Stream(device=dev)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    _lib.check(_lib.load().unite_clock_probe(buf.data_ptr(), n, interval_us, side.cuda_stream), "unite_clock_probe")
    t0 = time.perf_counter()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    k = 0
    while time.perf_counter() - t0 < seconds * 1.05:      # keep the launch queue fed for the whole probe (the loop blocks on back-pressure)
        for _ in range(10):
            fn()
        k += 10
        if k % 200 == 0:
            torch.cuda.current_stream().synchronize()
    e1.record()
    torch.cuda.synchronize()
    sm = buf.cpu().numpy().astype("float64")
    mhz = (sm[1:, 0] - sm[:-1, 0]) / (sm[1:, 1] - sm[:-1, 1]) * 100.0
    mhz = mhz[len(mhz) // 10:]                              # skip the ramp at the start
    mhz.sort()
    return dict(mean=mhz.mean(), p10=mhz[len(mhz) // 10], median=mhz[len(mhz) // 2], p90=mhz[len(mhz) * 9 // 10]), e0.elapsed_time(e1) * 1e3 / k


def main():
    dev = torch.device("cuda:0")
    M, N, K = 50432, 3072, 768
    a = torch.randn(M, K, device=dev).bfloat16()
    w = torch.randn(N, K, device=dev).bfloat16()
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    x = torch.randn(M, K, device=dev)
    y = torch.empty(M, K, dtype=torch.bfloat16, device=dev)
    g, b = torch.ones(K, device=dev), torch.zeros(K, device=dev)
    sq = torch.randn(8192, 8192, device=dev).bfloat16()
    sq_out = torch.empty(8192, 8192, dtype=torch.bfloat16, device=dev)

    def gemm(s, A=a, W=w, O=out):
        def f():
            with ops.plan(persistent=0, sched=s):
                ops.gemm(A, W, O)
        return f
    cases = [("idle (nothing but the probe)", lambda: None),
             ("layernorm_fwd 50432 x 768 (HBM-bound)", lambda: ops.layernorm_fwd(x, g, b, 1e-5, y)),
             ("teacher c_fc, tile kernel, sched 0", gemm(0)), ("teacher c_fc, tile kernel, sched 1", gemm(1)),
             ("teacher c_fc, vendor library (yardstick)", lambda: torch.matmul(a, w.t())),
             ("square 8192, tile kernel, sched 0", gemm(0, sq, sq, sq_out)), ("square 8192, tile kernel, sched 1", gemm(1, sq, sq, sq_out)),
             ("square 8192, vendor library (yardstick)", lambda: torch.matmul(sq, sq.t()))]
    for name, fn in cases:
        if name.startswith("idle"):
            time.sleep(0.2)
        c, us = probe_while(fn)
        print(f"{name:44s} clock mean {c['mean']:7.1f}  p10 {c['p10']:7.1f}  median {c['median']:7.1f}  p90 {c['p90']:7.1f} MHz | {us:8.1f} us per launch", flush=True)


if __name__ == "__main__":
    main()
