"""How fast does the vendor library (hipBLASLt behind torch.matmul) run the step's GEMM shapes, without any epilogue?  A yardstick for the
hand-written kernels only: nothing in unite_amd/ calls it.  Next to it the same product -- plain bf16 output (f32 for the weight-gradient
layout), no bias, no activation, no residual -- through unite_gemm_bf16 with its default planner.  Usage: python tools/blaslt_ref.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unite_amd import ops  # noqa: E402

SHAPES = [("teacher c_fc", 50432, 3072, 768), ("teacher c_proj", 50432, 768, 3072), ("teacher out_proj", 50432, 768, 768), ("teacher qkv", 50432, 2304, 768),
          ("student qkv", 10240, 2304, 768), ("student proj", 10240, 768, 768), ("student fc1", 10240, 3072, 768), ("student fc2", 10240, 768, 3072),
          ("wgrad fc1 (TN)", 3072, 768, 10240), ("wgrad proj (TN)", 768, 768, 10240), ("square 4096", 4096, 4096, 4096), ("square 8192", 8192, 8192, 8192)]


def main():
    dev = torch.device("cuda:0")
    for name, M, N, K in SHAPES:
        tn = "TN" in name
        a = torch.randn((K, M) if tn else (M, K), device=dev).bfloat16()
        w = torch.randn((K, N) if tn else (N, K), device=dev).bfloat16()
        f = (lambda: torch.matmul(a.t(), w)) if tn else (lambda: torch.matmul(a, w.t()))
        out = torch.empty(M, N, dtype=torch.float32 if tn else torch.bfloat16, device=dev)
        ws = torch.empty(220 << 20, dtype=torch.uint8, device=dev) if tn else None
        h = (lambda: ops.gemm(a, w, out, trans_a=True, trans_b=True, workspace=ws)) if tn else (lambda: ops.gemm(a, w, out))

        def best_of(fn):
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            best = 1e9
            for rep in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
            return best
        tb, th = best_of(f), best_of(h)
        fl = 2.0 * M * N * K / 1e6
        print(f"{name:18s} M={M:6d} N={N:5d} K={K:5d}  hipBLASLt {tb:8.1f} us {fl / tb:7.1f} TF/s | this build {th:8.1f} us {fl / th:7.1f} TF/s | ratio {th / tb:5.2f}", flush=True)


if __name__ == "__main__":
    main()
