"""How fast does the vendor library (hipBLASLt behind torch.matmul) run the step's GEMM shapes, without any epilogue?  A yardstick for the
hand-written kernels only: nothing in unite_amd/ calls it.  Usage: python tools/blaslt_ref.py"""
import torch

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
        for _ in range(5):
            f()
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                f()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
        print(f"{name:18s} M={M:6d} N={N:5d} K={K:5d}  {best:8.1f} us  {2.0 * M * N * K / best / 1e6:8.1f} TF/s", flush=True)


if __name__ == "__main__":
    main()
