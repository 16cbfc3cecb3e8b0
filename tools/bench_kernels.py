#!/usr/bin/env python
"""Micro-benchmarks of the hot kernels at the stage-1 shapes (B=32): time per launch with HIP events,
TFLOP/s vs the 2.5 PF bf16 dense MFMA peak.  Run on the GPU box:  python tools/bench_kernels.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unite_amd import ops

DEV = "cuda"
WS = torch.empty(200 << 20, dtype=torch.uint8, device=DEV)


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def gemm_case(name, M, N, K, ta=False, tb=False, out_dtype=torch.bfloat16, epi="", **kw):
    """epi: the epilogue of the step's launch of this shape -- "bias", "gelu" (bias + GELU + saved pre-activation), "qgelu", "dgelu",
    "res" (bias + stochastic-depth row scale + f32 residual)"""
    a = torch.randn((K, M) if ta else (M, K), device=DEV).to(torch.bfloat16)
    b = torch.randn((K, N) if tb else (N, K), device=DEV).to(torch.bfloat16)
    out = torch.empty(M, N, dtype=out_dtype, device=DEV)
    if epi in ("bias", "gelu", "qgelu", "res"):
        kw["bias"] = torch.randn(N, device=DEV)
    if epi == "gelu":
        kw.update(act=ops.ACT_GELU, aux_out=torch.empty(M, N, dtype=torch.bfloat16, device=DEV))
    if epi == "qgelu":
        kw.update(act=ops.ACT_QUICKGELU)
    if epi == "dgelu":
        kw.update(act=ops.ACT_DGELU, aux_in=torch.randn(M, N, device=DEV).to(torch.bfloat16))
    if epi == "res":
        kw.update(residual=torch.randn(M, N, device=DEV), row_scale=torch.ones((M + 319) // 320, device=DEV), rows_per_scale=320)
    name = f"{name} [{epi}]" if epi else name
    if ta and tb:
        kw["workspace"] = WS
    fl = 2.0 * M * N * K
    if os.environ.get("AB"):
        # tile kernels (policy 0) against the persistent kernel wherever it supports the problem (policy 2), interleaved in one process
        from unite_amd import _lib
        lib = _lib.load()
        ts = {0: [], 2: []}
        for rnd in range(3):
            for pol in (0, 2):
                lib.unite_gemm_set_policy(pol)
                ts[pol].append(timeit(lambda: ops.gemm(a, b, out, trans_a=ta, trans_b=tb, **kw), iters=10, warm=2))
        lib.unite_gemm_set_policy(-1)
        t0, t2 = min(ts[0]), min(ts[2])
        print(f"{name:34s} M={M:6d} N={N:5d} K={K:6d} tb={int(tb)}  tile {t0*1e6:8.1f} us {fl/t0/1e12:7.1f} TF/s | persistent {t2*1e6:8.1f} us {fl/t2/1e12:7.1f} TF/s  ({(t2/t0-1)*100:+5.1f} %)", flush=True)
        return t0
    t = timeit(lambda: ops.gemm(a, b, out, trans_a=ta, trans_b=tb, **kw))
    print(f"{name:34s} M={M:6d} N={N:5d} K={K:6d} ta={int(ta)} tb={int(tb)}  {t*1e6:9.1f} us  {fl/t/1e12:8.1f} TF/s  {fl/t/2.5e15*100:5.1f}% peak", flush=True)
    return t


def main():
    Ms, Mt = 10240, 50432
    print("== GEMM, student (M = 32*320)")
    gemm_case("qkv fwd", Ms, 2304, 768, epi="bias")
    gemm_case("proj fwd (f32 out)", Ms, 768, 768, out_dtype=torch.float32, epi="res")
    gemm_case("fc1 fwd", Ms, 3072, 768, epi="gelu")
    gemm_case("fc2 fwd (f32 out)", Ms, 768, 3072, out_dtype=torch.float32, epi="res")
    gemm_case("fc2 dgrad (NN)", Ms, 3072, 768, tb=True, epi="dgelu")
    gemm_case("fc1 dgrad (NN)", Ms, 768, 3072, tb=True)
    gemm_case("proj dgrad (NN)", Ms, 768, 768, tb=True)
    gemm_case("qkv dgrad (NN)", Ms, 768, 2304, tb=True)
    gemm_case("fc1 wgrad (TN, f32)", 3072, 768, Ms, ta=True, tb=True, out_dtype=torch.float32)
    gemm_case("fc2 wgrad (TN, f32)", 768, 3072, Ms, ta=True, tb=True, out_dtype=torch.float32)
    gemm_case("qkv wgrad (TN, f32)", 2304, 768, Ms, ta=True, tb=True, out_dtype=torch.float32)
    gemm_case("proj wgrad (TN, f32)", 768, 768, Ms, ta=True, tb=True, out_dtype=torch.float32)
    print("== GEMM, teacher (M = 256*197)")
    gemm_case("qkv", Mt, 2304, 768, epi="bias")
    gemm_case("out_proj (f32)", Mt, 768, 768, out_dtype=torch.float32, epi="res")
    gemm_case("c_fc", Mt, 3072, 768, epi="qgelu")
    gemm_case("c_proj (f32)", Mt, 768, 3072, out_dtype=torch.float32, epi="res")
    gemm_case("square 4096", 4096, 4096, 4096)
    gemm_case("square 8192", 8192, 8192, 8192)
    if os.environ.get("GEMM_ONLY"):
        return
    print("== attention")
    for (B, N, H, nm) in [(32, 320, 12, "student"), (256, 197, 12, "teacher")]:
        qkv = torch.randn(B * N, 3 * H * 64, device=DEV).to(torch.bfloat16)
        out = torch.empty(B * N, H * 64, dtype=torch.bfloat16, device=DEV)
        lse = torch.empty(B, H, N, device=DEV)
        t = timeit(lambda: ops.attn_fwd(qkv, out, lse, B, N, H, 0.125))
        fl = 4.0 * B * H * N * N * 64
        print(f"attn fwd {nm:8s} B={B} N={N}  {t*1e6:9.1f} us  {fl/t/1e12:7.1f} TF/s", flush=True)
        if nm == "student":
            do = torch.randn(B * N, H * 64, device=DEV).to(torch.bfloat16)
            dqkv = torch.empty_like(qkv); delta = torch.empty(B, H, N, device=DEV)
            t = timeit(lambda: ops.attn_bwd(qkv, out, do, lse, delta, dqkv, B, N, H, 0.125))
            print(f"attn bwd {nm:8s} B={B} N={N}  {t*1e6:9.1f} us  {2.5*fl/t/1e12:7.1f} TF/s (5 products)", flush=True)
    print("== teacher projection + attention in one kernel (unite_teacher_qkv_attn) vs the GEMM + attention pair")
    BT, L, H = 256, 197, 12
    D = H * 64
    hx = torch.randn(BT * L, D, device=DEV).to(torch.bfloat16)
    w_in = (torch.randn(3 * D, D, device=DEV) * D ** -0.5).to(torch.bfloat16)
    b_in = torch.randn(3 * D, device=DEV) * 0.1
    qkv = torch.empty(BT * L, 3 * D, dtype=torch.bfloat16, device=DEV)
    o = torch.empty(BT * L, D, dtype=torch.bfloat16, device=DEV)
    lse = torch.empty(BT, H, L, device=DEV)
    fl = 2.0 * BT * L * 3 * D * D + 4.0 * BT * H * L * L * 64

    def pair():
        ops.gemm(hx, w_in, qkv, bias=b_in)
        ops.attn_fwd(qkv, o, lse, BT, L, H, 0.125)
    t = timeit(pair)
    print(f"qkv GEMM + attention        {t*1e6:9.1f} us  {fl/t/1e12:7.1f} TF/s", flush=True)
    t = timeit(lambda: ops.teacher_qkv_attn(hx, w_in, b_in, o, BT, L, H, 0.125))
    print(f"fused kernel                {t*1e6:9.1f} us  {fl/t/1e12:7.1f} TF/s", flush=True)
    print("== streaming kernels")
    M, D = Ms, 768
    x = torch.randn(M, D, device=DEV); g = torch.ones(D, device=DEV); b = torch.zeros(D, device=DEV)
    y = torch.empty(M, D, dtype=torch.bfloat16, device=DEV); mean = torch.empty(M, device=DEV); rstd = torch.empty(M, device=DEV)
    t = timeit(lambda: ops.layernorm_fwd(x, g, b, 1e-6, y, mean=mean, rstd=rstd))
    print(f"layernorm fwd [{M},{D}]  {t*1e6:8.1f} us  {(M*D*6)/t/1e9:8.1f} GB/s", flush=True)
    n = 88_005_888
    p = torch.randn(n, device=DEV); gr = torch.randn(n, device=DEV); m = torch.zeros(n, device=DEV); v = torch.zeros(n, device=DEV)
    pb = torch.empty(n, dtype=torch.bfloat16, device=DEV)
    t = timeit(lambda: ops.adamw_flat(p, gr, m, v, pb, None, [1e-4], [0.05], 0.9, 0.95, 1e-8, 1))
    print(f"adamw flat n={n}  {t*1e6:8.1f} us  {n*30/t/1e9:8.1f} GB/s (30 B/elem)", flush=True)
    ws = torch.empty(ops.grad_norm_workspace(n), dtype=torch.uint8, device=DEV); nrm = torch.empty(1, device=DEV)
    t = timeit(lambda: ops.grad_norm_flat(gr, nrm, ws))
    print(f"grad norm n={n}  {t*1e6:8.1f} us  {n*4/t/1e9:8.1f} GB/s", flush=True)


if __name__ == "__main__":
    main()
