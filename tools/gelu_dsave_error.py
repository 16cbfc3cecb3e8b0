"""Error of the fc2 input gradient dz = (dy @ W2) * GELU'(z) against fp32 on the same bf16-rounded operands, for the two ways the backward
gets GELU'(z): recomputed from the saved bf16 z (UNITE_ACT_GELU + UNITE_ACT_DGELU) or saved by the forward as a 16-bit fixed-point number (UNITE_ACT_GELU_DSAVE +
UNITE_ACT_MULAUX, the default since round 4).  Prints the relative L2 error and the relative bias (mean error / mean |dz|) of both.
Usage: python tools/gelu_dsave_error.py [M]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unite_amd import ops  # noqa: E402


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 10240
    D, H = 768, 3072
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn(M, D, device=dev, generator=g).bfloat16()
    w1 = (torch.randn(H, D, device=dev, generator=g) * D ** -0.5).bfloat16()
    b1 = torch.randn(H, device=dev, generator=g) * 0.1
    w2 = (torch.randn(D, H, device=dev, generator=g) * H ** -0.5).bfloat16()
    dy = torch.randn(M, D, device=dev, generator=g).bfloat16()
    z32 = (x.float() @ w1.float().t() + b1).requires_grad_(True)
    torch.nn.functional.gelu(z32).sum().backward()
    ref = (dy.float() @ w2.float()) * z32.grad
    a = torch.empty(M, H, dtype=torch.bfloat16, device=dev)
    aux = torch.empty(M, H, dtype=torch.bfloat16, device=dev)
    dz = torch.empty(M, H, dtype=torch.bfloat16, device=dev)
    f32 = torch.empty(M, H, dtype=torch.float32, device=dev)
    for name, fa, ba in (("saved z, GELU' recomputed", ops.ACT_GELU, ops.ACT_DGELU), ("saved GELU'(z)", ops.ACT_GELU_DSAVE, ops.ACT_MULAUX)):
        ops.gemm(x, w1, a, bias=b1, act=fa, aux_out=aux)
        ops.gemm(dy, w2, dz, trans_b=True, act=ba, aux_in=aux)
        ops.gemm(dy, w2, f32, trans_b=True, act=ba, aux_in=aux)          # the same product before the bf16 rounding of the output
        for tag, out in (("bf16 out", dz.float()), ("f32 out ", f32)):
            e = out - ref
            print(f"{name:28s} {tag}: rel L2 {(e.norm() / ref.norm()).item():.3e}   bias {(e.mean() / ref.abs().mean()).item():+.2e}", flush=True)


if __name__ == "__main__":
    main()
