"""What the type of the frozen teacher's residual stream costs in accuracy: small CLIP towers (width 128, 3 blocks, 4 x 4 patch grid, 2 frames) with
seeded random weights and clips, the HIP teacher with an f32 / bf16 / f16 stream (UNITE_TEACHER_RES16) against the fp32 CPU oracle
(oracle/umt_oracle.py teacher_forward, pinned on the reference's own vectors).  Over `seeds` models: error of the CLS attention row (the mask
weights, clip.py:95-96,183) and cosine of the L2-normalised target features.  One fixture is 16 values -- too few to tell the streams apart
(any perturbation re-randomises the bf16 roundings downstream), hence the statistics.
Usage: python tools/teacher_stream_error.py [seeds]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.teacher_stream_stats import collect  # noqa: E402   (the checker lives with the tests: tests/teacher_stream_stats.py)


def main():
    seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    stats = collect(seeds)
    print(f"# {seeds} models x 4 frames x 16 attention values (mean value {1 / 16:.4f}); features: {stats['f32'][2].numel() // seeds} rows per model")
    for name, (ab, rl, cs) in stats.items():
        print(f"{name:5s} attention abs err rms {ab.pow(2).mean().sqrt():.2e}  max {ab.max():.2e} | rel err rms {rl.pow(2).mean().sqrt():.4f}  max {rl.max():.4f}"
              f" | feature cosine min {cs.min():.6f}  mean {cs.mean():.7f}", flush=True)


if __name__ == "__main__":
    main()
