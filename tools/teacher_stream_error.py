"""What the type of the frozen teacher's residual stream costs in accuracy: small CLIP towers (width 128, 3 blocks, 4 x 4 patch grid, 2 frames) with
seeded random weights and clips, the HIP teacher with an f32 / bf16 / f16 stream (UNITE_TEACHER_RES16) against the fp32 CPU oracle
(oracle/umt_oracle.py teacher_forward, pinned on the reference's own vectors).  Over `seeds` models: error of the CLS attention row (the mask
weights, clip.py:95-96,183) and cosine of the L2-normalised target features.  One fixture is 16 values -- too few to tell the streams apart
(any perturbation re-randomises the bf16 roundings downstream), hence the statistics.
Usage: python tools/teacher_stream_error.py [seeds]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import umt_oracle as O  # noqa: E402   (test infrastructure: this tool is a checker, not a product path)
from oracle.filler import fill_state_dict, make_videos  # noqa: E402
from tests.shapes import teacher_shapes  # noqa: E402


def collect(seeds: int):
    """{stream: (attention abs errors, attention relative errors, feature cosines)} over `seeds` seeded towers, each against the fp32 oracle"""
    from unite_amd.clip import VisionTransformer
    cfg = O.TeacherCfg(input_resolution=64, patch_size=16, width=128, layers=3, heads=2, output_dim=64, clip_return_layers=(1, 2))
    acc = {m: dict(abs=[], rel=[], cos=[]) for m in ("f32", "bf16", "f16")}
    for seed in range(seeds):
        sd = fill_state_dict(teacher_shapes(cfg), 1000 + seed)
        vid = make_videos(2, 2, 64, 64, seed=2000 + seed)
        ref_f, ref_a = O.teacher_forward(sd, vid, cfg, return_attn=True)
        for name, mode in (("f32", False), ("bf16", True), ("f16", "f16")):
            t = VisionTransformer(input_resolution=64, patch_size=16, width=128, layers=3, heads=2, output_dim=64, return_attn=True,
                                  clip_return_layers=[1, 2])
            t.load_state_dict(sd)
            t = t.to("cuda").eval()
            t.runtime().res16 = mode
            feats, attn = t(vid.to("cuda"))
            e = (attn.cpu() - ref_a).abs()
            acc[name]["abs"].append(e.flatten())
            acc[name]["rel"].append((e / ref_a.abs()).flatten())
            acc[name]["cos"].append(torch.nn.functional.cosine_similarity(feats.cpu().flatten(0, -2), ref_f.flatten(0, -2), dim=-1))
    return {k: (torch.cat(v["abs"]), torch.cat(v["rel"]), torch.cat(v["cos"])) for k, v in acc.items()}


def main():
    seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    stats = collect(seeds)
    print(f"# {seeds} models x 4 frames x 16 attention values (mean value {1 / 16:.4f}); features: {stats['f32'][2].numel() // seeds} rows per model")
    for name, (ab, rl, cs) in stats.items():
        print(f"{name:5s} attention abs err rms {ab.pow(2).mean().sqrt():.2e}  max {ab.max():.2e} | rel err rms {rl.pow(2).mean().sqrt():.4f}  max {rl.max():.4f}"
              f" | feature cosine min {cs.min():.6f}  mean {cs.mean():.7f}", flush=True)


if __name__ == "__main__":
    main()
