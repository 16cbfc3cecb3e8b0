"""TEST INFRASTRUCTURE: the HIP teacher with each residual-stream type (f32 / bf16 / f16 rows, UNITE_TEACHER_RES16) against the fp32 CPU oracle over
seeded small towers -- used by tests/test_model_gpu.py::test_teacher_residual_stream_types_error_statistics and by the command-line front end
tools/teacher_stream_error.py (the oracle is imported here, under tests/, only)."""
import torch

from oracle import umt_oracle as O
from oracle.filler import fill_state_dict, make_videos
from tests.shapes import teacher_shapes


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
