"""Child process of tests/test_ddp_gpu.py::test_rccl_path_one_rank: a ONE-rank process group on RCCL (backend 'nccl') with
UNITE_DDP_FORCE_COLLECTIVES=1, so that the reducer issues its bucket collectives as it does at N > 1: broadcast of the flat parameter buffer,
completion events from the main and the weight-gradient stream, ncclAvg all-reduce per bucket on the reducer's stream, the join in finish(),
per-bucket AdamW behind the collectives.  A mean over one rank changes nothing, so everything must equal the unwrapped step bit for bit.
usage: python tests/_ddp_rccl_worker.py <port> <out.pt>"""
import os
import sys
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["UNITE_DDP_FORCE_COLLECTIVES"] = "1"

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    port, out = sys.argv[1], sys.argv[2]
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", port
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    from functools import partial
    from oracle.filler import fill_state_dict, make_importance, make_videos
    from tests.shapes import TINY_S, TINY_T, student_shapes, teacher_shapes
    from unite_amd.clip import VisionTransformer as Teacher
    from unite_amd.ddp import DistributedDataParallel
    from unite_amd.engine_stage1 import StepState, stage1_step
    from unite_amd.modeling_adaptation import AdaptationVisionTransformer
    from unite_amd.optim_factory import create_optimizer
    from unite_amd.utils import NativeScalerWithGradNormCount

    def student():
        s = AdaptationVisionTransformer(img_size=32, patch_size=16, encoder_embed_dim=128, encoder_depth=3, encoder_num_heads=2,
                                        mlp_ratio=4, qkv_bias=True, norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), num_frames=2,
                                        tubelet_size=1, clip_decoder_embed_dim=128, clip_output_dim=64, clip_return_layers=[1, 2])
        s.load_state_dict(fill_state_dict(student_shapes(TINY_S), 3))
        return s.to(dev).train()

    t = Teacher(input_resolution=32, patch_size=16, width=128, layers=3, heads=2, output_dim=64, return_attn=True, clip_return_layers=[1, 2])
    t.load_state_dict(fill_state_dict(teacher_shapes(TINY_T), 1))
    t = t.to(dev)
    B = 4
    vid = make_videos(B, 2, 32, 32, seed=21).to(dev)
    imp = make_importance(B * 2, 4, seed=22).to(dev)
    res = {}

    def train(wrap, bucket):
        s = student()
        m = DistributedDataParallel(s, bucket_cap_mb=0) if wrap else s          # cap 0: one bucket per layer -> several collectives per step
        a = SimpleNamespace(opt="adamw", weight_decay=0.05, lr=2e-3, opt_eps=1e-8, opt_betas=[0.9, 0.95])
        opt = create_optimizer(a, s, skip_list=s.no_weight_decay())
        sc = NativeScalerWithGradNormCount()
        sc.bucket_adamw = bucket
        gns = []
        for _ in range(3):
            loss = stage1_step(m, t, vid, B, 0.5, 'attention', None, 'mixed', StepState(), clip_input_resolution=32, importance=imp)
            opt.zero_grad()
            gns.append(sc(loss, opt, clip_grad=None, parameters=None, reducer=getattr(m, "reducer", None)).item())
        torch.cuda.synchronize()
        launched = m.reducer.launched if wrap else 0
        return s.runtime().fp.param.clone().cpu(), s.runtime().fp.grad.clone().cpu(), gns, launched

    res["plain"] = train(False, False)
    res["ddp"] = train(True, False)
    res["ddp_bucket"] = train(True, True)
    res["backend"] = dist.get_backend()
    torch.save(res, out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
