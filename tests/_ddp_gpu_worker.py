"""Child process of tests/test_ddp_gpu.py: one rank of a world-2 data-parallel run on ONE MI355X over gloo (the RCCL
rendezvous needs one GPU per rank; everything else -- the runtime's tag ranges, completion events from the weight-gradient
stream, bucket launches, finish() before the grad-norm, accumulation under no_sync -- is the code the N > 1 bench runs).
usage: python tests/_ddp_gpu_worker.py <rank> <world> <port> <out.pt>"""
import os
import sys
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["GPU_MAX_HW_QUEUES"] = "4"      # two ranks share the one GPU here: keep the default number of hardware queues (unite_amd/__init__.py)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", port
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    from functools import partial
    from oracle.filler import fill_state_dict, make_importance, make_videos
    from tests.shapes import TINY_S, TINY_T, TINY_V, student_shapes, teacher_shapes, vit_shapes
    from unite_amd.clip import VisionTransformer as Teacher
    from unite_amd.ddp import DistributedDataParallel
    from unite_amd.engine_stage1 import StepState, TeacherAhead, stage1_step, student_phase
    from unite_amd.modeling_adaptation import AdaptationVisionTransformer
    from unite_amd.modeling_finetune import VisionTransformer as Vit
    from unite_amd.optim_factory import create_optimizer
    from unite_amd.utils import NativeScalerWithGradNormCount
    res = {}

    # ---------------- stage 1: student + teacher, B = 4 split 2 + 2
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
    per = B // world
    lo, hi = rank * per, (rank + 1) * per
    s = student()
    if rank == 1:                                   # DDP must broadcast rank 0's weights (run_stage1.py:809)
        with torch.no_grad():
            for p in s.parameters():
                p.add_(0.5)
    model = DistributedDataParallel(s)
    rt = s.runtime()
    assert rt.runner.wgrad_stream, "the weight-gradient stream must be on: its events are what the reducer waits for"
    for it in range(2):                             # twice: steady-state reuse of events / pending sets
        rt.fp.accumulate = False
        loss = stage1_step(model, t, vid[lo:hi].contiguous(), per, 0.5, 'attention', None, 'mixed', StepState(), clip_input_resolution=32,
                           importance=imp[lo * 2:hi * 2].contiguous())
        loss.backward()
        model.reducer.finish()
        torch.cuda.synchronize()
    res["s1.launched"] = model.reducer.launched
    res["s1.grad"] = rt.fp.grad.clone().cpu()
    res["s1.loss"] = loss.item()
    # the same two steps with the teacher one batch ahead on its own stream (what train_one_epoch / bench.py run by default): the reducer's
    # buckets and the teacher's stream are independent of each other, the reduced gradient is the same
    ahead = TeacherAhead(t, StepState(), dev, 0.5, 'attention', clip_input_resolution=32)
    mine, imp_mine = vid[lo:hi].contiguous(), imp[lo * 2:hi * 2].contiguous()
    nxt = ahead.launch(mine, importance=imp_mine)
    for it in range(2):
        cur, nxt = nxt, ahead.launch(mine, importance=imp_mine)
        rt.fp.accumulate = False
        loss = student_phase(model, mine, cur, per, 'mixed')
        loss.backward()
        model.reducer.finish()
        torch.cuda.synchronize()
    ahead.close()
    res["s1.ahead_grad"] = rt.fp.grad.clone().cpu()
    res["s1.ahead_loss"] = loss.item()
    if rank == 0:
        s_full = student()
        rf = s_full.runtime()
        rf.fp.accumulate = False
        lf = stage1_step(s_full, t, vid, B, 0.5, 'attention', None, 'mixed', StepState(), clip_input_resolution=32, importance=imp)
        lf.backward()
        torch.cuda.synchronize()
        res["s1.full_grad"], res["s1.full_loss"] = rf.fp.grad.clone().cpu(), lf.item()

    # ---------------- stage 2: classifier wrapped in DDP, update_freq = 2 (one clip per micro-batch and rank)
    def vit():
        v = Vit(img_size=32, patch_size=16, embed_dim=128, depth=2, num_heads=2, mlp_ratio=4, qkv_bias=True,
                norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), num_classes=5, all_frames=4, tubelet_size=1, use_mean_pooling=True,
                init_scale=0.001)
        v.load_state_dict(fill_state_dict(vit_shapes(TINY_V), 5))
        return v.to(dev).train()

    vid2 = make_videos(4, 4, 32, 32, seed=31).to(dev)
    lab = torch.tensor([1, 4, 0, 2], device=dev)
    v = vit()
    m2 = DistributedDataParallel(v)
    tags = [t_ for t_, _, _ in v.runtime().tag_ranges()]
    assert tags == ["head", 1, 0, "patch_embed"], tags
    args = SimpleNamespace(opt="adamw", weight_decay=0.05, lr=0.0, opt_eps=1e-8, opt_betas=[0.9, 0.999])     # lr 0: the step leaves weights alone
    opt = create_optimizer(args, v, skip_list=v.no_weight_decay())
    scaler = NativeScalerWithGradNormCount()
    opt.zero_grad()
    before = m2.reducer.launched
    for mb in range(2):
        i = rank * 2 + mb
        l2, _ = v.forward_loss(vid2[i:i + 1].contiguous(), lab[i:i + 1].contiguous())
        gn = scaler(l2 / 2, opt, clip_grad=None, parameters=None, update_grad=mb == 1, reducer=m2.reducer)
        if mb == 0:
            assert gn is None and m2.reducer.launched == before
    torch.cuda.synchronize()
    res["s2.launched"] = m2.reducer.launched - before
    res["s2.grad"] = v.runtime().fp.grad.clone().cpu()
    res["s2.gn"] = gn.item()
    if rank == 0:
        vf = vit()
        lf2, _ = vf.forward_loss(vid2, lab)
        vf.runtime().fp.accumulate = False
        lf2.backward()
        torch.cuda.synchronize()
        res["s2.full_grad"] = vf.runtime().fp.grad.clone().cpu()
    # ---------------- AdamW per gradient bucket (UNITE_BUCKET_ADAMW / scaler.bucket_adamw) vs one launch after the whole backward
    def train(bucket):
        s3 = student()
        m3 = DistributedDataParallel(s3)
        a3 = SimpleNamespace(opt="adamw", weight_decay=0.05, lr=2e-3, opt_eps=1e-8, opt_betas=[0.9, 0.95])
        opt3 = create_optimizer(a3, s3, skip_list=s3.no_weight_decay())
        sc = NativeScalerWithGradNormCount()
        sc.bucket_adamw = bucket
        gns = []
        res["s1.init_params"] = s3.runtime().fp.param.clone().cpu()
        for it in range(3):
            l3 = stage1_step(m3, t, mine, per, 0.5, 'attention', None, 'mixed', StepState(), clip_input_resolution=32, importance=imp_mine)
            opt3.zero_grad()
            gns.append(sc(l3, opt3, clip_grad=None, parameters=None, reducer=m3.reducer).item())
        torch.cuda.synchronize()
        return s3.runtime().fp.param.clone().cpu(), gns
    res["s1.opt_params"], res["s1.opt_gn"] = train(False)
    res["s1.bucket_params"], res["s1.bucket_gn"] = train(True)
    torch.save(res, out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
