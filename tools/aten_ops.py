import os, sys, torch
sys.path.insert(0, '.')
import unite_amd
from unite_amd.engine_stage1 import StepState, stage1_step
from unite_amd.optim_factory import create_optimizer
from unite_amd.utils import NativeScalerWithGradNormCount
from types import SimpleNamespace
dev = torch.device('cuda'); B, T = 8, 8
student = unite_amd.create_model("adaptation_umt_base_patch16_224", pretrained=False, drop_path_rate=0.1, drop_block_rate=None, use_learnable_pos_emb=False,
    use_checkpoint=False, checkpoint_num=0, clip_decoder_embed_dim=768, clip_output_dim=512, clip_norm_type='l2', num_frames=T,
    tubelet_size=1, clip_return_layers=[6, 7, 8, 9, 10, 11], clip_student_return_interval=1, use_cls_token=False).to(dev).train()
teacher = unite_amd.clip.clip_b16(pretrained=False, return_attn=True, clip_return_layers=[6, 7, 8, 9, 10, 11]).to(dev)
args = SimpleNamespace(opt="adamw", weight_decay=0.05, lr=1e-5, opt_eps=1e-8, opt_betas=[0.9, 0.95])
opt = create_optimizer(args, student, skip_list=student.no_weight_decay())
scaler = NativeScalerWithGradNormCount()
videos = torch.randn(B, 3, T, 224, 224, device=dev); state = StepState()
def step():
    loss = stage1_step(student, teacher, videos, B, 0.8, 'attention', None, 'mixed', state)
    opt.zero_grad()
    return loss, scaler(loss, opt, clip_grad=None, parameters=None)
for _ in range(3): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], record_shapes=False) as prof:
    for _ in range(2): step()
    torch.cuda.synchronize()
rows = sorted(prof.key_averages(), key=lambda e: -e.count)
for e in rows[:25]:
    print(f"{e.key[:50]:50s} count/step {e.count/2:7.1f}  cpu_us/step {e.cpu_time_total/2:9.1f}")
