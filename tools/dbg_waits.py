import os, sys, contextlib
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, traceback
from types import SimpleNamespace
import unite_amd
from unite_amd import ops
from unite_amd.engine_stage1 import StepState, TeacherAhead, student_phase
from unite_amd.optim_factory import create_optimizer
from unite_amd.utils import NativeScalerWithGradNormCount
dev = torch.device("cuda:0")
B, T = 8, 8
student = unite_amd.create_model("adaptation_umt_base_patch16_224", pretrained=False, drop_path_rate=0.1, drop_block_rate=None, use_learnable_pos_emb=False,
                                 use_checkpoint=False, checkpoint_num=0, clip_decoder_embed_dim=768, clip_output_dim=512, clip_norm_type='l2', num_frames=T,
                                 tubelet_size=1, clip_return_layers=[6, 7, 8, 9, 10, 11], clip_student_return_interval=1, use_cls_token=False).to(dev).train()
teacher = unite_amd.clip.clip_b16(pretrained=False, return_attn=True, clip_return_layers=[6, 7, 8, 9, 10, 11]).to(dev)
with contextlib.redirect_stdout(sys.stderr):
    opt = create_optimizer(SimpleNamespace(opt="adamw", weight_decay=0.05, lr=1e-4, opt_eps=1e-8, opt_betas=[0.9, 0.95]), student, skip_list=student.no_weight_decay())
scaler = NativeScalerWithGradNormCount()
videos = torch.randn(B, 3, T, 224, 224, device=dev)
state = StepState()
ahead = TeacherAhead(teacher, state, dev, 0.8, 'attention')
orig = torch.cuda.Stream.wait_event
log = []
def we(self, ev):
    if LOG[0]:
        fr = traceback.extract_stack(limit=4)
        log.append((self.cuda_stream == ahead.stream.cuda_stream, [f"{f.filename.split('/')[-1]}:{f.lineno}" for f in fr[:-1]]))
    return orig(self, ev)
torch.cuda.Stream.wait_event = we
LOG = [False]
touts = []
for i in range(6):
    LOG[0] = i == 5
    with ops.plan(sharing=0.8):
        if not touts:
            touts.append(ahead.launch(videos, inputs_ready=False))
        cur = touts.pop()
        if LOG[0]: log.append(("--- launch", []))
        touts.append(ahead.launch(videos, inputs_ready=False))
        if LOG[0]: log.append(("--- student", []))
        loss = student_phase(student, videos, cur, B, 'mixed')
        opt.zero_grad()
        scaler(loss, opt, clip_grad=None, parameters=None, reducer=None)
torch.cuda.synchronize()
for on_teacher, where in log:
    print("teacher-stream" if on_teacher is True else ("main/other" if on_teacher is False else on_teacher), " <- ".join(where))
