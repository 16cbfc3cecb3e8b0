#!/usr/bin/env python
"""Unprofiled timeline of the overlapped stage-1 step (bench.py's loop, B = 32): HIP events at the start / end of every teacher phase (on the
teacher's stream) and of every student step (on the main stream), read back after the run.  Prints, averaged over the steps and relative to
the end of the previous student step: when the teacher phase launched in the same iteration starts and ends, when the student step starts and
ends.  python tools/step_timeline.py [steps]"""
import contextlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from types import SimpleNamespace
import unite_amd
from unite_amd import ops
from unite_amd.engine_stage1 import StepState, TeacherAhead, student_phase
from unite_amd.optim_factory import create_optimizer
from unite_amd.utils import NativeScalerWithGradNormCount

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda:0")
torch.manual_seed(0)
B, T = 32, 8
student = unite_amd.create_model("adaptation_umt_base_patch16_224", pretrained=False, drop_path_rate=0.1, drop_block_rate=None, use_learnable_pos_emb=False,
                                 use_checkpoint=False, checkpoint_num=0, clip_decoder_embed_dim=768, clip_output_dim=512, clip_norm_type='l2', num_frames=T,
                                 tubelet_size=1, clip_return_layers=[6, 7, 8, 9, 10, 11], clip_student_return_interval=1, use_cls_token=False).to(dev).train()
teacher = unite_amd.clip.clip_b16(pretrained=False, return_attn=True, clip_return_layers=[6, 7, 8, 9, 10, 11]).to(dev)
with contextlib.redirect_stdout(sys.stderr):
    opt = create_optimizer(SimpleNamespace(opt="adamw", weight_decay=0.05, lr=1.5e-4 * B / 256, opt_eps=1e-8, opt_betas=[0.9, 0.95]), student,
                           skip_list=student.no_weight_decay())
scaler = NativeScalerWithGradNormCount()
videos = torch.randn(B, 3, T, 224, 224, device=dev)
state = StepState()
ahead = TeacherAhead(teacher, state, dev, 0.8, 'attention')
torch.cuda.synchronize()
ev = []
touts = []
E = lambda: torch.cuda.Event(enable_timing=True)
starts = []
_next_slot = ahead.next_slot


def next_slot(inputs_ready=None):
    slot = _next_slot(inputs_ready)
    e = E()
    e.record(ahead.stream)            # first thing on the teacher's stream behind its slot wait
    starts.append(e)
    return slot


ahead.next_slot = next_slot


for i in range(steps + 8):
    with ops.plan(sharing=ahead.sharing):
        if not touts:
            touts.append(ahead.launch(videos, inputs_ready=False))
        cur = touts.pop()
        t_launch_host = E()
        # teacher phase of the next batch: start marker = first thing on the teacher stream after its slot wait; end = TeacherOut.ready
        nxt = ahead.launch(videos, inputs_ready=False)
        touts.append(nxt)
        s0, s1 = E(), E()
        s0.record()
        loss = student_phase(student, videos, cur, B, 'mixed')
        sf = E()
        sf.record()                                  # forward (encoder, decoders, loss) enqueued up to here
        opt.zero_grad()
        scaler(loss, opt, clip_grad=None, parameters=None, reducer=None)
        s1.record()
        te = E()
        with torch.cuda.stream(ahead.stream):
            te.record()                              # behind the teacher phase just launched
        ev.append((s0, s1, te, starts[-1], sf))
torch.cuda.synchronize()
rows = []
for i in range(8, len(ev) - 1):
    prev_end = ev[i - 1][1]
    s0, s1, te, ts, sf = ev[i]
    rows.append((prev_end.elapsed_time(s0), prev_end.elapsed_time(s1), prev_end.elapsed_time(te), ev[i - 1][2].elapsed_time(te), prev_end.elapsed_time(ts), prev_end.elapsed_time(sf)))
n = len(rows)
avg = [sum(r[k] for r in rows) / n for k in range(6)]
print(f"{n} steps; relative to the end of the previous student step (ms): student forward ends {avg[5]:+.2f}, step ends {avg[1]:+.2f} (= step time); "
      f"the teacher phase launched in this iteration starts {avg[4]:+.2f}, ends {avg[2]:+.2f}; teacher phase to teacher phase {avg[3]:.2f}")
