#!/usr/bin/env python
"""Stage-2 step timing at BASELINE configs[2] shape on ONE GPU (ViT-B/16, 16 x 224^2 frames = 3136 tokens, 8 classes, B = 16):
not the headline metric -- a parity-test configuration -- but useful to track the all-token path."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from types import SimpleNamespace
import unite_amd
from unite_amd.optim_factory import create_optimizer, LayerDecayValueAssigner
from unite_amd.utils import NativeScalerWithGradNormCount

B, T = int(os.environ.get("B", 16)), int(os.environ.get("T", 16))
dev = torch.device("cuda")
m = unite_amd.create_model("vit_base_patch16_224", pretrained=False, num_classes=8, all_frames=T, tubelet_size=1, drop_path_rate=0.1,
                           use_mean_pooling=True, init_scale=0.001, use_learnable_pos_emb=False, fc_drop_rate=0.0, drop_rate=0.0,
                           attn_drop_rate=0.0, use_checkpoint=False, checkpoint_num=0).to(dev).train()
nl = m.get_num_layers()
asg = LayerDecayValueAssigner([0.65 ** (nl + 1 - i) for i in range(nl + 2)])
args = SimpleNamespace(opt="adamw", weight_decay=0.05, lr=1e-3 * B / 256, opt_eps=1e-8, opt_betas=[0.9, 0.999])
opt = create_optimizer(args, m, skip_list=m.no_weight_decay(), get_num_layer=asg.get_layer_id, get_layer_scale=asg.get_scale)
scaler = NativeScalerWithGradNormCount()
vid = torch.randn(B, 3, T, 224, 224, device=dev)
lab = torch.randint(0, 8, (B,), device=dev)
def step():
    opt.zero_grad()
    loss, _ = m.forward_loss(vid, lab)
    return loss, scaler(loss, opt, clip_grad=None)
for _ in range(3): loss, gn = step()
torch.cuda.synchronize(); t0 = time.perf_counter()
K = 10
for _ in range(K): loss, gn = step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
gf = 2693e9 * (T / 16.0) if T == 16 else None
print(f"stage2 ViT-B/16 B={B} T={T}: {dt*1e3:.1f} ms/step, {B/dt:.1f} clips/s, loss {loss.item():.4f}, grad_norm {gn.item():.4f}" +
      (f", {B/dt*gf/2.5e15*100:.1f}% of bf16 peak" if gf else ""))
