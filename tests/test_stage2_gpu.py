"""Stage-2 (source fine-tuning) path on the MI355X against the reference's golden vectors (-m gpu):
VisionTransformer forward (all tokens, mean-pool, fc_norm, head), CE loss, every parameter gradient, the engine with
gradient accumulation.  Tolerances as in test_model_gpu.py (bf16 operands): logits abs 2e-2 on O(1) values,
loss relative 2e-3, per-tensor gradient relative L2 <= 5e-2."""
import os
from functools import partial
from types import SimpleNamespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import umt_oracle as O  # noqa: E402
from oracle.filler import fill_state_dict, make_videos  # noqa: E402
from tests.shapes import TINY_V, vit_shapes  # noqa: E402

DEV = "cuda"


def _load(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name))
    return {k: z[k] for k in z.files}


def rel_l2(a, b):
    return ((a.float() - b.float()).norm() / (b.float().norm() + 1e-12)).item()


def build_vit(drop_path=0.0):
    from unite_amd.modeling_finetune import VisionTransformer
    return VisionTransformer(img_size=32, patch_size=16, embed_dim=128, depth=2, num_heads=2, mlp_ratio=4, qkv_bias=True,
                             norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), num_classes=5, all_frames=4, tubelet_size=1,
                             use_mean_pooling=True, init_scale=0.001, drop_path_rate=drop_path)


def test_vit_stage2_tiny_vs_reference_golden(golden_dir):
    z = _load(golden_dir, "vit_stage2_tiny.npz")
    m = build_vit()
    assert [(k, tuple(v.shape)) for k, v in m.state_dict().items()] == vit_shapes(TINY_V)
    m.load_state_dict(fill_state_dict(vit_shapes(TINY_V), int(z["in.seed_weights"])))
    m = m.to(DEV).train()
    vid = torch.from_numpy(z["in.videos"]).to(DEV)
    labels = torch.from_numpy(z["in.labels"]).to(DEV)
    logits = m(vid)
    torch.testing.assert_close(logits.detach().cpu(), torch.from_numpy(z["out.logits"]), atol=2e-2, rtol=2e-2)
    loss = torch.nn.functional.cross_entropy(logits, labels)
    assert abs(loss.item() - float(z["out.loss"])) <= 2e-3 * float(z["out.loss"])
    loss.backward()
    for k, p in m.named_parameters():
        ref = torch.from_numpy(z["g." + k])
        assert rel_l2(p.grad.cpu(), ref) <= 5e-2 or (p.grad.cpu() - ref).abs().max() <= 1e-6, (k, rel_l2(p.grad.cpu(), ref))
    # fused CE path gives the same loss and gradients
    rt = m.runtime()
    g1 = rt.fp.grad.clone()
    m.zero_grad()
    rt.fp.accumulate = False
    loss2, logits2 = m.forward_loss(vid, labels)
    loss2.backward()
    assert abs(loss2.item() - loss.item()) <= 1e-5 and torch.equal(logits2, logits.detach())
    assert rel_l2(rt.fp.grad, g1) <= 5e-3      # torch CE gradient vs the fused softmax-CE kernel, then bf16 GEMM operands


def test_stage2_engine_accumulation_and_layer_decay():
    """engine_for_finetuning.train_one_epoch with update_freq = 2 equals one step on the concatenated batch;
    layer-decay parameter groups (run_stage2.py:700-720 -> optim_factory.py:44-73) drive the fused optimizer."""
    from unite_amd.engine_for_finetuning import train_one_epoch
    from unite_amd.optim_factory import create_optimizer, LayerDecayValueAssigner
    from unite_amd.utils import NativeScalerWithGradNormCount
    sd = fill_state_dict(vit_shapes(TINY_V), 5)
    vids = make_videos(4, 4, 32, 32, seed=8)
    labels = torch.tensor([1, 4, 0, 2])
    results = []
    for update_freq in (2, 1):
        m = build_vit()
        m.load_state_dict(sd)
        m = m.to(DEV)
        nl = m.get_num_layers()
        assigner = LayerDecayValueAssigner([0.65 ** (nl + 1 - i) for i in range(nl + 2)])
        args = SimpleNamespace(opt="adamw", weight_decay=0.05, lr=1e-3, opt_eps=1e-8, opt_betas=[0.9, 0.999])
        opt = create_optimizer(args, m, skip_list=m.no_weight_decay(), get_num_layer=assigner.get_layer_id, get_layer_scale=assigner.get_scale)
        assert len(opt.param_groups) == 8 and min(g["lr_scale"] for g in opt.param_groups) == pytest.approx(0.65 ** 3)
        if update_freq == 2:
            loader = [(vids[:2], labels[:2], None, None), (vids[2:], labels[2:], None, None)]
        else:
            loader = [(vids, labels, None, None)]
        stats = train_one_epoch(m, torch.nn.CrossEntropyLoss(), loader, opt, torch.device(DEV), 0, NativeScalerWithGradNormCount(), None,
                                start_steps=0, lr_schedule_values=[1e-3], wd_schedule_values=None,
                                num_training_steps_per_epoch=1, update_freq=update_freq)
        assert np.isfinite(stats["loss"]) and "class_acc" in stats and "grad_norm" in stats
        results.append({k: v.detach().cpu().clone() for k, v in m.state_dict().items()})
    for k in results[0]:
        # two half-batches accumulated == one full batch (mean CE): same update up to bf16 / summation-order effects
        assert (results[0][k] - results[1][k]).abs().max() <= 2.5e-3, k
        assert (results[0][k] - results[1][k]).abs().mean() <= 1e-4, k


def test_vit_stage2_drop_path_vs_oracle():
    m = build_vit(drop_path=0.4)
    sd = fill_state_dict(vit_shapes(TINY_V), 5)
    m.load_state_dict(sd)
    m = m.to(DEV).train()
    vid = make_videos(3, 4, 32, 32, seed=6)
    rt = m.runtime()
    u = torch.rand(2, 2, 3, generator=torch.Generator().manual_seed(3))
    keep = (1 - torch.linspace(0, 0.4, 2)).view(-1, 1, 1)
    scales = torch.floor(keep + u) / keep
    rt.runner.drop_path_scales = lambda B, training: scales.to(DEV)
    logits = m(vid.to(DEV))
    # oracle with the same keep decisions
    x = O.im2col(vid, 16, 1) @ sd["patch_embed.proj.weight"].reshape(128, -1).t() + sd["patch_embed.proj.bias"]
    x = x + O.sinusoid_table(16, 128)
    rates = [r.item() for r in torch.linspace(0, 0.4, 2)]
    for i in range(2):
        x = O.vit_block(x, sd, f"blocks.{i}.", 2, 1e-6, rates[i], True, (u[i, 0], u[i, 1]))
    ref = torch.nn.functional.linear(O.layer_norm(x.mean(1), sd["fc_norm.weight"], sd["fc_norm.bias"], 1e-6), sd["head.weight"], sd["head.bias"])
    torch.testing.assert_close(logits.detach().cpu(), ref, atol=2e-2, rtol=2e-2)


def test_stage2_full_size_batch_split_equivalence():
    """BASELINE config 3 at its full sequence length (ViT-B/16, 16 f x 224^2 = 3136 tokens: the flash-style tiled attention kernels,
    forward and both backward kernels) through a size-independent property: CE over 2 clips = mean of the two single-clip losses, and
    the gradients are the mean of the single-clip gradients.  No oracle at this size (a CPU pass takes minutes)."""
    import unite_amd
    m = unite_amd.create_model("vit_base_patch16_224", pretrained=False, num_classes=8, all_frames=16, tubelet_size=1,
                               use_mean_pooling=True, drop_path_rate=0.0, init_scale=0.001).to(DEV).train()
    g = torch.Generator().manual_seed(11)
    with torch.no_grad():
        m.head.weight.copy_(torch.randn(8, 768, generator=g) * 0.05)          # init_scale 0.001 would make the logits ~0
    vid = make_videos(2, 16, 224, 224, seed=12).to(DEV)
    labels = torch.tensor([3, 6], device=DEV)
    rt = m.runtime()

    def run(lo, hi):
        rt.fp.accumulate = False
        logits = m(vid[lo:hi].contiguous())
        loss = torch.nn.functional.cross_entropy(logits.float(), labels[lo:hi])
        loss.backward()
        torch.cuda.synchronize()
        assert torch.isfinite(logits).all()
        return loss.item(), rt.fp.grad.clone()

    l_all, g_all = run(0, 2)
    l_a, g_a = run(0, 1)
    l_b, g_b = run(1, 2)
    assert abs(l_all - 0.5 * (l_a + l_b)) <= 1e-4 * max(1.0, abs(l_all))
    assert g_all.abs().max() > 0
    assert rel_l2(g_all, 0.5 * (g_a + g_b)) <= 2e-3


def test_stage2_parity_at_the_config3_batch():
    """BASELINE config 3 at its per-GPU batch (B = 16 clips of 16 f x 224^2 = 50 176 token rows per GEMM: the persistent / 256^2 tile
    kernels and split-K plans of that size, the tiled attention kernels at 3136 tokens) against eight B = 2 steps on the same clips and
    labels: the cross-entropy over 16 clips is the mean of the eight losses (1e-4) and the gradient the mean of the gradients (2e-3
    relative L2); the B = 2 shape of this model is tied to the reference by test_vit_stage2_tiny_vs_reference_golden (same kernels)."""
    import unite_amd
    m = unite_amd.create_model("vit_base_patch16_224", pretrained=False, num_classes=8, all_frames=16, tubelet_size=1,
                               use_mean_pooling=True, drop_path_rate=0.0, init_scale=0.001).to(DEV).train()
    g = torch.Generator().manual_seed(21)
    with torch.no_grad():
        m.head.weight.copy_(torch.randn(8, 768, generator=g) * 0.05)
    B = 16
    vid = make_videos(B, 16, 224, 224, seed=22).to(DEV)
    labels = torch.randint(0, 8, (B,), generator=g).to(DEV)
    rt = m.runtime()

    def run(lo, hi):
        rt.fp.accumulate = False
        logits = m(vid[lo:hi].contiguous())
        loss = torch.nn.functional.cross_entropy(logits.float(), labels[lo:hi])
        loss.backward()
        torch.cuda.synchronize()
        return loss.item(), rt.fp.grad.clone()

    l_all, g_all = run(0, B)
    ls, gsum = [], torch.zeros_like(g_all)
    for j in range(B // 2):
        l, gj = run(2 * j, 2 * j + 2)
        ls.append(l)
        gsum += gj
    assert abs(l_all - sum(ls) / len(ls)) <= 1e-4 * max(1.0, abs(l_all))
    assert rel_l2(g_all, gsum / len(ls)) <= 2e-3


@pytest.mark.timeout(1200)
def test_stage2_full_size_vs_oracle():
    """BASELINE config 3's MODEL at full size -- vit_base_patch16_224(all_frames=16, num_classes=8): 3136 tokens per clip, i.e. the tiled attention
    kernels (attention_tiled.hip), the 16 f x 196 sinusoid table, fc_norm(mean over 3136 tokens), the 8-class head -- against the fp32 CPU oracle
    (modeling_finetune.py:356-383 restated; one forward + backward of two clips), at the B = 2 shape whose eight sub-batches
    test_stage2_parity_at_the_config3_batch averages into the B = 16 step: logits, cross-entropy, the global gradient norm and gradient tensors
    from the first, a middle and the last layer.  Tolerances: bf16 operands through 12 blocks (logits abs 6e-2 = 1.5 x the measured 0.037 on values up to 6, loss relative
    2e-3, gradient norm 2e-2, per-tensor gradients relative L2 5e-2)."""
    import unite_amd
    cfg = O.VitCfg()                                     # img 224, patch 16, 768 x 12 x 12, 8 classes, 16 frames
    assert cfg.num_patches == 3136
    m = unite_amd.create_model("vit_base_patch16_224", pretrained=False, num_classes=8, all_frames=16, tubelet_size=1,
                               use_mean_pooling=True, drop_path_rate=0.0, init_scale=0.001)
    sd = fill_state_dict(vit_shapes(cfg), 41)
    sd["head.weight"] = sd["head.weight"] * 4.0          # spread the logits: a cross-entropy that depends on them
    assert [(k, tuple(v.shape)) for k, v in m.state_dict().items()] == vit_shapes(cfg)
    m.load_state_dict(sd)
    m = m.to(DEV).train()
    B = 2
    vid = make_videos(B, 16, 224, 224, seed=42)
    labels = torch.tensor([5, 2])
    # ---- oracle: fp32 autograd on the CPU
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    keys = ["patch_embed.proj.weight", "blocks.0.attn.qkv.weight", "blocks.0.attn.q_bias", "blocks.5.mlp.fc1.weight", "blocks.11.attn.proj.weight",
            "blocks.11.mlp.fc2.bias", "fc_norm.weight", "head.weight"]
    leaf = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref_logits = O.vit_classifier_forward(leaf, vid, cfg)
    ref_loss = torch.nn.functional.cross_entropy(ref_logits, labels)
    ref_loss.backward()
    ref_gn = O.grad_norm([p.grad for p in leaf.values()]).item()
    ref_g = {k: leaf[k].grad.clone() for k in keys}
    ref_logits, ref_loss = ref_logits.detach(), ref_loss.item()
    del leaf
    # ---- HIP path
    logits = m(vid.to(DEV))
    loss = torch.nn.functional.cross_entropy(logits.float(), labels.to(DEV))
    loss.backward()
    print(f"[stage2 full size] logits max err {(logits.detach().float().cpu() - ref_logits).abs().max().item():.3e} (|logits| <= {ref_logits.abs().max().item():.2f}), "
          f"loss {loss.item():.5f} vs {ref_loss:.5f}")
    torch.testing.assert_close(logits.detach().float().cpu(), ref_logits, atol=6e-2, rtol=2e-2)
    assert abs(loss.item() - ref_loss) <= 2e-3 * max(1.0, abs(ref_loss)), (loss.item(), ref_loss)
    gn = m.runtime().fp.grad.norm().item()
    assert abs(gn - ref_gn) <= 2e-2 * ref_gn, (gn, ref_gn)
    got = dict(m.named_parameters())
    for k in keys:
        assert rel_l2(got[k].grad.cpu(), ref_g[k]) <= 5e-2, (k, rel_l2(got[k].grad.cpu(), ref_g[k]))
