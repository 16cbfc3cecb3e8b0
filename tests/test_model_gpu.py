"""End-to-end parity (-m gpu): the HIP path (unite_amd models on libunite_hip.so) against
  (a) the CPU oracle on identical weights / clips / masks, and
  (b) the golden vectors the REFERENCE itself produced (tests/golden, oracle/make_golden.py).
Stated tolerances (bf16 GEMM operands, fp32 accumulation / statistics):
  loss: relative 1e-3 (the north-star bound);  L2-normalised features: cosine >= 0.999;
  teacher CLS attention: abs 2e-3 of values ~1/196;  gradients: relative L2 error per tensor <= 5e-2,
  global grad-norm relative 2e-2.
"""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import umt_oracle as O  # noqa: E402
from oracle.filler import fill_state_dict, make_importance, make_videos  # noqa: E402
from tests.shapes import TINY_S, TINY_T, student_shapes, teacher_shapes  # noqa: E402

DEV = "cuda"


def _load(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name))
    return {k: z[k] for k in z.files}


def cos_min(a, b):
    return torch.nn.functional.cosine_similarity(a.flatten(0, -2).float(), b.flatten(0, -2).float(), dim=-1).min().item()


def rel_l2(a, b):
    return ((a.float() - b.float()).norm() / (b.float().norm() + 1e-12)).item()


def build_tiny():
    from unite_amd.modeling_adaptation import AdaptationVisionTransformer
    from unite_amd.clip import VisionTransformer
    from functools import partial
    s = AdaptationVisionTransformer(img_size=32, patch_size=16, encoder_embed_dim=128, encoder_depth=3, encoder_num_heads=2,
                                    mlp_ratio=4, qkv_bias=True, norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), num_frames=2,
                                    tubelet_size=1, clip_decoder_embed_dim=128, clip_output_dim=64, clip_return_layers=[1, 2])
    t = VisionTransformer(input_resolution=32, patch_size=16, width=128, layers=3, heads=2, output_dim=64, return_attn=True,
                          clip_return_layers=[1, 2])
    return s, t


def test_state_dict_contract_tiny():
    s, t = build_tiny()
    assert [(k, tuple(v.shape)) for k, v in s.state_dict().items()] == student_shapes(TINY_S)
    assert sorted((k, tuple(v.shape)) for k, v in t.state_dict().items()) == sorted(teacher_shapes(TINY_T))


@pytest.mark.parametrize("stream", ["f32", "default"])
def test_teacher_tiny_vs_reference_golden(golden_dir, stream):
    z = _load(golden_dir, "teacher_tiny.npz")
    _, t = build_tiny()
    t.load_state_dict(fill_state_dict(teacher_shapes(TINY_T), int(z["in.seed_weights"])))
    t = t.to(DEV).eval()
    if stream == "f32":
        t.runtime().res16 = False          # the reference's own residual-stream type (UNITE_TEACHER_RES16=0)
    else:
        assert t.runtime().res16 == "f16"  # the default: IEEE-half rows, as OpenAI's CLIP runs
    feats, attn = t(torch.from_numpy(z["in.videos"]).to(DEV))
    ref_f, ref_a = torch.from_numpy(z["out.feats"]), torch.from_numpy(z["out.attn"])
    assert feats.shape == ref_f.shape and attn.shape == ref_a.shape
    assert cos_min(feats.cpu(), ref_f) >= 0.999
    # the fixture's CLS attention rows are 4 values of 0.026 .. 0.49 (a 2 x 2 patch grid), not the 1 / 196 of the full model.  With f32 rows the
    # bound is 2.4 % .. 9.6 % of a value; measured on MI355X (round 4): max abs error 4.1e-3, max relative error 2.4 %, feature cosine 0.99994 --
    # bf16 operands of the score product, nothing to spare.  The f16 rows of the default get the bound every other teacher test of this file
    # states (2e-3 + 5 %): on these 16 values they land at 7.3e-3 / 6 %, while over 1536 values of 24 seeded towers their error is that of
    # the f32 rows (1.43e-3 against 1.45e-3 rms, profiles/r04_teacher_stream_error.txt) -- any perturbation re-draws the bf16 roundings downstream.
    torch.testing.assert_close(attn.cpu(), ref_a, atol=2e-3, rtol=2e-2 if stream == "f32" else 5e-2)
    assert cos_min(feats.cpu(), ref_f) >= 0.9998


def test_teacher_patch14_vs_oracle():
    """CLIP-L/14 geometry (patch 14: im2col / conv1 rows padded 588 -> 592; 2 x 2 grid here) against the oracle, whose
    teacher_forward is pinned on the reference's patch-16 vectors above and is patch-size agnostic."""
    from unite_amd.clip import VisionTransformer
    cfg = O.TeacherCfg(input_resolution=28, patch_size=14, width=128, layers=2, heads=2, output_dim=64, clip_return_layers=(0, 1))
    t = VisionTransformer(input_resolution=28, patch_size=14, width=128, layers=2, heads=2, output_dim=64, return_attn=True,
                          clip_return_layers=[0, 1])
    sd = fill_state_dict(teacher_shapes(cfg), 77)
    t.load_state_dict(sd)
    t = t.to(DEV).eval()
    vid = make_videos(3, 2, 28, 28, seed=78)
    feats, attn = t(vid.to(DEV))
    ref_f, ref_a = O.teacher_forward(sd, vid, cfg, return_attn=True)
    assert feats.shape == ref_f.shape and attn.shape == ref_a.shape
    assert cos_min(feats.cpu(), ref_f) >= 0.999
    torch.testing.assert_close(attn.cpu(), ref_a, atol=2e-3, rtol=5e-2)


def test_teacher_last_block_not_a_tap_vs_oracle():
    """taps (0, 1) of 3 blocks: the last block then only contributes its CLS attention row (clip.py:95-96) -- the HIP teacher
    stops after that block's qkv projection; features and attention must still match the oracle.  Visible-row targets of a
    model whose last block IS a tap go through the pruned path (out_proj + MLP on the listed rows only): compared row by row."""
    from unite_amd.clip import VisionTransformer
    vid = make_videos(2, 2, 32, 32, seed=41)
    for taps in ((0, 1), (1, 2)):
        cfg = O.TeacherCfg(input_resolution=32, patch_size=16, width=128, layers=3, heads=2, output_dim=64, clip_return_layers=taps)
        t = VisionTransformer(input_resolution=32, patch_size=16, width=128, layers=3, heads=2, output_dim=64, return_attn=True,
                              clip_return_layers=list(taps))
        sd = fill_state_dict(teacher_shapes(cfg), 40)
        t.load_state_dict(sd)
        t = t.to(DEV).eval()
        feats, attn = t(vid.to(DEV))
        ref_f, ref_a = O.teacher_forward(sd, vid, cfg, return_attn=True)
        assert cos_min(feats.cpu(), ref_f) >= 0.999
        torch.testing.assert_close(attn.cpu(), ref_a, atol=2e-3, rtol=5e-2)
        # a subset of rows in scrambled order (row = frame * 5 + 1 + patch: CLS rows are skipped)
        rows = torch.tensor([6, 1, 19, 13, 2, 17], dtype=torch.int32, device=DEV)
        t.forward_attention(vid.to(DEV))
        sub = t.visible_targets(rows, 6).view(2, 6, 64).cpu()
        frame, patch = (rows.cpu() // 5).long(), (rows.cpu() % 5 - 1).long()
        want = ref_f.reshape(2, 2, 2, 4, 64)[:, frame // 2, frame % 2, patch]          # (K, B, T, HW, C) indexed per row
        assert cos_min(sub, want) >= 0.999


def test_stage1_step_with_resized_patch14_teacher_vs_oracle():
    """cfg-5 geometry in miniature: student 32 x 32 @ patch 16, teacher 28 x 28 @ patch 14 (both 2 x 2 grids), clips resized
    32 -> 28 bicubically for the teacher only (run_stage1.py:362-370)."""
    from unite_amd.clip import VisionTransformer
    from unite_amd.engine_stage1 import stage1_step, StepState
    s, _ = build_tiny()
    tcfg = O.TeacherCfg(input_resolution=28, patch_size=14, width=128, layers=3, heads=2, output_dim=64, clip_return_layers=(1, 2))
    t = VisionTransformer(input_resolution=28, patch_size=14, width=128, layers=3, heads=2, output_dim=64, return_attn=True,
                          clip_return_layers=[1, 2])
    ssd, tsd = fill_state_dict(student_shapes(TINY_S), 61), fill_state_dict(teacher_shapes(tcfg), 62)
    s.load_state_dict(ssd)
    t.load_state_dict(tsd)
    s, t = s.to(DEV).train(), t.to(DEV).eval()
    B = 3
    vid = make_videos(B, 2, 32, 32, seed=63)
    imp = make_importance(B * 2, 4, seed=64)
    loss = stage1_step(s, t, vid.to(DEV), B, 0.5, 'attention', None, 'mixed', StepState(), clip_input_resolution=28, importance=imp.to(DEV))
    mask = O.mask_from_importance(imp, 2, B)
    ssd_g = {k: v.clone().requires_grad_(True) for k, v in ssd.items()}
    ref, _, _, _ = O.stage1_loss(ssd_g, tsd, vid, mask, TINY_S, tcfg)
    assert abs(loss.item() - ref.item()) <= 1e-3 * abs(ref.item())
    loss.backward()
    ref.backward()
    for k, p in s.named_parameters():
        assert rel_l2(p.grad.cpu(), ssd_g[k].grad) <= 5e-2, k


def test_student_tiny_vs_reference_golden(golden_dir):
    """forward (x_clip, x_vis), loss, every parameter gradient, and 3 AdamW steps -- all against the reference's vectors."""
    from unite_amd.optim_factory import create_optimizer
    from unite_amd.utils import NativeScalerWithGradNormCount
    z = _load(golden_dir, "student_tiny.npz")
    s, _ = build_tiny()
    s.load_state_dict(fill_state_dict(student_shapes(TINY_S), int(z["in.seed_weights"])))
    s = s.to(DEV).train()
    vid = torch.from_numpy(z["in.videos"]).to(DEV)
    mask = torch.from_numpy(z["in.mask"]).to(DEV)
    tgt = torch.from_numpy(z["out.targets"]).to(DEV)
    # generic autograd path: model(x, mask, clip_only=True) -> loss in torch -> backward
    out = s(vid, mask, clip_only=True)
    assert cos_min(out.detach().cpu(), torch.from_numpy(z["out.x_clip"])) >= 0.999
    loss = (2 - 2 * (out * tgt).sum(dim=-1)).mean()
    assert abs(loss.item() - float(z["out.loss"])) <= 1e-3 * abs(float(z["out.loss"]))
    loss.backward()
    worst = 0.0
    for k, p in s.named_parameters():
        e = rel_l2(p.grad.cpu(), torch.from_numpy(z["g." + k]))
        worst = max(worst, e)
        assert e <= 5e-2, (k, e)
    # x_vis path
    with torch.no_grad():
        x_vis, x_clip = s(vid, mask, clip_only=False)
    assert rel_l2(x_vis.cpu(), torch.from_numpy(z["out.x_vis"])) <= 2e-2
    # fused loss path == generic path
    K, B, n_vis, C = tgt.shape
    rt = s.runtime()
    vis, nv = rt.tokens_from_mask(mask)
    assert nv == n_vis
    g_generic = rt.fp.grad.clone()
    s.zero_grad()
    rt.fp.accumulate = False
    loss2 = s.forward_loss(vid, vis, nv, tgt.reshape(K * B * n_vis, C))
    loss2.backward()
    assert abs(loss2.item() - loss.item()) <= 2e-4 * abs(loss.item())
    assert rel_l2(rt.fp.grad, g_generic) <= 1e-2
    # 3 optimizer steps through create_optimizer (param groups as the reference factory makes them)
    args = SimpleNamespace(opt="adamw", weight_decay=float(z["opt.wd"]), lr=float(z["opt.lr"]), opt_eps=float(z["opt.eps"]),
                           opt_betas=[float(b) for b in z["opt.betas"]])
    opt = create_optimizer(args, s, skip_list=s.no_weight_decay())
    names = {id(p): n for n, p in s.named_parameters()}
    got = {("decay" if g["weight_decay"] > 0 else "no_decay"): [names[id(p)] for p in g["params"]] for g in opt.param_groups}
    assert got["decay"] == list(z["groups.decay"]) and got["no_decay"] == list(z["groups.no_decay"])
    scaler = NativeScalerWithGradNormCount()
    for it in range(3):
        opt.zero_grad()
        l = s.forward_loss(vid, vis, nv, tgt.reshape(K * B * n_vis, C))
        gn = scaler(l, opt, clip_grad=None, parameters=s.parameters())
        assert abs(l.item() - z["out.losses3"][it]) <= 1e-3 * z["out.losses3"][it], (it, l.item(), z["out.losses3"][it])      # measured <= 6e-4
        assert abs(gn.item() - z["out.gnorms3"][it]) <= 1e-2 * z["out.gnorms3"][it], (it, gn.item(), z["out.gnorms3"][it])  # measured <= 1e-3
    sd = s.state_dict()
    lr = float(z["opt.lr"])
    for k in [f[len("after3."):] for f in z if f.startswith("after3.")]:
        a, r, g1 = sd[k].cpu(), torch.from_numpy(z["after3." + k]), torch.from_numpy(z["g." + k])
        d = (a - r).abs()
        # Adam normalises the step, so an element whose gradient is ~0 moves by up to lr per step in EITHER direction whatever the
        # rounding: only elements with a resolved gradient pin the update rule.  Where |g_ref| >= rms(g_ref) (the first step's
        # gradient; a third of the elements) the trajectory is held to 0.1 lr after three steps (measured <= 0.07 lr; a wrong
        # bias correction moves EVERY element by >= 0.5 lr at step 1), and the mean over all elements to 0.05 lr (measured 0.03).
        strong = g1.abs() >= g1.pow(2).mean().sqrt()
        assert strong.float().mean() > 0.15, k
        assert d[strong].max() <= 0.1 * lr, (k, (d[strong].max() / lr).item())
        assert d.mean() <= 0.05 * lr, (k, (d.mean() / lr).item())


def test_saved_gelu_derivative_equals_recomputed_tiny(golden_dir):
    """UNITE_GELU_DSAVE: the MLP's backward from the derivative the forward saved (16-bit fixed point) against the default path, which recomputes
    GELU' from the saved bf16 pre-activation -- same loss (the forward values are the same), gradients within 5e-3, and both within the golden bound."""
    z = _load(golden_dir, "student_tiny.npz")
    grads = []
    for dsave in (False, True):
        s, _ = build_tiny()
        s.load_state_dict(fill_state_dict(student_shapes(TINY_S), int(z["in.seed_weights"])))
        s = s.to(DEV).train()
        s.runtime().gelu_dsave = dsave
        out = s(torch.from_numpy(z["in.videos"]).to(DEV), torch.from_numpy(z["in.mask"]).to(DEV), clip_only=True)
        loss = (2 - 2 * (out * torch.from_numpy(z["out.targets"]).to(DEV)).sum(dim=-1)).mean()
        assert abs(loss.item() - float(z["out.loss"])) <= 1e-3 * abs(float(z["out.loss"]))
        loss.backward()
        for k, p in s.named_parameters():
            assert rel_l2(p.grad.cpu(), torch.from_numpy(z["g." + k])) <= 5e-2, (dsave, k)
        grads.append(s.runtime().fp.grad.clone())
    assert rel_l2(grads[1], grads[0]) <= 5e-3


def test_parameters_without_gradient_follow_torch_semantics():
    """torch semantics of `p.grad is None`: (a) blocks above the highest tap are not executed under clip_only
    (modeling_adaptation.py:165-166) -- their parameters must not be weight-decayed or moment-updated and must not enter the gradient
    norm; (b) frozen decoders (freeze_clip_decoders, run_stage1.py:586-590: requires_grad = False) still get gradient values written by
    the hand-scheduled backward, which must stay out of the norm and of the update.  Reference behaviour: torch.optim.AdamW skips
    p.grad is None; utils.get_grad_norm_ (:631-643) filters on it."""
    from functools import partial
    from unite_amd.modeling_adaptation import AdaptationVisionTransformer
    from unite_amd.optim_factory import create_optimizer
    from unite_amd.utils import NativeScalerWithGradNormCount
    cfg = O.StudentCfg(img_size=32, patch_size=16, embed_dim=128, depth=3, num_heads=2, num_frames=2, tubelet_size=1,
                       clip_decoder_embed_dim=128, clip_output_dim=64, clip_return_layers=(0, 1))
    s = AdaptationVisionTransformer(img_size=32, patch_size=16, encoder_embed_dim=128, encoder_depth=3, encoder_num_heads=2, mlp_ratio=4,
                                    qkv_bias=True, norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), num_frames=2, tubelet_size=1,
                                    clip_decoder_embed_dim=128, clip_output_dim=64, clip_return_layers=[0, 1])
    sd = fill_state_dict(student_shapes(cfg), 3)
    s.load_state_dict(sd)
    for n, p in s.named_parameters():
        if n.startswith("clip_decoder.1."):
            p.requires_grad = False                                   # (b) a frozen decoder
    s = s.to(DEV).train()
    vid = make_videos(2, 2, 32, 32, seed=5)
    mask = O.mask_from_importance(make_importance(4, 4, seed=6), 2, 2)
    tgt = torch.nn.functional.normalize(torch.randn(2, 2, 4, 64, generator=torch.Generator().manual_seed(7)), dim=-1)
    args = SimpleNamespace(opt="adamw", weight_decay=0.05, lr=1e-2, opt_eps=1e-8, opt_betas=[0.9, 0.95])
    opt = create_optimizer(args, s, skip_list=s.no_weight_decay())
    rt = s.runtime()
    vis, nv = rt.tokens_from_mask(mask.to(DEV))
    before = {k: v.detach().clone() for k, v in s.state_dict().items()}
    opt.zero_grad()
    loss = s.forward_loss(vid.to(DEV), vis, nv, tgt.reshape(-1, 64).to(DEV))
    gn = NativeScalerWithGradNormCount()(loss, opt, clip_grad=None, parameters=None)
    # oracle: gradients of the used, trainable parameters only
    leaf = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    out_ref = O.student_forward(leaf, vid, mask, cfg, clip_only=True)
    O.umt_loss(out_ref, tgt).backward()
    used = [k for k in leaf if leaf[k].grad is not None and not k.startswith("clip_decoder.1.")]
    assert not any(k.startswith("encoder.blocks.2.") for k in used) and any(k.startswith("encoder.blocks.1.") for k in used)
    gn_ref = torch.sqrt(sum((leaf[k].grad ** 2).sum() for k in used)).item()
    assert abs(gn.item() - gn_ref) <= 2e-2 * gn_ref, (gn.item(), gn_ref)
    after = s.state_dict()
    for k in before:
        moved = (after[k] - before[k]).abs().max().item()
        if k.startswith("encoder.blocks.2.") or k.startswith("clip_decoder.1."):
            assert moved == 0.0, (k, moved)                          # no gradient -> untouched (no weight decay, no moment update)
        else:
            assert moved > 0.0, k


def test_drop_path_and_accumulation_tiny(golden_dir):
    """stochastic depth with given keep-vectors vs the oracle; second backward without zero_grad accumulates."""
    z = _load(golden_dir, "student_tiny.npz")
    from functools import partial
    from unite_amd.modeling_adaptation import AdaptationVisionTransformer
    s = AdaptationVisionTransformer(img_size=32, patch_size=16, encoder_embed_dim=128, encoder_depth=3, encoder_num_heads=2,
                                    mlp_ratio=4, qkv_bias=True, norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), num_frames=2,
                                    tubelet_size=1, clip_decoder_embed_dim=128, clip_output_dim=64, clip_return_layers=[1, 2],
                                    drop_path_rate=0.5)
    sd = fill_state_dict(student_shapes(TINY_S), 3)
    s.load_state_dict(sd)
    s = s.to(DEV).train()
    vid = torch.from_numpy(z["in.videos"])
    mask = torch.from_numpy(z["in.mask"])
    tgt = torch.from_numpy(z["out.targets"])
    rt = s.runtime()
    u = torch.rand(3, 2, 2, generator=torch.Generator().manual_seed(5))
    rates = torch.linspace(0, 0.5, 3)
    keep = (1 - rates).view(-1, 1, 1)
    scales = torch.floor(keep + u) / keep
    rt.runner.drop_path_scales = lambda B, training: scales.to(DEV)
    leaf = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    out_ref = O.student_forward(leaf, vid, mask, TINY_S, clip_only=True, drop_path_rate=0.5, training=True, dp_rand=u)
    loss_ref = O.umt_loss(out_ref, tgt)
    loss_ref.backward()
    out = s(vid.to(DEV), mask.to(DEV), clip_only=True)
    loss = (2 - 2 * (out * tgt.to(DEV)).sum(dim=-1)).mean()
    assert abs(loss.item() - loss_ref.item()) <= 1e-3 * abs(loss_ref.item())
    loss.backward()
    for k, p in s.named_parameters():
        assert rel_l2(p.grad.cpu(), leaf[k].grad) <= 5e-2, k
    g1 = rt.fp.grad.clone()
    out = s(vid.to(DEV), mask.to(DEV), clip_only=True)
    ((2 - 2 * (out * tgt.to(DEV)).sum(dim=-1)).mean()).backward()       # no zero_grad: gradients add up
    assert rel_l2(rt.fp.grad, 2 * g1) <= 1e-3


def test_stage1_vitb_vs_reference_golden(golden_dir):
    """Full-size ViT-B/16 student + CLIP-B/16 teacher on 2 clips of 8x224x224: the engine's step (teacher -> mask from
    the stored permutation -> visible targets -> fused loss -> backward) against numbers produced by the reference."""
    import unite_amd
    from unite_amd.engine_stage1 import stage1_step, StepState
    z = _load(golden_dir, "stage1_vitb_cfg1.npz")
    B = int(z["in.B"])
    student = unite_amd.create_model("adaptation_umt_base_patch16_224", pretrained=False, drop_path_rate=0.0, num_frames=8,
                                     tubelet_size=1, clip_decoder_embed_dim=768, clip_output_dim=512,
                                     clip_return_layers=[6, 7, 8, 9, 10, 11], use_cls_token=False, use_learnable_pos_emb=False,
                                     use_checkpoint=False, checkpoint_num=0, clip_norm_type='l2', clip_student_return_interval=1,
                                     drop_block_rate=None)
    teacher = unite_amd.clip.clip_b16(pretrained=False, return_attn=True, clip_return_layers=[6, 7, 8, 9, 10, 11])
    assert sum(p.numel() for p in student.parameters()) == 88005888
    student.load_state_dict(fill_state_dict(student_shapes(O.StudentCfg()), int(z["in.seed_student"])))
    teacher.load_state_dict(fill_state_dict(teacher_shapes(O.TeacherCfg()), int(z["in.seed_teacher"])))
    student, teacher = student.to(DEV).train(), teacher.to(DEV)
    vid = make_videos(B, 8, 224, 224, int(z["in.seed_videos"])).to(DEV)
    imp = make_importance(B * 8, 196, int(z["in.seed_importance"])).to(DEV)
    st = StepState()
    loss = stage1_step(student, teacher, vid, B, float(z["in.mask_ratio"]), 'attention', None, 'mixed', st, importance=imp)
    # teacher side
    attn = teacher.runtime().ws.bufs["attn"]
    torch.testing.assert_close(attn.cpu(), torch.from_numpy(z["out.attn"]), atol=3e-4, rtol=5e-2)
    tg = teacher.runtime().ws.bufs["targets"].view(6, B, 320, 512)
    torch.testing.assert_close(tg[:, :, :8, :8].cpu(), torch.from_numpy(z["out.targets_corner"]), atol=6e-3, rtol=0)
    assert st.mask.view(B, -1).sum(1).tolist() == [1248] * B
    # loss within 1e-3 relative of the reference's fp32 CPU value
    ref_loss = float(z["out.loss"])
    assert abs(loss.item() - ref_loss) <= 1e-3 * abs(ref_loss), (loss.item(), ref_loss)
    loss.backward()
    rt = student.runtime()
    gn = rt.fp.grad.norm().item()
    assert abs(gn - float(z["out.grad_norm"])) <= 2e-2 * float(z["out.grad_norm"]), (gn, float(z["out.grad_norm"]))
    pg = dict(student.named_parameters())
    for f in z:
        if f.startswith("gnorm."):
            k = f[len("gnorm."):]
            g = pg[k].grad
            assert abs(g.norm().item() - float(z[f])) <= 5e-2 * float(z[f]) + 1e-7, (k, g.norm().item(), float(z[f]))
            corner = g.reshape(g.shape[0], -1)[:8, :8].cpu()
            ref = torch.from_numpy(z["gcorner." + k])
            assert (corner - ref).norm() <= 0.15 * ref.norm() + 1e-7, k


def test_stage1_loss_curve_vs_reference_golden(golden_dir):
    """north_star: "loss curves matching the CPU reference within 1e-3 relative".  24 optimisation steps of the full-size
    ViT-B/16 student + CLIP-B/16 teacher (B = 2, fresh clips and a stored mask permutation per step, cosine schedule with warm-up,
    AdamW through create_optimizer, drop_path 0) through the product engine pieces -- stage1_step, NativeScalerWithGradNormCount,
    the fused AdamW -- against the per-step loss / grad-norm THE REFERENCE's own classes produced on the CPU in fp32
    (oracle/make_golden_curve.py -> tests/golden/stage1_curve.npz; loss falls 1.99 -> 0.36).  Reference loop: run_stage1.py:294-505."""
    import unite_amd
    from unite_amd.engine_stage1 import stage1_step, StepState
    from unite_amd.optim_factory import create_optimizer
    from unite_amd.utils import NativeScalerWithGradNormCount, cosine_scheduler
    z = _load(golden_dir, "stage1_curve.npz")
    B, steps = int(z["in.B"]), int(z["in.steps"])
    student = unite_amd.create_model("adaptation_umt_base_patch16_224", pretrained=False, drop_path_rate=0.0, num_frames=8,
                                     tubelet_size=1, clip_decoder_embed_dim=768, clip_output_dim=512,
                                     clip_return_layers=[6, 7, 8, 9, 10, 11], use_cls_token=False, use_learnable_pos_emb=False,
                                     use_checkpoint=False, checkpoint_num=0, clip_norm_type='l2', clip_student_return_interval=1,
                                     drop_block_rate=None)
    teacher = unite_amd.clip.clip_b16(pretrained=False, return_attn=True, clip_return_layers=[6, 7, 8, 9, 10, 11])
    student.load_state_dict(fill_state_dict(student_shapes(O.StudentCfg()), int(z["in.seed_student"])))
    teacher.load_state_dict(fill_state_dict(teacher_shapes(O.TeacherCfg()), int(z["in.seed_teacher"])))
    student, teacher = student.to(DEV).train(), teacher.to(DEV)
    args = SimpleNamespace(opt="adamw", weight_decay=float(z["opt.wd"]), lr=float(z["opt.lr"]), opt_eps=float(z["opt.eps"]),
                           opt_betas=[float(b) for b in z["opt.betas"]])
    opt = create_optimizer(args, student, skip_list=student.no_weight_decay())
    lr = cosine_scheduler(float(z["opt.lr"]), float(z["opt.min_lr"]), 2, steps // 2, warmup_epochs=1, warmup_steps=int(z["opt.warmup_steps"]))
    np.testing.assert_allclose(lr, z["out.lr"], rtol=1e-12)
    scaler, st = NativeScalerWithGradNormCount(), StepState()
    losses, gnorms = [], []
    for it in range(steps):
        for g in opt.param_groups:                                  # run_stage1.py:326-338
            g["lr"] = lr[it] * g["lr_scale"]
        vid = make_videos(B, 8, 224, 224, int(z["in.seed_videos0"]) + it).to(DEV)
        imp = make_importance(B * 8, 196, int(z["in.seed_importance0"]) + it).to(DEV)
        loss = stage1_step(student, teacher, vid, B, float(z["in.mask_ratio"]), 'attention', None, 'mixed', st, importance=imp)
        opt.zero_grad()
        gn = scaler(loss, opt, clip_grad=None, parameters=None)
        losses.append(loss)
        gnorms.append(gn)
    losses = torch.stack([l.detach() for l in losses]).cpu().numpy()
    gnorms = torch.stack([g.detach() for g in gnorms]).cpu().numpy()
    rel = np.abs(losses - z["out.loss"]) / z["out.loss"]
    assert rel.max() <= 1e-3, (rel.max(), int(rel.argmax()), losses.tolist())          # every step of the curve, 1e-3 relative
    grel = np.abs(gnorms - z["out.grad_norm"]) / z["out.grad_norm"]
    assert grel.max() <= 2e-2, (grel.max(), int(grel.argmax()))
    sd = student.state_dict()
    for f in z:
        if f.startswith("after."):
            k = f[len("after."):]
            a, r = sd[k].reshape(sd[k].shape[0], -1)[:8, :8].cpu(), torch.from_numpy(z[f])
            # 24 Adam steps: an element moves by <= sum(lr) ~ 1.2e-2 in all; trajectories agree to a small fraction of that
            assert (a - r).abs().mean() <= 0.02 * float(z["out.lr"].sum()), (k, (a - r).abs().mean().item())


def test_stage1_vitl_cfg5_vs_oracle():
    """BASELINE config 5 at B = 1: ViT-L/16 student (16 frames -> 640 visible tokens, taps 18..23, decoders 1024 -> 768) with
    the CLIP-L/14 teacher at 196 x 196 (clips resized 224 -> 196), against the fp32 CPU oracle on the same seeded weights."""
    import unite_amd
    from unite_amd.engine_stage1 import stage1_step, StepState
    taps = [18, 19, 20, 21, 22, 23]
    scfg = O.StudentCfg(embed_dim=1024, depth=24, num_heads=16, num_frames=16, clip_decoder_embed_dim=1024, clip_output_dim=768,
                        clip_return_layers=tuple(taps))
    tcfg = O.TeacherCfg(input_resolution=196, patch_size=14, width=1024, layers=24, heads=16, output_dim=768, clip_return_layers=tuple(taps))
    student = unite_amd.create_model("adaptation_umt_large_patch16_224", pretrained=False, drop_path_rate=0.0, num_frames=16,
                                     tubelet_size=1, clip_decoder_embed_dim=1024, clip_output_dim=768, clip_return_layers=taps,
                                     use_cls_token=False, use_learnable_pos_emb=False, use_checkpoint=False, checkpoint_num=0,
                                     clip_norm_type='l2', clip_student_return_interval=1, drop_block_rate=None)
    teacher = unite_amd.clip.clip_l14(pretrained=False, input_resolution=196, return_attn=True, clip_return_layers=taps)
    ssd, tsd = fill_state_dict(student_shapes(scfg), 91), fill_state_dict(teacher_shapes(tcfg), 92)
    student.load_state_dict(ssd)
    teacher.load_state_dict(tsd)
    student, teacher = student.to(DEV).train(), teacher.to(DEV)
    B = 1
    vid = make_videos(B, 16, 224, 224, 93)
    imp = make_importance(B * 16, 196, 94)
    loss = stage1_step(student, teacher, vid.to(DEV), B, 0.8, 'attention', None, 'mixed', StepState(), clip_input_resolution=196,
                       importance=imp.to(DEV))
    loss.backward()
    mask = O.mask_from_importance(imp, 40, B)
    ssd_g = {k: v.clone().requires_grad_(True) for k, v in ssd.items()}
    ref, _, _, attn_ref = O.stage1_loss(ssd_g, tsd, vid, mask, scfg, tcfg)
    ref.backward()
    torch.testing.assert_close(teacher.runtime().ws.bufs["attn"].cpu(), attn_ref, atol=3e-4, rtol=5e-2)
    assert abs(loss.item() - ref.item()) <= 1e-3 * abs(ref.item()), (loss.item(), ref.item())
    gn_ref = torch.sqrt(sum((v.grad ** 2).sum() for v in ssd_g.values() if v.grad is not None)).item()
    gn = student.runtime().fp.grad.norm().item()
    assert abs(gn - gn_ref) <= 2e-2 * gn_ref, (gn, gn_ref)
    worst = max(rel_l2(p.grad.cpu(), ssd_g[k].grad) for k, p in student.named_parameters())
    assert worst <= 8e-2, worst          # 24 layers of bf16 operands; the per-tensor bound for ViT-B's 12 is 5e-2


@pytest.mark.parametrize("policy", [0, 2], ids=["tile-kernels", "persistent-kernel"])
def test_streams_on_off_bit_identical_full_size(policy):
    """BASELINE config-2 shapes at B = 8: the step with every side stream on (teacher frame ranges on three streams, overlapped target tail, weight
    gradients on their own stream) produces bit-identical gradients to the same step on one stream -- the arithmetic is the same
    and deterministic, so any difference would be a missing event / buffer-reuse race."""
    import unite_amd
    from unite_amd.engine_stage1 import stage1_step, StepState
    student = unite_amd.create_model("adaptation_umt_base_patch16_224", pretrained=False, drop_path_rate=0.0, num_frames=8,
                                     tubelet_size=1, clip_decoder_embed_dim=768, clip_output_dim=512,
                                     clip_return_layers=[6, 7, 8, 9, 10, 11], use_cls_token=False, use_learnable_pos_emb=False,
                                     use_checkpoint=False, checkpoint_num=0, clip_norm_type='l2', clip_student_return_interval=1,
                                     drop_block_rate=None)
    teacher = unite_amd.clip.clip_b16(pretrained=False, return_attn=True, clip_return_layers=[6, 7, 8, 9, 10, 11])
    student.load_state_dict(fill_state_dict(student_shapes(O.StudentCfg()), 3))
    teacher.load_state_dict(fill_state_dict(teacher_shapes(O.TeacherCfg()), 4))
    student, teacher = student.to(DEV).train(), teacher.to(DEV)
    B = 8
    vid = make_videos(B, 8, 224, 224, 5).to(DEV)
    imp = make_importance(B * 8, 196, 6).to(DEV)
    rt, trt = student.runtime(), teacher.runtime()
    trt.min_frames_per_stream = 16                      # 64 frames here: three frame ranges on three streams, as at B = 32
    grads, attns = [], []
    # The planner picks the GEMM kernel by tile count, and a frame range has a third of the rows: pin the kernel family (tile kernels only /
    # the persistent kernel wherever it applies), or the one-stream and the three-stream teacher would differ in the last bit for an
    # arithmetic reason (the persistent kernel adds the bias before the products, the tile kernels after them) and hide what this test is
    # looking for.
    from unite_amd import _lib
    lib = _lib.load()
    assert lib.unite_gemm_set_policy(policy) == 0
    try:
        for on in (False, True, True):                      # the concurrent form twice: steady-state buffer reuse included
            rt.runner.wgrad_stream, trt.two_streams = on, on
            st = StepState()
            st.overlap_targets = on
            rt.fp.accumulate = False
            loss = stage1_step(student, teacher, vid, B, 0.8, 'attention', None, 'mixed', st, importance=imp)
            loss.backward()
            torch.cuda.synchronize()
            grads.append(rt.fp.grad.clone())
            attns.append(trt.ws.bufs["attn"].clone())
            assert torch.isfinite(loss).item() and 1.0 < loss.item() < 2.5
    finally:
        assert lib.unite_gemm_set_policy(-1) == 0
    assert torch.equal(attns[0], attns[1]) and torch.equal(attns[1], attns[2])
    assert torch.equal(grads[0], grads[1]) and torch.equal(grads[1], grads[2])


def test_data_parallel_equivalence_full_size():
    """The property data-parallel training rests on, at BASELINE config-2 model sizes without the oracle: the step on 8 clips equals
    the mean of the steps on its two halves of 4 (loss) and the mean of their gradients (every clip is independent in teacher,
    masks and student; the loss is a mean over clips).  Tolerance: fp32 reductions in a different order + the split-K / tile plan
    of the GEMMs changing with M."""
    import unite_amd
    from unite_amd.engine_stage1 import stage1_step, StepState
    student = unite_amd.create_model("adaptation_umt_base_patch16_224", pretrained=False, drop_path_rate=0.0, num_frames=8,
                                     tubelet_size=1, clip_decoder_embed_dim=768, clip_output_dim=512,
                                     clip_return_layers=[6, 7, 8, 9, 10, 11], use_cls_token=False, use_learnable_pos_emb=False,
                                     use_checkpoint=False, checkpoint_num=0, clip_norm_type='l2', clip_student_return_interval=1,
                                     drop_block_rate=None)
    teacher = unite_amd.clip.clip_b16(pretrained=False, return_attn=True, clip_return_layers=[6, 7, 8, 9, 10, 11])
    student.load_state_dict(fill_state_dict(student_shapes(O.StudentCfg()), 13))
    teacher.load_state_dict(fill_state_dict(teacher_shapes(O.TeacherCfg()), 14))
    student, teacher = student.to(DEV).train(), teacher.to(DEV)
    B = 8
    vid = make_videos(B, 8, 224, 224, 15).to(DEV)
    imp = make_importance(B * 8, 196, 16).to(DEV)
    rt = student.runtime()

    def run(lo, hi):
        rt.fp.accumulate = False
        loss = stage1_step(student, teacher, vid[lo:hi].contiguous(), hi - lo, 0.8, 'attention', None, 'mixed', StepState(),
                           importance=imp[lo * 8:hi * 8].contiguous())
        loss.backward()
        torch.cuda.synchronize()
        return loss.item(), rt.fp.grad.clone()

    l_all, g_all = run(0, B)
    l_a, g_a = run(0, B // 2)
    l_b, g_b = run(B // 2, B)
    assert abs(l_all - 0.5 * (l_a + l_b)) <= 2e-5 * abs(l_all)
    g_mean = 0.5 * (g_a + g_b)
    assert rel_l2(g_all, g_mean) <= 2e-3


def test_stage1_step_with_more_than_256_patches_per_frame_vs_oracle():
    """The geometry of `clip_l14_336` as a teacher in miniature (reference clip.py:281-295: 24 x 24 = 576 patches per frame; here 18 x 18 = 324 at
    288 / 16 for both models): frames of more than 256 positions go through the 1024-thread form of the mask kernels, the teacher's 325-token
    sequences and the student's 2 x 81 visible tokens through the tiled attention kernels.  Loss and every gradient against the oracle."""
    from functools import partial
    from unite_amd.clip import VisionTransformer
    from unite_amd.modeling_adaptation import AdaptationVisionTransformer
    from unite_amd.engine_stage1 import stage1_step, StepState
    scfg = O.StudentCfg(img_size=288, patch_size=16, embed_dim=128, depth=2, num_heads=2, num_frames=2, tubelet_size=1,
                        clip_decoder_embed_dim=128, clip_output_dim=64, clip_return_layers=(0, 1))
    tcfg = O.TeacherCfg(input_resolution=288, patch_size=16, width=128, layers=2, heads=2, output_dim=64, clip_return_layers=(0, 1))
    s = AdaptationVisionTransformer(img_size=288, patch_size=16, encoder_embed_dim=128, encoder_depth=2, encoder_num_heads=2, mlp_ratio=4,
                                    qkv_bias=True, norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), num_frames=2, tubelet_size=1,
                                    clip_decoder_embed_dim=128, clip_output_dim=64, clip_return_layers=[0, 1])
    t = VisionTransformer(input_resolution=288, patch_size=16, width=128, layers=2, heads=2, output_dim=64, return_attn=True,
                          clip_return_layers=[0, 1])
    ssd, tsd = fill_state_dict(student_shapes(scfg), 71), fill_state_dict(teacher_shapes(tcfg), 72)
    s.load_state_dict(ssd)
    t.load_state_dict(tsd)
    s, t = s.to(DEV).train(), t.to(DEV).eval()
    B, N = 2, 324
    vid = make_videos(B, 2, 288, 288, seed=73)
    imp = make_importance(B * 2, N, seed=74)
    st = StepState()
    loss = stage1_step(s, t, vid.to(DEV), B, 0.75, 'attention', None, 'mixed', st, clip_input_resolution=288, importance=imp.to(DEV))
    mask = O.mask_from_importance(imp, N - int(N * 0.75), B)
    assert torch.equal(st.mask.view(B, -1).cpu().bool(), mask)
    ssd_g = {k: v.clone().requires_grad_(True) for k, v in ssd.items()}
    ref, _, _, _ = O.stage1_loss(ssd_g, tsd, vid, mask, scfg, tcfg)
    assert abs(loss.item() - ref.item()) <= 1e-3 * abs(ref.item()), (loss.item(), ref.item())
    loss.backward()
    ref.backward()
    for k, p in s.named_parameters():
        assert rel_l2(p.grad.cpu(), ssd_g[k].grad) <= 5e-2, k
    # the attention-guided sampler itself at this frame size: exact counts per frame from the teacher's own CLS attention
    st2 = StepState()
    stage1_step(s, t, vid.to(DEV), B, 0.75, 'attention', None, 'mixed', st2, clip_input_resolution=288)
    assert (~st2.mask.view(B * 2, N).cpu().bool()).sum(1).tolist() == [N - int(N * 0.75)] * (B * 2)


@pytest.mark.parametrize("which", ["source", "target"])
def test_stage1_loss_data_slices_vs_oracle(which):
    """clip_loss_data = 'source' / 'target' (run_stage1.py:418-427: the distillation loss over the first n_source clips of the mixed
    batch, or over the rest) with an explicit tube mask handed in as the dataset would (bool_masked_pos, mask_type != 'attention'):
    loss and every gradient against the oracle evaluated on that slice of the batch."""
    from unite_amd.engine_stage1 import stage1_step, StepState
    s, t = build_tiny()
    ssd, tsd = fill_state_dict(student_shapes(TINY_S), 41), fill_state_dict(teacher_shapes(TINY_T), 42)
    s.load_state_dict(ssd)
    t.load_state_dict(tsd)
    s, t = s.to(DEV).train(), t.to(DEV)
    B, T, N = 3, TINY_S.num_frames, (TINY_S.img_size // TINY_S.patch_size) ** 2
    vid = make_videos(B, T, TINY_S.img_size, TINY_S.img_size, seed=43)
    # tube masks: one spatial pattern per clip, repeated over the frames, half of the patches visible
    g = torch.Generator().manual_seed(44)
    frame_mask = torch.stack([torch.randperm(N, generator=g) >= N // 2 for _ in range(B)])          # (B, N) True = masked
    mask = frame_mask[:, None, :].expand(B, T, N).reshape(B, T * N).contiguous()
    n_source = 2
    lo, hi = (0, n_source) if which == "source" else (n_source, B)
    rt = s.runtime()
    rt.fp.accumulate = False
    loss = stage1_step(s, t, vid.to(DEV), n_source, 0.5, 'tube', mask.to(DEV), which, StepState(), clip_input_resolution=TINY_T.input_resolution)
    loss.backward()
    ssd_g = {k: v.clone().requires_grad_(True) for k, v in ssd.items()}
    ref, *_ = O.stage1_loss(ssd_g, tsd, vid[lo:hi], mask[lo:hi], TINY_S, TINY_T)
    ref.backward()
    assert abs(loss.item() - ref.item()) <= 1e-3 * abs(ref.item())
    for k, p in s.named_parameters():
        assert rel_l2(p.grad.cpu(), ssd_g[k].grad) <= 5e-2, k


def test_train_one_epoch_synthetic():
    """The drop-in engine on a synthetic loader: attention-guided masks, 6 steps, loss goes down, meters come back."""
    import unite_amd
    from unite_amd.engine_stage1 import train_one_epoch
    from unite_amd.optim_factory import create_optimizer
    from unite_amd.utils import NativeScalerWithGradNormCount, cosine_scheduler
    s, t = build_tiny()
    torch.manual_seed(0)
    s.load_state_dict(fill_state_dict(student_shapes(TINY_S), 3))
    t.load_state_dict(fill_state_dict(teacher_shapes(TINY_T), 1))
    s, t = s.to(DEV), t.to(DEV)
    args = SimpleNamespace(opt="adamw", weight_decay=0.05, lr=2e-3, opt_eps=1e-8, opt_betas=[0.9, 0.95], log_freq=2, epochs=1,
                           clip_loss_data="mixed", seed=0)
    opt = create_optimizer(args, s, skip_list=s.no_weight_decay())
    vid = make_videos(4, 2, 32, 32, seed=9)
    loader = [(vid, torch.full((4,), -1), torch.zeros(4, dtype=torch.long)) for _ in range(6)]
    lr = cosine_scheduler(2e-3, 1e-5, 1, 6)
    stats = train_one_epoch(s, loader, None, opt, torch.device(DEV), 0, NativeScalerWithGradNormCount(), max_norm=None,
                            start_steps=0, lr_schedule_values=lr, wd_schedule_values=None, teacher_model=t,
                            clip_input_resolution=32, clip_loss_type='l2', mask_type='attention', mask_ratio=0.5, args=args)
    assert set(stats) >= {"loss", "loss_clip", "grad_norm", "lr", "min_lr", "weight_decay", "loss_scale"}
    assert np.isfinite(stats["loss"]) and 0.0 < stats["loss"] < 4.0
    m = s._unite_stage1_state.mask.view(8, 4)
    assert (m.sum(1) == 2).all()                    # N_vis = 4 - int(4*0.5) = 2 visible per frame
    assert stats["loss"] < 2.3


def test_graphed_step_equals_eager_step():
    """graph_step.GraphedStage1Step (the whole stage-1 step captured once in a HIP graph, per-step scalars -- lr table, Adam bias corrections,
    mask and stochastic-depth seeds -- in device memory) against the eager step on the same seeds, schedule and clips: per-step loss and
    gradient norm and the parameters after six steps are bit-identical (same kernels, same order on every stream; the loss value itself to 2e-6:
    its reduction uses float atomics)."""
    from functools import partial
    from unite_amd.engine_stage1 import stage1_step, StepState
    from unite_amd.graph_step import GraphedStage1Step
    from unite_amd.modeling_adaptation import AdaptationVisionTransformer
    from unite_amd.optim_factory import create_optimizer
    from unite_amd.utils import NativeScalerWithGradNormCount, cosine_scheduler
    B, steps = 4, 6
    vids = [make_videos(B, 2, 32, 32, seed=50 + i).to(DEV) for i in range(steps)]
    lr = cosine_scheduler(2e-3, 1e-5, 1, steps)

    def build():
        torch.manual_seed(123)
        s = AdaptationVisionTransformer(img_size=32, patch_size=16, encoder_embed_dim=128, encoder_depth=3, encoder_num_heads=2, mlp_ratio=4,
                                        qkv_bias=True, norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), num_frames=2, tubelet_size=1,
                                        clip_decoder_embed_dim=128, clip_output_dim=64, clip_return_layers=[1, 2], drop_path_rate=0.2)
        _, t = build_tiny()
        s.load_state_dict(fill_state_dict(student_shapes(TINY_S), 3))
        t.load_state_dict(fill_state_dict(teacher_shapes(TINY_T), 1))
        s, t = s.to(DEV).train(), t.to(DEV)
        args = SimpleNamespace(opt="adamw", weight_decay=0.05, lr=2e-3, opt_eps=1e-8, opt_betas=[0.9, 0.95])
        opt = create_optimizer(args, s, skip_list=s.no_weight_decay())
        st = StepState()
        st.seed = 77
        return s, t, opt, NativeScalerWithGradNormCount(), st

    def run(graph: bool):
        s, t, opt, scaler, st = build()
        g = GraphedStage1Step(s, t, opt, scaler, tuple(vids[0].shape), 0.5, clip_grad=None, clip_input_resolution=32, state=st, warmup=2) if graph else None
        out = []
        for i in range(steps):
            for grp in opt.param_groups:
                grp["lr"] = lr[i] * grp["lr_scale"]
            if g is not None:
                loss, gn = g(vids[i])
            else:
                loss = stage1_step(s, t, vids[i], B, 0.5, 'attention', None, 'mixed', st, clip_input_resolution=32)
                opt.zero_grad()
                gn = scaler(loss, opt, clip_grad=None, parameters=None)
            out.append(torch.stack([loss.detach(), gn.detach()]).clone())
        torch.cuda.synchronize()
        if g is not None:
            assert g.graph is not None and g.calls == steps          # steps 0-1 eager, step 2 captured (+ replayed), steps 3-5 replays
        return torch.stack(out).cpu(), s.runtime().fp.param.clone().cpu()

    eager, p_eager = run(False)
    graph, p_graph = run(True)
    assert torch.isfinite(eager).all() and (eager[:, 0] > 0).all()
    # gradient norms and parameters: bit-identical.  The loss VALUE is a sum of per-workgroup float atomics (decoder_tail_fwd), whose order is
    # not fixed from launch to launch -- last-bit differences there do not reach the gradients
    assert torch.equal(eager[:, 1], graph[:, 1]), (eager, graph)
    torch.testing.assert_close(eager[:, 0], graph[:, 0], rtol=2e-6, atol=0)
    assert torch.equal(p_eager, p_graph)
    assert eager[-1, 0] < eager[0, 0]                                   # and it trains


def test_teacher_ahead_equals_sequential_step():
    """engine_stage1.TeacherAhead (the frozen teacher's whole phase of batch i+1 on its own stream while the student trains on batch i,
    two output slots) against stage1_step batch by batch, on different clips per step: per-step gradient norms and the parameters after six
    steps are bit-identical (the teacher does not depend on the student: run_stage1.py:371 no_grad, frozen weights), the loss to 2e-6 (float
    atomics in its reduction)."""
    from functools import partial
    from unite_amd.engine_stage1 import stage1_step, student_phase, StepState, TeacherAhead
    from unite_amd.modeling_adaptation import AdaptationVisionTransformer
    from unite_amd.optim_factory import create_optimizer
    from unite_amd.utils import NativeScalerWithGradNormCount, cosine_scheduler
    B, steps = 4, 6
    vids = [make_videos(B, 2, 32, 32, seed=150 + i).to(DEV) for i in range(steps)]
    lr = cosine_scheduler(2e-3, 1e-5, 1, steps)

    def run(ahead_on: bool):
        torch.manual_seed(123)
        s = AdaptationVisionTransformer(img_size=32, patch_size=16, encoder_embed_dim=128, encoder_depth=3, encoder_num_heads=2, mlp_ratio=4,
                                        qkv_bias=True, norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), num_frames=2, tubelet_size=1,
                                        clip_decoder_embed_dim=128, clip_output_dim=64, clip_return_layers=[1, 2], drop_path_rate=0.2)
        _, t = build_tiny()
        s.load_state_dict(fill_state_dict(student_shapes(TINY_S), 3))
        t.load_state_dict(fill_state_dict(teacher_shapes(TINY_T), 1))
        s, t = s.to(DEV).train(), t.to(DEV)
        args = SimpleNamespace(opt="adamw", weight_decay=0.05, lr=2e-3, opt_eps=1e-8, opt_betas=[0.9, 0.95])
        opt = create_optimizer(args, s, skip_list=s.no_weight_decay())
        scaler, st = NativeScalerWithGradNormCount(), StepState()
        st.seed = 77
        ahead = TeacherAhead(t, st, DEV, 0.5, 'attention', clip_input_resolution=32) if ahead_on else None
        out, masks = [], []
        nxt = ahead.launch(vids[0]) if ahead_on else None
        for i in range(steps):
            for grp in opt.param_groups:
                grp["lr"] = lr[i] * grp["lr_scale"]
            if ahead_on:
                cur = nxt
                nxt = ahead.launch(vids[i + 1]) if i + 1 < steps else None
                loss = student_phase(s, vids[i], cur, B, 'mixed')
                m = cur.mask
            else:
                loss = stage1_step(s, t, vids[i], B, 0.5, 'attention', None, 'mixed', st, clip_input_resolution=32)
                m = st.mask
            opt.zero_grad()
            gn = scaler(loss, opt, clip_grad=None, parameters=None)
            out.append(torch.stack([loss.detach(), gn.detach()]).clone())
            torch.cuda.current_stream().synchronize()                  # the slot's mask is overwritten two launches later
            masks.append(m.clone().cpu())
        torch.cuda.synchronize()
        if ahead is not None:
            ahead.close()
        return torch.stack(out).cpu(), s.runtime().fp.param.clone().cpu(), torch.stack(masks)

    seq, p_seq, m_seq = run(False)
    ahd, p_ahd, m_ahd = run(True)
    assert torch.isfinite(seq).all() and (seq[:, 0] > 0).all()
    assert torch.equal(m_seq, m_ahd)
    assert torch.equal(seq[:, 1], ahd[:, 1]), (seq, ahd)
    torch.testing.assert_close(seq[:, 0], ahd[:, 0], rtol=2e-6, atol=0)
    assert torch.equal(p_seq, p_ahd)


def _busy(ms_target=25.0):
    """~ms_target of matrix products on the CURRENT stream: whatever is enqueued behind them starts that much later"""
    a = torch.randn(4096, 4096, device=DEV, dtype=torch.bfloat16)
    b = torch.empty_like(a)
    for _ in range(max(1, int(ms_target / 0.25))):          # a 4096^3 bf16 product takes ~0.15-0.3 ms
        torch.mm(a, a, out=b)
    return b


def test_teacher_ahead_device_batch_produced_right_before_launch():
    """Round-2 advisor finding: TeacherAhead.launch ordered the teacher's stream only behind the mark of an EARLIER launch, so a device batch
    produced on the student's stream right before the call (here: behind ~25 ms of other work, into a freshly recycled allocator block)
    was read by the teacher before its producer had run -- masks and targets of another clip.  With the default ``inputs_ready=None`` the
    teacher's stream waits for the caller's stream position: masks, gradient norms and parameters equal the sequential step's."""
    from functools import partial
    from unite_amd.engine_stage1 import stage1_step, student_phase, StepState, TeacherAhead
    from unite_amd.optim_factory import create_optimizer
    from unite_amd.utils import NativeScalerWithGradNormCount, cosine_scheduler
    B, steps = 4, 5
    host = [make_videos(B, 2, 32, 32, seed=650 + i) for i in range(steps)]
    lr = cosine_scheduler(2e-3, 1e-5, 1, steps)

    def run(ahead_on: bool):
        s, t = build_tiny()
        s.load_state_dict(fill_state_dict(student_shapes(TINY_S), 3))
        t.load_state_dict(fill_state_dict(teacher_shapes(TINY_T), 1))
        s, t = s.to(DEV).train(), t.to(DEV)
        args = SimpleNamespace(opt="adamw", weight_decay=0.05, lr=2e-3, opt_eps=1e-8, opt_betas=[0.9, 0.95])
        opt = create_optimizer(args, s, skip_list=s.no_weight_decay())
        scaler, st = NativeScalerWithGradNormCount(), StepState()
        st.seed = 31
        ahead = TeacherAhead(t, st, DEV, 0.5, 'attention', clip_input_resolution=32) if ahead_on else None
        staged = [h.to(DEV) for h in host]
        torch.cuda.synchronize()

        def produce(i):
            _busy()                                   # the producer below sits behind this on the student's stream
            return staged[i] * 1.0                    # a NEW device tensor (recycled block), written by a kernel enqueued just now
        out, masks = [], []
        vid_next = produce(0)
        nxt = ahead.launch(vid_next) if ahead_on else None
        for i in range(steps):
            for grp in opt.param_groups:
                grp["lr"] = lr[i] * grp["lr_scale"]
            vid = vid_next
            if ahead_on:
                cur = nxt
                if i + 1 < steps:
                    vid_next = produce(i + 1)
                    nxt = ahead.launch(vid_next)
                loss = student_phase(s, vid, cur, B, 'mixed')
                m = cur.mask
            else:
                loss = stage1_step(s, t, vid, B, 0.5, 'attention', None, 'mixed', st, clip_input_resolution=32)
                m = st.mask
                if i + 1 < steps:
                    vid_next = produce(i + 1)
            opt.zero_grad()
            gn = scaler(loss, opt, clip_grad=None, parameters=None)
            out.append(torch.stack([loss.detach(), gn.detach()]).clone())
            torch.cuda.current_stream().synchronize()
            masks.append(m.clone().cpu())
        torch.cuda.synchronize()
        return torch.stack(out).cpu(), s.runtime().fp.param.clone().cpu(), torch.stack(masks)

    seq, p_seq, m_seq = run(False)
    ahd, p_ahd, m_ahd = run(True)
    assert torch.equal(m_seq, m_ahd)
    assert torch.equal(seq[:, 1], ahd[:, 1]), (seq, ahd)
    assert torch.equal(p_seq, p_ahd)


def test_train_one_epoch_teacher_ahead_with_a_device_side_loader():
    """A loader that PRODUCES its batch on the device at next() (synthetic clips, a device-side transform): train_one_epoch fetches it with the
    teacher's stream current, so the producer is ordered in front of the teacher phase and the student reaches the clip behind
    TeacherOut.ready.  Equal to the sequential order, bit for bit."""
    from unite_amd.engine_stage1 import train_one_epoch
    from unite_amd.optim_factory import create_optimizer
    from unite_amd.utils import NativeScalerWithGradNormCount, cosine_scheduler
    steps = 5
    host = [make_videos(2, 2, 32, 32, seed=700 + i) for i in range(steps)]
    lr = cosine_scheduler(2e-3, 1e-5, 1, steps)

    class DeviceLoader:
        def __init__(self):
            self.staged = [h.to(DEV) for h in host]
            torch.cuda.synchronize()

        def __len__(self):
            return steps

        def __iter__(self):
            for i in range(steps):
                _busy(8.0)                            # on whatever stream is current when the engine asks for the batch
                yield self.staged[i] * 1.0, torch.zeros(2, 8, dtype=torch.bool), torch.zeros(2, dtype=torch.long)

    def run(ahead):
        s, t = build_tiny()
        s.load_state_dict(fill_state_dict(student_shapes(TINY_S), 3))
        t.load_state_dict(fill_state_dict(teacher_shapes(TINY_T), 1))
        s, t = s.to(DEV), t.to(DEV)
        args = SimpleNamespace(opt="adamw", weight_decay=0.05, lr=2e-3, opt_eps=1e-8, opt_betas=[0.9, 0.95], log_freq=0, epochs=1,
                               clip_loss_data="mixed", seed=5, teacher_ahead=ahead)
        opt = create_optimizer(args, s, skip_list=s.no_weight_decay())
        stats = train_one_epoch(s, DeviceLoader(), None, opt, torch.device(DEV), 0, NativeScalerWithGradNormCount(), max_norm=None,
                                start_steps=0, lr_schedule_values=lr, wd_schedule_values=None, teacher_model=t, clip_input_resolution=32,
                                clip_loss_type='l2', mask_type='attention', mask_ratio=0.5, args=args)
        torch.cuda.synchronize()
        return stats, s.runtime().fp.param.clone().cpu()

    st_a, p_a = run(True)
    st_s, p_s = run(False)
    assert torch.equal(p_a, p_s)
    assert st_a["grad_norm"] == st_s["grad_norm"]


def test_teacher_residual_stream_types_error_statistics():
    """What the DEFAULT residual-stream type of the frozen teacher (IEEE half) costs against the reference's fp32 rows, where 16 values of one fixture
    cannot tell (test_teacher_tiny_vs_reference_golden): 12 seeded towers x 64 CLS-attention values each against the fp32 CPU oracle
    (tools/teacher_stream_error.py; 24 towers in profiles/r04_teacher_stream_error.txt: 1.43e-3 / 1.45e-3 / 1.74e-3 rms for f16 / f32 / bf16 rows).
    The f16 rows must be as accurate as the f32 rows (rms within 5 %, mean feature cosine within 2e-6), and better than the bf16 rows."""
    from tests.teacher_stream_stats import collect
    st = collect(12)
    rms = {k: v[0].pow(2).mean().sqrt().item() for k, v in st.items()}
    cos = {k: v[2].mean().item() for k, v in st.items()}
    assert rms["f32"] <= 2.0e-3                                   # values average 1 / 16: 3 % -- the bf16 operands of the score product
    assert rms["f16"] <= 1.05 * rms["f32"], rms
    assert rms["bf16"] >= 1.08 * rms["f16"], rms                  # the stream type that never became the default is measurably worse
    assert abs(cos["f16"] - cos["f32"]) <= 2e-6 and cos["f16"] >= 0.9999, cos


def test_teacher_bf16_residual_stream_vs_f32_stream():
    """UNITE_TEACHER_RES16=1 (opt-in: its attention error is outside the golden test's bound): the frozen teacher's residual
    stream, taps included, kept in bf16.  Against the f32 stream of the same weights and clips: CLS attention within 2e-2 absolute (five keys
    per frame here, probabilities ~0.2), every target row's cosine > 0.999."""
    _, t16 = build_tiny()
    _, t32 = build_tiny()
    sd = fill_state_dict(teacher_shapes(TINY_T), 1)
    t16.load_state_dict(sd)
    t32.load_state_dict(sd)
    t16, t32 = t16.to(DEV), t32.to(DEV)
    t16.runtime().res16, t32.runtime().res16 = True, False
    vid = make_videos(4, 2, 32, 32, seed=5).to(DEV)
    f16, a16 = t16(vid)
    f32, a32 = t32(vid)
    torch.testing.assert_close(a16, a32, atol=2e-2, rtol=0)
    cos = (f16 * f32).sum(-1)
    assert float(cos.min()) > 0.999, float(cos.min())
    assert not torch.equal(a16, a32)                      # the switch does change the arithmetic


def test_teacher_f16_residual_stream_vs_f32_stream():
    """The default residual stream of the frozen teacher, taps included: IEEE half (what OpenAI's CLIP itself runs with).  Against the f32 stream
    (UNITE_TEACHER_RES16=0) of the same weights and clips: CLS attention within 6e-3 absolute (measured 4.0e-3 -- the size of either arm's own
    error against the golden vectors: a perturbation of 2^-12 is enough to re-draw the bf16 roundings of everything downstream; the bf16 stream
    needs 2e-2), every target row's cosine > 0.9999."""
    _, t16 = build_tiny()
    _, t32 = build_tiny()
    sd = fill_state_dict(teacher_shapes(TINY_T), 1)
    t16.load_state_dict(sd)
    t32.load_state_dict(sd)
    t16, t32 = t16.to(DEV), t32.to(DEV)
    t16.runtime().res16, t32.runtime().res16 = "f16", False
    vid = make_videos(4, 2, 32, 32, seed=5).to(DEV)
    f16, a16 = t16(vid)
    f32, a32 = t32(vid)
    torch.testing.assert_close(a16, a32, atol=6e-3, rtol=0)
    cos = (f16 * f32).sum(-1)
    assert float(cos.min()) > 0.9999, float(cos.min())
    assert not torch.equal(a16, a32)                      # the switch does change the arithmetic
    assert t16.runtime().ws.bufs[t16.runtime().ws.prefix + "x.a"].dtype == torch.float16


@pytest.mark.parametrize("mask_type", ["attention", "tube"])
def test_train_one_epoch_teacher_ahead_equals_sequential(mask_type):
    """train_one_epoch with its default schedule (teacher one batch ahead: the loader is read one batch early, host batches are copied on the
    teacher's stream, source + target loaders concatenated) against args.teacher_ahead=False (run_stage1.py's order) on the same loaders of
    host tensors: same meters, bit-identical parameters after the epoch."""
    from unite_amd.engine_stage1 import train_one_epoch
    from unite_amd.optim_factory import create_optimizer
    from unite_amd.utils import NativeScalerWithGradNormCount, cosine_scheduler
    steps = 5
    g = torch.Generator().manual_seed(9)

    def tube(B):          # loader-side masks (mask_type 'tube': run_stage1.py:344-358): one spatial pattern per clip, half of the 4 patches visible
        fm = torch.stack([torch.randperm(4, generator=g) >= 2 for _ in range(B)])
        return fm[:, None, :].expand(B, 2, 4).reshape(B, 8).contiguous()
    src = [(make_videos(2, 2, 32, 32, seed=300 + i), tube(2), torch.zeros(2, dtype=torch.long)) for i in range(steps)]
    tgt = [(make_videos(2, 2, 32, 32, seed=400 + i), tube(2), torch.zeros(2, dtype=torch.long)) for i in range(2)]   # shorter: re-iterated
    lr = cosine_scheduler(2e-3, 1e-5, 1, steps)

    def run(ahead):
        s, t = build_tiny()
        s.load_state_dict(fill_state_dict(student_shapes(TINY_S), 3))
        t.load_state_dict(fill_state_dict(teacher_shapes(TINY_T), 1))
        s, t = s.to(DEV), t.to(DEV)
        args = SimpleNamespace(opt="adamw", weight_decay=0.05, lr=2e-3, opt_eps=1e-8, opt_betas=[0.9, 0.95], log_freq=1, epochs=1,
                               clip_loss_data="mixed", seed=5, teacher_ahead=ahead)
        opt = create_optimizer(args, s, skip_list=s.no_weight_decay())
        stats = train_one_epoch(s, src, tgt, opt, torch.device(DEV), 0, NativeScalerWithGradNormCount(), max_norm=None, start_steps=0,
                                lr_schedule_values=lr, wd_schedule_values=None, teacher_model=t, clip_input_resolution=32,
                                clip_loss_type='l2', mask_type=mask_type, mask_ratio=0.5, args=args)
        torch.cuda.synchronize()
        return stats, s.runtime().fp.param.clone().cpu()

    st_a, p_a = run(True)
    st_s, p_s = run(False)
    assert torch.equal(p_a, p_s)
    assert st_a["grad_norm"] == st_s["grad_norm"]
    assert abs(st_a["loss"] - st_s["loss"]) <= 2e-6 * abs(st_s["loss"])


def test_decoders_on_side_stream_bit_identical():
    """ViTRunner.side_decoders (UNITE_DECODER_STREAM=1: the student's decoders -- projection, fused normalise + loss, their backward -- on a side
    stream beside the encoder blocks; off by default) gives bit-identical gradients to the one-stream order, twice in a row (buffer reuse)."""
    from unite_amd.engine_stage1 import stage1_step, StepState
    vid = make_videos(4, 2, 32, 32, seed=77).to(DEV)
    imp = make_importance(8, 4, seed=78).to(DEV)
    grads = []
    for on in (False, True):
        s, t = build_tiny()
        s.load_state_dict(fill_state_dict(student_shapes(TINY_S), 3))
        t.load_state_dict(fill_state_dict(teacher_shapes(TINY_T), 1))
        s, t = s.to(DEV).train(), t.to(DEV)
        rt = s.runtime()
        rt.runner.side_decoders = on
        for _ in range(2):
            rt.fp.accumulate = False
            loss = stage1_step(s, t, vid, 4, 0.5, 'attention', None, 'mixed', StepState(), clip_input_resolution=32, importance=imp)
            loss.backward()
            torch.cuda.synchronize()
        grads.append(rt.fp.grad.clone())
        assert torch.isfinite(loss).item()
    assert torch.equal(grads[0], grads[1])


@pytest.mark.parametrize("kind", ["mse", "smooth_l1", "l1"])
def test_stage1_other_clip_loss_types_vs_oracle(kind):
    """clip_loss_type 'mse' / 'smooth_l1' / 'l1' (run_stage1.py:403-408,433-434; no shipped config selects them): loss and every gradient of
    the tiny student against the oracle's nn.*Loss on the same decoder outputs and targets."""
    from unite_amd.engine_stage1 import stage1_step, StepState
    s, t = build_tiny()
    ssd, tsd = fill_state_dict(student_shapes(TINY_S), 61), fill_state_dict(teacher_shapes(TINY_T), 62)
    s.load_state_dict(ssd)
    t.load_state_dict(tsd)
    s, t = s.to(DEV).train(), t.to(DEV).eval()
    B = 3
    vid = make_videos(B, 2, 32, 32, seed=63)
    imp = make_importance(B * 2, 4, seed=64)
    s.runtime().fp.accumulate = False
    loss = stage1_step(s, t, vid.to(DEV), B, 0.5, 'attention', None, 'mixed', StepState(), clip_input_resolution=32, importance=imp.to(DEV),
                       clip_loss_type=kind)
    mask = O.mask_from_importance(imp, 2, B)
    ssd_g = {k: v.clone().requires_grad_(True) for k, v in ssd.items()}
    ref, _, _, _ = O.stage1_loss(ssd_g, tsd, vid, mask, TINY_S, TINY_T, clip_loss_type=kind)
    assert abs(loss.item() - ref.item()) <= 2e-3 * abs(ref.item())
    loss.backward()
    ref.backward()
    # l1: the gradient is sign(out - target), so wherever bf16 rounding moves an output across its target the two paths differ by a full
    # +-1/n in that element -- the per-tensor error is inherently larger than for the smooth losses (measured 0.13 on the patch embedding)
    tol = 0.25 if kind == "l1" else 5e-2
    for k, p in s.named_parameters():
        assert rel_l2(p.grad.cpu(), ssd_g[k].grad) <= tol, k


def test_stage1_parity_at_the_benchmark_batch(golden_dir):
    """BASELINE configs[1] AS bench.py TIMES IT -- B = 32, the teacher launched on its own stream (TeacherAhead), every GEMM of both phases
    planned for a shared GPU (the stage-1 planner weight, TeacherAhead.DEFAULT_SHARING = 0.9: other tiles, other split-K factors than at B = 2) -- against sixteen B = 2 steps of the
    sequential engine on the same clips and stored mask permutations (drop_path 0): the B = 32 loss is the mean of the sixteen losses to
    2e-5 relative and the gradient the mean of the sixteen gradients to 2e-3 relative L2 (clips are independent, the loss is a mean over
    clips); the first B = 2 step is the golden step of tests/golden/stage1_vitb_cfg1.npz and is held to the REFERENCE's loss (1e-3) here
    as well, which ties the B = 32 launches to the reference."""
    import unite_amd
    from unite_amd.engine_stage1 import stage1_step, student_phase, StepState, TeacherAhead
    z = _load(golden_dir, "stage1_vitb_cfg1.npz")
    student = unite_amd.create_model("adaptation_umt_base_patch16_224", pretrained=False, drop_path_rate=0.0, num_frames=8,
                                     tubelet_size=1, clip_decoder_embed_dim=768, clip_output_dim=512,
                                     clip_return_layers=[6, 7, 8, 9, 10, 11], use_cls_token=False, use_learnable_pos_emb=False,
                                     use_checkpoint=False, checkpoint_num=0, clip_norm_type='l2', clip_student_return_interval=1,
                                     drop_block_rate=None)
    teacher = unite_amd.clip.clip_b16(pretrained=False, return_attn=True, clip_return_layers=[6, 7, 8, 9, 10, 11])
    student.load_state_dict(fill_state_dict(student_shapes(O.StudentCfg()), int(z["in.seed_student"])))
    teacher.load_state_dict(fill_state_dict(teacher_shapes(O.TeacherCfg()), int(z["in.seed_teacher"])))
    student, teacher = student.to(DEV).train(), teacher.to(DEV)
    NB, B = 16, 32
    vid = torch.cat([make_videos(2, 8, 224, 224, int(z["in.seed_videos"]) + 100 * j) for j in range(NB)]).to(DEV)
    imp = torch.cat([make_importance(2 * 8, 196, int(z["in.seed_importance"]) + 100 * j) for j in range(NB)]).to(DEV)
    rt = student.runtime()
    torch.cuda.synchronize()
    # the benchmark's launch form
    st = StepState()
    ahead = TeacherAhead(teacher, st, DEV, float(z["in.mask_ratio"]), 'attention')
    tout = ahead.launch(vid, importance=imp, inputs_ready=False)
    rt.fp.accumulate = False
    with ahead.student():
        loss32 = student_phase(student, vid, tout, B, 'mixed')
        loss32.backward()
    torch.cuda.synchronize()
    l32, g32 = loss32.item(), rt.fp.grad.clone()
    assert st.mask.view(B, -1).sum(1).tolist() == [1248] * B
    # sixteen sequential B = 2 steps
    ls, gsum = [], torch.zeros_like(g32)
    for j in range(NB):
        rt.fp.accumulate = False
        loss = stage1_step(student, teacher, vid[2 * j:2 * j + 2].contiguous(), 2, float(z["in.mask_ratio"]), 'attention', None, 'mixed',
                           StepState(), importance=imp[16 * j:16 * j + 16].contiguous())
        loss.backward()
        torch.cuda.synchronize()
        ls.append(loss.item())
        gsum += rt.fp.grad
    ref_loss = float(z["out.loss"])
    assert abs(ls[0] - ref_loss) <= 1e-3 * abs(ref_loss), (ls[0], ref_loss)          # sub-batch 0 IS the golden step
    mean_l = sum(ls) / NB
    assert abs(l32 - mean_l) <= 2e-5 * abs(mean_l), (l32, mean_l)
    assert rel_l2(g32, gsum / NB) <= 2e-3


def test_stage1_vitl_parity_at_batch_8():
    """BASELINE config 5 (ViT-L/16 student, 16 frames -> 640 visible tokens, CLIP-L/14 teacher at 196) at B = 8 with the teacher ahead and
    the shared-GPU planner weight, against four B = 2 steps of the sequential engine: loss = mean of the losses (2e-5), gradient = mean of
    the gradients (2e-3 relative L2).  The B = 1 step of this configuration is tied to the oracle by test_stage1_vitl_cfg5_vs_oracle."""
    import unite_amd
    from unite_amd.engine_stage1 import stage1_step, student_phase, StepState, TeacherAhead
    taps = [18, 19, 20, 21, 22, 23]
    scfg = O.StudentCfg(embed_dim=1024, depth=24, num_heads=16, num_frames=16, clip_decoder_embed_dim=1024, clip_output_dim=768,
                        clip_return_layers=tuple(taps))
    tcfg = O.TeacherCfg(input_resolution=196, patch_size=14, width=1024, layers=24, heads=16, output_dim=768, clip_return_layers=tuple(taps))
    student = unite_amd.create_model("adaptation_umt_large_patch16_224", pretrained=False, drop_path_rate=0.0, num_frames=16,
                                     tubelet_size=1, clip_decoder_embed_dim=1024, clip_output_dim=768, clip_return_layers=taps,
                                     use_cls_token=False, use_learnable_pos_emb=False, use_checkpoint=False, checkpoint_num=0,
                                     clip_norm_type='l2', clip_student_return_interval=1, drop_block_rate=None)
    teacher = unite_amd.clip.clip_l14(pretrained=False, input_resolution=196, return_attn=True, clip_return_layers=taps)
    student.load_state_dict(fill_state_dict(student_shapes(scfg), 91))
    teacher.load_state_dict(fill_state_dict(teacher_shapes(tcfg), 92))
    student, teacher = student.to(DEV).train(), teacher.to(DEV)
    B = 8
    vid = make_videos(B, 16, 224, 224, 95).to(DEV)
    imp = make_importance(B * 16, 196, 96).to(DEV)
    rt = student.runtime()
    torch.cuda.synchronize()
    st = StepState()
    ahead = TeacherAhead(teacher, st, DEV, 0.8, 'attention', clip_input_resolution=196)
    tout = ahead.launch(vid, importance=imp, inputs_ready=False)
    rt.fp.accumulate = False
    with ahead.student():
        loss8 = student_phase(student, vid, tout, B, 'mixed')
        loss8.backward()
    torch.cuda.synchronize()
    l8, g8 = loss8.item(), rt.fp.grad.clone()
    ls, gsum = [], torch.zeros_like(g8)
    for j in range(B // 2):
        rt.fp.accumulate = False
        loss = stage1_step(student, teacher, vid[2 * j:2 * j + 2].contiguous(), 2, 0.8, 'attention', None, 'mixed', StepState(),
                           clip_input_resolution=196, importance=imp[32 * j:32 * j + 32].contiguous())
        loss.backward()
        torch.cuda.synchronize()
        ls.append(loss.item())
        gsum += rt.fp.grad
    mean_l = sum(ls) / len(ls)
    assert abs(l8 - mean_l) <= 2e-5 * abs(mean_l), (l8, mean_l)
    assert rel_l2(g8, gsum / len(ls)) <= 2e-3
