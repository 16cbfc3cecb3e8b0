"""The CPU oracle (oracle/umt_oracle.py) against vectors produced by the reference itself
(oracle/make_golden.py -> tests/golden/*.npz).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import umt_oracle as O
from oracle.filler import fill_state_dict, make_importance, make_videos

TOL = 1e-5   # same fp32 ops, different reduction order


def _load(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name))
    return {k: z[k] for k in z.files}


def _sd(z, prefix="w."):
    return {k[len(prefix):]: torch.from_numpy(v) for k, v in z.items() if k.startswith(prefix)}


from tests.shapes import TINY_S, TINY_T, TINY_V, student_shapes, teacher_shapes, vit_shapes  # noqa: E402


def _tiny_teacher_sd(z):
    return fill_state_dict(teacher_shapes(TINY_T), int(z["in.seed_weights"]))


def _tiny_student_sd(z):
    return fill_state_dict(student_shapes(TINY_S), int(z["in.seed_weights"]))


def test_teacher_tiny(golden_dir):
    z = _load(golden_dir, "teacher_tiny.npz")
    feats, attn = O.teacher_forward(_tiny_teacher_sd(z), torch.from_numpy(z["in.videos"]), TINY_T)
    np.testing.assert_allclose(feats.numpy(), z["out.feats"], atol=TOL, rtol=0)
    np.testing.assert_allclose(attn.numpy(), z["out.attn"], atol=TOL, rtol=0)
    assert (attn.sum(-1) < 1).all()          # CLS column removed (clip.py:183)


def test_student_tiny_forward_backward(golden_dir):
    z = _load(golden_dir, "student_tiny.npz")
    tz = _load(golden_dir, "teacher_tiny.npz")
    sd = {k: v.clone().requires_grad_(True) for k, v in _tiny_student_sd(z).items()}
    vid = torch.from_numpy(z["in.videos"])
    imp = torch.from_numpy(z["in.importance"])
    n_vis = O.n_visible(4, float(z["in.mask_ratio"]))
    mask = O.mask_from_importance(imp, n_vis, vid.shape[0])
    assert np.array_equal(mask.numpy(), z["in.mask"])
    loss, out, tgt, _ = O.stage1_loss(sd, _tiny_teacher_sd(tz), vid, mask, TINY_S, TINY_T)
    np.testing.assert_allclose(tgt.numpy(), z["out.targets"], atol=TOL, rtol=0)
    np.testing.assert_allclose(out.detach().numpy(), z["out.x_clip"], atol=TOL, rtol=0)
    np.testing.assert_allclose(loss.item(), z["out.loss"], atol=TOL, rtol=0)
    loss.backward()
    for k, p in sd.items():
        np.testing.assert_allclose(p.grad.numpy(), z["g." + k], atol=2e-5, rtol=1e-4, err_msg=k)
    with torch.no_grad():
        x_vis, x_clip = O.student_forward(sd, vid, mask, TINY_S, clip_only=False)
    np.testing.assert_allclose(x_vis.numpy(), z["out.x_vis"], atol=TOL, rtol=0)


def test_student_tiny_three_adamw_steps(golden_dir):
    """param grouping (optim_factory.py:76-118) + torch.optim.AdamW through the reference's factory."""
    z = _load(golden_dir, "student_tiny.npz")
    tsd = _tiny_teacher_sd(_load(golden_dir, "teacher_tiny.npz"))
    sd = {k: v.clone() for k, v in _tiny_student_sd(z).items()}
    groups = O.parameter_group_names([(k, tuple(v.shape)) for k, v in sd.items()], 0.05,
                                     skip_list={'pos_embed', 'cls_token', 'mask_token', 'clip_mask_token', 'clip_pos_embed'})
    assert groups["decay"]["params"] == list(z["groups.decay"])
    assert groups["no_decay"]["params"] == list(z["groups.no_decay"])
    vid = torch.from_numpy(z["in.videos"])
    mask = torch.from_numpy(z["in.mask"])
    m = {k: torch.zeros_like(v) for k, v in sd.items()}
    v2 = {k: torch.zeros_like(v) for k, v in sd.items()}
    lr, eps = float(z["opt.lr"]), float(z["opt.eps"])
    b1, b2 = [float(b) for b in z["opt.betas"]]
    for step in range(1, 4):
        leaf = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        loss, *_ = O.stage1_loss(leaf, tsd, vid, mask, TINY_S, TINY_T)
        loss.backward()
        np.testing.assert_allclose(loss.item(), z["out.losses3"][step - 1], atol=2e-5, rtol=0)
        gn = O.grad_norm([p.grad for p in leaf.values()])
        np.testing.assert_allclose(gn.item(), z["out.gnorms3"][step - 1], rtol=1e-4)
        for gname, g in groups.items():
            for k in g["params"]:
                O.adamw_step(sd[k], leaf[k].grad, m[k], v2[k], step, lr, b1, b2, eps, g["weight_decay"])
    # Adam divides by sqrt(v): an element whose gradient is ~0 moves by up to lr per step whatever the
    # rounding of g, so the bound is a fraction of 3*lr, not the fp32 ulp.
    checked = [k[len("after3."):] for k in z if k.startswith("after3.")]
    assert len(checked) == 8
    for k in checked:
        np.testing.assert_allclose(sd[k].numpy(), z["after3." + k], atol=2e-4, rtol=1e-4, err_msg=k)
        assert np.mean(np.abs(sd[k].numpy() - z["after3." + k])) < 2e-6, k


def test_vit_stage2_tiny(golden_dir):
    z = _load(golden_dir, "vit_stage2_tiny.npz")
    cfg = TINY_V
    sd = {k: v.requires_grad_(True) for k, v in fill_state_dict(vit_shapes(cfg), int(z["in.seed_weights"])).items()}
    logits = O.vit_classifier_forward(sd, torch.from_numpy(z["in.videos"]), cfg)
    np.testing.assert_allclose(logits.detach().numpy(), z["out.logits"], atol=TOL, rtol=0)
    loss = torch.nn.functional.cross_entropy(logits, torch.from_numpy(z["in.labels"]))
    np.testing.assert_allclose(loss.item(), z["out.loss"], atol=TOL, rtol=0)
    loss.backward()
    for k, p in sd.items():
        np.testing.assert_allclose(p.grad.numpy(), z["g." + k], atol=2e-5, rtol=1e-4, err_msg=k)


def test_utils(golden_dir):
    z = _load(golden_dir, "utils.npz")
    np.testing.assert_allclose(O.cosine_scheduler(1.5e-4, 1e-5, 4, 5, warmup_epochs=1), z["cos.a"], rtol=1e-12)
    np.testing.assert_allclose(O.cosine_scheduler(1.5e-4, 1e-5, 3, 7, warmup_epochs=1, warmup_steps=4,
                                                  start_warmup_value=1e-6), z["cos.b"], rtol=1e-12)
    attn = torch.from_numpy(z["greedy.attn"])
    assert np.array_equal(O.get_greedy_masks(attn, 0.75, 2).numpy(), z["greedy.k2_r075"])
    g3 = O.get_greedy_masks(attn, 0.8, 3)
    assert np.array_equal(g3.numpy(), z["greedy.k3_r08"])
    assert ((~g3).sum(0) <= 1).all()          # committee members are disjoint
    # layer-decay grouping on stage-2 names
    nl = 2
    scales = [0.65 ** (nl + 1 - i) for i in range(nl + 2)]
    for name, sc in zip(z["layer.names"], z["layer.scale"]):
        assert scales[O.get_num_layer_for_vit(str(name), len(scales))] == pytest.approx(float(sc))


def test_stage1_vitb_cfg1(golden_dir):
    """Full-size ViT-B/16 student + CLIP-B/16 teacher, 2 clips of 8x224x224 (BASELINE configs[0] shape)."""
    z = _load(golden_dir, "stage1_vitb_cfg1.npz")
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    scfg, tcfg = O.StudentCfg(), O.TeacherCfg()
    ssd = {k: v.requires_grad_(True) for k, v in fill_state_dict(student_shapes(scfg), int(z["in.seed_student"])).items()}
    tsd = fill_state_dict(teacher_shapes(tcfg), int(z["in.seed_teacher"]))
    B = int(z["in.B"])
    vid = make_videos(B, 8, 224, 224, int(z["in.seed_videos"]))
    imp = make_importance(B * 8, 196, int(z["in.seed_importance"]))
    mask = O.mask_from_importance(imp, O.n_visible(196, float(z["in.mask_ratio"])), B)
    assert mask.sum(1).tolist() == [1248] * B
    loss, out, tgt, attn = O.stage1_loss(ssd, tsd, vid, mask, scfg, tcfg)
    np.testing.assert_allclose(attn.numpy(), z["out.attn"], atol=TOL, rtol=0)
    np.testing.assert_allclose(tgt[:, :, :8, :8].numpy(), z["out.targets_corner"], atol=TOL, rtol=0)
    np.testing.assert_allclose(out[:, :, :8, :8].detach().numpy(), z["out.x_clip_corner"], atol=5e-5, rtol=0)
    np.testing.assert_allclose(loss.item(), z["out.loss"], atol=2e-5, rtol=0)
    loss.backward()
    gn = O.grad_norm([p.grad for p in ssd.values()])
    np.testing.assert_allclose(gn.item(), z["out.grad_norm"], rtol=1e-4)
    for k in [f[len("gnorm."):] for f in z if f.startswith("gnorm.")]:
        np.testing.assert_allclose(ssd[k].grad.norm().item(), z["gnorm." + k], rtol=2e-4, err_msg=k)
        g = ssd[k].grad.reshape(ssd[k].shape[0], -1)[:8, :8]
        np.testing.assert_allclose(g.numpy(), z["gcorner." + k], atol=1e-6, rtol=2e-3, err_msg=k)


def test_stage1_curve_first_steps(golden_dir):
    """The oracle's full-size stage-1 training loop (oracle step + adamw_step + grad_norm + cosine_scheduler, the bench's CPU
    baseline) against the first steps of the loss curve the REFERENCE produced (oracle/make_golden_curve.py: its own model
    classes, create_optimizer, get_grad_norm_, cosine_scheduler).  Bounded to 3 steps of B = 2 to keep the CPU suite short."""
    z = _load(golden_dir, "stage1_curve.npz")
    scfg, tcfg = O.StudentCfg(), O.TeacherCfg()
    ssd = fill_state_dict(student_shapes(scfg), int(z["in.seed_student"]))
    tsd = fill_state_dict(teacher_shapes(tcfg), int(z["in.seed_teacher"]))
    B, steps = int(z["in.B"]), int(z["in.steps"])
    lr = O.cosine_scheduler(float(z["opt.lr"]), float(z["opt.min_lr"]), 2, steps // 2, warmup_epochs=1, warmup_steps=int(z["opt.warmup_steps"]))
    np.testing.assert_allclose(lr, z["out.lr"], rtol=1e-12)
    m = {k: torch.zeros_like(v) for k, v in ssd.items()}
    v = {k: torch.zeros_like(v) for k, v in ssd.items()}
    b1, b2 = (float(b) for b in z["opt.betas"])
    torch.set_num_threads(8)
    for it in range(3):
        vid = make_videos(B, 8, 224, 224, int(z["in.seed_videos0"]) + it)
        mask = O.mask_from_importance(make_importance(B * 8, 196, int(z["in.seed_importance0"]) + it), 40, B)
        leaf = {k: p.requires_grad_(True) for k, p in ssd.items()}
        loss, *_ = O.stage1_loss(leaf, tsd, vid, mask, scfg, tcfg)
        loss.backward()
        gn = O.grad_norm([p.grad for p in leaf.values()])
        np.testing.assert_allclose(loss.item(), z["out.loss"][it], rtol=2e-5)
        np.testing.assert_allclose(gn.item(), z["out.grad_norm"][it], rtol=2e-4)
        with torch.no_grad():
            for k, p in leaf.items():
                p.requires_grad_(False)
                # no-decay group: 1-D tensors, .bias, and no_weight_decay() names (optim_factory.py:76-118)
                O.adamw_step(p, p.grad, m[k], v[k], it + 1, float(lr[it]), b1, b2, float(z["opt.eps"]), float(z["opt.wd"]) if p.ndim > 1 else 0.0)
                p.grad = None


def test_stage3_step_vs_reference_golden(golden_dir):
    """oracle.stage3_loss / stage3_pseudo_labels / clip_similarity against the reference's OWN stage-3 train_one_epoch (run_stage3.py:340-710,
    executed from its syntax tree by oracle/make_golden_stage3.py: tests/golden/stage3_step.npz) for all seven selection strategies: the
    classifier logits of the three student passes, the committee masks, the zero-shot similarities, source / target / total loss, the
    select ratio, every per-tensor gradient norm and eight gradient tensors."""
    z = _load(golden_dir, "stage3_step.npz")
    ssd0 = fill_state_dict(student_shapes(TINY_S), int(z["in.seed_student"]))
    tsd = fill_state_dict(teacher_shapes(TINY_T), int(z["in.seed_teacher"]))
    t = {k[3:]: torch.from_numpy(v) for k, v in z.items() if k.startswith("in.") and v.dtype.kind in "fi" and v.ndim > 0}
    B_t = t["videos_t"].shape[0]
    sims = O.clip_similarity(t["img_feats"], t["text_feats"], B_t)
    _, attn = O.teacher_forward(tsd, t["videos_t_aug"], TINY_T, return_attn=True)
    masks = O.get_greedy_masks(attn, float(z["in.mask_ratio"]), 2)
    for strat in [str(s) for s in z["in.strategies"]]:
        pre = strat + "."
        assert np.array_equal(masks.numpy(), z[pre + "masks"])
        if pre + "similarities" in z:
            np.testing.assert_allclose(sims.numpy(), z[pre + "similarities"], atol=2e-6, rtol=0)
        ssd = {k: v.clone().requires_grad_(True) for k, v in ssd0.items()}
        loss, loss_s, loss_t, sel = O.stage3_loss(ssd, tsd, t["cls_w"], t["cls_b"], t["videos_s"], t["labels_s"], t["videos_t"], t["videos_t_aug"],
                                                  t["labels_t"], TINY_S, TINY_T, float(z["in.mask_ratio"]), strat, clip_probs_t=sims,
                                                  clip_threshold=float(z["in.clip_threshold"]))
        np.testing.assert_allclose(float(loss_s), z[pre + "loss_s"], atol=TOL, rtol=0, err_msg=strat)
        np.testing.assert_allclose(float(loss_t), z[pre + "loss_t"], atol=TOL, rtol=0, err_msg=strat)
        np.testing.assert_allclose(float(loss), z[pre + "loss"], atol=TOL, rtol=0, err_msg=strat)
        assert float(sel.float().mean()) == float(z[pre + "select_ratio"]), strat
        loss.backward()
        gn = O.grad_norm([p.grad for k, p in ssd.items() if p.grad is not None])
        np.testing.assert_allclose(float(gn), z[pre + "grad_norm"], rtol=2e-5, err_msg=strat)
        for k, p in ssd.items():
            if k.startswith("clip_decoder."):
                assert p.grad is None or float(p.grad.abs().max()) == 0.0          # run_stage3.py:475 discards the decoder outputs
                continue
            np.testing.assert_allclose(float(p.grad.norm()), z[pre + "gnorm." + k], rtol=1e-4, atol=1e-6, err_msg=k)
            if pre + "g." + k in z:
                np.testing.assert_allclose(p.grad.numpy(), z[pre + "g." + k], atol=5e-5, rtol=1e-4, err_msg=k)
