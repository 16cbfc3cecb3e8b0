"""Stage hand-off loaders (unite_amd/checkpoint.py) against the behaviour of run_stage1.py:518-602, run_stage2.py:349-438 and
run_stage3.py:829-924 (read as text: the drivers cannot be imported here -- wandb / decord / src.knn).  CPU only."""
import os
from functools import partial
from types import SimpleNamespace

import pytest
import torch

from unite_amd import checkpoint as C
from unite_amd import utils


def _student():
    from unite_amd.modeling_adaptation import AdaptationVisionTransformer
    return AdaptationVisionTransformer(img_size=32, patch_size=16, encoder_embed_dim=128, encoder_depth=2, encoder_num_heads=2,
                                       mlp_ratio=4, qkv_bias=True, norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), num_frames=2,
                                       tubelet_size=1, clip_decoder_embed_dim=128, clip_output_dim=64, clip_return_layers=[1])


def _vit():
    from unite_amd.modeling_finetune import VisionTransformer
    return VisionTransformer(img_size=32, patch_size=16, embed_dim=128, depth=2, num_heads=2, mlp_ratio=4, qkv_bias=True,
                             norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), num_classes=5, all_frames=2, tubelet_size=1,
                             use_mean_pooling=True, init_scale=0.001)


def _randomise(m, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(torch.randn(p.shape, generator=g))
    return m


def test_load_state_dict_prefix_and_reports():
    lin = torch.nn.Sequential(torch.nn.Linear(3, 2), torch.nn.Linear(2, 2))
    sd = {"pre.0.weight": torch.ones(2, 3), "pre.0.bias": torch.zeros(2), "pre.9.weight": torch.ones(1)}
    missing, unexpected = utils.load_state_dict(lin, sd, prefix="pre.")
    assert torch.equal(lin[0].weight, torch.ones(2, 3))
    assert sorted(missing) == ["pre.1.bias", "pre.1.weight"] and unexpected == ["pre.9.weight"]


def test_stage1_to_stage2_to_stage3_handoff(tmp_path):
    # "UMT K710" style encoder-only file -> stage-1 student: every key gains 'encoder.' (run_stage1.py:526)
    enc = _randomise(_vit(), 1)
    pre = {k: v for k, v in enc.state_dict().items() if not k.startswith(("head.", "fc_norm."))}
    pre["norm.weight"], pre["norm.bias"] = torch.full((128,), 2.0), torch.full((128,), 3.0)
    f0 = tmp_path / "umt.pth"
    torch.save({"model": {"backbone." + k if k.startswith("blocks.0.") else k: v for k, v in pre.items()}}, f0)
    dec = _randomise(_student(), 2)
    fdec = tmp_path / "dec.pth"
    torch.save({k: v for k, v in dec.state_dict().items()}, fdec)
    student = _student()
    args = SimpleNamespace(student_init=str(f0), model_key="model|module", student_prefix="", clip_decoder_init=str(fdec),
                           freeze_clip_decoders=True, num_frames=2)
    C.load_student_from_ckpt(args, student)
    sd = student.state_dict()
    # 'encoder.' + key; note the reference adds the prefix BEFORE stripping 'backbone.', so 'backbone.x' becomes 'encoder.backbone.x'
    # and is not found: those parameters keep their initial values (reported as missing) -- same here
    assert torch.equal(sd["encoder.blocks.1.attn.qkv.weight"], pre["blocks.1.attn.qkv.weight"])
    assert not torch.equal(sd["encoder.blocks.0.attn.qkv.weight"], pre["blocks.0.attn.qkv.weight"])
    assert torch.equal(sd["encoder.norm.weight"], torch.full((128,), 2.0))
    assert torch.equal(sd["clip_decoder.0.head.weight"], dec.state_dict()["clip_decoder.0.head.weight"])
    assert all(not p.requires_grad for n, p in student.named_parameters() if n.startswith("clip_decoder."))
    assert all(p.requires_grad for n, p in student.named_parameters() if n.startswith("encoder."))

    # stage 1 -> stage 2: 'encoder.' stripped, decoders / encoder.norm unused, head missing (run_stage2.py:384-393)
    f1 = tmp_path / "stage1.pth"
    torch.save({"model": student.state_dict(), "epoch": 3}, f1)
    vit = _vit()
    head0 = vit.head.weight.detach().clone()
    C.load_from_ckpt(SimpleNamespace(finetune=str(f1), model_key="model|module", model_prefix="", delete_head=True, nb_classes=5,
                                     num_frames=2), vit)
    assert torch.equal(vit.state_dict()["blocks.1.mlp.fc2.weight"], sd["encoder.blocks.1.mlp.fc2.weight"])
    assert torch.equal(vit.head.weight, head0)

    # stage 2 -> stage 3: a classifier file has no 'encoder.' prefix -> it is added (run_stage3.py:843-846); a stage-1 file is taken as is
    f2 = tmp_path / "stage2.pth"
    torch.save({"model": _randomise(_vit(), 7).state_dict()}, f2)
    s3 = _student()
    a3 = SimpleNamespace(student_init=str(f2), model_key="model|module", student_prefix="", clip_decoder_init=None,
                         freeze_clip_decoders=False, num_frames=2)
    C.load_student_from_ckpt_stage3(a3, s3)
    assert torch.equal(s3.state_dict()["encoder.blocks.0.norm1.weight"], torch.load(f2, weights_only=True)["model"]["blocks.0.norm1.weight"])
    s3b = _student()
    a3.student_init = str(f1)
    C.load_student_from_ckpt_stage3(a3, s3b)
    assert torch.equal(s3b.state_dict()["encoder.blocks.1.attn.proj.weight"], sd["encoder.blocks.1.attn.proj.weight"])
    assert torch.equal(s3b.state_dict()["clip_decoder.0.norm.weight"], sd["clip_decoder.0.norm.weight"])


def test_head_slicing_for_k710_checkpoints(tmp_path):
    from unite_amd.modeling_finetune import VisionTransformer
    m = VisionTransformer(img_size=32, patch_size=16, embed_dim=128, depth=1, num_heads=2, mlp_ratio=4, qkv_bias=True,
                          norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), num_classes=400, all_frames=2, tubelet_size=1)
    sd = m.state_dict()
    sd["head.weight"], sd["head.bias"] = torch.arange(710.0)[:, None].expand(710, 128).clone(), torch.arange(710.0)
    f = tmp_path / "k710.pth"
    torch.save({"module": sd}, f)
    C.load_from_ckpt(SimpleNamespace(finetune=str(f), model_key="model|module", model_prefix="", delete_head=False, nb_classes=400,
                                     num_frames=2), m)
    assert torch.equal(m.head.bias.detach(), torch.arange(400.0))                       # run_stage2.py:370-372


def test_interpolate_pos_embed_time_then_space():
    D = 4
    model = SimpleNamespace(pos_embed=torch.zeros(1, 1 + 16 * 16, D), patch_embed=SimpleNamespace(num_patches=16 * 16, tubelet_size=1))
    # checkpoint: 8 frames of a 2 x 2 grid (+ 1 class token); target: 16 frames of 4 x 4
    t = torch.arange(8.0).view(8, 1, 1).expand(8, 4, D)                                   # value = frame index
    ck = {"pos_embed": torch.cat([torch.full((1, 1, D), -1.0), t.reshape(1, 32, D)], dim=1)}
    # the reference keeps the class token only in the spatial step; its temporal step views ALL tokens as (t, hw): give it none here
    ck["pos_embed"] = ck["pos_embed"][:, 1:]
    model.pos_embed = torch.zeros(1, 16 * 16, D)
    out = C.interpolate_pos_embed(ck, model, num_frames=16)["pos_embed"]
    assert out.shape == (1, 16 * 16, D)
    frames = out.view(16, 16, D)
    assert torch.allclose(frames[:, 0, 0], frames[:, 5, 2])                               # constant over space stays constant
    ref = torch.nn.functional.interpolate(torch.arange(8.0).view(1, 1, 8), size=16, mode="linear").view(16)
    assert torch.allclose(frames[:, 0, 0], ref, atol=1e-6)                                # half-pixel linear in time (align_corners=False)
    # equal sizes (8 frames of 4 x 4 -> 8 frames of 4 x 4): the table is left alone
    same = {"pos_embed": torch.randn(1, 8 * 16, D)}
    keep = same["pos_embed"].clone()
    m2 = SimpleNamespace(pos_embed=torch.zeros(1, 8 * 16, D), patch_embed=SimpleNamespace(num_patches=8 * 16, tubelet_size=1))
    assert torch.equal(C.interpolate_pos_embed(same, m2, num_frames=8)["pos_embed"], keep)


def test_clip_weight_inflation_and_pos_resize_vs_reference_golden(golden_dir):
    """unite_amd.clip.inflate_weight / load_state_dict against THE REFERENCE's own functions (src/models/clip.py:191-231, imported as is by
    oracle/make_golden_ckpt.py): 2-D conv1 weights inflated along time (center / mean) and the position table bicubic-resized from a
    3 x 3 to a 2 x 2 grid; every other tensor loads unchanged.  Tolerance: bit-exact for the inflation, 1e-6 for the bicubic resize."""
    import os
    import numpy as np
    import torch
    from unite_amd import clip
    z = np.load(os.path.join(golden_dir, "clip_ckpt.npz"))
    w2d = torch.from_numpy(z["in.w2d"])
    assert torch.equal(clip.inflate_weight(w2d, 3, center=True), torch.from_numpy(z["out.inflate_center_t3"]))
    assert torch.equal(clip.inflate_weight(w2d, 2, center=False), torch.from_numpy(z["out.inflate_mean_t2"]))
    for tag, center in (("c", True), ("m", False)):
        model = clip.VisionTransformer(input_resolution=32, patch_size=16, width=64, layers=2, heads=1, output_dim=32, kernel_size=1,
                                       return_attn=True, clip_return_layers=[1])
        sd = {k: v.clone() for k, v in model.state_dict().items()}
        for k in ("conv1.weight", "positional_embedding", "proj", "ln_pre.weight"):
            sd[k] = torch.from_numpy(z[f"in.{tag}.{k}"])
        clip.load_state_dict(model, sd, input_resolution=32, patch_size=16, center=center)
        got = model.state_dict()
        assert torch.equal(got["conv1.weight"], torch.from_numpy(z[f"out.{tag}.conv1.weight"]))
        torch.testing.assert_close(got["positional_embedding"], torch.from_numpy(z[f"out.{tag}.positional_embedding"]), atol=1e-6, rtol=0)
        assert torch.equal(got["proj"], torch.from_numpy(z[f"out.{tag}.proj"]))
        assert torch.equal(got["ln_pre.weight"], torch.from_numpy(z[f"in.{tag}.ln_pre.weight"]))


class _Payload:
    pass


def test_read_checkpoint_refuses_pickles_without_opt_in(tmp_path, monkeypatch):
    """third-party checkpoint files are read with the tensor-only loader; an argparse.Namespace (what the reference saves under 'args')
    is allow-listed, anything else needs UNITE_UNSAFE_CHECKPOINT_LOAD=1."""
    import argparse
    import torch
    from unite_amd.checkpoint import read_checkpoint

    ok = tmp_path / "ok.pth"
    torch.save({"model": {"w": torch.ones(2)}, "args": argparse.Namespace(lr=1e-3), "epoch": 3}, ok)
    ck = read_checkpoint(str(ok))
    assert ck["epoch"] == 3 and ck["args"].lr == 1e-3
    bad = tmp_path / "bad.pth"
    torch.save({"model": {"w": torch.ones(2)}, "extra": _Payload()}, bad)
    monkeypatch.delenv("UNITE_UNSAFE_CHECKPOINT_LOAD", raising=False)
    with pytest.raises(RuntimeError, match="UNITE_UNSAFE_CHECKPOINT_LOAD"):
        read_checkpoint(str(bad))


def test_stage2_hand_off_vs_reference_golden(golden_dir, tmp_path):
    """checkpoint.load_from_ckpt against the reference's OWN load_from_ckpt (run_stage2.py:349-438 executed from its syntax tree by
    oracle/make_golden_posembed.py): model-key selection, the 710 -> 400 head rows / --delete_head, 'encoder.' / 'backbone.' prefix stripping
    and the position table -- linear in time then bicubic in space, class token kept -- on three small checkpoints."""
    import json
    from collections import OrderedDict
    from types import SimpleNamespace
    import numpy as np
    from unite_amd import checkpoint, utils
    z = np.load(os.path.join(golden_dir, "stage2_ckpt.npz"))
    for tag in ("a", "b", "c"):
        c = json.loads(str(z[f"{tag}.in.cfg"]))
        sd = OrderedDict((k[len(tag) + 4:], torch.from_numpy(z[k])) for k in z.files if k.startswith(f"{tag}.in.") and not k.endswith(".cfg"))
        path = str(tmp_path / f"{tag}.pth")
        torch.save({c["key"]: sd} if c["key"] else sd, path)
        n_new = c["t_new"] * c["s_new"] ** 2
        model = SimpleNamespace(patch_embed=SimpleNamespace(num_patches=n_new, tubelet_size=1), pos_embed=torch.zeros(1, n_new + c["extra"], 16))
        args = SimpleNamespace(finetune=path, model_key="model|module", delete_head=c["delete_head"], nb_classes=c["nb"], num_frames=c["t_new"],
                               model_prefix="")
        got = {}
        keep = utils.load_state_dict
        utils.load_state_dict = lambda m, d, prefix='': got.update(sd=d)
        try:
            checkpoint.load_from_ckpt(args, model)
        finally:
            utils.load_state_dict = keep
        assert list(got["sd"].keys()) == [str(k) for k in z[f"{tag}.out.keys"]], tag
        for k, v in got["sd"].items():
            ref = torch.from_numpy(z[f"{tag}.out.{k}"])
            assert v.shape == ref.shape, (tag, k)
            torch.testing.assert_close(v, ref, atol=1e-6, rtol=1e-6)
