"""-m gpu: the stage-1 driver as a user launches it -- ``python -m unite_amd.run_stage1 --config ... --synthetic`` in a fresh process
(reference stage1.sh:15-17 -> run_stage1.py main): YAML + flags through unite_amd.cli, ViT-B/16 student + CLIP-B/16 teacher, three
synthetic steps, checkpoint and log written, and a second launch resumes from checkpoint-latest.pth."""
import json
import os
import subprocess
import sys

import pytest
import torch
import yaml

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(600)
def test_run_stage1_synthetic(tmp_path):
    cfg = tmp_path / "stage1.yaml"
    cfg.write_text(yaml.safe_dump(dict(
        model="adaptation_umt_base_patch16_224", num_frames=8, tubelet_size=1, clip_decoder_embed_dim=768, clip_output_dim=512,
        clip_return_layers=[6, 7, 8, 9, 10, 11], clip_teacher="clip_b16", clip_return_attn=True, clip_loss_data="mixed", mask_type="attention",
        mask_ratio=0.8, drop_path=0.1, opt="adamw", opt_betas=[0.9, 0.95], lr=1.5e-4, warmup_epochs=0, epochs=1, batch_size=2,
        log_freq=1, use_cls_token=False, save_ckpt_freq=1)))
    out = tmp_path / "run"
    cmd = [sys.executable, "-m", "unite_amd.run_stage1", "--config", str(cfg), "--synthetic", "--synthetic_steps", "3", "--output_dir", str(out),
           "--batch_size", "2", "--seed", "3"]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    log = [json.loads(l) for l in open(out / "log.txt")]
    assert len(log) == 1 and log[0]["epoch"] == 0 and log[0]["n_parameters"] == 88005888
    assert 0.5 < log[0]["train_loss"] < 2.5 and log[0]["train_grad_norm"] > 0
    # run_stage1.py:798-799: lr and min_lr scaled by the global batch / 256; the meter averages the cosine schedule over the three steps
    assert 1e-5 * 2 / 256 < log[0]["train_lr"] < 1.5e-4 * 2 / 256 and abs(log[0]["train_min_lr"] - log[0]["train_lr"]) < 1e-12
    ck = torch.load(out / "checkpoint-latest.pth", map_location="cpu", weights_only=True)
    assert set(ck) >= {"model", "optimizer", "epoch", "scaler", "args"} and ck["epoch"] == 0 and len(ck["model"]) == 184
    assert (out / "checkpoint-0.pth").exists() and (out / "config.yaml").exists()
    # second launch: auto-resume finds checkpoint-latest.pth, start_epoch = 1 = epochs -> nothing left to train
    r2 = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=500)
    assert r2.returncode == 0, r2.stdout[-2000:] + r2.stderr[-3000:]
    assert "Resume checkpoint" in r2.stdout and len(open(out / "log.txt").readlines()) == 1


def _seeded_init(seed, build):
    """the driver seeds torch with seed + rank and builds its model on the CPU before moving it: reproducible here"""
    import numpy as np
    torch.manual_seed(seed)
    np.random.seed(seed)
    return {k: v.clone() for k, v in build().state_dict().items()}


@pytest.mark.timeout(900)
def test_run_stage2_synthetic(tmp_path):
    """``python -m unite_amd.run_stage2 --config ... --synthetic`` (reference stage2.sh -> run_stage2.py:455-848): two epochs with update_freq 2,
    layer-wise lr decay, --lp_ft_epochs 1 (blocks 0-8 + patch embedding frozen in epoch 0: run_stage2.py:741-746, trainable from epoch 1: :758),
    validation every epoch with the best checkpoint kept, final_test + merge, the scalar logger fed per iteration."""
    from types import SimpleNamespace
    cfg = tmp_path / "stage2.yaml"
    cfg.write_text(yaml.safe_dump(dict(
        model="vit_base_patch16_224", nb_classes=5, num_frames=4, num_segments=1, tubelet_size=1, use_mean_pooling=True, init_scale=0.001,
        drop_path=0.0, opt="adamw", opt_betas=[0.9, 0.999], lr=1e-3, min_lr=1e-6, warmup_epochs=0, epochs=2, batch_size=2, update_freq=2,
        layer_decay=0.65, lr_schedule="cosine", eval_freq=1, save_ckpt_freq=1, frozen_layers="", lp_ft_epochs=1, test_best=False,
        weight_decay=0.05, smoothing=0.0, input_size=224)))
    out = tmp_path / "run"
    cmd = [sys.executable, "-m", "unite_amd.run_stage2", "--config", str(cfg), "--synthetic", "--synthetic_steps", "2", "--output_dir", str(out),
           "--seed", "3"]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    log = [json.loads(l) for l in open(out / "log.txt")]
    assert [l.get("epoch") for l in log[:2]] == [0, 1] and "Final top-1" in log[2]
    assert log[0]["train_loss"] > 0 and "val_acc1" in log[0] and "train_grad_norm" in log[0]
    assert (out / "checkpoint-best.pth").exists() and (out / "checkpoint-latest.pth").exists() and (out / "0.txt").exists()
    scal = out / "scalars.jsonl"
    if scal.exists():            # no tensorboard writer installed: launch.ScalarLog
        rows = [json.loads(l) for l in open(scal)]
        keys = {r_["key"] for r_ in rows}
        assert {"loss/loss", "opt/lr", "opt/grad_norm", "perf/val_acc1"} <= keys
        assert sum(r_["key"] == "loss/loss" for r_ in rows) == 2 * 4       # every iteration of both epochs (2 steps x update_freq 2)
    # trainable-parameter schedule: epoch 0 leaves blocks 0-8 and the patch embedding exactly at their initial values
    from unite_amd import run_stage2
    a = SimpleNamespace(model="vit_base_patch16_224", nb_classes=5, num_frames=4, num_segments=1, tubelet_size=1, use_learnable_pos_emb=False,
                        fc_drop_rate=0.0, drop=0.0, drop_path=0.0, attn_drop_rate=0.0, use_checkpoint=False, checkpoint_num=0,
                        use_mean_pooling=True, init_scale=0.001, head_type="linear", head_hidden_dim=2048)
    ck0 = torch.load(out / "checkpoint-0.pth", map_location="cpu", weights_only=True)["model"]
    ck1 = torch.load(out / "checkpoint-1.pth", map_location="cpu", weights_only=True)["model"]
    a.head_type = getattr(torch.load(out / "checkpoint-0.pth", map_location="cpu", weights_only=True)["args"], "head_type", "linear")
    a.head_hidden_dim = getattr(torch.load(out / "checkpoint-0.pth", map_location="cpu", weights_only=True)["args"], "head_hidden_dim", 2048)
    init = _seeded_init(3, lambda: run_stage2.get_model(a))
    for k in ("patch_embed.proj.weight", "blocks.0.attn.qkv.weight", "blocks.8.mlp.fc2.bias"):
        assert torch.equal(ck0[k], init[k]), k                          # frozen in epoch 0
        assert not torch.equal(ck1[k], ck0[k]), k                       # released at epoch lp_ft_epochs
    for k in ("blocks.9.attn.qkv.weight", "blocks.11.mlp.fc1.weight", "head.weight"):
        assert not torch.equal(ck0[k], init[k]), k


@pytest.mark.timeout(900)
def test_run_stage3_synthetic(tmp_path):
    """``python -m unite_amd.run_stage3 --config ... --synthetic`` (reference stage3.sh -> run_stage3.py:992-1414): one epoch of two
    source + target steps with the default clip_matchORconf selection (zero-shot probabilities from the frozen CLIP image tower against
    seeded text features), validation, checkpoints and src_classifier_latest.pth; the source classifier and clip_decoder.* are not
    optimised (Appendix A-8: the classifier is outside the optimizer; the decoders get no gradient in stage 3)."""
    from types import SimpleNamespace
    cfg = tmp_path / "stage3.yaml"
    cfg.write_text(yaml.safe_dump(dict(
        model="adaptation_umt_base_patch16_224", num_frames=8, tubelet_size=1, clip_decoder_embed_dim=768, clip_output_dim=512,
        clip_return_layers=[6], clip_teacher="clip_b16", clip_return_attn=True, mask_type="attention", mask_ratio=0.8, masking_type="clip_attention",
        drop_path=0.0, opt="adamw", opt_betas=[0.9, 0.95], lr=1e-4, warmup_epochs=0, epochs=1, batch_size=2, log_freq=1, use_cls_token=False,
        save_ckpt_freq=1, nb_classes=5, src_classifier_type="linear", class_loss_src_ratio=1.0, selection_strategy="clip_matchORconf",
        clip_threshold=0.3, val_interval=1, return_aug_for_val=True, input_size=224)))
    out = tmp_path / "run"
    cmd = [sys.executable, "-m", "unite_amd.run_stage3", "--config", str(cfg), "--synthetic", "--synthetic_steps", "2", "--output_dir", str(out),
           "--seed", "4"]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    log = [json.loads(l) for l in open(out / "log.txt")]
    assert len(log) == 2 and log[0]["epoch"] == 0 and log[0]["train_loss"] > 0 and "train_select_ratio" in log[0] and "val_acc1" in log[0]
    # the run ends with final_test + merge (run_stage3.py:1393-1409): per-view logits in 0.txt, the headline accuracies as the last log line
    assert set(log[1]) == {"Final top-1", "Final Top-5"} and 0.0 <= log[1]["Final top-1"] <= log[1]["Final Top-5"] <= 100.0
    views = open(out / "0.txt").read().splitlines()
    assert len(views) == 1 + 2 * 4 and views[1].startswith("video_0_") and views[1].count(",") == 4       # 2 batches x 4 views, 5 logits each
    assert (out / "checkpoint-latest.pth").exists() and (out / "src_classifier_latest.pth").exists()
    ck = torch.load(out / "checkpoint-latest.pth", map_location="cpu", weights_only=True)["model"]
    from unite_amd import run_stage3
    a = SimpleNamespace(model="adaptation_umt_base_patch16_224", drop_path=0.0, use_learnable_pos_emb=False, use_checkpoint=False, checkpoint_num=0,
                        clip_decoder_embed_dim=768, clip_output_dim=512, clip_norm_type="l2", num_frames=8, tubelet_size=1, clip_return_layers=[6],
                        clip_student_return_interval=1, use_cls_token=False)
    init = _seeded_init(4, lambda: run_stage3.get_model(a))
    assert torch.equal(ck["clip_decoder.0.head.weight"], init["clip_decoder.0.head.weight"])      # no gradient -> AdamW leaves it alone
    assert not torch.equal(ck["encoder.blocks.0.attn.qkv.weight"], init["encoder.blocks.0.attn.qkv.weight"])
    head = torch.load(out / "src_classifier_latest.pth", map_location="cpu", weights_only=True)
    assert set(head) == {"weight", "bias"} and head["weight"].shape == (5, 768)


def _write_videos(root, n, seed):
    import numpy as np
    rng = np.random.RandomState(seed)
    lines = []
    for i in range(n):
        F, H, W = int(rng.randint(12, 40)), int(rng.choice([240, 256, 270])), int(rng.choice([320, 340, 256]))
        np.save(root / f"clip{seed}_{i}.npy", rng.randint(0, 256, size=(F, H, W, 3), dtype=np.uint8))
        lines.append(f"clip{seed}_{i}.npy {i % 4}")
    ann = root / f"list{seed}.txt"
    ann.write_text("\n".join(lines) + "\n")
    return ann


def test_device_loader_batches_equal_the_cpu_pipeline(tmp_path):
    """unite_amd.datasets end to end on .npy videos of different sizes: every batch of the DeviceLoader equals, BIT FOR BIT, what the
    reference's per-clip CPU pipeline produces from the same draws -- crop + Pillow-bilinear resize (oracle/pil_resize.py, pinned on Pillow),
    flip, /255, normalise, (C,T,H,W) (build.py:32-54, mae.py:217-219) -- and carries the tube masks and labels of its samples."""
    import random
    import types
    import numpy as np
    from oracle.pil_resize import crop_resize_bilinear
    from unite_amd.datasets import build_pretraining_dataset, DeviceLoader
    ann = _write_videos(tmp_path, 6, 1)
    args = types.SimpleNamespace(input_size=224, mask_type="tube", mask_ratio=0.8, window_size=(8, 14, 14), color_jitter=0.0, flip=True,
                                 prefix=str(tmp_path), split=" ", num_segments=8, num_frames=8, umt_step=1, use_decord=True, num_sample=1)
    ds = build_pretraining_dataset(args, str(ann))
    dev = torch.device("cuda:0")
    loader = DeviceLoader(ds, 3, dev, sampler=None, num_workers=0, drop_last=True)
    assert len(loader) == 2
    np.random.seed(11)
    random.seed(12)
    got = [(v.clone().cpu(), m.clone(), t.clone()) for v, m, t in loader]
    np.random.seed(11)
    random.seed(12)
    from oracle import umt_oracle as O
    from unite_amd.data import IMAGENET_DEFAULT_MEAN, IMAGENET_DEFAULT_STD
    for bi in range(2):
        videos, masks, targets = got[bi]
        assert tuple(videos.shape) == (3, 3, 8, 224, 224) and tuple(masks.shape) == (3, 1568) and targets.tolist() == [(3 * bi + j) % 4 for j in range(3)]
        for j in range(3):
            frames, box, flip, mask, _ = ds[3 * bi + j]                   # the same draws again, in the same order
            clip = np.stack([crop_resize_bilinear(f, box, (224, 224)) for f in frames.numpy()])        # (T,224,224,3) uint8
            ref = O.clip_to_tensor(torch.from_numpy(clip)[None], IMAGENET_DEFAULT_MEAN, IMAGENET_DEFAULT_STD,
                                   torch.tensor([1 if flip else 0], dtype=torch.uint8))[0]            # flip, /255, normalise, (C,T,H,W)
            assert torch.equal(videos[j], ref), (bi, j, float((videos[j] - ref).abs().max()))
            assert np.array_equal(masks[j].numpy(), mask)


@pytest.mark.timeout(600)
def test_run_stage1_on_npy_videos_with_a_target_domain(tmp_path):
    """``python -m unite_amd.run_stage1`` WITHOUT --synthetic: source and target lists of .npy videos through unite_amd.datasets
    (run_stage1.py:654-745: the smaller target list is repeated to the source's length, both loaders step together, B_s + B_t clips per step,
    lr scaled by the doubled batch), two loader workers, attention masks from the teacher."""
    src, tgt = _write_videos(tmp_path, 8, 2), _write_videos(tmp_path, 3, 3)
    cfg = tmp_path / "stage1.yaml"
    cfg.write_text(yaml.safe_dump(dict(
        model="adaptation_umt_base_patch16_224", num_frames=8, tubelet_size=1, clip_decoder_embed_dim=768, clip_output_dim=512,
        clip_return_layers=[6, 7, 8, 9, 10, 11], clip_teacher="clip_b16", clip_return_attn=True, clip_loss_data="mixed", mask_type="attention",
        mask_ratio=0.8, drop_path=0.1, opt="adamw", opt_betas=[0.9, 0.95], lr=1.5e-4, warmup_epochs=0, epochs=1, batch_size=2, log_freq=1,
        use_cls_token=False, save_ckpt_freq=1, ann_file_train=str(src), ann_file_train_target=str(tgt), prefix=str(tmp_path), split=" ",
        num_segments=8, num_workers=2, flip=True, input_size=224)))
    out = tmp_path / "run"
    cmd = [sys.executable, "-m", "unite_amd.run_stage1", "--config", str(cfg), "--output_dir", str(out), "--batch_size", "2", "--seed", "5"]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "Repeating target dataset 3 times" in r.stdout and "Batch size = 4" in r.stdout and "Number of training steps per epoch = 4" in r.stdout
    log = [json.loads(l) for l in open(out / "log.txt")]
    assert len(log) == 1 and 0.5 < log[0]["train_loss"] < 2.5 and log[0]["train_grad_norm"] > 0


@pytest.mark.timeout(600)
def test_bench_data_parallel_flow_rehearsed_with_one_rank(tmp_path):
    """bench.py as the driver launches it, except that the process group has ONE rank (UNITE_DDP_FORCE_COLLECTIVES=1): backend nccl = RCCL, the
    student under the data-parallel wrapper, bucket all-reduces on the reducer's stream, barrier + synchronize around the timed steps.  stdout
    must be exactly the JSON record (RCCL prints a banner at communicator creation: it belongs on stderr) and the loss must be a trained one."""
    env = dict(os.environ, UNITE_DDP_FORCE_COLLECTIVES="1", MASTER_PORT="29611")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "2", "--batch", "4", "--no-cpu-baseline",
                        "--no-roofline"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[:2000]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 1 and rec["steps"] == 3 and rec["unit"] == "clips/s" and rec["value"] > 0 and 0.0 < rec["final_loss"] < 4.0
    assert rec["config"]["global_batch"] == 4 and rec["scaling"] == "weak" and rec["higher_is_better"] is True


@pytest.mark.timeout(900)
def test_bench_starts_its_own_ranks_without_a_launcher(tmp_path):
    """`python bench.py --gpus 2` with no torchrun around it (how a driver's `--gpus N` leg may call it): bench.py starts the two ranks itself
    before touching the GPU (self_start: fresh children under torch.distributed.run on 127.0.0.1), relays rank 0's ONE JSON line and returns the
    launcher's code.  Two ranks share the one GPU of the test box over gloo, so the hardware-queue rule must leave HIP's default pool alone."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "GPU_MAX_HW_QUEUES")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "2", "--warmup", "1",
                        "--batch", "4", "--no-cpu-baseline", "--no-roofline"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[:2000]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["ranks_seen"] == 2 and rec["config"]["global_batch"] == 8 and rec["config"]["parallelism"] == "dp2"
    assert rec["value"] > 0 and 0.0 < rec["final_loss"] < 4.0 and "share" in rec["hw_queues"]


def _write_cls_videos(root, n, seed, n_cls=4):
    """labelled .npy clips of 240 x 320 (landscape) / 320 x 240 (portrait) frames: the short side is NOT the 224 the transforms ask for, so
    validation / test views go through the device resize"""
    import numpy as np
    rng = np.random.RandomState(seed)
    lines = []
    for i in range(n):
        F = int(rng.randint(10, 24))
        hw = (320, 240) if i % 3 == 2 else (240, 320)
        np.save(root / f"cls{seed}_{i}.npy", rng.randint(0, 256, size=(F, hw[0], hw[1], 3), dtype=np.uint8))
        lines.append(f"cls{seed}_{i}.npy {i % n_cls}")
    ann = root / f"cls_list{seed}.txt"
    ann.write_text("\n".join(lines) + "\n")
    return ann


@pytest.mark.timeout(900)
def test_run_stage2_on_npy_videos(tmp_path):
    """``python -m unite_amd.run_stage2`` WITHOUT --synthetic (run_stage2.py:492-563): train / validation / test lists through
    unite_amd/datasets_cls.py -- RandAugment + every draw in two loader workers, the pixels on the GPU -- one epoch, validation, final_test over
    2 temporal x 3 spatial views per test video, merge."""
    tr, va, te = _write_cls_videos(tmp_path, 8, 11), _write_cls_videos(tmp_path, 4, 12), _write_cls_videos(tmp_path, 3, 13)
    cfg = tmp_path / "stage2.yaml"
    cfg.write_text(yaml.safe_dump(dict(
        model="vit_base_patch16_224", nb_classes=4, num_frames=4, num_segments=1, tubelet_size=1, use_mean_pooling=True, init_scale=0.001,
        drop_path=0.0, opt="adamw", opt_betas=[0.9, 0.999], lr=1e-3, min_lr=1e-6, warmup_epochs=0, epochs=1, batch_size=2, update_freq=1,
        layer_decay=0.65, lr_schedule="cosine", eval_freq=1, save_ckpt_freq=1, frozen_layers="", test_best=False, weight_decay=0.05, smoothing=0.0,
        input_size=224, short_side_size=224, data_set="Kinetics_sparse", ann_file_train=str(tr), ann_file_val=str(va), ann_file_test=str(te),
        prefix=str(tmp_path), split=" ", sampling_rate=0, test_num_segment=2, test_num_crop=3, num_workers=2, aa="rand-m7-n4-mstd0.5-inc1",
        reprob=0.25, remode="pixel", recount=1, train_interpolation="bicubic", num_sample=1, train_fraction=1.0, dist_eval=True)))
    out = tmp_path / "run"
    cmd = [sys.executable, "-m", "unite_amd.run_stage2", "--config", str(cfg), "--output_dir", str(out), "--seed", "6"]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "Use Dataset: Kinetics_sparse" in r.stdout and "Number of training steps per epoch = 4" in r.stdout
    log = [json.loads(l) for l in open(out / "log.txt")]
    assert log[0]["epoch"] == 0 and log[0]["train_loss"] > 0 and "val_acc1" in log[0] and "Final top-1" in log[1]
    views = open(out / "0.txt").read().splitlines()
    assert len(views) == 1 + 3 * 2 * 3 and {v.split(" ")[0] for v in views[1:]} == {f"cls13_{i}" for i in range(3)}


@pytest.mark.timeout(900)
def test_run_stage3_on_npy_videos(tmp_path):
    """``python -m unite_amd.run_stage3`` WITHOUT --synthetic (run_stage3.py:1042-1145): the source list in train mode, the (shorter, repeated)
    target list in validation mode with its RandAugment second view, validation, final_test + merge."""
    src, tgt, va, te = (_write_cls_videos(tmp_path, n, s) for n, s in ((6, 21), (4, 22), (4, 23), (2, 24)))
    cfg = tmp_path / "stage3.yaml"
    cfg.write_text(yaml.safe_dump(dict(
        model="adaptation_umt_base_patch16_224", num_frames=8, tubelet_size=1, clip_decoder_embed_dim=768, clip_output_dim=512,
        clip_return_layers=[6], clip_teacher="clip_b16", clip_return_attn=True, mask_type="attention", mask_ratio=0.8, masking_type="clip_attention",
        drop_path=0.0, opt="adamw", opt_betas=[0.9, 0.95], lr=1e-4, warmup_epochs=0, epochs=1, batch_size=2, batch_size_val=4, log_freq=1,
        use_cls_token=False, save_ckpt_freq=1, nb_classes=4, src_classifier_type="linear", class_loss_src_ratio=1.0, selection_strategy="consORconf",
        val_interval=1, return_aug_for_val=True, input_size=224, short_side_size=224, data_set="Kinetics_sparse", ann_file_train=str(src),
        ann_file_train_target=str(tgt), ann_file_val=str(va), ann_file_test=str(te), prefix=str(tmp_path), split=" ", sampling_rate=0,
        test_num_segment=2, test_num_crop=3, num_workers=2, aa="rand-m7-n4-mstd0.5-inc1", reprob=0.25, remode="pixel", recount=1,
        train_interpolation="bicubic", num_sample=1, train_fraction=1.0, train_repetitions=0)))
    out = tmp_path / "run"
    cmd = [sys.executable, "-m", "unite_amd.run_stage3", "--config", str(cfg), "--output_dir", str(out), "--seed", "8"]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "Repeating target dataset 2 times" in r.stdout and "Number of training steps per epoch = 3" in r.stdout
    log = [json.loads(l) for l in open(out / "log.txt")]
    assert log[0]["epoch"] == 0 and log[0]["train_loss"] > 0 and "train_select_ratio" in log[0] and "val_acc1" in log[0] and "Final top-1" in log[1]
    assert len(open(out / "0.txt").read().splitlines()) == 1 + 2 * 2 * 3
