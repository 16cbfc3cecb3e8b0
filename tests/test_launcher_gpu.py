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
