# stage-2 and stage-3 drivers under torchrun with one rank on RCCL, the reducer's collectives forced (outputs under /tmp: checkpoints are large)
W=/tmp/rl23; rm -rf $W; mkdir -p $W gpurun_out/rl23
python - <<'PY'
import yaml
yaml.safe_dump(dict(model="vit_base_patch16_224", nb_classes=5, num_frames=4, num_segments=1, tubelet_size=1, use_mean_pooling=True, init_scale=0.001,
    drop_path=0.0, opt="adamw", opt_betas=[0.9, 0.999], lr=1e-3, min_lr=1e-6, warmup_epochs=0, epochs=2, batch_size=2, update_freq=2, layer_decay=0.65,
    lr_schedule="cosine", eval_freq=1, save_ckpt_freq=1, frozen_layers="", lp_ft_epochs=1, test_best=False, weight_decay=0.05, smoothing=0.0,
    input_size=224), open("/tmp/rl23/stage2.yaml", "w"))
PY
export UNITE_DDP_FORCE_COLLECTIVES=1
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29631 -m unite_amd.run_stage2 --config $W/stage2.yaml \
  --synthetic --synthetic_steps 2 --output_dir $W/run2 --seed 3 > gpurun_out/rl23/stage2.log 2>&1
echo "stage 2 rc=$?"; cut -c1-160 $W/run2/log.txt
python - <<'PY'
import yaml
yaml.safe_dump(dict(model="adaptation_umt_base_patch16_224", num_frames=8, tubelet_size=1, clip_decoder_embed_dim=768, clip_output_dim=512,
    clip_return_layers=[6], clip_teacher="clip_b16", clip_return_attn=True, mask_type="attention", mask_ratio=0.8, masking_type="clip_attention",
    drop_path=0.0, opt="adamw", opt_betas=[0.9, 0.95], lr=1e-4, warmup_epochs=0, epochs=1, batch_size=2, log_freq=1, use_cls_token=False,
    save_ckpt_freq=1, nb_classes=5, src_classifier_type="linear", class_loss_src_ratio=1.0, selection_strategy="clip_matchORconf",
    clip_threshold=0.3, val_interval=1, return_aug_for_val=True, input_size=224), open("/tmp/rl23/stage3.yaml", "w"))
PY
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29632 -m unite_amd.run_stage3 --config $W/stage3.yaml \
  --synthetic --synthetic_steps 2 --output_dir $W/run3 --seed 4 > gpurun_out/rl23/stage3.log 2>&1
echo "stage 3 rc=$?"; cut -c1-160 $W/run3/log.txt; ls $W/run3
grep -E "Traceback|Error" gpurun_out/rl23/stage2.log gpurun_out/rl23/stage3.log | head
