mkdir -p gpurun_out/soak
cat > gpurun_out/soak/stage1.yaml <<'Y'
model: adaptation_umt_base_patch16_224
num_frames: 8
tubelet_size: 1
clip_decoder_embed_dim: 768
clip_output_dim: 512
clip_return_layers: [6, 7, 8, 9, 10, 11]
clip_teacher: clip_b16
clip_return_attn: true
clip_loss_data: mixed
mask_type: attention
mask_ratio: 0.8
drop_path: 0.1
opt: adamw
opt_betas: [0.9, 0.95]
lr: 0.00015
warmup_epochs: 1
epochs: 4
batch_size: 32
log_freq: 50
use_cls_token: false
save_ckpt_freq: 100
checkpoints_enabled: false
Y
python -m unite_amd.run_stage1 --config gpurun_out/soak/stage1.yaml --synthetic --synthetic_steps 150 --output_dir gpurun_out/soak/run --batch_size 32 --seed 1 > gpurun_out/soak/out.log 2>&1
echo rc=$?
grep -E "Epoch \[[0-9]\]: +\[ *(0|50|100|149)/150\]|Averaged stats|Training time" gpurun_out/soak/out.log | cut -c1-220
cat gpurun_out/soak/run/log.txt
