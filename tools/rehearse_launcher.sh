# the stage-1 driver under torchrun with one rank on RCCL and the reducer's collectives forced (one-GPU rehearsal of the multi-GPU launch path)
mkdir -p gpurun_out/rl
cat > gpurun_out/rl/stage1.yaml <<'Y'
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
warmup_epochs: 0
epochs: 2
batch_size: 8
log_freq: 5
use_cls_token: false
save_ckpt_freq: 1
Y
rm -rf gpurun_out/rl/run
UNITE_DDP_FORCE_COLLECTIVES=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29621 -m unite_amd.run_stage1 \
  --config gpurun_out/rl/stage1.yaml --synthetic --synthetic_steps 10 --output_dir gpurun_out/rl/run --batch_size 8 --seed 1 > gpurun_out/rl/out.log 2>&1
echo rc=$?
grep -E "distributed|Averaged stats|Training time|Error|error|Traceback" gpurun_out/rl/out.log | cut -c1-200 | head -12
cat gpurun_out/rl/run/log.txt 2>/dev/null | cut -c1-200; ls gpurun_out/rl/run 2>/dev/null
