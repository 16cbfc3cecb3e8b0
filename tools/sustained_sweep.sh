#!/bin/bash
# sustained (energy-proxy) time of every GEMM shape of the stage-1 step with its real epilogue, under each tile kernel: which kernel costs least?
O=gpurun_out/r4f; mkdir -p $O
T="t_c_fc:50432,3072,768,bias,qgelu t_c_proj:50432,768,3072,f32,bias,res t_out_proj:50432,768,768,f32,bias,res t_qkv:50432,2304,768,bias"
S="s_qkv:10240,2304,768,bias s_proj:10240,768,768,f32,bias,res s_fc1:10240,3072,768,bias s_fc2:10240,768,3072,f32,bias,res s_dec:10240,512,768,f32,bias"
D="d_fc2:10240,3072,768,nt d_fc1:10240,768,3072,nt,f32 d_proj:10240,768,768,nt d_qkv:10240,768,2304,nt,f32"
W="w_fc1:3072,768,10240,tn w_fc2:768,3072,10240,tn w_qkv:2304,768,10240,tn w_proj:768,768,10240,tn"
for k in deep256 deep128 wide; do
  UNITE_GEMM_KERNEL=$k timeout -k 10 170 python tools/sustained.py 0.2 $T $S $D $W > $O/sweep_$k.txt 2>&1 || exit 1
done
UNITE_GEMM_KERNEL=wide UNITE_GEMM_WIDE_DIRECT=1 timeout -k 10 170 python tools/sustained.py 0.2 $T $S $D > $O/sweep_wide_direct.txt 2>&1 || exit 1
# persistent kernel where it is supported (k-contiguous B only)
PPS=$(for s in $T $S; do echo "$s,pp"; done)
timeout -k 10 170 python tools/sustained.py 0.2 $PPS > $O/sweep_pp.txt 2>&1 || exit 1
# planner's own choice, alone and at the step's sharing weight
timeout -k 10 170 python tools/sustained.py 0.2 $T $S $D $W > $O/sweep_planner.txt 2>&1
UNITE_GEMM_PLAN_WORK=0.8 timeout -k 10 170 python tools/sustained.py 0.2 $T $S $D $W > $O/sweep_planner_w08.txt 2>&1
grep -h -v amdgpu $O/sweep_*.txt
