mkdir -p gpurun_out/r3v
O=$PWD/gpurun_out/r3v/gemm_ab.txt
: > $O
S='50432,3072,768,bias,qgelu 50432,768,3072,f32,bias,res 50432,768,768,f32,bias,res 10240,768,3072,f32,bias,res 10240,3072,768,bias,gelu'
for i in 1 2; do
echo "== previous round, tile kernels" >> $O; (cd ab_r02 && UNITE_GEMM_PP=0 python tools/gemm_time.py $S 2>&1 | grep -v amdgpu.ids) >> $O
echo "== this round, tile kernels" >> $O; UNITE_GEMM_PP=0 python tools/gemm_time.py $S 2>&1 | grep -v amdgpu.ids >> $O
done
echo "== previous round, teacher_fused_time" >> $O; (cd ab_r02 && python tools/teacher_fused_time.py 2>&1 | tail -3) >> $O
echo "== this round, teacher_fused_time" >> $O; python tools/teacher_fused_time.py 2>&1 | tail -3 >> $O
cat $O
