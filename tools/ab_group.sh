mkdir -p gpurun_out/r3m
O=gpurun_out/r3m/group.txt
S='50432,3072,768,bias,qgelu 50432,2304,768,bias 10240,3072,768,bias,gelu 10240,2304,768,bias 50432,768,3072,f32,bias,res'
: > $O
for G in 0 4 6 8 12 16; do echo "== group rows $G" >> $O; UNITE_GEMM_PP=0 UNITE_GEMM_GROUP_ROWS=$G python tools/gemm_time.py $S 2>&1 | grep -v amdgpu.ids >> $O; done
for G in 0 8 0 6 12; do echo "== bench group rows $G" >> $O; UNITE_GEMM_GROUP_ROWS=$G python bench.py --no-cpu-baseline --no-roofline --steps 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])" >> $O; done
cat $O
