mkdir -p gpurun_out/r3l
O=gpurun_out/r3l/walk.txt
S='50432,3072,768,bias,qgelu 50432,768,3072,f32,bias,res 50432,768,768,f32,bias,res'
echo "== tile kernels" > $O
UNITE_GEMM_PP=0 python tools/gemm_time.py $S >> $O 2>&1
for W in 0 2 3 4 6 8; do echo "== pp walk $W" >> $O; UNITE_GEMM_PP=2 UNITE_PP_WALK=$W python tools/gemm_time.py $S >> $O 2>&1; done
echo "== bench default" >> $O
python bench.py --no-cpu-baseline --no-roofline --steps 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])" >> $O
for W in 2 4 8; do echo "== bench teacher pp=2 walk $W" >> $O; UNITE_TEACHER_PP=2 UNITE_PP_WALK=$W python bench.py --no-cpu-baseline --no-roofline --steps 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])" >> $O; done
echo "== bench teacher pp=1 walk 4" >> $O; UNITE_TEACHER_PP=1 UNITE_PP_WALK=4 python bench.py --no-cpu-baseline --no-roofline --steps 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])" >> $O
echo "== bench default" >> $O
python bench.py --no-cpu-baseline --no-roofline --steps 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])" >> $O
cat $O
