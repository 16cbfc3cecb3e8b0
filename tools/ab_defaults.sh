mkdir -p gpurun_out/r3aq
O=$PWD/gpurun_out/r3aq/ab.txt
: > $O
ms() { python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print(d['ms_per_step'])"; }
run() { python bench.py --no-cpu-baseline --no-roofline --steps 40 2>/dev/null | ms; }
for i in 1 2; do
echo "== default" >> $O; run >> $O
for w in 0.6 0.7 0.9; do echo "== UNITE_GEMM_SHARING=$w" >> $O; UNITE_GEMM_SHARING=$w run >> $O; done
for s in 2 4; do echo "== UNITE_TEACHER_AHEAD_SLOTS=$s" >> $O; UNITE_TEACHER_AHEAD_SLOTS=$s run >> $O; done
echo "== UNITE_TEACHER_FUSED=0" >> $O; UNITE_TEACHER_FUSED=0 run >> $O
echo "== UNITE_WGRAD_STREAMS=2" >> $O; UNITE_WGRAD_STREAMS=2 run >> $O
echo "== UNITE_DECODER_STREAM=1" >> $O; UNITE_DECODER_STREAM=1 run >> $O
echo "== UNITE_GEMM_PP=0 (student on tile kernels too)" >> $O; UNITE_GEMM_PP=0 run >> $O
done
cat $O
