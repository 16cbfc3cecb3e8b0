mkdir -p gpurun_out/r3ah
O=$PWD/gpurun_out/r3ah/ab.txt
: > $O
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm" > gpurun_out/r3ah/gemm_tests.log 2>&1; echo "gemm tests rc=$?" >> $O; tail -2 gpurun_out/r3ah/gemm_tests.log >> $O
UNITE_SPLITK_SEPARATE=0 timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm" > gpurun_out/r3ah/gemm_tests_inlaunch.log 2>&1; echo "gemm tests (in-launch reduce) rc=$?" >> $O; tail -2 gpurun_out/r3ah/gemm_tests_inlaunch.log >> $O
ms() { python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print(d['ms_per_step'])"; }
run() { python bench.py --no-cpu-baseline --no-roofline --steps 40 2>/dev/null | ms; }
for i in 1 2 3; do
echo "== previous round" >> $O; (cd ab_r02 && run) >> $O
echo "== this tree (plan model 3)" >> $O; run >> $O
echo "== this tree, plan model 2" >> $O; UNITE_PLAN_MODEL=2 run >> $O
done
cat $O
