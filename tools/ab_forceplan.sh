mkdir -p gpurun_out/r3ac
O=$PWD/gpurun_out/r3ac/ab.txt
: > $O
ms() { python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print(d['ms_per_step'])"; }
run() { python bench.py --no-cpu-baseline --no-roofline --steps 40 2>/dev/null | ms; }
echo "== previous round" >> $O; (cd ab_r02 && run) >> $O
echo "== plan model 2" >> $O; UNITE_PLAN_MODEL=2 run >> $O
for fp in 2,2 2,3 2,4 2,6 1,3 1,5 1,8; do echo "== force $fp" >> $O; UNITE_GEMM_FORCE_PLAN=$fp run >> $O; done
echo "== previous round" >> $O; (cd ab_r02 && run) >> $O
echo "== plan model 2" >> $O; UNITE_PLAN_MODEL=2 run >> $O
cat $O
