mkdir -p gpurun_out/r3ab
O=$PWD/gpurun_out/r3ab/ab.txt
: > $O
ms() { python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print(d['ms_per_step'])"; }
run() { python bench.py --no-cpu-baseline --no-roofline --steps 40 2>/dev/null | ms; }
for i in 1 2; do
  echo "== previous round" >> $O; (cd ab_r02 && run) >> $O
  echo "== this round" >> $O; run >> $O
  echo "== plan model 2" >> $O; UNITE_PLAN_MODEL=2 run >> $O
  echo "== plan model 2, separate reduce always" >> $O; UNITE_PLAN_MODEL=2 UNITE_SPLITK_SEPARATE=1 run >> $O
  echo "== plan model 2, in-launch reduce always" >> $O; UNITE_PLAN_MODEL=2 UNITE_SPLITK_SEPARATE=0 run >> $O
done
UNITE_PLAN_MODEL=2 UNITE_GEMM_PLAN_DEBUG=1 python bench.py --no-cpu-baseline --no-roofline --steps 2 --warmup 1 2>&1 | grep "ta 1 tb 1 w 0.80" | sort | uniq -c >> $O
cat $O
