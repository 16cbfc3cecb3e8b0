mkdir -p gpurun_out/r3y
O=$PWD/gpurun_out/r3y/ab.txt
: > $O
ms() { python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print(d['ms_per_step'])"; }
run() { python bench.py --no-cpu-baseline --no-roofline --steps 40 2>/dev/null | ms; }
for i in 1 2; do
  echo "== previous round" >> $O; (cd ab_r02 && run) >> $O
  echo "== this round" >> $O; run >> $O
  echo "== this round, UNITE_GEMM_PLAN_WORK=0.8" >> $O; UNITE_GEMM_PLAN_WORK=0.8 run >> $O
  echo "== this round, separate column sums" >> $O; UNITE_WGRAD_ROWSUM=0 run >> $O
  echo "== this round, separate column sums + separate split-K reduce" >> $O; UNITE_WGRAD_ROWSUM=0 UNITE_SPLITK_SEPARATE=1 run >> $O
  echo "== this round, in-launch reduce everywhere" >> $O; UNITE_SPLITK_SEPARATE=0 run >> $O
done
cat $O
