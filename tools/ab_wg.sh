mkdir -p gpurun_out/r3ag
O=$PWD/gpurun_out/r3ag/ab.txt
: > $O
ms() { python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print(d['ms_per_step'])"; }
run() { python bench.py --no-cpu-baseline --no-roofline --steps 40 2>/dev/null | ms; }
for i in 1 2 3; do
echo "== previous round" >> $O; (cd ab_r02 && run) >> $O
echo "== this tree (plan model 3)" >> $O; run >> $O
echo "== this tree, plan model 2" >> $O; UNITE_PLAN_MODEL=2 run >> $O
echo "== this tree, plan model 2, in-launch reduce" >> $O; UNITE_PLAN_MODEL=2 UNITE_SPLITK_SEPARATE=0 run >> $O
done
cat $O
