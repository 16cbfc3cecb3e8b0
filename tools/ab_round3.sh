mkdir -p gpurun_out/r3o
O=gpurun_out/r3o/ab.txt
: > $O
run() { python bench.py --no-cpu-baseline --no-roofline --steps 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
echo "== default" >> $O; run >> $O
echo "== res16" >> $O; UNITE_TEACHER_RES16=1 run >> $O
echo "== default" >> $O; run >> $O
echo "== res16" >> $O; UNITE_TEACHER_RES16=1 run >> $O
for W in 0.0 0.5 0.8 1.0; do echo "== wgrad sharing $W" >> $O; UNITE_WGRAD_SHARING=$W run >> $O; done
cat $O
