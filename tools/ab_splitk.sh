mkdir -p gpurun_out/r3w
O=$PWD/gpurun_out/r3w/ab.txt
: > $O
ms() { python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print(d['ms_per_step'])"; }
for i in 1 2; do
  echo "== stage 1, previous round" >> $O; (cd ab_r02 && python bench.py --no-cpu-baseline --no-roofline --steps 40 2>/dev/null | ms) >> $O
  for m in 0 1 2; do echo "== stage 1, split-K reduce mode $m" >> $O; UNITE_SPLITK_SEPARATE=$m python bench.py --no-cpu-baseline --no-roofline --steps 40 2>/dev/null | ms >> $O; done
done
echo "== stage3, previous round" >> $O; (cd ab_r02 && python tools/bench_configs.py stage3 2>/dev/null | ms) >> $O
for m in 0 1 2; do echo "== stage3, split-K reduce mode $m" >> $O; UNITE_SPLITK_SEPARATE=$m python tools/bench_configs.py stage3 2>/dev/null | ms >> $O; done
cat $O
