mkdir -p gpurun_out/r3x
O=$PWD/gpurun_out/r3x/ab.txt
: > $O
ms() { python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print(d['ms_per_step'])"; }
for i in 1 2 3; do
  for t in ab_r02 ab_6292352 ab_dbfbb57 .; do echo "== $t" >> $O; (cd $t && python bench.py --no-cpu-baseline --no-roofline --steps 40 2>/dev/null | ms) >> $O; done
done
cat $O
