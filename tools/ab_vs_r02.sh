# same-box A/B of this tree against the previous round's tree (git archive <round-2 commit> | tar -x -C ab_r02, make -C ab_r02/unite_amd/csrc):
# stage-1 bench and the other configs, interleaved in ONE gpurun call.  Usage: bash tools/ab_vs_r02.sh
mkdir -p gpurun_out/r3u
O=$PWD/gpurun_out/r3u/ab.txt
: > $O
ms() { python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print(d['ms_per_step'])"; }
for i in 1 2 3; do
  echo "== stage 1, previous round" >> $O; (cd ab_r02 && python bench.py --no-cpu-baseline --no-roofline --steps 40 2>/dev/null | ms) >> $O
  echo "== stage 1, this round" >> $O; python bench.py --no-cpu-baseline --no-roofline --steps 40 2>/dev/null | ms >> $O
done
for c in stage2 stage3 vitl; do
  for i in 1 2; do
    echo "== $c, previous round" >> $O; (cd ab_r02 && python tools/bench_configs.py $c 2>/dev/null | ms) >> $O
    echo "== $c, this round" >> $O; python tools/bench_configs.py $c 2>/dev/null | ms >> $O
  done
done
cat $O
