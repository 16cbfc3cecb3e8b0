#!/bin/bash
# bash tools/ab_tree_cfg.sh <dir of the other tree> <config> [rounds]: bench.py --config <n> of the other tree and of this one, interleaved
OTHER=$1; CFG=$2; N=${3:-3}
for r in $(seq 1 $N); do
  for t in $OTHER .; do
    (cd $t && timeout -k 10 300 python bench.py --config $CFG --steps 15 --warmup 4 --no-cpu-baseline --no-roofline 2>/dev/null) | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg $CFG [$t] round $r: %.3f ms/step' % r['ms_per_step'], flush=True)" || exit 1
  done
done
