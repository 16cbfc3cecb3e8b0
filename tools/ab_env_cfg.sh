#!/bin/bash
# bash tools/ab_env_cfg.sh <config> "<ENV_1>" "<ENV_2>" ...: bench.py --config <n> under each setting, two interleaved rounds
CFG=$1; shift
for r in 1 2; do
  for e in "$@"; do
    env $e timeout -k 10 300 python bench.py --config $CFG --steps 15 --warmup 4 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg $CFG [$e] round $r: %.3f ms/step' % r['ms_per_step'], flush=True)" || exit 1
  done
done
