#!/bin/bash
# same-call comparison of the whole stage-1 step under up to four environment settings, three interleaved rounds:
#   bash tools/ab_env_step3.sh <tag> "<ENV_1>" "<ENV_2>" ["<ENV_3>" ["<ENV_4>"]]
TAG=$1; shift
O=gpurun_out/ab_$TAG; mkdir -p $O
for r in 1 2 3; do
  i=0
  for e in "$@"; do
    i=$((i+1))
    env $e timeout -k 10 200 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-roofline > $O/bench_${i}_r$r.json 2> $O/bench_${i}_r$r.err || exit 1
    python - <<PY
import json
r=json.load(open("$O/bench_${i}_r$r.json")); print("[%s] round $r: %.3f ms/step  %.1f clips/s  loss %.5f" % ("$e" or "default", r["ms_per_step"], r["value"], r["final_loss"]), flush=True)
PY
  done
done
