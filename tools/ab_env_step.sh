#!/bin/bash
# same-call A/B of the whole stage-1 step under two environment settings, three interleaved runs each:
#   bash tools/ab_env_step.sh <tag> "<ENV_A>" "<ENV_B>"      e.g.  bash tools/ab_env_step.sh rows "UNITE_GEMM_GROUP_ROWS=0" ""
TAG=$1; A=$2; B=$3
O=gpurun_out/ab_$TAG; mkdir -p $O
for r in 1 2 3; do
  i=0
  for e in "$A" "$B"; do
    i=$((i+1))
    env $e timeout -k 10 200 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-roofline > $O/bench_${i}_r$r.json 2> $O/bench_${i}_r$r.err || exit 1
    python - <<PY
import json
r=json.load(open("$O/bench_${i}_r$r.json")); print("[%s] run $r: %.3f ms/step  %.1f clips/s  loss %.5f" % ("$e" or "default", r["ms_per_step"], r["value"], r["final_loss"]), flush=True)
PY
  done
done
