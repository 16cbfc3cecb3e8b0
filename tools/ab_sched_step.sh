#!/bin/bash
# same-call A/B of the whole stage-1 step under the two main-loop schedules of the tile GEMM kernels (UNITE_GEMM_SCHED=0|1), three interleaved runs
O=gpurun_out/r4b; mkdir -p $O
for r in 1 2 3; do
  for s in 0 1; do
    UNITE_GEMM_SCHED=$s timeout -k 10 200 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline > $O/bench_s${s}_r$r.json 2> $O/bench_s${s}_r$r.err || exit 1
    python - <<PY
import json
r=json.load(open("$O/bench_s${s}_r$r.json")); print("sched $s run $r: %.3f ms/step  %.1f clips/s  loss %.5f" % (r["ms_per_step"], r["value"], r["final_loss"]), flush=True)
PY
  done
done
