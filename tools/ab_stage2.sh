mkdir -p gpurun_out/r3ak
O=$PWD/gpurun_out/r3ak/ab.txt
: > $O
ms() { python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print(d['ms_per_step'])"; }
for i in 1 2; do
echo "== stage2, previous round" >> $O; (cd ab_r02 && python tools/bench_configs.py stage2 2>/dev/null | ms) >> $O
echo "== stage2, this tree" >> $O; python tools/bench_configs.py stage2 2>/dev/null | ms >> $O
echo "== stage2, plan model 3" >> $O; UNITE_PLAN_MODEL=3 python tools/bench_configs.py stage2 2>/dev/null | ms >> $O
echo "== stage2, plan model 3 + in-launch reduce" >> $O; UNITE_PLAN_MODEL=3 UNITE_SPLITK_SEPARATE=0 python tools/bench_configs.py stage2 2>/dev/null | ms >> $O
echo "== stage2, plan model 2 + in-launch reduce" >> $O; UNITE_SPLITK_SEPARATE=0 python tools/bench_configs.py stage2 2>/dev/null | ms >> $O
done
UNITE_GEMM_PLAN_DEBUG=1 python tools/bench_configs.py stage2 2>&1 | grep "ta 1 tb 1" | sort | uniq -c | sort -rn | head -8 >> $O
UNITE_PLAN_MODEL=3 UNITE_GEMM_PLAN_DEBUG=1 python tools/bench_configs.py stage2 2>&1 | grep "ta 1 tb 1" | sort | uniq -c | sort -rn | head -8 >> $O
cat $O
