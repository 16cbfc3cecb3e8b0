mkdir -p gpurun_out/r3ai
O=$PWD/gpurun_out/r3ai/ab.txt
: > $O
ms() { python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print(d['ms_per_step'])"; }
run() { UNITE_PLAN_MODEL=2 python bench.py --no-cpu-baseline --no-roofline --steps 40 2>/dev/null | ms; }
for i in 1 2 3; do
echo "== LayerNorm forward at 32 registers" >> $O; run >> $O
echo "== LayerNorm forward at 48 registers (cannot sit beside a GEMM workgroup)" >> $O; UNITE_HIP_LIB=$PWD/ab_fat/unite_amd/lib/libunite_hip.so run >> $O
done
cat $O
