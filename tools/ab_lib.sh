# A/B of two builds of libunite_hip.so in one call: bash tools/ab_lib.sh <other.so> [rounds]   (this tree's build vs <other.so>)
mkdir -p gpurun_out/ablib
O=$PWD/gpurun_out/ablib/ab.txt
: > $O
ms() { python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print(d['ms_per_step'])"; }
run() { python bench.py --no-cpu-baseline --no-roofline --steps 40 2>/dev/null | ms; }
for i in $(seq 1 ${2:-3}); do
echo "== other build ($1)" >> $O; UNITE_HIP_LIB=$PWD/$1 run >> $O
echo "== this build" >> $O; run >> $O
done
cat $O
