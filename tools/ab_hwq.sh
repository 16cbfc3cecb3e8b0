mkdir -p gpurun_out/rccl
O=$PWD/gpurun_out/rccl/hwq.txt
: > $O
ms() { python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print(d['ms_per_step'], d['final_loss'])"; }
run() { python bench.py --no-cpu-baseline --no-roofline --steps 30 2>/dev/null | ms; }
echo "== plain, default HW queues" >> $O; run >> $O
for q in 4 6 8 12 16; do
echo "== forced one-rank RCCL collectives, GPU_MAX_HW_QUEUES=$q" >> $O; GPU_MAX_HW_QUEUES=$q UNITE_DDP_FORCE_COLLECTIVES=1 run >> $O
done
for q in 8 16; do echo "== plain, GPU_MAX_HW_QUEUES=$q" >> $O; GPU_MAX_HW_QUEUES=$q run >> $O; done
cat $O
