# one-GPU rehearsal of the data-parallel bench flow on RCCL (one rank, collectives forced): bash tools/rehearse_rccl.sh
mkdir -p gpurun_out/rccl
O=$PWD/gpurun_out/rccl/rehearsal.txt
: > $O
ms() { python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print(d['ms_per_step'], d['final_loss'], d['final_grad_norm'])"; }
run() { python bench.py --no-cpu-baseline --no-roofline --steps 30 2>>gpurun_out/rccl/err.log | ms; }
echo "== plain (no wrapper)" >> $O; run >> $O
echo "== one-rank RCCL group, bucket all-reduces forced (torch.distributed nccl)" >> $O; UNITE_DDP_FORCE_COLLECTIVES=1 run >> $O
echo "== + AdamW per bucket" >> $O; UNITE_DDP_FORCE_COLLECTIVES=1 UNITE_BUCKET_ADAMW=1 run >> $O
echo "== + libunite_comm.so instead of the process group" >> $O; UNITE_DDP_FORCE_COLLECTIVES=1 UNITE_COMM_NATIVE=1 run >> $O
echo "== plain (no wrapper)" >> $O; run >> $O
cat $O
