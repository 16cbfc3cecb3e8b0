# kernel trace of the one-rank RCCL rehearsal through torch.distributed and through libunite_comm.so: per-queue busy time and the RCCL kernels
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/rccl_native; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export UNITE_DDP_FORCE_COLLECTIVES=1
for mode in torch native; do
  if [ $mode = native ]; then export UNITE_COMM_NATIVE=1; fi
  rocprofv3 --kernel-trace --output-format csv -d $O/trace_$mode -- python $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline > $O/trace_$mode.log 2>&1
  T=$(find $O/trace_$mode -name "*kernel_trace.csv" | head -1)
  echo "== $mode" >> $O/summary.txt
  (cd $R && python tools/trace_queues.py $T >> $O/summary.txt; python tools/prof_summary.py $T 40 | grep -i "rccl\|reduce\|nccl\|kernel time\|Reduce" >> $O/summary.txt)
  find $O/trace_$mode -name "*.csv" -delete
done
cat $O/summary.txt
