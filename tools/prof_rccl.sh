R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/rccl; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export UNITE_DDP_FORCE_COLLECTIVES=1
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline > $O/trace.log 2>&1
cd $R
T=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python tools/prof_summary.py $T 14 > $O/kernels.txt
python tools/trace_queues.py $T > $O/queues.txt
find $O -name "*.csv" -delete
head -16 $O/kernels.txt; cat $O/queues.txt
