R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3ae; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for t in ab_r02 .; do
  n=$(echo $t | tr -d './'); n=${n:-r03}
  cd $R/$t
  UNITE_PLAN_MODEL=2 rocprofv3 --kernel-trace --output-format csv -d $O/over_$n -- python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline > $O/over_$n.log 2>&1
  python $R/tools/trace_queues.py $(find $O/over_$n -name "*kernel_trace.csv" | head -1) > $O/queues_$n.txt
done
find $O -name "*.csv" -delete
cat $O/queues_*.txt
