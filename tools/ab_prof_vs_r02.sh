R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3aa; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for t in ab_r02 .; do
  n=$(echo $t | tr -d './'); n=${n:-r03}
  cd $R/$t
  rocprofv3 --kernel-trace --output-format csv -d $O/serial_$n -- python bench.py --serial --steps 12 --warmup 3 --no-cpu-baseline --no-roofline > $O/serial_$n.log 2>&1
  rocprofv3 --kernel-trace --output-format csv -d $O/over_$n -- python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline > $O/over_$n.log 2>&1
  python $R/tools/prof_summary.py $(find $O/serial_$n -name "*kernel_trace.csv" | head -1) 40 > $O/serial_$n.txt
  python $R/tools/prof_summary.py $(find $O/over_$n -name "*kernel_trace.csv" | head -1) 40 > $O/over_$n.txt
  python $R/tools/trace_gaps.py $(find $O/over_$n -name "*kernel_trace.csv" | head -1) | head -3 > $O/gaps_$n.txt
done
find $O -name "*.csv" -delete
head -4 $O/gaps_*.txt
