# The round's artifact run on the GPU box (tools/README.md): bench line, rocprofv3 kernel stats of `bench.py --serial`, HBM traffic
# (FETCH_SIZE / WRITE_SIZE in their own passes) and MFMA-utilisation counters per kernel.  Every rocprofv3 command has python directly
# after `--`; counters are collected with --kernel-trace only.  Usage (from the repo root on the box):  bash tools/final_prof.sh r02
set -e
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $O/bench_line.json 2> $O/bench.log
tail -c 900 $O/bench_line.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --serial --steps 12 --warmup 3 --no-cpu-baseline --no-roofline > $O/stats.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline > $O/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python $R/bench.py --serial --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python $R/bench.py --serial --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $O/pmc_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma -- python $R/bench.py --serial --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $O/pmc_mfma.log 2>&1
cd $R
STATS=$(find $O/stats -name "*kernel_stats.csv" | head -1)
TRACE=$(find $O/stats -name "*kernel_trace.csv" | head -1)
python tools/prof_summary.py $TRACE 60 > $O/${TAG}_bench_kernel_stats.txt
cp $STATS $O/${TAG}_bench_kernel_stats.csv
python tools/pmc_traffic.py $(find $O/pmc_fetch -name "*counter_collection.csv" | head -1) $(find $O/pmc_write -name "*counter_collection.csv" | head -1) $O/${TAG}_gemm_traffic.json
python tools/pmc_kernels.py $(find $O/pmc_mfma -name "*counter_collection.csv" | head -1) $O/${TAG}_mfma_counters.csv > $O/${TAG}_mfma_counters.txt
python tools/trace_gaps.py $(find $O/trace -name "*kernel_trace.csv" | head -1) > $O/${TAG}_overlapped_step_gaps.txt
python tools/prof_summary.py $(find $O/trace -name "*kernel_trace.csv" | head -1) 40 > $O/${TAG}_overlapped_kernel_stats.txt
cp $O/bench_line.json $O/${TAG}_bench_line.json
ls -la $O
