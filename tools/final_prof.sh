set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $O/bench_line.json 2> $O/bench.log
tail -c 600 $O/bench_line.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --serial --steps 10 --warmup 3 --no-cpu-baseline > $O/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python $R/bench.py --serial --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python $R/bench.py --serial --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $O/pmc_write.log 2>&1
find $O -name "*.csv" | head -20
