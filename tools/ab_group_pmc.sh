R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3m; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export UNITE_GEMM_PP=0
for G in 0 8; do
  export UNITE_GEMM_GROUP_ROWS=$G
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/f$G -- python $R/tools/gemm_one.py 50432 3072 768 6 > $O/f$G.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/w$G -- python $R/tools/gemm_one.py 50432 3072 768 6 > $O/w$G.log 2>&1
  echo "== group rows $G" >> $O/pmc.txt
  python $R/tools/pmc_traffic_by_kernel.py $(find $O/f$G -name "*counter_collection.csv") $(find $O/w$G -name "*counter_collection.csv") | grep -E "kernel|gemm" >> $O/pmc.txt
done
find $O -name "*.csv" -delete
cat $O/pmc.txt
