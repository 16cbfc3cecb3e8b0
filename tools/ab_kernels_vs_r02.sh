mkdir -p gpurun_out/r3z
O=$PWD/gpurun_out/r3z
for i in 1 2; do
(cd ab_r02 && GEMM_ONLY=1 python tools/bench_kernels.py 2>/dev/null) > $O/r02_$i.txt
GEMM_ONLY=1 python tools/bench_kernels.py 2>/dev/null > $O/r03_$i.txt
done
paste -d'|' <(cut -c1-95 $O/r02_2.txt) <(cut -c70-95 $O/r03_2.txt) <(cut -c70-95 $O/r02_1.txt) <(cut -c70-95 $O/r03_1.txt)
