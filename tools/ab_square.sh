for G in 0 4 8 16; do UNITE_GEMM_PP=0 UNITE_GEMM_GROUP_ROWS=$G python tools/gemm_time.py 8192,8192,8192 4096,4096,4096 50432,3072,3072 2>&1 | grep -v amdgpu; done
UNITE_GEMM_PP=2 python tools/gemm_time.py 8192,8192,8192 4096,4096,4096 50432,3072,3072 2>&1 | grep -v amdgpu
