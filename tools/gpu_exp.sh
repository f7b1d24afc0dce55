cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
BG_SQUARE=0 timeout 500 python tools/gemm_p256_check.py 8640 15360 17280 30720 > gpurun_out/r3_g1_check.log 2>&1; grep "BIT-EQ\|False" gpurun_out/r3_g1_check.log | head -20; grep -A14 "^M = " gpurun_out/r3_g1_check.log | tail -64
