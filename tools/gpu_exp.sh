cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 400 python tools/gemm_p256_check.py 15360 17280 30720 > gpurun_out/r3_fold_final_check.log 2>&1; grep "BIT-EQ\|False" gpurun_out/r3_fold_final_check.log | head; grep -A12 "^M = " gpurun_out/r3_fold_final_check.log | grep "M =\|qkv fold\|ffn1 fold"
timeout 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r3_fold_final_pytest.log 2>&1; tail -3 gpurun_out/r3_fold_final_pytest.log
timeout 600 python tools/face_ldm_ab.py "10=2" "10=0" > gpurun_out/r3_fold_final_ab.log 2>&1; tail -7 gpurun_out/r3_fold_final_ab.log
