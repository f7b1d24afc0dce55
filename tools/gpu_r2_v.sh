#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 300 python tools/gemm_variant_check.py 6 17280 138752 > $O/gemm_variant6.log 2>&1; cat $O/gemm_variant6.log
timeout 600 python -m pytest tests/test_gpu_parity.py -q -k "gemm or denoiser or fold or embed or encoder_layer" > $O/pytest_r2v.log 2>&1; echo "pytest rc=$?" >> $O/pytest_r2v.log; tail -4 $O/pytest_r2v.log
