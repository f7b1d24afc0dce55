#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 150 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py -q -x -k "gemm_bf16 or gemm_fp16 or gemm_split or gemm_layernorm_fold or 256_tile or denoiser_bf16 or vae_decode_bf16" > $O/pytest_r2y.log 2>&1; echo "pytest rc=$?" >> $O/pytest_r2y.log; tail -4 $O/pytest_r2y.log
