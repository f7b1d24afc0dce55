#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python tools/gemm_mode_bench.py 5 138752 0,127,254,381 > $O/gemm_stagger_m138752.log 2>&1; cat $O/gemm_stagger_m138752.log
timeout 600 python tools/gemm_mode_bench.py 5 30720 0,127,254 > $O/gemm_stagger_m30720.log 2>&1; cat $O/gemm_stagger_m30720.log
