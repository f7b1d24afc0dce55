#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 300 python -m pytest tests/test_gpu_round2.py -q -k "256_tile or gemv" > $O/pytest_r2w.log 2>&1; echo "pytest rc=$?" >> $O/pytest_r2w.log; tail -4 $O/pytest_r2w.log
