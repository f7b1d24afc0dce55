#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cascade.py -q -k "vae_decode or vae_encode or cascade_matches or pipeline_driver" > $O/pytest_r2x.log 2>&1; echo "pytest rc=$?" >> $O/pytest_r2x.log; tail -4 $O/pytest_r2x.log
