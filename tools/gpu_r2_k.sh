#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_round2.py tests/test_gpu_cascade.py tests/test_gpu_parity.py -k "implicit or cascade_matches or vae" -q > $O/pytest_r2k.log 2>&1; echo "pytest rc=$?" >> $O/pytest_r2k.log
tail -5 $O/pytest_r2k.log
timeout 900 python tools/vae_bench.py > $O/vae_bench.log 2>&1; tail -40 $O/vae_bench.log
timeout 600 python bench.py --no-cpu-baseline --no-extra > $O/bench_k.log 2>&1; tail -1 $O/bench_k.log | cut -c1-260
