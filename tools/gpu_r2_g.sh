#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_fullsize_oracle.py tests/test_gpu_round2.py -k "cfg2 or cfg3_deepcad_edgez or batch_row or varlen_equals" -q > $O/pytest_r2g.log 2>&1; echo "pytest rc=$?" >> $O/pytest_r2g.log
tail -4 $O/pytest_r2g.log
timeout 900 python bench.py --no-cpu-baseline > $O/bench_g.log 2>&1; echo "bench rc=$?" >> $O/bench_g.log
tail -2 $O/bench_g.log | cut -c1-300
