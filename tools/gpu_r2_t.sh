#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python bench.py --no-cpu-baseline > $O/bench_t.log 2>&1; echo "bench rc=$?" >> $O/bench_t.log
tail -2 $O/bench_t.log | cut -c1-300
