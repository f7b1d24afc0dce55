#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_round2.py -k "split_streams or hip_graph" -q > $O/pytest_r2i.log 2>&1; echo "pytest rc=$?" >> $O/pytest_r2i.log
tail -4 $O/pytest_r2i.log
for sp in 1 2 3 4; do timeout 300 python bench.py --no-cpu-baseline --no-extra --no-roofline --split $sp > $O/bench_split$sp.log 2>&1; echo "split $sp: $(tail -1 $O/bench_split$sp.log | cut -c80-260)"; done
timeout 900 python bench.py --no-cpu-baseline > $O/bench_i.log 2>&1; echo "bench rc=$?" >> $O/bench_i.log
tail -2 $O/bench_i.log | cut -c1-300
