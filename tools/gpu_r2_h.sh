#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py -k "attention or varlen or compacted" -q > $O/pytest_r2h.log 2>&1; echo "pytest rc=$?" >> $O/pytest_r2h.log
tail -4 $O/pytest_r2h.log
timeout 600 python tools/dual_stream_probe.py > $O/dual_stream_probe.log 2>&1; cat $O/dual_stream_probe.log | tail -12
timeout 900 python bench.py --no-cpu-baseline --no-extra > $O/bench_h.log 2>&1; echo "bench rc=$?" >> $O/bench_h.log
tail -2 $O/bench_h.log | cut -c1-300
