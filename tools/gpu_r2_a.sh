#!/bin/bash
# round-2 call A: CU census, new parity tests, bench (default + CU-mate tile walk A/B)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
hipcc --offload-arch=gfx950 -O2 tools/cu_census.hip -o /tmp/cu_census > $O/cu_census.log 2>&1 && /tmp/cu_census >> $O/cu_census.log 2>&1
timeout 900 python -m pytest tests/test_gpu_round2.py tests/test_gpu_fullsize_oracle.py "tests/test_gpu_cascade.py::test_cascade_matches_oracle_cascade" -q -s > $O/pytest_r2a.log 2>&1; echo "pytest rc=$?" >> $O/pytest_r2a.log
timeout 600 python bench.py > $O/bench_a.log 2>&1; echo "bench rc=$?" >> $O/bench_a.log
BG_TUNE="5=2" timeout 300 python bench.py --no-cpu-baseline --steps 30 > $O/bench_walk2.log 2>&1
timeout 300 python bench.py --no-cpu-baseline --steps 30 > $O/bench_walk0.log 2>&1
BG_TUNE="5=2" timeout 300 python bench.py --no-cpu-baseline --steps 30 >> $O/bench_walk2.log 2>&1
tail -3 $O/pytest_r2a.log; tail -1 $O/bench_walk0.log | cut -c1-300; tail -1 $O/bench_walk2.log | cut -c1-300
