#!/bin/bash
# round-2 call B: var-len tests, kernel parity suite (plumbing changed), bench (var-len headline + dense extra)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_round2.py -k "varlen" -q -x > $O/pytest_r2b_varlen.log 2>&1; echo "pytest rc=$?" >> $O/pytest_r2b_varlen.log
timeout 900 python -m pytest tests/test_gpu_parity.py -q > $O/pytest_r2b_parity.log 2>&1; echo "pytest rc=$?" >> $O/pytest_r2b_parity.log
timeout 600 python bench.py --no-cpu-baseline > $O/bench_b.log 2>&1; echo "bench rc=$?" >> $O/bench_b.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
tail -5 $O/pytest_r2b_varlen.log; tail -5 $O/pytest_r2b_parity.log; tail -2 $O/bench_b.log | cut -c1-1500; tail -3 $O/smoke.log
