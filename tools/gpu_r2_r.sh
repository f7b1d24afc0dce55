#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/test_gpu_cascade.py tests/test_gpu_round2.py -q -k "cascade or graph or bench_collective" > $O/pytest_r2r.log 2>&1; echo "pytest rc=$?" >> $O/pytest_r2r.log
tail -6 $O/pytest_r2r.log
for g in nographs graphs; do
  timeout 300 python tools/cascade_bench.py 16 x $g > $O/cascade_b16_$g.log 2>&1; tail -19 $O/cascade_b16_$g.log
done
timeout 300 python tools/cascade_bench.py 256 datalike > $O/cascade_datalike_auto.log 2>&1; tail -19 $O/cascade_datalike_auto.log
