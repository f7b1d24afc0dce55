#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -k "vae or decode or encode or im2col or cascade_matches" -q > $O/pytest_r2p.log 2>&1; echo "pytest rc=$?" >> $O/pytest_r2p.log
tail -5 $O/pytest_r2p.log
timeout 300 python tools/cascade_bench.py 256 datalike > $O/cascade_datalike.log 2>&1; tail -22 $O/cascade_datalike.log
timeout 600 python tools/cascade_bench.py 256 > $O/cascade_random.log 2>&1; tail -22 $O/cascade_random.log
