#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests -m gpu -k "vae or cascade_matches or pipeline or decode or encode" -q > $O/pytest_r2l.log 2>&1; echo "pytest rc=$?" >> $O/pytest_r2l.log
tail -15 $O/pytest_r2l.log
timeout 900 python tools/vae_bench.py > $O/vae_bench.log 2>&1; tail -60 $O/vae_bench.log
