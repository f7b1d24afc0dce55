#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests -m gpu -k "vae or decode or encode or groupnorm or im2col" -q > $O/pytest_r2n.log 2>&1; echo "pytest rc=$?" >> $O/pytest_r2n.log
tail -15 $O/pytest_r2n.log
timeout 900 python tools/vae_bench.py 15360 460800 one_call_program,im2col > $O/vae_bench_n.log 2>&1; tail -40 $O/vae_bench_n.log
