#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -k "vae or decode or encode or groupnorm or im2col or attn or upsample" -q > $O/pytest_r2o.log 2>&1; echo "pytest rc=$?" >> $O/pytest_r2o.log
tail -8 $O/pytest_r2o.log
rm -rf $O/vae_prof
timeout 900 rocprofv3 --kernel-trace --stats -d $O/vae_prof -o vae -- python tools/vae_bench.py 15360 460800 one_call_program > $O/vae_prof.log 2>&1
grep -A12 one_call_program $O/vae_prof.log | head -20
