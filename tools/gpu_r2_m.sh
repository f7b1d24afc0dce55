#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
export TMPDIR=/tmp
rm -rf $O/vae_prof
timeout 900 rocprofv3 --kernel-trace --stats -d $O/vae_prof -o vae -- python tools/vae_bench.py 15360 460800 one_call_program > $O/vae_prof.log 2>&1
tail -20 $O/vae_prof.log
f=$(find $O/vae_prof -name "*kernel_stats.csv" | head -1); echo $f; head -30 "$f"
