#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd $R
bash tools/gpu_round.sh all
timeout 600 python tools/vae_bench.py 15360 460800 one_call_program > $O/vae_bench_final.log 2>&1; tail -22 $O/vae_bench_final.log
timeout 600 python tools/cascade_bench.py 256 > $O/cascade_random_final.log 2>&1; tail -20 $O/cascade_random_final.log
