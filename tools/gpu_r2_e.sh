#!/bin/bash
# round-2 call E: var-len short-kernel debug, residual-prefetch A/B of the SPLIT epilogue
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 300 python tools/debug_varlen_attn.py > $O/debug_varlen.log 2>&1; cat $O/debug_varlen.log | tail -20
timeout 300 python tools/gemm_mode_bench.py 5 > $O/gemm_mode_bench_m30720.log 2>&1; grep "outproj\|ffn2" $O/gemm_mode_bench_m30720.log
GEMM_M=17280 timeout 300 python tools/gemm_mode_bench.py 5 > $O/gemm_mode_bench_m17280.log 2>&1; cat $O/gemm_mode_bench_m17280.log
BG_TUNE="3=1" timeout 300 python bench.py --no-cpu-baseline --no-extra --no-roofline > $O/bench_pf1.log 2>&1; tail -1 $O/bench_pf1.log | cut -c1-250
timeout 300 python bench.py --no-cpu-baseline --no-extra --no-roofline > $O/bench_pf0.log 2>&1; tail -1 $O/bench_pf0.log | cut -c1-250
BG_TUNE="3=1" timeout 300 python bench.py --no-cpu-baseline --no-extra --no-roofline >> $O/bench_pf1.log 2>&1; tail -1 $O/bench_pf1.log | cut -c1-250
