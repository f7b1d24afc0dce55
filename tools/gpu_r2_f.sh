#!/bin/bash
# round-2 call F: attention / var-len tests after the MASKED split, attention bench, edge-net kernel profile, bench
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_round2.py -k "attention or compact or varlen" -q > $O/pytest_r2f.log 2>&1; echo "pytest rc=$?" >> $O/pytest_r2f.log
tail -4 $O/pytest_r2f.log
timeout 600 python tools/attn_bench.py > $O/attn_bench.log 2>&1; grep '"N": 1800\|"N": 4000' $O/attn_bench.log
export TMPDIR=/tmp; cd /tmp
for mode in varlen dense; do
  rm -rf $O/edge_prof_$mode
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/edge_prof_$mode -o edge -- python $R/tools/edge_eval.py $mode > $O/edge_prof_$mode.log 2>&1
  f=$(find $O/edge_prof_$mode -name "*kernel_stats.csv" | head -1); echo "== $mode"; head -12 "$f" | cut -c1-200
done
cd $R
timeout 900 python bench.py > $O/bench_f.log 2>&1; echo "bench rc=$?" >> $O/bench_f.log
tail -2 $O/bench_f.log | cut -c1-300
find $O -type f -size +8M -delete
