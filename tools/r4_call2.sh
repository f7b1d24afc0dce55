#!/bin/bash
# round 4, GPU call 2: phase groups of the 256 x 256 split kernel (sweep), small-launch tiles at the reference's batch 16, new parity cases
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/c2
mkdir -p $O
cd $R
timeout 900 python tools/gemm_split_bench.py 3 > $O/gemm_split_bench.log 2>&1; echo "rc=$?" >> $O/gemm_split_bench.log
grep -E "BIT-EQUALITY|rc=" $O/gemm_split_bench.log
timeout 900 python -m pytest tests/test_gpu_round4.py tests/test_gpu_parity.py tests/test_gpu_fullsize_oracle.py -m gpu -q -k "round4 or golden or cfg4_abc_edgez or cfg5_furniture_cfg_edgepos" --durations=8 > $O/pytest_subset.log 2>&1; echo "pytest rc=$?" >> $O/pytest_subset.log
tail -12 $O/pytest_subset.log
for t in "15=-1" "15=0"; do BG_TUNE="$t" timeout 300 python tools/cascade_bench.py 16 > $O/cascade_b16_$t.log 2>&1; echo "rc=$?" >> $O/cascade_b16_$t.log; done
BG_TUNE="15=0" timeout 300 python tools/cascade_bench.py 16 x auto > $O/cascade_b16_graphs.log 2>&1
grep -h -A1 "stage_s\|total_s" $O/cascade_b16_*.log | grep -v "^--"
timeout 600 python tools/edge_ab.py "8=-1" "8=0" "8=36" > $O/edge_ab.log 2>&1; echo "rc=$?" >> $O/edge_ab.log
grep -v amdgpu.ids $O/edge_ab.log | cut -c1-330 | tail -10
cp gpurun_out/parity_r04.json gpurun_out/parity_fullsize.json $O/ 2>/dev/null
