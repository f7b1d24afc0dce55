#!/bin/bash
# round-2 call D: attention correctness (fixed max exchange), setprio A/B, PMC passes of the attention kernel, bench
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_round2.py -k "long_attention or compact or compacted or rescale or wide_logits" -q > $O/pytest_r2d.log 2>&1; echo "pytest rc=$?" >> $O/pytest_r2d.log
tail -6 $O/pytest_r2d.log
timeout 600 python tools/attn_bench.py > $O/attn_bench_prio0.log 2>&1; grep '"N": 1800\|"N": 4000' $O/attn_bench_prio0.log | grep dense
BG_TUNE="7=1" timeout 600 python tools/attn_bench.py > $O/attn_bench_prio1.log 2>&1; grep '"N": 1800\|"N": 4000' $O/attn_bench_prio1.log | grep dense
export TMPDIR=/tmp; cd /tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_UNALIGNED_STALL"; do
  i=$((i+1)); rm -rf $O/attn_pmc$i
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/attn_pmc$i -o pmc -- python $R/tools/attn_pmc.py > $O/attn_pmc$i.log 2>&1
done
cd $R
python tools/pmc_dirs.py $O/attn_pmc1 $O/attn_pmc2 > $O/attn_pmc_summary.json 2>&1; head -c 3000 $O/attn_pmc_summary.json
timeout 900 python bench.py --no-cpu-baseline > $O/bench_d.log 2>&1; echo "bench rc=$?" >> $O/bench_d.log
tail -2 $O/bench_d.log | cut -c1-300
find $O -type f -size +8M -delete
