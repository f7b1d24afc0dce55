#!/bin/bash
# One gpurun call: GPU tests, smoke, bench (which runs the rank-local ABC / furniture loops in full itself since round 6), rocprofv3 kernel
# stats, PMC passes on the bench's step mix, a PMC pass over the attention micro-benchmark.
# Everything lands in gpurun_out/.     bash tools/gpu_round.sh [all|test|bench|prof|pmc|attnpmc]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
WHAT=${1:-all}
PMC_STEPS=20; PMC_WARMUP=5        # the driver's own step mix (python bench.py --steps 20 --warmup 5): tools/pmc_summary.py <round> 20 5
if [[ $WHAT == all || $WHAT == test ]]; then
  timeout 1300 python -m pytest tests -m gpu -q --durations=25 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
  tail -3 $O/pytest_gpu.log; tail -2 $O/smoke.log
fi
if [[ $WHAT == all || $WHAT == bench ]]; then
  timeout 1500 python bench.py > $O/bench.log 2>&1; echo "bench rc=$?" >> $O/bench.log
  tail -2 $O/bench.log | cut -c1-260
fi
if [[ $WHAT == all || $WHAT == prof ]]; then
  export TMPDIR=/tmp
  cd /tmp
  # the bench command as the driver runs it (product default: two sample groups in flight), and with serialised launches
  for sp in 0 1; do
    rm -rf $O/prof_stats_split$sp
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats_split$sp -o bench -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-extra --split $sp > $O/prof_stats_split$sp.log 2>&1
    echo "rocprof stats (split $sp) rc=$?" >> $O/prof_stats_split$sp.log
  done
  cd $R
fi
if [[ $WHAT == all || $WHAT == pmc ]]; then
  export TMPDIR=/tmp
  cd /tmp
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    rm -rf $O/pmc$i
    # (the 8-counter SQ pass writes 8 rows per launch: 10 steps keep its CSV small; the per-kernel averages do not need the mix)
    st=$PMC_STEPS; [[ $i == 1 ]] && st=10
    timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc$i -o pmc -- python $R/bench.py --steps $st --warmup $PMC_WARMUP --no-cpu-baseline --no-roofline --no-extra --split 1 > $O/pmc$i.log 2>&1
    echo "pmc$i rc=$? ($set)" >> $O/pmc$i.log
  done
  cd $R
fi
if [[ $WHAT == all || $WHAT == attnpmc ]]; then
  export TMPDIR=/tmp
  cd /tmp
  rm -rf $O/pmc_attn
  timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/pmc_attn -o pmc -- python $R/tools/attn_bench.py > $O/pmc_attn.log 2>&1
  echo "pmc_attn rc=$?" >> $O/pmc_attn.log
  cd $R
  python tools/attn_pmc_summary.py r06 > $O/attn_pmc_summary.log 2>&1
  cp profiles/r06/attn_pmc_per_launch_shape.json $O/ 2>/dev/null
fi
if [[ $WHAT == all || $WHAT == vaeprof ]]; then
  # VAE decode of configs[2]'s output on ONE stream (no concurrent halves: kernel durations add up to the pass)
  export TMPDIR=/tmp
  cd /tmp
  rm -rf $O/vae_prof
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/vae_prof -o vae -- python $R/tools/vae_bench.py 15360 460800 one_call_program serial > $O/vae_prof.log 2>&1
  echo "vaeprof rc=$?" >> $O/vae_prof.log
  rm -f $O/vae_prof/*trace.csv
  cd $R
  timeout 300 python tools/vae_concurrent_bench.py > $O/vae_concurrent_bench.log 2>&1
fi
if [[ $WHAT == driver ]]; then
  # the command the driver runs at round end
  timeout 1700 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.log 2>&1; echo "bench rc=$?" >> $O/bench_driver_cmd.log
  tail -2 $O/bench_driver_cmd.log | cut -c1-260
fi
du -ah $O | sort -h | tail -30 > $O/listing.txt 2>&1
find $O -type f -size +16M -print -delete >> $O/listing.txt 2>&1
