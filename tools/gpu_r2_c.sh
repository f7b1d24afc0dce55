#!/bin/bash
# round-2 call C: transpose-read probe, new attention kernel tests + micro-bench, bench with edge-net extra
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
hipcc --offload-arch=gfx950 -O2 tools/tr_probe.hip -o /tmp/tr_probe > /dev/null 2>&1 && /tmp/tr_probe > $O/tr_probe.log 2>&1
cat $O/tr_probe.log | head -3
timeout 900 python -m pytest tests/test_gpu_round2.py -k "long_attention or compact or compacted or rescale or wide_logits" -q -x > $O/pytest_r2c.log 2>&1; echo "pytest rc=$?" >> $O/pytest_r2c.log
tail -4 $O/pytest_r2c.log
timeout 600 python tools/attn_bench.py > $O/attn_bench.log 2>&1; cat $O/attn_bench.log
timeout 900 python bench.py --no-cpu-baseline > $O/bench_c.log 2>&1; echo "bench rc=$?" >> $O/bench_c.log
tail -2 $O/bench_c.log | cut -c1-600
