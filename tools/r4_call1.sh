#!/bin/bash
# round 4, GPU call 1: validate the pipelined split GEMM + the applied epilogue diets, A/B them in situ, first full bench.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/c1
mkdir -p $O
cd $R
timeout 600 python tools/gemm_split_bench.py 5 > $O/gemm_split_bench.log 2>&1; echo "rc=$?" >> $O/gemm_split_bench.log
grep -E "BIT-EQUALITY|rc=" $O/gemm_split_bench.log
if grep -q "BIT-EQUALITY OK" $O/gemm_split_bench.log; then PIPE_OK=1; else PIPE_OK=0; export BG_TUNE="12=1"; echo "PIPE KERNEL NOT BIT-EQUAL: falling back to 12=1 for the rest"; fi
timeout 900 python -m pytest tests/test_gpu_round4.py tests/test_gpu_round3.py tests/test_gpu_round2.py tests/test_gpu_parity.py -m gpu -q -x -k "split_pipe or p256 or implicit or vae or varlen or split_streams" --durations=8 > $O/pytest_subset.log 2>&1; echo "pytest rc=$?" >> $O/pytest_subset.log
tail -4 $O/pytest_subset.log
BG_SPLITS=1,2 timeout 600 python tools/face_ldm_ab.py "12=1" "12=0" "12=0,13=1" "12=1,10=2" > $O/face_ldm_ab.log 2>&1; echo "rc=$?" >> $O/face_ldm_ab.log
cat $O/face_ldm_ab.log | tail -14
timeout 600 python tools/edge_ab.py "12=1" "12=0" > $O/edge_ab.log 2>&1; echo "rc=$?" >> $O/edge_ab.log
grep -v amdgpu.ids $O/edge_ab.log | cut -c1-250 | tail -8
for t in "14=0" "14=1"; do BG_TUNE="$t" timeout 300 python tools/vae_bench.py 15360 460800 one_call_program > $O/vae_bench_$t.log 2>&1; echo "rc=$?" >> $O/vae_bench_$t.log; done
grep -h total_s $O/vae_bench_14=*.log
timeout 900 python bench.py > $O/bench.log 2>&1; echo "bench rc=$?" >> $O/bench.log
tail -2 $O/bench.log | cut -c1-400
echo PIPE_OK=$PIPE_OK
