set -u
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 200 python tools/gemm_p256_check.py 17280 30720 > gpurun_out/r3_p256_clean.log 2>&1; grep -c "True" gpurun_out/r3_p256_clean.log; grep "BIT-EQ" gpurun_out/r3_p256_clean.log
timeout 1300 python -m pytest tests -m gpu -q --durations=25 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -4 gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log; tail -2 gpurun_out/smoke.log
timeout 900 python bench.py > gpurun_out/bench.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench.log
tail -2 gpurun_out/bench.log | cut -c1-400
