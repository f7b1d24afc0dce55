cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
(timeout 200 python tools/gemm_instr.py 15360; timeout 200 python tools/gemm_instr.py 30720) > gpurun_out/r3_instr.log 2>&1; tail -14 gpurun_out/r3_instr.log
