#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -q -k "vae or decode or encode or pipeline or smoke" > $O/pytest_r2u.log 2>&1; echo "pytest rc=$?" >> $O/pytest_r2u.log
tail -4 $O/pytest_r2u.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
