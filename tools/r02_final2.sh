#!/bin/bash
set -u
ROOT=$GRAFT_REPO_ROOT
cd $ROOT
mkdir -p gpurun_out/r02_c
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/r02_c/pytest.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/r02_c/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r02_c/smoke.log 2>&1; echo "smoke rc=$?"
bash tools/profile_round.sh r02_c > gpurun_out/r02_c/profile.log 2>&1
tail -25 gpurun_out/r02_c/profile.log
