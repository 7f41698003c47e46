#!/bin/bash
set -u
ROOT=$GRAFT_REPO_ROOT
cd $ROOT
mkdir -p gpurun_out/r02_d
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/r02_d/pytest.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/r02_d/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r02_d/smoke.log 2>&1; echo "smoke rc=$?"
bash tools/profile_round.sh r02_d > gpurun_out/r02_d/profile.log 2>&1
tail -25 gpurun_out/r02_d/profile.log
