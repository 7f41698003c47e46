#!/bin/bash
set -u
TAG=${1:-round}   # output directory under gpurun_out/ and prefix of the profile pass: bash tools/round_end.sh r02_d
ROOT=$GRAFT_REPO_ROOT
cd $ROOT
mkdir -p gpurun_out/${TAG}
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/${TAG}/pytest.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/${TAG}/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}/smoke.log 2>&1; echo "smoke rc=$?"
bash tools/profile_round.sh ${TAG} > gpurun_out/${TAG}/profile.log 2>&1
tail -25 gpurun_out/${TAG}/profile.log
