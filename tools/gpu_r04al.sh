#!/bin/bash
mkdir -p gpurun_out
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/r04al_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04al_pytest.log
tail -4 gpurun_out/r04al_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 90 python tools/fuzz_gpu.py 60 909 > gpurun_out/r04al_fuzz.txt 2>&1; tail -2 gpurun_out/r04al_fuzz.txt
