#!/bin/bash
mkdir -p gpurun_out
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r04l_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04l_pytest.log
tail -8 gpurun_out/r04l_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5
timeout 200 python tools/fuzz_gpu.py 120 31 > gpurun_out/r04l_fuzz.txt 2>&1; tail -2 gpurun_out/r04l_fuzz.txt
