#!/bin/bash
mkdir -p gpurun_out
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q -k "oversized or python_surface or depth2_upper" > gpurun_out/r04ag_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04ag_pytest.log
tail -12 gpurun_out/r04ag_pytest.log
