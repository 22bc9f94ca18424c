#!/bin/bash
mkdir -p gpurun_out
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
export TMPDIR=/tmp
timeout 300 python -m pytest tests -m gpu -x -q -k "deep_models" > gpurun_out/r04ak_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04ak_pytest.log
tail -12 gpurun_out/r04ak_pytest.log
