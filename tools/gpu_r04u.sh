#!/bin/bash
# round 4: models padded up to the cluster kernels' shapes: full GPU suite + fuzz
mkdir -p gpurun_out
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
export TMPDIR=/tmp
timeout 1300 python -m pytest tests -m gpu -x -q -k "not golden_cases and not trained" > gpurun_out/r04u_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04u_pytest.log
tail -12 gpurun_out/r04u_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
