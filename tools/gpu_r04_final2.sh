#!/bin/bash
# round 4: full GPU suite + smoke on the committed build
mkdir -p gpurun_out
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
export TMPDIR=/tmp
timeout 1300 python -m pytest tests -m gpu -x -q > gpurun_out/r04final_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04final_pytest.log
tail -4 gpurun_out/r04final_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout 100 python tools/fuzz_gpu.py 60 777 2>&1 | tail -1
