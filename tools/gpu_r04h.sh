#!/bin/bash
# Round 4: the float64-list leg (bench.py's `value`): cast threads, pinned landing block for the labels.
mkdir -p gpurun_out
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
export TMPDIR=/tmp
nproc
leg() { python bench.py --no_cpu_baseline --no_extra_configs --steps 20 --warmup 5 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print({k: d[k] for k in ('value','value_host_buffers','value_device','ms_per_step')})"; }
{
echo "default (16 threads) $(leg)"
for T in 4 8 24 32 48; do echo "UIS_CAST_THREADS=$T $(UIS_CAST_THREADS=$T leg)"; done
echo "default again $(leg)"
} 2>&1 | tee gpurun_out/r04h_f64.txt
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "float64 or host_buffer or python_surface" 2>&1 | tail -3
