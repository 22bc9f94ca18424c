#!/bin/bash
# round 4: k_window's winners ranked by one wave; phases (launch-per-sub-step build with clocks) + configs[2]
mkdir -p gpurun_out
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -k "look_ahead or calculate_score or golden or probes or level" > gpurun_out/r04q_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04q_pytest.log
tail -3 gpurun_out/r04q_pytest.log
UIS_NO_WINDOW_LAUNCH=1 UIS_LIB_PATH=$PWD/build/variants/seltiming.so python bench.py --timed device --no_cpu_baseline --no_host_buffers --no_extra_configs --config 2 --steps 1 --warmup 0 2>&1 >/dev/null | grep "window timing" | tail -2 | tee gpurun_out/r04q_window_phases.txt
timeout 300 python bench.py --config 2 --steps 3 --warmup 1 --no_extra_configs --no_cpu_baseline 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print({k: d.get(k) for k in ('value', 'value_host_buffers', 'value_device', 'ms_per_step')}, d.get('roofline', {}).get('kernel'))" | tee -a gpurun_out/r04q_bench.txt
UIS_LIB_PATH=$PWD/build/variants/timing.so timeout 300 python bench.py --config 2 --steps 1 --warmup 0 --no_extra_configs --no_cpu_baseline 2>&1 >/dev/null | grep "window launch timing" | tail -4 | tee gpurun_out/r04q_win_timing.txt
UIS_NO_WINDOW_LAUNCH=1 timeout 300 python bench.py --config 2 --steps 3 --warmup 1 --no_extra_configs --no_cpu_baseline 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print({k: d.get(k) for k in ('value', 'value_host_buffers', 'value_device', 'ms_per_step')}, d.get('roofline', {}).get('kernel'))" | tee -a gpurun_out/r04q_bench.txt
