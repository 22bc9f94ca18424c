#!/bin/bash
# round 4: weight fragments requested a k-block ahead in the wave-per-row-tile dense stages: parity + stage clocks + rates
mkdir -p gpurun_out
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -k "look_ahead or more_utterances or golden" > gpurun_out/r04s_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04s_pytest.log
tail -3 gpurun_out/r04s_pytest.log
UIS_LIB_PATH=$PWD/build/variants/timing.so timeout 300 python bench.py --config 2 --steps 1 --warmup 0 --no_extra_configs --no_cpu_baseline 2>&1 >/dev/null | grep "window launch timing" | head -6 | tee gpurun_out/r04s_win_timing.txt
for cfg in 2 3; do
  timeout 300 python bench.py --config $cfg --steps 3 --warmup 1 --no_extra_configs --no_cpu_baseline 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print({k: d.get(k) for k in ('value', 'value_host_buffers', 'value_device', 'ms_per_step')}, d.get('roofline', {}).get('kernel'))" | tee -a gpurun_out/r04s_bench.txt
done
UIS_NO_WINDOW_LAUNCH=1 timeout 300 python bench.py --config 2 --steps 3 --warmup 1 --no_extra_configs --no_cpu_baseline 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print({k: d.get(k) for k in ('value', 'value_host_buffers', 'value_device', 'ms_per_step')}, d.get('roofline', {}).get('kernel'))" | tee -a gpurun_out/r04s_bench.txt
