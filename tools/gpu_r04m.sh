#!/bin/bash
# round 4: look_ahead >= 2 in one launch (k_decode_big<WIN>): parity, then configs[2] A/B against the launch-per-sub-step path
mkdir -p gpurun_out
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -k "look_ahead or calculate_score or golden or probes or python_surface or level" > gpurun_out/r04m_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04m_pytest.log
tail -15 gpurun_out/r04m_pytest.log
for env in "" "UIS_NO_WINDOW_LAUNCH=1"; do
  echo "== configs[2] $env" | tee -a gpurun_out/r04m_c2.txt
  env $env timeout 300 python bench.py --config 2 --steps 3 --warmup 1 --no_extra_configs --no_cpu_baseline 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print({k: d.get(k) for k in ('value', 'value_host_buffers', 'value_device', 'ms_per_step')}, d.get('roofline', {}).get('kernel'))" | tee -a gpurun_out/r04m_c2.txt
done
