#!/bin/bash
# round 4: the float64 -> float32 cast with streaming stores, A/B on the float64-list leg
mkdir -p gpurun_out
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q -k "f64 or python_surface or trained or smoke or in_between" > gpurun_out/r04z_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04z_pytest.log
tail -3 gpurun_out/r04z_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
for rep in 1 2 3; do
for env in "UIS_X=0" "UIS_CAST_PLAIN_STORES=1"; do
  echo "== $env" | tee -a gpurun_out/r04z_f64.txt
  env $env python bench.py --steps 20 --warmup 3 --no_cpu_baseline --no_extra_configs 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print({k: d.get(k) for k in ('value', 'value_host_buffers', 'value_device', 'ms_per_step')})" | tee -a gpurun_out/r04z_f64.txt
done
done
timeout 100 python tools/fuzz_gpu.py 60 5 > gpurun_out/r04z_fuzz.txt 2>&1; tail -1 gpurun_out/r04z_fuzz.txt
