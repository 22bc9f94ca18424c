#!/bin/bash
# round 4: final check of the build -- full GPU suite, smoke, the default bench line
mkdir -p gpurun_out
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
export TMPDIR=/tmp
timeout 1300 python -m pytest tests -m gpu -x -q > gpurun_out/r04y_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04y_pytest.log
tail -4 gpurun_out/r04y_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
python bench.py > gpurun_out/r04y_bench.json 2> gpurun_out/r04y_bench.err; tail -c 600 gpurun_out/r04y_bench.err
python - <<'PY'
import json
d = json.load(open('gpurun_out/r04y_bench.json'))
print({k: d[k] for k in ('value', 'value_leg', 'value_predict_f64', 'value_host_buffers', 'value_device', 'ms_per_step')})
print(d['roofline']['kernel'], d['roofline']['avg_launch_us'], d['roofline']['frac'], d['roofline']['effective'])
for e in d['extra_configs'] or []:
    print({k: e.get(k) for k in ('config', 'value', 'kernel', 'frac', 'effective_frac', 'parity', 'error')})
PY
