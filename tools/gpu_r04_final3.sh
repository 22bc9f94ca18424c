#!/bin/bash
# round 4: the driver's sequence on HEAD -- GPU tests, smoke, default bench
mkdir -p gpurun_out
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/r04final3_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04final3_pytest.log; tail -3 gpurun_out/r04final3_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
python bench.py > gpurun_out/r04_bench_c1.json 2> gpurun_out/r04_bench_c1.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.load(open('gpurun_out/r04_bench_c1.json'))
print({k: d[k] for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'scaling', 'vs_baseline', 'dtype', 'value_host_buffers', 'value_device')})
print(d['roofline']['kernel'], d['roofline']['avg_launch_us'], d['roofline']['frac'], d['roofline']['effective'], d['roofline']['traffic'])
print(d['cpu_baseline'])
for e in d['extra_configs'] or []:
    print({k: e.get(k) for k in ('config', 'value', 'kernel', 'frac', 'effective_frac', 'parity', 'error')})
PY
