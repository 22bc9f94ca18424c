#!/bin/bash
# Round 4, first call: the GPU tests on this round's library + the new bench line (value = the
# float64-list leg) + a first U-sweep on round 3's kernels (the baseline the generalised
# k_decode_rs has to beat).
mkdir -p gpurun_out
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04a_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04a_pytest.log
tail -5 gpurun_out/r04a_pytest.log
timeout 600 python bench.py > gpurun_out/r04a_bench_c1.json 2> gpurun_out/r04a_bench_c1.err
python - <<'PY'
import json
d = json.load(open('gpurun_out/r04a_bench_c1.json'))
print({k: d[k] for k in ('value', 'value_leg', 'value_predict_f64', 'value_host_buffers', 'value_device', 'ms_per_step')})
print(d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['effective'], d['roofline']['traffic'])
print(d['cpu_baseline'])
for e in d['extra_configs'] or []:
    print({k: e.get(k) for k in ('config', 'value', 'kernel', 'frac', 'effective_frac', 'parity', 'error')})
PY
# U-sweep, 500-frame utterances, device leg only
for U in 1 8 64 65 96 128 256 257 1024; do
  python bench.py --utterances $U --timed device --steps 5 --warmup 2 --no_cpu_baseline --no_host_buffers --no_extra_configs 2>/dev/null \
    | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('U=$U', d['value'], d['ms_per_step'], d['roofline']['kernel'])"
done | tee gpurun_out/r04a_usweep.txt
