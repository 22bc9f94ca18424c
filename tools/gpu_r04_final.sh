#!/bin/bash
# round 4: final check of the committed build -- full GPU suite, smoke, a longer soak, look-ahead U-sweep, ragged lines
mkdir -p gpurun_out
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
export TMPDIR=/tmp
timeout 1300 python -m pytest tests -m gpu -x -q > gpurun_out/r04final_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04final_pytest.log
tail -4 gpurun_out/r04final_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
python - <<'PY' > gpurun_out/r04_usweep_lookahead.json
import json, subprocess, sys
out = []
for u in (8, 64, 256, 1024):
    for env in ({}, {'UIS_NO_WINDOW_LAUNCH': '1'}):
        import os
        e = dict(os.environ); e.update(env)
        r = subprocess.run([sys.executable, 'bench.py', '--config', '2', '--utterances', str(u), '--frames', '200', '--timed', 'device',
                            '--steps', '3', '--warmup', '1', '--no_cpu_baseline', '--no_host_buffers', '--no_extra_configs'],
                           capture_output=True, text=True, env=e)
        d = json.loads(r.stdout.strip().splitlines()[-1])
        out.append({'utterances': u, 'one_launch': not env, 'frames_per_s': d['value'], 'ms_per_pass': d['ms_per_step'],
                    'kernel': d['roofline']['kernel'], 'frac': d['roofline']['frac']})
print(json.dumps({'workload': 'U utterances x 200 frames x 256-dim, beam 50, look_ahead 2, test_iteration 2, device leg', 'sweep': out}, indent=1))
PY
python bench.py --ragged --no_cpu_baseline --no_extra_configs > gpurun_out/r04_bench_c1_ragged.json 2>/dev/null
python bench.py --config 2 --ragged --no_cpu_baseline --no_extra_configs --no_host_buffers > gpurun_out/r04_bench_c2_ragged.json 2>/dev/null
timeout 300 python tools/fuzz_gpu.py 240 2026 > gpurun_out/r04final_fuzz.txt 2>&1
timeout 300 python tools/stress_resident.py 100 >> gpurun_out/r04final_fuzz.txt 2>&1
timeout 200 python tools/stress_persistent.py >> gpurun_out/r04final_fuzz.txt 2>&1
tail -4 gpurun_out/r04final_fuzz.txt
python - <<'PY'
import json
for r in json.load(open('gpurun_out/r04_usweep_lookahead.json'))['sweep']:
    print(r)
for f in ('r04_bench_c1_ragged', 'r04_bench_c2_ragged'):
    d = json.load(open('gpurun_out/%s.json' % f))
    print(f, d['value'], d['value_leg'], d.get('value_device'), d['ms_per_step'], d['roofline']['kernel'])
PY
