#!/bin/bash
# Round-4 evidence, second pass (after the one-launch look-ahead decode and the barrier-shadow changes):
# the default bench line again (extra_configs carry configs[2] on k_decode_big<WIN>), configs[2] / [3] on their own,
# kernel traces and PMC passes for configs[2], stage clocks, the U-sweep, soaks.  Everything under gpurun_out/.
mkdir -p gpurun_out
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
export TMPDIR=/tmp
python bench.py > gpurun_out/r04_bench_c1.json 2> gpurun_out/r04_bench_c1.err
python bench.py --config 2 --no_cpu_baseline > gpurun_out/r04_bench_c2.json 2>/dev/null
UIS_NO_WINDOW_LAUNCH=1 python bench.py --config 2 --no_cpu_baseline > gpurun_out/r04_bench_c2_stepwise.json 2>/dev/null
python bench.py --config 3 --no_cpu_baseline > gpurun_out/r04_bench_c3.json 2>/dev/null
python bench.py --config 4 --no_cpu_baseline > gpurun_out/r04_bench_c4.json 2>/dev/null
python bench.py --config 3 --ragged --no_cpu_baseline --no_host_buffers > gpurun_out/r04_bench_c3_ragged.json 2>/dev/null
# U-sweep: utterances per GPU x 500 frames, device leg
python - <<'PY' > gpurun_out/r04_usweep.json
import json, subprocess, sys
out = []
for u in (1, 8, 32, 64, 65, 72, 96, 128, 192, 256, 257, 512, 1024):
    r = subprocess.run([sys.executable, 'bench.py', '--utterances', str(u), '--timed', 'device', '--steps', '5', '--warmup', '2',
                        '--no_cpu_baseline', '--no_host_buffers', '--no_extra_configs'], capture_output=True, text=True)
    d = json.loads(r.stdout.strip().splitlines()[-1])
    out.append({'utterances': u, 'frames_per_s': d['value'], 'ms_per_pass': d['ms_per_step'], 'kernel': d['roofline']['kernel'],
                'us_per_decode_step': round(d['roofline']['avg_launch_us'] / 1000.0, 2), 'frac': d['roofline']['frac'],
                'effective_frac': d['roofline']['effective']['frac']})
print(json.dumps({'workload': 'U utterances x 500 frames x 256-dim, beam 10, test_iteration 2 (1000 decode steps), device leg',
                  'sweep': out}, indent=1))
PY
# kernel traces
for c in 2 3; do
  BENCH_ARGS="--config $c --steps 3 --warmup 1 --timed device --no_cpu_baseline --no_host_buffers" ./tools/gpu_prof.sh > /dev/null 2>&1
  cp gpurun_out/kernel_stats.csv gpurun_out/r04_kernel_stats_c$c.csv
done
UIS_NO_WINDOW_LAUNCH=1 BENCH_ARGS="--config 2 --steps 3 --warmup 1 --timed device --no_cpu_baseline --no_host_buffers" ./tools/gpu_prof.sh > /dev/null 2>&1
cp gpurun_out/kernel_stats.csv gpurun_out/r04_kernel_stats_c2_stepwise.csv
# PMC passes for configs[2] (separate runs per counter group)
BENCH_ARGS="--config 2 --no_extra_configs --no_host_buffers --timed device" ./tools/gpu_pmc.sh > /dev/null 2>&1
cp gpurun_out/pmc.log gpurun_out/r04_pmc_c2.txt
# stage clocks (diagnostic builds)
B="--timed device --no_cpu_baseline --no_host_buffers --no_extra_configs"
UIS_LIB_PATH=$PWD/build/variants/timing.so python bench.py $B --config 2 --steps 1 --warmup 0 2>&1 >/dev/null | grep "window launch timing" > gpurun_out/r04_window_launch_timing.txt
{
echo "--- configs[3] share: k_decode_big<WS>"
UIS_LIB_PATH=$PWD/build/variants/timing.so python bench.py $B --config 3 --steps 1 --warmup 0 2>&1 >/dev/null | grep "resident timing" | tail -4
echo "--- configs[4]: k_decode_resident (default)"
UIS_LIB_PATH=$PWD/build/variants/timing.so python bench.py $B --config 4 --steps 2 --warmup 1 2>&1 >/dev/null | grep "resident timing" | tail -4
} > gpurun_out/r04_resident_timing_b.txt 2>&1
UIS_NO_WINDOW_LAUNCH=1 UIS_LIB_PATH=$PWD/build/variants/seltiming.so python bench.py $B --config 2 --steps 1 --warmup 0 2>&1 >/dev/null | grep "window timing" | tail -2 > gpurun_out/r04_window_phases.txt
# soak
timeout 200 python tools/fuzz_gpu.py 150 77 > gpurun_out/r04_fuzz.txt 2>&1
timeout 300 python tools/stress_resident.py 100 >> gpurun_out/r04_fuzz.txt 2>&1
timeout 200 python tools/stress_persistent.py >> gpurun_out/r04_fuzz.txt 2>&1
python - <<'PY'
import json
d = json.load(open('gpurun_out/r04_bench_c1.json'))
print({k: d[k] for k in ('value', 'value_leg', 'value_predict_f64', 'value_host_buffers', 'value_device', 'ms_per_step')})
print(d['roofline']['kernel'], d['roofline']['avg_launch_us'], d['roofline']['frac'], d['roofline']['effective'], d['roofline']['traffic'])
print(d['cpu_baseline']['value'], d['cpu_baseline']['sample'])
for e in d['extra_configs'] or []:
    print({k: e.get(k) for k in ('config', 'value', 'kernel', 'frac', 'effective_frac', 'parity', 'error')})
for f in ('r04_bench_c2', 'r04_bench_c2_stepwise', 'r04_bench_c3', 'r04_bench_c4', 'r04_bench_c3_ragged'):
    try:
        d = json.load(open('gpurun_out/%s.json' % f))
        print(f, d['value'], d['value_leg'], d.get('value_device'), d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'])
    except Exception as e:
        print(f, 'ERR', e)
for r in json.load(open('gpurun_out/r04_usweep.json'))['sweep']:
    print(r)
PY
head -5 gpurun_out/r04_kernel_stats_c2.csv; head -6 gpurun_out/r04_kernel_stats_c2_stepwise.csv; cat gpurun_out/r04_window_launch_timing.txt gpurun_out/r04_resident_timing_b.txt gpurun_out/r04_window_phases.txt; grep -A12 "group 1" gpurun_out/r04_pmc_c2.txt | head -30; tail -4 gpurun_out/r04_fuzz.txt
