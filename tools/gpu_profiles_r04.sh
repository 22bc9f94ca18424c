#!/bin/bash
# Round-4 evidence, everything under gpurun_out/ (copied to profiles/ afterwards):
#   the default bench line (configs[1]: value = the float64-list leg, + extra_configs), the other configs on their
#   own, a U-sweep, ragged runs, the RCCL path with one rank, single-utterance latency, rocprofv3 kernel traces,
#   PMC passes (-> r04_traffic.json), per-phase clocks, streaming latency, a fuzz soak.
mkdir -p gpurun_out
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
export TMPDIR=/tmp
python bench.py > gpurun_out/r04_bench_c1.json 2> gpurun_out/r04_bench_c1.err
python bench.py --config 3 --no_cpu_baseline > gpurun_out/r04_bench_c3.json 2>/dev/null
python bench.py --ragged --no_cpu_baseline --no_extra_configs > gpurun_out/r04_bench_c1_ragged.json 2>/dev/null
python bench.py --config 3 --ragged --no_cpu_baseline --no_host_buffers > gpurun_out/r04_bench_c3_ragged.json 2>/dev/null
timeout 600 python bench.py --gpus 1 --force_dist --no_cpu_baseline --no_extra_configs \
  > gpurun_out/r04_force_dist.json 2> gpurun_out/r04_force_dist.err
echo "rc=$?" >> gpurun_out/r04_force_dist.err
python bench.py --utterances 1 --frames 1000 --timed device --no_cpu_baseline --no_extra_configs --no_host_buffers > gpurun_out/r04_latency_1utt.json 2>/dev/null
python bench.py --utterances 8 --frames 1000 --timed device --no_cpu_baseline --no_extra_configs --no_host_buffers > gpurun_out/r04_latency_8utt.json 2>/dev/null
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
BENCH_ARGS="--steps 10 --warmup 3 --no_cpu_baseline --no_extra_configs" ./tools/gpu_prof.sh > gpurun_out/r04_prof_stats.log 2>&1
cp gpurun_out/kernel_stats.csv gpurun_out/r04_kernel_stats_bench.csv
grep '"metric"' gpurun_out/prof_bench.log > gpurun_out/r04_bench_line_under_rocprof.json
for c in 2 3 4; do
  BENCH_ARGS="--config $c --steps 3 --warmup 1 --timed device --no_cpu_baseline --no_host_buffers" ./tools/gpu_prof.sh > /dev/null 2>&1
  cp gpurun_out/kernel_stats.csv gpurun_out/r04_kernel_stats_c$c.csv
done
# PMC passes (separate runs per counter group)
BENCH_ARGS="--no_extra_configs --no_host_buffers --timed device" ./tools/gpu_pmc.sh > /dev/null 2>&1
cp gpurun_out/pmc.log gpurun_out/r04_pmc_per_kernel.txt
python - <<'PY'
import csv, json, re
# FETCH_SIZE / WRITE_SIZE per kernel from the PMC log, launch duration from the kernel trace
txt = open('gpurun_out/r04_pmc_per_kernel.txt').read()
kern = {}
cur = None
for line in txt.splitlines():
    if line.startswith('==') or not line.strip():
        continue
    if not line.startswith(' '):
        cur = line.strip().replace('void ', '').split('<')[0]
        continue
    m = re.match(r'\s+(\S+)\s+n=\s*(\d+)\s+mean=\s*([0-9.eE+-]+)', line)
    if m and cur and m.group(1) in ('FETCH_SIZE', 'WRITE_SIZE'):
        kern.setdefault(cur, {})[m.group(1)] = float(m.group(3))
dur = {}
for r in csv.DictReader(open('gpurun_out/r04_kernel_stats_bench.csv')):
    dur[r['Name'].replace('void ', '').split('<')[0].split('(')[0]] = float(r['AverageNs']) / 1e3
out = {'source': 'profiles/r04_pmc_per_kernel.txt (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, bench.py --steps 1 '
                 '--warmup 0 --no_cpu_baseline --no_extra_configs --no_host_buffers --timed device); avg_launch_us from '
                 'profiles/r04_kernel_stats_bench.csv',
       'correction': 'FETCH_SIZE x2 on gfx950 for 16-byte-per-lane coalesced reads (MI355X_MICROARCH.md, HBM section); WRITE_SIZE '
                     'uncorrected; both in KiB per dispatch',
       'kernels': {}}
for k, v in kern.items():
    if 'FETCH_SIZE' in v and 'WRITE_SIZE' in v:
        out['kernels'][k] = {'fetch_size_kib': round(v['FETCH_SIZE'], 1), 'write_size_kib': round(v['WRITE_SIZE'], 1),
                             'avg_launch_us': round(dur[k], 1) if k in dur else None}
json.dump(out, open('gpurun_out/r04_traffic.json', 'w'), indent=1)
print(json.dumps(out['kernels']))
PY
# per-phase clocks (diagnostic builds)
B="--timed device --no_cpu_baseline --no_host_buffers --no_extra_configs"
{
echo "--- configs[1]: k_decode_rs, fixed-shape class"
UIS_LIB_PATH=$PWD/build/variants/timing.so python bench.py $B --steps 3 --warmup 1 2>&1 >/dev/null | grep "resident timing" | tail -4
echo "--- configs[1]: owner-select kernel (k_decode_resident, --flags 2048)"
UIS_LIB_PATH=$PWD/build/variants/timing.so python bench.py $B --steps 3 --warmup 1 --flags 2048 2>&1 >/dev/null | grep "resident timing" | tail -4
echo "--- configs[4]: k_decode_resident (default)"
UIS_LIB_PATH=$PWD/build/variants/timing.so python bench.py $B --config 4 --steps 2 --warmup 1 2>&1 >/dev/null | grep "resident timing" | tail -4
echo "--- configs[4]: k_decode_rs<wide> (--flags 4096)"
UIS_LIB_PATH=$PWD/build/variants/timing.so python bench.py $B --config 4 --steps 2 --warmup 1 --flags 4096 2>&1 >/dev/null | grep "resident timing" | tail -4
echo "--- configs[3] share: k_decode_big<WS>"
UIS_LIB_PATH=$PWD/build/variants/timing.so python bench.py $B --config 3 --steps 1 --warmup 0 2>&1 >/dev/null | grep "resident timing" | tail -4
} > gpurun_out/r04_resident_timing.txt 2>&1
UIS_LIB_PATH=$PWD/build/variants/seltiming.so python bench.py $B --config 2 --steps 1 --warmup 0 2>&1 >/dev/null | grep "window timing" | tail -2 > gpurun_out/r04_window_phases.txt
# online decoding latency
timeout 300 python tools/stream_latency.py > gpurun_out/r04_stream_latency.json 2> gpurun_out/r04_stream_latency.err
# soak
timeout 200 python tools/fuzz_gpu.py 150 > gpurun_out/r04_fuzz.txt 2>&1
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
for f in ('r04_bench_c3', 'r04_bench_c1_ragged', 'r04_bench_c3_ragged', 'r04_latency_1utt', 'r04_latency_8utt', 'r04_force_dist'):
    try:
        d = json.load(open('gpurun_out/%s.json' % f))
        print(f, d['value'], d['value_leg'], d.get('value_device'), d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'])
    except Exception as e:
        print(f, 'ERR', e)
for r in json.load(open('gpurun_out/r04_usweep.json'))['sweep']:
    print(r)
PY
head -6 gpurun_out/r04_kernel_stats_bench.csv; cat gpurun_out/r04_resident_timing.txt gpurun_out/r04_window_phases.txt; tail -4 gpurun_out/r04_fuzz.txt
