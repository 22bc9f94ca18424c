#!/bin/bash
# round 4: rocprofv3 kernel trace + bench line of a depth-2 decode (k_decode_deep), 1024 and 64 utterances
mkdir -p gpurun_out
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
export TMPDIR=/tmp
python bench.py --rnn_depth 2 --utterances 1024 --timed device --steps 3 --warmup 1 --no_cpu_baseline --no_host_buffers --no_extra_configs > gpurun_out/r04_bench_depth2_1024.json 2>/dev/null
python bench.py --rnn_depth 2 --timed device --steps 5 --warmup 2 --no_host_buffers --no_extra_configs > gpurun_out/r04_bench_depth2_64.json 2>/dev/null
BENCH_ARGS="--rnn_depth 2 --utterances 1024 --steps 3 --warmup 1 --timed device --no_cpu_baseline --no_host_buffers --no_extra_configs" ./tools/gpu_prof.sh > /dev/null 2>&1
cp gpurun_out/kernel_stats.csv gpurun_out/r04_kernel_stats_depth2.csv
head -5 gpurun_out/r04_kernel_stats_depth2.csv
python - <<'PY'
import json
for f in ('r04_bench_depth2_1024', 'r04_bench_depth2_64'):
    d = json.load(open('gpurun_out/%s.json' % f))
    print(f, d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['effective'], (d.get('cpu_baseline') or {}).get('sample'))
PY
