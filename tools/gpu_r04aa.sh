#!/bin/bash
# round 4: the library built for gfx950:xnack- against the default target, interleaved
mkdir -p gpurun_out
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
export TMPDIR=/tmp
for rep in 1 2; do
for lib in uisrnn_amd/libuisrnn_hip.so build/variants/xnack_off.so; do
  for cfg in 1 2 4; do
  echo "== $lib config $cfg" | tee -a gpurun_out/r04aa_xnack.txt
  UIS_LIB_PATH=$PWD/$lib timeout 300 python bench.py --config $cfg --steps 5 --warmup 2 --timed device --no_host_buffers --no_cpu_baseline --no_extra_configs 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print({k: d.get(k) for k in ('value', 'ms_per_step')}, d['roofline']['kernel'], d['roofline']['avg_launch_us'])" | tee -a gpurun_out/r04aa_xnack.txt
  done
done
done
