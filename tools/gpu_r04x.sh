#!/bin/bash
# round 4: configs[4] on k_decode_resident: control-word placement and shape class against the step time
mkdir -p gpurun_out
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
export TMPDIR=/tmp
run() {
  echo "== $*" | tee -a gpurun_out/r04x_c4.txt
  env "$@" timeout 300 python bench.py --config 4 --steps 3 --warmup 1 --timed device --no_host_buffers --no_extra_configs --no_cpu_baseline 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print({k: d.get(k) for k in ('value', 'ms_per_step', 'setup_passes')}, d['roofline']['kernel'], d['roofline']['avg_launch_us'])" | tee -a gpurun_out/r04x_c4.txt
}
run UIS_X=0
run UIS_CTL_OFFSET=0
run UIS_CTL_OFFSET=8192
run UIS_CTL_OFFSET=1048576
run UIS_CTL_OFFSET=1056768
run UIS_NO_SHAPE_CLASSES=1
UIS_LIB_PATH=$PWD/build/variants/timing.so python bench.py --timed device --no_cpu_baseline --no_host_buffers --no_extra_configs --config 4 --steps 3 --warmup 1 2>&1 | grep -E "resident timing\] workgroup|metric" | tail -7 | cut -c1-260 | tee -a gpurun_out/r04x_c4.txt
