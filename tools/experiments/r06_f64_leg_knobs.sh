#!/bin/bash
# the float64-list leg (bench.py's `value`) at configs[1] / [4] / the configs[3] share under slice-schedule and cast-team knobs
cd ${GRAFT_REPO_ROOT:-/root/repo}
export UIS_BENCH_NO_PMC=1
for cfg in "" "--config 4" "--config 3"; do
for rep in 1 2; do
for e in "-" "UIS_SPLIT_FRAMES=32" "UIS_NO_SPLIT=1" $EXTRA_ENVS; do
  ee=$e; [ "$e" = "-" ] && ee=""
  v=$(env $ee timeout 300 python bench.py $cfg --no_cpu_baseline --no_extra_configs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['value_host_buffers'], d['value_device'], d['ms_per_step'])")
  echo "[$cfg] rep=$rep [$e] $v"
done
done
done
