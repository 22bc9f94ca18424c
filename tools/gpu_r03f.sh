#!/bin/bash
# A/B of library variants on the owner-select kernels: configs[1] with --flags 2048, configs[4], streaming pushes
mkdir -p gpurun_out
for i in 1 2; do
  for v in base ${VARIANTS:-}; do
    if [ $v = base ]; then L=""; else L="UIS_LIB_PATH=$PWD/build/variants/$v.so"; fi
    a=$(env $L python bench.py --flags 2048 --steps 10 --warmup 3 --no_cpu_baseline --no_host_buffers --no_extra_configs 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.readlines()[-1])['value'])")
    b=$(env $L python bench.py --config 4 --steps 5 --warmup 2 --no_cpu_baseline --no_host_buffers 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.readlines()[-1])['value'])")
    c=$(env $L python bench.py --flags 128 --steps 5 --warmup 2 --no_cpu_baseline --no_host_buffers --no_extra_configs 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.readlines()[-1])['value'])")
    echo "$v $i owner-select $a config4 $b stepwise $c"
  done
done
