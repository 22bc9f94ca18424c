#!/bin/bash
# A/B of library variants on the PCIe-inclusive rates (value_host_buffers, value_predict_f64)
mkdir -p gpurun_out
for i in $(seq 1 ${ROUNDS:-2}); do
  for v in base ${VARIANTS:-}; do
    if [ $v = base ]; then L=""; else L="UIS_LIB_PATH=$PWD/build/variants/$v.so"; fi
    env $L python bench.py --steps 10 --warmup 3 --no_cpu_baseline --no_extra_configs > gpurun_out/r03e_${v}_$i.json 2>/dev/null
    python -c "import json; d=json.load(open('gpurun_out/r03e_${v}_$i.json')); print('$v', $i, d['value'], d['value_host_buffers'], d['value_predict_f64'])"
  done
done
