#!/bin/bash
# A/B on one box: workspace arena on/off, default (launch-per-step) and resident decode
for rep in 1 2; do
  for flags in 0 64; do
    for arena in 1 0; do
      if [ $arena = 0 ]; then export UIS_NO_ARENA=1; else unset UIS_NO_ARENA; fi
      v=$(timeout 120 python bench.py --steps 5 --warmup 1 --no_cpu_baseline --flags $flags | grep -o '"value": [0-9.]*')
      echo "rep=$rep flags=$flags arena=$arena $v"
    done
  done
done
