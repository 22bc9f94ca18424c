#!/bin/bash
# Round 5, last GPU call: the batched weight-slice refills (lds_fill_512) -- GPU suite on the new library, the default
# bench line, then interleaved A/B against the previous build (build/variants/old.so = HEAD~ compiled from git archive).
#   gpurun --timeout 470 -- tools/experiments/r05_refill_ab.sh
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd "$REPO" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
Q="--no_cpu_baseline --no_host_buffers --no_extra_configs"
val() { python -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline'].get('frac'))" "$1" 2>/dev/null; }
echo "=== suite"; timeout 200 python -m pytest tests -m gpu -x -q > gpurun_out/refill_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/refill_pytest.log; tail -4 gpurun_out/refill_pytest.log
echo "=== default bench"; timeout 200 python bench.py > gpurun_out/refill_bench_c1.json 2> gpurun_out/refill_bench_c1.err; echo "rc=$? $(val gpurun_out/refill_bench_c1.json)"
python - <<'PY'
import json
try:
  d = json.loads(open('gpurun_out/refill_bench_c1.json').read().strip().splitlines()[-1])
  for e in d['extra_configs']:
    print('extra', e.get('config'), e.get('value'), e.get('parity'), e.get('error'))
except Exception as ex:
  print('no line', ex)
PY
: > gpurun_out/refill_ab.txt
for cfg in 3 2; do
  for rep in 1 2; do
    for lib in new old; do
      p=$REPO/uisrnn_amd/libuisrnn_hip.so; [ $lib = old ] && p=$REPO/build/variants/old.so
      UIS_LIB_PATH=$p UIS_BENCH_NO_PMC=1 timeout 120 python bench.py --config $cfg --timed device --steps 5 --warmup 1 $Q 2>/dev/null | tail -1 > gpurun_out/.ab.json
      echo "config=$cfg rep=$rep $lib $(val gpurun_out/.ab.json)" | tee -a gpurun_out/refill_ab.txt
    done
  done
done
echo "=== fuzz"; timeout 100 python tools/fuzz_gpu.py 30 4099 > gpurun_out/refill_fuzz.txt 2>&1; tail -2 gpurun_out/refill_fuzz.txt
