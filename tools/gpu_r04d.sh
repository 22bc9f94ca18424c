#!/bin/bash
# Round 4, fourth call: k_window with the digit-wise threshold search / one row reservation; the float64 leg
# with the persistent cast pool and finer copies.
mkdir -p gpurun_out
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r04d_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04d_pytest.log
tail -6 gpurun_out/r04d_pytest.log
B="--timed device --no_cpu_baseline --no_host_buffers --no_extra_configs"
{
echo "--- k_window phases, configs[2] (diagnostic build)"
UIS_LIB_PATH=$PWD/build/variants/seltiming.so python bench.py $B --config 2 --steps 1 --warmup 0 2>&1 >/dev/null | grep "window timing" | tail -2
echo "--- configs[2]"
python bench.py $B --config 2 --steps 3 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['kernel_ms_profile_pass'])"
echo "--- configs[1], all three legs"
for i in 1 2; do
python bench.py --no_cpu_baseline --no_extra_configs --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print({k: d[k] for k in ('value','value_leg','value_predict_f64','value_host_buffers','value_device','ms_per_step')})"
done
echo "--- float64 leg, one copy piece per projection chunk (as before)"
echo "(see the library variant build/variants/f64_1piece.so)"
[ -f build/variants/f64_1piece.so ] && UIS_LIB_PATH=$PWD/build/variants/f64_1piece.so python bench.py --no_cpu_baseline --no_extra_configs --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print({k: d[k] for k in ('value','value_leg','value_predict_f64','value_host_buffers','value_device','ms_per_step')})"
} 2>&1 | tee gpurun_out/r04d_numbers.txt
timeout 100 python tools/fuzz_gpu.py 70 11 > gpurun_out/r04d_fuzz.txt 2>&1; tail -3 gpurun_out/r04d_fuzz.txt
