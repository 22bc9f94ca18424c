#!/bin/bash
# Round 4, fifth call: k_decode_rs with descriptor addressing / exact dims / host-side lp_new; k_window
# scoring per hypothesis.
mkdir -p gpurun_out
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r04e_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04e_pytest.log
tail -6 gpurun_out/r04e_pytest.log
B="--timed device --no_cpu_baseline --no_host_buffers --no_extra_configs"
one() { python bench.py $B "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['avg_launch_us'], d['roofline']['frac'], d['roofline']['effective']['frac'])"; }
{
for i in 1 2 3; do echo "c1 $(one --steps 20 --warmup 5)"; done
echo "c1 generic class $(UIS_RS_NO_C1=1 one --steps 10 --warmup 3)"
echo "c4 $(one --config 4 --steps 5 --warmup 2)"
echo "U=128 $(one --utterances 128 --steps 5 --warmup 2)"
echo "--- k_window phases, configs[2] (diagnostic build)"
UIS_LIB_PATH=$PWD/build/variants/seltiming.so python bench.py $B --config 2 --steps 1 --warmup 0 2>&1 >/dev/null | grep "window timing" | tail -2
echo "c2 $(one --config 2 --steps 3 --warmup 1)"
} 2>&1 | tee gpurun_out/r04e_numbers.txt
timeout 100 python tools/fuzz_gpu.py 70 13 > gpurun_out/r04e_fuzz.txt 2>&1; tail -3 gpurun_out/r04e_fuzz.txt
