#!/bin/bash
# Round 4, third call: parity of the shape classes (opt-in ones through UIS_FLAG_REPLICATED_SELECT), then
# per-phase clocks: k_window at configs[2], the wide class against the owner-select kernel at configs[4].
mkdir -p gpurun_out
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r04c_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04c_pytest.log
tail -6 gpurun_out/r04c_pytest.log
B="--timed device --no_cpu_baseline --no_host_buffers --no_extra_configs"
{
echo "--- k_window phases, configs[2] (diagnostic build)"
UIS_LIB_PATH=$PWD/build/variants/seltiming.so python bench.py $B --config 2 --steps 1 --warmup 0 2>&1 >/dev/null | grep "window timing" | tail -2
echo "--- configs[4], wide class (flags 4096)"
UIS_LIB_PATH=$PWD/build/variants/timing.so python bench.py $B --config 4 --steps 2 --warmup 1 --flags 4096 2>&1 >/dev/null | grep "resident timing" | tail -4
echo "--- configs[4], owner select (default)"
UIS_LIB_PATH=$PWD/build/variants/timing.so python bench.py $B --config 4 --steps 2 --warmup 1 2>&1 >/dev/null | grep "resident timing" | tail -4
echo "--- configs[1], fixed-shape class"
UIS_LIB_PATH=$PWD/build/variants/timing.so python bench.py $B --steps 3 --warmup 1 2>&1 >/dev/null | grep "resident timing" | tail -4
echo "--- 128 utterances, two per wave (flags 4096)"
UIS_LIB_PATH=$PWD/build/variants/timing.so python bench.py $B --utterances 128 --steps 2 --warmup 1 --flags 4096 2>&1 >/dev/null | grep "resident timing" | tail -4
echo "--- 128 utterances, owner select"
UIS_LIB_PATH=$PWD/build/variants/timing.so python bench.py $B --utterances 128 --steps 2 --warmup 1 2>&1 >/dev/null | grep "resident timing" | tail -4
} 2>&1 | tee gpurun_out/r04c_timing.txt
timeout 150 python tools/fuzz_gpu.py 100 7 > gpurun_out/r04c_fuzz.txt 2>&1; tail -3 gpurun_out/r04c_fuzz.txt
