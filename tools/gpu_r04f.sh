#!/bin/bash
# Round 4, sixth call: shape classes (compile-time beam / cap) for k_decode_resident and k_decode_big<WS>,
# A/B against the run-time instantiations (UIS_NO_SHAPE_CLASSES=1); k_window with its survivors' arrays in LDS.
mkdir -p gpurun_out
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r04f_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04f_pytest.log
tail -6 gpurun_out/r04f_pytest.log
B="--timed device --no_cpu_baseline --no_host_buffers --no_extra_configs"
one() { python bench.py $B "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['avg_launch_us'], d['roofline']['frac'], d['roofline']['effective']['frac'])"; }
{
for i in 1 2; do
echo "c4 class    $(one --config 4 --steps 5 --warmup 2)"
echo "c4 runtime  $(UIS_NO_SHAPE_CLASSES=1 one --config 4 --steps 5 --warmup 2)"
echo "U=128 class   $(one --utterances 128 --steps 5 --warmup 2)"
echo "U=128 runtime $(UIS_NO_SHAPE_CLASSES=1 one --utterances 128 --steps 5 --warmup 2)"
echo "c3 class    $(one --config 3 --steps 3 --warmup 1)"
echo "c3 runtime  $(UIS_NO_SHAPE_CLASSES=1 one --config 3 --steps 3 --warmup 1)"
done
echo "c1 $(one --steps 20 --warmup 5)"
echo "--- k_window phases, configs[2] (diagnostic build)"
UIS_LIB_PATH=$PWD/build/variants/seltiming.so python bench.py $B --config 2 --steps 1 --warmup 0 2>&1 >/dev/null | grep "window timing" | tail -2
echo "c2 $(one --config 2 --steps 3 --warmup 1)"
} 2>&1 | tee gpurun_out/r04f_numbers.txt
timeout 100 python tools/fuzz_gpu.py 70 17 > gpurun_out/r04f_fuzz.txt 2>&1; tail -3 gpurun_out/r04f_fuzz.txt
