#!/bin/bash
# Round 4, seventh call: the wide class of k_decode_rs again (after the descriptor / lp_new changes took its
# spills away) against the owner-select kernel's shape class at configs[4]; two per wave at 128 utterances.
mkdir -p gpurun_out
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r04g_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04g_pytest.log
tail -6 gpurun_out/r04g_pytest.log
B="--timed device --no_cpu_baseline --no_host_buffers --no_extra_configs"
one() { python bench.py $B "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['avg_launch_us'], d['roofline']['frac'], d['roofline']['effective']['frac'])"; }
{
for i in 1 2; do
echo "c4 owner select (class) $(one --config 4 --steps 5 --warmup 2)"
echo "c4 wide                 $(one --config 4 --steps 5 --warmup 2 --flags 4096)"
echo "c4 wide fixed shape     $(UIS_RS_WIDE_C4=1 one --config 4 --steps 5 --warmup 2 --flags 4096)"
done
echo "U=128 owner select      $(one --utterances 128 --steps 5 --warmup 2)"
echo "U=128 two per wave      $(one --utterances 128 --steps 5 --warmup 2 --flags 4096)"
echo "U=128 two per wave fixed $(UIS_RS_UPW2_C1=1 one --utterances 128 --steps 5 --warmup 2 --flags 4096)"
echo "U=65 owner select       $(one --utterances 65 --steps 5 --warmup 2)"
echo "U=65 two per wave       $(one --utterances 65 --steps 5 --warmup 2 --flags 4096)"
echo "c1 $(one --steps 20 --warmup 5)"
echo "c2 $(one --config 2 --steps 3 --warmup 1)"
} 2>&1 | tee gpurun_out/r04g_numbers.txt
