#!/bin/bash
# Round 4, second call: the generalised k_decode_rs (shape classes) -- parity, then A/B numbers.
mkdir -p gpurun_out
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r04b_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04b_pytest.log
tail -15 gpurun_out/r04b_pytest.log
B="--timed device --no_cpu_baseline --no_host_buffers --no_extra_configs"
one() { python bench.py $B "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['effective']['frac'])"; }
{
for i in 1 2; do
echo "c1 fixed-shape   $(one --steps 10 --warmup 3)"
echo "c1 generic       $(UIS_RS_NO_C1=1 one --steps 10 --warmup 3)"
done
echo "c4 wide          $(one --config 4 --steps 5 --warmup 2)"
echo "c4 wide fixed    $(UIS_RS_WIDE_C4=1 one --config 4 --steps 5 --warmup 2)"
echo "c4 resident      $(one --config 4 --steps 5 --warmup 2 --flags 2048)"
for U in 65 96 128; do
echo "U=$U upw2         $(one --utterances $U --steps 5 --warmup 2)"
echo "U=$U upw2 fixed   $(UIS_RS_UPW2_C1=1 one --utterances $U --steps 5 --warmup 2)"
echo "U=$U resident     $(one --utterances $U --steps 5 --warmup 2 --flags 2048)"
echo "U=$U big<WS>      $(UIS_BIG_MIN_U=65 one --utterances $U --steps 5 --warmup 2 --flags 2048)"
done
for U in 192 256; do
echo "U=$U resident     $(one --utterances $U --steps 5 --warmup 2)"
echo "U=$U big<WS>      $(UIS_BIG_MIN_U=65 one --utterances $U --steps 5 --warmup 2)"
done
echo "c2 1 stream      $(one --config 2 --steps 2 --warmup 1)"
echo "c2 2 streams     $(one --config 2 --steps 2 --warmup 1 --streams 2)"
echo "c2 4 streams     $(one --config 2 --steps 2 --warmup 1 --streams 4)"
echo "c3 uniform       $(one --config 3 --steps 2 --warmup 1)"
echo "c3 ragged        $(one --config 3 --steps 2 --warmup 1 --ragged)"
} 2>&1 | tee gpurun_out/r04b_ab.txt
timeout 200 python tools/fuzz_gpu.py 60 > gpurun_out/r04b_fuzz.txt 2>&1; tail -3 gpurun_out/r04b_fuzz.txt
