#!/bin/bash
# round 4: full GPU suite + fuzz (look-ahead on the one-launch path) + look-ahead with many utterances, one launch against per sub-step
mkdir -p gpurun_out
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
export TMPDIR=/tmp
timeout 1300 python -m pytest tests -m gpu -x -q > gpurun_out/r04p_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04p_pytest.log
tail -6 gpurun_out/r04p_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout 260 python tools/fuzz_gpu.py 150 41 > gpurun_out/r04p_fuzz.txt 2>&1; tail -3 gpurun_out/r04p_fuzz.txt
for env in "" "UIS_NO_WINDOW_LAUNCH=1"; do
  echo "== 1024 utt x 200 frames, beam 10, look_ahead 2 $env" | tee -a gpurun_out/r04p_many.txt
  env $env timeout 300 python bench.py --config 2 --utterances 1024 --frames 200 --beam_size 10 --steps 2 --warmup 1 --no_extra_configs --no_cpu_baseline --timed device --no_host_buffers 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print({k: d.get(k) for k in ('value', 'ms_per_step')}, d.get('roofline', {}).get('kernel'))" | tee -a gpurun_out/r04p_many.txt
done
