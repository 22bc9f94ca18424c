#!/bin/bash
# round 4: rnn_depth 2 with the upper layer's input gates on the LDS-weight kernel: parity + rate
mkdir -p gpurun_out
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -k "depth2_upper or golden or rnn_step" > gpurun_out/r04ae_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04ae_pytest.log
tail -8 gpurun_out/r04ae_pytest.log
sed -i 's/for n_utt in (64, 1024):/for n_utt in (1024,):/' /dev/null
cat > /tmp/prof.py <<'PY'
import sys, time, json
sys.path.insert(0, '.')
import numpy as np
from uisrnn_amd import _capi, synth
n_utt = 1024
params = synth.tracker_params(256, 512, 2, seed=0)
seqs, _ = synth.make_utterances(10_000, n_utt, 500, 256)
frames = np.concatenate(seqs).astype(np.float32)
offsets = np.concatenate([[0], np.cumsum([len(s) for s in seqs])]).astype(np.int64)
dec = _capi.Decoder(params)
out = dec.decode(frames, offsets, 10, 1, 2, max_clusters=16)
t0 = time.perf_counter(); out = dec.decode(frames, offsets, 10, 1, 2, max_clusters=16); dt = time.perf_counter() - t0
prof = dec.decode(frames, offsets, 10, 1, 2, max_clusters=16, flags=_capi.UIS_FLAG_PROFILE)['stats']
print(json.dumps({'utterances': n_utt, 'rnn_depth': 2, 'frames_per_s': round(n_utt * 500 / dt), 'us_per_decode_step': round(dt * 1e3, 2), 'kernel': prof['decode_kernel'],
                  'us_per_step': {k: round(1e3 * v / max(prof['n_steps'], 1), 2) for k, v in prof['kernel_ms'].items()}}))
PY
for env in "UIS_X=0" "UIS_WT_NO_UPPER=1"; do
  echo "== $env" | tee -a gpurun_out/r04ae_depth2.txt
  env $env python /tmp/prof.py 2>&1 | tail -1 | tee -a gpurun_out/r04ae_depth2.txt
done
