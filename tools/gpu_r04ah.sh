#!/bin/bash
# round 4: k_decode_deep (rnn_depth >= 2 in one launch): parity, rates against the launch-per-step path
mkdir -p gpurun_out
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q -k "deep_models or depth2_upper or small_models or rnn_step" > gpurun_out/r04ah_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04ah_pytest.log
tail -15 gpurun_out/r04ah_pytest.log
cat > /tmp/rate.py <<'PY'
import sys, time, json
sys.path.insert(0, '.')
import numpy as np
from uisrnn_amd import _capi, synth
for n_utt in (64, 1024):
    params = synth.tracker_params(256, 512, 2, seed=0)
    seqs, _ = synth.make_utterances(10_000, n_utt, 500, 256)
    frames = np.concatenate(seqs).astype(np.float32)
    offsets = np.concatenate([[0], np.cumsum([len(s) for s in seqs])]).astype(np.int64)
    dec = _capi.Decoder(params)
    for fl in (0, _capi.UIS_FLAG_STEPWISE):
        out = dec.decode(frames, offsets, 10, 1, 2, max_clusters=16, flags=fl)
        t0 = time.perf_counter(); out = dec.decode(frames, offsets, 10, 1, 2, max_clusters=16, flags=fl); dt = time.perf_counter() - t0
        print(json.dumps({'utterances': n_utt, 'rnn_depth': 2, 'frames_per_s': round(n_utt * 500 / dt), 'us_per_decode_step': round(dt * 1e3, 2),
                          'status': out['status'], 'kernel': out['stats']['decode_kernel']}))
    dec.close()
PY
timeout 300 python /tmp/rate.py 2>&1 | tail -4 | tee gpurun_out/r04ah_deep.txt
