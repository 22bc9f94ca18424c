#!/bin/bash
# round 4: what the launch-per-step region costs -- rnn_depth 2 at hidden size 512, hidden size 320 (no one-launch kernel)
mkdir -p gpurun_out
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
export TMPDIR=/tmp
cat > /tmp/rate.py <<'PY'
import sys, time, json
sys.path.insert(0, '.')
import numpy as np
from uisrnn_amd import _capi, synth
for dim, hid, depth, n_utt, fl in ((256, 512, 1, 64, 0), (256, 512, 1, 64, _capi.UIS_FLAG_STEPWISE), (256, 512, 2, 64, 0), (256, 320, 1, 64, 0),
                                   (256, 512, 2, 1024, 0), (256, 512, 1, 1024, _capi.UIS_FLAG_STEPWISE)):
    params = synth.tracker_params(dim, hid, depth, seed=0)
    seqs, _ = synth.make_utterances(10_000, n_utt, 500, dim)
    frames = np.concatenate(seqs).astype(np.float32)
    offsets = np.concatenate([[0], np.cumsum([len(s) for s in seqs])]).astype(np.int64)
    dec = _capi.Decoder(params)
    out = dec.decode(frames, offsets, 10, 1, 2, max_clusters=16, flags=fl)
    t0 = time.perf_counter()
    for _ in range(2):
        out = dec.decode(frames, offsets, 10, 1, 2, max_clusters=16, flags=fl)
    dt = (time.perf_counter() - t0) / 2
    print(json.dumps({'observation_dim': dim, 'rnn_hidden_size': hid, 'rnn_depth': depth, 'utterances': n_utt, 'frames_per_s': round(n_utt * 500 / dt),
                      'us_per_decode_step': round(dt * 1e6 / 1000, 2), 'status': out['status'], 'kernel': out['stats']['decode_kernel']}))
    dec.close()
PY
python /tmp/rate.py 2>&1 | tail -6 | tee gpurun_out/r04ac_stepwise_region.txt
