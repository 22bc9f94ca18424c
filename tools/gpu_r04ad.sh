#!/bin/bash
# round 4: rnn_depth 2 at hidden size 512 on the launch-per-step path: where the time goes
mkdir -p gpurun_out
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
export TMPDIR=/tmp
cat > /tmp/prof.py <<'PY'
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
    out = dec.decode(frames, offsets, 10, 1, 2, max_clusters=16)
    out = dec.decode(frames, offsets, 10, 1, 2, max_clusters=16, flags=_capi.UIS_FLAG_PROFILE)
    st = out['stats']
    print(json.dumps({'utterances': n_utt, 'kernel': st['decode_kernel'], 'n_steps': st['n_steps'], 'rnn_rows': st['rnn_rows'],
                      'us_per_step': {k: round(1e3 * v / max(st['n_steps'], 1), 2) for k, v in st['kernel_ms'].items()},
                      'launches': st['kernel_launches']}))
    dec.close()
PY
python /tmp/prof.py 2>&1 | tail -2 | tee gpurun_out/r04ad_depth2_profile.txt
