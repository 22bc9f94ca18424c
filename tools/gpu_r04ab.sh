#!/bin/bash
# round 4: k_decode_small with a window sub-step as its select (small models, look_ahead >= 2): parity + rates
mkdir -p gpurun_out
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
export TMPDIR=/tmp
cat > /tmp/rate.py <<'PY'
import sys, time, json
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
import numpy as np
from uisrnn_amd import _capi
import golden_util
case = golden_util.load_case('d32_lookahead3')
rng = np.random.default_rng(3)
dim = case['params']['observation_dim']
cents = rng.standard_normal((3, dim))
seqs = [(cents[np.repeat(rng.integers(0, 3, size=60), 4)[:200]] * 0.4 + 0.1 * rng.standard_normal((200, dim))) for _ in range(64)]
frames = np.concatenate(seqs).astype(np.float32)
offsets = np.concatenate([[0], np.cumsum([len(s) for s in seqs])]).astype(np.int64)
dec = _capi.Decoder(case['params'])
for look, beam in ((2, 5), (3, 3)):
    for fl in (0, _capi.UIS_FLAG_STEPWISE):
        out = dec.decode(frames, offsets, beam, look, 1, max_clusters=24, flags=fl)
        t0 = time.perf_counter()
        for _ in range(3):
            out = dec.decode(frames, offsets, beam, look, 1, max_clusters=24, flags=fl)
        dt = (time.perf_counter() - t0) / 3
        print(json.dumps({'model': 'd32_lookahead3 (hidden %d, depth %d)' % (case['params']['rnn_hidden_size'], case['params']['rnn_depth']),
                          'look_ahead': look, 'beam': beam, 'frames_per_s': round(64 * 200 / dt), 'us_per_sub_step': round(dt * 1e6 / 200, 2),
                          'status': out['status'], 'kernel': out['stats']['decode_kernel']}))
PY
python /tmp/rate.py 2>&1 | tail -4 | tee gpurun_out/r04ab_small_lookahead.txt
