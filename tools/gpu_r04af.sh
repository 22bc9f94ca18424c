#!/bin/bash
# round 4: launch-per-step region at 64 utterances: plain launches against hipGraph replay (UIS_FLAG_GRAPH)
mkdir -p gpurun_out
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
export TMPDIR=/tmp
cat > /tmp/rate.py <<'PY'
import sys, time, json
sys.path.insert(0, '.')
import numpy as np
from uisrnn_amd import _capi, synth
for dim, hid, depth in ((256, 512, 2), (256, 320, 1), (64, 96, 2)):
    params = synth.tracker_params(dim, hid, depth, seed=0)
    seqs, _ = synth.make_utterances(10_000, 64, 500, dim)
    frames = np.concatenate(seqs).astype(np.float32)
    offsets = np.concatenate([[0], np.cumsum([len(s) for s in seqs])]).astype(np.int64)
    dec = _capi.Decoder(params)
    for fl in (0, _capi.UIS_FLAG_GRAPH, 0, _capi.UIS_FLAG_GRAPH):
        out = dec.decode(frames, offsets, 10, 1, 2, max_clusters=16, flags=fl)
        t0 = time.perf_counter()
        for _ in range(3):
            out = dec.decode(frames, offsets, 10, 1, 2, max_clusters=16, flags=fl)
        dt = (time.perf_counter() - t0) / 3
        print(json.dumps({'observation_dim': dim, 'rnn_hidden_size': hid, 'rnn_depth': depth, 'graph': bool(fl), 'frames_per_s': round(64 * 500 / dt),
                          'us_per_decode_step': round(dt * 1e3, 2), 'status': out['status'], 'kernel': out['stats']['decode_kernel']}))
    dec.close()
PY
python /tmp/rate.py 2>&1 | tail -12 | tee gpurun_out/r04af_graph.txt
