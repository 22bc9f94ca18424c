#!/bin/bash
# round 4: k_decode_deep: full GPU suite, fuzz (small-branch models of depth 2-3 with hidden 65..99 land on it), rates
mkdir -p gpurun_out
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
export TMPDIR=/tmp
timeout 1300 python -m pytest tests -m gpu -x -q > gpurun_out/r04ai_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04ai_pytest.log
tail -6 gpurun_out/r04ai_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout 200 python tools/fuzz_gpu.py 120 4242 > gpurun_out/r04ai_fuzz.txt 2>&1; tail -3 gpurun_out/r04ai_fuzz.txt
timeout 300 python /dev/stdin <<'PY' | tee gpurun_out/r04ai_deep.txt
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
