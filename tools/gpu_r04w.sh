#!/bin/bash
# round 4: hidden size 65 .. 128 on the cluster kernels (HP = 128 instantiations): parity, rates, more-utterances paths
mkdir -p gpurun_out
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -k "in_between or more_utterances or resident or small_models" > gpurun_out/r04w_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04w_pytest.log
tail -8 gpurun_out/r04w_pytest.log
cat > /tmp/rate.py <<'PY'
import sys, time, json
sys.path.insert(0, '.')
import numpy as np
from uisrnn_amd import _capi, synth
for dim, hid, n_utt in ((64, 100, 64), (100, 128, 64), (100, 128, 512)):
    params = synth.tracker_params(dim, hid, 1, seed=0)
    seqs, _ = synth.make_utterances(10_000, n_utt, 500, dim)
    frames = np.concatenate(seqs).astype(np.float32)
    offsets = np.concatenate([[0], np.cumsum([len(s) for s in seqs])]).astype(np.int64)
    dec = _capi.Decoder(params)
    out = dec.decode(frames, offsets, 10, 1, 2, max_clusters=16)
    t0 = time.perf_counter()
    for _ in range(3):
        out = dec.decode(frames, offsets, 10, 1, 2, max_clusters=16)
    dt = (time.perf_counter() - t0) / 3
    print(json.dumps({'observation_dim': dim, 'rnn_hidden_size': hid, 'utterances': n_utt, 'frames_per_s': round(n_utt * 500 / dt),
                      'ms_per_pass': round(dt * 1e3, 2), 'status': out['status'], 'kernel': out['stats']['decode_kernel']}))
PY
for env in "UIS_X=0" "UIS_PAD_TO_16_ONLY=1"; do
  echo "== $env" | tee -a gpurun_out/r04w_padded_shapes.txt
  env $env python /tmp/rate.py 2>&1 | tail -3 | tee -a gpurun_out/r04w_padded_shapes.txt
done
timeout 150 python tools/fuzz_gpu.py 100 123 > gpurun_out/r04w_fuzz.txt 2>&1; tail -2 gpurun_out/r04w_fuzz.txt
