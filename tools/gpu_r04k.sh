#!/bin/bash
# Round 4: k_decode_small (one workgroup per utterance, small models of any depth).
mkdir -p gpurun_out
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r04k_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04k_pytest.log
tail -25 gpurun_out/r04k_pytest.log
python - <<'PY' 2>&1 | tee gpurun_out/r04k_small.txt
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import golden_util
from uisrnn_amd import _capi, weights
from oracle import oracle
# frames/s of small models: one launch against four-plus launches per step
def run(params, seqs, beam, tau, flags, reps=5):
  dec = _capi.Decoder(params)
  frames, offsets = oracle.pack(seqs)
  out = dec.decode(frames, offsets, beam, 1, tau, flags=flags)
  t0 = time.perf_counter()
  for _ in range(reps):
    out = dec.decode(frames, offsets, beam, 1, tau, flags=flags)
  dt = (time.perf_counter() - t0) / reps
  return out, dt
rng = np.random.default_rng(1)
for name, (dim, hid, depth) in (('hidden 8, depth 1 (tests/uisrnn_test.py)', (16, 8, 1)), ('hidden 8, depth 2 (tests/integration_test.py)', (2, 8, 2)),
                                ('hidden 24, depth 3', (20, 24, 3)), ('hidden 100, depth 2, dim 100', (100, 100, 2))):
  params = weights.init_params(dim, hid, depth, sigma2=0.1, transition_bias=0.2, crp_alpha=1.0, seed=3)
  cents = rng.standard_normal((3, dim))
  for n_utt in (1, 64, 512):
    seqs = [(cents[np.repeat(rng.integers(0, 3, size=500 // 4 + 1), 4)[:500]] * 0.4 + 0.1 * rng.standard_normal((500, dim))) for _ in range(n_utt)]
    a, ta = run(params, seqs, 10, 2, 0)
    b, tb = run(params, seqs, 10, 2, _capi.UIS_FLAG_STEPWISE, reps=2)
    same = np.array_equal(a['labels'], b['labels'])
    print('%-48s %4d utt x 500: %-16s %8.2f ms (%.2f us/step, %9.0f frames/s) | launch per step %8.2f ms (%.2f us/step) | x%.1f same=%s' % (
        name, n_utt, a['stats']['decode_kernel'], 1e3 * ta, 1e3 * ta, n_utt * 500 / ta, 1e3 * tb, 1e3 * tb, tb / ta, same))
PY
timeout 100 python tools/fuzz_gpu.py 70 29 > gpurun_out/r04k_fuzz.txt 2>&1; tail -2 gpurun_out/r04k_fuzz.txt
