"""Repeat the one-launch decode on the benchmark workload and compare every run bit for bit.

  python tools/stress_resident.py [runs] [observation_dim] [rnn_hidden_size]
"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from uisrnn_amd import _capi, synth
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 256
hid = int(sys.argv[3]) if len(sys.argv) > 3 else 512
params = synth.tracker_params(dim, hid, 1, seed=0)
seqs, _ = synth.make_utterances(10_000, 64, 500, dim)
frames = np.concatenate(seqs).astype(np.float32)
offsets = (np.arange(65) * 500).astype(np.int64)
dec = _capi.Decoder(params)
ref = dec.decode(frames, offsets, 10, 1, 2, want_beam_scores=True, flags=_capi.UIS_FLAG_STEPWISE)
bad = 0
t0 = time.time()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
for i in range(n):
  out = dec.decode(frames, offsets, 10, 1, 2, want_beam_scores=True, flags=_capi.UIS_FLAG_RESIDENT)
  same = (np.array_equal(out['labels'], ref['labels']) and
          np.array_equal(out['beam_scores'].view(np.uint32), ref['beam_scores'].view(np.uint32)))
  bad += not same
print('runs', n, 'mismatching', bad, 'seconds', round(time.time() - t0, 1))
# ragged, many utterances per XCD, interleaved with other work on the device
seqs2, _ = synth.make_utterances(20_000, 300, [10 + (7 * u) % 90 for u in range(300)], dim)
f2 = np.concatenate(seqs2).astype(np.float32)
o2 = np.concatenate([[0], np.cumsum([len(s) for s in seqs2])]).astype(np.int64)
r2 = dec.decode(f2, o2, 10, 1, 2, want_beam_scores=True, flags=_capi.UIS_FLAG_STEPWISE)
bad2 = 0
for i in range(max(n // 2, 1)):
  out = dec.decode(f2, o2, 10, 1, 2, want_beam_scores=True, flags=_capi.UIS_FLAG_RESIDENT)
  bad2 += not (np.array_equal(out['labels'], r2['labels']) and
               np.array_equal(out['beam_scores'].view(np.uint32), r2['beam_scores'].view(np.uint32)))
print('ragged runs', max(n // 2, 1), 'mismatching', bad2)
sys.exit(1 if bad or bad2 else 0)
