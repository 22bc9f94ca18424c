import sys, time
sys.path.insert(0, '.')
import numpy as np
from uisrnn_amd import _capi, synth
params = synth.tracker_params(256, 512, 1, seed=0)
seqs, _ = synth.make_utterances(10_000, 64, 500, 256)
frames = np.concatenate(seqs).astype(np.float32)
offsets = (np.arange(65) * 500).astype(np.int64)
dec = _capi.Decoder(params)
for _ in range(2): dec.decode(frames, offsets, 10, 1, 2)
ts = []
for _ in range(5):
  t0 = time.perf_counter(); out = dec.decode(frames, offsets, 10, 1, 2); ts.append(time.perf_counter() - t0)
print('host-buffer uis_decode: best %.2f ms wall (device part %.2f ms) -> %.0f frames/s PCIe-inclusive' % (min(ts) * 1e3, out['stats']['decode_ms'], 32000 / min(ts)))
