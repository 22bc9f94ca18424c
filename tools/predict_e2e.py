"""The Python surface end to end: UISRNN.predict on a list of float64 arrays (what a user of the
reference passes, uisrnn/uisrnn.py:564-590) -> lists of ints, wall clock per call, next to the
same workload through the lower entry points.  Workload = BASELINE configs[1] (64 x 500 x 256).

  python tools/predict_e2e.py [n_utt=64] [n_frames=500] [reps=8]
"""
import sys
import time
sys.path.insert(0, '.')
import numpy as np
import uisrnn_amd
from uisrnn_amd import synth, weights

n_utt = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n_frames = int(sys.argv[2]) if len(sys.argv) > 2 else 500
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
model_args, _, inference_args = uisrnn_amd.parse_arguments([])
model = uisrnn_amd.UISRNN(model_args)
model.load_params(weights.load_checkpoint('tests/golden/trained_d256.uisrnn'))
seqs, _ = synth.make_utterances(10_000, n_utt, n_frames, 256)
total = n_utt * n_frames


def best(fn):
  ts = []
  for _ in range(reps):
    t0 = time.perf_counter()
    fn()
    ts.append(time.perf_counter() - t0)
  return min(ts[2:]) if len(ts) > 2 else min(ts)  # (the first calls grow buffers and try placements)

dec = model._get_decoder(None)  # pylint: disable=protected-access
frames32 = np.concatenate(seqs).astype(np.float32)
offsets = (np.arange(n_utt + 1) * n_frames).astype(np.int64)
ba, la, ti = inference_args.beam_size, inference_args.look_ahead, inference_args.test_iteration
t_predict = best(lambda: model.predict(seqs, inference_args))
t_f64 = best(lambda: dec.decode_f64(seqs, ba, la, ti))
t_cast = best(lambda: np.concatenate(seqs).astype(np.float32))
out = np.empty((total, 256), dtype=np.float32)
def cast_copy():
  o = 0
  for s in seqs:
    out[o:o + len(s)] = s
    o += len(s)
t_cast2 = best(cast_copy)
t_f32 = best(lambda: dec.decode(frames32, offsets, ba, la, ti))
for name, t in (('UISRNN.predict (float64 list -> int lists)', t_predict),
                ('uis_decode_f64 (float64 list -> packed int32)', t_f64),
                ('uis_decode (caller-cast float32, pageable)', t_f32),
                ('numpy cast alone: concatenate + astype', t_cast),
                ('numpy cast alone: slice assignment', t_cast2)):
  print('{:52s} {:8.2f} ms  {:10.0f} frames/s'.format(name, 1e3 * t, total / t), flush=True)
