"""How much would dealing configs[4]'s utterances to the XCDs by their row counts gain?  (round 5, the verdict's item 5)

An upper bound, measured: the rows every utterance emits (each decoded alone: uis_stats.rnn_rows), then the same list
in three orders -- as given (utterance u on XCD u % 8), dealt longest-processing-time-first by the rows of the WHOLE
utterance (knowledge nobody has before decoding), and dealt by the rows of its first 32 frames (what a first launch of
a decode in several launches could know).  Frames/s from the library's own clock, best of five.

  gpurun -- python tools/experiments/redeal_bound.py > gpurun_out/redeal_bound.txt
"""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from uisrnn_amd import _capi, synth, weights  # pylint: disable=wrong-import-position

os.environ['UIS_NO_SPLIT'] = '1'
N_UTT, N_FRAMES, DIM, BEAM, CAP, NCL = 64, 500, 512, 20, 11, 8
params = weights.load_checkpoint(os.path.join(ROOT, 'tests', 'golden', 'trained_d512.uisrnn'))
seqs, _ = synth.make_utterances(10_000, N_UTT, N_FRAMES, DIM)
dec = _capi.Decoder(params)


def pack(order, n=None):
  frames = np.concatenate([seqs[u][:n] for u in order], axis=0).astype(np.float32)
  lens = [seqs[u][:n].shape[0] for u in order]
  return frames, np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)


def rows_alone(n):
  out = []
  for u in range(N_UTT):
    r = dec.decode(*pack([u], n), BEAM, 1, 2, max_clusters=CAP)
    assert r['status'] == 0
    out.append(r['stats']['rnn_rows'])
  return np.array(out, dtype=np.float64)


def lpt(weight):
  """position p = cluster + NCL * rank; at most N_UTT / NCL utterances per cluster, heaviest first to the lightest cluster"""
  load, count, order = np.zeros(NCL), np.zeros(NCL, dtype=int), [None] * N_UTT
  for u in np.argsort(-weight, kind='stable'):
    c = min((c for c in range(NCL) if count[c] < N_UTT // NCL), key=lambda c: load[c])
    order[c + NCL * count[c]] = int(u)
    load[c] += weight[u]
    count[c] += 1
  return order


def rate(order):
  frames, offsets = pack(order)
  best, labels = 1e9, None
  for _ in range(5):
    r = dec.decode(frames, offsets, BEAM, 1, 2, max_clusters=CAP)
    assert r['status'] == 0 and r['stats']['decode_kernel'] == 'k_decode_resident', r['stats']['decode_kernel']
    best, labels = min(best, r['stats']['decode_ms']), r['labels']
  return N_UTT * N_FRAMES / best * 1e3, labels, offsets


full, head = rows_alone(None), rows_alone(32)
print('rows per utterance, whole: min %d  mean %.0f  max %d;  first 32 frames: min %d mean %.0f max %d;  correlation %.3f'
      % (full.min(), full.mean(), full.max(), head.min(), head.mean(), head.max(), np.corrcoef(full, head)[0, 1]))
orders = {'as given': list(range(N_UTT)), 'by the whole utterance\'s rows': lpt(full), 'by the first 32 frames\' rows': lpt(head)}
base_labels = None
for name, order in orders.items():
  load = np.zeros(NCL)
  for p, u in enumerate(order):
    load[p % NCL] += full[u]
  fps, labels, offsets = rate(order)
  per_utt = {u: labels[offsets[k]:offsets[k + 1]] for k, u in enumerate(order)}
  if base_labels is None:
    base_labels = per_utt
  same = all(np.array_equal(per_utt[u], base_labels[u]) for u in range(N_UTT))
  print('%-32s heaviest XCD / mean rows %.3f   %.4f M frames/s   labels as before: %s' % (name, load.max() / load.mean(), fps / 1e6, same))
