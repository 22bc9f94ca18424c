"""Record what the REFERENCE's own functions return, function by function.

Runs only in the dev container (it imports /root/reference through make_golden.import_reference);
writes two committed fixtures:

  fn_scores.npz UISRNN._calculate_score (uisrnn/uisrnn.py:455-477): for a few small decodes every
               array it returns -- one per (window, beam hypothesis) -- placed in predict_single's
               padded score_set (uisrnn.py:534-545, +inf outside), dense [windows, beam_size, cmax
               (, cmax)].  Several dozen BeamStates per case, fresh clusters, +inf padding of
               hypotheses with fewer clusters and of beam rows past the live beam included.
  fn_evals.npz  evals.compute_sequence_match_accuracy (uisrnn/evals.py:40-73) on 200 random label
               sequence pairs (1..64 distinct ids, lengths 1..400) + the reference's known answers.

  python tests/golden/make_scores.py
"""

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

import make_golden  # pylint: disable=wrong-import-position

# (fixture of make_golden.py whose model / utterances are reused, utterance, frames kept,
#  beam_size, look_ahead, test_iteration, cmax)
SCORE_CASES = [
    ('tiny_d16', 0, 14, 4, 1, 2, 8),
    ('tiny_d16', 5, 1, 3, 1, 1, 4),        # the single-frame utterance: one window, one hypothesis
    ('d32_lookahead3', 2, 9, 3, 2, 1, 8),  # look_ahead 2 with a ragged last window
    ('d20_h24_depth3', 0, 10, 5, 1, 2, 8),
    ('tracker_d256', 0, 16, 10, 1, 2, 8),  # the benchmark shape (D = 256, H = 512, beam 10)
    # BASELINE configs[2]'s shape: the model the reference trained (make_trained.py), beam 50,
    # look_ahead 2, max_clusters 12 -> arrays [windows, 50, 13, 13]; 8 frames x test_iteration 2
    ('trained_d256_l2_n40', 1, 8, 50, 2, 2, 13),
]


def record_scores(uisrnn, params, seq, beam_size, look_ahead, test_iteration, cmax):
  """predict_single with _calculate_score wrapped: dense [windows, beam, cmax (, cmax)]."""
  model, inference_args = make_golden.reference_model(uisrnn, params)
  inference_args.beam_size = beam_size
  inference_args.look_ahead = look_ahead
  inference_args.test_iteration = test_iteration
  total = test_iteration * seq.shape[0]
  n_win = (total + look_ahead - 1) // look_ahead
  out = np.full([n_win, beam_size] + [cmax] * look_ahead, np.inf, dtype=np.float32)
  state = {'calls': 0, 'win': 0, 'rank': 0, 'in_score': False, 'replayed': False}
  orig = model._calculate_score  # pylint: disable=protected-access
  orig_update = model._update_beam_state  # pylint: disable=protected-access

  def update(beam_state, look_ahead_seq, cluster_seq):
    # outside _calculate_score this is the replay of a window's winners (uisrnn.py:551-559): the
    # next _calculate_score call opens the next window
    if not state['in_score']:
      state['replayed'] = True
    return orig_update(beam_state, look_ahead_seq, cluster_seq)

  def score(beam_state, look_ahead_seq):
    if state['replayed']:
      state['win'] += 1
      state['rank'] = 0
      state['replayed'] = False
    state['in_score'] = True
    try:
      arr = orig(beam_state, look_ahead_seq)
    finally:
      state['in_score'] = False
    a = np.asarray(arr, dtype=np.float32)
    assert all(d <= cmax for d in a.shape), (a.shape, cmax)
    # (a ragged last window returns fewer dimensions: its scores sit at index 0 of the missing ones)
    idx = (state['win'], state['rank']) + tuple(slice(0, d) for d in a.shape) + (0,) * (look_ahead - a.ndim)
    out[idx] = a
    state['rank'] += 1
    state['calls'] += 1
    return arr

  model._update_beam_state = update  # pylint: disable=protected-access
  model._calculate_score = score  # pylint: disable=protected-access
  labels = model.predict(seq, inference_args)
  assert state['win'] == n_win - 1, (state['win'], n_win)
  return out, np.array([int(x) for x in labels], dtype=np.int32), state['calls']


def main():
  import golden_util  # pylint: disable=import-outside-toplevel
  uisrnn = make_golden.import_reference()
  store = {'n_cases': np.int64(len(SCORE_CASES))}
  for i, (name, utt, keep, beam, look, tau, cmax) in enumerate(SCORE_CASES):
    case = golden_util.load_trained(name) if name.startswith('trained_') else golden_util.load_case(name)
    seq = np.asarray(case['seqs'][utt], dtype=np.float64)[:keep]
    arr, labels, calls = record_scores(uisrnn, case['params'], seq, beam, look, tau, cmax)
    store['case_{}'.format(i)] = np.array([name], dtype='U32')
    store['cfg_{}'.format(i)] = np.array([utt, keep, beam, look, tau, cmax], dtype=np.int64)
    store['scores_{}'.format(i)] = arr
    store['labels_{}'.format(i)] = labels
    print('{}: utterance {} x {} frames, beam {}, look_ahead {}, test_iteration {}: {} _calculate_score calls, '
          '{} finite scores'.format(name, utt, keep, beam, look, tau, calls, int(np.isfinite(arr).sum())))
  np.savez_compressed(os.path.join(HERE, 'fn_scores.npz'), **store)

  # ---- evals.compute_sequence_match_accuracy
  from uisrnn import evals  # pylint: disable=import-outside-toplevel
  rng = np.random.default_rng(77)
  pairs = [([0, 0, 1, 2, 2], [1, 1, 0, 2, 0]), ([0, 0, 1, 1], [1, 1, 0, 0]), ([0, 1], [2, 2])]
  for _ in range(200):
    n = int(rng.integers(1, 401))
    ka, kb = int(rng.integers(1, 65)), int(rng.integers(1, 65))
    ida = rng.choice(65536, size=ka, replace=False)
    idb = rng.choice(65536, size=kb, replace=False)
    a = ida[rng.integers(0, ka, size=n)]
    if rng.random() < 0.5:   # correlated with a: a noisy relabelling
      b = idb[(np.searchsorted(np.sort(ida), a) + (rng.random(n) < 0.2) * rng.integers(0, kb, size=n)) % kb]
    else:
      b = idb[rng.integers(0, kb, size=n)]
    pairs.append((a.tolist(), b.tolist()))
  acc = np.array([evals.compute_sequence_match_accuracy(list(a), list(b)) for a, b in pairs], dtype=np.float64)
  lens = np.array([len(a) for a, _ in pairs], dtype=np.int64)
  np.savez_compressed(os.path.join(HERE, 'fn_evals.npz'),
                      a=np.concatenate([np.asarray(a, dtype=np.int64) for a, _ in pairs]),
                      b=np.concatenate([np.asarray(b, dtype=np.int64) for _, b in pairs]),
                      lens=lens, accuracy=acc)
  print('evals: {} pairs, accuracy {:.3f} .. {:.3f}'.format(len(pairs), acc.min(), acc.max()))


if __name__ == '__main__':
  main()
