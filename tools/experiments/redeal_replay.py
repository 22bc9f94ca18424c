"""configs[4]: what a RE-DEAL of the utterances to the XCDs between the two passes of test_iteration = 2 would gain
(round 6, the verdict's item 4: "pass 1's row counts ARE whole-utterance knowledge for pass 2").

k_decode_resident can stop after any step and resume; with a position -> utterance table it could stop at step N (the
end of pass 1), read every utterance's row count, deal the utterances to the XCDs longest-rows-first and run pass 2
balanced.  Before building that: the measured prediction.  Every utterance decoded alone with test_iteration 1 and 2
gives its rows in pass 1 and in pass 2; the list is then decoded in three orders -- as given, dealt by the rows of
pass 1 (what the re-deal would know), dealt by the rows of the whole decode (R5.8's bound, knowledge nobody has) --
with test_iteration 1 (= pass 1 alone) and 2 (both passes); pass 2's time under a deal is the difference.  A decode
that runs pass 1 as given and pass 2 re-dealt takes  T1(as given) + T2(by pass 1's rows) + one relaunch (0.15 ms,
LABNOTES R5.6).

  gpurun -- python tools/experiments/redeal_replay.py > gpurun_out/r06_redeal_replay.txt
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


def pack(order):
  frames = np.concatenate([seqs[u] for u in order], axis=0).astype(np.float32)
  lens = [seqs[u].shape[0] for u in order]
  return frames, np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)


def rows_alone(tau):
  out = []
  for u in range(N_UTT):
    r = dec.decode(*pack([u]), BEAM, 1, tau, max_clusters=CAP)
    assert r['status'] == 0
    out.append(r['stats']['rnn_rows'])
  return np.array(out, dtype=np.float64)


def lpt(weight):
  load, count, order = np.zeros(NCL), np.zeros(NCL, dtype=int), [None] * N_UTT
  for u in np.argsort(-weight, kind='stable'):
    c = min((c for c in range(NCL) if count[c] < N_UTT // NCL), key=lambda c: load[c])
    order[c + NCL * count[c]] = int(u)
    load[c] += weight[u]
    count[c] += 1
  return order


def ms(order, tau):
  frames, offsets = pack(order)
  best = 1e9
  for _ in range(7):
    r = dec.decode(frames, offsets, BEAM, 1, tau, max_clusters=CAP)
    assert r['status'] == 0 and r['stats']['decode_kernel'] == 'k_decode_resident', r['stats']['decode_kernel']
    best = min(best, r['stats']['decode_ms'])
  return best


def imbalance(order, weight):
  load = np.zeros(NCL)
  for p, u in enumerate(order):
    load[p % NCL] += weight[u]
  return load.max() / load.mean()


p1 = rows_alone(1)
both = rows_alone(2)
p2 = both - p1
print('rows per utterance: pass 1 min %d mean %.0f max %d; pass 2 min %d mean %.0f max %d; correlation(pass 1, pass 2) %.3f'
      % (p1.min(), p1.mean(), p1.max(), p2.min(), p2.mean(), p2.max(), np.corrcoef(p1, p2)[0, 1]))
orders = {'as given': list(range(N_UTT)), 'by pass 1\'s rows': lpt(p1), 'by the whole decode\'s rows': lpt(both)}
t = {}
for name, order in orders.items():
  t1, t12 = ms(order, 1), ms(order, 2)
  t[name] = (t1, t12 - t1, t12)
  print('%-28s heaviest XCD / mean: pass 1 %.3f  pass 2 %.3f | pass 1 alone %.3f ms, pass 2 (difference) %.3f ms, both %.3f ms = %.4f M frames/s'
        % (name, imbalance(order, p1), imbalance(order, p2), t1, t12 - t1, t12, N_UTT * N_FRAMES / t12 / 1e3))
relaunch = 0.15
pred = t['as given'][0] + t['by pass 1\'s rows'][1] + relaunch
print('pass 1 as given + pass 2 dealt by pass 1\'s rows + one relaunch (%.2f ms): %.3f ms = %.4f M frames/s (as given: %.4f M; dealt by the whole decode\'s rows from the start: %.4f M)'
      % (relaunch, pred, N_UTT * N_FRAMES / pred / 1e3, N_UTT * N_FRAMES / t['as given'][2] / 1e3, N_UTT * N_FRAMES / t['by the whole decode\'s rows'][2] / 1e3))
