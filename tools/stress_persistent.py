"""Stress of the persistent streaming launch (UIS_FLAG_PERSISTENT): many sessions of many small
pushes (random counts, silent utterances, label requests in between, idle gaps that make the
launch leave), every session compared with the offline decode bit for bit.

  python tools/stress_persistent.py [sessions] [utterances] [observation_dim] [rnn_hidden_size]
"""
import os, sys, time
sys.path.insert(0, '.')
import numpy as np
from uisrnn_amd import _capi, synth
n_sessions = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n_utt = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dim = int(sys.argv[3]) if len(sys.argv) > 3 else 256
hid = int(sys.argv[4]) if len(sys.argv) > 4 else 512
os.environ.setdefault('UIS_PERSIST_IDLE_MS', '3')
params = synth.tracker_params(dim, hid, 1, seed=0)
rng = np.random.default_rng(5)
lo_len, hi_len = [int(v) for v in os.environ.get('STRESS_LEN', '40,200').split(',')]
lens = [int(x) for x in rng.integers(lo_len, hi_len, size=n_utt)]
seqs, _ = synth.make_utterances(40_000, n_utt, lens, dim)
dec = _capi.Decoder(params)
frames = np.concatenate(seqs).astype(np.float32)
offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
ref = dec.decode(frames, offsets, 10, 1, 1, want_beam_scores=True)
bad, pushes, t0 = 0, 0, time.time()
for sess in range(n_sessions):
  dec.stream_begin(n_utt, 10, max(lens), flags=int(os.environ.get('STRESS_FLAGS', _capi.UIS_FLAG_PERSISTENT)))
  pos = [0] * n_utt
  while any(p < n for p, n in zip(pos, lens)):
    top = int(rng.choice([int(v) for v in os.environ.get('STRESS_TOPS', '1,1,1,2,5,16').split(',')]))
    chunks = []
    for u in range(n_utt):
      take = min(lens[u] - pos[u], int(rng.integers(0, top + 1)))
      chunks.append(seqs[u][pos[u]:pos[u] + take] if take else None)
      pos[u] += take
    dec.stream_push(chunks)
    pushes += 1
    r = rng.random() if not os.environ.get('STRESS_PLAIN') else 1.0
    if r < 0.02:
      time.sleep(0.01)          # longer than UIS_PERSIST_IDLE_MS: the launch leaves
    elif r < 0.05:
      labels, _, _, status = dec.stream_labels()
      assert status == 0 and [len(x) for x in labels] == pos, (status, [len(x) for x in labels], pos)
  labels, scores, overflow, status = dec.stream_labels()
  info = np.empty((n_utt, 10), dtype=np.float32)
  dec._check(dec._lib.uis_last_decode_info(dec._handle, None, info.ctypes.data_as(_capi._fp)), 'info')
  dec.stream_end()
  same = status == 0 and all(np.array_equal(labels[u], ref['labels'][offsets[u]:offsets[u + 1]]) for u in range(n_utt))
  same = same and np.array_equal(info.view(np.uint32), ref['beam_scores'].view(np.uint32))
  bad += not same
  if not same and bad <= 2:
    badu = [u for u in range(n_utt) if not np.array_equal(labels[u], ref['labels'][offsets[u]:offsets[u + 1]])]
    rows = [u for u in range(n_utt) if not np.array_equal(info[u].view(np.uint32), ref['beam_scores'][u].view(np.uint32))]
    print('  beam rows differing', rows[:10], 'lens', [lens[u] for u in rows[:10]])
    for u in rows[:2]:
      print('   u', u, 'got', info[u][:6], 'want', ref['beam_scores'][u][:6], 'score', scores[u], ref['scores'][u])
    print('  session', sess, 'status', status, 'bad utterances', len(badu), badu[:12], 'beam scores equal', np.array_equal(info.view(np.uint32), ref['beam_scores'].view(np.uint32)), flush=True)
print('sessions', n_sessions, 'pushes', pushes, 'mismatching sessions', bad, 'seconds', round(time.time() - t0, 1))
sys.exit(1 if bad else 0)
