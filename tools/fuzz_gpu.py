"""Randomised soak of the decode entry points against the C oracle (a checker run, like the tests):
random model shapes, beams, windows, utterance counts / lengths and path flags for a given
number of seconds; every decode has to match the oracle bit for bit (labels and beam scores),
and a streaming session fed the same utterances in random chunks has to match too.

  python tools/fuzz_gpu.py [seconds=60] [seed=1]
"""
import sys
import time
sys.path.insert(0, '.')
import os
import numpy as np
from uisrnn_amd import _capi, weights
from oracle import oracle

SECONDS = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
SEED = int(sys.argv[2]) if len(sys.argv) > 2 else 1
oracle.lib()


def bits(a):
  return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def draw(rng):
  big = rng.random() < 0.45  # shapes of the one-launch kernels
  if big:
    dim = int(rng.choice([30, 120, 128, 250, 256, 400, 512]))
    hid = int(rng.choice([70, 100, 128, 130, 200, 250, 256, 257, 300, 320, 370, 384, 390, 500, 512]))   # (65 .. 512: padded up to the kernels' shapes -- 257 .. 384 embedded, round 6)
    depth = 1
    look = int(rng.choice([1, 1, 1, 2, 2, 3]))   # look_ahead >= 2: k_decode_big<WIN>
    beam = int(rng.integers(1, 33))   # up to the wide class of the single-wave select
    # (161 and more: k_decode_big<WS> where the single-wave select applies -- and k_decode_coh with UIS_FLAG_COHORTS --, 257 and more: k_decode_big elsewhere)
    n_utt = int(rng.choice([1, 2, 5, 8, 9, 17, 33, 64, 70, 100, 128, 170, 257, 300, 530]))
    if look > 1:
      beam = int(rng.integers(1, 13 if look == 2 else 7))
      n_utt = int(rng.choice([1, 3, 8, 9, 40, 64, 70, 270]))
    max_len = 0  # set below from the oracle's budget
  else:
    dim = int(rng.integers(1, 80))
    hid = int(rng.integers(1, 100))
    depth = int(rng.integers(1, 4))
    look = int(rng.choice([1, 1, 2, 3]))
    beam = int(rng.integers(1, 12 if look > 1 else 48))
    n_utt = int(rng.integers(1, 12))
    max_len = 14 if look == 3 else 30
  tau = int(rng.choice([1, 2, 2, 3]))
  if big:  # keep the oracle (a CPU) at a second or two per case
    max_len = int(np.clip(12000 // (n_utt * beam * tau * (1 if look == 1 else 6 ** (look - 1))), 3, 48))
  lengths = [int(v) for v in rng.integers(1, max_len + 1, size=n_utt)]
  if big and look == 1 and dim in (128, 256) and rng.random() < 0.25:
    # (round 5) a list of equal-length utterances of 128 frames or more, given in host memory: the decode runs as several
    # launches with the later frames travelling behind the earlier ones (k_decode_rs / k_decode_big<WS> resume); main()
    # draws the slice boundaries
    n_utt = int(rng.choice([1, 3, 8, 9]))
    beam = int(rng.integers(1, 11))
    lengths = [int(rng.integers(128, 200))] * n_utt
  return dim, hid, depth, beam, look, tau, lengths


def main():
  rng = np.random.default_rng(SEED)
  t_end = time.time() + SECONDS
  n_case = n_decode = n_stream = n_multi = 0
  while time.time() < t_end:
    dim, hid, depth, beam, look, tau, lengths = draw(rng)
    seed = int(rng.integers(1 << 30))
    params = weights.init_params(dim, hid, depth, sigma2=float(rng.choice([0.02, 0.08, 0.3])),
                                 transition_bias=float(rng.choice([0.05, 0.2, 0.5])),
                                 crp_alpha=float(rng.choice([0.3, 1.0, 3.0])), seed=seed)
    params['rnn_init_hidden'] = (0.2 * rng.standard_normal((depth, hid))).astype(np.float32)
    n_spk = int(rng.integers(1, 5))
    cents = rng.standard_normal((n_spk, dim))
    seqs = []
    for n in lengths:
      ids = np.repeat(rng.integers(0, n_spk, size=n // 4 + 1), 4)[:n]
      seqs.append((cents[ids] * 0.4 + 0.1 * rng.standard_normal((n, dim))).astype(np.float64))
    ref = oracle.decode(params, seqs, beam, look, tau, n_threads=8)
    frames, offsets = oracle.pack(seqs)
    cap = max(int(ref['max_clusters'].max()) + look - 1, 2)
    dec = _capi.Decoder(params)
    tag = (dim, hid, depth, beam, look, tau, lengths, seed)
    flag_sets = [0, _capi.UIS_FLAG_STEPWISE, _capi.UIS_FLAG_SMALL_TILES, _capi.UIS_FLAG_OWNER_SELECT,
                 _capi.UIS_FLAG_REPLICATED_SELECT,  # every class of k_decode_rs, also where it is not the default
                 _capi.UIS_FLAG_COHORTS,            # k_decode_coh (two cohorts in flight) where k_decode_big<WS> is the default (-DUIS_WITH_COHORTS builds; else ignored)
                 _capi.UIS_FLAG_AGENT_FLAGS,        # (round 6) the hand-offs' phase words at agent scope

                 int(rng.choice([_capi.UIS_FLAG_NO_DEDUP, _capi.UIS_FLAG_GENERIC_SELECT | _capi.UIS_FLAG_STEPWISE,
                                 _capi.UIS_FLAG_GRAPH | _capi.UIS_FLAG_STEPWISE]))]
    os.environ['UIS_SPLIT_MIN_MB'] = '0'                # (the lists here are far below the size the library spends a launch on)
    if len(set(lengths)) == 1 and lengths[0] >= 128:  # the several-launch decode: random slice boundaries (or the library's own)
      cuts = sorted(set(int(v) for v in rng.integers(20, lengths[0], size=int(rng.integers(0, 4)))))
      if cuts:
        os.environ['UIS_SPLIT_FRAMES'] = ','.join(str(c) for c in cuts)
      else:
        os.environ.pop('UIS_SPLIT_FRAMES', None)
    for fl in flag_sets:
      out = dec.decode(frames, offsets, beam, look, tau, max_clusters=cap, flags=fl, want_beam_scores=True)
      assert out['status'] == 0, ('status', out['status'], fl, tag)
      for u in range(len(seqs)):
        assert np.array_equal(out['labels'][offsets[u]:offsets[u + 1]], ref['labels'][u]), ('labels', fl, u, tag)
      assert np.array_equal(bits(out['beam_scores']), bits(ref['beam_scores'])), ('scores', fl, tag)
      n_decode += 1
      n_multi += out['stats']['decode_launches'] >= 2
    out = dec.decode_f64(seqs, beam, look, tau, max_clusters=cap, want_beam_scores=True)  # predict()'s own input type
    assert out['status'] == 0 and np.array_equal(bits(out['beam_scores']), bits(ref['beam_scores'])), ('f64', tag)
    assert np.array_equal(out['labels'], np.concatenate([ref['labels'][u] for u in range(len(seqs))])), ('f64 labels', tag)
    n_decode += 1
    if look == 1 and tau == 1:  # online decoding = predict_single with test_iteration 1
      for fl in (0, _capi.UIS_FLAG_PERSISTENT, _capi.UIS_FLAG_STEPWISE):
        try:
          dec.stream_begin(len(seqs), beam, max(lengths), max_clusters=cap, flags=fl)
        except _capi.HipLibraryError as e:
          if fl == _capi.UIS_FLAG_PERSISTENT and e.status == _capi.UIS_ERR_UNSUPPORTED:
            continue
          raise
        pos = [0] * len(seqs)
        while any(p < n for p, n in zip(pos, lengths)):
          chunk = []
          for u, n in enumerate(lengths):
            k = int(min(n - pos[u], rng.integers(0, 7)))
            chunk.append(seqs[u][pos[u]:pos[u] + k].astype(np.float32))
            pos[u] += k
          if sum(len(c) for c in chunk) == 0:
            continue
          dec.stream_push(chunk)
        lab, sc, _, status = dec.stream_labels()
        dec.stream_end()
        assert status == 0, ('stream status', status, fl, tag)
        for u in range(len(seqs)):
          assert np.array_equal(lab[u], ref['labels'][u]), ('stream labels', fl, u, tag)
        assert np.array_equal(bits(sc), bits(ref['beam_scores'][:, 0])), ('stream scores', fl, tag)
        n_stream += 1
    dec.close()
    n_case += 1
  print('fuzz: cases', n_case, 'decodes', n_decode, '(of them in several launches:', n_multi, ') streaming sessions', n_stream,
        'mismatching 0', flush=True)


if __name__ == '__main__':
  main()
