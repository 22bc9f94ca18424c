"""Online decoding (uis_stream_*): any chunking == one offline decode, bit for bit.

The reference has no streaming entry point; the contract tested here is the one stated in
include/uisrnn_hip.h: a session that has received an utterance's frames in whatever pieces
holds exactly the beam of predict_single(test_iteration=1) over those frames
(uisrnn/uisrnn.py:479-562) -- labels, best score, the whole final beam.  The offline side is
checked against the CPU oracle as well.
"""

import numpy as np
import pytest

from uisrnn_amd import _capi
from uisrnn_amd import synth
from uisrnn_amd import weights

pytestmark = pytest.mark.gpu


def _bits(a):
  return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def _stream(dec, seqs, beam, schedule, max_frames, max_clusters=0, check_every_push=None, flags=0):
  """schedule: list of per-push lists of frame counts (one per utterance)."""
  dec.stream_begin(len(seqs), beam, max_frames, max_clusters=max_clusters, flags=flags)
  try:
    pos = [0] * len(seqs)
    for counts in schedule:
      chunks = []
      for u, n in enumerate(counts):
        chunks.append(seqs[u][pos[u]:pos[u] + n] if n else None)
        pos[u] += n
      dec.stream_push(chunks)
      if check_every_push is not None:
        check_every_push(dec, pos)
    assert pos == [len(s) for s in seqs], 'schedule does not cover the utterances'
    labels, scores, overflow, status = dec.stream_labels()
    info = np.empty((len(seqs), beam), dtype=np.float32)
    dec._check(dec._lib.uis_last_decode_info(dec._handle, None, info.ctypes.data_as(_capi._fp)), 'info')
    return labels, scores, overflow, status, info
  finally:
    dec.stream_end()


def _offline(dec, seqs, beam, max_clusters=0):
  lens = [len(s) for s in seqs]
  offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
  frames = np.concatenate(seqs).astype(np.float32)
  out = dec.decode(frames, offsets, beam, 1, 1, max_clusters=max_clusters, want_beam_scores=True)
  return out, offsets


def _random_schedule(rng, lens, max_chunk):
  left = list(lens)
  schedule = []
  while any(left):
    counts = []
    for u, n in enumerate(left):
      take = int(min(n, rng.integers(0, max_chunk + 1)))   # 0 = this utterance is silent in this push
      counts.append(take)
      left[u] -= take
    if any(counts):
      schedule.append(counts)
  return schedule


@pytest.mark.parametrize('max_chunk', [1, 7, 40])
def test_any_chunking_equals_offline_decode(max_chunk, oracle_lib):
  params = synth.tracker_params(256, 512, 1, seed=21)
  lens = [60, 33, 1, 90, 17, 45, 72, 8, 64]
  seqs, _ = synth.make_utterances(12_000, len(lens), lens, 256)
  dec = _capi.Decoder(params)
  off, offsets = _offline(dec, seqs, 10)
  ref = oracle_lib.decode(params, seqs, 10, 1, 1, n_threads=8)
  rng = np.random.default_rng(max_chunk)
  schedule = _random_schedule(rng, lens, max_chunk)
  # a push runs its steps in ONE launch of the resident decode kernel where that applies (this
  # shape: UIS_FLAG_RESIDENT demands it) and as four kernels per step under UIS_FLAG_STEPWISE
  for flags in (_capi.UIS_FLAG_RESIDENT, _capi.UIS_FLAG_STEPWISE, 0):
    labels, scores, overflow, status, beam = _stream(dec, seqs, 10, schedule, 100, flags=flags)
    assert status == 0 and not overflow.any()
    for u in range(len(seqs)):
      assert np.array_equal(labels[u], off['labels'][offsets[u]:offsets[u + 1]]), u
      assert np.array_equal(labels[u], ref['labels'][u]), u
    assert np.array_equal(_bits(scores), _bits(off['scores']))
    assert np.array_equal(_bits(beam), _bits(off['beam_scores']))
    assert np.array_equal(_bits(beam), _bits(ref['beam_scores']))
  # the handle is usable for ordinary decodes again
  again, _ = _offline(dec, seqs, 10)
  assert np.array_equal(again['labels'], off['labels'])


def test_prefix_labels_after_every_push(oracle_lib):
  """After every push the session equals an offline decode of the prefixes received so far."""
  params = synth.tracker_params(256, 512, 1, seed=22)
  lens = [24, 10, 31]
  seqs, _ = synth.make_utterances(12_100, 3, lens, 256)
  dec = _capi.Decoder(params)
  checker = _capi.Decoder(params)

  def check(d, pos):
    labels, scores, _, status = d.stream_labels()
    assert status == 0
    prefixes = [seqs[u][:pos[u]] for u in range(3)]
    keep = [u for u in range(3) if pos[u] > 0]
    if not keep:
      return
    off, offsets = _offline(checker, [prefixes[u] for u in keep], 6)
    for k, u in enumerate(keep):
      assert np.array_equal(labels[u], off['labels'][offsets[k]:offsets[k + 1]]), (u, pos)
      assert _bits(scores[u]) == _bits(off['scores'][k])
    for u in range(3):
      if pos[u] == 0:
        assert len(labels[u]) == 0 and scores[u] == 0.0

  schedule = [[5, 0, 2], [0, 0, 9], [7, 10, 0], [12, 0, 20]]
  _stream(dec, seqs, 6, schedule, 40, check_every_push=check)
  _stream(dec, seqs, 6, schedule, 40, check_every_push=check, flags=_capi.UIS_FLAG_STEPWISE)


def test_many_utterances_one_frame_pushes(oracle_lib):
  """More utterances than workgroups per cluster, one frame per push, some silent: the
  one-launch step kernel against the offline decode."""
  params = synth.tracker_params(256, 256, 1, seed=25)   # hidden 256: two ranks share a feature tile
  lens = [6 + (5 * u) % 9 for u in range(70)]
  seqs, _ = synth.make_utterances(12_400, len(lens), lens, 256)
  dec = _capi.Decoder(params)
  off, offsets = _offline(dec, seqs, 10)
  schedule = [[1 if (n > t and (u + t) % 4) else 0 for u, n in enumerate(lens)] for t in range(14)]
  left = [n - sum(row[u] for row in schedule) for u, n in enumerate(lens)]
  schedule.append(left)
  labels, scores, overflow, status, beam = _stream(dec, seqs, 10, schedule, 16, flags=_capi.UIS_FLAG_RESIDENT)
  assert status == 0 and not overflow.any()
  for u in range(len(seqs)):
    assert np.array_equal(labels[u], off['labels'][offsets[u]:offsets[u + 1]]), u
  assert np.array_equal(_bits(beam), _bits(off['beam_scores']))


@pytest.mark.parametrize('max_chunk', [1, 5, 16])
def test_persistent_launch_any_chunking_equals_offline_decode(max_chunk, oracle_lib):
  """UIS_FLAG_PERSISTENT: the decode kernel stays on the device between pushes and takes them from
  the host-memory mailbox; labels are back-traced inside it.  Same contract as every other path."""
  params = synth.tracker_params(256, 512, 1, seed=31)
  lens = [60, 33, 1, 90, 17, 45, 72, 8, 64, 2, 29]
  seqs, _ = synth.make_utterances(12_500, len(lens), lens, 256)
  dec = _capi.Decoder(params)
  off, offsets = _offline(dec, seqs, 10)
  ref = oracle_lib.decode(params, seqs, 10, 1, 1, n_threads=8)
  schedule = _random_schedule(np.random.default_rng(100 + max_chunk), lens, max_chunk)
  prefix_checks = []

  def check(d, pos):   # labels after every few pushes: prefixes of the final answer are NOT expected
    if len(prefix_checks) % 7 == 0:      # (the best hypothesis may change), lengths and status are
      labels, scores, _, status = d.stream_labels()
      assert status == 0 and [len(x) for x in labels] == pos
    prefix_checks.append(1)

  labels, scores, overflow, status, beam = _stream(dec, seqs, 10, schedule, 100, flags=_capi.UIS_FLAG_PERSISTENT,
                                                   check_every_push=check)
  assert status == 0 and not overflow.any()
  for u in range(len(seqs)):
    assert np.array_equal(labels[u], off['labels'][offsets[u]:offsets[u + 1]]), u
    assert np.array_equal(labels[u], ref['labels'][u]), u
  assert np.array_equal(_bits(scores), _bits(off['scores']))
  assert np.array_equal(_bits(beam), _bits(off['beam_scores']))
  again, _ = _offline(dec, seqs, 10)   # the launch has left: ordinary decodes work again
  assert np.array_equal(again['labels'], off['labels'])


def test_persistent_launch_leaves_when_idle_and_comes_back(oracle_lib, monkeypatch):
  """Idle for longer than UIS_PERSIST_IDLE_MS: the kernel writes its tables back and leaves; the
  next push starts a new one; a push too large for the mailbox goes the ordinary way in between."""
  import time
  monkeypatch.setenv('UIS_PERSIST_IDLE_MS', '5')
  params = synth.tracker_params(256, 512, 1, seed=32)
  lens = [70, 70, 41, 70, 12, 70, 70, 55, 70]
  seqs, _ = synth.make_utterances(12_600, len(lens), lens, 256)
  dec = _capi.Decoder(params)
  off, offsets = _offline(dec, seqs, 10)
  schedule = []
  left = list(lens)
  for take in (1, 1, 3, 1, 40, 2, 1, 1, 16, 1, 99):   # 40 and 99: more than 16 frames per utterance
    counts = [min(n, take) for n in left]
    left = [n - c for n, c in zip(left, counts)]
    if any(counts):
      schedule.append(counts)
  assert not any(left)
  naps = iter([0, 0.05, 0, 0, 0.05, 0, 0.02, 0, 0, 0.05, 0, 0, 0])

  def nap(d, pos):
    time.sleep(next(naps))
    if len(pos) and pos[0] in (2, 6):
      labels, _, _, status = d.stream_labels()   # once with the launch gone, once (usually) with it resident
      assert status == 0 and [len(x) for x in labels] == pos

  labels, scores, overflow, status, beam = _stream(dec, seqs, 10, schedule, 100, flags=_capi.UIS_FLAG_PERSISTENT,
                                                   check_every_push=nap)
  assert status == 0 and not overflow.any()
  for u in range(len(seqs)):
    assert np.array_equal(labels[u], off['labels'][offsets[u]:offsets[u + 1]]), u
  assert np.array_equal(_bits(scores), _bits(off['scores']))
  assert np.array_equal(_bits(beam), _bits(off['beam_scores']))


def test_persistent_launch_rows_that_change_hands_between_pushes(oracle_lib):
  """Ragged counts (0 .. 16 frames per utterance and push, 64 utterances, back-to-back pushes
  without label requests in between): which cluster a chunk row belongs to would change from push
  to push if the clusters did not own fixed row ranges -- and a stale dirty line left in the
  previous owner's XCD-private L2 would then be free to overwrite the new owner's data (this
  test caught exactly that: rare score differences, never on uniform pushes)."""
  params = synth.tracker_params(256, 512, 1, seed=33)
  rng = np.random.default_rng(7)
  lens = [int(x) for x in rng.integers(40, 160, size=64)]
  seqs, _ = synth.make_utterances(12_700, len(lens), lens, 256)
  dec = _capi.Decoder(params)
  off, offsets = _offline(dec, seqs, 10)
  for rep in range(3):
    left = list(lens)
    schedule = []
    while any(left):
      top = int(rng.choice([1, 5, 16]))
      counts = [int(min(n, rng.integers(0, top + 1))) for n in left]
      left = [n - c for n, c in zip(left, counts)]
      if any(counts):
        schedule.append(counts)
    labels, scores, overflow, status, beam = _stream(dec, seqs, 10, schedule, max(lens), flags=_capi.UIS_FLAG_PERSISTENT)
    assert status == 0 and not overflow.any()
    for u in range(len(seqs)):
      assert np.array_equal(labels[u], off['labels'][offsets[u]:offsets[u + 1]]), (rep, u)
    assert np.array_equal(_bits(beam), _bits(off['beam_scores'])), rep


def test_cluster_cap_in_a_session_is_reported_by_every_path(oracle_lib):
  """A hypothesis that outgrows max_clusters: uis_stream_labels returns UIS_ERR_CLUSTER_CAP and the
  per-utterance flags, from the per-step kernels, the one launch per push and the resident launch
  (whose flags travel through the mailbox); the utterance that fits still has the oracle's labels."""
  params = weights.init_params(256, 512, 1, sigma2=0.5, transition_bias=0.5, crp_alpha=50.0, seed=15)
  rng = np.random.default_rng(16)
  seqs = [rng.standard_normal((30, 256)), rng.standard_normal((2, 256)), rng.standard_normal((25, 256))]
  ref = oracle_lib.decode(params, seqs, 8, 1, 1, n_threads=3)
  assert ref['max_clusters'][0] > 8 and ref['max_clusters'][1] <= 8
  dec = _capi.Decoder(params)
  schedule = [[5, 2, 5]] + [[5, 0, 5]] * 4 + [[5, 0, 0]]
  for flags in (_capi.UIS_FLAG_STEPWISE, _capi.UIS_FLAG_RESIDENT, _capi.UIS_FLAG_PERSISTENT):
    labels, scores, overflow, status, _ = _stream(dec, seqs, 8, schedule, 30, max_clusters=8, flags=flags)
    assert status == _capi.UIS_ERR_CLUSTER_CAP, flags
    assert overflow.tolist() == [1, 0, 1], flags
    assert np.array_equal(labels[1], ref['labels'][1])


def test_persistent_launch_that_fails_its_placement_check_is_reported():
  """UIS_FLAG_TEST_MISPLACED makes one workgroup claim another XCD: the resident launch gives up
  at its first in-launch barrier and ends; the push reports it (the session's state is gone),
  and the handle goes on with ordinary launches."""
  params = synth.tracker_params(256, 512, 1, seed=34)
  seqs, _ = synth.make_utterances(12_800, 4, 12, 256)
  dec = _capi.Decoder(params)
  dec.stream_begin(4, 6, 16, flags=_capi.UIS_FLAG_PERSISTENT | _capi.UIS_FLAG_TEST_MISPLACED)
  with pytest.raises(_capi.HipLibraryError):
    dec.stream_push([s[:3] for s in seqs])
  dec.stream_end()
  off, offsets = _offline(dec, seqs, 6)
  labels, scores, overflow, status, beam = _stream(dec, seqs, 6, [[5] * 4, [7] * 4], 16)
  assert status == 0
  for u in range(4):
    assert np.array_equal(labels[u], off['labels'][offsets[u]:offsets[u + 1]])


def test_persistent_flag_is_refused_where_it_cannot_work():
  params = weights.init_params(20, 24, 1, sigma2=0.08, transition_bias=0.2, seed=4)
  dec = _capi.Decoder(params)
  opts = _capi.make_opts(4, 1, 1, 0, _capi.UIS_FLAG_PERSISTENT, 0)
  assert dec._lib.uis_stream_begin(dec._handle, 2, opts, 8) == _capi.UIS_ERR_UNSUPPORTED
  dec.stream_begin(2, 4, 8)    # the handle is still good
  dec.stream_end()


def test_odd_model_shapes_general_select_and_depth(oracle_lib):
  """Padded dims, depth 2 (k_dense_upper_in), a beam too wide for the fast select kernel."""
  rng = np.random.default_rng(3)
  params = weights.init_params(20, 24, 2, sigma2=0.08, transition_bias=0.2, seed=4)
  params['rnn_init_hidden'] = (0.2 * rng.standard_normal((2, 24))).astype(np.float32)
  cents = rng.standard_normal((3, 20))
  seqs = []
  for n in (15, 4, 22):
    ids = np.repeat(rng.integers(0, 3, size=n // 4 + 1), 4)[:n]
    seqs.append((cents[ids] * 0.4 + 0.1 * rng.standard_normal((n, 20))).astype(np.float64))
  for beam in (5, 40):
    ref = oracle_lib.decode(params, seqs, beam, 1, 1, n_threads=4)
    cap = max(int(ref['max_clusters'].max()), 2)
    dec = _capi.Decoder(params)
    labels, scores, overflow, status, beam_scores = _stream(
        dec, seqs, beam, _random_schedule(rng, [15, 4, 22], 6), 30, max_clusters=cap)
    assert status == 0
    for u in range(3):
      assert np.array_equal(labels[u], ref['labels'][u]), (beam, u)
    assert np.array_equal(_bits(beam_scores), _bits(ref['beam_scores']))


def test_session_errors():
  params = synth.tracker_params(256, 512, 1, seed=23)
  dec = _capi.Decoder(params)
  seqs, _ = synth.make_utterances(12_200, 2, 5, 256)
  assert dec._lib.uis_stream_labels(dec._handle, None, None, None) == _capi.UIS_ERR_INVALID_ARG  # nothing open
  dec.stream_begin(2, 4, 8)
  with pytest.raises(_capi.HipLibraryError):          # one session per handle
    dec.stream_begin(2, 4, 8)
  with pytest.raises(_capi.HipLibraryError):          # offline decode refused meanwhile
    _offline(dec, seqs, 4)
  dec.stream_push([seqs[0], seqs[1]])
  with pytest.raises(_capi.HipLibraryError):          # 5 + 5 > max_frames 8
    dec.stream_push([seqs[0], None])
  labels, _, _, _ = dec.stream_labels()                # the failed push changed nothing
  assert [len(x) for x in labels] == [5, 5]
  dec.stream_end()
  dec.stream_end()                                     # idempotent
  opts = _capi.make_opts(4, 1, 2, 0, 0, 0)             # test_iteration 2 is not online
  assert dec._lib.uis_stream_begin(dec._handle, 2, opts, 8) == _capi.UIS_ERR_UNSUPPORTED


def test_python_online_session_matches_predict():
  import uisrnn_amd
  model_args, _, inference_args = uisrnn_amd.parse_arguments([])
  model = uisrnn_amd.UISRNN(model_args)
  model.load_params(synth.tracker_params(256, 512, 1, seed=24))
  seqs, _ = synth.make_utterances(12_300, 3, [50, 20, 35], 256)
  inference_args.test_iteration = 1
  offline = model.predict(seqs, inference_args)
  with model.online(3, inference_args, max_frames=64) as session:
    session.push([seqs[0][:10], None, seqs[2][:35]])
    partial = session.labels()
    assert [len(x) for x in partial] == [10, 0, 35] and partial[2] == offline[2]
    session.push([seqs[0][10:], seqs[1], None])
    assert session.labels() == offline
    with pytest.raises(TypeError):
      session.push([seqs[0].astype(np.float32), None, None])   # the reference's float64 rule
  block = np.stack([s[:20] for s in seqs])            # [U, n, D]: every utterance gets the same count
  with model.online(3, inference_args, max_frames=64) as session:
    session.push(block[:, :7])
    session.push(block[:, 7:])
    assert session.labels() == [x[:20] for x in model.predict([s[:20] for s in seqs], inference_args)]
  with model.online(3, inference_args, max_frames=64, persistent=True) as session:
    assert session.persistent
    for lo in range(0, 50, 5):
      session.push([s[lo:lo + 5] if lo < len(s) else None for s in seqs])
    assert session.labels() == offline
  inference_args.look_ahead = 2
  with pytest.raises(ValueError):
    model.online(3, inference_args, max_frames=64)
