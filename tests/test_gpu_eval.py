"""Evaluation on the device: uis_eval_* against the reference's known answers
(tests/evals_test.py:36-58 of google/uis-rnn) and against the host mirror of
uisrnn/evals.py:40-73 (scipy's linear_sum_assignment) on random label pairs.  Integer work:
the matched counts must be identical, and so must the float64 accuracies."""

import numpy as np
import pytest

import golden_util
from uisrnn_amd import _capi
from uisrnn_amd import evals
from uisrnn_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def decoder():
  return _capi.Decoder(golden_util.load_case('tiny_d16')['params'])


def test_reference_known_answers(decoder):
  """The three known-answer cases of the reference's evals_test.py, in one launch."""
  pairs = [([0, 0, 1, 2, 2], [3, 3, 4, 4, 1], 0.8),     # tests/evals_test.py:36-42
           ([0, 0, 1, 2, 2], [3, 3, 4, 1, 1], 1.0),     # :44-50
           ([1, 1], [1, 2], 0.5)]                       # :52-58
  got = evals.sequence_match_accuracies_device(
      decoder, [p[0] for p in pairs], [p[1] for p in pairs])
  assert got == [p[2] for p in pairs]
  # symmetry (:60-70)
  rng = np.random.default_rng(3)
  a = rng.permutation([1] * 10 + [2] * 20 + [3] * 30 + [4] * 40).tolist()
  b = rng.permutation([1] * 10 + [2] * 20 + [3] * 30 + [4] * 40).tolist()
  ab, ba = evals.sequence_match_accuracies_device(decoder, [a, b], [b, a])
  assert ab == ba == evals.compute_sequence_match_accuracy(a, b)
  # the reference's argument errors (:72-82)
  with pytest.raises(ValueError):
    evals.sequence_match_accuracies_device(decoder, [[0, 0, 1, 2]], [[3, 3, 4, 4, 1]])
  with pytest.raises(ValueError):
    evals.sequence_match_accuracies_device(decoder, [[]], [[]])
  with pytest.raises(TypeError):
    evals.sequence_match_accuracies_device(decoder, [np.array([1, 2])], [[1, 2]])


def test_random_pairs_against_the_host_function(decoder):
  """Many shapes: equal / different numbers of ids (1 .. 64), strings, noise levels."""
  rng = np.random.default_rng(11)
  seqs1, seqs2 = [], []
  for trial in range(120):
    n = int(rng.integers(1, 700))
    k1 = int(rng.integers(1, 65 if trial % 7 == 0 else 9))
    k2 = int(rng.integers(1, 65 if trial % 5 == 0 else 9))
    a = rng.integers(0, k1, size=n)
    perm = rng.permutation(max(k1, k2))
    b = np.where(rng.random(n) < rng.random(), perm[a] % k2, rng.integers(0, k2, size=n))
    if trial % 3 == 0:
      seqs1.append(['spk{}'.format(v) for v in a])   # ids of any hashable kind
    else:
      seqs1.append((a * 37 + 5).tolist())            # sparse integer ids
    seqs2.append(b.tolist())
  got = evals.sequence_match_accuracies_device(decoder, seqs1, seqs2)
  want = [evals.compute_sequence_match_accuracy(x, y) for x, y in zip(seqs1, seqs2)]
  assert got == want


def test_limits_are_reported(decoder):
  off = np.array([0, 70], dtype=np.int64)
  many = np.arange(70, dtype=np.int32)           # 70 distinct ids > 64
  with pytest.raises(_capi.HipLibraryError, match='64 distinct'):
    decoder.eval_matched(many, many, off)
  big = np.full(70, 70000, dtype=np.int32)       # label value out of range
  with pytest.raises(_capi.HipLibraryError, match='65536'):
    decoder.eval_matched(big, big, off)
  # 64 ids are fine, and empty utterances give 0
  ok = (np.arange(128, dtype=np.int32) % 64)
  assert decoder.eval_matched(ok, ok[::-1].copy(), np.array([0, 128, 128], np.int64)).tolist() == [128, 0]


def test_predict_and_evaluate_keeps_labels_on_the_device(oracle_lib):
  """UISRNN.predict_and_evaluate: accuracies from the labels still resident after the decode
  equal the host function applied to the returned labels; also through device pointers."""
  import torch
  import uisrnn_amd
  model_args, _, inference_args = uisrnn_amd.parse_arguments([])
  model = uisrnn_amd.UISRNN(model_args)
  params = synth.tracker_params(256, 512, 1, seed=0)
  model.load_params(params)
  lengths = [120, 64, 200, 33, 90]
  seqs, spk = synth.make_utterances(9600, len(lengths), lengths, 256)
  truth = [['u{}_{}'.format(u, s) for s in ids] for u, ids in enumerate(spk)]
  predicted, acc = model.predict_and_evaluate(seqs, truth, inference_args)
  ref = oracle_lib.decode(params, seqs, 10, 1, 2, n_threads=4)
  assert predicted == [l.tolist() for l in ref['labels']]
  assert acc == [uisrnn_amd.compute_sequence_match_accuracy(t, p) for t, p in zip(truth, predicted)]
  assert min(acc) > 0.9
  # device pointers in, nothing but the counts out
  flat_p = torch.tensor(np.concatenate(predicted), dtype=torch.int32, device='cuda')
  flat_t = torch.tensor(np.concatenate([evals.dense_ids(t) for t in truth]), dtype=torch.int32,
                        device='cuda')
  off = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64)
  matched = model._get_decoder().eval_matched_device(flat_p.data_ptr(), flat_t.data_ptr(), off)
  assert [m / n for m, n in zip(matched.tolist(), lengths)] == acc


def test_recorded_reference_accuracies(decoder):
  """k_eval pinned to the reference itself: 203 label-sequence pairs whose accuracies
  google/uis-rnn's evals.compute_sequence_match_accuracy (uisrnn/evals.py:40-73) returned in the
  dev container (tests/golden/make_scores.py); identical float64 values."""
  data = np.load(golden_util.GOLDEN_DIR + '/fn_evals.npz')
  seqs1, seqs2, pos = [], [], 0
  for n in data['lens']:
    seqs1.append(data['a'][pos:pos + n].tolist())
    seqs2.append(data['b'][pos:pos + n].tolist())
    pos += n
  got = evals.sequence_match_accuracies_device(decoder, seqs1, seqs2)
  assert got == data['accuracy'].tolist()


def test_predict_and_evaluate_with_more_ids_than_the_device_kernel_takes(oracle_lib):
  """A ground truth with more than 64 distinct ids in one utterance: k_eval refuses it
  (UIS_ERR_UNSUPPORTED), the reference's evals.py has no such limit -- predict_and_evaluate falls
  back to the host function and returns the same values."""
  import uisrnn_amd
  model_args, _, inference_args = uisrnn_amd.parse_arguments([])
  model = uisrnn_amd.UISRNN(model_args)
  model.load_params(synth.tracker_params(256, 512, 1, seed=0))
  lengths = [150, 40]
  seqs, spk = synth.make_utterances(9700, len(lengths), lengths, 256)
  truth = [list(range(lengths[0])), spk[1].tolist()]   # 150 distinct ids in utterance 0
  predicted, acc = model.predict_and_evaluate(seqs, truth, inference_args)
  assert acc == [uisrnn_amd.compute_sequence_match_accuracy(t, p) for t, p in zip(truth, predicted)]
  with pytest.raises(_capi.HipLibraryError):   # (the device entry point itself still says no)
    evals.sequence_match_accuracies_device(model._get_decoder(), [truth[0]], [predicted[0]])
