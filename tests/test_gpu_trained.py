"""The reference's own behavioural tests at the boundary, on checkpoints the reference trained
(tests/golden/make_trained.py: fit + save by google/uis-rnn, predict() outputs recorded):

  * tests/uisrnn_test.py:26-70        one training label -> predict returns [0] * 10 for an
                                      array, for a list, and through parallel_predict
  * tests/integration_test.py:116-134 four clusters on a square, rnn_depth 2: accuracy == 1.0,
                                      also after save() / load()
  * SURVEY.md 8(d)                    the D=256 / H=512 model trained by the reference's fit on
                                      synthetic d-vectors, utterances of 100 / 500 / 1000 frames
                                      (2000 decode steps: scores ~3e5, where near-ties are sub-ulp)

Everything goes through uisrnn_amd.UISRNN.load() (checkpoints read without PyTorch) and the C
ABI.  HIP vs oracle: bit-exact.  HIP vs the reference's recorded outputs: labels identical (or
the alternative the reference re-scored, golden_util.accept_labels), scores within 1e-4 rel."""

import os

import numpy as np
import pytest

import golden_util
import uisrnn_amd
from uisrnn_amd import _capi

pytestmark = pytest.mark.gpu


def _model(case_name, **model_flags):
  argv = []
  for k, v in model_flags.items():
    argv += ['--' + k, str(v)]
  model_args, _, inference_args = uisrnn_amd.parse_arguments(argv)
  model = uisrnn_amd.UISRNN(model_args)
  model.load(os.path.join(golden_util.GOLDEN_DIR, golden_util.TRAINED_CASES[case_name]))
  return model, inference_args


def _set(inference_args, cfg):
  inference_args.beam_size, inference_args.look_ahead, inference_args.test_iteration = cfg


def test_single_label_model_predicts_all_zeros():
  case = golden_util.load_trained('trained_single')
  model, inference_args = _model('trained_single', observation_dim=16, rnn_hidden_size=8, rnn_depth=1)
  _set(inference_args, case['cfg'])
  assert case['cfg'][2] == 1                                   # test_iteration = 1 as in the reference test
  assert model.predict(case['seqs'][0], inference_args) == [0] * 10          # ndarray
  got = model.predict(case['seqs'][1:], inference_args)                     # list of two
  assert isinstance(got, list) and got == [[0] * 10, [0] * 10]
  got = uisrnn_amd.parallel_predict(model, case['seqs'][1:], inference_args)
  assert got == [[0] * 10, [0] * 10]
  assert [l.tolist() for l in case['labels']] == [[0] * 10] * 3  # what the reference returned


def test_four_cluster_toy_accuracy_is_one(tmp_path):
  case = golden_util.load_trained('trained_toy4')
  model, inference_args = _model('trained_toy4', observation_dim=2, rnn_hidden_size=8, rnn_depth=2)
  _set(inference_args, case['cfg'])
  truth = case['truth'].tolist()
  predicted = model.predict(case['seqs'][0], inference_args)
  assert uisrnn_amd.compute_sequence_match_accuracy(predicted, truth) == 1.0
  assert predicted == case['labels'][0].tolist()
  # save / load round trip (integration_test.py:121-134)
  path = str(tmp_path / 'toy4.uisrnn')
  model.save(path)
  model_args, _, _ = uisrnn_amd.parse_arguments(
      ['--observation_dim', '2', '--rnn_hidden_size', '8', '--rnn_depth', '2'])
  loaded = uisrnn_amd.UISRNN(model_args)
  loaded.load(path)
  again, acc = loaded.predict_and_evaluate([case['seqs'][0]], [truth], inference_args)
  assert again[0] == predicted and acc == [1.0]
  assert loaded.transition_bias == model.transition_bias


@pytest.mark.parametrize('name', [n for n in golden_util.trained_names() if n.startswith(('trained_d256', 'trained_d512'))])
def test_trained_d256_d512_against_reference_and_oracle(name, oracle_lib):
  case = golden_util.load_trained(name)
  beam, look, tau = case['cfg']
  dec = _capi.Decoder(case['params'])
  frames, offsets = oracle_lib.pack(case['seqs'])
  ref = oracle_lib.decode(case['params'], case['seqs'], beam, look, tau, n_threads=8)
  for flags in (0, _capi.UIS_FLAG_STEPWISE):
    out = dec.decode(frames, offsets, beam, look, tau, want_beam_scores=True, flags=flags)
    assert out['status'] == 0
    for u in range(len(case['seqs'])):
      got = out['labels'][offsets[u]:offsets[u + 1]]
      assert np.array_equal(got, ref['labels'][u])                           # HIP == oracle
      assert golden_util.accept_labels(case, u, got, out['scores'][u]), (name, u)  # == reference
    assert np.array_equal(out['scores'].view(np.uint32), ref['scores'].view(np.uint32))
    assert np.array_equal(out['beam_scores'].view(np.uint32), ref['beam_scores'].view(np.uint32))
    same = [u for u in range(len(case['seqs']))
            if np.array_equal(ref['labels'][u], case['labels'][u])]
    np.testing.assert_allclose(out['scores'][same], case['best'][same], rtol=1e-4)


def test_config_shapes_with_trained_models_against_the_oracle(oracle_lib):
  """BASELINE configs[4] (D=512, H=512, beam 20; the model the reference trained at D=512) and
  configs[2] (beam 50, look_ahead 2; the D=256 model) at sizes well beyond a handful of
  utterances: HIP vs oracle element-wise and bit-exact, and the labels are a sane diarization."""
  from uisrnn_amd import evals, synth, weights
  # configs[4]: 16 utterances x 150 frames, beam 20, cluster cap 11 (= the one-launch decode's limit)
  p512 = weights.load_checkpoint(os.path.join(golden_util.GOLDEN_DIR, 'trained_d512.uisrnn'))
  seqs, truth = synth.make_utterances(40_000, 16, 150, 512)
  ref = oracle_lib.decode(p512, seqs, 20, 1, 2, n_threads=16)
  assert int(ref['max_clusters'].max()) <= 11
  dec = _capi.Decoder(p512)
  frames, offsets = oracle_lib.pack(seqs)
  for flags in (_capi.UIS_FLAG_RESIDENT, _capi.UIS_FLAG_STEPWISE):
    out = dec.decode(frames, offsets, 20, 1, 2, max_clusters=11, flags=flags, want_beam_scores=True)
    assert out['status'] == 0
    for u in range(len(seqs)):
      assert np.array_equal(out['labels'][offsets[u]:offsets[u + 1]], ref['labels'][u])
    assert np.array_equal(out['beam_scores'].view(np.uint32), ref['beam_scores'].view(np.uint32))
  acc = [evals.compute_sequence_match_accuracy(ref['labels'][u].tolist(), truth[u].tolist()) for u in range(len(seqs))]
  assert np.mean(acc) > 0.97
  # configs[2]: 12 utterances x 80 frames, beam 50, look_ahead 2
  p256 = weights.load_checkpoint(os.path.join(golden_util.GOLDEN_DIR, 'trained_d256.uisrnn'))
  seqs, truth = synth.make_utterances(41_000, 12, 80, 256)
  ref = oracle_lib.decode(p256, seqs, 50, 2, 2, n_threads=12)
  cap = int(ref['max_clusters'].max()) + 1
  dec = _capi.Decoder(p256)
  frames, offsets = oracle_lib.pack(seqs)
  out = dec.decode(frames, offsets, 50, 2, 2, max_clusters=max(cap, 12), want_beam_scores=True)
  assert out['status'] == 0
  for u in range(len(seqs)):
    assert np.array_equal(out['labels'][offsets[u]:offsets[u + 1]], ref['labels'][u])
  assert np.array_equal(out['beam_scores'].view(np.uint32), ref['beam_scores'].view(np.uint32))
