"""The CPU oracle against the reference's own outputs (tests/golden/*.npz).

The fixtures were produced by importing google/uis-rnn in the dev container
(tests/golden/make_golden.py).  Parity bar (BASELINE.json north_star): cluster-id
sequences identical, log-scores within RELATIVE 1e-4 (scores are float32
accumulations that reach 1e4..1e5; the observed error is <= 3e-7).
"""

import numpy as np
import pytest

import golden_util

RTOL = 1e-4
CASES = golden_util.case_names()


def test_fixtures_present():
  assert set(CASES) >= {'tiny_d16', 'toy_d2_depth2', 'd32_lookahead3',
                        'd20_h24_depth3', 'tracker_d256', 'tracker_d256_long', 'tracker_d64_h300'}


@pytest.mark.parametrize('name', CASES)
def test_decode_matches_reference(name, oracle_lib):
  case = golden_util.load_case(name)
  for run in case['runs']:
    out = oracle_lib.decode(case['params'], case['seqs'], run['beam_size'],
                            run['look_ahead'], run['test_iteration'],
                            n_threads=4)
    for u, ref_labels in enumerate(run['labels']):
      assert np.array_equal(out['labels'][u], ref_labels), (name, run, u)
    np.testing.assert_allclose(out['scores'], run['best'], rtol=RTOL)
    # the whole final beam, not just the winner
    fin = np.isfinite(run['beam'])
    assert np.array_equal(np.isfinite(out['beam_scores']), fin)
    np.testing.assert_allclose(out['beam_scores'][fin], run['beam'][fin],
                               rtol=RTOL)


@pytest.mark.parametrize('name', CASES)
def test_core_rnn_matches_reference(name, oracle_lib):
  """CoreRNN.forward (uisrnn/uisrnn.py:45-52) on recorded inputs."""
  case = golden_util.load_case(name)
  unit = case['unit']
  for x, h, mean_ref, h_ref in zip(unit['unit_x'], unit['unit_h'],
                                   unit['unit_mean'], unit['unit_hout']):
    mean, hout = oracle_lib.rnn_step(case['params'], x, h)
    np.testing.assert_allclose(mean, mean_ref, rtol=RTOL, atol=1e-6)
    np.testing.assert_allclose(hout, h_ref, rtol=RTOL, atol=1e-6)


@pytest.mark.parametrize('name', CASES)
def test_weighted_mse_matches_reference(name, oracle_lib):
  """loss_func.weighted_mse_loss incl. the exact-zero quirk (loss_func.py:36,41)."""
  case = golden_util.load_case(name)
  unit = case['unit']
  saw_inf = False
  for a, b, ref in zip(unit['mse_a'], unit['mse_b'], unit['mse_val']):
    got = oracle_lib.weighted_mse(case['params'], a, b)
    if np.isinf(ref):
      saw_inf = True
      assert np.isinf(got) and got > 0
    else:
      np.testing.assert_allclose(got, ref, rtol=1e-5)
  assert saw_inf  # the fixture contains a first-difference-zero row


def test_constants_are_rnn_of_zero_input(oracle_lib):
  """(m0, h1) = CoreRNN(0, rnn_init_hidden), uisrnn/uisrnn.py:435-439."""
  case = golden_util.load_case('toy_d2_depth2')  # non-zero rnn_init_hidden, depth 2
  params = case['params']
  m0, h1 = oracle_lib.constants(params)
  mean, hout = oracle_lib.rnn_step(
      params, np.zeros(params['observation_dim'], np.float32),
      params['rnn_init_hidden'])
  assert np.array_equal(m0, mean) and np.array_equal(h1, hout)


def test_edge_cases(oracle_lib):
  case = golden_util.load_case('tiny_d16')
  params = case['params']
  seq = case['seqs'][0]
  # no utterances, empty utterance, one frame
  out = oracle_lib.decode(params, [], 10, 1, 2)
  assert out['labels'] == []
  out = oracle_lib.decode(params, [seq[:0], seq[:1], seq], 10, 1, 2)
  assert len(out['labels'][0]) == 0 and out['scores'][0] == 0.0
  assert out['labels'][1].tolist() == [0] or len(out['labels'][1]) == 1
  # a list decodes like its elements one by one, in any thread count
  one = [oracle_lib.decode(params, [s], 10, 1, 2)['labels'][0]
         for s in case['seqs']]
  many = oracle_lib.decode(params, case['seqs'], 10, 1, 2, n_threads=3)
  for a, b in zip(one, many['labels']):
    assert np.array_equal(a, b)
  # beam 1 is greedy: still a valid labelling
  greedy = oracle_lib.decode(params, [seq], 1, 1, 1)['labels'][0]
  assert greedy[0] == 0 and greedy.max() <= len(set(greedy.tolist()))


def test_labels_are_first_appearance_ordered(oracle_lib):
  """With test_iteration=1 the trace starts at 0 and new ids appear in order (quirk 9)."""
  case = golden_util.load_case('d32_lookahead3')
  for look_ahead in (1, 2, 3):
    out = oracle_lib.decode(case['params'], case['seqs'], 5, look_ahead, 1)
    for labels in out['labels']:
      seen = -1
      for lab in labels:
        assert lab <= seen + 1
        seen = max(seen, lab)


def test_ragged_last_window(oracle_lib):
  """tau*N not a multiple of look_ahead: the last window is shorter (uisrnn.py:532-533)."""
  case = golden_util.load_case('d32_lookahead3')
  seq = case['seqs'][0][:7]
  out = oracle_lib.decode(case['params'], [seq], 4, 3, 1)
  assert len(out['labels'][0]) == 7
  assert np.isfinite(out['scores'][0])


def test_numerics_version_matches_header(oracle_lib):
  import os
  import re
  header = open(os.path.join(os.path.dirname(__file__), '..', 'include',
                             'uis_numerics.h')).read()
  version = int(re.search(r'#define UIS_NUMERICS_VERSION (\d+)', header).group(1))
  assert oracle_lib.numerics_version() == version


# ---- models the REFERENCE trained (tests/golden/make_trained.py): its own behavioural tests


@pytest.mark.parametrize('name', golden_util.trained_names())
def test_trained_fixtures_match_reference(name, oracle_lib):
  """Checkpoints written by the reference's fit() + save(); labels and scores recorded from its
  predict().  Where the oracle's labels differ, only the alternative the reference itself
  re-scored (make_trained.rescore_with_reference) is accepted -- golden_util.accept_labels."""
  case = golden_util.load_trained(name)
  beam, look, tau = case['cfg']
  out = oracle_lib.decode(case['params'], case['seqs'], beam, look, tau, n_threads=4)
  for u in range(len(case['seqs'])):
    assert golden_util.accept_labels(case, u, out['labels'][u], out['scores'][u]), (name, u)
    if np.array_equal(out['labels'][u], case['labels'][u]):
      np.testing.assert_allclose(out['scores'][u], case['best'][u], rtol=RTOL)
      fin = np.isfinite(case['beam'][u])
      np.testing.assert_allclose(out['beam_scores'][u][fin], case['beam'][u][fin], rtol=RTOL)


def test_trained_single_label_is_all_zeros(oracle_lib):
  """tests/uisrnn_test.py:26-70 of the reference: one training label -> predict gives [0]*10."""
  case = golden_util.load_trained('trained_single')
  for lab in case['labels']:
    assert lab.tolist() == [0] * 10
  out = oracle_lib.decode(case['params'], case['seqs'], *case['cfg'])
  assert all(l.tolist() == [0] * 10 for l in out['labels'])


def test_trained_toy4_accuracy_is_one(oracle_lib):
  """tests/integration_test.py:116-118 of the reference: four clusters, depth 2, accuracy 1.0."""
  from uisrnn_amd import evals
  case = golden_util.load_trained('trained_toy4')
  assert float(case['accuracy']) == 1.0
  out = oracle_lib.decode(case['params'], case['seqs'], *case['cfg'])
  acc = evals.compute_sequence_match_accuracy(out['labels'][0].tolist(), case['truth'].tolist())
  assert acc == 1.0


def test_exact_ties_are_broken_like_the_reference_does(oracle_lib):
  """probes.json 'exact_ties': candidates with EXACTLY equal scores (an all-zero network, priors
  only; quirk 8, uisrnn.py:549 np.argsort).  The reference's choice on this stack equals the
  decoder's rule -- lowest flat index first -- for score arrays on both sides of numpy's
  16-element insertion-sort threshold."""
  import json
  import os
  with open(os.path.join(golden_util.GOLDEN_DIR, 'probes.json')) as f:
    probes = json.load(f)
  assert len(probes['exact_ties']) >= 4
  for probe in probes['exact_ties']:
    dim, hidden, beam, n_frames, seed, look = probe['spec']
    params, seq = golden_util.tie_probe_case(dim, hidden, beam, n_frames, seed)
    out = oracle_lib.decode(params, [seq], beam, look, 1)
    assert out['labels'][0].tolist() == probe['labels'], probe['spec']
    assert len(set(probe['labels'])) > 1     # (not the trivial all-zeros answer)
  # ... and where numpy's unstable argsort picks ANOTHER of the equally good candidates (more of
  # them tie at once with crp_alpha 2 / 3): the decoder's labels differ from the reference's, and
  # the reference's own scorer gives both sequences the same float32 (recorded by make_golden.py)
  assert len(probes['exact_ties_unstable']) >= 3
  for probe in probes['exact_ties_unstable']:
    dim, hidden, beam, n_frames, seed, look = probe['spec']
    params, seq = golden_util.tie_probe_case(dim, hidden, beam, n_frames, seed)
    params['crp_alpha'] = probe['crp_alpha']
    out = oracle_lib.decode(params, [seq], beam, look, 1)
    assert out['labels'][0].tolist() == probe['decoder_labels'], probe['spec']
    assert probe['decoder_labels'] != probe['reference_labels']
    assert np.float32(probe['decoder_labels_rescored']) == np.float32(probe['reference_labels_rescored'])
    assert abs(float(out['scores'][0]) - probe['reference_labels_rescored']) <= 1e-4 * abs(probe['reference_labels_rescored'])


def test_reference_probes_at_the_edges(oracle_lib):
  """tests/golden/probes.json (make_golden.py --probes): what the reference does on a frame whose
  first component equals m0[0] (loss_func.py:36,41 -> the fresh-cluster candidate is inf) and on
  non-finite frames.  The oracle reproduces the finite cases exactly; where the reference's beam
  empties it raises (ValueError or IndexError, depending on the step) and the oracle returns -1
  labels, which the Python host turns into EmptyBeamError (both exception types)."""
  import json
  import os
  import uisrnn_amd
  with open(os.path.join(golden_util.GOLDEN_DIR, 'probes.json')) as f:
    probes = json.load(f)
  case = golden_util.load_case('tiny_d16')
  params, seq = case['params'], case['seqs'][0]
  m0, _ = oracle_lib.constants(params)

  def oracle_labels(x):
    return oracle_lib.decode(params, [x], 5, 1, 1)['labels'][0].tolist()

  assert oracle_labels(seq) == probes['clean']['labels']
  q = seq.copy(); q[4, 0] = np.float64(m0[0])
  assert oracle_labels(q) == probes['first_component_equals_m0_frame4']['labels']
  q = seq.copy(); q[0, 0] = np.float64(m0[0])
  assert probes['first_component_equals_m0_frame0']['raises'] == 'ValueError'
  assert set(oracle_labels(q)) == {-1}
  bad = seq.copy(); bad[3, 2] = np.nan
  assert probes['nan_mid_frame']['raises'] == 'ValueError'
  assert set(oracle_labels(bad)) == {-1}
  bad = seq.copy(); bad[0, 0] = np.inf
  assert probes['inf_first_frame']['raises'] == 'IndexError'
  assert set(oracle_labels(bad)) == {-1}
  assert issubclass(uisrnn_amd.EmptyBeamError, ValueError)
  assert issubclass(uisrnn_amd.EmptyBeamError, IndexError)
  # documented deviation: with NaN in the LAST frame the reference keeps NaN-scored hypotheses
  # (numpy sorts NaN behind inf, trim_zeros does not drop it) and returns labels; here the
  # beam is empty
  assert 'labels' in probes['nan_last_frame']


def _score_cases():
  data = np.load(golden_util.GOLDEN_DIR + '/fn_scores.npz')
  for i in range(int(data['n_cases'])):
    name = str(data['case_{}'.format(i)][0])
    utt, keep, beam, look, tau, cmax = (int(v) for v in data['cfg_{}'.format(i)])
    yield name, utt, keep, beam, look, tau, cmax, data['scores_{}'.format(i)], data['labels_{}'.format(i)]


def test_calculate_score_arrays_match_reference(oracle_lib):
  """UISRNN._calculate_score (uisrnn/uisrnn.py:455-477), array by array: every candidate score of
  every window as the reference returned it (tests/golden/make_scores.py), +inf padding included --
  not only the survivors' scores."""
  n_states = 0
  for name, utt, keep, beam, look, tau, cmax, ref, ref_labels in _score_cases():
    case = golden_util.load_trained(name) if name.startswith('trained_') else golden_util.load_case(name)
    seq = np.asarray(case['seqs'][utt], dtype=np.float64)[:keep]
    got = oracle_lib.candidate_scores(case['params'], seq, beam, look, tau, cmax)
    assert got.shape == ref.shape
    assert np.array_equal(np.isinf(got), np.isinf(ref)), (name, 'where the +inf padding sits')
    fin = np.isfinite(ref)
    np.testing.assert_allclose(got[fin], ref[fin], rtol=RTOL)
    out = oracle_lib.decode(case['params'], [seq], beam, look, tau)
    assert np.array_equal(out['labels'][0], ref_labels)
    n_states += int(np.isfinite(ref.reshape(ref.shape[0], ref.shape[1], -1)).any(axis=2).sum())
  assert n_states >= 20 * 5


def test_sequence_match_accuracy_matches_reference():
  """evals.compute_sequence_match_accuracy (uisrnn/evals.py:40-73) on 203 recorded pairs: the host
  mirror gives the reference's float64 accuracies exactly."""
  from uisrnn_amd import evals
  data = np.load(golden_util.GOLDEN_DIR + '/fn_evals.npz')
  pos = 0
  for n, acc in zip(data['lens'], data['accuracy']):
    a = data['a'][pos:pos + n].tolist()
    b = data['b'][pos:pos + n].tolist()
    pos += n
    assert evals.compute_sequence_match_accuracy(a, b) == acc
