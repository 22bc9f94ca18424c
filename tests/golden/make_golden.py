"""Generate the golden fixtures in tests/golden/ by running the REFERENCE itself.

Runs only in the dev container (it imports /root/reference); the fixtures it
writes are committed and travel to the GPU box, the reference does not.

  python tests/golden/make_golden.py            # all cases
  python tests/golden/make_golden.py tiny_d16   # one case
  python tests/golden/make_golden.py --probes   # probes.json: the reference at the edges (nan frames, quirk 7)

What is recorded per case (an .npz):
  * the model parameters (small models) or the seed that regenerates them
    through uisrnn_amd.synth / uisrnn_amd.weights (D=256/H=512 models),
  * the test sequences (small) or their generator seeds,
  * the reference's predict() label sequences, the best hypothesis'
    neg_likelihood and the whole final beam's neg_likelihoods, captured by
    wrapping UISRNN._update_beam_state (uisrnn/uisrnn.py:388) -- the calls made
    outside _calculate_score after the last window are the final beam in rank
    order (uisrnn/uisrnn.py:551-559),
  * unit-level vectors: CoreRNN.forward in/out and weighted_mse_loss values.

Shims needed to import the reference here: a stub
`colortimelog` module and an empty sys.argv.  The reference's code is not
modified or copied.
"""

import os
import sys
import time
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
REFERENCE = '/root/reference'


def import_reference():
  """Import the reference package with the two shims."""
  if 'colortimelog' not in sys.modules:
    stub = types.ModuleType('colortimelog')

    class Logger:  # the three members the reference touches
      def __init__(self, verbosity):
        self.verbosity = verbosity

      def print(self, level, msg):
        del level, msg

      def info(self, msg):
        del msg

    stub.Logger = Logger
    sys.modules['colortimelog'] = stub
  if REFERENCE not in sys.path:
    sys.path.insert(0, REFERENCE)
  argv = sys.argv
  sys.argv = argv[:1]
  try:
    import uisrnn  # pylint: disable=import-outside-toplevel
  finally:
    sys.argv = argv
  return uisrnn


def reference_model(uisrnn, params):
  """A reference UISRNN carrying exactly `params` (uisrnn_amd.weights dict)."""
  import torch  # pylint: disable=import-outside-toplevel
  from uisrnn_amd import weights  # pylint: disable=import-outside-toplevel
  argv = sys.argv
  sys.argv = argv[:1]
  try:
    model_args, _, inference_args = uisrnn.parse_arguments()
  finally:
    sys.argv = argv
  model_args.observation_dim = params['observation_dim']
  model_args.rnn_hidden_size = params['rnn_hidden_size']
  model_args.rnn_depth = params['rnn_depth']
  model_args.rnn_dropout = 0.0
  model_args.enable_cuda = False
  model_args.verbosity = 0
  model_args.crp_alpha = params['crp_alpha']
  model_args.transition_bias = params['transition_bias']
  model_args.sigma2 = 1.0
  model = uisrnn.UISRNN(model_args)
  state = {k: torch.from_numpy(np.array(v))
           for k, v in weights.state_dict_from_params(params).items()}
  model.rnn_model.load_state_dict(state)
  depth, hid = params['rnn_depth'], params['rnn_hidden_size']
  model.rnn_init_hidden = torch.nn.Parameter(torch.from_numpy(
      np.array(params['rnn_init_hidden'], dtype=np.float32).reshape(
          depth, 1, hid)))
  model.sigma2 = torch.nn.Parameter(
      torch.from_numpy(np.array(params['sigma2'], dtype=np.float32)))
  return model, inference_args


def run_reference(model, inference_args, seqs, beam_size, look_ahead,
                  test_iteration):
  """predict() per utterance, capturing the final beam's scores."""
  inference_args.beam_size = beam_size
  inference_args.look_ahead = look_ahead
  inference_args.test_iteration = test_iteration
  orig_update = model._update_beam_state  # pylint: disable=protected-access
  orig_score = model._calculate_score  # pylint: disable=protected-access
  state = {'in_score': False, 'tail': []}

  def update(beam_state, look_ahead_seq, cluster_seq):
    out = orig_update(beam_state, look_ahead_seq, cluster_seq)
    if not state['in_score']:
      state['tail'].append(float(out.neg_likelihood))
    return out

  def score(beam_state, look_ahead_seq):
    state['in_score'] = True
    state['tail'] = []
    try:
      return orig_score(beam_state, look_ahead_seq)
    finally:
      state['in_score'] = False

  model._update_beam_state = update  # pylint: disable=protected-access
  model._calculate_score = score  # pylint: disable=protected-access
  labels, best, beams, secs = [], [], [], []
  try:
    for seq in seqs:
      t0 = time.time()
      lab = model.predict(seq, inference_args)
      secs.append(time.time() - t0)
      labels.append(np.array([int(x) for x in lab], dtype=np.int32))
      tail = state['tail']
      row = np.full(beam_size, np.inf, dtype=np.float32)
      row[:len(tail)] = np.array(tail, dtype=np.float32)
      beams.append(row)
      best.append(np.float32(tail[0]) if tail else np.float32(np.inf))
  finally:
    model._update_beam_state = orig_update  # pylint: disable=protected-access
    model._calculate_score = orig_score  # pylint: disable=protected-access
  return labels, np.array(best, dtype=np.float32), np.stack(beams), secs


def unit_vectors(model, params, rng, count=6):
  """CoreRNN.forward and weighted_mse_loss samples from the reference."""
  import torch  # pylint: disable=import-outside-toplevel
  from uisrnn import loss_func  # pylint: disable=import-outside-toplevel
  dim, hid, depth = (params['observation_dim'], params['rnn_hidden_size'],
                     params['rnn_depth'])
  xs = (rng.standard_normal((count, dim)) * 0.3).astype(np.float32)
  hs = (rng.standard_normal((count, depth, hid)) * 0.5).astype(np.float32)
  xs[0] = 0.0
  hs[0] = np.array(params['rnn_init_hidden'], dtype=np.float32)
  means, houts, mses = [], [], []
  weight = 1 / (2 * model.sigma2)
  with torch.no_grad():
    for x, h in zip(xs, hs):
      mean, hout = model.rnn_model(
          torch.from_numpy(x).view(1, 1, dim),
          torch.from_numpy(h).view(depth, 1, hid))
      means.append(mean.view(dim).numpy().copy())
      houts.append(hout.view(depth, hid).numpy().copy())
  mse_a = (rng.standard_normal((count, dim)) * 0.2).astype(np.float32)
  mse_b = (rng.standard_normal((count, dim)) * 0.2).astype(np.float32)
  mse_b[1, 0] = mse_a[1, 0]  # quirk: first squared difference exactly zero
  for a, b in zip(mse_a, mse_b):
    mses.append(loss_func.weighted_mse_loss(
        input_tensor=torch.from_numpy(a), target_tensor=torch.from_numpy(b),
        weight=weight).detach().numpy())
  return {
      'unit_x': xs, 'unit_h': hs, 'unit_mean': np.stack(means),
      'unit_hout': np.stack(houts), 'mse_a': mse_a, 'mse_b': mse_b,
      'mse_val': np.array(mses, dtype=np.float32)}


def flat_params(params):
  """params dict -> flat {name: array} for np.savez."""
  out = {}
  for key, val in params.items():
    if isinstance(val, list):
      for l, arr in enumerate(val):
        out['p_{}_{}'.format(key, l)] = np.asarray(arr)
    elif val is None:
      continue
    else:
      out['p_' + key] = np.asarray(val)
  return out


CASES = {}


def case(name):
  def deco(fn):
    CASES[name] = fn
    return fn
  return deco


def random_small_params(dim, hid, depth, seed, sigma2, transition_bias,
                        crp_alpha=1.0, init_hidden_scale=0.0):
  from uisrnn_amd import weights  # pylint: disable=import-outside-toplevel
  params = weights.init_params(dim, hid, depth, sigma2=sigma2,
                               transition_bias=transition_bias,
                               crp_alpha=crp_alpha, seed=seed)
  if init_hidden_scale:
    rng = np.random.default_rng(seed + 1)
    params['rnn_init_hidden'] = (
        init_hidden_scale * rng.standard_normal((depth, hid))).astype(
            np.float32)
  return params


def clustered_sequences(rng, count, length, dim, centers, noise):
  """Few Gaussian blobs, random turn-taking; float64 like the reference's input."""
  cents = rng.standard_normal((centers, dim))
  seqs = []
  for _ in range(count):
    n = int(length if np.isscalar(length) else rng.integers(*length))
    ids = np.repeat(rng.integers(0, centers, size=n // 3 + 1), 3)[:n]
    seqs.append((cents[ids] + noise * rng.standard_normal((n, dim))).astype(
        np.float64))
  return seqs


@case('tiny_d16')
def case_tiny_d16(uisrnn):
  """Shape of tests/uisrnn_test.py (D=16, H=8, depth=1); tau = 1 and 2."""
  params = random_small_params(16, 8, 1, seed=11, sigma2=0.05,
                               transition_bias=0.2)
  rng = np.random.default_rng(12)
  seqs = clustered_sequences(rng, 4, (8, 30), 16, 3, 0.1)
  seqs.append(rng.random((10, 16)) / 10.0)  # like tests/uisrnn_test.py:45
  seqs.append(rng.random((1, 16)))          # single-frame utterance
  runs = [dict(beam_size=10, look_ahead=1, test_iteration=1),
          dict(beam_size=10, look_ahead=1, test_iteration=2),
          dict(beam_size=3, look_ahead=1, test_iteration=2),
          dict(beam_size=1, look_ahead=1, test_iteration=2)]
  return params, seqs, runs, True


@case('toy_d2_depth2')
def case_toy_d2(uisrnn):
  """Shape of tests/integration_test.py (D=2, H=8, depth=2), look_ahead 1..2."""
  params = random_small_params(2, 8, 2, seed=21, sigma2=0.02,
                               transition_bias=0.3, crp_alpha=0.7,
                               init_hidden_scale=0.3)
  rng = np.random.default_rng(22)
  seqs = clustered_sequences(rng, 3, (20, 41), 2, 4, 0.05)
  runs = [dict(beam_size=6, look_ahead=1, test_iteration=2),
          dict(beam_size=6, look_ahead=2, test_iteration=2),
          dict(beam_size=4, look_ahead=2, test_iteration=1)]
  return params, seqs, runs, True


@case('d32_lookahead3')
def case_d32(uisrnn):
  """D=32, H=32: look_ahead=3 with a ragged last window (tau*N % 3 != 0)."""
  params = random_small_params(32, 32, 1, seed=31, sigma2=0.05,
                               transition_bias=0.15)
  rng = np.random.default_rng(32)
  seqs = clustered_sequences(rng, 2, 31, 32, 3, 0.1)
  seqs += clustered_sequences(rng, 1, 10, 32, 2, 0.1)
  runs = [dict(beam_size=4, look_ahead=3, test_iteration=1),
          dict(beam_size=5, look_ahead=2, test_iteration=2),
          dict(beam_size=10, look_ahead=1, test_iteration=2)]
  return params, seqs, runs, True


@case('d20_h24_depth3')
def case_d20(uisrnn):
  """Dimensions that are not multiples of 16 and three GRU layers."""
  params = random_small_params(20, 24, 3, seed=41, sigma2=0.08,
                               transition_bias=0.1, crp_alpha=2.0,
                               init_hidden_scale=0.2)
  rng = np.random.default_rng(42)
  seqs = clustered_sequences(rng, 3, (12, 36), 20, 3, 0.15)
  runs = [dict(beam_size=10, look_ahead=1, test_iteration=2),
          dict(beam_size=4, look_ahead=2, test_iteration=2)]
  return params, seqs, runs, True


@case('tracker_d256')
def case_tracker(uisrnn):
  """BASELINE config shape: D=256, H=512, beam 10 -- synthetic tracker model.

  Parameters and utterances are regenerated from seeds (6.3 MB of weights do
  not belong in a fixture); only the reference's outputs are stored.
  """
  from uisrnn_amd import synth  # pylint: disable=import-outside-toplevel
  params = synth.tracker_params(256, 512, 1, seed=0)
  seqs, _ = synth.make_utterances(1000, 3, [40, 60, 25], 256)
  runs = [dict(beam_size=10, look_ahead=1, test_iteration=2),
          dict(beam_size=4, look_ahead=2, test_iteration=1)]
  return params, seqs, runs, False


@case('tracker_d256_long')
def case_tracker_long(uisrnn):
  """One 250-frame D=256/H=512 utterance (500 decode steps): near-ties accumulate with length."""
  from uisrnn_amd import synth  # pylint: disable=import-outside-toplevel
  params = synth.tracker_params(256, 512, 1, seed=0)
  seqs, _ = synth.make_utterances(1100, 1, [250], 256)
  runs = [dict(beam_size=10, look_ahead=1, test_iteration=2)]
  return params, seqs, runs, False


@case('tracker_d64_h300')
def case_tracker_h300(uisrnn):
  """rnn_hidden_size 300 (19 k-blocks: canonical segments of THREE): the size class the library embeds in its
  512-wide kernels with a zero k-block behind every segment (round 6, uis_decoder.hip: HidMap).  Recorded from the
  reference so that the embedding is pinned against google/uis-rnn itself, not only against the oracle.
  Parameters and utterances are regenerated from seeds."""
  from uisrnn_amd import synth  # pylint: disable=import-outside-toplevel
  params = synth.tracker_params(64, 300, 1, seed=3)
  seqs, _ = synth.make_utterances(1200, 3, [40, 25, 33], 64)
  runs = [dict(beam_size=10, look_ahead=1, test_iteration=2),
          dict(beam_size=5, look_ahead=2, test_iteration=1)]
  return params, seqs, runs, False


CHECKPOINT_CASE = 'd20_h24_depth3'  # depth 3, non-zero rnn_init_hidden


def write_reference_checkpoint(uisrnn):
  """<case>.uisrnn: the case's model written by the REFERENCE's own UISRNN.save()
  (uisrnn/uisrnn.py:135-147) -- the file format uisrnn_amd.weights reads without torch."""
  params = CASES[CHECKPOINT_CASE](uisrnn)[0]
  model, _ = reference_model(uisrnn, params)
  path = os.path.join(HERE, CHECKPOINT_CASE + '.uisrnn')
  model.save(path)
  print('wrote', path, os.path.getsize(path), 'bytes')


def write_probes(uisrnn):
  """probes.json: what the REFERENCE does at the edges the decoder documents as deviations or
  quirks -- non-finite frames (which exception, depending on where the beam empties:
  uisrnn/uisrnn.py:531 `max()` of an empty beam vs :561 `beam_set[0]`) and a frame whose first
  component equals m0[0] (loss_func.py:36,41: the fresh-cluster candidate becomes inf)."""
  import json  # pylint: disable=import-outside-toplevel
  from oracle import oracle  # pylint: disable=import-outside-toplevel
  params = CASES['tiny_d16'](uisrnn)[0]
  seqs = CASES['tiny_d16'](uisrnn)[1]
  model, inference_args = reference_model(uisrnn, params)
  inference_args.beam_size, inference_args.look_ahead, inference_args.test_iteration = 5, 1, 1
  seq = seqs[0]
  m0, _ = oracle.constants(params)

  def run(x):
    try:
      return {'labels': [int(v) for v in model.predict(x, inference_args)]}
    except Exception as exc:  # pylint: disable=broad-except
      return {'raises': type(exc).__name__, 'message': str(exc)}

  out = {'case': 'tiny_d16 utterance 0, beam 5, look_ahead 1, test_iteration 1', 'n_frames': len(seq)}
  bad = seq.copy(); bad[3, 2] = np.nan
  out['nan_mid_frame'] = run(bad)
  bad = seq.copy(); bad[len(seq) - 1, 2] = np.nan
  out['nan_last_frame'] = run(bad)
  bad = seq.copy(); bad[0, 0] = np.inf
  out['inf_first_frame'] = run(bad)
  q = seq.copy(); q[0, 0] = np.float64(m0[0])
  out['first_component_equals_m0_frame0'] = run(q)
  q = seq.copy(); q[4, 0] = np.float64(m0[0])
  out['first_component_equals_m0_frame4'] = run(q)
  out['clean'] = run(seq)
  # exact ties (np.argsort's order among equal scores, uisrnn.py:549).  An
  # all-zero network makes every cluster mean exactly 0, so candidates differ by their priors only;
  # with crp_alpha 1 "a new cluster" and "back to a cluster seen in one block" tie exactly, from
  # the third frame on.  The decoder's rule is lowest flat index first: recorded here is what the
  # reference does on this stack (numpy's argsort), for beams whose flattened score arrays are
  # shorter and longer than the 16 elements below which numpy's quicksort is an insertion sort.
  sys.path.insert(0, os.path.join(REPO, "tests"))
  from golden_util import tie_probe_case  # pylint: disable=import-outside-toplevel
  ties = []
  # spec = (observation_dim, rnn_hidden_size, beam_size, frames, seed, look_ahead)
  for spec in ((4, 4, 3, 10, 0, 1), (4, 4, 10, 14, 2, 1), (4, 4, 10, 30, 3, 1), (6, 8, 20, 40, 4, 1),
               (4, 4, 3, 9, 5, 2), (4, 4, 6, 12, 6, 2), (4, 4, 4, 10, 7, 3)):
    tie_params, tie_seq = tie_probe_case(*spec[:5])
    tie_model, tie_args = reference_model(uisrnn, tie_params)
    tie_args.beam_size, tie_args.look_ahead, tie_args.test_iteration = spec[2], spec[5], 1
    ties.append({'spec': list(spec), 'labels': [int(v) for v in tie_model.predict(tie_seq, tie_args)]})
  out['exact_ties'] = ties
  # ... and where it does NOT: with crp_alpha 2 or 3 more candidates tie at once, and numpy's
  # argsort (AVX-512 / introsort above 16 elements, not stable) picks another of the equally good
  # ones than lowest-index-first.  Recorded: both label sequences and what the REFERENCE's own
  # _update_beam_state (uisrnn.py:388-453) scores them at -- the same float32, bit for bit.
  from make_trained import rescore_with_reference  # pylint: disable=import-outside-toplevel
  unstable = []
  for alpha, spec in ((2.0, (4, 4, 30, 30, 12, 1)), (2.0, (4, 4, 5, 12, 13, 2)), (3.0, (4, 4, 30, 30, 12, 1))):
    tie_params, tie_seq = tie_probe_case(*spec[:5])
    tie_params['crp_alpha'] = alpha
    tie_model, tie_args = reference_model(uisrnn, tie_params)
    tie_args.beam_size, tie_args.look_ahead, tie_args.test_iteration = spec[2], spec[5], 1
    theirs = [int(v) for v in tie_model.predict(tie_seq, tie_args)]
    ours = oracle.decode(tie_params, [tie_seq], spec[2], spec[5], 1)['labels'][0].tolist()
    unstable.append({'spec': list(spec), 'crp_alpha': alpha, 'reference_labels': theirs, 'decoder_labels': ours,
                     'reference_labels_rescored': rescore_with_reference(tie_model, tie_seq, theirs, 1),
                     'decoder_labels_rescored': rescore_with_reference(tie_model, tie_seq, ours, 1)})
  out['exact_ties_unstable'] = unstable
  out['stack'] = 'numpy {} on a CPU with AVX-512 (np.argsort of float64: x86-simd-sort)'.format(np.__version__)
  with open(os.path.join(HERE, 'probes.json'), 'w') as f:
    json.dump(out, f, indent=1)
  print(json.dumps(out, indent=1))


def main():
  uisrnn = import_reference()
  import torch  # pylint: disable=import-outside-toplevel
  torch.set_num_threads(1)
  if sys.argv[1:] == ['--checkpoint']:
    write_reference_checkpoint(uisrnn)
    return
  if sys.argv[1:] == ['--probes']:
    write_probes(uisrnn)
    return
  names = sys.argv[1:] or list(CASES)
  for name in names:
    params, seqs, runs, store_inputs = CASES[name](uisrnn)
    model, inference_args = reference_model(uisrnn, params)
    out = {'n_runs': np.int64(len(runs)), 'n_utt': np.int64(len(seqs))}
    if store_inputs:
      out.update(flat_params(params))
      for u, seq in enumerate(seqs):
        out['seq_{}'.format(u)] = seq
    out.update(unit_vectors(model, params, np.random.default_rng(5)))
    for r, run in enumerate(runs):
      t0 = time.time()
      labels, best, beams, secs = run_reference(model, inference_args, seqs,
                                                **run)
      out['run{}_cfg'.format(r)] = np.array(
          [run['beam_size'], run['look_ahead'], run['test_iteration']],
          dtype=np.int64)
      for u, lab in enumerate(labels):
        out['run{}_labels_{}'.format(r, u)] = lab
      out['run{}_best'.format(r)] = best
      out['run{}_beam'.format(r)] = beams
      out['run{}_secs'.format(r)] = np.array(secs)
      print('{} run {} {}: {:.1f}s, frames/s {:.2f}'.format(
          name, r, run, time.time() - t0,
          sum(len(s) for s in seqs) / max(sum(secs), 1e-9)), flush=True)
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)
    print('wrote', name)


if __name__ == '__main__':
  main()
