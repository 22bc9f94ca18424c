"""Fixtures from models the REFERENCE itself trained (fit + save), and its predict() outputs.

Runs only in the dev container (imports /root/reference, see make_golden.py for the shims).
Nothing here is used by the product; the files it writes under tests/golden/ are what travels.

  python tests/golden/make_trained.py train_single     # tests/uisrnn_test.py:26-70 shape
  python tests/golden/make_trained.py train_toy4       # tests/integration_test.py:56-134 shape
  python tests/golden/make_trained.py train_d256       # SURVEY.md 8(d) model (D=256, H=512)
  python tests/golden/make_trained.py train_d512       # same recipe at D=512 (BASELINE configs[4])
  python tests/golden/make_trained.py predict_d256 100 # reference predict() on 100-frame utterances
  python tests/golden/make_trained.py predict_d256 500
  python tests/golden/make_trained.py predict_d256 1000
  python tests/golden/make_trained.py predict_d512 100 # the configs[4] shape: observation_dim 512, beam 20
  python tests/golden/make_trained.py predict_d256_l2 40 # the configs[2] shape: beam 50, look_ahead 2
  python tests/golden/make_trained.py predict_d256_l2 120  # ... 2 x 120 frames (round 5)
  python tests/golden/make_trained.py predict_d512 250     # configs[4] shape, 2 x 250 frames (round 5)
  python tests/golden/make_trained.py wholebox         # reference whole-box CPU rate (8 x 1 thread)

Outputs
  trained_single.uisrnn / .npz   D=16, H=8, depth 1, 50 iterations on 1000 single-label frames
                                 (reference test: predict must return [0]*10)
  trained_toy4.uisrnn / .npz     D=2, H=8, depth 2, 200 iterations, seeds 1/1/1
                                 (reference test: accuracy must be 1.0, also after load())
  trained_d256.uisrnn            D=256, H=512, depth 1, 300 iterations of the reference's fit on
                                 40 synthetic utterances x 300 frames (uisrnn_amd.synth, seeds
                                 5000..5039), np/random/torch seeds 1/1/1, lr 1e-3, batch 10
  trained_d256_n{100,500,1000}.npz  the reference's predict() on synth utterances with that model
  reference_cpu_rate.json        measured reference frames/s (one process and whole box)

Every .npz holds: the reference's label sequences, its best neg_likelihood and final beam, the
wall time of each predict, and -- the re-scoring harness of SURVEY.md section 7 hard part 1 --
for every utterance where the C oracle's label sequence differs from the reference's:
  alt_labels_<u>   the oracle's labels,
  alt_rescored_<u> the score the REFERENCE's own _update_beam_state (uisrnn/uisrnn.py:388-453)
                   gives the oracle's full trace, replayed step by step,
  alt_margin_<u>   the oracle's smallest relative decision margin on that utterance.
The tests accept such an alternative only if alt_rescored is within 1e-4 relative of the
reference's best score AND alt_margin shows a decision inside float32 rounding (<= 4 ulp).
"""

import json
import multiprocessing
import os
import random
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)

import make_golden  # noqa: E402  pylint: disable=wrong-import-position

D256_TRAIN_SEED = 5000
# 40 / 120: the look_ahead-2 fixtures (configs[2] shape; 120 frames = 240 decode steps, round 5)
D256_TEST_SEED = {100: 6100, 500: 6500, 1000: 7000, 40: 6040, 120: 6120}
D256_TEST_COUNT = {100: 4, 500: 2, 1000: 2, 40: 4, 120: 2}
D512_TEST_SEED = {100: 8100, 250: 8250}   # BASELINE configs[4]: observation_dim 512, beam 20 (250 frames: round 5)
D512_TEST_COUNT = {100: 4, 250: 2}


def _seed_all():
  import torch  # pylint: disable=import-outside-toplevel
  np.random.seed(1)
  random.seed(1)
  torch.manual_seed(1)


def _args(uisrnn):
  argv = sys.argv
  sys.argv = argv[:1]
  try:
    return uisrnn.parse_arguments()
  finally:
    sys.argv = argv


def _load_reference_model(uisrnn, model_args, path):
  """UISRNN.load with torch.load(weights_only=False)."""
  import torch  # pylint: disable=import-outside-toplevel
  model = uisrnn.UISRNN(model_args)
  orig = torch.load

  def load(*a, **kw):
    kw.setdefault('weights_only', False)
    return orig(*a, **kw)

  torch.load = load
  try:
    model.load(path)
  finally:
    torch.load = orig
  return model


def rescore_with_reference(model, seq, full_trace, test_iteration):
  """neg_likelihood the reference assigns to `full_trace` (cluster index per tiled frame).

  Replays uisrnn/uisrnn.py:388-453 one frame at a time on the reference's own BeamState.
  """
  import torch  # pylint: disable=import-outside-toplevel
  import uisrnn.uisrnn as ref  # pylint: disable=import-outside-toplevel
  tiled = np.tile(seq, (test_iteration, 1))
  tiled = torch.autograd.Variable(torch.from_numpy(tiled).float())
  state = ref.BeamState()
  with torch.no_grad():
    for t, cluster in enumerate(full_trace):
      state = model._update_beam_state(  # pylint: disable=protected-access
          state, tiled[t:t + 1, :], [int(cluster)])
  return float(state.neg_likelihood)


def oracle_alternatives(model, params, seqs, labels, run):
  """Where the oracle disagrees with the reference: its labels, re-scored by the reference."""
  from oracle import oracle  # pylint: disable=import-outside-toplevel
  out = {}
  res = oracle.decode(params, seqs, **run)
  for u, (seq, ref_lab) in enumerate(zip(seqs, labels)):
    out['oracle_margin_{}'.format(u)] = np.float32(res['margins'][u])
    if np.array_equal(res['labels'][u], ref_lab):
      continue
    tau = run['test_iteration']
    full = oracle.decode(params, [np.tile(seq, (tau, 1))], beam_size=run['beam_size'],
                         look_ahead=run['look_ahead'], test_iteration=1)
    trace = full['labels'][0]
    assert np.array_equal(trace[-len(seq):], res['labels'][u])
    out['alt_labels_{}'.format(u)] = res['labels'][u]
    out['alt_rescored_{}'.format(u)] = np.float32(
        rescore_with_reference(model, seq, trace, tau))
    out['alt_margin_{}'.format(u)] = np.float32(res['margins'][u])
    print('  utterance {}: oracle differs from the reference; rescored {} vs best {}'.format(
        u, out['alt_rescored_{}'.format(u)], 'see run_best'), flush=True)
  return out


def _record(model, inference_args, params, seqs, run, extra=None):
  labels, best, beams, secs = make_golden.run_reference(model, inference_args, seqs, **run)
  out = {'n_utt': np.int64(len(seqs)),
         'cfg': np.array([run['beam_size'], run['look_ahead'], run['test_iteration']],
                         dtype=np.int64),
         'best': best, 'beam': beams, 'secs': np.array(secs)}
  for u, lab in enumerate(labels):
    out['labels_{}'.format(u)] = lab
  out.update(oracle_alternatives(model, params, seqs, labels, run))
  if extra:
    out.update(extra)
  return out


def _params_of(path):
  from uisrnn_amd import weights  # pylint: disable=import-outside-toplevel
  return weights.load_checkpoint(path)


def train_single():
  """tests/uisrnn_test.py:26-70: one label, predict must be all zeros."""
  uisrnn = make_golden.import_reference()
  _seed_all()
  model_args, training_args, inference_args = _args(uisrnn)
  model_args.enable_cuda = False
  model_args.rnn_depth = 1
  model_args.rnn_hidden_size = 8
  model_args.observation_dim = 16
  model_args.verbosity = 0
  training_args.learning_rate = 0.01
  training_args.train_iteration = 50
  train_sequence = np.random.rand(1000, 16)
  train_cluster_id = np.array(['A'] * 1000)
  model = uisrnn.UISRNN(model_args)
  model.fit(train_sequence, train_cluster_id, training_args)
  path = os.path.join(HERE, 'trained_single.uisrnn')
  model.save(path)
  seqs = [np.random.rand(10, 16) / 10.0 for _ in range(3)]
  params = _params_of(path)
  run = dict(beam_size=10, look_ahead=1, test_iteration=1)
  out = _record(model, inference_args, params, seqs, run)
  for u, seq in enumerate(seqs):
    out['seq_{}'.format(u)] = seq
  np.savez_compressed(os.path.join(HERE, 'trained_single.npz'), **out)
  print('trained_single labels', [out['labels_{}'.format(u)].tolist() for u in range(3)])


def train_toy4():
  """tests/integration_test.py:56-134: four clusters on a square, depth 2, accuracy 1.0."""
  uisrnn = make_golden.import_reference()
  _seed_all()
  centers = {'A': np.array([0.0, 0.0]), 'B': np.array([0.0, 1.0]),
             'C': np.array([1.0, 0.0]), 'D': np.array([1.0, 1.0])}

  def gen(ids, sigma):
    pts = np.stack([centers[i] for i in ids])
    return pts + np.random.rand(*pts.shape) * sigma

  train_id = ['A'] * 400 + ['B'] * 300 + ['C'] * 200 + ['D'] * 100
  random.shuffle(train_id)
  train_seq = gen(train_id, 0.01)
  cuts = [0, 100, 300, 600, 1000]
  train_seqs = [train_seq[a:b] for a, b in zip(cuts[:-1], cuts[1:])]
  train_ids = [train_id[a:b] for a, b in zip(cuts[:-1], cuts[1:])]
  test_id = ['A'] * 10 + ['B'] * 20 + ['C'] * 30 + ['D'] * 40
  random.shuffle(test_id)
  test_seq = gen(test_id, 0.01)
  model_args, training_args, inference_args = _args(uisrnn)
  model_args.enable_cuda = False
  model_args.rnn_depth = 2
  model_args.rnn_hidden_size = 8
  model_args.observation_dim = 2
  model_args.verbosity = 0
  training_args.learning_rate = 0.01
  training_args.train_iteration = 200
  training_args.enforce_cluster_id_uniqueness = False
  model = uisrnn.UISRNN(model_args)
  model.fit(train_seqs, train_ids, training_args)
  path = os.path.join(HERE, 'trained_toy4.uisrnn')
  model.save(path)
  params = _params_of(path)
  run = dict(beam_size=10, look_ahead=1, test_iteration=2)
  out = _record(model, inference_args, params, [test_seq], run)
  out['seq_0'] = test_seq
  out['truth_0'] = np.array(test_id)
  acc = uisrnn.compute_sequence_match_accuracy(out['labels_0'].tolist(), test_id)
  out['accuracy'] = np.float64(acc)
  np.savez_compressed(os.path.join(HERE, 'trained_toy4.npz'), **out)
  print('trained_toy4 accuracy', acc)


def _d256_args(uisrnn, dim=256):
  model_args, training_args, inference_args = _args(uisrnn)
  model_args.enable_cuda = False
  model_args.observation_dim = dim
  model_args.rnn_hidden_size = 512
  model_args.rnn_depth = 1
  model_args.verbosity = 0
  return model_args, training_args, inference_args


def train_d256(dim=256):
  """the reference's fit, 300 iterations, on synthetic d-vectors
  (dim 512: the model of BASELINE configs[4], same recipe)."""
  from uisrnn_amd import synth  # pylint: disable=import-outside-toplevel
  uisrnn = make_golden.import_reference()
  _seed_all()
  model_args, training_args, _ = _d256_args(uisrnn, dim)
  training_args.learning_rate = 1e-3
  training_args.train_iteration = 300
  training_args.batch_size = 10
  seqs, ids = synth.make_utterances(D256_TRAIN_SEED, 40, 300, dim)
  ids = [['s{}'.format(int(i)) for i in row] for row in ids]
  model = uisrnn.UISRNN(model_args)
  t0 = time.time()
  model.fit(seqs, ids, training_args)
  print('fit: {:.0f}s, transition_bias {}, sigma2 mean {}'.format(
      time.time() - t0, model.transition_bias, float(model.sigma2.mean())), flush=True)
  model.save(os.path.join(HERE, 'trained_d{}.uisrnn'.format(dim)))


def _predict_d256_one(job):
  """Worker: one utterance through the reference (one torch thread)."""
  n_frames, u, dim, beam, look = job
  import torch  # pylint: disable=import-outside-toplevel
  torch.set_num_threads(1)
  from uisrnn_amd import synth  # pylint: disable=import-outside-toplevel
  uisrnn = make_golden.import_reference()
  model_args, _, inference_args = _d256_args(uisrnn, dim)
  path = os.path.join(HERE, 'trained_d{}.uisrnn'.format(dim))
  model = _load_reference_model(uisrnn, model_args, path)
  seeds = D256_TEST_SEED if dim == 256 else D512_TEST_SEED
  seq, truth = synth.make_utterance(seeds[n_frames] + u, n_frames, dim)
  run = dict(beam_size=beam, look_ahead=look, test_iteration=2)
  out = _record(model, inference_args, _params_of(path), [seq], run)
  acc = uisrnn.compute_sequence_match_accuracy(out['labels_0'].tolist(),
                                               [str(i) for i in truth])
  print('d={} n={} u={} secs {:.0f} accuracy {:.3f} best {}'.format(
      dim, n_frames, u, out['secs'][0], acc, out['best'][0]), flush=True)
  out['accuracy'] = np.float64(acc)
  return out


def predict_d256(n_frames, dim=256, beam=10, look=1, tag=''):
  """predict() of the reference on the trained model of that dim (256: beam 10; 512: the
  BASELINE configs[4] shape, beam 20), utterances regenerated from their seeds by the tests."""
  seeds, counts = (D256_TEST_SEED, D256_TEST_COUNT) if dim == 256 else (D512_TEST_SEED, D512_TEST_COUNT)
  count = counts[n_frames]
  with multiprocessing.get_context('spawn').Pool(min(count, 4)) as pool:
    parts = pool.map(_predict_d256_one, [(n_frames, u, dim, beam, look) for u in range(count)])
  out = {'n_utt': np.int64(count), 'cfg': parts[0]['cfg'], 'dim': np.int64(dim),
         'utt_seed': np.int64(seeds[n_frames]), 'n_frames': np.int64(n_frames),
         'best': np.concatenate([p['best'] for p in parts]),
         'beam': np.concatenate([p['beam'] for p in parts]),
         'secs': np.concatenate([p['secs'] for p in parts]),
         'accuracy': np.array([p['accuracy'] for p in parts])}
  for u, p in enumerate(parts):
    out['labels_{}'.format(u)] = p['labels_0']
    out['oracle_margin_{}'.format(u)] = p['oracle_margin_0']
    for key in ('alt_labels', 'alt_rescored', 'alt_margin'):
      if key + '_0' in p:
        out['{}_{}'.format(key, u)] = p[key + '_0']
  np.savez_compressed(os.path.join(HERE, 'trained_d{}{}_n{}.npz'.format(dim, tag, n_frames)), **out)
  print('wrote trained_d{}{}_n{}'.format(dim, tag, n_frames))


def _wholebox_worker(job):
  seed, n_frames = job
  import torch  # pylint: disable=import-outside-toplevel
  torch.set_num_threads(1)
  from uisrnn_amd import synth  # pylint: disable=import-outside-toplevel
  uisrnn = make_golden.import_reference()
  model_args, _, inference_args = _d256_args(uisrnn)
  model = _load_reference_model(uisrnn, model_args, os.path.join(HERE, 'trained_d256.uisrnn'))
  seq, _ = synth.make_utterance(seed, n_frames, 256)
  inference_args.beam_size, inference_args.look_ahead, inference_args.test_iteration = 10, 1, 2
  t0 = time.time()
  model.predict(seq, inference_args)
  return time.time() - t0


def wholebox():
  """os.cpu_count() processes x 1 torch thread, the structure of
  uisrnn/uisrnn.py:616-622; frames/s = sum of frames / slowest worker's wall."""
  cores = os.cpu_count()
  n_frames = 100
  jobs = [(8000 + u, n_frames) for u in range(cores)]
  with multiprocessing.get_context('spawn').Pool(cores) as pool:
    walls = pool.map(_wholebox_worker, jobs)
  one = _wholebox_worker((8000, n_frames))
  rec = {
      'box': 'dev container: {} vCPU {}'.format(cores, _cpu_name()),
      'cores': cores,
      'model': 'trained_d256.uisrnn (D=256, H=512, depth 1)',
      'workload': '{} utterances x {} frames, beam 10, look_ahead 1, test_iteration 2'.format(
          cores, n_frames),
      'whole_box_frames_per_s': cores * n_frames / max(walls),
      'whole_box_worker_walls_s': walls,
      'one_process_one_thread_frames_per_s': n_frames / one,
      'one_process_wall_s': one,
  }
  with open(os.path.join(HERE, 'reference_cpu_rate.json'), 'w') as f:
    json.dump(rec, f, indent=1)
  print(json.dumps(rec, indent=1))


def _cpu_name():
  try:
    with open('/proc/cpuinfo') as f:
      for line in f:
        if line.startswith('model name'):
          return line.split(':', 1)[1].strip()
  except OSError:
    pass
  return 'unknown'


def main():
  cmd = sys.argv[1]
  if cmd == 'train_single':
    train_single()
  elif cmd == 'train_toy4':
    train_toy4()
  elif cmd == 'train_d256':
    train_d256()
  elif cmd == 'train_d512':
    train_d256(512)
  elif cmd == 'predict_d256':
    predict_d256(int(sys.argv[2]))
  elif cmd == 'predict_d256_l2':  # the configs[2] shape: beam 50, look_ahead 2
    predict_d256(int(sys.argv[2]), beam=50, look=2, tag='_l2')
  elif cmd == 'predict_d512':
    predict_d256(int(sys.argv[2]), dim=512, beam=20)
  elif cmd == 'wholebox':
    wholebox()
  else:
    raise SystemExit('unknown command ' + cmd)


if __name__ == '__main__':
  main()
