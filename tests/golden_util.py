"""Load the fixtures written by tests/golden/make_golden.py."""

import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')

SEEDED_CASES = {
    # regenerated from seeds, see make_golden.py case_tracker
    'tracker_d256': dict(dim=256, hid=512, depth=1, seed=0, utt_seed=1000,
                         lengths=[40, 60, 25]),
    'tracker_d256_long': dict(dim=256, hid=512, depth=1, seed=0, utt_seed=1100,
                              lengths=[250]),
}


def case_names():
  return sorted(f[:-4] for f in os.listdir(GOLDEN_DIR) if f.endswith('.npz'))


def _params_from_npz(data):
  params = {}
  lists = {}
  for key in data.files:
    if not key.startswith('p_'):
      continue
    name = key[2:]
    base, _, idx = name.rpartition('_')
    if base in ('gru_weight_ih', 'gru_weight_hh', 'gru_bias_ih', 'gru_bias_hh'):
      lists.setdefault(base, {})[int(idx)] = data[key]
    else:
      val = data[key]
      params[name] = val.item() if val.ndim == 0 else val
  for base, items in lists.items():
    params[base] = [items[i] for i in range(len(items))]
  for key in ('observation_dim', 'rnn_hidden_size', 'rnn_depth'):
    params[key] = int(params[key])
  return params


def load_case(name):
  """Returns dict(params, seqs, runs=[dict(cfg, labels, best, beam)], unit)."""
  from uisrnn_amd import synth  # pylint: disable=import-outside-toplevel
  data = np.load(os.path.join(GOLDEN_DIR, name + '.npz'))
  n_utt = int(data['n_utt'])
  if name in SEEDED_CASES:
    spec = SEEDED_CASES[name]
    params = synth.tracker_params(spec['dim'], spec['hid'], spec['depth'],
                                  seed=spec['seed'])
    seqs, _ = synth.make_utterances(spec['utt_seed'], n_utt, spec['lengths'],
                                    spec['dim'])
  else:
    params = _params_from_npz(data)
    seqs = [data['seq_{}'.format(u)] for u in range(n_utt)]
  runs = []
  for r in range(int(data['n_runs'])):
    cfg = data['run{}_cfg'.format(r)]
    runs.append({
        'beam_size': int(cfg[0]), 'look_ahead': int(cfg[1]),
        'test_iteration': int(cfg[2]),
        'labels': [data['run{}_labels_{}'.format(r, u)] for u in range(n_utt)],
        'best': data['run{}_best'.format(r)],
        'beam': data['run{}_beam'.format(r)],
        'secs': data['run{}_secs'.format(r)],
    })
  unit = {k: data[k] for k in ('unit_x', 'unit_h', 'unit_mean', 'unit_hout',
                               'mse_a', 'mse_b', 'mse_val')}
  return {'params': params, 'seqs': seqs, 'runs': runs, 'unit': unit}
