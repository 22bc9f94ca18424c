"""Model parameters of the decode path as plain numpy arrays.

The reference keeps them in a torch ``CoreRNN`` module plus a few attributes
(uisrnn/uisrnn.py:32-52,83-107) and checkpoints them with ``torch.save``
(uisrnn/uisrnn.py:135-147).  The decoder only needs the float32 arrays, so the
host keeps a dict:

  observation_dim, rnn_hidden_size, rnn_depth            ints
  gru_weight_ih[l] (3H, D|H)   gru_weight_hh[l] (3H, H)  gate order r|z|n
  gru_bias_ih[l] (3H)          gru_bias_hh[l] (3H)
  linear_mean1_weight (H, H)   linear_mean1_bias (H)
  linear_mean2_weight (D, H)   linear_mean2_bias (D)
  rnn_init_hidden (depth, H)   sigma2 (D)
  transition_bias, crp_alpha                             python floats
  transition_bias_denominator                            float (carried for save())

PyTorch is used only to read / write the reference's checkpoint format.
"""

import numpy as np

_INITIAL_SIGMA2_VALUE = 0.1  # uisrnn/uisrnn.py:29


def init_params(observation_dim, rnn_hidden_size, rnn_depth, sigma2=None,
                transition_bias=None, crp_alpha=1.0, seed=None):
  """Fresh parameters, distributed like the reference constructor's.

  nn.GRU / nn.Linear draw U(-1/sqrt(fan), 1/sqrt(fan)); rnn_init_hidden is
  zero; sigma2 is 0.1 unless given (uisrnn/uisrnn.py:92-106).  The draws are
  numpy's, not torch's: a fresh model is random either way.
  """
  rng = np.random.default_rng(seed)
  dim, hid, depth = int(observation_dim), int(rnn_hidden_size), int(rnn_depth)

  def uni(shape, fan):
    k = 1.0 / np.sqrt(fan)
    return rng.uniform(-k, k, size=shape).astype(np.float32)

  in_dims = [dim] + [hid] * (depth - 1)
  return {
      'observation_dim': dim,
      'rnn_hidden_size': hid,
      'rnn_depth': depth,
      'gru_weight_ih': [uni((3 * hid, in_dims[l]), hid) for l in range(depth)],
      'gru_weight_hh': [uni((3 * hid, hid), hid) for l in range(depth)],
      'gru_bias_ih': [uni((3 * hid,), hid) for l in range(depth)],
      'gru_bias_hh': [uni((3 * hid,), hid) for l in range(depth)],
      'linear_mean1_weight': uni((hid, hid), hid),
      'linear_mean1_bias': uni((hid,), hid),
      'linear_mean2_weight': uni((dim, hid), hid),
      'linear_mean2_bias': uni((dim,), hid),
      'rnn_init_hidden': np.zeros((depth, hid), dtype=np.float32),
      'sigma2': np.full(
          (dim,), _INITIAL_SIGMA2_VALUE if sigma2 is None else sigma2,
          dtype=np.float32),
      'transition_bias': transition_bias,
      'transition_bias_denominator': 0.0,
      'crp_alpha': crp_alpha,
  }


def _np32(value):
  if hasattr(value, 'detach'):
    value = value.detach().cpu().numpy()
  return np.ascontiguousarray(np.asarray(value, dtype=np.float32))


def params_from_state(rnn_state_dict, rnn_init_hidden, sigma2, transition_bias,
                      crp_alpha, transition_bias_denominator=0.0):
  """Convert the pieces of a reference model / checkpoint to the params dict.

  rnn_state_dict has the CoreRNN keys (uisrnn/uisrnn.py:35-43):
  gru.weight_ih_l{k}, gru.weight_hh_l{k}, gru.bias_ih_l{k}, gru.bias_hh_l{k},
  linear_mean1.weight/.bias, linear_mean2.weight/.bias.
  """
  depth = 0
  while 'gru.weight_ih_l{}'.format(depth) in rnn_state_dict:
    depth += 1
  if depth == 0:
    raise ValueError('rnn_state_dict has no gru.weight_ih_l0')
  w_hh0 = _np32(rnn_state_dict['gru.weight_hh_l0'])
  hid = w_hh0.shape[1]
  w2 = _np32(rnn_state_dict['linear_mean2.weight'])
  dim = w2.shape[0]
  init_hidden = _np32(rnn_init_hidden).reshape(depth, hid)
  return {
      'observation_dim': dim,
      'rnn_hidden_size': hid,
      'rnn_depth': depth,
      'gru_weight_ih': [
          _np32(rnn_state_dict['gru.weight_ih_l{}'.format(l)])
          for l in range(depth)],
      'gru_weight_hh': [
          _np32(rnn_state_dict['gru.weight_hh_l{}'.format(l)])
          for l in range(depth)],
      'gru_bias_ih': [
          _np32(rnn_state_dict['gru.bias_ih_l{}'.format(l)])
          for l in range(depth)],
      'gru_bias_hh': [
          _np32(rnn_state_dict['gru.bias_hh_l{}'.format(l)])
          for l in range(depth)],
      'linear_mean1_weight': _np32(rnn_state_dict['linear_mean1.weight']),
      'linear_mean1_bias': _np32(rnn_state_dict['linear_mean1.bias']),
      'linear_mean2_weight': w2,
      'linear_mean2_bias': _np32(rnn_state_dict['linear_mean2.bias']),
      'rnn_init_hidden': init_hidden,
      'sigma2': _np32(sigma2).reshape(dim),
      'transition_bias': (
          None if transition_bias is None else float(transition_bias)),
      'transition_bias_denominator': float(transition_bias_denominator),
      'crp_alpha': float(crp_alpha),
  }


def state_dict_from_params(params):
  """Inverse of params_from_state: CoreRNN-keyed dict of numpy arrays."""
  out = {}
  for l in range(params['rnn_depth']):
    out['gru.weight_ih_l{}'.format(l)] = params['gru_weight_ih'][l]
    out['gru.weight_hh_l{}'.format(l)] = params['gru_weight_hh'][l]
    out['gru.bias_ih_l{}'.format(l)] = params['gru_bias_ih'][l]
    out['gru.bias_hh_l{}'.format(l)] = params['gru_bias_hh'][l]
  out['linear_mean1.weight'] = params['linear_mean1_weight']
  out['linear_mean1.bias'] = params['linear_mean1_bias']
  out['linear_mean2.weight'] = params['linear_mean2_weight']
  out['linear_mean2.bias'] = params['linear_mean2_bias']
  return out


def load_checkpoint(filepath):
  """Read a checkpoint written by the reference's UISRNN.save().

  Format (uisrnn/uisrnn.py:141-147): torch.save of a dict with keys
  rnn_state_dict, rnn_init_hidden (numpy), transition_bias,
  transition_bias_denominator, crp_alpha, sigma2 (numpy).  The numpy members
  need weights_only=False on torch >= 2.6 (the reference's own load() at
  uisrnn/uisrnn.py:155 fails there).
  """
  import torch  # pylint: disable=import-outside-toplevel
  var_dict = torch.load(filepath, map_location='cpu', weights_only=False)
  return params_from_state(
      var_dict['rnn_state_dict'], var_dict['rnn_init_hidden'],
      var_dict['sigma2'], var_dict['transition_bias'], var_dict['crp_alpha'],
      var_dict.get('transition_bias_denominator', 0.0))


def save_checkpoint(params, filepath):
  """Write the reference's checkpoint format (uisrnn/uisrnn.py:141-147)."""
  import torch  # pylint: disable=import-outside-toplevel
  state = {k: torch.from_numpy(np.array(v, dtype=np.float32))
           for k, v in state_dict_from_params(params).items()}
  depth, hid = params['rnn_depth'], params['rnn_hidden_size']
  torch.save({
      'rnn_state_dict': state,
      'rnn_init_hidden': np.asarray(
          params['rnn_init_hidden'], dtype=np.float32).reshape(depth, 1, hid),
      'transition_bias': params['transition_bias'],
      'transition_bias_denominator': params.get(
          'transition_bias_denominator', 0.0),
      'crp_alpha': params['crp_alpha'],
      'sigma2': np.asarray(params['sigma2'], dtype=np.float32)}, filepath)
