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

Checkpoints in the reference's torch.save format are read (read_torch_zip) and
written (write_torch_zip) without PyTorch.
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


# ---------------------------------------------------------------------------
# Reading the reference's checkpoint WITHOUT PyTorch.
#
# torch.save (zip format, torch >= 1.6) writes <name>/data.pkl -- a protocol-2
# pickle in which every tensor is torch._utils._rebuild_tensor_v2(storage,
# offset, size, stride, ...) and every storage a persistent id
# ('storage', torch.<T>Storage, key, device, numel) -- plus one raw
# little-endian file <name>/data/<key> per storage.  The numpy members of the
# reference's dict (rnn_init_hidden, sigma2) are ordinary numpy pickles.
# The unpickler below resolves exactly the globals such a file needs and
# nothing else, so loading a checkpoint cannot run arbitrary code.

_STORAGE_DTYPES = {
    'FloatStorage': np.float32, 'DoubleStorage': np.float64,
    'HalfStorage': np.float16, 'LongStorage': np.int64,
    'IntStorage': np.int32, 'ShortStorage': np.int16,
    'CharStorage': np.int8, 'ByteStorage': np.uint8, 'BoolStorage': np.bool_,
}


class _StorageType:
  def __init__(self, name):
    self.dtype = np.dtype(_STORAGE_DTYPES[name])


def _rebuild_tensor_v2(storage, storage_offset, size, stride, *unused):
  size, stride = tuple(size), tuple(stride)
  if not size:
    return storage[storage_offset].copy()
  view = np.lib.stride_tricks.as_strided(
      storage[storage_offset:], shape=size,
      strides=tuple(s * storage.itemsize for s in stride))
  return np.array(view, order='C')  # own, contiguous copy


def _rebuild_parameter(data, requires_grad, backward_hooks, *unused):
  return data


def read_torch_zip(filepath):
  """The object stored by torch.save(obj, filepath), tensors as numpy arrays.

  Pure Python (zipfile + a restricted pickle.Unpickler); raises ValueError for
  files that are not torch zip archives and pickle.UnpicklingError for pickles
  that reference anything outside a checkpoint's vocabulary.
  """
  import collections  # pylint: disable=import-outside-toplevel
  import pickle       # pylint: disable=import-outside-toplevel
  import zipfile      # pylint: disable=import-outside-toplevel
  if not zipfile.is_zipfile(filepath):
    raise ValueError('{} is not a torch zip checkpoint'.format(filepath))
  with zipfile.ZipFile(filepath) as archive:
    names = archive.namelist()
    pkl = [n for n in names if n.endswith('/data.pkl') or n == 'data.pkl']
    if len(pkl) != 1:
      raise ValueError('{}: no data.pkl inside'.format(filepath))
    prefix = pkl[0][:-len('data.pkl')]
    byteorder = 'little'
    if prefix + 'byteorder' in names:
      byteorder = archive.read(prefix + 'byteorder').decode().strip()
    storages = {}

    def persistent_load(pid):
      if not (isinstance(pid, tuple) and pid and pid[0] == 'storage'):
        raise pickle.UnpicklingError('unexpected persistent id')
      storage_type, key, numel = pid[1], pid[2], pid[4]
      if key not in storages:
        raw = archive.read('{}data/{}'.format(prefix, key))
        dtype = storage_type.dtype.newbyteorder(
            '<' if byteorder == 'little' else '>')
        arr = np.frombuffer(raw, dtype=dtype, count=int(numel))
        storages[key] = arr.astype(storage_type.dtype, copy=False)
      return storages[key]

    try:  # numpy >= 2 (numpy.core is a deprecated alias there and will go away)
      import numpy._core.multiarray as _np_multiarray  # pylint: disable=import-outside-toplevel
    except ImportError:
      import numpy.core.multiarray as _np_multiarray  # pylint: disable=import-outside-toplevel
    allowed = {
        ('collections', 'OrderedDict'): collections.OrderedDict,
        ('torch._utils', '_rebuild_tensor_v2'): _rebuild_tensor_v2,
        ('torch._utils', '_rebuild_parameter'): _rebuild_parameter,
        ('numpy', 'ndarray'): np.ndarray,
        ('numpy', 'dtype'): np.dtype,
        ('numpy.core.multiarray', '_reconstruct'): _np_multiarray._reconstruct,
        ('numpy._core.multiarray', '_reconstruct'): _np_multiarray._reconstruct,
        ('numpy.core.multiarray', 'scalar'): _np_multiarray.scalar,
        ('numpy._core.multiarray', 'scalar'): _np_multiarray.scalar,
        ('_codecs', 'encode'): __import__('_codecs').encode,
    }

    class _Unpickler(pickle.Unpickler):
      def find_class(self, module, name):
        if module == 'torch' and name in _STORAGE_DTYPES:
          return _StorageType(name)
        try:
          return allowed[(module, name)]
        except KeyError:
          raise pickle.UnpicklingError(
              'checkpoint refers to {}.{}, which a uis-rnn checkpoint does '
              'not need'.format(module, name)) from None

    import io  # pylint: disable=import-outside-toplevel
    unpickler = _Unpickler(io.BytesIO(archive.read(pkl[0])))
    unpickler.persistent_load = persistent_load
    return unpickler.load()


def load_checkpoint(filepath):
  """Read a checkpoint written by the reference's UISRNN.save().

  Format (uisrnn/uisrnn.py:141-147): torch.save of a dict with keys
  rnn_state_dict, rnn_init_hidden (numpy), transition_bias,
  transition_bias_denominator, crp_alpha, sigma2 (numpy).  Read without
  PyTorch (read_torch_zip); only the pre-1.6 non-zip format still goes through
  torch.load (weights_only=False: the numpy members make the reference's own
  load() at uisrnn/uisrnn.py:155 fail on torch >= 2.6).
  """
  try:
    var_dict = read_torch_zip(filepath)
  except ValueError:
    import torch  # pylint: disable=import-outside-toplevel
    var_dict = torch.load(filepath, map_location='cpu', weights_only=False)
  return params_from_state(
      var_dict['rnn_state_dict'], var_dict['rnn_init_hidden'],
      var_dict['sigma2'], var_dict['transition_bias'], var_dict['crp_alpha'],
      var_dict.get('transition_bias_denominator', 0.0))


# ---------------------------------------------------------------------------
# Writing that format WITHOUT PyTorch.  The pickle is produced by the pure-Python
# pickler with two stand-ins: _TorchGlobal (written as a bare GLOBAL opcode --
# the module is never imported here) and _Tensor (reduces to
# torch._utils._rebuild_tensor_v2 over a storage persistent id, exactly what
# torch.save emits for a contiguous CPU float tensor).


class _TorchGlobal:
  def __init__(self, module, name):
    self.module, self.name = module, name

  def __call__(self, *args):  # (the pickler only checks that a reduce function is callable)
    raise TypeError('{}.{} is a name in a pickle, not a function'.format(self.module, self.name))


class _Storage:
  def __init__(self, key, array):
    self.key, self.array = key, array


class _Tensor:
  def __init__(self, storage, shape):
    self.storage, self.shape = storage, tuple(int(n) for n in shape)

  def __reduce__(self):
    stride, acc = [], 1
    for n in reversed(self.shape):
      stride.append(acc)
      acc *= n
    import collections  # pylint: disable=import-outside-toplevel
    return (_TorchGlobal('torch._utils', '_rebuild_tensor_v2'),
            (self.storage, 0, self.shape, tuple(reversed(stride)), False,
             collections.OrderedDict()))


def write_torch_zip(obj, filepath, tensor_keys=('rnn_state_dict',)):
  """torch.save(obj, filepath) for a dict whose `tensor_keys` members are dicts
  of float32 numpy arrays (written as torch tensors); everything else --
  numpy arrays, floats, None -- is pickled as it is.  No PyTorch involved; the
  result loads with torch.load(..., weights_only=False) and read_torch_zip."""
  import collections  # pylint: disable=import-outside-toplevel
  import io           # pylint: disable=import-outside-toplevel
  import pickle       # pylint: disable=import-outside-toplevel
  import zipfile      # pylint: disable=import-outside-toplevel
  storages = []
  out = {}
  for key, val in obj.items():
    if key in tensor_keys:
      tensors = collections.OrderedDict()
      for name, arr in val.items():
        arr = np.ascontiguousarray(arr, dtype=np.float32)
        storage = _Storage(str(len(storages)), arr)
        storages.append(storage)
        tensors[name] = _Tensor(storage, arr.shape)
      out[key] = tensors
    else:
      out[key] = val

  class _Pickler(pickle._Pickler):  # pylint: disable=protected-access
    dispatch = dict(pickle._Pickler.dispatch)  # pylint: disable=protected-access

    def persistent_id(self, item):
      if isinstance(item, _Storage):
        return ('storage', _TorchGlobal('torch', 'FloatStorage'), item.key, 'cpu',
                int(item.array.size))
      return None

    def _save_torch_global(self, item):
      self.write(pickle.GLOBAL + item.module.encode() + b'\n' + item.name.encode() + b'\n')
      self.memoize(item)

    dispatch[_TorchGlobal] = _save_torch_global

  buf = io.BytesIO()
  _Pickler(buf, protocol=2).dump(out)
  name = 'archive'
  with zipfile.ZipFile(filepath, 'w', compression=zipfile.ZIP_STORED) as archive:
    archive.writestr(name + '/data.pkl', buf.getvalue())
    archive.writestr(name + '/byteorder', 'little')
    for storage in storages:
      archive.writestr('{}/data/{}'.format(name, storage.key),
                       storage.array.astype('<f4', copy=False).tobytes())
    archive.writestr(name + '/version', '3\n')


def save_checkpoint(params, filepath):
  """Write the reference's checkpoint format (uisrnn/uisrnn.py:141-147), without
  PyTorch: a file the reference's own UISRNN.load() reads (where its torch.load
  accepts numpy members, i.e. torch < 2.6 or weights_only=False)."""
  depth, hid = params['rnn_depth'], params['rnn_hidden_size']
  write_torch_zip({
      'rnn_state_dict': state_dict_from_params(params),
      'rnn_init_hidden': np.asarray(
          params['rnn_init_hidden'], dtype=np.float32).reshape(depth, 1, hid),
      'transition_bias': params['transition_bias'],
      'transition_bias_denominator': params.get(
          'transition_bias_denominator', 0.0),
      'crp_alpha': params['crp_alpha'],
      'sigma2': np.asarray(params['sigma2'], dtype=np.float32)}, filepath)
