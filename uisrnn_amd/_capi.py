"""ctypes binding of the C ABI declared in include/uisrnn_hip.h.

The reference has no FFI; its boundary is the Python method set on
``uisrnn.UISRNN`` (uisrnn/__init__.py:26-30, uisrnn/uisrnn.py:479-623).  This
module is the thin layer between that Python surface (uisrnn_amd/uisrnn.py)
and ``libuisrnn_hip.so``.  There is deliberately NO fallback: if the HIP
library is missing or no gfx950 device is usable, the calls raise.
"""

import ctypes
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libuisrnn_hip.so')

UIS_OK = 0
UIS_ABI_VERSION = 6   # include/uisrnn_hip.h
UIS_ERR_INVALID_ARG = -1
UIS_ERR_DIM_MISMATCH = -2
UIS_ERR_NO_DEVICE = -3
UIS_ERR_HIP = -4
UIS_ERR_OOM = -5
UIS_ERR_CLUSTER_CAP = -6
UIS_ERR_UNSUPPORTED = -7

UIS_FLAG_NO_DEDUP = 0x1
UIS_FLAG_GRAPH = 0x2
UIS_FLAG_PROFILE = 0x4
UIS_FLAG_GENERIC_SELECT = 0x8
UIS_FLAG_RESIDENT = 0x40
UIS_FLAG_STEPWISE = 0x80
UIS_FLAG_TEST_MISPLACED = 0x100
UIS_FLAG_TEST_STALL = 0x4000
UIS_FLAG_SMALL_TILES = 0x200
UIS_FLAG_PERSISTENT = 0x400
UIS_FLAG_OWNER_SELECT = 0x800
UIS_FLAG_REPLICATED_SELECT = 0x1000
UIS_FLAG_DEBUG_SCORES = 0x2000
UIS_FLAG_CLUSTER_BARRIERS = 0x8000
UIS_FLAG_COHORTS = 0x10000
UIS_FLAG_AGENT_FLAGS = 0x20000
UIS_BUILD_COHORTS = 0x1

UIS_N_KERNELS = 8
KERNEL_NAMES = ('input_proj', 'select', 'gru', 'head1', 'head2', 'backtrace',
                'upper_in', 'expand')

# uis_stats.decode_kernel (UIS_DK_* | UIS_DF_* << 8): the kernel family that ran the decode steps
DECODE_KERNELS = {0: 'none', 1: 'stepwise', 2: 'k_decode_rs', 3: 'k_decode_resident', 4: 'k_decode_big',
                  5: 'k_decode_big<WS>', 6: 'k_decode_small',
                  7: 'k_decode_big<WIN>', 8: 'k_decode_deep', 9: 'k_decode_coh'}
DENSE_FAMILIES = {0: '', 1: 'k_dense', 2: 'k_big', 3: 'k_wt'}


RS_VARIANTS = {0: '', 1: '', 2: '', 3: '<2 per wave>', 4: '<wide>', 5: '<2 per wave>', 6: '<wide>'}


def decode_kernel_name(code):
  name = DECODE_KERNELS.get(code & 0xff, 'unknown')
  fam = DENSE_FAMILIES.get((code >> 8) & 0xff, '')
  return name + RS_VARIANTS.get((code >> 16) & 0xff, '') + (':' + fam if fam else '')


_fp = ctypes.POINTER(ctypes.c_float)
_fpp = ctypes.POINTER(_fp)


class ModelDesc(ctypes.Structure):
  """struct uis_model_desc (include/uisrnn_hip.h)."""
  _fields_ = [
      ('observation_dim', ctypes.c_int32),
      ('rnn_hidden_size', ctypes.c_int32),
      ('rnn_depth', ctypes.c_int32),
      ('reserved0', ctypes.c_int32),
      ('gru_weight_ih', _fpp),
      ('gru_weight_hh', _fpp),
      ('gru_bias_ih', _fpp),
      ('gru_bias_hh', _fpp),
      ('linear_mean1_weight', _fp),
      ('linear_mean1_bias', _fp),
      ('linear_mean2_weight', _fp),
      ('linear_mean2_bias', _fp),
      ('rnn_init_hidden', _fp),
      ('sigma2', _fp),
      ('transition_bias', ctypes.c_double),
      ('crp_alpha', ctypes.c_double),
  ]


class DecodeOpts(ctypes.Structure):
  """struct uis_decode_opts (include/uisrnn_hip.h)."""
  _fields_ = [
      ('beam_size', ctypes.c_int32),
      ('look_ahead', ctypes.c_int32),
      ('test_iteration', ctypes.c_int32),
      ('max_clusters', ctypes.c_int32),
      ('flags', ctypes.c_uint32),
      ('n_streams', ctypes.c_int32),
      ('level_cap', ctypes.c_int32),
      ('reserved', ctypes.c_int32 * 1),
  ]


class Stats(ctypes.Structure):
  """struct uis_stats (include/uisrnn_hip.h)."""
  _fields_ = [
      ('n_steps', ctypes.c_int32),
      ('max_clusters_seen', ctypes.c_int32),
      ('rnn_rows', ctypes.c_int64),
      ('rnn_rows_nodedup', ctypes.c_int64),
      ('candidates', ctypes.c_int64),
      ('decode_ms', ctypes.c_double),
      ('kernel_ms', ctypes.c_double * UIS_N_KERNELS),
      ('kernel_launches', ctypes.c_int64 * UIS_N_KERNELS),
      ('n_overflow', ctypes.c_int32),
      ('n_streams', ctypes.c_int32),
      ('decode_kernel', ctypes.c_int32),
      ('decode_launches', ctypes.c_int32),
  ]

  def as_dict(self):
    return {
        'n_steps': self.n_steps,
        'max_clusters_seen': self.max_clusters_seen,
        'rnn_rows': self.rnn_rows,
        'rnn_rows_nodedup': self.rnn_rows_nodedup,
        'candidates': self.candidates,
        'decode_ms': self.decode_ms,
        'decode_launches': self.decode_launches,
        'kernel_ms': {n: self.kernel_ms[i] for i, n in enumerate(KERNEL_NAMES)},
        'kernel_launches': {
            n: self.kernel_launches[i] for i, n in enumerate(KERNEL_NAMES)},
        'n_overflow': self.n_overflow,
        'n_streams': self.n_streams,
        'decode_kernel': decode_kernel_name(self.decode_kernel),
    }


def _f32(a):
  return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def _ptr(a):
  return a.ctypes.data_as(_fp)


def make_desc(params):
  """Build a ModelDesc from a parameter dict (see uisrnn_amd/weights.py).

  Returns (desc, keepalive): keepalive holds the numpy buffers the struct
  points into and must outlive every use of desc.
  """
  depth = int(params['rnn_depth'])
  keep = []

  def arr(a, shape):
    a = _f32(a)
    if tuple(a.shape) != tuple(shape):
      raise ValueError('parameter has shape {} but {} is required'.format(
          a.shape, shape))
    keep.append(a)
    return a

  dim = int(params['observation_dim'])
  hid = int(params['rnn_hidden_size'])

  def ptr_array(key, shapes):
    arrs = [arr(params[key][l], shapes[l]) for l in range(depth)]
    pa = (_fp * depth)(*[_ptr(a) for a in arrs])
    keep.append(pa)
    return ctypes.cast(pa, _fpp)

  in_dims = [dim] + [hid] * (depth - 1)
  desc = ModelDesc()
  desc.observation_dim = dim
  desc.rnn_hidden_size = hid
  desc.rnn_depth = depth
  desc.gru_weight_ih = ptr_array(
      'gru_weight_ih', [(3 * hid, in_dims[l]) for l in range(depth)])
  desc.gru_weight_hh = ptr_array(
      'gru_weight_hh', [(3 * hid, hid)] * depth)
  desc.gru_bias_ih = ptr_array('gru_bias_ih', [(3 * hid,)] * depth)
  desc.gru_bias_hh = ptr_array('gru_bias_hh', [(3 * hid,)] * depth)
  desc.linear_mean1_weight = _ptr(arr(params['linear_mean1_weight'], (hid, hid)))
  desc.linear_mean1_bias = _ptr(arr(params['linear_mean1_bias'], (hid,)))
  desc.linear_mean2_weight = _ptr(arr(params['linear_mean2_weight'], (dim, hid)))
  desc.linear_mean2_bias = _ptr(arr(params['linear_mean2_bias'], (dim,)))
  desc.rnn_init_hidden = _ptr(arr(params['rnn_init_hidden'], (depth, hid)))
  desc.sigma2 = _ptr(arr(params['sigma2'], (dim,)))
  desc.transition_bias = float(params['transition_bias'])
  desc.crp_alpha = float(params['crp_alpha'])
  return desc, keep


def make_opts(beam_size, look_ahead, test_iteration, max_clusters=0, flags=0,
              n_streams=0, level_cap=0):
  opts = DecodeOpts()
  opts.level_cap = int(level_cap)
  opts.n_streams = int(n_streams)
  opts.beam_size = int(beam_size)
  opts.look_ahead = int(look_ahead)
  opts.test_iteration = int(test_iteration)
  opts.max_clusters = int(max_clusters)
  opts.flags = int(flags)
  return opts


class HipLibraryError(RuntimeError):
  """The HIP decoder library is missing, failed to load or reported an error."""
  status = None   # the uis_status of a failed call, where there was one


_lib = None
_hip_runtime = None   # what share_hip_runtime_with_torch() did, for whoever asks (tests, bench.py)


def share_hip_runtime_with_torch():
  """ONE HIP runtime per process, whichever of {this library, PyTorch} is loaded first (round 6).

  The PyTorch-ROCm wheel bundles its own libamdhip64.so (SONAME libamdhip64.so.7, the system ROCm's too).  With
  torch imported FIRST the dynamic linker resolves this library's `NEEDED libamdhip64.so.7` to torch's copy, already
  loaded under that SONAME: one runtime, everything works.  With this library first the system runtime is loaded, a
  later `import torch` brings a SECOND runtime (its NEEDED name is `libamdhip64.so`: no SONAME match) and
  torch.cuda.init() reports "No HIP GPUs are available" (measured, round 4).  So: where a torch installation is found
  and not yet imported, its libamdhip64.so is dlopen()ed (RTLD_GLOBAL, by path, WITHOUT importing torch) before this
  library -- both load orders then end in the first situation.  UIS_HIP_RUNTIME=system keeps the system runtime (for
  processes that never import torch)."""
  global _hip_runtime
  if _hip_runtime is not None:
    return _hip_runtime
  if os.environ.get('UIS_HIP_RUNTIME', 'auto') == 'system':
    _hip_runtime = 'system (UIS_HIP_RUNTIME)'
  elif 'torch' in sys.modules:
    _hip_runtime = 'torch (imported before this library)'
  else:
    _hip_runtime = 'system (no torch installation found)'
    try:
      import importlib.util  # pylint: disable=import-outside-toplevel
      spec = importlib.util.find_spec('torch')
      bundled = os.path.join(os.path.dirname(spec.origin), 'lib', 'libamdhip64.so') if spec and spec.origin else None
      if bundled and os.path.exists(bundled):
        ctypes.CDLL(bundled, mode=ctypes.RTLD_GLOBAL)
        _hip_runtime = 'torch (preloaded: ' + bundled + ')'
    except (ImportError, OSError, ValueError) as e:
      _hip_runtime = 'system (torch\'s runtime could not be preloaded: {})'.format(e)
  return _hip_runtime


def load_library(path=None):
  """dlopen libuisrnn_hip.so and declare every entry point of the header."""
  global _lib
  if _lib is not None and path is None:
    return _lib
  path = path or os.environ.get('UIS_LIB_PATH') or LIB_PATH
  if not os.path.exists(path):
    raise HipLibraryError(
        '{} not found: build it with `python -m uisrnn_amd.build` (hipcc, '
        'gfx950). There is no CPU fallback for the decode path.'.format(path))
  share_hip_runtime_with_torch()
  lib = ctypes.CDLL(path)
  lib.uis_abi_version.restype = ctypes.c_int32
  lib.uis_abi_version.argtypes = []
  if lib.uis_abi_version() != UIS_ABI_VERSION:
    # a stale in-tree build: its structs would not match the ctypes mirrors below
    raise HipLibraryError(
        '{} was built for ABI {} but this package expects {}: rebuild with '
        '`python -m uisrnn_amd.build --force`'.format(path, lib.uis_abi_version(), UIS_ABI_VERSION))
  i32 = ctypes.c_int32
  i32p = ctypes.POINTER(ctypes.c_int32)
  i64p = ctypes.POINTER(ctypes.c_int64)
  lib.uis_abi_version.restype = i32
  lib.uis_abi_version.argtypes = []
  lib.uis_numerics_version.restype = i32
  lib.uis_numerics_version.argtypes = []
  lib.uis_build_flags.restype = ctypes.c_uint32
  lib.uis_build_flags.argtypes = []
  lib.uis_device_count.restype = i32
  lib.uis_device_count.argtypes = []
  lib.uis_create.restype = i32
  lib.uis_create.argtypes = [
      ctypes.POINTER(ModelDesc), i32, ctypes.POINTER(ctypes.c_void_p)]
  lib.uis_destroy.restype = None
  lib.uis_destroy.argtypes = [ctypes.c_void_p]
  lib.uis_decode.restype = i32
  lib.uis_decode.argtypes = [
      ctypes.c_void_p, _fp, i64p, i32, ctypes.POINTER(DecodeOpts), i32p, _fp,
      ctypes.POINTER(Stats)]
  lib.uis_decode_device.restype = i32
  lib.uis_decode_device.argtypes = [
      ctypes.c_void_p, ctypes.c_void_p, i64p, i32, ctypes.POINTER(DecodeOpts),
      ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(Stats)]
  lib.uis_decode_f64.restype = i32
  lib.uis_decode_f64.argtypes = [
      ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), i64p, i32, ctypes.POINTER(DecodeOpts),
      i32p, _fp, ctypes.POINTER(Stats)]
  lib.uis_last_decode_info.restype = i32
  lib.uis_last_decode_info.argtypes = [ctypes.c_void_p, i32p, _fp]
  lib.uis_last_decode_shape.restype = i32
  lib.uis_last_decode_shape.argtypes = [ctypes.c_void_p, i32p, i32p]
  lib.uis_debug_scores.restype = i32
  lib.uis_debug_scores.argtypes = [ctypes.c_void_p, _fp, ctypes.c_int64]
  lib.uis_model_constants.restype = i32
  lib.uis_model_constants.argtypes = [ctypes.c_void_p, _fp, _fp]
  lib.uis_rnn_step.restype = i32
  lib.uis_rnn_step.argtypes = [ctypes.c_void_p, _fp, _fp, _fp, _fp]
  lib.uis_stream_begin.restype = i32
  lib.uis_stream_begin.argtypes = [ctypes.c_void_p, i32, ctypes.POINTER(DecodeOpts), ctypes.c_int64]
  lib.uis_stream_push.restype = i32
  lib.uis_stream_push.argtypes = [ctypes.c_void_p, _fp, i32p]
  lib.uis_stream_labels.restype = i32
  lib.uis_stream_labels.argtypes = [ctypes.c_void_p, i32p, _fp, i32p]
  lib.uis_stream_end.restype = i32
  lib.uis_stream_end.argtypes = [ctypes.c_void_p]
  lib.uis_eval_accuracy.restype = i32
  lib.uis_eval_accuracy.argtypes = [ctypes.c_void_p, i32p, i32p, i64p, i32, i64p]
  lib.uis_eval_accuracy_device.restype = i32
  lib.uis_eval_accuracy_device.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, i64p, i32, i64p]
  lib.uis_eval_last_decode.restype = i32
  lib.uis_eval_last_decode.argtypes = [ctypes.c_void_p, i32p, i32, i64p]
  lib.uis_host_alloc.restype = i32
  lib.uis_host_alloc.argtypes = [ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p)]
  lib.uis_host_free.restype = None
  lib.uis_host_free.argtypes = [ctypes.c_void_p]
  lib.uis_last_error.restype = ctypes.c_char_p
  lib.uis_last_error.argtypes = []
  _lib = lib
  return lib


EXPORTED_SYMBOLS = (
    'uis_abi_version', 'uis_numerics_version', 'uis_build_flags', 'uis_device_count', 'uis_create', 'uis_destroy',
    'uis_decode', 'uis_decode_f64', 'uis_decode_device', 'uis_last_decode_info', 'uis_last_decode_shape',
    'uis_debug_scores',
    'uis_model_constants', 'uis_rnn_step', 'uis_stream_begin', 'uis_stream_push',
    'uis_stream_labels', 'uis_stream_end', 'uis_eval_accuracy', 'uis_eval_accuracy_device',
    'uis_eval_last_decode', 'uis_host_alloc', 'uis_host_free', 'uis_last_error')


def last_error(lib):
  msg = lib.uis_last_error()
  return msg.decode('utf-8', 'replace') if msg else ''


class Decoder:
  """Owns one uis_handle (one HIP device)."""

  def __init__(self, params, device=0):
    self._lib = load_library()
    self.params = params
    self.observation_dim = int(params['observation_dim'])
    desc, keep = make_desc(params)
    handle = ctypes.c_void_p()
    rc = self._lib.uis_create(ctypes.byref(desc), int(device),
                              ctypes.byref(handle))
    del keep
    if rc != UIS_OK:
      raise HipLibraryError('uis_create failed ({}): {}'.format(
          rc, last_error(self._lib)))
    self._handle = handle
    self.device = int(device)

  def close(self):
    if getattr(self, '_handle', None):
      self._lib.uis_destroy(self._handle)
      self._handle = None

  def __del__(self):
    try:
      self.close()
    except Exception:  # pylint: disable=broad-except
      pass

  def constants(self):
    """(m0 [D], h1 [depth, H]) as computed on the device at create time."""
    m0 = np.empty(self.observation_dim, dtype=np.float32)
    h1 = np.empty((int(self.params['rnn_depth']),
                   int(self.params['rnn_hidden_size'])), dtype=np.float32)
    self._check(self._lib.uis_model_constants(
        self._handle, m0.ctypes.data_as(_fp), h1.ctypes.data_as(_fp)),
                'uis_model_constants')
    return m0, h1

  def rnn_step(self, x, h_in):
    """CoreRNN.forward of one row on the device: x [D], h_in [depth, H]."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    h_in = np.ascontiguousarray(h_in, dtype=np.float32)
    mean = np.empty(self.observation_dim, dtype=np.float32)
    h_out = np.empty_like(h_in)
    self._check(self._lib.uis_rnn_step(
        self._handle, x.ctypes.data_as(_fp), h_in.ctypes.data_as(_fp),
        mean.ctypes.data_as(_fp), h_out.ctypes.data_as(_fp)), 'uis_rnn_step')
    return mean, h_out

  def _check(self, rc, what):
    if rc == UIS_OK or rc == UIS_ERR_CLUSTER_CAP:
      return rc
    msg = last_error(self._lib)
    if rc == UIS_ERR_DIM_MISMATCH:
      raise ValueError(msg)
    err = HipLibraryError('{} failed ({}): {}'.format(what, rc, msg))
    err.status = rc
    raise err

  def decode(self, frames, offsets, beam_size, look_ahead, test_iteration,
             max_clusters=0, flags=0, want_beam_scores=False, n_streams=0, level_cap=0):
    """Decode packed host utterances.

    Args:
      frames: float32 [sum N, D] C-contiguous.
      offsets: int64 [U + 1].
    Returns:
      dict with labels (int32 [sum N]), scores (float32 [U]), overflow
      (int32 [U]), stats (dict) and optionally beam_scores [U, beam_size].
    """
    frames = np.ascontiguousarray(frames, dtype=np.float32)
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    n_utt = offsets.shape[0] - 1
    total = int(offsets[-1])
    if frames.ndim != 2 or frames.shape[0] != total:
      raise ValueError('frames must be [offsets[-1], D]')
    if total and frames.shape[1] != self.observation_dim:
      raise ValueError('frames do not match observation_dim')
    labels = np.empty(total, dtype=np.int32)
    scores = np.empty(n_utt, dtype=np.float32)
    opts = make_opts(beam_size, look_ahead, test_iteration, max_clusters, flags,
                     n_streams, level_cap)
    stats = Stats()
    rc = self._lib.uis_decode(
        self._handle, frames.ctypes.data_as(_fp),
        offsets.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), n_utt,
        ctypes.byref(opts), labels.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
        scores.ctypes.data_as(_fp), ctypes.byref(stats))
    rc = self._check(rc, 'uis_decode')
    out = {'labels': labels, 'scores': scores, 'stats': stats.as_dict(),
           'status': rc}
    overflow = np.zeros(n_utt, dtype=np.int32)
    beam_scores = (np.empty((n_utt, int(beam_size)), dtype=np.float32)
                   if want_beam_scores else None)
    rc2 = self._lib.uis_last_decode_info(
        self._handle, overflow.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
        beam_scores.ctypes.data_as(_fp) if want_beam_scores else None)
    self._check(rc2, 'uis_last_decode_info')
    out['overflow'] = overflow
    if want_beam_scores:
      out['beam_scores'] = beam_scores
    return out

  def decode_f64(self, sequences, beam_size, look_ahead, test_iteration,
                 max_clusters=0, flags=0, want_beam_scores=False, n_streams=0, level_cap=0):
    """Decode a list of [N_u, D] float64 arrays as predict() receives them (uis_decode_f64:
    the library casts to float32 on its own threads while earlier chunks travel to the device).

    Returns the dict of decode(); labels are packed in utterance order.
    """
    seqs = [np.ascontiguousarray(s, dtype=np.float64) for s in sequences]  # (no copy when already so)
    n_utt = len(seqs)
    for s in seqs:
      if s.ndim != 2 or (s.shape[0] and s.shape[1] != self.observation_dim):
        raise ValueError('frames do not match observation_dim')
    lens = np.array([s.shape[0] for s in seqs], dtype=np.int64)
    total = int(lens.sum())
    ptrs = (ctypes.c_void_p * max(n_utt, 1))(*[s.ctypes.data if s.shape[0] else None for s in seqs])
    labels = np.empty(total, dtype=np.int32)
    scores = np.empty(n_utt, dtype=np.float32)
    opts = make_opts(beam_size, look_ahead, test_iteration, max_clusters, flags, n_streams, level_cap)
    stats = Stats()
    rc = self._lib.uis_decode_f64(
        self._handle, ptrs, lens.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), n_utt,
        ctypes.byref(opts), labels.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
        scores.ctypes.data_as(_fp), ctypes.byref(stats))
    rc = self._check(rc, 'uis_decode_f64')
    out = {'labels': labels, 'scores': scores, 'stats': stats.as_dict(), 'status': rc}
    overflow = np.zeros(n_utt, dtype=np.int32)
    beam_scores = (np.empty((n_utt, int(beam_size)), dtype=np.float32)
                   if want_beam_scores else None)
    rc2 = self._lib.uis_last_decode_info(
        self._handle, overflow.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
        beam_scores.ctypes.data_as(_fp) if want_beam_scores else None)
    self._check(rc2, 'uis_last_decode_info')
    out['overflow'] = overflow
    if want_beam_scores:
      out['beam_scores'] = beam_scores
    return out

  def last_overflow(self, n_utt=None):
    """Per-utterance flags of the last decode: bit 0 = a survivor hit the cluster cap, bit 1 = an
    intermediate level of a look-ahead window was full (uis_last_decode_info).

    The buffer is sized from the library (uis_last_decode_shape), never from the caller's idea of
    the batch: a decode that was refused before it started leaves zero utterances behind, and the
    result is then an empty array.  `n_utt`, if given, only pads / cuts the returned array."""
    n_lib, beam = ctypes.c_int32(0), ctypes.c_int32(0)
    self._check(self._lib.uis_last_decode_shape(self._handle, ctypes.byref(n_lib), ctypes.byref(beam)),
                'uis_last_decode_shape')
    overflow = np.zeros(max(int(n_lib.value), 1), dtype=np.int32)
    self._check(self._lib.uis_last_decode_info(
        self._handle, overflow.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), None), 'uis_last_decode_info')
    overflow = overflow[:int(n_lib.value)]
    if n_utt is not None and int(n_utt) != overflow.shape[0]:
      padded = np.zeros(int(n_utt), dtype=np.int32)
      padded[:min(int(n_utt), overflow.shape[0])] = overflow[:int(n_utt)]
      return padded
    return overflow

  def debug_scores(self, n_windows, n_utt, beam_size, max_clusters, look_ahead=1):
    """The candidate scores of the last decode (flags included UIS_FLAG_DEBUG_SCORES):
    float32 [n_windows, n_utt, beam_size] + [max_clusters + 1] * look_ahead, +inf where
    _calculate_score's padded array (uisrnn/uisrnn.py:534-545) holds +inf.  n_windows =
    ceil(test_iteration * longest utterance / look_ahead)."""
    out = np.empty([int(n_windows), int(n_utt), int(beam_size)] + [int(max_clusters) + 1] * int(look_ahead),
                   dtype=np.float32)
    self._check(self._lib.uis_debug_scores(self._handle, out.ctypes.data_as(_fp), out.size), 'uis_debug_scores')
    return out

  # ---- online decoding (uis_stream_*)
  def stream_begin(self, n_utt, beam_size, max_frames, max_clusters=0, flags=0):
    """Open a streaming session for n_utt utterances (test_iteration 1, look_ahead 1)."""
    opts = make_opts(beam_size, 1, 1, max_clusters, flags, 0)
    rc = self._lib.uis_stream_begin(self._handle, int(n_utt), ctypes.byref(opts), int(max_frames))
    self._check(rc, 'uis_stream_begin')
    self._stream_n = int(n_utt)
    self._stream_have = np.zeros(int(n_utt), dtype=np.int64)
    self._stream_counts = np.zeros(int(n_utt), dtype=np.int32)   # (reused by every push: its pointer is bound once)
    self._stream_counts_ptr = self._stream_counts.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))

  def stream_push(self, chunks):
    """chunks: one [n_u, D] array (or None / empty) per utterance: its new frames -- or ONE
    [n_utt, n, D] array when every utterance has the same number of new frames (one cast, no
    per-utterance Python work: the low-latency way to feed a live session)."""
    if len(chunks) != self._stream_n:
      raise ValueError('one chunk (or None) per utterance')
    dim = self.observation_dim
    if isinstance(chunks, np.ndarray) and chunks.ndim == 3:
      if chunks.shape[2] != dim:
        raise ValueError('chunk does not match observation_dim')
      frames = np.ascontiguousarray(chunks, dtype=np.float32)
      counts = np.full(self._stream_n, chunks.shape[1], dtype=np.int32)
      rc = self._lib.uis_stream_push(
          self._handle, frames.ctypes.data_as(_fp) if frames.size else None,
          counts.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
      self._check(rc, 'uis_stream_push')
      self._stream_have += counts
      return
    # (round 6) the list form without per-chunk numpy work: ONE concatenate over the caller's arrays (it checks that
    # they are 2-dimensional and of one width), one cast, the counts written into a buffer whose ctypes pointer was
    # made when the session opened -- a one-frame push of 64 utterances spent 56 us here, against 44 us in the library
    counts = self._stream_counts
    try:
      counts[:] = [c.shape[0] for c in chunks]   # (a None entry has no shape: the careful path below)
      frames64 = np.concatenate(chunks)
      if frames64.ndim != 2 or frames64.dtype != np.float64 or len({c.dtype for c in chunks}) != 1:
        raise TypeError
    except (TypeError, ValueError, AttributeError, IndexError):
      # None / empty / non-array / non-float64 entries: the careful path, chunk by chunk
      parts = []
      counts[:] = 0
      for u, chunk in enumerate(chunks):
        if chunk is None or len(chunk) == 0:
          continue
        arr = np.ascontiguousarray(chunk, dtype=np.float32)
        if arr.ndim != 2 or arr.shape[1] != dim:
          raise ValueError('chunk does not match observation_dim')
        parts.append(arr)
        counts[u] = arr.shape[0]
      frames64 = np.concatenate(parts) if parts else np.zeros((0, dim), dtype=np.float32)
    if frames64.shape[0] and frames64.shape[1] != dim:
      raise ValueError('chunk does not match observation_dim')
    frames = frames64.astype(np.float32, copy=False)
    rc = self._lib.uis_stream_push(self._handle, frames.ctypes.data_as(_fp) if len(frames) else None, self._stream_counts_ptr)
    self._check(rc, 'uis_stream_push')
    self._stream_have += counts

  def stream_labels(self):
    """Best-hypothesis labels of everything received so far: (list of int32 arrays, scores, overflow, status)."""
    total = int(self._stream_have.sum())
    labels = np.empty(max(total, 1), dtype=np.int32)
    scores = np.empty(self._stream_n, dtype=np.float32)
    overflow = np.zeros(self._stream_n, dtype=np.int32)
    rc = self._lib.uis_stream_labels(
        self._handle, labels.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), scores.ctypes.data_as(_fp),
        overflow.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
    rc = self._check(rc, 'uis_stream_labels')
    bounds = np.concatenate([[0], np.cumsum(self._stream_have)]).astype(np.int64)
    per_utt = [labels[bounds[u]:bounds[u + 1]].copy() for u in range(self._stream_n)]
    return per_utt, scores, overflow, rc

  def stream_end(self):
    self._check(self._lib.uis_stream_end(self._handle), 'uis_stream_end')

  def decode_host(self, frames_ptr, offsets, beam_size, look_ahead, test_iteration,
                  labels_ptr, scores_ptr, max_clusters=0, flags=0):
    """uis_decode on raw HOST pointers (e.g. pinned buffers from uis_host_alloc or torch)."""
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    n_utt = offsets.shape[0] - 1
    opts = make_opts(beam_size, look_ahead, test_iteration, max_clusters, flags, 0)
    stats = Stats()
    rc = self._lib.uis_decode(
        self._handle, ctypes.cast(ctypes.c_void_p(int(frames_ptr)), _fp),
        offsets.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), n_utt,
        ctypes.byref(opts),
        ctypes.cast(ctypes.c_void_p(int(labels_ptr)), ctypes.POINTER(ctypes.c_int32)),
        ctypes.cast(ctypes.c_void_p(int(scores_ptr)), _fp) if scores_ptr else None,
        ctypes.byref(stats))
    rc = self._check(rc, 'uis_decode')
    return {'stats': stats.as_dict(), 'status': rc}

  # ---- evaluation on the device (uis_eval_*)
  @staticmethod
  def _eval_args(offsets):
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    n_utt = offsets.shape[0] - 1
    matched = np.zeros(max(n_utt, 1), dtype=np.int64)
    return offsets, n_utt, matched

  def eval_matched(self, labels_a, labels_b, offsets):
    """Matched positions per utterance under the best label mapping (host int32 arrays in)."""
    labels_a = np.ascontiguousarray(labels_a, dtype=np.int32)
    labels_b = np.ascontiguousarray(labels_b, dtype=np.int32)
    offsets, n_utt, matched = self._eval_args(offsets)
    if labels_a.shape != labels_b.shape or labels_a.shape[0] != int(offsets[-1]):
      raise ValueError('both label arrays must hold offsets[-1] entries')
    i32p = ctypes.POINTER(ctypes.c_int32)
    self._check(self._lib.uis_eval_accuracy(
        self._handle, labels_a.ctypes.data_as(i32p), labels_b.ctypes.data_as(i32p),
        offsets.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), n_utt,
        matched.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))), 'uis_eval_accuracy')
    return matched[:n_utt]

  def eval_matched_device(self, d_labels_a_ptr, d_labels_b_ptr, offsets):
    """Same with both label sequences resident in HBM (raw device pointers)."""
    offsets, n_utt, matched = self._eval_args(offsets)
    self._check(self._lib.uis_eval_accuracy_device(
        self._handle, ctypes.c_void_p(int(d_labels_a_ptr)), ctypes.c_void_p(int(d_labels_b_ptr)),
        offsets.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), n_utt,
        matched.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))), 'uis_eval_accuracy_device')
    return matched[:n_utt]

  def eval_last_decode(self, truth, n_utt):
    """Matched positions of the last decode()'s labels (still in HBM) against `truth`."""
    truth = np.ascontiguousarray(truth, dtype=np.int32)
    matched = np.zeros(max(int(n_utt), 1), dtype=np.int64)
    self._check(self._lib.uis_eval_last_decode(
        self._handle, truth.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), int(n_utt),
        matched.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))), 'uis_eval_last_decode')
    return matched[:int(n_utt)]

  def decode_device(self, d_frames_ptr, offsets, beam_size, look_ahead,
                    test_iteration, d_labels_ptr, d_scores_ptr, max_clusters=0,
                    flags=0, n_streams=0):
    """Decode with frames/labels/scores already resident in HBM (raw pointers)."""
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    n_utt = offsets.shape[0] - 1
    opts = make_opts(beam_size, look_ahead, test_iteration, max_clusters, flags,
                     n_streams)
    stats = Stats()
    rc = self._lib.uis_decode_device(
        self._handle, ctypes.c_void_p(int(d_frames_ptr)),
        offsets.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), n_utt,
        ctypes.byref(opts), ctypes.c_void_p(int(d_labels_ptr)),
        ctypes.c_void_p(int(d_scores_ptr) if d_scores_ptr else None),
        ctypes.byref(stats))
    rc = self._check(rc, 'uis_decode_device')
    return {'stats': stats.as_dict(), 'status': rc}
