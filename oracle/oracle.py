"""ctypes wrapper of oracle/liboracle.so (the CPU checker).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  uisrnn_amd/ never imports this module.
"""

import ctypes
import os
import subprocess

import numpy as np

from uisrnn_amd import _capi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'liboracle.so')
_lib = None


def build(force=False):
  """Compile liboracle.so with gcc (oracle/Makefile)."""
  if force and os.path.exists(LIB_PATH):
    os.remove(LIB_PATH)
  subprocess.check_call(['make', '-s', '-C', _HERE])
  return LIB_PATH


def lib():
  global _lib
  if _lib is None:
    build()  # make: a no-op when liboracle.so is newer than its sources
    _lib = ctypes.CDLL(LIB_PATH)
    _lib.uis_oracle_numerics_version.restype = ctypes.c_int32
    fp = ctypes.POINTER(ctypes.c_float)
    i32p = ctypes.POINTER(ctypes.c_int32)
    i64p = ctypes.POINTER(ctypes.c_int64)
    _lib.uis_oracle_decode.restype = ctypes.c_int32
    _lib.uis_oracle_decode.argtypes = [
        ctypes.POINTER(_capi.ModelDesc), fp, i64p, ctypes.c_int32,
        ctypes.POINTER(_capi.DecodeOpts), ctypes.c_int32, i32p, fp, fp, fp,
        i32p, i64p]
    _lib.uis_oracle_rnn_step.restype = ctypes.c_int32
    _lib.uis_oracle_rnn_step.argtypes = [
        ctypes.POINTER(_capi.ModelDesc), fp, fp, fp, fp]
    _lib.uis_oracle_weighted_mse.restype = ctypes.c_float
    _lib.uis_oracle_weighted_mse.argtypes = [
        ctypes.POINTER(_capi.ModelDesc), fp, fp]
    _lib.uis_oracle_constants.restype = ctypes.c_int32
    _lib.uis_oracle_constants.argtypes = [
        ctypes.POINTER(_capi.ModelDesc), fp, fp]
  return _lib


def numerics_version():
  return int(lib().uis_oracle_numerics_version())


def _fp(a):
  return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def pack(sequences):
  """list of [N_i, D] arrays -> (float32 [sum N, D], int64 offsets [U+1])."""
  lens = [int(s.shape[0]) for s in sequences]
  offsets = np.zeros(len(sequences) + 1, dtype=np.int64)
  offsets[1:] = np.cumsum(lens)
  dim = sequences[0].shape[1] if sequences else 0
  frames = np.empty((int(offsets[-1]), dim), dtype=np.float32)
  for seq, start in zip(sequences, offsets[:-1]):
    frames[start:start + seq.shape[0]] = seq  # float64 -> float32 (RNE), as .float()
  return frames, offsets


def decode(params, sequences, beam_size=10, look_ahead=1, test_iteration=2,
           n_threads=1):
  """Oracle decode of a list of [N_i, D] arrays.

  Returns dict: labels (list of int32 arrays), scores [U], beam_scores [U,B],
  margins [U], max_clusters [U], rnn_calls, candidates.
  """
  frames, offsets = pack(sequences)
  n_utt = len(sequences)
  desc, keep = _capi.make_desc(params)
  opts = _capi.make_opts(beam_size, look_ahead, test_iteration)
  labels = np.full(int(offsets[-1]), -7, dtype=np.int32)
  scores = np.zeros(n_utt, dtype=np.float32)
  beam_scores = np.zeros((n_utt, beam_size), dtype=np.float32)
  margins = np.zeros(n_utt, dtype=np.float32)
  max_clusters = np.zeros(n_utt, dtype=np.int32)
  counters = np.zeros(2, dtype=np.int64)
  rc = lib().uis_oracle_decode(
      ctypes.byref(desc), _fp(frames),
      offsets.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), n_utt,
      ctypes.byref(opts), int(n_threads),
      labels.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), _fp(scores),
      _fp(beam_scores), _fp(margins),
      max_clusters.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
      counters.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)))
  del keep
  if rc != 0:
    raise RuntimeError('uis_oracle_decode failed: {}'.format(rc))
  return {
      'labels': [labels[offsets[u]:offsets[u + 1]].copy() for u in range(n_utt)],
      'scores': scores, 'beam_scores': beam_scores, 'margins': margins,
      'max_clusters': max_clusters, 'rnn_calls': int(counters[0]),
      'candidates': int(counters[1]),
  }


def candidate_scores(params, sequence, beam_size, look_ahead, test_iteration, cmax):
  """Every candidate score of every window of ONE utterance: float32
  [windows, beam_size] + [cmax] * look_ahead, +inf where the reference's padded score_set
  (uisrnn/uisrnn.py:534-545) holds +inf."""
  frames, _ = pack([sequence])
  desc, keep = _capi.make_desc(params)
  opts = _capi.make_opts(beam_size, look_ahead, test_iteration)
  total = test_iteration * frames.shape[0]
  n_win = (total + look_ahead - 1) // look_ahead
  out = np.empty([n_win, beam_size] + [cmax] * look_ahead, dtype=np.float32)
  fn = lib().uis_oracle_candidate_scores
  fn.restype = ctypes.c_int32
  rc = fn(ctypes.byref(desc), _fp(frames), ctypes.c_int64(frames.shape[0]), ctypes.byref(opts),
          ctypes.c_int32(cmax), _fp(out), None)
  del keep
  if rc != 0:
    raise RuntimeError('uis_oracle_candidate_scores failed: {}'.format(rc))
  return out


def rnn_step(params, x, h_in):
  """CoreRNN.forward restatement: x [D], h_in [depth, H] -> (mean, h_out)."""
  desc, keep = _capi.make_desc(params)
  x = np.ascontiguousarray(x, dtype=np.float32)
  h_in = np.ascontiguousarray(h_in, dtype=np.float32)
  mean = np.empty(params['observation_dim'], dtype=np.float32)
  h_out = np.empty_like(h_in)
  lib().uis_oracle_rnn_step(ctypes.byref(desc), _fp(x), _fp(h_in), _fp(mean),
                            _fp(h_out))
  del keep
  return mean, h_out


def weighted_mse(params, mean, x):
  desc, keep = _capi.make_desc(params)
  mean = np.ascontiguousarray(mean, dtype=np.float32)
  x = np.ascontiguousarray(x, dtype=np.float32)
  v = lib().uis_oracle_weighted_mse(ctypes.byref(desc), _fp(mean), _fp(x))
  del keep
  return np.float32(v)


def constants(params):
  desc, keep = _capi.make_desc(params)
  m0 = np.empty(params['observation_dim'], dtype=np.float32)
  h1 = np.empty((params['rnn_depth'], params['rnn_hidden_size']),
                dtype=np.float32)
  lib().uis_oracle_constants(ctypes.byref(desc), _fp(m0), _fp(h1))
  del keep
  return m0, h1
