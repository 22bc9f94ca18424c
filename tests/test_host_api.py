"""Host-side mirror of the reference interface: arguments, errors, weights, C ABI surface.

No GPU needed.  Expected strings / defaults are the reference's
(uisrnn/arguments.py:30-205, uisrnn/uisrnn.py:510-521,585-590,614-615).
"""

import ctypes
import os
import re
import sys

import numpy as np
import pytest

import uisrnn_amd
from uisrnn_amd import _capi
from uisrnn_amd import arguments
from uisrnn_amd import synth
from uisrnn_amd import weights

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _args(**over):
  model_args, _, inference_args = uisrnn_amd.parse_arguments([])
  model_args.observation_dim = 16
  model_args.rnn_hidden_size = 8
  model_args.transition_bias = 0.2
  model_args.sigma2 = 0.05
  for key, val in over.items():
    setattr(model_args, key, val)
  return model_args, inference_args


def test_parse_arguments_defaults():
  model_args, training_args, inference_args = uisrnn_amd.parse_arguments([])
  assert vars(model_args) == dict(
      observation_dim=256, rnn_hidden_size=512, rnn_depth=1, rnn_dropout=0.2,
      transition_bias=None, crp_alpha=1.0, sigma2=None, verbosity=3,
      enable_cuda=True)
  assert vars(inference_args) == dict(beam_size=10, look_ahead=1,
                                      test_iteration=2)
  assert training_args.optimizer == 'adam'
  assert training_args.train_iteration == 20000
  assert training_args.learning_rate == 1e-3
  assert training_args.enforce_cluster_id_uniqueness is True


def test_parse_arguments_flags_and_str2bool():
  model_args, _, inference_args = uisrnn_amd.parse_arguments(
      ['--observation_dim', '32', '-s', '5', '--look_ahead', '2',
       '--enable_cuda', 'no', '--transition_bias', '0.1'])
  assert model_args.observation_dim == 32 and model_args.enable_cuda is False
  assert model_args.transition_bias == 0.1
  assert inference_args.beam_size == 5 and inference_args.look_ahead == 2
  assert arguments.str2bool('Yes') is True and arguments.str2bool('0') is False
  with pytest.raises(Exception):
    arguments.str2bool('maybe')
  with pytest.raises(SystemExit):
    uisrnn_amd.parse_arguments(['--no_such_flag', '1'])


def test_predict_argument_errors():
  """Same exception types and messages as uisrnn/uisrnn.py:510-521,590,614-615."""
  model_args, inference_args = _args()
  model = uisrnn_amd.UISRNN(model_args)
  with pytest.raises(TypeError, match='test_sequence should be a numpy array of float type.'):
    model.predict_single([[1.0] * 16], inference_args)
  with pytest.raises(TypeError, match='numpy array of float type'):
    model.predict_single(np.zeros((4, 16), dtype=np.float32), inference_args)
  with pytest.raises(ValueError, match='test_sequence must be 2-dim array.'):
    model.predict_single(np.zeros(16), inference_args)
  with pytest.raises(ValueError, match='does not match the dimension specified by args.observation_dim'):
    model.predict_single(np.zeros((4, 15)), inference_args)
  with pytest.raises(ValueError):
    model.predict([np.zeros((4, 16)), np.zeros((4, 3))], inference_args)
  with pytest.raises(TypeError, match='test_sequences should be either a list or numpy array.'):
    model.predict('abc', inference_args)
  with pytest.raises(TypeError, match='test_sequences must be a list.'):
    uisrnn_amd.parallel_predict(model, np.zeros((4, 16)), inference_args)
  assert model.predict([], inference_args) == []
  with pytest.raises(NotImplementedError):
    model.fit(np.zeros((4, 16)), ['a'] * 4, None)


def test_untrained_transition_bias_raises_like_reference():
  model_args, inference_args = _args(transition_bias=None)
  model = uisrnn_amd.UISRNN(model_args)
  with pytest.raises(TypeError):
    model.predict(np.zeros((4, 16)), inference_args)


@pytest.mark.skipif(_capi.load_library().uis_device_count() > 0,
                    reason='a GPU is visible: covered by the gpu tests')
def test_no_gpu_fails_loudly():
  """No silent CPU fallback: without a device the decode path raises."""
  model_args, inference_args = _args()
  model = uisrnn_amd.UISRNN(model_args)
  with pytest.raises(_capi.HipLibraryError, match='no HIP device|no CPU fallback'):
    model.predict(np.zeros((4, 16)), inference_args)


def test_product_never_imports_the_oracle():
  pkg = os.path.join(ROOT, 'uisrnn_amd')
  for dirpath, _, files in os.walk(pkg):
    for name in files:
      if name.endswith(('.py', '.hip', '.h')):
        text = open(os.path.join(dirpath, name)).read()
        assert not re.search(r'^\s*(from|import)\s+oracle', text, re.M), name
        assert not re.search(r'#\s*include\s*[<"][^>"]*oracle', text), name
        assert 'liboracle' not in text, name
        assert not re.search(r'uis_oracle_\w+\s*\(', text), name  # no calls into the checker


def test_library_exports_every_declared_symbol():
  header = open(os.path.join(ROOT, 'include', 'uisrnn_hip.h')).read()
  declared = set(re.findall(r'^\s*(?:const\s+char\*|int32_t|uint32_t|void)\s+(uis_\w+)\s*\(',
                            header, re.M))
  assert declared == set(_capi.EXPORTED_SYMBOLS)
  lib = _capi.load_library()
  for name in declared:
    assert hasattr(lib, name), name
  assert lib.uis_abi_version() == _capi.UIS_ABI_VERSION == 6
  version = int(re.search(r'#define UIS_NUMERICS_VERSION (\d+)', open(
      os.path.join(ROOT, 'include', 'uis_numerics.h')).read()).group(1))
  assert lib.uis_numerics_version() == version


def test_header_is_plain_c_and_a_c_program_links_against_the_library(tmp_path):
  """The drop-in boundary is a C ABI (plain pointers and sizes, no C++ or torch types): the header
  compiles as pedantic C99, and a C program built with gcc -- no hipcc, no Python -- links against
  libuisrnn_hip.so and gets the library's own answers from the entry points that need no GPU
  (versions, device count, argument checks and their error text).  No compute call."""
  import shutil
  import subprocess
  gcc = shutil.which('gcc')
  if gcc is None:
    pytest.skip('no gcc')
  _capi.load_library()  # (built)
  src = tmp_path / 'abi_check.c'
  src.write_text(r'''
#include <stdio.h>
#include <string.h>
#include "uisrnn_hip.h"
int main(void) {
  uis_decode_opts opts;
  uis_stats stats;
  uis_handle* h = NULL;
  int64_t offsets[1] = {0};
  memset(&opts, 0, sizeof opts);
  memset(&stats, 0, sizeof stats);
  printf("abi %d numerics %d devices %d\n", (int)uis_abi_version(), (int)uis_numerics_version(), (int)uis_device_count());
  printf("create(NULL) %d\n", (int)uis_create(NULL, 0, &h));
  {  /* (the message is read AFTER the call it explains: two statements, not two arguments of one printf) */
    const int rc = (int)uis_decode(NULL, NULL, offsets, 0, &opts, NULL, NULL, &stats);
    printf("decode(NULL) %d [%s]\n", rc, uis_last_error());
  }
  printf("sizes %d %d\n", (int)sizeof(uis_decode_opts), (int)sizeof(uis_stats));
  uis_destroy(NULL);
  return uis_abi_version() == UIS_ABI_VERSION ? 0 : 1;
}
''')
  exe = tmp_path / 'abi_check'
  lib_dir = os.path.dirname(_capi.LIB_PATH)
  subprocess.check_call([gcc, '-std=c99', '-pedantic', '-Wall', '-Wextra', '-Werror', '-I', os.path.join(ROOT, 'include'),
                         str(src), '-o', str(exe), '-L', lib_dir, '-luisrnn_hip', '-Wl,-rpath,' + lib_dir])
  out = subprocess.run([str(exe)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
  assert out.returncode == 0, (out.stdout, out.stderr)
  lines = out.stdout.strip().splitlines()
  assert lines[0].startswith('abi {} numerics '.format(_capi.UIS_ABI_VERSION)), lines
  assert lines[1] == 'create(NULL) {}'.format(_capi.UIS_ERR_INVALID_ARG), lines
  assert lines[2].startswith('decode(NULL) {} ['.format(_capi.UIS_ERR_INVALID_ARG)) and 'null handle' in lines[2], lines
  # the ctypes mirror of the two structs has the C compiler's sizes
  assert lines[3] == 'sizes {} {}'.format(ctypes.sizeof(_capi.DecodeOpts), ctypes.sizeof(_capi.Stats)), lines


def test_driver_build_entry_point():
  """__graft_entry__.build() is what the driver runs as its "does it build" check."""
  sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
  import __graft_entry__
  __graft_entry__.build()
  # the header, the library and the Python constants agree on the ABI version
  header = open(os.path.join(os.path.dirname(__graft_entry__.__file__), 'include', 'uisrnn_hip.h')).read()
  assert int(re.search(r'#define UIS_ABI_VERSION (\d+)', header).group(1)) == _capi.UIS_ABI_VERSION
  flags = dict(re.findall(r'#define (UIS_FLAG_\w+)\s+(0x[0-9a-fA-F]+)u', header))
  for name, value in flags.items():
    assert getattr(_capi, name) == int(value, 16), name


def test_initial_cluster_cap_keeps_wide_beams_on_the_fast_path():
  import argparse
  from uisrnn_amd import uisrnn as host
  def cap(beam, look=1, explicit=0):
    return host._initial_cluster_cap(argparse.Namespace(beam_size=beam, look_ahead=look, max_clusters=explicit))
  assert cap(10) == 16 and cap(15) == 16          # 15 * 17 = 255 candidates: fits as it is
  assert cap(20) == 11 and 20 * (cap(20) + 1) <= 256
  assert cap(28) == 8 and cap(29) == 16           # below 8 clusters it is not worth it
  assert cap(20, look=2) == 16                    # look_ahead >= 2 has its own kernel
  assert cap(20, explicit=32) == 32               # the caller's word wins


def test_output_result_known_answer(tmp_path, monkeypatch):
  """uisrnn.output_result (uisrnn/utils.py:253-285): the expected text below is what the
  reference prints for these arguments (recorded from the reference in the dev container)."""
  monkeypatch.chdir(tmp_path)
  model_args, training_args, _ = uisrnn_amd.parse_arguments([])
  text = uisrnn_amd.output_result(model_args, training_args, [(0.93, 120), (1.0, 64), (0.5, 7)])
  expected = (
      'Config:\n  sigma_alpha: 1.0\n  sigma_beta: 1.0\n  crp_alpha: 1.0\n  learning rate: 0.001\n'
      '  regularization: 1e-05\n  batch size: 10\n\nPerformance:\n  averaged accuracy: 0.810000\n'
      '  accuracy numbers for all testing sequences:\n    0.930000\n    1.000000\n    0.500000\n'
      + '=' * 80 + '\n')
  assert text == expected
  assert (tmp_path / 'layer_512_1_0.2_result.txt').read_text() == expected
  uisrnn_amd.output_result(model_args, training_args, [(1.0, 3)])       # appends, like the reference
  assert (tmp_path / 'layer_512_1_0.2_result.txt').read_text().count('Config:') == 2


def test_struct_layouts_match_header():
  """ctypes mirrors of the header structs (LP64): sizes and a few offsets."""
  assert ctypes.sizeof(_capi.ModelDesc) == 16 + 10 * 8 + 16
  assert _capi.ModelDesc.transition_bias.offset == 96
  assert ctypes.sizeof(_capi.DecodeOpts) == 32
  assert _capi.Stats.kernel_ms.offset == 40
  assert ctypes.sizeof(_capi.Stats) == 40 + 8 * 8 + 8 * 8 + 16
  assert _capi.Stats.decode_kernel.offset == 40 + 8 * 8 + 8 * 8 + 8


def test_c_abi_rejects_bad_arguments_without_a_device():
  lib = _capi.load_library()
  assert lib.uis_create(None, 0, None) == _capi.UIS_ERR_INVALID_ARG
  assert b'null' in lib.uis_last_error()
  params = weights.init_params(4, 4, 1, sigma2=0.1, transition_bias=1.5, seed=0)
  desc, keep = _capi.make_desc(params)
  handle = ctypes.c_void_p()
  rc = lib.uis_create(ctypes.byref(desc), 0, ctypes.byref(handle))
  assert rc == _capi.UIS_ERR_INVALID_ARG  # transition_bias outside (0, 1)
  del keep


def test_params_round_trips(tmp_path):
  params = weights.init_params(6, 5, 2, sigma2=0.3, transition_bias=0.25,
                               crp_alpha=1.5, seed=3)
  assert params['gru_weight_ih'][0].shape == (15, 6)
  assert params['gru_weight_ih'][1].shape == (15, 5)
  state = weights.state_dict_from_params(params)
  assert set(state) == {
      'gru.weight_ih_l0', 'gru.weight_hh_l0', 'gru.bias_ih_l0', 'gru.bias_hh_l0',
      'gru.weight_ih_l1', 'gru.weight_hh_l1', 'gru.bias_ih_l1', 'gru.bias_hh_l1',
      'linear_mean1.weight', 'linear_mean1.bias', 'linear_mean2.weight',
      'linear_mean2.bias'}
  back = weights.params_from_state(state, params['rnn_init_hidden'],
                                   params['sigma2'], 0.25, 1.5)
  assert back['rnn_depth'] == 2 and back['observation_dim'] == 6
  # the reference's checkpoint format (uisrnn/uisrnn.py:141-147)
  path = str(tmp_path / 'model.uisrnn')
  weights.save_checkpoint(params, path)
  import torch
  raw = torch.load(path, weights_only=False)
  assert set(raw) == {'rnn_state_dict', 'rnn_init_hidden', 'transition_bias',
                      'transition_bias_denominator', 'crp_alpha', 'sigma2'}
  assert raw['rnn_init_hidden'].shape == (2, 1, 5)
  loaded = weights.load_checkpoint(path)
  for key in ('linear_mean2_weight', 'sigma2', 'rnn_init_hidden'):
    assert np.array_equal(loaded[key], params[key])
  assert np.array_equal(loaded['gru_weight_hh'][1], params['gru_weight_hh'][1])
  assert loaded['transition_bias'] == 0.25 and loaded['crp_alpha'] == 1.5
  # and through the model object
  model_args, _ = _args(observation_dim=6, rnn_hidden_size=5, rnn_depth=2)
  model = uisrnn_amd.UISRNN(model_args)
  model.load(path)
  assert model.transition_bias == 0.25
  assert model.rnn_init_hidden.shape == (2, 1, 5)
  model.save(str(tmp_path / 'again.uisrnn'))


def _same_params(a, b):
  for key, val in a.items():
    if isinstance(val, list):
      assert all(np.array_equal(x, y) and x.dtype == y.dtype for x, y in zip(val, b[key])), key
    elif isinstance(val, np.ndarray):
      assert np.array_equal(val, b[key]) and val.dtype == b[key].dtype, key
    else:
      assert val == b[key], key


def test_reference_checkpoint_is_read_without_torch(monkeypatch):
  """tests/golden/d20_h24_depth3.uisrnn was written by the REFERENCE's UISRNN.save()
  (tests/golden/make_golden.py --checkpoint); it must load with torch unimportable and give
  exactly the parameters the golden case was generated from."""
  import golden_util
  path = os.path.join(golden_util.GOLDEN_DIR, 'd20_h24_depth3.uisrnn')
  expect = golden_util.load_case('d20_h24_depth3')['params']
  monkeypatch.setitem(sys.modules, 'torch', None)   # `import torch` now raises ImportError
  with pytest.raises(ImportError):
    import torch  # noqa: F401  pylint: disable=unused-import,import-outside-toplevel
  got = weights.load_checkpoint(path)
  assert got['rnn_depth'] == 3 and got['rnn_hidden_size'] == 24 and got['observation_dim'] == 20
  assert np.abs(got['rnn_init_hidden']).max() > 0
  for key in ('gru_weight_ih', 'gru_weight_hh', 'gru_bias_ih', 'gru_bias_hh', 'linear_mean1_weight',
              'linear_mean1_bias', 'linear_mean2_weight', 'linear_mean2_bias', 'rnn_init_hidden', 'sigma2'):
    a, b = got[key], expect[key]
    if isinstance(a, list):
      assert all(np.array_equal(x, np.asarray(y, dtype=np.float32)) for x, y in zip(a, b)), key
    else:
      assert np.array_equal(a, np.asarray(b, dtype=np.float32).reshape(a.shape)), key
  assert got['transition_bias'] == pytest.approx(float(expect['transition_bias']))
  assert got['crp_alpha'] == pytest.approx(float(expect['crp_alpha']))


def test_checkpoint_is_written_without_torch(tmp_path, monkeypatch):
  """save() produces the reference's torch.save file with torch unimportable; torch.load (the
  reference's reader) then returns real tensors with the same bits."""
  params = weights.init_params(12, 10, 2, sigma2=0.3, transition_bias=0.15, crp_alpha=2.0, seed=5)
  params['rnn_init_hidden'] = np.random.default_rng(6).standard_normal((2, 10)).astype(np.float32)
  path = str(tmp_path / 'written.uisrnn')
  with monkeypatch.context() as mp:
    mp.setitem(sys.modules, 'torch', None)
    weights.save_checkpoint(params, path)
    _same_params(weights.load_checkpoint(path), {**params, 'transition_bias_denominator': 0.0})
  import torch
  raw = torch.load(path, weights_only=False)
  assert list(raw) == ['rnn_state_dict', 'rnn_init_hidden', 'transition_bias',
                       'transition_bias_denominator', 'crp_alpha', 'sigma2']
  state = weights.state_dict_from_params(params)
  assert list(raw['rnn_state_dict']) == list(state)
  for key, val in state.items():
    got = raw['rnn_state_dict'][key]
    assert isinstance(got, torch.Tensor) and got.dtype == torch.float32 and got.is_contiguous()
    assert np.array_equal(got.numpy(), val), key
  assert raw['rnn_init_hidden'].shape == (2, 1, 10) and raw['crp_alpha'] == 2.0
  # torch can also take it through its own nn.Module machinery, like the reference's load()
  gru = torch.nn.GRU(12, 10, 2)
  gru.load_state_dict({k[len('gru.'):]: v for k, v in raw['rnn_state_dict'].items() if k.startswith('gru.')})


def test_torch_free_reader_matches_torch_load(tmp_path):
  import pickle
  import torch
  params = weights.init_params(7, 9, 2, sigma2=0.2, transition_bias=0.4, seed=8)
  path = str(tmp_path / 'm.uisrnn')
  weights.save_checkpoint(params, path)
  raw = torch.load(path, weights_only=False)
  via_torch = weights.params_from_state(raw['rnn_state_dict'], raw['rnn_init_hidden'], raw['sigma2'],
                                        raw['transition_bias'], raw['crp_alpha'],
                                        raw['transition_bias_denominator'])
  _same_params(via_torch, weights.load_checkpoint(path))
  # views: transposed, offset into a shared storage, 0-d, other dtypes
  base = torch.arange(24, dtype=torch.float32).reshape(4, 6)
  obj = {'t': base.t(), 'window': base[1:3, 2:5], 'scalar': torch.tensor(3.5),
         'long': torch.arange(5), 'half': torch.ones(3, dtype=torch.float16), 'list': [base[0], 'text', 7]}
  torch.save(obj, str(tmp_path / 'views.pt'))
  got = weights.read_torch_zip(str(tmp_path / 'views.pt'))
  assert np.array_equal(got['t'], base.t().numpy()) and got['t'].flags['C_CONTIGUOUS']
  assert np.array_equal(got['window'], base[1:3, 2:5].numpy())
  assert float(got['scalar']) == 3.5 and got['long'].dtype == np.int64 and got['half'].dtype == np.float16
  assert np.array_equal(got['list'][0], base[0].numpy()) and got['list'][1:] == ['text', 7]
  # nothing outside a checkpoint's vocabulary is ever resolved, let alone called

  class Evil:
    def __reduce__(self):
      return (os.system, ('true',))
  torch.save({'x': Evil()}, str(tmp_path / 'evil.pt'))
  with pytest.raises(pickle.UnpicklingError):
    weights.read_torch_zip(str(tmp_path / 'evil.pt'))
  with pytest.raises(ValueError):
    weights.read_torch_zip(__file__)


def test_synthetic_generators_are_deterministic():
  a, ida = synth.make_utterance(7, 50, 32)
  b, idb = synth.make_utterance(7, 50, 32)
  assert a.dtype == np.float64 and a.shape == (50, 32)
  assert np.array_equal(a, b) and np.array_equal(ida, idb)
  p1 = synth.tracker_params(32, 48, 2, seed=1)
  p2 = synth.tracker_params(32, 48, 2, seed=1)
  assert np.array_equal(p1['gru_weight_hh'][1], p2['gru_weight_hh'][1])
  assert p1['gru_weight_ih'][1].shape == (144, 48)
  with pytest.raises(ValueError):
    synth.tracker_params(64, 32)
  assert synth.relabel_first_occurrence([5, 5, 2, 5, 9]) == [0, 0, 1, 0, 2]


def test_tracker_model_diarizes_synthetic_speech(oracle_lib):
  """The closed-form benchmark model behaves like a trained one (accuracy ~1)."""
  params = synth.tracker_params(256, 512, 1, seed=0)
  seqs, ids = synth.make_utterances(100, 3, 120, 256)
  out = oracle_lib.decode(params, seqs, 10, 1, 2, n_threads=3)
  for labels, truth in zip(out['labels'], ids):
    acc = uisrnn_amd.compute_sequence_match_accuracy(labels.tolist(),
                                                     truth.tolist())
    assert acc > 0.97
  assert out['max_clusters'].max() <= 7


def test_sequence_match_accuracy_known_answers():
  """Known answers of the reference's tests/evals_test.py:36-82."""
  acc = uisrnn_amd.compute_sequence_match_accuracy
  assert acc([0, 0, 1, 2, 2], [3, 3, 4, 4, 1]) == 0.8
  assert acc([0, 0, 0, 1, 2], [3, 3, 3, 4, 1]) == 1.0
  assert acc([1, 1], [1, 2]) == 0.5
  assert acc(['a', 'b', 'b'], [1, 2, 2]) == 1.0
  a, b = [0, 1, 1, 2, 0, 3], [4, 4, 5, 6, 4, 4]
  assert acc(a, b) == acc(b, a)
  with pytest.raises(TypeError):
    acc(np.array([0, 1]), [0, 1])
  with pytest.raises(ValueError):
    acc([0, 1], [0])
  with pytest.raises(ValueError):
    acc([], [])


def test_predict_splits_a_list_that_does_not_fit_the_device():
  """The reference's predict takes a list of any size (uisrnn.py:588-589).  When the library reports that
  the decode state of a batch does not fit (UIS_ERR_OOM), the host layer decodes the list in halves --
  recursively -- and hands the labels back in the caller's order.  (Plumbing test with a stand-in decoder:
  no device here.)"""
  from uisrnn_amd import uisrnn as host

  class StandIn:
    """Decodes at most `room` utterances at a time; labels = the utterance's own tag, so order is checkable."""
    def __init__(self, room):
      self.room, self.batches = room, []

    def decode_f64(self, seqs, beam_size, look_ahead, test_iteration, max_clusters=0, flags=0, level_cap=0):
      if len(seqs) > self.room:
        err = _capi.HipLibraryError('uis_decode_f64 failed (-5): decode state would need 999 GB')
        err.status = _capi.UIS_ERR_OOM
        raise err
      self.batches.append(len(seqs))
      labels = np.concatenate([np.full(s.shape[0], int(s[0, 0]), dtype=np.int32) for s in seqs])
      return {'status': 0, 'labels': labels, 'overflow': np.zeros(len(seqs), dtype=np.int32),
              'stats': {'decode_kernel': 'stand-in'}}

  model_args, _, inference_args = arguments.parse_arguments([])
  model_args.observation_dim = 4
  model = uisrnn_amd.UISRNN(model_args)
  seqs = [np.full((3 + k % 4, 4), float(k)) for k in range(11)]
  dec = StandIn(room=3)
  out = model._decode_batch(seqs, inference_args, decoder=dec)  # pylint: disable=protected-access
  assert [lab[0] for lab in out] == list(range(11))
  assert [len(lab) for lab in out] == [s.shape[0] for s in seqs]
  assert sum(dec.batches) == 11 and max(dec.batches) <= 3
  # a single utterance that does not fit is an error, not a loop
  with pytest.raises(_capi.HipLibraryError):
    model._decode_batch(seqs[:1], inference_args, decoder=StandIn(room=0))  # pylint: disable=protected-access
  assert host is not None


def test_predict_retries_an_overflowing_look_ahead_window_with_more_room():
  """Round 5 (plumbing test with a stand-in decoder: no device here).  The library reports a look-ahead window whose
  live prefixes outgrow a level's capacity per utterance (UIS_ERR_UNSUPPORTED + bit 1 of the flags); the host layer
  decodes the other utterances again at the same capacity and the affected ones with eight times the room, up to
  the library's maximum, and only then raises LookAheadWindowError -- numbered like the caller's list, with every
  decodable utterance's labels -- also when the list had to be split first because its state did not fit."""
  from uisrnn_amd import uisrnn as host

  class StandIn:
    """Utterance k needs a level capacity of need[k] (its tag); decodes at most `room` utterances at a time."""
    def __init__(self, room=99):
      self.room, self.calls, self.flags = room, [], np.zeros(0, dtype=np.int32)

    def decode_f64(self, seqs, beam_size, look_ahead, test_iteration, max_clusters=0, flags=0, level_cap=0):
      cap = level_cap or 32768
      self.calls.append((len(seqs), cap))
      if len(seqs) > self.room:
        err = _capi.HipLibraryError('uis_decode_f64 failed (-5): decode state would need 999 GB')
        err.status = _capi.UIS_ERR_OOM
        self.flags = np.zeros(0, dtype=np.int32)
        raise err
      self.flags = np.array([2 if int(s[0, 1]) > cap else 0 for s in seqs], dtype=np.int32)
      if self.flags.any():
        err = _capi.HipLibraryError('uis_decode_f64 failed (-7): a look-ahead window held more prefixes than a level')
        err.status = _capi.UIS_ERR_UNSUPPORTED
        raise err
      labels = np.concatenate([np.full(s.shape[0], int(s[0, 0]), dtype=np.int32) for s in seqs])
      return {'status': 0, 'labels': labels, 'overflow': np.zeros(len(seqs), dtype=np.int32),
              'stats': {'decode_kernel': 'stand-in'}}

    def last_overflow(self, n_utt=None):
      return self.flags

  model_args, _, inference_args = arguments.parse_arguments([])
  model_args.observation_dim = 4
  inference_args.look_ahead = 3
  model = uisrnn_amd.UISRNN(model_args)
  need = [100, 40000, 100, 300000, 100, 100, 9_000_000, 100]
  seqs = []
  for k, n in enumerate(need):
    s = np.zeros((2 + k % 3, 4))
    s[:, 0], s[:, 1] = k, n
    seqs.append(s)
  ok = [k for k, n in enumerate(need) if n <= host._MAX_LEVEL_CAP]  # pylint: disable=protected-access
  # everything that fits some capacity comes back, in the caller's order; the one that fits none is named
  for room in (99, 3):
    dec = StandIn(room)
    with pytest.raises(uisrnn_amd.LookAheadWindowError) as info:
      model._decode_batch(seqs, inference_args, decoder=dec)  # pylint: disable=protected-access
    assert info.value.utterances == (6,), (room, info.value.utterances)
    assert [None if r is None else r[0] for r in info.value.results] == [k if k in ok else None for k in range(8)]
    assert max(cap for _, cap in dec.calls) == host._MAX_LEVEL_CAP  # pylint: disable=protected-access
  # without the hopeless one: no exception at all
  dec = StandIn()
  out = model._decode_batch(seqs[:6], inference_args, decoder=dec)  # pylint: disable=protected-access
  assert [r[0] for r in out] == list(range(6)) and [len(r) for r in out] == [s.shape[0] for s in seqs[:6]]
  assert sorted({cap for _, cap in dec.calls}) == [32768, 262144, 524287]
  # one utterance whose window does not fit the DEVICE at the larger capacity (UIS_ERR_OOM from a single-utterance decode):
  # only that one is named, the others of its retry group come back (round 6)
  class TightDevice(StandIn):
    def decode_f64(self, seqs, beam_size, look_ahead, test_iteration, max_clusters=0, flags=0, level_cap=0):
      if (level_cap or 32768) > 32768 and any(int(s[0, 0]) == 3 for s in seqs):
        self.calls.append((len(seqs), level_cap))
        err = _capi.HipLibraryError('uis_decode_f64 failed (-5): decode state would need 999 GB')
        err.status = _capi.UIS_ERR_OOM
        self.flags = np.zeros(0, dtype=np.int32)
        raise err
      return StandIn.decode_f64(self, seqs, beam_size, look_ahead, test_iteration, max_clusters, flags, level_cap)
  dec = TightDevice()
  with pytest.raises(uisrnn_amd.LookAheadWindowError) as info:
    model._decode_batch(seqs[:6], inference_args, decoder=dec)  # pylint: disable=protected-access
  assert info.value.utterances == (3,), info.value.utterances
  assert [None if r is None else r[0] for r in info.value.results] == [0, 1, 2, None, 4, 5]
  # args.level_cap: where the retries start
  inference_args.level_cap = 64
  dec = StandIn()
  out = model._decode_batch(seqs[:3], inference_args, decoder=dec)  # pylint: disable=protected-access
  assert [r[0] for r in out] == [0, 1, 2] and dec.calls[0] == (3, 64)


_REFERENCE_TESTS = '/root/reference/tests'


@pytest.mark.skipif(not os.path.isdir(_REFERENCE_TESTS), reason='the reference checkout is not on this machine')
def test_the_references_own_fit_free_tests_pass_against_this_package():
  """Drop-in at the Python surface: google/uis-rnn's OWN unit tests -- the files as they lie under
  /root/reference/tests, unmodified, nothing copied -- run with `uisrnn` resolving to this package: all of
  evals_test.py (get_list_inverse_index, compute_sequence_match_accuracy: values, symmetry, error cases)
  and uisrnn_test.py's test_save_and_load (UISRNN(args).save / .load round trip through the torch-free
  checkpoint writer).  The others in uisrnn_test.py / integration_test.py call fit(): out of scope
  (SURVEY.md section 8).  A subprocess: the reference's tests call parse_arguments() on sys.argv."""
  import subprocess
  prog = r'''
import sys, unittest
sys.path.insert(0, {root!r})
import uisrnn_amd
from uisrnn_amd import evals
sys.modules['uisrnn'] = uisrnn_amd          # `import uisrnn` / `from uisrnn import evals` in the reference's files
sys.modules['uisrnn.evals'] = evals
sys.path.insert(0, {ref!r})
sys.argv = ['reference-tests']
import evals_test, uisrnn_test
loader = unittest.TestLoader()
suite = unittest.TestSuite([loader.loadTestsFromModule(evals_test),
                            loader.loadTestsFromName('uisrnn_test.TestUISRNN.test_save_and_load')])
result = unittest.TextTestRunner(verbosity=0, stream=sys.stderr).run(suite)
assert sys.modules['uisrnn'] is uisrnn_amd and evals_test.evals is evals
print('ran', result.testsRun, 'failures', len(result.failures), 'errors', len(result.errors), 'skipped', len(result.skipped))
sys.exit(0 if result.wasSuccessful() else 1)
'''.format(root=ROOT, ref=_REFERENCE_TESTS)
  out = subprocess.run([sys.executable, '-c', prog], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300,
                       cwd='/tmp')  # (not the reference's directory: `uisrnn` must resolve through sys.modules only)
  assert out.returncode == 0, (out.stdout, out.stderr[-3000:])
  assert out.stdout.strip().splitlines()[-1] == 'ran 8 failures 0 errors 0 skipped 0', (out.stdout, out.stderr[-2000:])


def test_library_and_torch_share_one_hip_runtime_in_either_load_order():
  """Round 6: loading libuisrnn_hip.so BEFORE PyTorch used to bring two HIP runtimes into the process (torch then finds
  no GPU).  _capi.share_hip_runtime_with_torch() preloads torch's bundled libamdhip64.so (same SONAME as the system
  one) where a torch installation exists: ONE runtime mapped whichever comes first; UIS_HIP_RUNTIME=system opts out."""
  import subprocess
  probe = ('import sys; sys.path.insert(0, {root!r})\n'
           '{first}\n'
           'from uisrnn_amd import _capi\n'
           '_capi.load_library()\n'
           '{second}\n'
           'maps = sorted(set(l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l))\n'
           'print(len(maps), _capi._hip_runtime)\n')
  for first, second, env, want in (('', 'import torch', {}, '1 torch (preloaded'),
                                   ('import torch', '', {}, '1 torch (imported before'),
                                   ('', '', {'UIS_HIP_RUNTIME': 'system'}, '1 system (UIS_HIP_RUNTIME)')):
    out = subprocess.run([sys.executable, '-c', probe.format(root=ROOT, first=first, second=second)],
                         capture_output=True, text=True, timeout=600, env=dict(os.environ, **env))
    assert out.returncode == 0 and out.stdout.startswith(want), (first, second, out.stdout, out.stderr[-2000:])
