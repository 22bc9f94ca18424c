"""Full-size GPU checks (BASELINE.json sizes) through size-independent properties.

The oracle needs ~12 s per 1000-frame utterance, so at full size only a sample is compared
element-wise; everything else is checked through invariants of the decode:
  * batch independence -- an utterance decodes the same alone and inside a big batch,
  * idempotence -- decoding twice gives the same bits,
  * scheduling invariance -- row de-duplication / generic select / groups do not change bits,
  * label structure -- every label sequence is a valid first-appearance-ordered trace suffix.
"""

import numpy as np
import pytest

from uisrnn_amd import _capi
from uisrnn_amd import synth

pytestmark = pytest.mark.gpu


def _decode(dec, seqs, beam, look, tau, **kw):
  lens = [len(s) for s in seqs]
  offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
  frames = np.concatenate(seqs).astype(np.float32)
  out = dec.decode(frames, offsets, beam, look, tau, want_beam_scores=True, **kw)
  assert out['status'] == 0
  return out, offsets


def test_config2_full_size_properties(oracle_lib):
  """configs[1]: 64 utterances x 500 frames x 256-dim, beam 10 (the benchmark workload)."""
  params = synth.tracker_params(256, 512, 1, seed=0)
  seqs, truth = synth.make_utterances(10_000, 64, 500, 256)
  dec = _capi.Decoder(params)
  out, off = _decode(dec, seqs, 10, 1, 2)
  again, _ = _decode(dec, seqs, 10, 1, 2)
  assert np.array_equal(out['labels'], again['labels'])
  assert np.array_equal(out['beam_scores'].view(np.uint32), again['beam_scores'].view(np.uint32))
  for flags, streams in ((_capi.UIS_FLAG_NO_DEDUP, 0), (_capi.UIS_FLAG_GENERIC_SELECT, 0), (0, 4),
                         (_capi.UIS_FLAG_STEPWISE, 0), (_capi.UIS_FLAG_RESIDENT, 0)):
    alt, _ = _decode(dec, seqs, 10, 1, 2, flags=flags, n_streams=streams)
    assert np.array_equal(out['labels'], alt['labels'])
    assert np.array_equal(out['beam_scores'].view(np.uint32), alt['beam_scores'].view(np.uint32))
  # batch independence on a sample, and the oracle on the same sample
  sample = [0, 17, 63]
  for u in sample:
    alone, _ = _decode(dec, [seqs[u]], 10, 1, 2)
    assert np.array_equal(alone['labels'], out['labels'][off[u]:off[u + 1]])
    assert np.array_equal(alone['beam_scores'].view(np.uint32), out['beam_scores'][u:u + 1].view(np.uint32))
  ref = oracle_lib.decode(params, [seqs[u] for u in sample], 10, 1, 2, n_threads=3)
  for k, u in enumerate(sample):
    assert np.array_equal(ref['labels'][k], out['labels'][off[u]:off[u + 1]])
    assert ref['scores'][k].view(np.uint32) == out['scores'][u].view(np.uint32)
  # it is a diarizer: accuracy against the generator's speakers
  from uisrnn_amd import evals
  acc = [evals.compute_sequence_match_accuracy(out['labels'][off[u]:off[u + 1]].tolist(), truth[u].tolist())
         for u in range(64)]
  assert np.mean(acc) > 0.98


def test_config4_shape_one_gpu_share():
  """configs[3] per-GPU share: 1024 utterances x 1000 frames would be 1 M frames; run 1024 x 100
  (same batch width, shorter) and check structure + batch independence."""
  params = synth.tracker_params(256, 512, 1, seed=0)
  seqs, _ = synth.make_utterances(20_000, 1024, 100, 256)
  dec = _capi.Decoder(params)
  out, off = _decode(dec, seqs, 10, 1, 2, flags=_capi.UIS_FLAG_RESIDENT)   # 128 utterances per XCD
  assert out['stats']['n_steps'] == 200
  step, _ = _decode(dec, seqs, 10, 1, 2, flags=_capi.UIS_FLAG_STEPWISE)
  assert np.array_equal(out['labels'], step['labels'])
  assert np.array_equal(out['beam_scores'].view(np.uint32), step['beam_scores'].view(np.uint32))
  for u in (0, 511, 1023):
    alone, _ = _decode(dec, [seqs[u]], 10, 1, 2)
    assert np.array_equal(alone['labels'], out['labels'][off[u]:off[u + 1]])
  labels = out['labels'].reshape(1024, 100)
  assert labels.min() >= 0 and labels.max() < 16
  assert np.isfinite(out['scores']).all()


def test_first_appearance_order_and_ragged_batch():
  """test_iteration=1: each trace starts at 0 and new ids appear in order; lengths 0..300 mixed."""
  params = synth.tracker_params(256, 512, 1, seed=5)
  lengths = [0, 1, 2, 300, 17, 0, 128, 255, 3, 64]
  seqs, _ = synth.make_utterances(30_000, len(lengths), lengths, 256)
  dec = _capi.Decoder(params)
  for look in (1, 2):
    out, off = _decode(dec, seqs, 10, look, 1)
    for u, n in enumerate(lengths):
      lab = out['labels'][off[u]:off[u + 1]]
      assert len(lab) == n
      seen = -1
      for v in lab:
        assert 0 <= v <= seen + 1
        seen = max(seen, v)
    assert out['scores'][0] == 0.0 and out['scores'][5] == 0.0  # empty utterances


def test_config5_and_config3_shapes(oracle_lib):
  """configs[4]: D=512, H=512, beam 20; configs[2]: beam 50, look_ahead 2 -- short, vs the oracle."""
  params = synth.tracker_params(512, 512, 1, seed=0)
  seqs, _ = synth.make_utterances(40_000, 4, [40, 25, 33, 12], 512)
  dec = _capi.Decoder(params)
  ref = oracle_lib.decode(params, seqs, 20, 1, 2, n_threads=4)
  out, off = _decode(dec, seqs, 20, 1, 2, max_clusters=int(ref['max_clusters'].max()))
  for u in range(4):
    assert np.array_equal(ref['labels'][u], out['labels'][off[u]:off[u + 1]])
  assert np.array_equal(ref['beam_scores'].view(np.uint32), out['beam_scores'].view(np.uint32))
  params = synth.tracker_params(256, 512, 1, seed=0)
  seqs, _ = synth.make_utterances(41_000, 2, [30, 21], 256)
  dec = _capi.Decoder(params)
  ref = oracle_lib.decode(params, seqs, 50, 2, 2, n_threads=2)
  out, off = _decode(dec, seqs, 50, 2, 2, max_clusters=int(ref['max_clusters'].max()) + 1)
  for u in range(2):
    assert np.array_equal(ref['labels'][u], out['labels'][off[u]:off[u + 1]])
  assert np.array_equal(ref['beam_scores'].view(np.uint32), out['beam_scores'].view(np.uint32))


def test_handles_release_their_memory():
  """create / decode / destroy in a loop leaves the device memory where it was."""
  import gc
  import torch
  params = synth.tracker_params(256, 512, 1, seed=0)
  seqs, _ = synth.make_utterances(60_000, 8, 40, 256)
  torch.cuda.init()
  dec = _capi.Decoder(params)
  _decode(dec, seqs, 10, 1, 2)
  dec.close()
  gc.collect()
  torch.cuda.synchronize()
  free0, _ = torch.cuda.mem_get_info()
  for _ in range(20):
    dec = _capi.Decoder(params)
    _decode(dec, seqs, 10, 1, 2)
    _decode(dec, seqs, 5, 2, 1)
    dec.close()
  gc.collect()
  torch.cuda.synchronize()
  free1, _ = torch.cuda.mem_get_info()
  assert free0 - free1 < 64 * 1024 * 1024, (free0, free1)


def test_one_launch_decode_repeats_bit_for_bit():
  """The in-launch hand-offs of k_decode_resident are timing dependent in principle: the same
  batch, decoded 40 times back to back (tools/stress_resident.py runs hundreds), must give the
  launch-per-step path's bits every time."""
  params = synth.tracker_params(256, 512, 1, seed=0)
  lengths = [120 + (37 * u) % 200 for u in range(64)]
  seqs, _ = synth.make_utterances(30_000, 64, lengths, 256)
  dec = _capi.Decoder(params)
  ref, _ = _decode(dec, seqs, 10, 1, 2, flags=_capi.UIS_FLAG_STEPWISE)
  for _ in range(40):
    out, _ = _decode(dec, seqs, 10, 1, 2, flags=_capi.UIS_FLAG_RESIDENT)
    assert np.array_equal(out['labels'], ref['labels'])
    assert np.array_equal(out['beam_scores'].view(np.uint32), ref['beam_scores'].view(np.uint32))


# ------------------------------------------------------------ the multi-rank job, rehearsed on ONE device (round 6)

import json
import os
import socket
import subprocess
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _torchrun_bench(extra, backend='gloo', nproc=2, timeout=900):
  """bench.py under torch.distributed.run exactly as the driver launches it (plus --allow_shared_device)."""
  out = None
  for _ in range(2):   # (a rendezvous can lose a race on a busy box: once more)
    with socket.socket() as sock:
      sock.bind(('127.0.0.1', 0))
      port = sock.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', UIS_BENCH_NO_PMC='1')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'LOCAL_WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
      env.pop(k, None)
    out = subprocess.run(
        [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(nproc),
         '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.join(_ROOT, 'bench.py'),
         '--gpus', str(nproc), '--allow_shared_device', '--backend', backend, '--check_gather',
         '--no_cpu_baseline', '--no_extra_configs'] + list(extra),
        capture_output=True, text=True, timeout=timeout, cwd=_ROOT, env=env)
    if out.returncode == 0:
      break
  return out


def _record(name, text):
  try:
    os.makedirs(os.path.join(_ROOT, 'gpurun_out'), exist_ok=True)
    with open(os.path.join(_ROOT, 'gpurun_out', name), 'w') as f:
      f.write(text)
  except OSError:
    pass


@pytest.mark.parametrize('extra', [['--steps', '3', '--warmup', '1'],
                                   ['--config', '3', '--ragged', '--utterances', '64', '--steps', '2', '--warmup', '1']],
                         ids=['configs1', 'configs3_ragged_64'])
def test_two_ranks_run_the_real_decoder_on_one_device(extra):
  """The driver's multi-GPU launch (torchrun, one rank per GPU, all three legs, final label gather) has only ever run
  with a stand-in decoder or one rank.  Here TWO ranks run the real library on the one device this box has
  (`--allow_shared_device`: a rehearsal, never a scaling number -- the line says `shared_device` and counts devices):
  two processes whose cooperative launches each want every CU, two cast teams on disjoint core sets, two pinned staging
  blocks, the gather over gloo -- and EVERY label of both ranks against the CPU oracle (`--check_gather`)."""
  out = _torchrun_bench(extra)
  tag = 'rehearsal_' + ('c3' if '--config' in extra else 'c1')
  _record(tag + '.log', out.stdout[-6000:] + '\n--- stderr ---\n' + out.stderr[-6000:])
  assert out.returncode == 0, out.stderr[-3000:]
  lines = [l for l in out.stdout.splitlines() if l.startswith('{"metric"')]
  assert len(lines) == 1, out.stdout[-2000:]
  rec = json.loads(lines[0])
  _record(tag + '.json', lines[0] + '\n')
  assert rec['shared_device'] is True and rec['n_gpus'] == 1 and rec['ranks'] == 2
  assert rec['gather_check']['identical'] is True, rec['gather_check']
  assert rec['gather_check']['ranks'] == 2 and rec['gather_check']['utterances'] == 128
  assert rec['value_predict_f64'] > 0 and rec['value_host_buffers'] > 0 and rec['value_device'] > 0
  assert rec['rank_host_cores'] >= 1
  assert rec['per_rank_ms']['max'] <= rec['ms_per_step'] * 1.001


def test_two_ranks_on_one_device_over_rccl_or_its_refusal_on_record():
  """The same rehearsal over the `nccl` backend (= RCCL).  RCCL may refuse two ranks on one device ("Duplicate GPU
  detected"): then the message goes on record (gpurun_out/rehearsal_nccl.log) and the test passes -- what must not
  happen is a hang or a wrong label."""
  try:
    out = _torchrun_bench(['--steps', '2', '--warmup', '1', '--no_host_buffers'], backend='nccl', timeout=600)
  except subprocess.TimeoutExpired as e:
    _record('rehearsal_nccl.log', 'TIMEOUT after {} s\n'.format(e.timeout))
    pytest.fail('two ranks on one device over RCCL hung')
  _record('rehearsal_nccl.log', 'rc={}\n'.format(out.returncode) + out.stdout[-4000:] + '\n--- stderr ---\n' + out.stderr[-8000:])
  if out.returncode == 0:
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith('{"metric"')][-1])
    assert rec['gather_check']['identical'] is True and rec['shared_device'] is True
  else:
    assert 'Duplicate GPU' in out.stderr or 'NCCL' in out.stderr or 'nccl' in out.stderr, out.stderr[-2000:]


def test_library_first_then_torch_finds_the_gpu():
  """Load order (round 6): this library first, PyTorch second used to leave torch without a GPU (two HIP runtimes in one
  process).  _capi.share_hip_runtime_with_torch() makes both orders end with ONE runtime: decode, THEN import torch and
  use the device from it, then decode again on torch's current stream's device."""
  code = (
      'import sys; sys.path.insert(0, {root!r})\n'
      'import numpy as np\n'
      'from uisrnn_amd import _capi, synth\n'
      'assert "torch" not in sys.modules\n'
      'params = synth.tracker_params(256, 512, 1, seed=0)\n'
      'seqs, _ = synth.make_utterances(500, 2, [30, 20], 256)\n'
      'dec = _capi.Decoder(params)\n'
      'a = dec.decode_f64(seqs, 10, 1, 2)\n'
      'assert a["status"] == 0\n'
      'import torch\n'
      'torch.cuda.init()\n'
      'assert torch.cuda.device_count() >= 1\n'
      'x = torch.arange(8, device="cuda").sum().item()\n'
      'assert x == 28\n'
      'b = dec.decode_f64(seqs, 10, 1, 2)\n'
      'assert np.array_equal(a["labels"], b["labels"])\n'
      'maps = sorted(set(l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l))\n'
      'assert len(maps) == 1, maps\n'
      'print("ok", _capi._hip_runtime)\n').format(root=_ROOT)
  out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=600, cwd=_ROOT)
  assert out.returncode == 0 and 'ok torch (preloaded' in out.stdout, out.stdout[-1000:] + out.stderr[-3000:]
