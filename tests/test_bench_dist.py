"""bench.py's multi-rank plumbing on CPU: the timed region (barrier + sync on both sides, MAX
over ranks), the final label gather and the self-spawn under torch.distributed.run -- world
size 2 over gloo.  The decode itself needs a GPU; here a stand-in step function takes its place."""

import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_RANK_SCRIPT = textwrap.dedent('''
    import json, os, sys, time
    sys.path.insert(0, {root!r})
    import torch, torch.distributed as dist
    import bench
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    dist.init_process_group('gloo', rank=rank, world_size=world)
    labels = torch.full((5,), rank, dtype=torch.int32)
    gathered = torch.empty(world * 5, dtype=torch.int32)
    calls = []
    def step():
      time.sleep(0.02 * (rank + 1))          # rank 1 is the slow one
      dist.all_gather_into_tensor(gathered, labels)
      calls.append(1)
    elapsed = bench.timed_region(step, lambda: None, steps=3, warmup=2, dist=dist,
                                 reduce_device=torch.device('cpu'))
    assert len(calls) == 5
    assert gathered.view(world, 5)[1].tolist() == [1] * 5
    sys.stdout.write(json.dumps({{'rank': rank, 'elapsed': elapsed}}) + chr(10))  # one write per record
    sys.stdout.flush()
    dist.barrier()
    dist.destroy_process_group()
''')


def test_timed_region_world_2_gloo(tmp_path):
  script = tmp_path / 'rank.py'
  script.write_text(_RANK_SCRIPT.format(root=ROOT))
  import socket
  out = None
  for attempt in range(3):   # (a rendezvous can fail for reasons outside the code under test)
    with socket.socket() as sock:
      sock.bind(('127.0.0.1', 0))
      port = sock.getsockname()[1]
    out = subprocess.run(
        [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
         '--master-addr', '127.0.0.1', '--master-port', str(port), str(script)],
        capture_output=True, text=True, timeout=300, cwd=ROOT)
    if out.returncode == 0:   # (anything else: the rendezvous lost a race or timed out on a busy box -- once more)
      break
  assert out.returncode == 0, out.stderr[-3000:]
  import json
  import re
  # (the two ranks share the pipe: records may end up on one line)
  recs = [json.loads(r) for r in re.findall(r'\{[^{}]*\}', out.stdout)]
  assert sorted(r['rank'] for r in recs) == [0, 1]
  # every rank reports the SAME number: the slowest rank's (3 steps x 40 ms)
  assert abs(recs[0]['elapsed'] - recs[1]['elapsed']) < 1e-9
  assert recs[0]['elapsed'] >= 0.11


_MAIN_SCRIPT = textwrap.dedent('''
    import ctypes, json, os, sys
    sys.path.insert(0, {root!r})
    import numpy as np
    from uisrnn_amd import _capi

    class StandInDecoder:
      """What bench.main needs from _capi.Decoder, without a GPU: labels = the rank."""
      def __init__(self, params, device=0):
        self.calls = 0
      def decode_device(self, d_frames_ptr, offsets, beam_size, look_ahead, test_iteration,
                        d_labels_ptr, d_scores_ptr, max_clusters=0, flags=0, n_streams=0):
        n = int(offsets[-1])
        lab = np.ctypeslib.as_array(ctypes.cast(ctypes.c_void_p(int(d_labels_ptr)),
                                                ctypes.POINTER(ctypes.c_int32)), shape=(max(n, 1),))
        lab[:n] = int(os.environ['RANK']) + 7
        self.calls += 1
        steps = int(test_iteration) * int(max(np.diff(offsets))) if len(offsets) > 1 else 0
        rows = steps * (len(offsets) - 1) * 3
        names = _capi.KERNEL_NAMES
        return {{'status': 0, 'stats': {{'n_steps': steps, 'decode_ms': 1.0, 'n_streams': 1,
                 'rnn_rows': rows, 'rnn_rows_nodedup': 2 * rows, 'decode_kernel': 'k_decode_rs',
                 'kernel_ms': {{k: (0.5 if k == 'gru' else 0.0) for k in names}},
                 'kernel_launches': {{k: (1 if k == 'gru' else 0) for k in names}}}}}}
      def decode_host(self, frames_ptr, offsets, beam_size, look_ahead, test_iteration, labels_ptr,
                      scores_ptr, max_clusters=0, flags=0):
        return self.decode_device(frames_ptr, offsets, beam_size, look_ahead, test_iteration, labels_ptr,
                                  scores_ptr, max_clusters, flags)
      def decode_f64(self, seqs, beam_size, look_ahead, test_iteration, max_clusters=0, flags=0, level_cap=0):
        assert all(s.dtype == np.float64 for s in seqs)
        n = sum(s.shape[0] for s in seqs)
        self.calls += 1
        return {{'status': 0, 'labels': np.full(n, int(os.environ['RANK']) + 7, dtype=np.int32)}}
    _capi.Decoder = StandInDecoder
    import bench
    res = bench.main(['--gpus', '2', '--device', 'cpu', '--backend', 'gloo', '--utterances', '3',
                      '--frames', '24', '--steps', '2', '--warmup', '1', '--no_cpu_baseline', '--ragged',
                      '--model', 'tracker'] + {extra!r})
    if int(os.environ['RANK']) == 0:
      assert res is not None
    else:
      assert res is None
''')


import pytest


@pytest.mark.parametrize('extra', [[], ['--config', '3'], ['--timed', 'device']],
                         ids=['config1_predict_f64', 'config3_ragged', 'timed_device'])
def test_bench_main_world_2_gloo_with_a_stand_in_decoder(tmp_path, extra):
  """bench.main() itself, two ranks over gloo: launch plumbing, per-rank core pinning, ragged
  utterances dealt by shard_utterances, the padded label gather, the timed region over the
  float64-list leg on EVERY rank (the multi-GPU line is PCIe-inclusive), the other two legs, ONE
  JSON line from rank 0 with the per-rank spread -- for configs[1] and for the configs[3] share
  (`--config 3 --ragged`, the driver's 8-GPU workload).  (The decode is a stand-in: no GPU here.)"""
  script = tmp_path / 'main_rank.py'
  script.write_text(_MAIN_SCRIPT.format(root=ROOT, extra=extra))
  import json
  import socket
  out = None
  for attempt in range(3):
    with socket.socket() as sock:
      sock.bind(('127.0.0.1', 0))
      port = sock.getsockname()[1]
    out = subprocess.run(
        [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
         '--master-addr', '127.0.0.1', '--master-port', str(port), str(script)],
        capture_output=True, text=True, timeout=600, cwd=ROOT)
    if out.returncode == 0:
      break
  assert out.returncode == 0, out.stderr[-3000:]
  lines = [l for l in out.stdout.splitlines() if l.startswith('{"metric"')]
  assert len(lines) == 1, out.stdout[-2000:]
  rec = json.loads(lines[0])
  assert rec['n_gpus'] == 2 and rec['steps'] == 2 and rec['warmup'] == 1
  assert rec['scaling'] == 'weak' and rec['unit'] == 'frames/s'
  assert rec['per_rank_ms']['min'] <= rec['per_rank_ms']['max'] <= rec['ms_per_step'] * 1.5
  assert rec['setup_passes'] >= 1 and rec['setup_ms'] >= 0.0
  assert 0.0 <= rec['roofline']['frac'] <= 1.0
  assert rec['roofline']['effective']['frac'] >= rec['roofline']['frac']
  assert 'ragged' in rec['config']['workload']
  # whole-job frames: 6 utterances of 12..24 frames, 2 timed steps
  assert 6 * 12 * 2 / (rec['ms_per_step'] * 2e-3) <= rec['value'] <= 6 * 24 * 2 / (rec['ms_per_step'] * 2e-3) * 1.01
  # `value` is the float64-list leg unless --timed says otherwise; all three rates are there
  want = 'device' if '--timed' in extra else 'predict_f64'
  assert rec['value_leg'] == want and rec['value'] == rec['value_' + want]
  assert rec['value_predict_f64'] > 0 and rec['value_host_buffers'] > 0 and rec['value_device'] > 0
  assert rec['roofline']['kernel'] == 'k_decode_rs'   # named by the (stand-in) library, not re-derived


def test_gpus_flag_refuses_to_fold_ranks_onto_one_device():
  """`bench.py --gpus 2` on a box with fewer than 2 GPUs must fail loudly, not run one rank."""
  env = dict(os.environ)
  for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
    env.pop(k, None)
  out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2'],
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
  import torch
  if torch.cuda.device_count() < 2:
    assert out.returncode != 0
    assert 'HIP device(s) visible' in (out.stderr + out.stdout)


def test_gpus_flag_must_agree_with_the_launcher():
  env = dict(os.environ, WORLD_SIZE='2', RANK='0', LOCAL_RANK='0')
  out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '4'],
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
  assert out.returncode != 0 and 'WORLD_SIZE=2' in (out.stderr + out.stdout)


def test_flop_model_matches_the_survey():
  """62.9 MFLOP per input frame for configs[1], 167.8 for configs[4]."""
  sys.path.insert(0, ROOT)
  import bench
  assert abs(bench.flops_per_frame(bench.CONFIGS[1]) / 1e6 - 62.9) < 0.2
  assert abs(bench.flops_per_frame(bench.CONFIGS[4]) / 1e6 - 167.8) < 0.5
  assert abs(bench.flops_per_frame(bench.CONFIGS[2]) / 1e6 - 944) < 10
  assert bench.bytes_per_step(bench.CONFIGS[1]) == 103504
  # --rnn_depth (not a BASELINE config): every further GRU layer adds its input-side and hidden-side gates,
  # 2 x 3 H^2 MACs per hypothesis and step each
  deep = dict(bench.CONFIGS[1], rnn_depth=2)
  hid, beam, tau = deep['rnn_hidden_size'], deep['beam_size'], deep['test_iteration']
  extra = tau * 2.0 * (6 * hid * hid) * beam
  assert abs(bench.flops_per_frame(deep) - bench.flops_per_frame(bench.CONFIGS[1]) - extra) < 1.0
  args = bench.parse(['--rnn_depth', '2', '--utterances', '3'])
  assert args.rnn_depth == 2 and args.utterances == 3


_FAKE_PROFILER = textwrap.dedent('''\
    #!{python}
    # a stand-in for rocprofv3: writes the counter file the real one writes and runs nothing
    import json, os, sys, time
    args = sys.argv[1:]
    out_dir = args[args.index('-d') + 1]
    counter = args[args.index('--pmc') + 1]
    cmd = args[args.index('--') + 1:]
    assert '--kernel-trace' in args and os.environ.get('UIS_BENCH_CHILD') == '1'
    assert cmd[-9:] == ['--timed', 'device', '--steps', '1', '--warmup', '0', '--no_cpu_baseline', '--no_host_buffers', '--no_extra_configs'], cmd
    assert '--utterances' in cmd and '--gpus' not in cmd
    if os.environ.get('FAKE_PROFILER_HANGS'):
      time.sleep(60)
    os.makedirs(os.path.join(out_dir, 'host', '123'))
    with open(os.path.join(out_dir, 'host', '123', 'p_counter_collection.csv'), 'w') as f:
      f.write('"Dispatch_Id","Kernel_Name","Counter_Name","Counter_Value"\\n')
      values = {{'FETCH_SIZE': [1000.0, 1002.0, 5000.0], 'WRITE_SIZE': [300.0, 301.0, 302.0]}}[counter]
      for k, v in enumerate(values):
        f.write('%d,"void k_decode_rs<512, 256, 3, 1, 10, 16, false>(DevModel, DecodeState)","%s",%r\\n' % (k, counter, v))
      f.write('9,"k_backtrace(DecodeState, int*, float*, float*)","%s",77.0\\n' % counter)
      f.write('10,"void k_decode_rs<512, 256, 3, 1, 10, 16, false>(DevModel, DecodeState)","GRBM_GUI_ACTIVE",5.0\\n')
    print('noise')
    print(json.dumps({{'roofline': {{'kernel': 'k_decode_rs'}}}}))
    sys.exit(139)   # (the profiled interpreter is seen to die in its exit handlers: the rows decide, not the status)
''')


def test_traffic_sub_runs_read_the_counter_files(tmp_path, monkeypatch):
  """`roofline.traffic` as bench.py measures it: the sub-run's command line, the dominant kernel's rows of the counter
  file (median over its dispatches, FETCH_SIZE doubled, KiB), a sub-run that hangs (its whole process group goes, the
  committed record takes over) -- with a stand-in for rocprofv3."""
  sys.path.insert(0, ROOT)
  import bench
  fake = tmp_path / 'rocprofv3'
  fake.write_text(_FAKE_PROFILER.format(python=sys.executable))
  fake.chmod(0o755)
  monkeypatch.setenv('PATH', str(tmp_path) + os.pathsep + os.environ['PATH'])
  monkeypatch.delenv('UIS_BENCH_CHILD', raising=False)
  monkeypatch.delenv('UIS_BENCH_NO_PMC', raising=False)
  rec = bench.measured_traffic(['--gpus', '1', '--steps', '7', '--utterances', '64', '--no_cpu_baseline'], timeout_s=30)
  assert rec['kernel'] == 'k_decode_rs'
  assert rec['fetch_size_kib'] == 1002.0 and rec['write_size_kib'] == 301.0
  assert rec['bytes_per_launch'] == int((2 * 1002.0 + 301.0) * 1024)
  assert rec['bytes_per_launch_raw'] == int((1002.0 + 301.0) * 1024)   # (the counters untouched, beside the guide's corrected figure)
  assert rec['min_max_dispatches'] == {'FETCH_SIZE': [1000.0, 5000.0, 3], 'WRITE_SIZE': [300.0, 302.0, 3]}
  monkeypatch.setenv('FAKE_PROFILER_HANGS', '1')
  import time
  t0 = time.time()
  assert bench.measured_traffic(['--utterances', '64'], timeout_s=2) is None
  assert time.time() - t0 < 20
  # ... and a sub-run never starts sub-runs of its own
  monkeypatch.setenv('UIS_BENCH_CHILD', '1')
  assert bench.measured_traffic(['--utterances', '64']) is None


def test_host_memory_of_eight_ranks_of_the_configs3_share():
  """The driver's 8-GPU run: 8 processes x (float64 list + packed float32 + pinned float32 + the library's pinned
  staging block) of the configs[3] share must fit a node's host memory with room to spare (the bound bench.py reports
  as `per_rank_host_gb`)."""
  sys.path.insert(0, ROOT)
  import bench
  per_rank = bench.host_bytes_per_rank(bench.CONFIGS[3])
  frames = 1024 * 1000
  assert per_rank >= frames * 256 * (8 + 4 + 4 + 4)
  assert 8 * per_rank < 48e9, per_rank          # 8 x 5.3 GB
  assert 8 * bench.host_bytes_per_rank(bench.CONFIGS[1]) < 2e9


def test_a_rehearsal_on_a_shared_device_can_never_pass_for_a_scaling_line():
  """`--allow_shared_device` is parsed, is dropped from the counter sub-runs' command, and without it `--gpus 2` on a
  one-device box still refuses (test_gpus_flag_refuses_to_fold_ranks_onto_one_device)."""
  sys.path.insert(0, ROOT)
  import bench
  args = bench.parse(['--gpus', '2', '--allow_shared_device', '--check_gather'])
  assert args.allow_shared_device and args.check_gather
  assert not bench.parse([]).allow_shared_device
  src = open(os.path.join(ROOT, 'bench.py')).read()
  assert "'n_gpus': min(world, n_dev) if shared_device else world" in src and "'shared_device': shared_device" in src


def test_rank_sequences_are_reproducible_from_any_rank():
  """--check_gather: rank 0 regenerates every rank's utterances; ragged lists are ONE job-wide list dealt longest
  first, equal-length ones are seeded by the rank."""
  sys.path.insert(0, ROOT)
  import bench
  import numpy as np
  cfg = dict(bench.CONFIGS[1], utterances_per_gpu=3, frames=20)
  for ragged in (False, True):
    args = bench.parse(['--ragged'] if ragged else [])
    a0, job = bench.rank_sequences(cfg, args, 0, 2)
    a1, job1 = bench.rank_sequences(cfg, args, 1, 2)
    b1, _ = bench.rank_sequences(cfg, args, 1, 2)
    assert job == job1 == sum(s.shape[0] for s in a0 + a1)
    assert all(np.array_equal(x, y) for x, y in zip(a1, b1)) and len(a1) == len(b1)
    assert not np.array_equal(a0[0][:10], a1[0][:10])
