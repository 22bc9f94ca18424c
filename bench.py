#!/usr/bin/env python3
"""Benchmark of the MI355X UIS-RNN decode path (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config 1|2|3|4]

One "step" = one pass of the hot path (uisrnn.UISRNN.predict over a list,
uisrnn/uisrnn.py:564-590) over one batch of synthetic utterances.  --config picks
the BASELINE.json `configs[i]` workload (per GPU):

  1 (default)  64 utterances x 500 frames x 256-dim, beam 10, look_ahead 1   <- the metric's config
  2            beam 50, look_ahead 2, 64 utterances x 1000 frames (wide-beam stress)
  3            the per-GPU share of configs[3]: 1024 utterances x 1000 frames, beam 10
  4            observation_dim 512, rnn_hidden_size 512, beam 20, 64 x 500 frames

Multi-GPU (the reference's analogue is parallel_predict,
uisrnn/uisrnn.py:593-623): utterances are independent, so every rank decodes its own
utterances (weak scaling) and the only collective is the final all_gather of the int32
labels (RCCL over xGMI).  `--gpus N` with N > 1 and no WORLD_SIZE in the environment
re-executes itself under `python -m torch.distributed.run --nproc-per-node N`; under
torchrun (the driver's launch) it reads RANK / LOCAL_RANK / WORLD_SIZE.  It refuses to
run when fewer than N HIP devices are visible.

The frame stream and the label buffer live in HBM before the timed region starts
(torch is used for device memory and torch.distributed only).  `value` is that
HBM-resident rate; `value_host_buffers` is the same pass through uis_decode from
pinned host memory (H2D of the frames and D2H of the labels inside the clock,
SURVEY.md 8d).

Prints ONE JSON line on rank 0 with the contract fields plus
  roofline     -- the dominant kernel timed with HIP events on the decode stream
  cpu_baseline -- the CPU oracle (oracle/, a port of the reference algorithm) timed on
                  this box's host cores on a bounded sample (the GPU labels are checked
                  against it), next to the reference's own measured rates
                  (tests/golden/reference_cpu_rate.json, recorded in the dev container:
                  google/uis-rnn cannot travel to the GPU box).
"""

import argparse
import json
import os
import socket
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense f32 MFMA peak
PEAK_HBM_GBS = 8000.0

CONFIGS = {
    1: dict(workload='configs[1]: 64 utt x 500 frames x 256-dim, beam 10, look_ahead 1, '
                     'test_iteration 2, <=4 speakers',
            utterances_per_gpu=64, frames=500, observation_dim=256, rnn_hidden_size=512,
            rnn_depth=1, beam_size=10, look_ahead=1, test_iteration=2, max_clusters=16),
    2: dict(workload='configs[2]: beam 50, look_ahead 2, 64 utt x 1000 frames x 256-dim, '
                     'test_iteration 2',
            utterances_per_gpu=64, frames=1000, observation_dim=256, rnn_hidden_size=512,
            rnn_depth=1, beam_size=50, look_ahead=2, test_iteration=2, max_clusters=12),
    3: dict(workload='configs[3] per-GPU share: 1024 utt x 1000 frames x 256-dim, beam 10, '
                     'look_ahead 1, test_iteration 2',
            utterances_per_gpu=1024, frames=1000, observation_dim=256, rnn_hidden_size=512,
            rnn_depth=1, beam_size=10, look_ahead=1, test_iteration=2, max_clusters=16),
    4: dict(workload='configs[4]: observation_dim 512, rnn_hidden_size 512, beam 20, '
                     '64 utt x 500 frames, look_ahead 1, test_iteration 2',
            utterances_per_gpu=64, frames=500, observation_dim=512, rnn_hidden_size=512,
            rnn_depth=1, beam_size=20, look_ahead=1, test_iteration=2, max_clusters=11),
}


def flops_per_frame(cfg, clusters=4):
  """algorithmic FLOPs per input frame (test_iteration decode steps each)."""
  dim, hid = cfg['observation_dim'], cfg['rnn_hidden_size']
  beam, look, tau = cfg['beam_size'], cfg['look_ahead'], cfg['test_iteration']
  p = 3 * hid * dim + 3 * hid * hid + hid * hid + dim * hid
  if look == 1:
    per_step = 2.0 * p * beam + 3.0 * dim * beam * (clusters + 1)
  else:
    # per window of `look` frames: the first look-1 sub-steps evaluate CoreRNN for every
    # prefix, the last one for the winners only
    prefixes, total = beam, 0.0
    for _ in range(look - 1):
      prefixes *= clusters + 1
      total += 2.0 * p * prefixes
    per_step = (total + 2.0 * p * beam) / look
  return tau * per_step


def bytes_per_step(cfg, clusters=4):
  """minimum state traffic per decode step and utterance."""
  dim, hid, beam = cfg['observation_dim'], cfg['rnn_hidden_size'], cfg['beam_size']
  return 4 * dim * (1 + beam * clusters + 2 * beam) + 8 * cfg['rnn_depth'] * hid * beam + 8 * beam


def committed_traffic(kernel):
  """HBM bytes per launch of `kernel` from the newest committed PMC run (profiles/), or None.

  PMC counters cannot be read from inside the benchmark process; they are collected with
  rocprofv3 in separate passes (tools/gpu_pmc.sh) and committed.  FETCH_SIZE is doubled
  (gfx950 counts 64 B per 128-B request for wide coalesced reads).
  """
  import glob
  files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_traffic.json')))
  if not files:
    return None
  try:
    entry = json.load(open(files[-1]))['kernels'][kernel]
    return {'bytes_per_launch': int((2.0 * entry['fetch_size_kib'] + entry['write_size_kib']) * 1024),
            'source': os.path.relpath(files[-1], ROOT)}
  except (KeyError, ValueError):
    return None


def reference_rates():
  """The reference's own measured predict() rates (dev container), or None."""
  try:
    with open(os.path.join(ROOT, 'tests', 'golden', 'reference_cpu_rate.json')) as f:
      rec = json.load(f)
    return {'whole_box_frames_per_s': round(rec['whole_box_frames_per_s'], 2),
            'one_process_one_thread_frames_per_s': round(rec['one_process_one_thread_frames_per_s'], 2),
            'cores': rec['cores'], 'box': rec['box'], 'workload': rec['workload'],
            'source': 'tests/golden/reference_cpu_rate.json (google/uis-rnn run by '
                      'tests/golden/make_trained.py wholebox)'}
  except (OSError, KeyError, ValueError):
    return None


def parse(argv=None):
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=None)
  ap.add_argument('--steps', type=int, default=None)
  ap.add_argument('--warmup', type=int, default=None)
  ap.add_argument('--config', type=int, default=1, choices=sorted(CONFIGS))
  ap.add_argument('--utterances', type=int, default=None, help='utterances per GPU')
  ap.add_argument('--frames', type=int, default=None)
  ap.add_argument('--beam_size', type=int, default=None)
  ap.add_argument('--model', default='auto', choices=['auto', 'trained', 'tracker'],
                  help='trained = tests/golden/trained_d{256,512}.uisrnn (the reference\'s fit, '
                       'SURVEY.md 8d); tracker = closed-form weights (uisrnn_amd.synth); '
                       'auto = trained where it applies')
  ap.add_argument('--no_cpu_baseline', action='store_true')
  ap.add_argument('--no_host_buffers', action='store_true',
                  help='skip the PCIe-inclusive pass (value_host_buffers)')
  ap.add_argument('--cpu_sample', type=int, default=0,
                  help='utterances in the CPU-baseline sample (0 = auto)')
  ap.add_argument('--flags', type=int, default=0, help='UIS_FLAG_* for the timed run')
  ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'],
                  help='collective backend for N > 1 (nccl = RCCL over xGMI)')
  ap.add_argument('--force_dist', action='store_true',
                  help='initialise the process group and run the gather even with one rank '
                       '(exercises the RCCL calls on a single-GPU box)')
  ap.add_argument('--streams', type=int, default=0,
                  help='utterance groups decoded concurrently (0 = library default)')
  return ap.parse_args(argv)


def free_port():
  with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
    s.bind(('127.0.0.1', 0))
    return s.getsockname()[1]


def respawn_under_torchrun(n_gpus, argv):
  """--gpus N without a launcher: one rank per GPU through torch.distributed.run."""
  cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
         '--nproc-per-node', str(n_gpus), '--master-addr', '127.0.0.1',
         '--master-port', str(free_port()), os.path.abspath(__file__)] + list(argv)
  os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
  os.execv(sys.executable, cmd)


def timed_region(step_fn, sync_fn, steps, warmup, dist=None, reduce_device=None):
  """W untimed steps, then EXACTLY `steps` steps bracketed by barrier + device sync on both
  sides; returns the MAX over ranks of the elapsed seconds.

  step_fn(): one pass of the hot path (incl. the final label gather when dist is set);
  sync_fn(): wait for this rank's device.  `dist` is torch.distributed or None.
  """
  import torch  # pylint: disable=import-outside-toplevel
  for _ in range(warmup):
    step_fn()
  sync_fn()
  if dist is not None:
    dist.barrier()
  sync_fn()
  t0 = time.perf_counter()
  for _ in range(steps):
    step_fn()
  sync_fn()
  if dist is not None:
    dist.barrier()
  elapsed = time.perf_counter() - t0
  if dist is not None:
    t = torch.tensor([elapsed], dtype=torch.float64, device=reduce_device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
  return elapsed


def load_model(cfg, which):
  """(params, description): a checkpoint trained by the reference where one exists, else closed-form weights."""
  from uisrnn_amd import synth, weights  # pylint: disable=import-outside-toplevel
  dim, hid = cfg['observation_dim'], cfg['rnn_hidden_size']
  path = os.path.join(ROOT, 'tests', 'golden', 'trained_d{}.uisrnn'.format(dim))
  can_train = dim in (256, 512) and hid == 512 and cfg['rnn_depth'] == 1 and os.path.exists(path)
  if which == 'trained' and not can_train:
    raise SystemExit('--model trained needs tests/golden/trained_d{256,512}.uisrnn and a matching config')
  if which in ('auto', 'trained') and can_train:
    return (weights.load_checkpoint(path),
            'tests/golden/trained_d{}.uisrnn: trained by the reference\'s fit(), 300 iterations on synthetic '
            'd-vectors (tests/golden/make_trained.py, SURVEY.md 8d)'.format(dim))
  return (synth.tracker_params(dim, hid, cfg['rnn_depth'], seed=0),
          'closed-form tracker (uisrnn_amd.synth)')


def main(argv=None):
  args = parse(argv)
  env_world = os.environ.get('WORLD_SIZE')
  if env_world is None and args.gpus is not None and args.gpus > 1:
    import torch  # pylint: disable=import-outside-toplevel
    n_dev = torch.cuda.device_count()
    if n_dev < args.gpus:
      raise SystemExit('bench.py --gpus {}: only {} HIP device(s) visible; one rank per GPU is '
                       'required (no folding of ranks onto one device)'.format(args.gpus, n_dev))
    respawn_under_torchrun(args.gpus, sys.argv[1:] if argv is None else argv)
  world = int(env_world or '1')
  if args.gpus is not None and args.gpus != world:
    raise SystemExit('bench.py --gpus {} but WORLD_SIZE={}'.format(args.gpus, world))
  rank = int(os.environ.get('RANK', '0'))
  local_rank = int(os.environ.get('LOCAL_RANK', '0'))

  import torch  # device memory + torch.distributed only
  from uisrnn_amd import _capi, synth

  cfg = dict(CONFIGS[args.config])
  if args.utterances is not None:
    cfg['utterances_per_gpu'] = args.utterances
  if args.frames is not None:
    cfg['frames'] = args.frames
  if args.beam_size is not None:
    cfg['beam_size'] = args.beam_size
  if args.utterances is not None or args.frames is not None or args.beam_size is not None:
    cfg['workload'] += ' [overridden: {} utt x {} frames, beam {}]'.format(
        cfg['utterances_per_gpu'], cfg['frames'], cfg['beam_size'])
  big = cfg['utterances_per_gpu'] * cfg['frames'] > 200_000 or cfg['look_ahead'] > 1
  steps = args.steps if args.steps is not None else (3 if big else 10)
  # five warm-up passes: the decoder tries its control-word placements on passes 2-5 of a shape
  warmup = args.warmup if args.warmup is not None else (5 if cfg['utterances_per_gpu'] * cfg['frames'] <= 2_000_000 else 1)

  n_dev = torch.cuda.device_count()
  if n_dev < 1:
    raise RuntimeError('bench.py needs an MI355X: no HIP device is visible')
  if local_rank >= n_dev:
    raise RuntimeError('rank {} (local rank {}) has no GPU of its own: {} HIP device(s) visible, '
                       'one rank per GPU is required'.format(rank, local_rank, n_dev))
  dev_index = local_rank
  torch.cuda.set_device(dev_index)
  use_dist = world > 1 or args.force_dist
  if use_dist:
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29511')
    if args.backend == 'nccl':
      dist.init_process_group('nccl', rank=rank, world_size=world,
                              device_id=torch.device('cuda', dev_index))
    else:
      dist.init_process_group('gloo', rank=rank, world_size=world)
  else:
    dist = None
  dev = torch.device('cuda', dev_index)

  dim, hid = cfg['observation_dim'], cfg['rnn_hidden_size']
  n_utt, n_frames = cfg['utterances_per_gpu'], cfg['frames']
  beam, look, tau = cfg['beam_size'], cfg['look_ahead'], cfg['test_iteration']
  params, model_note = load_model(cfg, args.model)
  cfg['model'] = model_note
  seqs, _ = synth.make_utterances(10_000 + rank * n_utt, n_utt, n_frames, dim)
  frames = np.concatenate(seqs, axis=0).astype(np.float32)
  offsets = np.arange(n_utt + 1, dtype=np.int64) * n_frames
  total_frames = n_utt * n_frames

  decoder = _capi.Decoder(params, device=dev_index)
  d_frames = torch.from_numpy(frames).to(dev)
  d_labels = torch.empty(total_frames, dtype=torch.int32, device=dev)
  d_scores = torch.empty(n_utt, dtype=torch.float32, device=dev)
  gather_dev = dev if args.backend == 'nccl' else torch.device('cpu')
  gathered = (torch.empty(world * total_frames, dtype=torch.int32, device=gather_dev)
              if use_dist else None)
  state = {'last': None, 'cap': cfg['max_clusters']}

  def decode_once(flags):
    while True:
      out = decoder.decode_device(d_frames.data_ptr(), offsets, beam, look, tau,
                                  d_labels.data_ptr(), d_scores.data_ptr(),
                                  max_clusters=state['cap'], flags=flags,
                                  n_streams=args.streams)
      if out['status'] == 0:
        return out
      # a surviving hypothesis needed more clusters than the tables hold: the Python host's
      # policy (uisrnn_amd/uisrnn.py) is to decode again with twice the room; only warm-up
      # passes may do that -- a timed pass that retried would be counted as failed
      if state.get('timing'):
        raise RuntimeError('decode hit the cluster cap inside the timed region')
      state['cap'] *= 2

  def one_step():
    state['last'] = decode_once(args.flags)
    if use_dist:  # the final gather: the only collective of the path (RCCL over xGMI)
      dist.all_gather_into_tensor(
          gathered, d_labels if args.backend == 'nccl' else d_labels.cpu())

  # set-up passes before anything is timed: the first settles the cluster cap, the next four are
  # the decoder's trials of its control-word placement for this shape, so
  # that they stay outside the clock whatever --warmup says
  for _ in range(5 if total_frames <= 2_000_000 else 1):
    decode_once(args.flags)
  state['timing'] = True
  elapsed = timed_region(one_step, torch.cuda.synchronize, steps, warmup, dist, gather_dev)
  ms_per_step = 1e3 * elapsed / max(steps, 1)
  value = world * total_frames * steps / elapsed
  stats = state['last']['stats'] if state['last'] else {}

  result = None
  if rank == 0:
    # ---- PCIe-inclusive rate: pinned host frames in, labels out, same passes
    host_rate = None
    if not args.no_host_buffers:
      pin_frames = torch.from_numpy(frames).pin_memory()
      pin_labels = torch.empty(total_frames, dtype=torch.int32).pin_memory()
      pin_scores = torch.empty(n_utt, dtype=torch.float32).pin_memory()
      def host_step():
        rc = decoder.decode_host(pin_frames.data_ptr(), offsets, beam, look, tau,
                                 pin_labels.data_ptr(), pin_scores.data_ptr(),
                                 max_clusters=state['cap'], flags=args.flags)
        if rc['status'] != 0:
          raise RuntimeError('host-buffer decode hit the cluster cap')
      el = timed_region(host_step, torch.cuda.synchronize, max(steps // 2, 1), 1)
      host_rate = total_frames * max(steps // 2, 1) / el
      if not np.array_equal(pin_labels.numpy(), d_labels.cpu().numpy()):
        raise RuntimeError('host-buffer decode and device-buffer decode disagree')

    # ---- roofline of the dominant kernel: HIP events around every launch
    prof = decoder.decode_device(d_frames.data_ptr(), offsets, beam, look, tau,
                                 d_labels.data_ptr(), d_scores.data_ptr(),
                                 max_clusters=state['cap'],
                                 flags=args.flags | _capi.UIS_FLAG_PROFILE)['stats']
    n_steps = prof['n_steps']
    resident = prof['kernel_launches']['select'] == 0 and prof['kernel_launches']['gru'] == 1
    rows_algo = prof['rnn_rows_nodedup'] / max(n_steps, 1)
    rows_exec = prof['rnn_rows'] / max(n_steps, 1)
    fpf = flops_per_frame(cfg)
    ceiling = PEAK_F32_MFMA_TFLOPS * 1e12 / fpf
    bps = bytes_per_step(cfg)
    if resident:
      # ONE launch = the whole beam search.  Algorithmic work of the launch: every surviving
      # hypothesis of every step takes the hidden-side GRU matvec (3H x H), linear_mean1
      # (H x H) and linear_mean2 (D x H); the input-side projection is k_dense_input_proj's.
      # (more utterances than workgroups -- 32 per XCD-sized cluster of CUs -- run the variant whose
      # dense stages give a wave a whole row tile, unless flag 0x200 keeps the split-K passes)
      n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
      kernel = 'k_decode_big' if n_utt > n_cu - n_cu % 32 and not args.flags & 0x200 else 'k_decode_resident'
      kclass = 'gru'
      per_row = 2.0 * (3 * hid * hid + hid * hid + dim * hid)
      flop_per_launch = per_row * prof['rnn_rows_nodedup']
      flop_exec = per_row * prof['rnn_rows']
      algo_bytes = int(4 * (3 * hid * hid + hid * hid + dim * hid) + n_utt * n_steps * bps)
    else:
      # the class that takes the most device time; for the GRU GEMM one launch = one step's
      # hidden-side matvecs (3H x H MACs per surviving hypothesis / prefix)
      kclass = max(('gru', 'head1', 'head2', 'select', 'expand', 'upper_in'),
                   key=lambda k: prof['kernel_ms'][k])
      # which family of dense kernels the library picks (uis_decoder.hip, launch_rnn): a wave per row
      # tile above a row capacity of 1280 (weight slice in LDS, hidden size 256 / 512) or 2048 (others)
      level = beam
      for j in range(1, look):
        level = min(level * (state['cap'] + j), 32768)
      cap_rows = 0 if args.flags & 0x200 else n_utt * (beam if look == 1 else level)
      fam = 'k_wt' if hid in (256, 512) and cap_rows > 1280 else ('k_big' if cap_rows > 2048 else 'k_dense')
      kernel = {'gru': fam + '_gru', 'head1': fam + '_head1' if fam != 'k_wt' else 'k_wt_head<1>',
                'head2': fam + '_head2' if fam != 'k_wt' else 'k_wt_head<2>',
                'select': 'k_select_fast', 'expand': 'k_window', 'upper_in': 'k_dense_upper_in'}[kclass]
      launches_gru = max(prof['kernel_launches']['gru'], 1)
      rows_per_launch = prof['rnn_rows_nodedup'] / launches_gru
      per_row = {'gru': 2.0 * 3 * hid * hid, 'head1': 2.0 * hid * hid, 'head2': 2.0 * dim * hid}.get(kclass, 0.0)
      flop_per_launch = per_row * rows_per_launch
      flop_exec = per_row * prof['rnn_rows'] / launches_gru
      algo_bytes = int(4 * 3 * hid * hid + rows_per_launch * 4 * (hid + 3 * hid + hid))
    k_launches = max(prof['kernel_launches'][kclass], 1)
    avg_us = 1e3 * prof['kernel_ms'][kclass] / k_launches
    achieved = flop_per_launch / (avg_us * 1e-6) / 1e12 if flop_per_launch else 0.0
    executed = flop_exec / (avg_us * 1e-6) / 1e12 if flop_exec else 0.0
    roofline = {
        'bound': 'mfma', 'kernel': kernel,
        'decode_path': 'one launch ({})'.format(kernel) if resident else 'launch per step',
        'achieved': round(achieved, 3),
        'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
        'frac': round(achieved / PEAK_F32_MFMA_TFLOPS, 4),
        'traffic': committed_traffic(kernel),
        'avg_launch_us': round(avg_us, 3), 'launches': k_launches,
        'algorithmic_bytes_per_launch': algo_bytes,
        # `achieved` counts the algorithmic rows (one CoreRNN step per surviving hypothesis,
        # SURVEY.md 8d); row de-duplication (8 f1) executes fewer -- what the MFMA pipes ran:
        'executed': {'tflops': round(executed, 3), 'frac': round(executed / PEAK_F32_MFMA_TFLOPS, 4)},
        'rows_per_step_algorithmic': round(rows_algo, 1),
        'rows_per_step_executed': round(rows_exec, 1),
        'ceiling_frames_per_s_fp32': round(ceiling, 0),
        'path_frac_fp32': round(value / world / ceiling, 4),
        # SURVEY.md 8(d) asks for both fractions; HBM is not the binding one (~300 FLOP/B)
        'path_frac_hbm': round(value / world * tau * bps / (PEAK_HBM_GBS * 1e9), 4),
        'kernel_ms_profile_pass': {k: round(v, 3) for k, v in prof['kernel_ms'].items()},
    }
    # ---- CPU baseline: the oracle on this box's cores, bounded sample
    cpu = None
    if not args.no_cpu_baseline and world == 1:  # (N > 1: the other ranks would wait at the barrier)
      from oracle import oracle
      cores = os.cpu_count() or 1
      threads = min(cores, 64)
      sample = args.cpu_sample or min(n_utt, max(threads, 1))
      if look > 1:
        sample = min(sample, 8)
      sample_frames = n_frames if look == 1 else min(n_frames, 100)
      sample_seqs = [s[:sample_frames] for s in seqs[:sample]]
      t0 = time.perf_counter()
      ref = oracle.decode(params, sample_seqs, beam, look, tau, n_threads=threads)
      cpu_s = time.perf_counter() - t0
      if sample_frames == n_frames:
        got = d_labels.cpu().numpy()
        parity = all(
            np.array_equal(got[u * n_frames:(u + 1) * n_frames], ref['labels'][u])
            for u in range(sample))
      else:  # truncated utterances: decode exactly those on the GPU for the comparison
        chk = decoder.decode(np.concatenate(sample_seqs).astype(np.float32),
                             np.arange(sample + 1, dtype=np.int64) * sample_frames,
                             beam, look, tau, max_clusters=state['cap'])
        parity = all(
            np.array_equal(chk['labels'][u * sample_frames:(u + 1) * sample_frames], ref['labels'][u])
            for u in range(sample))
      cpu = {'value': round(sample * sample_frames / cpu_s, 2), 'unit': 'frames/s',
             'cores': threads, 'kind': 'port',
             'reference': reference_rates(),
             'sample': '{} of the {} utterances ({} frames each), {} threads, {:.1f}s; GPU labels '
                       'identical: {}'.format(sample, n_utt, sample_frames, threads, cpu_s, parity)}
    result = {
        'metric': 'diarization frames/sec (whole node), beam=10, 256-dim',
        'value': round(value, 1), 'unit': 'frames/s', 'n_gpus': world,
        'steps': steps, 'warmup': warmup,
        'ms_per_step': round(ms_per_step, 3), 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32',
        'data': 'synthetic',
        'config': dict(cfg, parallelism='utterance-sharded x{}'.format(world),
                       max_clusters=state['cap']),
        'value_host_buffers': round(host_rate, 1) if host_rate else None,
        'decode_ms_device': round(stats.get('decode_ms', 0.0), 3),
        'n_streams': stats.get('n_streams', 0),
        'roofline': roofline, 'cpu_baseline': cpu,
    }
  # the JSON line is the last thing on stdout: whatever native libraries buffered there (RCCL's
  # banner goes through C stdio) is flushed by every rank before rank 0 prints
  sys.stdout.flush()
  try:
    import ctypes  # pylint: disable=import-outside-toplevel
    ctypes.CDLL(None).fflush(None)
  except OSError:
    pass
  if use_dist:
    dist.barrier()
  if result is not None:
    print(json.dumps(result), flush=True)
  if use_dist:
    dist.destroy_process_group()
  return result


if __name__ == '__main__':
  main()
