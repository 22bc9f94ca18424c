#!/usr/bin/env python3
"""Benchmark of the MI355X UIS-RNN decode path (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W]

One "step" = one pass of the hot path (uisrnn.UISRNN.predict over a list,
uisrnn/uisrnn.py:564-590) over one batch of synthetic utterances:
BASELINE.json configs[1] -- 64 utterances x 500 frames x 256-dim, beam_size=10,
look_ahead=1, test_iteration=2, <= 4 speakers -- per GPU (weak scaling:
utterances are independent, every rank decodes its own 64; the only
collective is the final all_gather of the int32 labels over RCCL).

The frame stream and the label buffer live in HBM before the timed region
starts (torch is used for device memory and torch.distributed only).

Prints ONE JSON line on rank 0 with the contract fields plus
  roofline     -- the dominant kernel timed with HIP events on the decode stream
                  (UIS_FLAG_PROFILE pass): k_decode_resident, the one-launch beam
                  search, where it applies (this workload); k_dense_gru, the
                  hidden-side GRU GEMM, on the launch-per-step path (--flags 128)
  cpu_baseline -- the CPU oracle (oracle/, a port of the reference algorithm)
                  timed on this box's host cores on a bounded sample; the GPU
                  labels are checked against it.
"""

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense f32 MFMA peak
PEAK_HBM_GBS = 8000.0

CONFIG = dict(workload='configs[1]: 64 utt x 500 frames x 256-dim, beam 10, '
                       'look_ahead 1, test_iteration 2, <=4 speakers',
              utterances_per_gpu=64, frames=500, observation_dim=256,
              rnn_hidden_size=512, rnn_depth=1, beam_size=10, look_ahead=1,
              test_iteration=2, model='closed-form tracker (uisrnn_amd.synth)')


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


def reference_rate_note():
  """The reference's own predict() rate, recorded when the fixtures were generated.

  google/uis-rnn cannot run on the GPU box (it is not shipped there); its wall time per
  utterance was stored in tests/golden/tracker_d256.npz by make_golden.py (dev container,
  8 vCPU Xeon, one torch thread, same model family and beam as this benchmark).
  """
  try:
    data = np.load(os.path.join(ROOT, 'tests', 'golden', 'tracker_d256.npz'))
    frames = sum(len(data['run0_labels_{}'.format(u)]) for u in range(int(data['n_utt'])))
    secs = float(np.sum(data['run0_secs']))
    return ('google/uis-rnn predict(): {:.1f} frames/s ({} frames in {:.0f} s, beam 10, D=256, '
            'H=512, 1 thread, dev container; tests/golden/tracker_d256.npz)'.format(
                frames / secs, frames, secs))
  except Exception:  # pylint: disable=broad-except
    return None


def parse():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=10)
  ap.add_argument('--warmup', type=int, default=2)
  ap.add_argument('--utterances', type=int, default=CONFIG['utterances_per_gpu'])
  ap.add_argument('--frames', type=int, default=CONFIG['frames'])
  ap.add_argument('--beam_size', type=int, default=CONFIG['beam_size'])
  ap.add_argument('--no_cpu_baseline', action='store_true')
  ap.add_argument('--cpu_sample', type=int, default=0,
                  help='utterances in the CPU-baseline sample (0 = auto)')
  ap.add_argument('--flags', type=int, default=0, help='UIS_FLAG_* for the timed run')
  ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'],
                  help='collective backend for N > 1 (nccl = RCCL over xGMI; gloo only to '
                       'exercise the multi-rank path on a box with fewer GPUs than ranks)')
  ap.add_argument('--force_dist', action='store_true',
                  help='initialise the process group and run the gather even with one rank '
                       '(exercises the RCCL calls on a single-GPU box)')
  ap.add_argument('--streams', type=int, default=0,
                  help='utterance groups decoded concurrently (0 = library default)')
  return ap.parse_args()


def main():
  args = parse()
  import torch  # device memory + torch.distributed only
  from uisrnn_amd import _capi, synth

  world = int(os.environ.get('WORLD_SIZE', '1'))
  rank = int(os.environ.get('RANK', '0'))
  local_rank = int(os.environ.get('LOCAL_RANK', '0'))
  n_dev = torch.cuda.device_count()
  if n_dev < 1:
    raise RuntimeError('bench.py needs an MI355X: no HIP device is visible')
  # one GPU per rank; if the launcher narrowed visibility to one device per process the
  # modulo maps every rank to its only device
  dev_index = local_rank % n_dev
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

  dim, hid = CONFIG['observation_dim'], CONFIG['rnn_hidden_size']
  n_utt, n_frames = args.utterances, args.frames
  params = synth.tracker_params(dim, hid, CONFIG['rnn_depth'], seed=0)
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
  beam, look, tau = args.beam_size, CONFIG['look_ahead'], CONFIG['test_iteration']

  def one_step(flags):
    out = decoder.decode_device(d_frames.data_ptr(), offsets, beam, look, tau,
                                d_labels.data_ptr(), d_scores.data_ptr(),
                                max_clusters=16, flags=flags,
                                n_streams=args.streams)
    if out['status'] != 0:
      raise RuntimeError('decode hit the cluster cap in the benchmark workload')
    if use_dist:  # the final gather: the only collective of the path (RCCL over xGMI)
      dist.all_gather_into_tensor(
          gathered, d_labels if args.backend == 'nccl' else d_labels.cpu())
    return out

  for _ in range(args.warmup):
    one_step(args.flags)
  if use_dist:
    dist.barrier()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  last = None
  for _ in range(args.steps):
    last = one_step(args.flags)
  torch.cuda.synchronize()
  if use_dist:
    dist.barrier()
  elapsed = time.perf_counter() - t0
  if use_dist:
    t = torch.tensor([elapsed], dtype=torch.float64, device=gather_dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

  ms_per_step = 1e3 * elapsed / max(args.steps, 1)
  value = world * total_frames * args.steps / elapsed
  stats = last['stats'] if last else {}

  result = None
  if rank == 0:
    # ---- roofline of the dominant kernel: HIP events around every launch
    prof = decoder.decode_device(d_frames.data_ptr(), offsets, beam, look, tau,
                                 d_labels.data_ptr(), d_scores.data_ptr(),
                                 max_clusters=16,
                                 flags=args.flags | _capi.UIS_FLAG_PROFILE)['stats']
    n_steps = prof['n_steps']
    gru_ms = prof['kernel_ms']['gru']
    gru_launches = max(prof['kernel_launches']['gru'], 1)
    resident = prof['kernel_launches']['select'] == 0 and prof['kernel_launches']['gru'] == 1
    # start/stop events of hipExtLaunchKernelGGL = the dispatch's own begin/end timestamps
    # (cross-check: rocprofv3 --kernel-trace average in profiles/)
    avg_us = 1e3 * gru_ms / gru_launches
    rows_algo = prof['rnn_rows_nodedup'] / max(n_steps, 1)
    flop_per_frame = tau * (2.0 * (3 * hid * dim + 3 * hid * hid + hid * hid + dim * hid) * beam
                            + 3.0 * dim * beam * 5)
    if resident:
      # ONE launch = the whole beam search.  Algorithmic work of the launch: every surviving
      # hypothesis of every step takes the hidden-side GRU matvec (3H x H), linear_mean1
      # (H x H) and linear_mean2 (D x H); the input-side projection is k_dense_input_proj's.
      kernel = 'k_decode_resident'
      flop_per_launch = 2.0 * (3 * hid * hid + hid * hid + dim * hid) * prof['rnn_rows_nodedup']
      # SURVEY.md 8(d), per decode step and utterance: x_t, the candidates' means, the winners'
      # h / mean in and out, back-pointers = 4D(1 + B*K + 2B) + 8*H*B + 8B with K = 4 clusters;
      # the weights once.  (Served by the XCD's L2 for the most part: `traffic` is what reached HBM.)
      algo_bytes = int(4 * (3 * hid * hid + hid * hid + dim * hid) +
                       n_utt * n_steps * (4 * dim * (1 + beam * 4 + 2 * beam) + 8 * hid * beam + 8 * beam))
    else:
      # one launch = one step's hidden-side GRU matvecs (3H x H MACs per surviving hypothesis)
      kernel = 'k_dense_gru'
      flop_per_launch = 2.0 * 3 * hid * hid * rows_algo
      # W_hh once + per row: h in, gi0 in (3H), h' out
      algo_bytes = int(4 * 3 * hid * hid + rows_algo * 4 * (hid + 3 * hid + hid))
    achieved = flop_per_launch / (avg_us * 1e-6) / 1e12
    roofline = {
        'bound': 'mfma', 'kernel': kernel,
        'decode_path': 'one launch (k_decode_resident)' if resident else 'launch per step',
        'achieved': round(achieved, 3),
        'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
        'frac': round(achieved / PEAK_F32_MFMA_TFLOPS, 4),
        'traffic': committed_traffic(kernel),
        'avg_launch_us': round(avg_us, 3), 'launches': gru_launches,
        'algorithmic_bytes_per_launch': algo_bytes,
        'rows_per_step_algorithmic': round(rows_algo, 1),
        'rows_per_step_executed': round(prof['rnn_rows'] / max(n_steps, 1), 1),
        'path_frac_fp32': round(value / world * flop_per_frame / (PEAK_F32_MFMA_TFLOPS * 1e12), 4),
        # SURVEY.md 8(d) asks for both fractions; HBM is not the binding one (arithmetic intensity ~300 FLOP/B)
        'path_frac_hbm': round(value / world * tau * (4 * dim * (1 + beam * 4 + 2 * beam) + 8 * hid * beam + 8 * beam)
                               / (PEAK_HBM_GBS * 1e9), 4),
        'kernel_ms_profile_pass': {k: round(v, 3) for k, v in prof['kernel_ms'].items()},
    }
    # ---- CPU baseline: the oracle on this box's cores, bounded sample
    cpu = None
    if not args.no_cpu_baseline:
      from oracle import oracle
      cores = os.cpu_count() or 1
      threads = min(cores, 64)
      sample = args.cpu_sample or min(n_utt, max(threads, 1))
      t0 = time.perf_counter()
      ref = oracle.decode(params, seqs[:sample], beam, look, tau, n_threads=threads)
      cpu_s = time.perf_counter() - t0
      got = d_labels.cpu().numpy()
      parity = all(
          np.array_equal(got[u * n_frames:(u + 1) * n_frames], ref['labels'][u])
          for u in range(sample))
      cpu = {'value': round(sample * n_frames / cpu_s, 2), 'unit': 'frames/s',
             'cores': threads, 'kind': 'port',
             'reference_python': reference_rate_note(),
             'sample': '{} of the {} utterances, {} threads, {:.1f}s; GPU labels '
                       'identical: {}'.format(sample, n_utt, threads, cpu_s, parity)}
    result = {
        'metric': 'diarization frames/sec (whole node), beam=10, 256-dim',
        'value': round(value, 1), 'unit': 'frames/s', 'n_gpus': world,
        'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(ms_per_step, 3), 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32',
        'data': 'synthetic', 'config': dict(CONFIG, utterances_per_gpu=n_utt,
                                            frames=n_frames, beam_size=beam,
                                            parallelism='utterance-sharded x{}'.format(world)),
        'decode_ms_device': round(stats.get('decode_ms', 0.0), 3),
        'n_streams': stats.get('n_streams', 0),
        'roofline': roofline, 'cpu_baseline': cpu,
    }
    print(json.dumps(result), flush=True)
  if use_dist:
    dist.barrier()
    dist.destroy_process_group()
  return result


if __name__ == '__main__':
  main()
