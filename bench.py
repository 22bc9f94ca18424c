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
run when fewer than N HIP devices are visible.  Every rank pins itself to its share of the
host cores (the library's float64 -> float32 cast threads follow the affinity mask).

What the ONE JSON line (rank 0) says, field by field:
  value                 frames/s of the whole job through the entry UISRNN.predict uses
                        (uis_decode_f64: the list of float64 arrays the caller holds; cast + H2D of
                        the frames + decode + D2H of the labels inside the clock) -- the quantity
                        SURVEY.md 8(d) defines, on EVERY rank when N > 1.  `--timed` picks another
                        leg for the timed region (profiling); the line says which (`value_leg`)
  value_predict_f64     the same number under an explicit name
  value_host_buffers    the same passes through uis_decode from PINNED float32 host memory
                        (H2D of the frames and D2H of the labels inside the clock)
  value_device          the same passes with the frame stream and the label buffer resident in
                        HBM when the clock starts (uis_decode_device): no PCIe in the clock
  setup_passes/_ms      untimed decodes before the warm-up (cluster cap, control-word placement)
  per_rank_ms           min / max over ranks of a rank's own time per step
  roofline              the dominant kernel (named by the library: uis_stats.decode_kernel), timed
                        with HIP events on the decode stream; `traffic` = its HBM bytes per launch from
                        rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE sub-runs of this very command (round 5;
                        UIS_BENCH_NO_PMC=1 or no rocprofv3: the newest committed record, marked as such):
                        `frac` = what the MFMA pipes EXECUTED (rows after de-duplication) / peak,
                        never above 1; `effective` = the algorithmic rows (one CoreRNN step per
                        surviving hypothesis) / peak -- de-duplication's credit
  cpu_baseline          the CPU oracle (oracle/, a port of the reference algorithm) timed on this
                        box's host cores on a bounded sample (the GPU labels are checked against
                        it; N > 1: rank 0 runs it after the closing barrier, on its share of the
                        cores); the reference's own measured rates (dev container, google/uis-rnn
                        cannot travel) as reference_* scalars
  extra_configs         configs[2], the configs[3] share and configs[4] at their stated sizes,
                        a few passes each, with frac and a parity check against the oracle
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
  # (every GRU layer: input-side and hidden-side gates; layer 0 reads the observation, the others the layer below)
  p = 3 * hid * dim + 3 * hid * hid + (cfg['rnn_depth'] - 1) * 6 * hid * hid + hid * hid + dim * hid
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


def committed_traffic(kernel, avg_launch_us=None):
  """HBM bytes per launch of `kernel` from the newest committed PMC run (profiles/), or None.

  PMC counters cannot be read from inside the benchmark process; they are collected with
  rocprofv3 in separate passes (tools/gpu_pmc.sh) and committed.  FETCH_SIZE is doubled
  (gfx950 counts 64 B per 128-B request for wide coalesced reads).  The record carries the
  kernel's launch duration at the time it was taken; `stale` says whether today's duration has
  moved more than 10 % away from it (the kernel changed: collect the counters again).
  """
  import glob
  base = kernel.split('<')[0]
  files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_traffic.json')))
  for path in reversed(files):
    try:
      kernels = json.load(open(path))['kernels']
      entry = kernels[kernel] if kernel in kernels else kernels[base]  # (an instantiation's own record first)
      out = {'bytes_per_launch': int((2.0 * entry['fetch_size_kib'] + entry['write_size_kib']) * 1024),
             'bytes_per_launch_raw': int((entry['fetch_size_kib'] + entry['write_size_kib']) * 1024),
             'source': os.path.relpath(path, ROOT)}
      then = entry.get('avg_launch_us')
      if then and avg_launch_us:
        out['avg_launch_us_then'] = then
        out['stale'] = bool(abs(avg_launch_us - then) > 0.10 * then)
      return out
    except (KeyError, ValueError, OSError):
      continue
  return None


def _no_core_dumps():
  import resource  # pylint: disable=import-outside-toplevel
  resource.setrlimit(resource.RLIMIT_CORE, (0, 0))


def measured_traffic(argv, timeout_s=180):
  """HBM bytes per launch of `kernel`, MEASURED for this very run (round 5): two sub-runs of this script under
  `rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes: the two counters do not fit one;
  MI355X_MICROARCH.md, HBM section), one timed pass each of the same workload with the frames resident, the
  counters of the dominant kernel's dispatches averaged.  FETCH_SIZE is doubled (gfx950 tallies a 128-byte
  request of a wide coalesced read at 64 bytes), WRITE_SIZE is taken as reported (uncalibrated), both arrive in
  KiB.  The sub-runs go FIRST, before this process touches the device: run next to a parent that holds a HIP context
  and its buffers the same launch showed 1.7x the fetches and 8x the writes (443 / 417 MB against 254 / 49 MB), so
  they get the GPU to themselves; the kernel they name (their own JSON line's roofline.kernel) comes back with the
  record and the caller attaches it only if its own run names the same kernel.  None when rocprofv3 is not there,
  when this process is itself such a sub-run, or when a pass fails -- the caller then keeps the newest committed
  record and says so."""
  import csv  # pylint: disable=import-outside-toplevel
  import glob  # pylint: disable=import-outside-toplevel
  import shutil  # pylint: disable=import-outside-toplevel
  import subprocess  # pylint: disable=import-outside-toplevel
  import tempfile  # pylint: disable=import-outside-toplevel
  if os.environ.get('UIS_BENCH_CHILD') or os.environ.get('UIS_BENCH_NO_PMC'):
    return None
  prof = shutil.which('rocprofv3') or ('/opt/rocm/bin/rocprofv3' if os.path.exists('/opt/rocm/bin/rocprofv3') else None)
  if not prof:
    print('bench.py: rocprofv3 not found; roofline.traffic falls back to the committed record', file=sys.stderr)
    return None
  keep = []  # the workload's arguments; everything about timing, legs and extras is set here
  skip_next = False
  for a in argv:
    if skip_next:
      skip_next = False
      continue
    if a in ('--steps', '--warmup', '--gpus', '--timed', '--cpu_sample', '--backend'):
      skip_next = True
      continue
    if a in ('--no_cpu_baseline', '--no_host_buffers', '--no_extra_configs', '--force_dist', '--check_gather',
             '--allow_shared_device'):
      continue
    keep.append(a)
  env = dict(os.environ, UIS_BENCH_CHILD='1', TMPDIR='/tmp')
  sizes, spread, kernel, base = {}, {}, None, None
  t0 = time.perf_counter()
  for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
    out_dir = tempfile.mkdtemp(prefix='uis_pmc_', dir='/tmp')
    try:
      cmd = [prof, '--kernel-trace', '--pmc', counter, '-d', out_dir, '-o', 'p', '--output-format', 'csv', '--',
             sys.executable, os.path.abspath(__file__)] + keep + [
                 '--timed', 'device', '--steps', '1', '--warmup', '0', '--no_cpu_baseline', '--no_host_buffers', '--no_extra_configs']
      # (no check of the exit status: under --pmc the profiled interpreter can die in its exit handlers AFTER the counter
      # files are complete -- seen on this image, also from a shell; what decides is whether the kernel's rows are there)
      # (a group of its own: on a timeout the profiler AND the interpreter under it go, so that nothing of a sub-run
      # is still on the device when the timed passes start)
      proc = subprocess.Popen(cmd, cwd='/tmp', env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, preexec_fn=_no_core_dumps,
                              start_new_session=True, text=True)
      try:
        stdout, _ = proc.communicate(timeout=timeout_s)
      except subprocess.TimeoutExpired:
        import signal  # pylint: disable=import-outside-toplevel
        try:
          os.killpg(proc.pid, signal.SIGKILL)
        except OSError:
          pass
        proc.communicate()
        raise
      line = [l for l in stdout.splitlines() if l.startswith('{')]
      kernel = json.loads(line[-1])['roofline']['kernel']   # (what the library says ran, in the sub-run's own line)
      base = kernel.split('<')[0].split(':')[-1]
      vals = []
      for path in glob.glob(os.path.join(out_dir, '**', '*counter_collection.csv'), recursive=True):
        with open(path) as f:
          for row in csv.DictReader(f):
            if row.get('Counter_Name') == counter and base + '<' in row.get('Kernel_Name', '') + '<':
              vals.append(float(row['Counter_Value']))
      if not vals:
        print('bench.py: no {} rows of {} in the counter files under {}'.format(counter, base, out_dir), file=sys.stderr)
        return None
      # (a launch's counter comes in one row per dispatch; the set-up passes run the same kernel: the median of
      # what should be all alike, and the spread for whoever reads the line)
      vals.sort()
      sizes[counter] = vals[len(vals) // 2]
      spread[counter] = [round(vals[0], 1), round(vals[-1], 1), len(vals)]
    except (subprocess.SubprocessError, OSError, ValueError, KeyError, IndexError) as e:
      print('bench.py: the {} sub-run failed ({}: {}); roofline.traffic falls back to the committed record'.format(
          counter, type(e).__name__, str(e)[-300:]), file=sys.stderr)
      return None
    finally:
      shutil.rmtree(out_dir, ignore_errors=True)
  return {'kernel': kernel, 'bytes_per_launch': int((2.0 * sizes['FETCH_SIZE'] + sizes['WRITE_SIZE']) * 1024),
          # (the counters as rocprofv3 reports them, no correction: the lower bound if part of the kernel's reads -- narrow or
          # sc1 loads -- were tallied at their full size; `bytes_per_launch` is the guide's corrected figure)
          'bytes_per_launch_raw': int((sizes['FETCH_SIZE'] + sizes['WRITE_SIZE']) * 1024),
          'fetch_size_kib': round(sizes['FETCH_SIZE'], 1), 'write_size_kib': round(sizes['WRITE_SIZE'], 1),
          'source': 'measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE sub-runs of this script '
                    '(separate passes, device leg, one timed pass), median over the dispatches of ' + base,
          'correction': 'bytes_per_launch = 2 x FETCH_SIZE + WRITE_SIZE (MI355X_MICROARCH.md, HBM: gfx950 tallies 64 B per 128-B request of a '
                        '16-byte-per-lane coalesced read -- this kernel\'s loads; WRITE_SIZE uncalibrated, as reported); '
                        'bytes_per_launch_raw = FETCH_SIZE + WRITE_SIZE untouched; the truth lies between them; KiB per dispatch',
          'min_max_dispatches': spread, 'seconds': round(time.perf_counter() - t0, 1)}


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
  ap.add_argument('--look_ahead', type=int, default=None)
  ap.add_argument('--rnn_hidden_size', type=int, default=None,
                  help='hidden size of the model (not a BASELINE config: the closed-form tracker weights at that size)')
  ap.add_argument('--max_clusters', type=int, default=None, help='cluster cap the decode starts with (it doubles on overflow)')
  ap.add_argument('--rnn_depth', type=int, default=None,
                  help='GRU layers of the model (not a BASELINE config: the closed-form tracker weights at that depth)')
  ap.add_argument('--ragged', action='store_true',
                  help='utterance lengths uniform in [frames / 2, frames]; the whole job\'s utterances are '
                       'dealt to the ranks by uisrnn_amd.distributed.shard_utterances (longest first)')
  ap.add_argument('--model', default='auto', choices=['auto', 'trained', 'tracker'],
                  help='trained = tests/golden/trained_d{256,512}.uisrnn (the reference\'s fit, '
                       'SURVEY.md 8d); tracker = closed-form weights (uisrnn_amd.synth); '
                       'auto = trained where it applies')
  ap.add_argument('--timed', default='predict_f64', choices=['predict_f64', 'host_buffers', 'device'],
                  help='which leg the timed region (and `value`) is: predict_f64 = uis_decode_f64 from the '
                       'float64 arrays predict() receives (default, SURVEY.md 8d); host_buffers = uis_decode '
                       'from pinned float32; device = uis_decode_device, everything resident in HBM')
  ap.add_argument('--no_cpu_baseline', action='store_true')
  ap.add_argument('--no_host_buffers', action='store_true',
                  help='skip the PCIe-inclusive passes (value_host_buffers, value_predict_f64)')
  ap.add_argument('--no_extra_configs', action='store_true',
                  help='skip the short runs of the other BASELINE configs (extra_configs)')
  ap.add_argument('--cpu_sample', type=int, default=0,
                  help='utterances in the CPU-baseline sample (0 = auto)')
  ap.add_argument('--flags', type=int, default=0, help='UIS_FLAG_* for the timed run')
  ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'],
                  help='collective backend for N > 1 (nccl = RCCL over xGMI)')
  ap.add_argument('--device', default='cuda', choices=['cuda', 'cpu'],
                  help='cpu: plumbing tests only (gloo, a stand-in decoder; tests/test_bench_dist.py)')
  ap.add_argument('--force_dist', action='store_true',
                  help='initialise the process group and run the gather even with one rank '
                       '(exercises the RCCL calls on a single-GPU box)')
  ap.add_argument('--streams', type=int, default=0,
                  help='utterance groups decoded concurrently (0 = library default)')
  ap.add_argument('--allow_shared_device', action='store_true',
                  help='REHEARSAL ONLY (tests/test_gpu_scale.py): ranks beyond the visible devices fold onto them '
                       '(rank r -> device r mod #devices), so that the multi-rank job runs the real decoder on a '
                       'one-GPU box.  The line then carries "shared_device": true, `n_gpus` = the number of DEVICES '
                       'and `ranks` = the world size: it can never pass for a scaling number')
  ap.add_argument('--check_gather', action='store_true',
                  help='rank 0 compares the gathered labels of EVERY rank, all utterances, with the CPU oracle '
                       '(outside the timed regions) and puts the verdict into the line (`gather_check`)')
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


def pin_rank_to_cores(local_rank, local_world):
  """Give this rank its share of the host cores (8 ranks x the library's cast threads would
  otherwise all land on the same cores).  Returns the number of cores this rank may use."""
  try:
    cores = sorted(os.sched_getaffinity(0))
    if local_world > 1 and len(cores) >= local_world:
      per = len(cores) // local_world
      mine = cores[local_rank * per:(local_rank + 1) * per]
      os.sched_setaffinity(0, mine)
      return len(mine)
    return len(cores)
  except (AttributeError, OSError):
    return os.cpu_count() or 1


def timed_region(step_fn, sync_fn, steps, warmup, dist=None, reduce_device=None, per_rank=None):
  """W untimed steps, then EXACTLY `steps` steps bracketed by barrier + device sync on both
  sides; returns the MAX over ranks of the elapsed seconds.

  step_fn(): one pass of the hot path (incl. the final label gather when dist is set);
  sync_fn(): wait for this rank's device.  `dist` is torch.distributed or None.
  per_rank: optional list that receives every rank's own seconds BEFORE the closing barrier
  (how long it alone needed: the spread shows imbalance).
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
  own = time.perf_counter() - t0
  if dist is not None:
    dist.barrier()
  elapsed = time.perf_counter() - t0
  if dist is not None:
    t = torch.tensor([elapsed], dtype=torch.float64, device=reduce_device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    if per_rank is not None:
      every = torch.empty(dist.get_world_size(), dtype=torch.float64, device=reduce_device)
      dist.all_gather_into_tensor(every, torch.tensor([own], dtype=torch.float64, device=reduce_device))
      per_rank.extend(float(v) for v in every.tolist())
  elif per_rank is not None:
    per_rank.append(own)
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


def rank_sequences(cfg, args, rank, world):
  """(this rank's utterances, frames of the whole job): deterministic in (cfg, args, rank, world), so that any rank
  can regenerate any other rank's list (--check_gather)."""
  from uisrnn_amd import distributed, synth  # pylint: disable=import-outside-toplevel
  n_utt, n_frames, dim = cfg['utterances_per_gpu'], cfg['frames'], cfg['observation_dim']
  if args.ragged:
    # the WHOLE job's utterances (same list on every rank), dealt longest first
    rng = np.random.default_rng(4242)
    lengths = rng.integers(max(n_frames // 2, 1), n_frames + 1, size=n_utt * world)
    mine = distributed.shard_utterances(lengths, world)[rank]
    return [synth.make_utterance(10_000 + int(i), int(lengths[i]), dim)[0] for i in mine], int(lengths.sum())
  return synth.make_utterances(10_000 + rank * n_utt, n_utt, n_frames, dim)[0], world * n_utt * n_frames


def host_bytes_per_rank(cfg):
  """Host memory one rank of this benchmark holds at its peak (bytes, an upper bound for equal-length utterances):
  the float64 list predict() receives, its packed float32 copy, the pinned float32 block of the host-buffer leg, the
  library's own pinned staging block (float32 frames, allocated on the first host-side decode) and the label /
  offset arrays.  8 ranks of the configs[3] share: 8 x 5.3 GB (tests/test_bench_dist.py asserts the bound)."""
  frames = cfg['utterances_per_gpu'] * cfg['frames']
  dim = cfg['observation_dim']
  return frames * dim * (8 + 4 + 4 + 4) + frames * 4 * 3 + (cfg['utterances_per_gpu'] + 1) * 8 * 2


class Workload:
  """One BASELINE config on this rank: the model, this rank's utterances, the buffers in HBM."""

  def __init__(self, cfg, args, rank, world, dev, dev_index):
    import torch  # pylint: disable=import-outside-toplevel
    from uisrnn_amd import _capi  # pylint: disable=import-outside-toplevel
    self.cfg, self.args, self.torch, self.capi = cfg, args, torch, _capi
    self.dim, self.hid = cfg['observation_dim'], cfg['rnn_hidden_size']
    self.beam, self.look, self.tau = cfg['beam_size'], cfg['look_ahead'], cfg['test_iteration']
    self.params, cfg['model'] = load_model(cfg, args.model)
    self.seqs, self.job_frames = rank_sequences(cfg, args, rank, world)
    self.n_utt = len(self.seqs)
    lens = np.array([s.shape[0] for s in self.seqs], dtype=np.int64)
    self.offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    self.rank_frames = int(self.offsets[-1])
    self.frames = (np.concatenate(self.seqs, axis=0).astype(np.float32) if self.n_utt
                   else np.zeros((0, self.dim), np.float32))
    self.decoder = _capi.Decoder(self.params, device=dev_index)
    self.d_frames = torch.from_numpy(self.frames).to(dev)
    self.d_labels = torch.empty(max(self.rank_frames, 1), dtype=torch.int32, device=dev)
    self.d_scores = torch.empty(max(self.n_utt, 1), dtype=torch.float32, device=dev)
    self.cap = cfg['max_clusters']
    self.timing = False
    self.last = None

  def decode_once(self, flags):
    while True:
      out = self.decoder.decode_device(self.d_frames.data_ptr(), self.offsets, self.beam, self.look, self.tau,
                                       self.d_labels.data_ptr(), self.d_scores.data_ptr(),
                                       max_clusters=self.cap, flags=flags, n_streams=self.args.streams)
      if out['status'] == 0:
        self.last = out
        return out
      # a surviving hypothesis needed more clusters than the tables hold: the Python host's
      # policy (uisrnn_amd/uisrnn.py) is to decode again with twice the room; only set-up / warm-up
      # passes may do that -- a timed pass that retried would be counted as failed
      if self.timing:
        raise RuntimeError('decode hit the cluster cap inside the timed region')
      self.cap *= 2

  def setup(self, sync_fn):
    """Untimed decodes before anything is measured.  Returns (passes, total ms).

    The first pass settles the cluster cap and says which kernel the library runs for this shape
    (uis_stats.decode_kernel); k_decode_rs and the launch-per-step path have no placement trials
    (two passes), the other one-launch kernels try four placements of their control words on the
    next decodes (five passes)."""
    sync_fn()
    t0 = time.perf_counter()
    first = self.decode_once(self.args.flags)
    name = first['stats']['decode_kernel']
    small = name == 'k_decode_rs' or name.startswith('stepwise')
    passes = (2 if small else 5) if self.rank_frames <= 2_000_000 else 1
    for _ in range(passes - 1):
      self.decode_once(self.args.flags)
    sync_fn()
    return passes, 1e3 * (time.perf_counter() - t0)

  def roofline(self, value_rank):
    """The dominant kernel's line, from one profiled pass (HIP events around every launch)."""
    cfg, hid, dim = self.cfg, self.hid, self.dim
    prof = self.decoder.decode_device(self.d_frames.data_ptr(), self.offsets, self.beam, self.look, self.tau,
                                      self.d_labels.data_ptr(), self.d_scores.data_ptr(), max_clusters=self.cap,
                                      flags=self.args.flags | self.capi.UIS_FLAG_PROFILE)['stats']
    n_steps = prof['n_steps']
    resident = prof['kernel_launches']['select'] == 0 and prof['kernel_launches']['gru'] == 1
    fpf = flops_per_frame(cfg)
    ceiling = PEAK_F32_MFMA_TFLOPS * 1e12 / fpf
    bps = bytes_per_step(cfg)
    if resident:
      # ONE launch = the whole beam search.  Algorithmic work of the launch: every surviving
      # hypothesis of every step takes the hidden-side GRU matvec (3H x H), linear_mean1
      # (H x H) and linear_mean2 (D x H); the input-side projection is k_dense_input_proj's.
      kernel = prof['decode_kernel']  # named by the library (uis_stats.decode_kernel): its dispatch rule, not a copy of it
      kclass = 'gru'
      per_row = 2.0 * (3 * hid * hid + (cfg['rnn_depth'] - 1) * 6 * hid * hid + hid * hid + dim * hid)
      flop_algo = per_row * prof['rnn_rows_nodedup']
      flop_exec = per_row * prof['rnn_rows']
      algo_bytes = int(4 * (3 * hid * hid + hid * hid + dim * hid) + self.n_utt * n_steps * bps)
    else:
      # the class that takes the most device time; for the GRU GEMM one launch = one step's
      # hidden-side matvecs (3H x H MACs per surviving hypothesis / prefix)
      kclass = max(('gru', 'head1', 'head2', 'select', 'expand', 'upper_in'),
                   key=lambda k: prof['kernel_ms'][k])
      fam = prof['decode_kernel'].split(':')[-1]  # 'stepwise:k_wt' -> the dense kernels' family, from the library
      kernel = {'gru': fam + '_gru', 'head1': fam + '_head1' if fam != 'k_wt' else 'k_wt_head<1>',
                'head2': fam + '_head2' if fam != 'k_wt' else 'k_wt_head<2>',
                'select': 'k_select_fast', 'expand': 'k_window', 'upper_in': 'k_dense_upper_in'}[kclass]
      launches_gru = max(prof['kernel_launches']['gru'], 1)
      per_row = {'gru': 2.0 * 3 * hid * hid, 'head1': 2.0 * hid * hid, 'head2': 2.0 * dim * hid}.get(kclass, 0.0)
      flop_algo = per_row * prof['rnn_rows_nodedup'] / launches_gru
      flop_exec = per_row * prof['rnn_rows'] / launches_gru
      algo_bytes = int(4 * 3 * hid * hid + prof['rnn_rows_nodedup'] / launches_gru * 4 * (hid + 3 * hid + hid))
    k_launches = max(prof['kernel_launches'][kclass], 1)
    avg_us = 1e3 * prof['kernel_ms'][kclass] / k_launches
    effective = flop_algo / (avg_us * 1e-6) / 1e12 if flop_algo else 0.0
    executed = flop_exec / (avg_us * 1e-6) / 1e12 if flop_exec else 0.0
    return {
        'bound': 'mfma', 'kernel': kernel,
        'decode_path': 'one launch ({})'.format(kernel) if resident else 'launch per step',
        # what the silicon did: the rows the MFMA pipes executed (after row de-duplication)
        'achieved': round(executed, 3), 'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
        'frac': round(min(executed / PEAK_F32_MFMA_TFLOPS, 1.0), 4),
        # the algorithmic rows (one CoreRNN step per surviving hypothesis) over the same
        # time: de-duplication's credit (8 f1); may exceed 1
        'effective': {'tflops': round(effective, 3), 'frac': round(effective / PEAK_F32_MFMA_TFLOPS, 4)},
        'traffic': committed_traffic(kernel, avg_us),
        'avg_launch_us': round(avg_us, 3), 'launches': k_launches,
        'algorithmic_bytes_per_launch': algo_bytes,
        'rows_per_step_algorithmic': round(prof['rnn_rows_nodedup'] / max(n_steps, 1), 1),
        'rows_per_step_executed': round(prof['rnn_rows'] / max(n_steps, 1), 1),
        'ceiling_frames_per_s_fp32': round(ceiling, 0),
        'path_frac_fp32': round(value_rank / ceiling, 4),
        # SURVEY.md 8(d) asks for both fractions; HBM is not the binding one (~300 FLOP/B)
        'path_frac_hbm': round(value_rank * self.tau * bps / (PEAK_HBM_GBS * 1e9), 4),
        'kernel_ms_profile_pass': {k: round(v, 3) for k, v in prof['kernel_ms'].items()},
    }

  def oracle_check(self, sample, sample_frames, threads):
    """The CPU oracle on `sample` utterances cut to `sample_frames`: (seconds, labels identical)."""
    from oracle import oracle  # pylint: disable=import-outside-toplevel
    sample_seqs = [s[:sample_frames] for s in self.seqs[:sample]]
    t0 = time.perf_counter()
    ref = oracle.decode(self.params, sample_seqs, self.beam, self.look, self.tau, n_threads=threads)
    cpu_s = time.perf_counter() - t0
    whole = all(s.shape[0] <= sample_frames for s in self.seqs[:sample])
    if whole:
      got = self.d_labels.cpu().numpy()
      parity = all(np.array_equal(got[self.offsets[u]:self.offsets[u + 1]], ref['labels'][u]) for u in range(sample))
    else:  # truncated utterances: decode exactly those on the GPU for the comparison
      lens = np.array([s.shape[0] for s in sample_seqs], dtype=np.int64)
      offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
      chk = self.decoder.decode(np.concatenate(sample_seqs).astype(np.float32), offs,
                                self.beam, self.look, self.tau, max_clusters=self.cap)
      parity = all(np.array_equal(chk['labels'][offs[u]:offs[u + 1]], ref['labels'][u]) for u in range(sample))
    return cpu_s, bool(parity), int(sum(s.shape[0] for s in sample_seqs))


def check_gathered_labels(cfg, args, world, w, gathered, width):
  """--check_gather: what the LAST pass left in the gathered buffer (every rank's labels; with one rank: this rank's
  label buffer) against the CPU oracle, every utterance of every rank.  Returns a small record for the line."""
  from oracle import oracle  # pylint: disable=import-outside-toplevel
  threads = max(min(len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1), 64), 1)
  got_all = (gathered.cpu().numpy().reshape(world, max(width, 1)) if gathered is not None
             else w.d_labels.cpu().numpy()[None, :])
  t0 = time.perf_counter()
  bad, n_utt, n_frames = [], 0, 0
  for r in range(world):
    seqs = w.seqs if r == 0 and world == 1 else rank_sequences(cfg, args, r, world)[0]
    if not seqs:
      continue
    ref = oracle.decode(w.params, seqs, w.beam, w.look, w.tau, n_threads=threads)
    pos = 0
    for u, s in enumerate(seqs):
      n = s.shape[0]
      if not np.array_equal(got_all[r, pos:pos + n], ref['labels'][u]):
        bad.append((r, u))
      pos += n
    n_utt += len(seqs)
    n_frames += pos
  return {'identical': not bad, 'ranks': world, 'utterances': n_utt, 'frames': n_frames, 'mismatching': bad[:8],
          'oracle_threads': threads, 'seconds': round(time.perf_counter() - t0, 1)}


def run_extra_config(index, args, rank, dev, dev_index, sync_fn):
  """A few passes of another BASELINE config at its stated size: rate, frac, parity."""
  cfg = dict(CONFIGS[index])
  w = Workload(cfg, args, rank, 1, dev, dev_index)
  passes, setup_ms = w.setup(sync_fn)
  w.timing = True
  steps = 3
  el = timed_region(lambda: w.decode_once(args.flags), sync_fn, steps, 1)
  rate = w.rank_frames * steps / el
  roof = w.roofline(rate)
  # parity against the oracle on a sample it finishes in seconds (wide beams: truncated utterances)
  threads = min(os.cpu_count() or 1, 64)
  # (round 5: 16 utterances x 200 frames at look_ahead 2 -- one oracle thread per utterance, about 15 s -- and 16 x 250
  # otherwise; rounds 2-4 checked 4 x 60 and 8 x 250)
  sample = min(w.n_utt, 16)
  cut = min(cfg['frames'], 200 if w.look > 1 else 250)
  cpu_s, parity, _ = w.oracle_check(sample, cut, threads)
  out = {'config': index, 'workload': cfg['workload'], 'value': round(rate, 1), 'unit': 'frames/s',
         'ms_per_step': round(1e3 * el / steps, 3), 'steps': steps, 'setup_passes': passes,
         'setup_ms': round(setup_ms, 1), 'max_clusters': w.cap, 'model': cfg['model'],
         'kernel': roof['kernel'], 'frac': roof['frac'], 'effective_frac': roof['effective']['frac'],
         'avg_launch_us': roof['avg_launch_us'], 'path_frac_fp32': roof['path_frac_fp32'],
         'parity': 'labels identical to oracle: {} ({} utterances x {} frames, {:.1f}s on {} threads)'.format(
             parity, sample, cut, cpu_s, threads)}
  del w
  return out


def main(argv=None):
  args = parse(argv)
  env_world = os.environ.get('WORLD_SIZE')
  if env_world is None and args.gpus is not None and args.gpus > 1:
    import torch  # pylint: disable=import-outside-toplevel
    n_dev = torch.cuda.device_count()
    if n_dev < args.gpus and not (args.allow_shared_device and n_dev >= 1):
      raise SystemExit('bench.py --gpus {}: only {} HIP device(s) visible; one rank per GPU is '
                       'required (no folding of ranks onto one device)'.format(args.gpus, n_dev))
    respawn_under_torchrun(args.gpus, sys.argv[1:] if argv is None else argv)
  world = int(env_world or '1')
  if args.gpus is not None and args.gpus != world:
    raise SystemExit('bench.py --gpus {} but WORLD_SIZE={}'.format(args.gpus, world))
  rank = int(os.environ.get('RANK', '0'))
  local_rank = int(os.environ.get('LOCAL_RANK', '0'))
  local_world = int(os.environ.get('LOCAL_WORLD_SIZE', str(world)))
  rank_cores = pin_rank_to_cores(local_rank, local_world)

  # HBM traffic of the dominant kernel, measured for THIS command where rocprofv3 is there (two counter sub-runs of a
  # few seconds each, BEFORE this process creates its HIP context); otherwise the newest committed record stays
  live_traffic = None
  if world == 1 and rank == 0 and args.device == 'cuda':
    live_traffic = measured_traffic(sys.argv[1:] if argv is None else list(argv))

  import torch  # device memory + torch.distributed only

  cfg = dict(CONFIGS[args.config])
  if args.utterances is not None:
    cfg['utterances_per_gpu'] = args.utterances
  if args.frames is not None:
    cfg['frames'] = args.frames
  if args.beam_size is not None:
    cfg['beam_size'] = args.beam_size
  if args.rnn_depth is not None:
    cfg['rnn_depth'] = args.rnn_depth
  if args.look_ahead is not None:
    cfg['look_ahead'] = args.look_ahead
  if args.rnn_hidden_size is not None:
    cfg['rnn_hidden_size'] = args.rnn_hidden_size
    cfg['workload'] += ' [rnn_hidden_size {}]'.format(args.rnn_hidden_size)
  if args.max_clusters is not None:
    cfg['max_clusters'] = args.max_clusters
  if (args.utterances is not None or args.frames is not None or args.beam_size is not None or args.rnn_depth is not None or
      args.look_ahead is not None or args.max_clusters is not None):
    cfg['workload'] += ' [overridden: {} utt x {} frames, beam {}, look_ahead {}, rnn_depth {}, cluster cap {}]'.format(
        cfg['utterances_per_gpu'], cfg['frames'], cfg['beam_size'], cfg['look_ahead'], cfg['rnn_depth'], cfg['max_clusters'])
  if args.ragged:
    cfg['workload'] += ' [ragged: lengths uniform in [frames / 2, frames], longest-first sharding]'
  big = cfg['utterances_per_gpu'] * cfg['frames'] > 200_000 or cfg['look_ahead'] > 1
  steps = args.steps if args.steps is not None else (3 if big else 10)
  warmup = args.warmup if args.warmup is not None else (3 if cfg['utterances_per_gpu'] * cfg['frames'] <= 2_000_000 else 1)

  on_gpu = args.device == 'cuda'
  if on_gpu:
    n_dev = torch.cuda.device_count()
    if n_dev < 1:
      raise RuntimeError('bench.py needs an MI355X: no HIP device is visible')
    if local_rank >= n_dev and not args.allow_shared_device:
      raise RuntimeError('rank {} (local rank {}) has no GPU of its own: {} HIP device(s) visible, '
                         'one rank per GPU is required'.format(rank, local_rank, n_dev))
    dev_index = local_rank % n_dev   # (--allow_shared_device: the rehearsal folds ranks onto the devices there are)
    torch.cuda.set_device(dev_index)
    dev = torch.device('cuda', dev_index)
    sync_fn = torch.cuda.synchronize
  else:
    if args.backend != 'gloo':
      raise SystemExit('--device cpu is test plumbing: it needs --backend gloo and a stand-in decoder')
    dev_index, dev, sync_fn, n_dev = 0, torch.device('cpu'), (lambda: None), world
  shared_device = bool(on_gpu and local_world > n_dev)
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

  w = Workload(cfg, args, rank, world, dev, dev_index)
  tau = w.tau
  gather_dev = dev if args.backend == 'nccl' else torch.device('cpu')
  # one padded label buffer per rank: with --ragged the ranks hold different numbers of frames
  width = w.rank_frames
  if use_dist:
    wt = torch.tensor([width], dtype=torch.int64, device=gather_dev)
    dist.all_reduce(wt, op=dist.ReduceOp.MAX)
    width = int(wt.item())
    if w.d_labels.numel() < width:
      w.d_labels = torch.empty(width, dtype=torch.int32, device=dev)
  gathered = (torch.empty(world * max(width, 1), dtype=torch.int32, device=gather_dev) if use_dist else None)

  # ---- the three legs of one pass (all end with the labels of this rank's utterances):
  #   predict_f64   uis_decode_f64 from the float64 arrays predict() receives (cast + H2D + decode + D2H)
  #   host_buffers  uis_decode from pinned float32 (H2D + decode + D2H)
  #   device        uis_decode_device, frames and labels resident in HBM
  # (which legs run is a JOB-wide decision -- every leg's timed region issues barriers and gathers, so a rank
  # without utterances still enters every leg and takes part in its collectives with an empty decode)
  want_host = not args.no_host_buffers
  host = {}
  if w.n_utt and want_host:
    pin = (lambda t: t.pin_memory()) if on_gpu else (lambda t: t)
    host['frames'] = pin(torch.from_numpy(w.frames))
    host['labels'] = pin(torch.empty(max(w.rank_frames, 1), dtype=torch.int32))
    host['scores'] = pin(torch.empty(w.n_utt, dtype=torch.float32))

  def gather(src):  # the final gather: the only collective of the path (RCCL over xGMI)
    if use_dist:
      src = src[:max(width, 1)]
      dist.all_gather_into_tensor(gathered, src if args.backend == 'nccl' else src.cpu())

  def device_step():
    w.decode_once(args.flags)
    gather(w.d_labels)

  def host_step():
    if not w.n_utt:
      gather(w.d_labels)
      return
    rc = w.decoder.decode_host(host['frames'].data_ptr(), w.offsets, w.beam, w.look, tau,
                               host['labels'].data_ptr(), host['scores'].data_ptr(),
                               max_clusters=w.cap, flags=args.flags)
    if rc['status'] != 0:
      raise RuntimeError('host-buffer decode hit the cluster cap')
    if use_dist:  # the labels came back to the host (that is the leg): up again for the gather
      w.d_labels[:w.rank_frames].copy_(host['labels'][:w.rank_frames], non_blocking=True)
      gather(w.d_labels)

  f64_out = {}
  def f64_step():
    if not w.n_utt:
      gather(w.d_labels)
      return
    f64_out['r'] = w.decoder.decode_f64(w.seqs, w.beam, w.look, tau, max_clusters=w.cap, flags=args.flags)
    if f64_out['r']['status'] != 0:
      raise RuntimeError('float64-list decode hit the cluster cap')
    if use_dist:
      w.d_labels[:w.rank_frames].copy_(torch.from_numpy(f64_out['r']['labels']), non_blocking=True)
      gather(w.d_labels)

  legs = {'device': device_step}
  if want_host:
    legs['host_buffers'] = host_step
    legs['predict_f64'] = f64_step
  timed_leg = args.timed if args.timed in legs else 'device'   # (--no_host_buffers)

  setup_passes, setup_ms = w.setup(sync_fn)
  w.timing = True
  per_rank = []
  elapsed = timed_region(legs[timed_leg], sync_fn, steps, warmup, dist, gather_dev, per_rank)
  ms_per_step = 1e3 * elapsed / max(steps, 1)
  value = w.job_frames * steps / elapsed
  rates = {timed_leg: value}
  # ---- the other legs, same passes, every rank (a multi-GPU line carries all three rates too)
  n_other = max(steps // 2, 1)
  for name in ('device', 'host_buffers', 'predict_f64'):
    if name in legs and name not in rates:
      el = timed_region(legs[name], sync_fn, n_other, 1, dist, gather_dev)
      rates[name] = w.job_frames * n_other / el
  if host:  # the three legs decoded the same utterances: same labels
    dev_labels = w.d_labels[:w.rank_frames].cpu().numpy()
    if w.rank_frames and not (np.array_equal(host['labels'].numpy()[:w.rank_frames], dev_labels) and
                              np.array_equal(f64_out['r']['labels'], dev_labels)):
      raise RuntimeError('the device-buffer, host-buffer and float64-list decodes disagree')
  stats = w.last['stats'] if w.last else {}

  result = None
  if rank == 0:
    roofline = w.roofline(rates['device'] / world)  # (the path fractions are the HBM-resident leg's)
    if live_traffic and live_traffic.pop('kernel') == roofline['kernel']:
      roofline['traffic'] = live_traffic   # (measured for this very command, before this process took the device)
    # ---- CPU baseline: the oracle on this box's cores, bounded sample
    cpu = None
    if not args.no_cpu_baseline and on_gpu:
      # (N > 1: the timed regions are closed; the other ranks wait at the final barrier while rank 0
      # times the oracle on ITS share of the host cores)
      cores = rank_cores if world > 1 else (os.cpu_count() or 1)
      threads = max(min(cores, 64), 1)
      sample = args.cpu_sample or min(w.n_utt, max(threads, 1))
      if w.look > 1:
        sample = min(sample, 8)
      sample_frames = cfg['frames'] if w.look == 1 else min(cfg['frames'], 100)
      cpu_s, parity, n_sample_frames = w.oracle_check(sample, sample_frames, threads)
      ref = reference_rates() or {}
      cpu = {'value': round(n_sample_frames / cpu_s, 2), 'unit': 'frames/s',
             'cores': threads, 'kind': 'port',
             # the reference itself (google/uis-rnn cannot travel to this box): measured in the dev
             # container by tests/golden/make_trained.py wholebox on the same trained model
             'reference_whole_box_frames_per_s': ref.get('whole_box_frames_per_s'),
             'reference_one_process_frames_per_s': ref.get('one_process_one_thread_frames_per_s'),
             'reference_cores': ref.get('cores'),
             'reference': ref or None,
             'sample': '{} of the {} utterances ({} frames each), {} threads, {:.1f}s; GPU labels '
                       'identical: {}'.format(sample, w.n_utt, sample_frames, threads, cpu_s, parity)}
    gather_check = None
    if args.check_gather:
      gather_check = check_gathered_labels(cfg, args, world, w, gathered, width)
    extras = None
    if (not args.no_extra_configs and world == 1 and on_gpu and args.config == 1 and not args.ragged and
        args.utterances is None and args.frames is None and args.beam_size is None and args.rnn_depth is None and
        args.look_ahead is None and args.max_clusters is None and args.rnn_hidden_size is None):
      first = w.last
      w_cap = w.cap
      extras = []
      for idx in (2, 3, 4):
        try:
          extras.append(run_extra_config(idx, args, rank, dev, dev_index, sync_fn))
        except Exception as e:  # pylint: disable=broad-except
          extras.append({'config': idx, 'error': '{}: {}'.format(type(e).__name__, e)})
      w.last, w.cap = first, w_cap
    result = {
        'metric': 'diarization frames/sec (whole node), beam=10, 256-dim',
        'value': round(value, 1), 'unit': 'frames/s',
        # (a rehearsal on fewer devices than ranks says so, and counts DEVICES: never a scaling number)
        'n_gpus': min(world, n_dev) if shared_device else world, 'ranks': world, 'shared_device': shared_device,
        'steps': steps, 'warmup': warmup,
        'ms_per_step': round(ms_per_step, 3), 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32',
        'data': 'synthetic',
        'config': dict(cfg, parallelism='utterance-sharded x{}'.format(world),
                       max_clusters=w.cap),
        'value_leg': timed_leg,
        'value_definition': {
            'predict_f64': 'uis_decode_f64 -- the entry UISRNN.predict uses: float64 arrays in, cast + H2D of '
                           'the frames + decode + D2H of the labels inside the clock (SURVEY.md 8d), every rank',
            'host_buffers': 'uis_decode from pinned float32: H2D of the frames + decode + D2H of the labels '
                            'inside the clock',
            'device': 'uis_decode_device: frame stream and label buffer resident in HBM when the clock starts',
        }[timed_leg],
        'value_predict_f64': round(rates['predict_f64'], 1) if 'predict_f64' in rates else None,
        'value_host_buffers': round(rates['host_buffers'], 1) if 'host_buffers' in rates else None,
        'value_device': round(rates['device'], 1) if 'device' in rates else None,
        'setup_passes': setup_passes, 'setup_ms': round(setup_ms, 1),
        'per_rank_ms': {'min': round(1e3 * min(per_rank) / steps, 3), 'max': round(1e3 * max(per_rank) / steps, 3)},
        'rank_host_cores': rank_cores,
        'per_rank_host_gb': round(host_bytes_per_rank(cfg) / 1e9, 2),
        'gather_check': gather_check,
        'per_rank_memory_gb': round((w.rank_frames * w.dim * 4 + w.rank_frames * 3 * w.hid * 4) / 1e9, 2),
        'decode_ms_device': round(stats.get('decode_ms', 0.0), 3),
        'n_streams': stats.get('n_streams', 0),
        'roofline': roofline, 'cpu_baseline': cpu, 'extra_configs': extras,
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
