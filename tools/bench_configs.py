"""Timing of the non-headline BASELINE configs through the C ABI (one line per config)."""
import sys, time, json
sys.path.insert(0, '.')
import numpy as np
from uisrnn_amd import _capi, synth

def run(name, dim, hid, n_utt, n_frames, beam, look, tau, cap, flags=0, reps=2):
  params = synth.tracker_params(dim, hid, 1, seed=0)
  seqs, _ = synth.make_utterances(50_000, n_utt, n_frames, dim)
  frames = np.concatenate(seqs).astype(np.float32)
  offsets = (np.arange(n_utt + 1) * n_frames).astype(np.int64)
  dec = _capi.Decoder(params)
  best = None
  for _ in range(reps):
    out = dec.decode(frames, offsets, beam, look, tau, max_clusters=cap, flags=flags)
    ms = out['stats']['decode_ms']
    best = ms if best is None else min(best, ms)
  st = out['stats']
  prof = dec.decode(frames, offsets, beam, look, tau, max_clusters=cap, flags=flags | _capi.UIS_FLAG_PROFILE)['stats']
  n = max(prof['n_steps'], 1)
  print(json.dumps({'config': name, 'status': out['status'], 'frames_per_s': round(n_utt * n_frames / (best * 1e-3), 1),
                    'decode_ms': round(best, 2), 'us_per_step': round(1e3 * best / n, 2),
                    'rows_per_step': round(st['rnn_rows'] / n, 1), 'rows_nodedup': round(st['rnn_rows_nodedup'] / n, 1),
                    'cand_per_step': round(st['candidates'] / n, 1), 'maxK': st['max_clusters_seen'],
                    'kernel_us': {k: round(1e3 * v / n, 2) for k, v in prof['kernel_ms'].items() if v}}), flush=True)

STEP = _capi.UIS_FLAG_STEPWISE
which = sys.argv[1:] or ['c3', 'c5', 'c4']
if 'c2' in which:
  run('configs[1]: 64 utt x 500 frames, beam 10', 256, 512, 64, 500, 10, 1, 2, 16)
  run('configs[1], launch-per-step path', 256, 512, 64, 500, 10, 1, 2, 16, flags=STEP)
if 'small' in which:
  run('1 utt x 500 frames, beam 10', 256, 512, 1, 500, 10, 1, 2, 16)
  run('1 utt x 500 frames, launch-per-step path', 256, 512, 1, 500, 10, 1, 2, 16, flags=STEP)
  run('8 utt x 500 frames, beam 10', 256, 512, 8, 500, 10, 1, 2, 16)
  run('8 utt x 500 frames, launch-per-step path', 256, 512, 8, 500, 10, 1, 2, 16, flags=STEP)
  run('256 utt x 200 frames, beam 10', 256, 512, 256, 200, 10, 1, 2, 16)
  run('256 utt x 200 frames, launch-per-step path', 256, 512, 256, 200, 10, 1, 2, 16, flags=STEP)
if 'h256' in which:
  run('hidden 256, D 256: 64 utt x 500 frames, beam 10', 256, 256, 64, 500, 10, 1, 2, 16)
  run('hidden 256, D 256, launch-per-step path', 256, 256, 64, 500, 10, 1, 2, 16, flags=STEP)
  run('hidden 256, D 128: 64 utt x 500 frames, beam 10', 128, 256, 64, 500, 10, 1, 2, 16)
  run('hidden 256, D 128, launch-per-step path', 128, 256, 64, 500, 10, 1, 2, 16, flags=STEP)
if 'c3' in which:
  run('configs[2]: beam 50, look_ahead 2, 16 utt x 200 frames', 256, 512, 16, 200, 50, 2, 2, 12)
if 'c5' in which:
  run('configs[4]: D=512 H=512 beam 20, 64 utt x 200 frames', 512, 512, 64, 200, 20, 1, 2, 11)
  run('configs[4], launch-per-step path', 512, 512, 64, 200, 20, 1, 2, 11, flags=STEP)
if 'c4' in which:
  run('configs[3] per-GPU share: 1024 utt x 200 frames, beam 10', 256, 512, 1024, 200, 10, 1, 2, 16)
  run('configs[3] per-GPU share, launch-per-step path', 256, 512, 1024, 200, 10, 1, 2, 16, flags=STEP)
