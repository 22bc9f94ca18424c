"""Latency of uis_stream_push: one frame per utterance per push (and 16-frame chunks), the
launch that stays on the device (UIS_FLAG_PERSISTENT) against one launch per push and the
four-kernels-per-step path.  Prints JSON.

  python tools/stream_latency.py [utterances] [pushes]
"""
import json, sys, time
sys.path.insert(0, '.')
import numpy as np
from uisrnn_amd import _capi, synth, weights
import os
n_utt = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n_push = int(sys.argv[2]) if len(sys.argv) > 2 else 400
path = 'tests/golden/trained_d256.uisrnn'
params = weights.load_checkpoint(path) if os.path.exists(path) else synth.tracker_params(256, 512, 1, seed=0)
seqs, _ = synth.make_utterances(30_000, n_utt, n_push + 64 * 16, 256)
dec = _capi.Decoder(params)
out = {'utterances': n_utt, 'model': path if os.path.exists(path) else 'tracker'}
for name, flags in (('persistent_launch', _capi.UIS_FLAG_PERSISTENT), ('one_launch', _capi.UIS_FLAG_RESIDENT),
                    ('four_kernels_per_step', _capi.UIS_FLAG_STEPWISE)):
  dec.stream_begin(n_utt, 10, n_push + 64 * 16, flags=flags)
  for t in range(20):  # warm-up
    dec.stream_push([s[t:t + 1] for s in seqs])
  lat, lat_c, lat_block = [], [], []
  import ctypes
  ones = np.ones(n_utt, dtype=np.int32)
  for t in range(20, n_push):
    chunks = [s[t:t + 1] for s in seqs]
    if t % 4 == 1:   # through the Python host (packs the chunks with numpy)
      t0 = time.perf_counter()
      dec.stream_push(chunks)
      lat.append(time.perf_counter() - t0)
    elif t % 4 == 3:  # through the Python host, one [U, 1, D] float64 array
      block = np.stack(chunks)
      t0 = time.perf_counter()
      dec.stream_push(block)
      lat_block.append(time.perf_counter() - t0)
    else:       # the C entry point alone, frames already packed
      flat = np.ascontiguousarray(np.concatenate(chunks), dtype=np.float32)
      t0 = time.perf_counter()
      rc = dec._lib.uis_stream_push(dec._handle, flat.ctypes.data_as(_capi._fp),
                                    ones.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
      lat_c.append(time.perf_counter() - t0)
      assert rc == 0
      dec._stream_have += ones
  chunk_lat, chunk_lat_c = [], []
  sixteen = np.full(n_utt, 16, dtype=np.int32)
  for k in range(48):
    lo = n_push + 16 * k
    chunks = [s[lo:lo + 16] for s in seqs]
    if k % 2:
      t0 = time.perf_counter()
      dec.stream_push(chunks)
      chunk_lat.append(time.perf_counter() - t0)
    else:
      flat = np.ascontiguousarray(np.concatenate(chunks), dtype=np.float32)
      t0 = time.perf_counter()
      rc = dec._lib.uis_stream_push(dec._handle, flat.ctypes.data_as(_capi._fp),
                                    sixteen.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
      chunk_lat_c.append(time.perf_counter() - t0)
      assert rc == 0
      dec._stream_have += sixteen
  lab_lat = []
  for k in range(10):
    t0 = time.perf_counter()
    labels, scores, _, _ = dec.stream_labels()
    lab_lat.append(time.perf_counter() - t0)
  dec.stream_end()
  lat, chunk_lat, lat_c, chunk_lat_c = np.array(lat) * 1e6, np.array(chunk_lat) * 1e6, np.array(lat_c) * 1e6, np.array(chunk_lat_c) * 1e6
  out[name] = {'push_1_frame_us_median_c_abi': round(float(np.median(lat_c)), 1),
               'push_1_frame_us_median': round(float(np.median(lat)), 1),
               'push_1_frame_us_median_python_block': round(float(np.median(lat_block)) * 1e6, 1),
               'push_1_frame_us_p90': round(float(np.percentile(lat, 90)), 1),
               'push_16_frames_us_median_c_abi': round(float(np.median(chunk_lat_c)), 1),
               'push_16_frames_us_median': round(float(np.median(chunk_lat)), 1),
               'per_frame_step_in_16_chunk_us_c_abi': round(float(np.median(chunk_lat_c)) / 16, 1),
               'labels_us_median': round(float(np.median(lab_lat)) * 1e6, 1),
               'frames_per_utterance_at_labels': int(len(labels[0])),
               'score0': float(scores[0])}
assert out['one_launch']['score0'] == out['four_kernels_per_step']['score0'] == out['persistent_launch']['score0']
print(json.dumps(out))
