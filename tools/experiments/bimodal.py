"""Does the 1.28 M / 1.33 M bimodality of the bench follow the physical placement of the decoder's
buffers?  Several handles in one process, each created after a different amount of filler
allocations (kept alive), device-resident decodes timed by the library's own events."""
import sys
sys.path.insert(0, '.')
import numpy as np, torch
import bench
from uisrnn_amd import _capi, synth
cfg = dict(bench.CONFIGS[1])
params = bench.load_model(cfg, 'auto')[0]
seqs, _ = synth.make_utterances(10_000, 64, 500, 256)
frames = torch.from_numpy(np.concatenate(seqs).astype(np.float32)).cuda()
offsets = (np.arange(65) * 500).astype(np.int64)
labels = torch.empty(32000, dtype=torch.int32, device='cuda')
scores = torch.empty(64, dtype=torch.float32, device='cuda')
import os
def run(tag):
  dec = _capi.Decoder(params)
  ms = []
  for i in range(4):
    out = dec.decode_device(frames.data_ptr(), offsets, 10, 1, 2, labels.data_ptr(), scores.data_ptr())
    ms.append(out['stats']['decode_ms'])
  print(tag, 'decode_ms %.2f' % (sum(ms[1:]) / 3), flush=True)
  dec.close()
if sys.argv[1] == 'tune':
  # the placement tuner of the control words: handles created one after the other, ten decodes each
  for k in range(8):
    dec = _capi.Decoder(params)
    ms = [dec.decode_device(frames.data_ptr(), offsets, 10, 1, 2, labels.data_ptr(), scores.data_ptr())['stats']['decode_ms'] for i in range(10)]
    print('handle', k, 'decode_ms', ' '.join('%.2f' % v for v in ms), flush=True)
    dec.close()
  sys.exit(0)
if sys.argv[1] == 'offsets':
  # one handle, the control words at every offset given (bytes): device time of decodes 2-4 at each
  dec = _capi.Decoder(params)
  offs = [int(v) for v in sys.argv[2].split(',')]
  for rep in range(2):
    for off in offs:
      os.environ['UIS_CTL_OFFSET'] = str(off)
      ms = [dec.decode_device(frames.data_ptr(), offsets, 10, 1, 2, labels.data_ptr(), scores.data_ptr())['stats']['decode_ms'] for i in range(4)]
      print('offset', off, 'decode_ms %.2f' % min(ms[1:]), flush=True)
  sys.exit(0)
if sys.argv[1] == 'which':
  os.environ['UIS_ARENA_SHIFT'] = '0'
  for k in range(8):
    dec = _capi.Decoder(params)
    for i in range(2):
      dec.decode_device(frames.data_ptr(), offsets, 10, 1, 2, labels.data_ptr(), scores.data_ptr())
    out = dec.decode_device(frames.data_ptr(), offsets, 10, 1, 2, labels.data_ptr(), scores.data_ptr(), flags=_capi.UIS_FLAG_PROFILE)
    st = out['stats']
    print('handle', k, 'decode_ms %.2f' % st['decode_ms'], {n: round(v, 3) for n, v in st['kernel_ms'].items() if v}, flush=True)
    dec.close()
  sys.exit(0)
if sys.argv[1] == 'streams':
  # does the mode follow the hardware queue behind the handle's stream?  k dummy streams are
  # created (and kept) before each handle
  os.environ['UIS_ARENA_SHIFT'] = '0'
  dummies = []
  for k in (0, 0, 1, 0, 1, 1, 2, 0, 3, 0, 0, 1):
    for _ in range(k):
      dummies.append(torch.cuda.Stream())
    run('after %d more dummy streams (total %d)' % (k, len(dummies)))
  sys.exit(0)
shifts = [int(v) for v in sys.argv[1].split(',')]
print('arena base / shift -> decode ms')
for rep in range(2):
  for sh in shifts:
    os.environ['UIS_ARENA_SHIFT'] = str(sh)
    dec = _capi.Decoder(params)
    ms = []
    for i in range(4):
      out = dec.decode_device(frames.data_ptr(), offsets, 10, 1, 2, labels.data_ptr(), scores.data_ptr())
      ms.append(out['stats']['decode_ms'])
    print('shift', sh, 'decode_ms', ' '.join('%.2f' % v for v in ms[1:]), flush=True)
    dec.close()
