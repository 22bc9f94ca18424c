#!/usr/bin/env python3
"""Per-kernel register / scratch / spill table from `hipcc -Rpass-analysis=kernel-resource-usage`.

  python tools/resource_table.py [--out profiles/rNN_kernel_resources.txt] [--filter k_decode]

Compiles uisrnn_amd/csrc/uis_decoder.hip for gfx950 with the library's own flags (no GPU needed)
and prints one line per kernel instantiation.
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from uisrnn_amd import build as hip_build  # noqa: E402


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--out', default=None)
  ap.add_argument('--filter', default='')
  ap.add_argument('--log', default=None, help='parse an existing remark log instead of compiling')
  ap.add_argument('-D', dest='defines', action='append', default=[])
  args = ap.parse_args()
  if args.log:
    text = open(args.log).read()
  else:
    with tempfile.TemporaryDirectory() as tmp:
      cmd = [hip_build.hipcc()] + hip_build.FLAGS + ['-Rpass-analysis=kernel-resource-usage'] + \
            ['-D' + d for d in args.defines] + \
            ['-I', os.path.join(ROOT, 'include'), '-I', os.path.join(ROOT, 'uisrnn_amd', 'csrc')] + \
            hip_build.SOURCES + ['-o', os.path.join(tmp, 'lib.so')]
      text = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, check=True).stdout.decode()
  rows, cur = [], None
  for line in text.splitlines():
    m = re.search(r'remark: .*Function Name: (\S+)', line)
    if m:
      name = m.group(1)
      try:
        name = subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-cxxfilt', name], stdout=subprocess.PIPE,
                              check=True).stdout.decode().strip()
      except (OSError, subprocess.CalledProcessError):
        pass
      cur = {'name': re.sub(r'^void ', '', name).split('(')[0]}
      rows.append(cur)
      continue
    m = re.search(r'remark: .*?\s+(TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|SGPRs Spill|VGPRs Spill|'
                  r'Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)', line)
    if m and cur is not None:
      cur[m.group(1).split(' [')[0]] = int(m.group(2))
  cols = ['VGPRs', 'AGPRs', 'ScratchSize', 'VGPRs Spill', 'SGPRs Spill', 'TotalSGPRs', 'Occupancy']
  out = ['{:<72} {}'.format('kernel', ' '.join('{:>11}'.format(c) for c in cols))]
  for r in sorted(rows, key=lambda r: r['name']):
    if args.filter and args.filter not in r['name']:
      continue
    out.append('{:<72} {}'.format(r['name'][:72], ' '.join('{:>11}'.format(r.get(c, '-')) for c in cols)))
  text_out = '\n'.join(out) + '\n'
  sys.stdout.write(text_out)
  if args.out:
    with open(args.out, 'w') as f:
      f.write('# hipcc -Rpass-analysis=kernel-resource-usage, gfx950, flags of uisrnn_amd/build.py'
              + (' + -D' + ' -D'.join(args.defines) if args.defines else '') + '\n' + text_out)


if __name__ == '__main__':
  main()
