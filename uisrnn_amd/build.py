"""Build libuisrnn_hip.so (hipcc, gfx950 only) next to this file.

  python -m uisrnn_amd.build [--force]

The library is built in-tree so that it travels with the repository snapshot
to the GPU box; it is git-ignored.
"""

import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
SOURCES = [os.path.join(_HERE, 'csrc', 'uis_decoder.hip')]
DEPENDS = SOURCES + [
    os.path.join(_HERE, 'csrc', 'uis_kernels.hip'),
    os.path.join(_HERE, 'csrc', 'uis_kernels.h'),
    os.path.join(_HERE, 'csrc', 'uis_select_rs.hip'),
    os.path.join(_HERE, 'csrc', 'uis_decode_coh.hip'),
    os.path.join(_HERE, 'csrc', 'uis_eval.hip'),
    os.path.join(_ROOT, 'include', 'uis_numerics.h'),
    os.path.join(_ROOT, 'include', 'uisrnn_hip.h'),
]
OUTPUT = os.path.join(_HERE, 'libuisrnn_hip.so')

# -ffp-contract=off: the arithmetic order is spelled out with explicit fmaf()
# (include/uis_numerics.h); letting the compiler fuse a*b+c would break the
# bit-for-bit agreement with the oracle.
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off',
         '-fPIC', '-shared', '-fvisibility=hidden', '-Wall',
         '-Wno-unused-function']


def hipcc():
  for cand in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
    if cand and (os.path.sep not in cand or os.path.exists(cand)):
      return cand
  return 'hipcc'


def up_to_date():
  if not os.path.exists(OUTPUT):
    return False
  out_m = os.path.getmtime(OUTPUT)
  return all(os.path.getmtime(d) <= out_m for d in DEPENDS)


def build(force=False, verbose=False, defines=(), output=None, extra_flags=()):
  """Compile the library.  `defines` / `output` / `extra_flags` build tuning variants for A/B runs."""
  output = output or OUTPUT
  if output == OUTPUT and not force and up_to_date():
    return OUTPUT
  cmd = [hipcc()] + FLAGS + list(extra_flags) + ['-D' + d for d in defines] + [
      '-I', os.path.join(_ROOT, 'include'), '-I', os.path.join(_HERE, 'csrc'),
  ] + SOURCES + ['-o', output]
  if verbose:
    print(' '.join(cmd))
  subprocess.check_call(cmd)
  return output


if __name__ == '__main__':
  print(build(force='--force' in sys.argv, verbose=True))
