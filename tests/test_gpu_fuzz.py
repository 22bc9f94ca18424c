"""Randomised shapes: HIP path vs the oracle, bit for bit, over model / beam / window shapes
the hand-picked cases do not hit (odd feature-tile counts, D > 256, deep models in the wide
tile path, look_ahead with depth 2, beams larger than the fast select path)."""

import numpy as np
import pytest

from uisrnn_amd import _capi
from uisrnn_amd import weights

pytestmark = pytest.mark.gpu


def _bits(a):
  return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def _check(oracle_lib, dim, hid, depth, beam, look, tau, lengths, seed, flags=0, scale=0.4,
           alpha=1.0, tb=0.2, sigma2=0.08):
  rng = np.random.default_rng(seed)
  params = weights.init_params(dim, hid, depth, sigma2=sigma2, transition_bias=tb,
                               crp_alpha=alpha, seed=seed)
  params['rnn_init_hidden'] = (0.2 * rng.standard_normal((depth, hid))).astype(np.float32)
  cents = rng.standard_normal((3, dim))
  seqs = []
  for n in lengths:
    ids = np.repeat(rng.integers(0, 3, size=n // 4 + 1), 4)[:n]
    seqs.append((cents[ids] * scale + 0.1 * rng.standard_normal((n, dim))).astype(np.float64))
  ref = oracle_lib.decode(params, seqs, beam, look, tau, n_threads=8)
  frames, offsets = oracle_lib.pack(seqs)
  cap = max(int(ref['max_clusters'].max()) + look - 1, 2)
  dec = _capi.Decoder(params)
  for fl in (flags, flags | _capi.UIS_FLAG_STEPWISE):  # the default path and the launch-per-step path
    out = dec.decode(frames, offsets, beam, look, tau, max_clusters=cap, flags=fl,
                     want_beam_scores=True)
    assert out['status'] == 0, (dim, hid, depth, beam, look, tau)
    for u in range(len(seqs)):
      assert np.array_equal(out['labels'][offsets[u]:offsets[u + 1]], ref['labels'][u]), u
    assert np.array_equal(_bits(out['beam_scores']), _bits(ref['beam_scores']))


SHAPES = [
    # dim, hid, depth, beam, look, tau, lengths
    (300, 40, 1, 6, 1, 2, [9, 14, 5]),        # D > 256: two 256-float MSE chunks; Hp = 48: 3 feature tiles
    (7, 5, 1, 3, 1, 1, [11, 1, 6, 3]),        # everything smaller than one tile
    (33, 17, 2, 4, 2, 2, [8, 12]),            # depth 2 under look_ahead 2, odd sizes
    (48, 80, 3, 5, 1, 2, [10, 7, 13]),        # 5 feature tiles, depth 3
    (64, 64, 1, 40, 1, 1, [12, 9]),           # beam 40: 40*(K+1) > 256 candidates -> general select
    (20, 24, 1, 2, 3, 1, [10, 4]),            # look_ahead 3
    (16, 16, 1, 64, 1, 1, [6, 6]),            # beam 64
    (128, 96, 2, 8, 1, 2, [15, 15, 2, 9]),    # Hp = 96: 6 feature tiles (no XCD map), depth 2
    (250, 500, 1, 6, 1, 2, [9, 14, 5, 1, 20, 3, 8, 11, 2]),  # pads to 256 / 512: the one-launch decode, padded lanes
    (512, 512, 1, 5, 1, 1, [7, 12, 30]),      # one-launch decode, observation_dim 512
    (400, 505, 1, 4, 1, 3, [6]),              # one-launch decode, one utterance, tau 3
    (250, 250, 1, 6, 1, 2, [9, 14, 5, 1, 20, 3, 8, 11, 2, 30]),  # one-launch decode, hidden 256: two ranks per feature tile
    (120, 256, 1, 5, 1, 2, [12, 7, 40, 3]),   # one-launch decode, hidden 256, observation_dim 128: four ranks per mean tile
    (128, 500, 1, 4, 1, 1, [25, 6]),          # one-launch decode, hidden 512, observation_dim 128
    (512, 256, 1, 4, 1, 2, [10, 18]),         # one-launch decode, hidden 256 < observation_dim 512
]


@pytest.mark.parametrize('shape', SHAPES)
def test_random_shapes_bit_exact(shape, oracle_lib):
  dim, hid, depth, beam, look, tau, lengths = shape
  _check(oracle_lib, dim, hid, depth, beam, look, tau, lengths, seed=dim * 1000 + hid)


def test_wide_tiles_with_odd_shapes(oracle_lib):
  """Row capacity > 2048 switches to the 2x2 tiles: 3 and 5 feature tiles (odd), depth 2."""
  _check(oracle_lib, 40, 40, 2, 10, 1, 1, [4] * 215, seed=7)      # Hp = 48 -> 3 tiles
  _check(oracle_lib, 70, 72, 1, 10, 1, 1, [3] * 210, seed=8)      # Hp = 80 -> 5 tiles, Dp = 80
  _check(oracle_lib, 16, 16, 1, 10, 1, 1, [3] * 210, seed=9)      # one feature tile (< CT)


def test_many_clusters_random_model_look_ahead(oracle_lib):
  """An untrained model with a large crp_alpha opens clusters freely, under look_ahead 2."""
  _check(oracle_lib, 24, 20, 1, 4, 2, 1, [14, 9], seed=11, alpha=30.0, tb=0.5, sigma2=0.5, scale=1.0)
