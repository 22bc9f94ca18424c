"""Synthetic d-vector utterances and a closed-form speaker-tracking model.

There is no network for datasets or checkpoints and the reference's toy .npz
files are absent from the snapshot, so the benchmark and
the large parity cases run on:

  * make_utterance(): the generator of SURVEY.md 8(d) -- unit-norm Gaussian
    speaker centroids, geometric speaker turns (mean 20 frames), i.i.d.
    N(0, 0.05^2) noise per dimension, float64 [N, D] like the reference's
    `test_sequences` entries (demo.py:38-43);
  * tracker_params(): CoreRNN weights written down in closed form instead of
    trained: the GRU's update gate is held shut (z ~ 0) so h' ~ tanh(x), and
    the two-layer head passes it through (relu(h + 1) - 1), so the network's
    prediction for a cluster is ~ the last frame assigned to it and the
    decoder's running mean tracks the speaker centroid.  Every matrix carries
    dense N(0, eps^2) noise so no operand is zero-filled (zero-filled GEMM
    operands clock higher on MI355X and would flatter the benchmark).
    It behaves like the survey's 300-iteration trained model (<= 4-5 clusters
    on <= 4 speakers, accuracy ~1.0) and is reproducible from a seed on any
    box, which a 6.3 MB trained checkpoint is not.
"""

import numpy as np


def make_utterance(seed, num_frames, observation_dim=256, num_speakers=None,
                   mean_segment=20.0, noise_std=0.05):
  """One synthetic utterance.

  Returns:
    (sequence float64 [N, D], speaker ids int64 [N]).
  """
  rng = np.random.default_rng(seed)
  if num_speakers is None:
    num_speakers = int(rng.integers(2, 5))
  centroids = rng.standard_normal((num_speakers, observation_dim))
  centroids /= np.linalg.norm(centroids, axis=1, keepdims=True)
  ids = np.empty(num_frames, dtype=np.int64)
  pos = 0
  spk = int(rng.integers(num_speakers))
  while pos < num_frames:
    seg = int(rng.geometric(1.0 / mean_segment))
    ids[pos:pos + seg] = spk
    pos += seg
    if num_speakers > 1:
      spk = int((spk + 1 + rng.integers(num_speakers - 1)) % num_speakers)
  seq = centroids[ids] + noise_std * rng.standard_normal(
      (num_frames, observation_dim))
  return seq.astype(np.float64), ids


def make_utterances(base_seed, num_utterances, num_frames, observation_dim=256,
                    num_speakers=None):
  """`num_utterances` utterances with seeds base_seed + u."""
  seqs, ids = [], []
  for u in range(num_utterances):
    n = num_frames[u] if hasattr(num_frames, '__len__') else num_frames
    seq, spk = make_utterance(base_seed + u, int(n), observation_dim,
                              num_speakers)
    seqs.append(seq)
    ids.append(spk)
  return seqs, ids


def tracker_params(observation_dim=256, rnn_hidden_size=512, rnn_depth=1,
                   seed=0, eps=0.01, sigma2=0.005, transition_bias=0.05,
                   crp_alpha=1.0):
  """Closed-form speaker-tracking weights (see module docstring).

  Requires rnn_hidden_size >= observation_dim.
  """
  dim, hid, depth = int(observation_dim), int(rnn_hidden_size), int(rnn_depth)
  if hid < dim:
    raise ValueError('tracker_params needs rnn_hidden_size >= observation_dim')
  rng = np.random.default_rng(seed)

  def noise(shape):
    return (eps * rng.standard_normal(shape)).astype(np.float32)

  w_ih, w_hh, b_ih, b_hh = [], [], [], []
  for layer in range(depth):
    in_dim = dim if layer == 0 else hid
    wi = noise((3 * hid, in_dim))
    wh = noise((3 * hid, hid))
    bi = noise((3 * hid,))
    bh = noise((3 * hid,))
    # candidate gate n: copy the first `dim` inputs
    idx = np.arange(dim)
    wi[2 * hid + idx, idx] += 1.0
    # update gate z held shut: h' = (h - n) z + n ~ n
    bi[hid:2 * hid] -= 5.0
    bh[hid:2 * hid] -= 5.0
    w_ih.append(wi)
    w_hh.append(wh)
    b_ih.append(bi)
    b_hh.append(bh)
  w1 = noise((hid, hid))
  b1 = noise((hid,))
  w2 = noise((dim, hid))
  b2 = noise((dim,))
  idx = np.arange(dim)
  w1[idx, idx] += 1.0
  b1[idx] += 1.0   # relu(h + 1): h in (-1, 1) stays in the linear region
  w2[idx, idx] += 1.0
  # cancel the head's response to h = 0 so that m ~ h (the +1 offset and the
  # noise it picks up through w2 would otherwise bias every prediction)
  b2 -= (w2.astype(np.float64) @ np.maximum(b1.astype(np.float64), 0.0)).astype(
      np.float32)
  return {
      'observation_dim': dim,
      'rnn_hidden_size': hid,
      'rnn_depth': depth,
      'gru_weight_ih': w_ih,
      'gru_weight_hh': w_hh,
      'gru_bias_ih': b_ih,
      'gru_bias_hh': b_hh,
      'linear_mean1_weight': w1,
      'linear_mean1_bias': b1,
      'linear_mean2_weight': w2,
      'linear_mean2_bias': b2,
      'rnn_init_hidden': noise((depth, hid)),
      'sigma2': np.full((dim,), sigma2, dtype=np.float32),
      'transition_bias': float(transition_bias),
      'transition_bias_denominator': 0.0,
      'crp_alpha': float(crp_alpha),
  }


def relabel_first_occurrence(labels):
  """Rename cluster ids by order of first appearance (0, 1, 2, ...)."""
  mapping = {}
  out = []
  for lab in labels:
    lab = int(lab)
    if lab not in mapping:
      mapping[lab] = len(mapping)
    out.append(mapping[lab])
  return out
