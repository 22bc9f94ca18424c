"""uis_create pads rnn_depth-1 models up to the one-launch kernels' hidden sizes (128 / 256 / 512) where
that keeps the canonical K-segment length ceil(blocks / 8) of include/uis_numerics.h.  The claim behind
it -- zero-padding the hidden size inside such a range changes no bit of any result -- is pinned here
on the CPU oracle (test infrastructure): a model and its zero-padded twin give the same CoreRNN outputs
(uisrnn.py:45-52), labels and beam scores; outside the range (300 -> 512: segment length 3 -> 4) the
sums are associated differently and bits do move, which is why the library does not pad there -- it EMBEDS
there (round 6): hidden sizes 257 .. 384 have canonical segments of three k-blocks, the 512 shape segments of four,
so the zero units go behind every segment (unit j -> (j // 48) * 64 + j % 48) and every partial sum keeps its
terms: no bit moves (test_segment_embedding_*; on the device: tests/test_gpu_parity.py)."""
import numpy as np
import pytest

from uisrnn_amd import weights


def _bits(a):
  return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def _pad_hidden(params, hp):
  """The same model with rnn_hidden_size hp: the added units have zero weights, biases and initial state."""
  h, depth = params['rnn_hidden_size'], params['rnn_depth']
  assert hp >= h

  def rows(w):  # [3H, K] -> [3Hp, K], gate by gate (r | z | n, torch.nn.GRU's layout)
    out = np.zeros((3 * hp,) + w.shape[1:], dtype=np.float32)
    for g in range(3):
      out[g * hp:g * hp + h] = w[g * h:(g + 1) * h]
    return out

  def cols(w):  # [.., H] -> [.., Hp]
    out = np.zeros(w.shape[:-1] + (hp,), dtype=np.float32)
    out[..., :h] = w
    return out

  p = dict(params)
  p['rnn_hidden_size'] = hp
  # (layer 0 reads the observation, the upper layers the hidden vector below: their K axis is padded too)
  p['gru_weight_ih'] = [rows(w) if l == 0 else cols(rows(w)) for l, w in enumerate(params['gru_weight_ih'])]
  p['gru_weight_hh'] = [cols(rows(w)) for w in params['gru_weight_hh']]
  p['gru_bias_ih'] = [rows(b) for b in params['gru_bias_ih']]
  p['gru_bias_hh'] = [rows(b) for b in params['gru_bias_hh']]
  assert len(p['gru_weight_ih']) == depth
  w1 = np.zeros((hp, hp), dtype=np.float32)
  w1[:h, :h] = params['linear_mean1_weight']
  p['linear_mean1_weight'] = w1
  p['linear_mean1_bias'] = cols(params['linear_mean1_bias'])
  p['linear_mean2_weight'] = cols(params['linear_mean2_weight'])
  p['rnn_init_hidden'] = cols(params['rnn_init_hidden'])
  return p


def _model(dim, hidden, seed, depth=1):
  p = weights.init_params(dim, hidden, depth, sigma2=0.1, transition_bias=0.2, crp_alpha=1.0, seed=seed)
  p['rnn_init_hidden'] = (0.2 * np.random.default_rng(seed).standard_normal((depth, hidden))).astype(np.float32)
  return p


def _utterances(dim, seed):
  rng = np.random.default_rng(seed)
  cents = rng.standard_normal((3, dim))
  seqs = []
  for n in (17, 5, 26, 1, 12):
    ids = np.repeat(rng.integers(0, 3, size=n // 4 + 1), 4)[:n]
    seqs.append((cents[ids] * 0.4 + 0.1 * rng.standard_normal((n, dim))).astype(np.float64))
  return seqs


@pytest.mark.parametrize('dim,hidden,padded,depth', [(20, 200, 256, 1), (33, 130, 256, 1), (20, 100, 128, 1), (16, 70, 128, 1),
                                                    (12, 400, 512, 1), (24, 500, 512, 1), (20, 100, 128, 2), (12, 200, 256, 3)])
def test_zero_padding_the_hidden_size_inside_a_segment_class_moves_no_bit(dim, hidden, padded, depth, oracle_lib):
  params = _model(dim, hidden, seed=dim + hidden, depth=depth)
  twin = _pad_hidden(params, padded)
  rng = np.random.default_rng(hidden)
  for _ in range(4):
    x = rng.standard_normal(dim).astype(np.float32)
    h0 = rng.standard_normal((depth, hidden)).astype(np.float32)
    h0p = np.zeros((depth, padded), dtype=np.float32)
    h0p[:, :hidden] = h0
    mean, hout = oracle_lib.rnn_step(params, x, h0)
    mean_p, hout_p = oracle_lib.rnn_step(twin, x, h0p)
    assert np.array_equal(_bits(mean), _bits(mean_p))
    assert np.array_equal(_bits(hout), _bits(hout_p[:, :hidden]))
    assert not hout_p[:, hidden:].any()
  seqs = _utterances(dim, seed=hidden)
  for beam, look, tau in ((6, 1, 2), (4, 2, 1)):
    a = oracle_lib.decode(params, seqs, beam, look, tau, n_threads=4)
    b = oracle_lib.decode(twin, seqs, beam, look, tau, n_threads=4)
    for la, lb in zip(a['labels'], b['labels']):
      assert np.array_equal(la, lb)
    assert np.array_equal(_bits(a['beam_scores']), _bits(b['beam_scores']))


def test_padding_across_segment_classes_is_not_exact(oracle_lib):
  """Hidden size 300 (19 k-blocks, segments of 3) padded to 512 (segments of 4): other partial sums."""
  params = _model(16, 300, seed=3)
  twin = _pad_hidden(params, 512)
  rng = np.random.default_rng(9)
  moved = 0
  for _ in range(8):
    x = rng.standard_normal(16).astype(np.float32)
    h0 = rng.standard_normal((1, 300)).astype(np.float32)
    h0p = np.zeros((1, 512), dtype=np.float32)
    h0p[:, :300] = h0
    _, hout = oracle_lib.rnn_step(params, x, h0)
    _, hout_p = oracle_lib.rnn_step(twin, x, h0p)
    np.testing.assert_allclose(hout, hout_p[:, :300], rtol=1e-4, atol=1e-6)
    moved += int((_bits(hout) != _bits(hout_p[:, :300])).sum())
  assert moved > 0


def _embed_hidden(params, hp, seg, seg_p):
  """The same model with rnn_hidden_size hp, unit j moved to (j // seg) * seg_p + j % seg; zero units in between."""
  h, depth = params['rnn_hidden_size'], params['rnn_depth']
  pos = np.array([(j // seg) * seg_p + j % seg for j in range(h)])
  assert pos.max() < hp

  def rows(w):  # [3H, ..] -> [3Hp, ..]
    out = np.zeros((3 * hp,) + w.shape[1:], dtype=np.float32)
    for g in range(3):
      out[g * hp + pos] = w[g * h:(g + 1) * h]
    return out

  def cols(w):  # [.., H] -> [.., Hp]
    out = np.zeros(w.shape[:-1] + (hp,), dtype=np.float32)
    out[..., pos] = w
    return out

  p = dict(params)
  p['rnn_hidden_size'] = hp
  p['gru_weight_ih'] = [rows(w) if l == 0 else cols(rows(w)) for l, w in enumerate(params['gru_weight_ih'])]
  p['gru_weight_hh'] = [cols(rows(w)) for w in params['gru_weight_hh']]
  p['gru_bias_ih'] = [rows(b) for b in params['gru_bias_ih']]
  p['gru_bias_hh'] = [rows(b) for b in params['gru_bias_hh']]
  w1 = np.zeros((hp, hp), dtype=np.float32)
  w1[np.ix_(pos, pos)] = params['linear_mean1_weight']
  p['linear_mean1_weight'] = w1
  p['linear_mean1_bias'] = cols(params['linear_mean1_bias'])
  p['linear_mean2_weight'] = cols(params['linear_mean2_weight'])
  p['rnn_init_hidden'] = cols(params['rnn_init_hidden'])
  return p, pos


@pytest.mark.parametrize('dim,hidden,depth', [(16, 300, 1), (20, 257, 1), (12, 384, 1), (24, 320, 2), (16, 370, 1)])
def test_segment_embedding_of_hidden_sizes_257_to_384_into_512_moves_no_bit(dim, hidden, depth, oracle_lib):
  """What uis_create does for rnn_hidden_size 257 .. 384 since round 6 (HidMap): the model inside the 512-wide shape
  with a zero k-block behind each canonical segment of three.  On the CPU oracle: CoreRNN outputs, labels and beam
  scores of the model and of its embedded twin agree bit for bit."""
  params = _model(dim, hidden, seed=dim + hidden, depth=depth)
  twin, pos = _embed_hidden(params, 512, 48, 64)
  rng = np.random.default_rng(hidden)
  for _ in range(4):
    x = rng.standard_normal(dim).astype(np.float32)
    h0 = rng.standard_normal((depth, hidden)).astype(np.float32)
    h0p = np.zeros((depth, 512), dtype=np.float32)
    h0p[:, pos] = h0
    mean, hout = oracle_lib.rnn_step(params, x, h0)
    mean_p, hout_p = oracle_lib.rnn_step(twin, x, h0p)
    assert np.array_equal(_bits(mean), _bits(mean_p))
    assert np.array_equal(_bits(hout), _bits(hout_p[:, pos]))
    rest = np.ones(512, dtype=bool)
    rest[pos] = False
    assert not hout_p[:, rest].any()
  seqs = _utterances(dim, seed=hidden)
  for beam, look, tau in ((6, 1, 2), (4, 2, 1)):
    a = oracle_lib.decode(params, seqs, beam, look, tau, n_threads=4)
    b = oracle_lib.decode(twin, seqs, beam, look, tau, n_threads=4)
    for la, lb in zip(a['labels'], b['labels']):
      assert np.array_equal(la, lb)
    assert np.array_equal(_bits(a['beam_scores']), _bits(b['beam_scores']))
