"""Host-side mirror of ``uisrnn.UISRNN`` for the decode path.

Same constructor, ``load`` / ``save`` / ``predict`` / ``predict_single`` and
module-level ``parallel_predict`` as the reference (uisrnn/uisrnn.py:80-623),
same argument meaning and the same exceptions; the beam search itself runs in
libuisrnn_hip.so on the MI355X through uisrnn_amd._capi.  There is no CPU
implementation behind this class: without the HIP library and a gfx950 device
``predict`` raises.

Training (``fit``) is outside the scope of this package: train with the
reference and ``load()`` its checkpoint here.
"""

import threading

import numpy as np

from uisrnn_amd import _capi
from uisrnn_amd import weights

_DEFAULT_MAX_CLUSTERS = 16


def _initial_cluster_cap(args):
  """Device table size (clusters per hypothesis) of the first decode attempt.

  The fast select kernel and the one-launch decode need beam_size * (cap + 1) <= 256
  candidates (include/uisrnn_hip.h).  With the default cap 16 a beam wider than 15 would
  leave that path; real diarization rarely opens more than a handful of clusters, so a wide
  beam starts with the largest cap that still fits (never below 8) and the overflow retry
  below doubles it when an utterance does need more.
  """
  explicit = int(getattr(args, 'max_clusters', 0) or 0)
  if explicit:
    return explicit
  fits = 256 // max(int(args.beam_size), 1) - 1
  if int(args.look_ahead) == 1 and 8 <= fits < _DEFAULT_MAX_CLUSTERS:
    return fits
  return _DEFAULT_MAX_CLUSTERS
_MAX_CLUSTERS_LIMIT = 4096   # (the library's own limit: uis_decode_opts.max_clusters)
_DEFAULT_LEVEL_CAP = 32768   # include/uisrnn_hip.h: uis_decode_opts.level_cap's default ...
_MAX_LEVEL_CAP = 524287      # ... and its maximum


class EmptyBeamError(ValueError, IndexError):
  """Every candidate of some decode step was non-finite: no hypothesis survived.

  The reference fails in the same situation, with an exception that depends on WHERE the beam
  empties: `ValueError: max() arg is an empty sequence` at the next step's
  uisrnn/uisrnn.py:531 if frames remain, `IndexError: list index out of range` at
  uisrnn/uisrnn.py:561 after the last one (recorded in tests/golden/probes.json).  This class
  is both, so a caller's handler for either keeps working.  (With a NaN -- not inf -- score the
  reference can instead keep NaN-scored hypotheses, numpy sorts NaN behind inf; that is not
  reproduced: non-finite candidates are never selected here)
  """


class LookAheadWindowError(_capi.HipLibraryError):
  """look_ahead >= 2: inside a window some utterances had more live assignment prefixes than the
  device can hold (beam_size * clusters ^ (look_ahead - 1) hypotheses per intermediate level); the
  reference has no such bound -- it enumerates them one by one.  Round 5: predict() no longer raises
  this at the default capacity (32768 per level): the affected utterances are decoded again on
  their own with eight times the room, up to 524287 hypotheses per level, and only a window beyond
  THAT -- or beyond the device's memory -- ends here.  Lower look_ahead or beam_size for those
  utterances.

  Attributes:
    utterances: indices (into the list given to predict) of the affected utterances.
    results: the label lists of every OTHER utterance (None at the affected positions) -- they were
      decoded again without the affected ones, so nothing valid is thrown away.
  """
  utterances = ()
  results = None


class UISRNN:
  """Unbounded Interleaved-State RNN -- MI355X decode."""

  def __init__(self, args):
    """Construct from the model namespace (uisrnn/uisrnn.py:83-107).

    Weights are freshly initialised like the reference's; use load() or
    load_params() to install trained ones.
    """
    self.observation_dim = args.observation_dim
    self.params = weights.init_params(
        args.observation_dim, args.rnn_hidden_size, args.rnn_depth,
        sigma2=args.sigma2, transition_bias=args.transition_bias,
        crp_alpha=args.crp_alpha)
    self.estimate_sigma2 = args.sigma2 is None
    self.estimate_transition_bias = args.transition_bias is None
    self.device_index = int(getattr(args, 'device_index', 0))
    self.verbosity = getattr(args, 'verbosity', 3)
    self._decoder = None
    self._extra_decoders = {}
    self.last_stats = None
    self._single_pass = False
    self._state_lock = threading.Lock()  # last_stats / _single_pass: parallel_predict's workers share the model

  # ---- the attributes callers of the reference read and write
  @property
  def transition_bias(self):
    return self.params['transition_bias']

  @transition_bias.setter
  def transition_bias(self, value):
    self.params['transition_bias'] = value
    self._invalidate()

  @property
  def transition_bias_denominator(self):
    return self.params.get('transition_bias_denominator', 0.0)

  @property
  def crp_alpha(self):
    return self.params['crp_alpha']

  @crp_alpha.setter
  def crp_alpha(self, value):
    self.params['crp_alpha'] = value
    self._invalidate()

  @property
  def sigma2(self):
    return self.params['sigma2']

  @sigma2.setter
  def sigma2(self, value):
    self.params['sigma2'] = np.broadcast_to(
        np.asarray(value, dtype=np.float32), (self.observation_dim,)).copy()
    self._invalidate()

  @property
  def rnn_init_hidden(self):
    depth, hid = self.params['rnn_depth'], self.params['rnn_hidden_size']
    return self.params['rnn_init_hidden'].reshape(depth, 1, hid)

  def _invalidate(self):
    if self._decoder is not None:
      self._decoder.close()
    self._decoder = None
    for dec in self._extra_decoders.values():
      dec.close()
    self._extra_decoders = {}

  def load_params(self, params):
    """Install a parameter dict (uisrnn_amd.weights) wholesale."""
    if params['observation_dim'] != self.observation_dim:
      raise ValueError('parameters do not match args.observation_dim')
    self.params = params
    self._invalidate()

  def load(self, filepath):
    """Load a checkpoint written by the reference's save() (uisrnn.py:149-170)."""
    self.load_params(weights.load_checkpoint(filepath))

  def save(self, filepath):
    """Write the reference's checkpoint format (uisrnn.py:135-147)."""
    weights.save_checkpoint(self.params, filepath)

  def fit(self, *unused_args, **unused_kwargs):
    raise NotImplementedError(
        'Training is outside the scope of uisrnn_amd (decode path only): '
        'train with google/uis-rnn and load() the checkpoint.')

  fit_concatenated = fit

  def _get_decoder(self, device=None):
    if self.params['transition_bias'] is None:
      # the reference fails in np.log(None) (uisrnn.py:416-418)
      raise TypeError('transition_bias is None: fit or load a model first.')
    if device is None or device == self.device_index:
      if self._decoder is None:
        self._decoder = _capi.Decoder(self.params, self.device_index)
      return self._decoder
    if device not in self._extra_decoders:  # one handle per further GPU (parallel_predict)
      self._extra_decoders[device] = _capi.Decoder(self.params, device)
    return self._extra_decoders[device]

  def _check_sequence(self, test_sequence):
    """The reference's argument checks, uisrnn/uisrnn.py:510-521."""
    if (not isinstance(test_sequence, np.ndarray) or
        test_sequence.dtype != float):
      raise TypeError('test_sequence should be a numpy array of float type.')
    if test_sequence.ndim != 2:
      raise ValueError('test_sequence must be 2-dim array.')
    if test_sequence.shape[1] != self.observation_dim:
      raise ValueError('test_sequence does not match the dimension specified '
                       'by args.observation_dim.')

  def _decode_batch(self, sequences, args, flags=0, device=None, decoder=None, level_cap=0):
    """Decode a list of validated sequences in one lock-step batch.

    `decoder` (a _capi.Decoder built from self.params) overrides the model's own handle for
    `device`; parallel_predict passes one per worker thread.
    """
    decoder = decoder or self._get_decoder(device)
    # (args.level_cap, like args.max_clusters, is an extension: where the retries of a look-ahead window start)
    level_cap = level_cap or int(getattr(args, 'level_cap', 0) or 0)
    n_utt = len(sequences)
    results = [None] * n_utt
    pending = list(range(n_utt))
    cap = _initial_cluster_cap(args)
    stats = None
    while pending:
      # the float64 arrays go to the library as they are (uis_decode_f64): it casts to float32
      # once, like torch.from_numpy(seq).float() (uisrnn.py:525), on its own threads while the
      # earlier chunks are already travelling to the device
      sub = [sequences[u] for u in pending]
      sub_off = np.zeros(len(pending) + 1, dtype=np.int64)
      sub_off[1:] = np.cumsum([s.shape[0] for s in sub])
      try:
        out = decoder.decode_f64(sub, args.beam_size, args.look_ahead,
                                 args.test_iteration, max_clusters=cap, flags=flags, level_cap=level_cap)
      except _capi.HipLibraryError as err:
        if err.status == _capi.UIS_ERR_OOM and len(pending) > 1:
          # the decode state of this many utterances does not fit the device (or the pinned staging
          # block the host): the reference's predict takes a list of any size (uisrnn.py:588-589), so
          # the list goes in two halves, one after the other -- alternating members, so that a list
          # dealt longest-first stays even -- and each half may halve again
          level_full, first_err = [], None
          for part in (pending[0::2], pending[1::2]):
            try:
              part_labels = self._decode_batch([sequences[u] for u in part], args, flags, device, decoder, level_cap)
            except LookAheadWindowError as inner:
              # (a half whose look-ahead window overflowed: its indices and results are numbered inside
              # the half -- hand them up in the caller's numbering, and still decode the other half)
              part_labels = inner.results
              level_full.extend(part[k] for k in inner.utterances)
              first_err = first_err or inner
            for u, labels in zip(part, part_labels):
              results[u] = labels
          with self._state_lock:
            self._single_pass = False  # several decodes: no single resident label buffer
          if first_err is not None:
            level_full = sorted(level_full)
            exc = LookAheadWindowError('{} (utterances {})'.format(
                str(first_err).split(' (utterances')[0], level_full))
            exc.status, exc.utterances, exc.results = first_err.status, tuple(level_full), results
            raise exc from first_err
          return results
        if err.status != _capi.UIS_ERR_UNSUPPORTED or args.look_ahead < 2:
          raise
        # a look-ahead window of some utterances held more assignment prefixes than a level has
        # room for (bit 1 of their flags).  The flags come from the library, sized by the library:
        # a decode that was refused for its OPTIONS (look_ahead > 1024, beam_size > 32767) never
        # started and leaves no flags behind -- that error goes up as it is.
        flags_now = decoder.last_overflow()
        if flags_now.shape[0] != len(pending):
          raise
        level_full = [u for k, u in enumerate(pending) if flags_now[k] & 2]
        if not level_full:
          raise
        # The others' results are valid but were not handed out: decode them again on their own, at the
        # same capacity.  The affected ones get eight times the room per level (the library's default is
        # 32768 hypotheses per level, its maximum 524287), again and again; what ends the retries is the
        # maximum, or the device's memory (UIS_ERR_OOM for a single utterance).
        rest = [u for u in pending if u not in level_full]
        failed, first_err = [], None
        groups = [(rest, level_cap)]
        now_cap = level_cap or _DEFAULT_LEVEL_CAP
        if now_cap < _MAX_LEVEL_CAP:
          groups.append((level_full, min(now_cap * 8, _MAX_LEVEL_CAP)))
        else:
          failed, first_err = list(level_full), err
        for members, cap_level in groups:
          if not members:
            continue
          try:
            partial = self._decode_batch([sequences[u] for u in members], args, flags, device, decoder, cap_level)
          except LookAheadWindowError as inner:
            # (indices and results are numbered inside `members`: hand them up in the caller's numbering)
            partial = inner.results
            failed.extend(members[k] for k in inner.utterances)
            first_err = first_err or inner
          except _capi.HipLibraryError as inner:
            if inner.status != _capi.UIS_ERR_OOM or members is rest:
              raise
            # One utterance's window does not fit the device at this capacity (the halving inside that call got down
            # to a single utterance and still met UIS_ERR_OOM).  Which one is not known here: the others of this group
            # would fit -- decode the members one by one and name only those that do not (round 6).
            partial = []
            for u in members:
              try:
                one = [None] if len(members) == 1 else self._decode_batch([sequences[u]], args, flags, device, decoder, cap_level)
                one_err = inner if len(members) == 1 else None
              except LookAheadWindowError as e1:
                one, one_err = [None], e1
              except _capi.HipLibraryError as e1:
                if e1.status != _capi.UIS_ERR_OOM:
                  raise
                one, one_err = [None], e1
              partial.extend(one)
              if one_err is not None:
                failed.append(u)
                first_err = first_err or one_err
          for u, labels in zip(members, partial):
            results[u] = labels
        with self._state_lock:
          self._single_pass = False
        if failed:
          failed = sorted(failed)
          exc = LookAheadWindowError('{} (utterances {})'.format(str(first_err).split(' (utterances')[0], failed))
          exc.status, exc.utterances, exc.results = _capi.UIS_ERR_UNSUPPORTED, tuple(failed), results
          raise exc from first_err
        return results
      if stats is None:
        stats = out['stats']
      still = []
      for k, u in enumerate(pending):
        if out['overflow'][k]:
          still.append(u)
        else:
          labels = out['labels'][sub_off[k]:sub_off[k + 1]]
          if labels.size and labels[0] < 0:
            # every candidate of some step was non-finite (nan/inf in the input or the
            # weights): the reference ends up indexing an empty beam_set
            # (uisrnn/uisrnn.py:561) and raises the same exception type
            raise EmptyBeamError('the beam became empty (max() arg is an empty sequence / list '
                                 'index out of range in the reference): non-finite scores in '
                                 'utterance {}'.format(u))
          results[u] = labels.tolist()
      pending = still
      if pending:
        # a surviving hypothesis opened more clusters than the device tables
        # hold: decode those utterances again with twice the room
        cap *= 2
        if cap > _MAX_CLUSTERS_LIMIT:
          raise RuntimeError(
              'more than {} clusters per hypothesis'.format(_MAX_CLUSTERS_LIMIT))
    with self._state_lock:  # (parallel_predict: the last worker to finish wins, whole)
      self.last_stats = stats
      self._single_pass = cap == _initial_cluster_cap(args)  # no retry: one decode covered everything
    return results

  def predict_and_evaluate(self, test_sequences, test_cluster_ids, args):
    """predict() followed by the demo's accuracy step with the labels kept on the device.

    The reference's demo.py:58-64 calls predict and compute_sequence_match_accuracy per
    utterance on the host.  Here the batch is decoded once and the predicted labels, still in
    HBM, are matched against the ground truth by uis_eval_last_decode (confusion matrix +
    exact assignment per utterance in one launch).  Not part of the reference's API.

    Args:
      test_sequences: list of [N_i, D] float arrays.
      test_cluster_ids: list of lists of ground-truth ids (any hashable, e.g. str).
    Returns:
      (predicted label lists, accuracies) -- accuracies equal
      compute_sequence_match_accuracy(truth, predicted) exactly.

    The device kernel handles label sequences with at most 64 distinct ids each (ids below
    65536 after densification); a batch with a sequence beyond that -- the reference's evals.py
    has no such limit, and the cluster-cap retry lets predictions reach 1024 clusters -- is
    scored by the host function evals.compute_sequence_match_accuracy instead, same values.
    """
    from uisrnn_amd import _capi, evals  # pylint: disable=import-outside-toplevel
    if not isinstance(test_sequences, list) or not isinstance(test_cluster_ids, list):
      raise TypeError('test_sequences and test_cluster_ids must be lists')
    if len(test_sequences) != len(test_cluster_ids):
      raise ValueError('one list of cluster ids per test sequence')
    for seq, ids in zip(test_sequences, test_cluster_ids):
      self._check_sequence(seq)
      if len(ids) != seq.shape[0] or not len(ids):
        raise ValueError('sequence1 and sequence2 must be non-empty and of the same size')
    if not test_sequences:
      return [], []
    decoder = self._get_decoder()
    self._single_pass = False
    predicted = self._decode_batch(test_sequences, args)
    truth = np.concatenate([evals.dense_ids(list(ids)) for ids in test_cluster_ids])
    lens = np.array([s.shape[0] for s in test_sequences], dtype=np.int64)
    try:
      if self._single_pass:  # the labels of that one decode are still resident: no upload
        matched = decoder.eval_last_decode(truth, len(test_sequences))
      else:  # the cluster-cap retry decoded a subset last: hand the labels back
        offsets = np.zeros(len(lens) + 1, dtype=np.int64)
        offsets[1:] = np.cumsum(lens)
        flat = np.concatenate([evals.dense_ids(p) for p in predicted])
        matched = decoder.eval_matched(truth, flat, offsets)
    except _capi.HipLibraryError as err:
      if err.status != _capi.UIS_ERR_UNSUPPORTED:
        raise
      # more than 64 distinct ids in some sequence: the host function has no limit
      return predicted, [evals.compute_sequence_match_accuracy(list(ids), list(p))
                         for ids, p in zip(test_cluster_ids, predicted)]
    return predicted, [float(m) / int(n) for m, n in zip(matched, lens)]

  def predict_single(self, test_sequence, args):
    """Predict labels for one test sequence (uisrnn/uisrnn.py:479-562).

    Args:
      test_sequence: 2-dim float64 numpy array [N, D].
      args: inference namespace (beam_size, look_ahead, test_iteration).

    Returns:
      list of N ints: the predicted cluster id per frame.

    Raises:
      TypeError: test_sequence is not a float numpy array.
      ValueError: wrong rank or observation dimension.
    """
    self._check_sequence(test_sequence)
    return self._decode_batch([test_sequence], args)[0]

  def online(self, num_utterances, args, max_frames, persistent=False):
    """An OnlineSession (streaming decode; extension, see the class)."""
    if args.look_ahead != 1:
      raise ValueError('online decoding needs look_ahead 1')
    if self.transition_bias is None:
      raise TypeError('transition_bias is None: the model was never fit or loaded')
    return OnlineSession(self, num_utterances, args, max_frames, persistent)

  def predict(self, test_sequences, args):
    """Predict labels for one sequence or a list of them (uisrnn.py:564-590).

    A list is decoded as ONE lock-step batch on the GPU (the reference loops
    over it serially); the results are the same as calling predict_single on
    each element.

    Raises:
      TypeError: test_sequences is neither a list nor a numpy array.
      EmptyBeamError: non-finite scores emptied an utterance's beam (the reference raises
        ValueError / IndexError there; this is both).
      LookAheadWindowError: look_ahead >= 2 only -- some utterances had more live assignment
        prefixes inside a window than the device tables hold (a limit the reference does not
        have); the exception names them and carries the other utterances' results.
    """
    if isinstance(test_sequences, np.ndarray):
      return self.predict_single(test_sequences, args)
    if isinstance(test_sequences, list):
      for test_sequence in test_sequences:
        self._check_sequence(test_sequence)
      if not test_sequences:
        return []
      return self._decode_batch(test_sequences, args)
    raise TypeError('test_sequences should be either a list or numpy array.')


class OnlineSession:
  """Online (streaming) diarization of a fixed set of utterances.

  Not part of the reference's API -- google/uis-rnn only decodes offline, replaying the
  utterance `test_iteration` times (uisrnn/arguments.py:186-193) -- but the model is an
  online one.  A session keeps the beam on the GPU; frames are pushed as they arrive and
  `labels()` returns the currently best hypothesis for everything received.  Equivalent,
  bit for bit, to `predict` with test_iteration=1, look_ahead=1 on the frames received so
  far, whatever the chunking.

    with model.online(num_utterances=2, args=inference_args, max_frames=10000) as session:
      session.push([chunk_a, None])        # [n, D] float arrays; None = nothing new
      session.push([chunk_a2, chunk_b])
      labels = session.labels()            # list of lists of ints, one per utterance

  persistent=True (UIS_FLAG_PERSISTENT): the decode kernel stays on the GPU between pushes and is
  fed through a mailbox in pinned host memory -- the lowest push latency (no launch, no copy
  engine), at the price of occupying the whole device until the session closes or has been idle
  for UIS_PERSIST_IDLE_MS (default 50 ms).  Where the model's shape does not allow it the session
  silently uses ordinary launches.
  """

  def __init__(self, model, num_utterances, args, max_frames, persistent=False):
    self._model = model
    self._decoder = _capi.Decoder(model.params, model.device_index)  # own handle: one session per handle
    cap = _initial_cluster_cap(args)
    self.persistent = False
    if persistent:
      try:
        self._decoder.stream_begin(num_utterances, args.beam_size, max_frames, max_clusters=cap,
                                   flags=_capi.UIS_FLAG_PERSISTENT)
        self.persistent = True
      except _capi.HipLibraryError as err:
        if err.status != _capi.UIS_ERR_UNSUPPORTED:
          raise
    if not self.persistent:
      self._decoder.stream_begin(num_utterances, args.beam_size, max_frames, max_clusters=cap)
    self._open = True

  def push(self, chunks):
    """chunks: a list with one [n, D] float64 array (or None) per utterance, or one [U, n, D]
    float64 array when every utterance received the same number of frames."""
    if isinstance(chunks, np.ndarray) and chunks.ndim == 3:
      if chunks.dtype != float:
        raise TypeError('test_sequence should be a numpy array of float type.')
    else:
      # (round 6: one pass over the dtypes -- the reference's TypeError for anything but float64, uisrnn.py:511-513 --
      # the shapes are checked by the one concatenate in _capi.stream_push; the full per-chunk check of the
      # reference's messages only when something is off)
      try:
        plain = {c.dtype for c in chunks} == {np.dtype(float)}
      except AttributeError:
        plain = False
      if not plain:
        for chunk in chunks:
          if chunk is not None and len(chunk):
            self._model._check_sequence(np.asarray(chunk))
      try:
        self._decoder.stream_push(chunks)
      except ValueError:
        for chunk in chunks:   # (which chunk, in the reference's words)
          if chunk is not None and len(chunk):
            self._model._check_sequence(np.asarray(chunk))
        raise
      return
    self._decoder.stream_push(chunks)

  def labels(self):
    per_utt, _, overflow, _ = self._decoder.stream_labels()
    if overflow.any():
      raise RuntimeError('utterance(s) {} need more than max_clusters clusters per hypothesis; '
                         'open the session with a larger args.max_clusters'.format(
                             np.flatnonzero(overflow).tolist()))
    return [x.tolist() for x in per_utt]

  def close(self):
    if self._open:
      self._decoder.stream_end()
      self._decoder.close()
      self._open = False

  def __enter__(self):
    return self

  def __exit__(self, *exc):
    self.close()


def parallel_predict(model, test_sequences, args, num_processes=4, devices=None):
  """Drop-in for uisrnn.parallel_predict (uisrnn/uisrnn.py:593-623).

  The reference maps utterances over a forkserver process pool.  Here the
  utterances of one GPU are already decoded concurrently in one batch, so the
  workers are GPUs: the list is sharded (longest-processing-time) over
  min(num_processes, visible GPUs) devices, one decoder handle and one host
  thread per device (the C call releases the GIL), no communication during
  decode.  With one GPU this is model.predict.

  Args:
    devices: optional explicit list of HIP device indices (repeats allowed --
      used by the tests to exercise the path on a single-GPU box).

  Raises:
    TypeError: test_sequences is not a list.
  """
  if not isinstance(test_sequences, list):
    raise TypeError('test_sequences must be a list.')
  for test_sequence in test_sequences:
    model._check_sequence(test_sequence)  # pylint: disable=protected-access
  if not test_sequences:
    return []
  if devices is None:
    n_dev = max(_capi.load_library().uis_device_count(), 1)
    devices = list(range(max(1, min(int(num_processes), n_dev))))
  if len(devices) == 1:
    return model._decode_batch(test_sequences, args, device=devices[0])  # pylint: disable=protected-access
  import threading  # pylint: disable=import-outside-toplevel
  from uisrnn_amd import distributed  # pylint: disable=import-outside-toplevel
  shards = distributed.shard_utterances(
      [s.shape[0] for s in test_sequences], len(devices))
  # one handle per worker, created up front on this thread (handles are not thread-safe, so
  # a repeated device index gets its own)
  workers = [_Worker(model, device, slot) for slot, device in enumerate(devices)]
  results = [None] * len(test_sequences)
  errors = []

  def run(worker, shard):
    try:
      if shard:
        out = worker.decode([test_sequences[i] for i in shard], args)
        for i, labels in zip(shard, out):
          results[i] = labels
    except Exception as exc:  # pylint: disable=broad-except
      errors.append(exc)

  threads = [threading.Thread(target=run, args=(w, sh))
             for w, sh in zip(workers, shards)]
  for thread in threads:
    thread.start()
  for thread in threads:
    thread.join()
  for worker in workers:
    worker.close()
  if errors:
    raise errors[0]
  return results


class _Worker:
  """One decoder handle for one parallel_predict worker thread."""

  def __init__(self, model, device, slot):
    self._model = model
    # the first worker on the model's own device shares the model's handle; every other
    # worker (another device, or the same device again) owns a private one
    self._own = slot > 0 or device != model.device_index
    self._decoder = (_capi.Decoder(model.params, device) if self._own
                     else model._get_decoder(device))  # pylint: disable=protected-access

  def decode(self, sequences, args):
    return self._model._decode_batch(  # pylint: disable=protected-access
        sequences, args, decoder=self._decoder)

  def close(self):
    if self._own:
      self._decoder.close()
