"""Sequence-match accuracy, the step after predict() in the reference's demo.

Mirrors ``uisrnn.compute_sequence_match_accuracy`` (uisrnn/evals.py:40-73):
the best one-to-one mapping between the two label sets (Hungarian algorithm)
and the fraction of positions that agree under it.
"""

import numpy as np
from scipy import optimize


def get_list_inverse_index(unique_ids):
  """Position of every value of a list of distinct ids: {value: position} (uisrnn/evals.py:20-37;
  the reference's tests call it directly).

  Raises:
    TypeError: unique_ids is not a list.
  """
  if not isinstance(unique_ids, list):
    raise TypeError('unique_ids must be a list')
  return {value: position for position, value in enumerate(unique_ids)}


def compute_sequence_match_accuracy(sequence1, sequence2):
  """Accuracy between two label sequences under the best label permutation.

  Raises:
    TypeError: an argument is not a list.
    ValueError: the sequences are empty or differ in length.
  """
  if not isinstance(sequence1, list) or not isinstance(sequence2, list):
    raise TypeError('sequence1 and sequence2 must be lists')
  if not sequence1 or len(sequence1) != len(sequence2):
    raise ValueError(
        'sequence1 and sequence2 must be non-empty and of the same size')
  uniq1 = sorted(set(sequence1))
  uniq2 = sorted(set(sequence2))
  index1 = get_list_inverse_index(uniq1)
  index2 = get_list_inverse_index(uniq2)
  # a square matrix (zero padded): the optimum equals that of the reference's rectangular one
  size = max(len(uniq1), len(uniq2))
  overlap = np.zeros((size, size), dtype=np.int64)
  for a, b in zip(sequence1, sequence2):
    overlap[index1[a], index2[b]] += 1
  rows, cols = optimize.linear_sum_assignment(-overlap)
  return float(overlap[rows, cols].sum()) / len(sequence1)


def dense_ids(sequence):
  """Labels of any hashable kind -> int32 indices in sorted order (uisrnn/evals.py:58-61)."""
  index = {lab: i for i, lab in enumerate(sorted(set(sequence)))}
  return np.fromiter((index[lab] for lab in sequence), dtype=np.int32, count=len(sequence))


def sequence_match_accuracies_device(decoder, sequences1, sequences2):
  """compute_sequence_match_accuracy for MANY sequence pairs in one launch on the GPU.

  `decoder` is a uisrnn_amd._capi.Decoder (any model: the kernel only needs a device).  One
  workgroup per pair builds the confusion matrix and solves the assignment exactly
  (uisrnn_amd/csrc/uis_eval.hip); the quotient matched / length is formed here in float64
  like uisrnn/evals.py:72, so the values are identical to the host function's.

  Raises the host function's TypeError / ValueError for the same inputs.
  """
  if len(sequences1) != len(sequences2):
    raise ValueError('need as many first as second sequences')
  for seq1, seq2 in zip(sequences1, sequences2):
    if not isinstance(seq1, list) or not isinstance(seq2, list):
      raise TypeError('sequence1 and sequence2 must be lists')
    if not seq1 or len(seq1) != len(seq2):
      raise ValueError(
          'sequence1 and sequence2 must be non-empty and of the same size')
  if not sequences1:
    return []
  lens = np.array([len(s) for s in sequences1], dtype=np.int64)
  offsets = np.zeros(len(lens) + 1, dtype=np.int64)
  offsets[1:] = np.cumsum(lens)
  ids1 = np.concatenate([dense_ids(s) for s in sequences1])
  ids2 = np.concatenate([dense_ids(s) for s in sequences2])
  matched = decoder.eval_matched(ids1, ids2, offsets)
  return [float(m) / int(n) for m, n in zip(matched, lens)]
