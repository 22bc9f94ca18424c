"""Sequence-match accuracy, the step after predict() in the reference's demo.

Mirrors ``uisrnn.compute_sequence_match_accuracy`` (uisrnn/evals.py:40-73):
the best one-to-one mapping between the two label sets (Hungarian algorithm)
and the fraction of positions that agree under it.
"""

import numpy as np
from scipy import optimize


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
  index1 = {lab: i for i, lab in enumerate(uniq1)}
  index2 = {lab: i for i, lab in enumerate(uniq2)}
  # square matrix so that the assignment is total (uisrnn/evals.py pads the same way)
  size = max(len(uniq1), len(uniq2))
  overlap = np.zeros((size, size), dtype=np.int64)
  for a, b in zip(sequence1, sequence2):
    overlap[index1[a], index2[b]] += 1
  rows, cols = optimize.linear_sum_assignment(-overlap)
  return float(overlap[rows, cols].sum()) / len(sequence1)
