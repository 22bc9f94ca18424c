#!/usr/bin/env python3
"""Decode-path counterpart of the reference's demo.py (BASELINE.json configs[0]).

The reference demo trains on data/toy_training_data.npz, then for every test
utterance calls model.predict, compute_sequence_match_accuracy and output_result
(demo.py:26-79).  The toy .npz files are not part of the reference snapshot and
training is out of scope here, so this script
  * writes a toy TEST set in the reference's schema (demo.py:38-43: an .npz with
    object arrays `test_sequences` ([N_i, 256] float64) and `test_cluster_ids`
    (lists of str)) unless --test_data points at an existing one,
  * uses a checkpoint written by the reference's model.save() if --model is given,
    else the closed-form tracker weights of uisrnn_amd.synth,
  * and then follows the reference's predict / accuracy / summary flow.

  python demo.py [--model saved_model.uisrnn] [--test_data toy_testing_data.npz] [reference flags]
"""

import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import uisrnn_amd as uisrnn  # noqa: E402
from uisrnn_amd import synth  # noqa: E402


def write_toy_test_set(path, num_utterances=10, dim=256, seed=123):
  """A test set in the reference's .npz schema."""
  rng = np.random.default_rng(seed)
  seqs = np.empty(num_utterances, dtype=object)
  ids = np.empty(num_utterances, dtype=object)
  for u in range(num_utterances):
    seq, spk = synth.make_utterance(seed + u, int(rng.integers(80, 200)), dim)
    seqs[u] = seq
    ids[u] = ['{}_{}'.format(u, s) for s in spk]
  np.savez(path, test_sequences=seqs, test_cluster_ids=ids)
  return path


def main():
  own = argparse.ArgumentParser(add_help=False)
  own.add_argument('--model', default=None)
  own.add_argument('--test_data', default=None)
  own_args, rest = own.parse_known_args()
  model_args, _, inference_args = uisrnn.parse_arguments(rest)

  test_path = own_args.test_data
  if test_path is None:
    test_path = os.path.join(ROOT, 'toy_testing_data.npz')
    if not os.path.exists(test_path):
      write_toy_test_set(test_path, dim=model_args.observation_dim)
  test_data = np.load(test_path, allow_pickle=True)
  test_sequences = test_data['test_sequences'].tolist()
  test_cluster_ids = [list(ids) for ids in test_data['test_cluster_ids'].tolist()]

  model = uisrnn.UISRNN(model_args)
  if own_args.model:
    model.load(own_args.model)
  else:
    model.load_params(synth.tracker_params(
        model_args.observation_dim, model_args.rnn_hidden_size, model_args.rnn_depth))

  # the reference predicts one utterance at a time; a list is one GPU batch here
  # and the accuracy step (demo.py:61-64) runs on the labels while they are still in HBM
  predicted, accuracies = model.predict_and_evaluate(test_sequences, test_cluster_ids, inference_args)
  test_record = []
  for truth, labels, accuracy in zip(test_cluster_ids, predicted, accuracies):
    assert accuracy == uisrnn.compute_sequence_match_accuracy(truth, labels)  # host cross-check
    test_record.append((accuracy, len(truth)))  # demo.py:61-64
    print('Ground truth labels: {} ...'.format(truth[:8]))
    print('Predicted labels:    {} ...'.format(labels[:8]))
    print('accuracy {:.4f}  ({} frames)'.format(accuracy, len(labels)))
    print('-' * 60)
  print('Config: beam_size={} look_ahead={} test_iteration={} observation_dim={} '
        'rnn_hidden_size={} rnn_depth={}'.format(
            inference_args.beam_size, inference_args.look_ahead,
            inference_args.test_iteration, model_args.observation_dim,
            model_args.rnn_hidden_size, model_args.rnn_depth))
  print('Performance: averaged accuracy {:.6f}, accuracy numbers for all testing '
        'sequences: {}'.format(float(np.mean(accuracies)),
                               ' '.join('{:.4f}'.format(a) for a in accuracies)))
  # the reference's closing summary (demo.py:68-70; also appended to layer_*_result.txt)
  print(uisrnn.output_result(model_args, uisrnn.parse_arguments(rest)[1], test_record))
  stats = model.last_stats or {}
  print('Decode: {:.2f} ms on device, {} rnn rows'.format(
      stats.get('decode_ms', 0.0), stats.get('rnn_rows', 0)))


if __name__ == '__main__':
  main()
