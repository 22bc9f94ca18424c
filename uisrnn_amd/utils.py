"""The one utility of the reference's utils.py that sits on the inference side of demo.py.

demo.py:61-70 ends with `uisrnn.output_result(model_args, training_args, test_record)` after the
predictions; the other helpers of uisrnn/utils.py prepare TRAINING data and are out of scope.
"""

import numpy as np


def output_result(model_args, training_args, test_record):
  """Summary text of an experiment, as uisrnn.utils.output_result (uisrnn/utils.py:253-285).

  Args:
    model_args, training_args: namespaces from parse_arguments().
    test_record: list of (accuracy, length) pairs, one per test sequence.

  Returns:
    the summary string; it is also appended to
    'layer_<hidden>_<depth>_<dropout>_result.txt' in the working directory, like the reference.
  """
  accuracies = [record[0] for record in test_record]
  lines = [
      'Config:',
      '  sigma_alpha: {}'.format(training_args.sigma_alpha),
      '  sigma_beta: {}'.format(training_args.sigma_beta),
      '  crp_alpha: {}'.format(model_args.crp_alpha),
      '  learning rate: {}'.format(training_args.learning_rate),
      '  regularization: {}'.format(training_args.regularization_weight),
      '  batch size: {}'.format(training_args.batch_size),
      '',
      'Performance:',
      '  averaged accuracy: {:.6f}'.format(np.mean(accuracies)),
      '  accuracy numbers for all testing sequences:',
  ]
  lines.extend('    {:.6f}'.format(accuracy) for accuracy in accuracies)
  text = '\n'.join(lines) + '\n' + '=' * 80 + '\n'
  filename = 'layer_{}_{}_{:.1f}_result.txt'.format(
      model_args.rnn_hidden_size, model_args.rnn_depth, model_args.rnn_dropout)
  with open(filename, 'a') as handle:
    handle.write(text)
  return text
