"""Command-line / namespace configuration, mirroring uisrnn/arguments.py.

Same three namespaces, same flag names, types and defaults as the reference
(uisrnn/arguments.py:30-205), so code written against
``uisrnn.parse_arguments()`` keeps working.  Only the inference namespace
(beam_size, look_ahead, test_iteration; uisrnn/arguments.py:172-193) and the
model-shape flags are consumed by the decode path; the training flags are
parsed for interface compatibility and otherwise unused here.
"""

import argparse

_DEFAULT_OBSERVATION_DIM = 256


def str2bool(value):
  """'yes/true/t/y/1' -> True, 'no/false/f/n/0' -> False (case-insensitive)."""
  lowered = value.lower()
  if lowered in ('yes', 'true', 't', 'y', '1'):
    return True
  if lowered in ('no', 'false', 'f', 'n', '0'):
    return False
  raise argparse.ArgumentTypeError('Boolean value expected.')


_MODEL_FLAGS = (
    (('--observation_dim',), dict(default=_DEFAULT_OBSERVATION_DIM, type=int,
                                  help='Dimension of the embeddings (d-vectors).')),
    (('--rnn_hidden_size',), dict(default=512, type=int,
                                  help='Hidden units per GRU layer.')),
    (('--rnn_depth',), dict(default=1, type=int, help='Number of GRU layers.')),
    (('--rnn_dropout',), dict(default=0.2, type=float,
                              help='Dropout between GRU layers (training only).')),
    (('--transition_bias',), dict(default=None, type=float,
                                  help='p0 of the paper; None = estimated by training.')),
    (('--crp_alpha',), dict(default=1.0, type=float,
                            help='alpha of the Chinese restaurant process.')),
    (('--sigma2',), dict(default=None, type=float,
                         help='Observation variance; None = estimated by training.')),
    (('--verbosity',), dict(default=3, type=int, help='Logging verbosity.')),
    (('--enable_cuda',), dict(default=True, type=str2bool,
                              help='Kept for compatibility: this decoder always '
                                   'runs on the MI355X.')),
)

_TRAINING_FLAGS = (
    (('--optimizer', '-o'), dict(default='adam', choices=['adam'])),
    (('--learning_rate', '-l'), dict(default=1e-3, type=float)),
    (('--train_iteration', '-t'), dict(default=20000, type=int)),
    (('--batch_size', '-b'), dict(default=10, type=int)),
    (('--num_permutations',), dict(default=10, type=int)),
    (('--sigma_alpha',), dict(default=1.0, type=float)),
    (('--sigma_beta',), dict(default=1.0, type=float)),
    (('--regularization_weight', '-r'), dict(default=1e-5, type=float)),
    (('--grad_max_norm',), dict(default=5.0, type=float)),
    (('--enforce_cluster_id_uniqueness',), dict(default=True, type=str2bool)),
)

_INFERENCE_FLAGS = (
    (('--beam_size', '-s'), dict(default=10, type=int,
                                 help='Beam width of the decode.')),
    (('--look_ahead',), dict(default=1, type=int,
                             help='Frames scored jointly per decode window.')),
    (('--test_iteration',), dict(default=2, type=int,
                                 help='The sequence is decoded this many times '
                                      'back to back; the labels of the last '
                                      'pass are returned.')),
)


def _parser(description, flags):
  parser = argparse.ArgumentParser(description=description, add_help=False)
  for names, kwargs in flags:
    parser.add_argument(*names, **kwargs)
  return parser


def parse_arguments(argv=None):
  """Returns (model_args, training_args, inference_args) namespaces.

  Like the reference this reads sys.argv when argv is None, validates the
  union of all flags, then splits them.
  """
  model_parser = _parser('Model configurations.', _MODEL_FLAGS)
  training_parser = _parser('Training configurations.', _TRAINING_FLAGS)
  inference_parser = _parser('Inference configurations.', _INFERENCE_FLAGS)
  argparse.ArgumentParser(
      parents=[model_parser, training_parser, inference_parser]).parse_args(argv)
  model_args, _ = model_parser.parse_known_args(argv)
  training_args, _ = training_parser.parse_known_args(argv)
  inference_args, _ = inference_parser.parse_known_args(argv)
  return model_args, training_args, inference_args
