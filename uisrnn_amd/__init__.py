"""MI355X-native UIS-RNN beam-search decoder.

Drop-in for the inference path of google/uis-rnn: the names re-exported here
are the decode-path subset of uisrnn/__init__.py:26-30.
"""

from uisrnn_amd import arguments
from uisrnn_amd import evals
from uisrnn_amd import uisrnn as _uisrnn
from uisrnn_amd import utils

parse_arguments = arguments.parse_arguments
compute_sequence_match_accuracy = evals.compute_sequence_match_accuracy
output_result = utils.output_result
UISRNN = _uisrnn.UISRNN
parallel_predict = _uisrnn.parallel_predict
OnlineSession = _uisrnn.OnlineSession  # extension: streaming decode
EmptyBeamError = _uisrnn.EmptyBeamError
LookAheadWindowError = _uisrnn.LookAheadWindowError
