"""Multi-rank path on CPU: gloo, world_size 2 (the GPU path differs only in backend)."""

import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

import golden_util
from uisrnn_amd import distributed

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_utterances_balanced_and_complete():
  lengths = [500, 20, 300, 300, 7, 480, 60, 250]
  shards = distributed.shard_utterances(lengths, 3)
  assert sorted(i for s in shards for i in s) == list(range(len(lengths)))
  loads = [sum(lengths[i] for i in s) for s in shards]
  assert max(loads) - min(loads) <= max(lengths)
  assert distributed.shard_utterances(lengths, 3) == shards  # deterministic
  assert distributed.shard_utterances([5], 4) == [[0], [], [], []]
  assert distributed.shard_utterances([], 2) == [[], []]


def _free_port():
  sock = socket.socket()
  sock.bind(('127.0.0.1', 0))
  port = sock.getsockname()[1]
  sock.close()
  return port


def _worker(rank, world, port, case_name, queue):
  sys.path.insert(0, ROOT)
  sys.path.insert(0, os.path.join(ROOT, 'tests'))
  import torch.distributed as dist
  from oracle import oracle  # stands in for the GPU decode in this CPU test
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  dist.init_process_group('gloo', rank=rank, world_size=world)
  case = golden_util.load_case(case_name)
  run = case['runs'][0]

  def decode_fn(seqs):
    out = oracle.decode(case['params'], seqs, run['beam_size'], run['look_ahead'],
                        run['test_iteration'])
    return [l.tolist() for l in out['labels']]

  labels = distributed.predict_sharded(decode_fn, case['seqs'])
  queue.put((rank, labels))
  dist.barrier()
  dist.destroy_process_group()


def test_predict_sharded_gloo_world2():
  case_name = 'tiny_d16'
  case = golden_util.load_case(case_name)
  ctx = mp.get_context('spawn')
  queue = ctx.Queue()
  port = _free_port()
  procs = [ctx.Process(target=_worker, args=(r, 2, port, case_name, queue))
           for r in range(2)]
  for p in procs:
    p.start()
  results = dict(queue.get(timeout=120) for _ in range(2))
  for p in procs:
    p.join(timeout=60)
    assert p.exitcode == 0
  want = [l.tolist() for l in case['runs'][0]['labels']]
  assert results[0] == want and results[1] == want  # every rank has everything


def test_predict_sharded_single_process():
  from oracle import oracle
  case = golden_util.load_case('tiny_d16')
  run = case['runs'][1]
  labels = distributed.predict_sharded(
      lambda seqs: [l.tolist() for l in oracle.decode(
          case['params'], seqs, run['beam_size'], run['look_ahead'],
          run['test_iteration'])['labels']],
      case['seqs'], rank=0, world_size=1)
  assert labels == [l.tolist() for l in run['labels']]
