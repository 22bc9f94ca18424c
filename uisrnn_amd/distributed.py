"""Utterance sharding across the GPUs of a node.

Utterances are independent (the reference's own parallel_predict maps them over
a process pool, uisrnn/uisrnn.py:593-623), so the multi-GPU path is: one process
per GPU, every rank decodes its shard, and ONE collective at the end gathers the
int32 label sequences (torch.distributed: backend "nccl" is RCCL over xGMI on
MI355X; "gloo" on CPU for the tests).  There is no communication during decode.
"""

import numpy as np


def shard_utterances(lengths, world_size):
  """Longest-processing-time assignment of utterances to ranks.

  Decode time is proportional to the LONGEST utterance of a lock-step batch
  plus a per-frame cost, so utterances are dealt longest first to the rank with
  the least work so far.  Deterministic (ties by index).

  Returns:
    list of world_size lists of utterance indices (each in increasing order).
  """
  order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
  load = [0] * world_size
  shards = [[] for _ in range(world_size)]
  for i in order:
    r = min(range(world_size), key=lambda k: (load[k], k))
    shards[r].append(i)
    load[r] += int(lengths[i])
  return [sorted(s) for s in shards]


def predict_sharded(decode_fn, test_sequences, rank=None, world_size=None,
                    device=None):
  """Decode a list of utterances on all ranks and gather every label sequence.

  Args:
    decode_fn: callable(list of [N_i, D] arrays) -> list of label lists; on a
      GPU rank this is `lambda seqs: model.predict(seqs, inference_args)`.
    test_sequences: the FULL list, identical on every rank.
    rank, world_size: default to the initialised torch.distributed group.
    device: torch device of the gather buffers (cuda:<local_rank> for nccl).

  Returns:
    list of label lists for ALL utterances, on every rank.
  """
  import torch  # pylint: disable=import-outside-toplevel
  import torch.distributed as dist  # pylint: disable=import-outside-toplevel
  if world_size is None:
    world_size = dist.get_world_size() if dist.is_initialized() else 1
  if rank is None:
    rank = dist.get_rank() if dist.is_initialized() else 0
  lengths = [int(s.shape[0]) for s in test_sequences]
  shards = shard_utterances(lengths, world_size)
  mine = shards[rank]
  local = decode_fn([test_sequences[i] for i in mine]) if mine else []
  if world_size == 1:
    out = [None] * len(test_sequences)
    for i, lab in zip(mine, local):
      out[i] = list(lab)
    return out
  # one padded int32 buffer per rank: [sum of its utterance lengths]
  sizes = [sum(lengths[i] for i in s) for s in shards]
  width = max(max(sizes), 1)
  buf = torch.full((width,), -1, dtype=torch.int32, device=device)
  if mine:
    flat = np.concatenate([np.asarray(l, dtype=np.int32) for l in local]) \
        if sizes[rank] else np.zeros(0, np.int32)
    buf[:flat.shape[0]] = torch.from_numpy(flat).to(buf.device)
  gathered = torch.empty(world_size * width, dtype=torch.int32, device=device)
  dist.all_gather_into_tensor(gathered, buf)
  gathered = gathered.view(world_size, width).cpu().numpy()
  out = [None] * len(test_sequences)
  for r, shard in enumerate(shards):
    pos = 0
    for i in shard:
      out[i] = gathered[r, pos:pos + lengths[i]].tolist()
      pos += lengths[i]
  return out
