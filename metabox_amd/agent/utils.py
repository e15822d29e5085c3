"""Small training utilities shared by the agents (reference: src/agent/utils.py:9-48): the PPO rollout memory, the
DQN replay buffer, and the whole-object pickle checkpoints (``checkpoint0.pkl`` .. ``checkpoint20.pkl``)."""
import os
import pathlib
import pickle
import random
from collections import deque

import numpy as np
import torch


class Memory:
    """Per-segment storage of (state, action, log-prob, reward) lists."""
    FIELDS = ('actions', 'states', 'logprobs', 'rewards')

    def __init__(self):
        for f in self.FIELDS:
            setattr(self, f, [])

    def clear_memory(self):
        for f in self.FIELDS:
            getattr(self, f).clear()


class ReplayBuffer:
    """Uniform-sampling FIFO of (obs, action, reward, next_obs, done) transitions."""

    def __init__(self, max_size):
        self.buffer = deque(maxlen=max_size)

    def __len__(self):
        return len(self.buffer)

    def append(self, exp):
        self.buffer.append(exp)

    def sample(self, batch_size):
        cols = list(zip(*random.sample(self.buffer, batch_size)))
        obs, nxt = (torch.FloatTensor(np.array(cols[k])) for k in (0, 3))
        return obs, torch.tensor(cols[1]), torch.FloatTensor(cols[2]), nxt, torch.FloatTensor(cols[4])


def save_class(dir, file_name, saving_class):
    """Pickle `saving_class` to ``<dir><file_name>.pkl`` (dir is used as a string prefix, like the reference does).
    With torch.distributed initialised: an agent class that declares ``_mbx_replicated = True`` (RLEPSO / LDE / GLEET / DE-DDQN: under a process group
    ``Trainer.train`` only ever drives their data-parallel, gradient-synchronised ``train_batch``, so every rank holds the same parameters and reaches a
    checkpoint threshold at the same optimizer step) is written by rank 0 only; any other object (per-rank state: a tabular Q-function, a replay buffer) is
    written by EVERY rank, ranks > 0 under ``<file_name>.rank<r>.pkl``.  Writes go to a temporary file that is
    renamed over the target: a reader never sees a half-written pickle."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_rank() != 0:
        if getattr(saving_class, '_mbx_replicated', False):
            return
        file_name = f'{file_name}.rank{dist.get_rank()}'
    pathlib.Path(dir).mkdir(parents=True, exist_ok=True)
    target = f'{dir}{file_name}.pkl'
    tmp = f'{target}.tmp{os.getpid()}'
    with open(tmp, 'wb') as fh:
        pickle.dump(saving_class, fh, protocol=-1)
    os.replace(tmp, target)
