"""Agent utilities (reference: src/agent/utils.py:9-48)."""
import collections
import os
import pickle
import random

import numpy as np
import torch


class Memory:
    def __init__(self):
        self.actions, self.states, self.logprobs, self.rewards = [], [], [], []

    def clear_memory(self):
        del self.actions[:], self.states[:], self.logprobs[:], self.rewards[:]


class ReplayBuffer:
    def __init__(self, max_size):
        self.buffer = collections.deque(maxlen=max_size)

    def append(self, exp):
        self.buffer.append(exp)

    def sample(self, batch_size):
        obs, act, rew, nxt, done = zip(*random.sample(self.buffer, batch_size))
        return (torch.FloatTensor(np.array(obs)), torch.tensor(act), torch.FloatTensor(rew),
                torch.FloatTensor(np.array(nxt)), torch.FloatTensor(done))

    def __len__(self):
        return len(self.buffer)


def save_class(dir, file_name, saving_class):
    """Whole-object pickle checkpoint ``<dir><file_name>.pkl`` (21 files checkpoint0..20 per training run)."""
    os.makedirs(dir, exist_ok=True)
    with open(dir + file_name + '.pkl', 'wb') as f:
        pickle.dump(saving_class, f, -1)
