"""Training entry point (reference: src/trainer.py:59-187): epochs over the shuffled train set, one ``PBO_Env`` per
problem, ``agent.train_episode(env)``, ``.npy`` logs.  Training keeps the reference's single-environment semantics
(the environment is the B = 1 view of the fused kernels); figures are not produced."""
import os
import pickle

import numpy as np

from . import agent as _agents
from . import optimizer as _optimizers
from .environment import PBO_Env
from .tester import _lookup
from .utils import construct_problem_set


class Trainer(object):
    def __init__(self, config):
        self.config = config
        if config.resume_dir is None:
            self.agent = _lookup(_agents, config.train_agent)(config)
        else:
            with open(config.resume_dir + config.train_agent + '.pkl', 'rb') as f:
                self.agent = pickle.load(f)
            self.agent.update_setting(config)
        self.optimizer = _lookup(_optimizers, config.train_optimizer)(config)
        self.train_set, self.test_set = construct_problem_set(config)

    def save_log(self, epochs, steps, cost, returns, normalizer):
        log_dir = self.config.log_dir + f'/train/{self.agent.__class__.__name__}/{self.config.run_time}/log/'
        os.makedirs(log_dir, exist_ok=True)
        np.save(log_dir + 'return', np.stack((steps, returns), 0))
        for problem in self.train_set.data:
            name = str(problem)
            if len(cost[name]) == 0:
                continue
            while len(cost[name]) < len(epochs):
                cost[name].append(cost[name][-1])
                normalizer[name].append(normalizer[name][-1])
            np.save(log_dir + name + '_cost', np.stack((epochs, cost[name], normalizer[name]), 0))

    def train(self, max_epochs=None):
        exceed, epoch = False, 0
        cost_record = {str(p): [] for p in self.train_set.data}
        normalizer_record = {str(p): [] for p in self.train_set.data}
        return_record, learn_steps, epoch_steps = [], [], []
        while not exceed:
            learn_step = 0
            self.train_set.shuffle()
            for problem in self.train_set:
                env = PBO_Env(problem, self.optimizer)
                exceed, info = self.agent.train_episode(env)
                name = str(problem)
                learn_step = info['learn_steps']
                cost_record[name].append(info['gbest'])
                normalizer_record[name].append(info['normalizer'])
                return_record.append(info['return'])
                learn_steps.append(learn_step)
                if exceed:
                    break
            self.agent.train_epoch()
            epoch_steps.append(learn_step)
            self.save_log(epoch_steps, learn_steps, cost_record, return_record, normalizer_record)
            epoch += 1
            if max_epochs is not None and epoch >= max_epochs:
                break
        return {'epochs': epoch, 'learn_steps': learn_steps, 'returns': return_record}
