"""QLPSO agent: tabular Q-learning over 4 states x 4 actions (reference: src/agent/qlpso_agent.py).

Policy: softmax over the Q-row of the state, sampled with ``np.random.choice``.  Training: TD(0) with gamma = 0.8 and a learning
rate decaying linearly from 1 to 0.1 over ``max_learning_step`` updates.  ``rollout_batch`` hands the 4 x 4 table to the step
kernel, which then makes the decisions itself, `chunk` env steps per launch (``mbx_qlpso_rollout``).
"""
import numpy as np
import torch

from .basic_agent import Basic_Agent
from .utils import save_class


class QLPSO_Agent(Basic_Agent):
    def __init__(self, config):
        super().__init__(config)
        config.n_states = 4             # qlpso_agent.py:10-14
        config.n_actions = 4
        config.alpha_max = 1
        config.alpha_decay = True
        config.gamma = 0.8
        self.__config = config
        self.__q_table = np.zeros((config.n_states, config.n_actions))
        self.__alpha = config.alpha_max
        self.__max_learning_step = config.max_learning_step
        self.__global_ls = 0
        self.__cur_checkpoint = 0
        self.__checkpoint()

    def __checkpoint(self):
        if getattr(self.__config, 'agent_save_dir', None):
            save_class(self.__config.agent_save_dir, 'checkpoint' + str(self.__cur_checkpoint), self)
        self.__cur_checkpoint += 1

    @property
    def q_table(self):
        return self.__q_table

    def load_exported_weights(self, npz):
        self.__q_table = np.array(npz['q_table'], dtype=np.float64)
        return self

    def to(self, device):
        self.__config.device = device
        return self

    def update_setting(self, config):
        self.__config.max_learning_step = self.__max_learning_step = config.max_learning_step
        self.__config.agent_save_dir = config.agent_save_dir
        self.__global_ls = 0
        save_class(self.__config.agent_save_dir, 'checkpoint0', self)
        self.__config.save_interval = config.save_interval
        self.__cur_checkpoint = 1

    def __get_action(self, state):
        weights = np.exp(self.__q_table[state])
        return np.random.choice(self.__config.n_actions, size=1, p=weights / weights.sum())

    def train_episode(self, env):
        """TD(0) along one rollout (qlpso_agent.py:40-64): Q[s, a] += alpha (r + gamma max Q[s'] - Q[s, a]); alpha decays linearly
        from alpha_max to 0.1 over max_learning_step updates; the episode is cut when that budget is reached."""
        c, q = self.__config, self.__q_table
        total, state, finished = 0, env.reset(), False
        while not finished:
            action = self.__get_action(state)
            successor, reward, finished = env.step(action)
            total += reward
            q[state][action] += self.__alpha * (reward + c.gamma * q[successor].max() - q[state][action])
            self.__global_ls += 1
            if self.__global_ls >= c.save_interval * self.__cur_checkpoint:
                self.__checkpoint()
            if self.__global_ls >= self.__max_learning_step:
                break
            if c.alpha_decay:
                self.__alpha = c.alpha_max - (c.alpha_max - 0.1) * self.__global_ls / self.__max_learning_step
            state = successor
        summary = {'normalizer': env.optimizer.cost[0], 'gbest': env.optimizer.cost[-1], 'return': total, 'learn_steps': self.__global_ls}
        return self.__global_ls >= self.__max_learning_step, summary

    def rollout_episode(self, env):
        total, state, finished = 0, env.reset(), False
        while not finished:
            state, reward, finished = env.step(self.__get_action(state))
            total += reward
        return {'cost': env.optimizer.cost, 'fes': env.optimizer.fes, 'return': total}

    @torch.no_grad()
    def rollout_batch(self, env, max_steps=None, chunk=256):
        """Whole episodes of a BatchedPBO_Env with the tabular policy inside the step kernel."""
        bc = env.batch.cfg
        if max_steps is None:
            max_steps = bc.max_fes - bc.np                     # every step bills exactly one evaluation
        q = torch.from_numpy(np.ascontiguousarray(self.__q_table, dtype=np.float64)).to(env.batch.device)
        env.reset()
        left = max_steps
        while left > 0:
            env.batch.qlpso_rollout(q, min(chunk, left))
            left -= chunk
        res = env.results()
        return {'cost': res['cost'], 'fes': res['fes'], 'return': res['return'], 'steps': res['steps'], 'cost_len': res['cost_len']}
