"""Base class of MetaBBO agents (reference: src/agent/basic_agent.py:26-42).

``train_episode(env) -> (exceed_max_learning_step, {'normalizer','gbest','return','learn_steps'})``
``rollout_episode(env) -> {'cost','fes','return'}``; plus, new here, ``rollout_batch(env)`` for a
``BatchedPBO_Env`` returning the same three fields for every instance.
"""
from typing import Tuple


class Basic_Agent:
    def __init__(self, config):
        self.__config = config

    def update_setting(self, config):
        pass

    def train_episode(self, env) -> Tuple[bool, dict]:
        raise NotImplementedError

    def rollout_episode(self, env) -> dict:
        raise NotImplementedError

    def rollout_batch(self, env) -> dict:
        raise NotImplementedError

    def train_epoch(self):
        pass
