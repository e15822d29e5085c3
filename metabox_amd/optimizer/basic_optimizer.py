"""Base class of classic (non-learnable) optimizers (reference: src/optimizer/basic_optimizer.py:10-15):
``run_episode(problem) -> {'cost': [...], 'fes': int}``."""


class Basic_Optimizer:
    def __init__(self, config):
        self.__config = config

    def run_episode(self, problem):
        raise NotImplementedError
