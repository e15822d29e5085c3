"""Classic (non-learnable) optimizers implement ``run_episode(problem) -> {'cost': [...], 'fes': int}``
(reference protocol: src/optimizer/basic_optimizer.py:10-15)."""


class Basic_Optimizer:
    def __init__(self, config):
        self.__config = config

    def run_episode(self, problem):
        raise NotImplementedError('run_episode(problem) must be provided by the optimizer')
