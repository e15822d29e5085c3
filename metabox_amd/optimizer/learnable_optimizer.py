"""Base class of learnable backbone optimizers (reference: src/optimizer/learnable_optimizer.py:16-30).

Contract kept from the reference: ``init_population(problem) -> state``, ``update(action, problem) ->
(next_state, reward, is_done)``, public attributes ``fes``, ``cost`` (list, first entry = initial best, one entry
appended whenever ``fes >= log_index * log_interval``, last entry = final best), ``log_index``, ``log_interval``.
Added: ``make_batch(suite, problem_idx, seeds)`` returning the device batch that steps many instances at once.
"""
from typing import Any, Tuple


class Learnable_Optimizer:
    def __init__(self, config):
        self.__config = config

    def init_population(self, problem) -> Any:
        raise NotImplementedError

    def update(self, action: Any, problem) -> Tuple[Any]:
        raise NotImplementedError

    def make_batch(self, suite, problem_idx, seeds, early_stop=True):
        raise NotImplementedError
