"""Environments of the plugin surface.

``PBO_Env``        — the reference's single-instance glue (src/environment/basic_environment.py:6-22):
                     ``reset() = problem.reset(); optimizer.init_population(problem)``,
                     ``step(a) = optimizer.update(a, problem)``.  Works with any duck-typed optimizer/problem,
                     including user plugins written against MetaBox.
``BatchedPBO_Env`` — B independent (problem x run) pairs stepped in lock-step by one kernel launch
                     (``mbx_reset`` / ``mbx_step``); states / rewards / dones are device tensors.
"""
from typing import Any


class PBO_Env:
    def __init__(self, problem, optimizer):
        self.problem = problem
        self.optimizer = optimizer

    def reset(self):
        self.problem.reset()
        return self.optimizer.init_population(self.problem)

    def step(self, action: Any):
        return self.optimizer.update(action, self.problem)


class BatchedPBO_Env:
    """Lock-step batch of ``PBO_Env``s.

    problems : list of problem objects (one suite, one dimension)
    optimizer: a batched-capable Learnable_Optimizer (``make_batch``), e.g. RLEPSO_Optimizer
    problem_idx[i], seeds[i] : which problem instance i optimises and its Philox key; the result of an instance
    depends only on this pair, never on its position in the batch or on the number of GPUs.
    """

    def __init__(self, problems, optimizer, problem_idx, seeds, early_stop=True, suite=None):
        from ..suite import Suite
        self.problems = list(problems)
        self.suite = suite if suite is not None else Suite(self.problems)
        self.optimizer = optimizer
        self.batch = optimizer.make_batch(self.suite, problem_idx, seeds, early_stop=early_stop)
        self.B = self.batch.B
        self.problem_idx = list(problem_idx)

    @property
    def state_dim(self):
        return self.batch.state_dim

    @property
    def action_dim(self):
        return self.batch.action_dim

    def reset(self):
        return self.batch.reset()

    def step(self, actions):
        return self.batch.step(actions)

    def results(self):
        return self.batch.results()

    def close(self):
        self.batch.close()
