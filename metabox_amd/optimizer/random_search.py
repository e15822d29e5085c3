"""Random_search — NP = 100 uniform samples per step, the normaliser of the AEI metric
(reference: src/optimizer/random_search.py:5-58; baseline use src/logger.py:94-120).  The sampling and evaluation run
in the batched kernel (metabox_amd/csrc/mbx_rs.hpp); ``run_episode`` is the B = 1 view, ``run_batch`` steps many
(problem x run) pairs at once."""
import numpy as np
import torch

from .._abi import ALGO_RANDOM_SEARCH
from .basic_optimizer import Basic_Optimizer


class Random_search(Basic_Optimizer):
    def __init__(self, config):
        super().__init__(config)
        self.__config = config
        self.__NP = 100
        self.log_index = None
        self.cost = None
        self.log_interval = config.log_interval

    def make_batch(self, suite, problem_idx, seeds, early_stop=True):
        from ..suite import Batch
        c = self.__config
        return Batch(suite, ALGO_RANDOM_SEARCH, problem_idx, seeds, self.__NP, c.maxFEs, c.log_interval, c.n_logpoint,
                     early_stop=early_stop)

    def run_batch(self, suite, problem_idx, seeds):
        """-> dict of device tensors (cost [B, n_logpoint+1] padded, fes [B], cost_len [B])."""
        c = self.__config
        batch = self.make_batch(suite, problem_idx, seeds)
        batch.reset()
        for _ in range(-(-(c.maxFEs - self.__NP) // self.__NP)):
            batch.step(None)
        res = batch.results()
        torch.cuda.synchronize()
        batch.close()
        return res

    def run_episode(self, problem):
        problem.reset()
        suite = problem._bound_suite()
        seed = int(np.random.randint(0, 2 ** 31 - 1)) * 2654435761 + int(np.random.randint(0, 2 ** 31 - 1))
        res = self.run_batch(suite, [problem._suite_index], [seed])
        n = int(res['cost_len'][0].item())
        self.cost = [float(v) for v in res['cost'][0, :n].cpu().numpy()]
        return {'cost': self.cost, 'fes': int(res['fes'][0].item())}
