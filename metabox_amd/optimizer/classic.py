"""Classic baselines of the test harness (reference: src/optimizer/deap_de.py, deap_pso.py, deap_cmaes.py).

``DEAP_DE``     NP = 50, F = 0.5, Cr = 0.5, donors by ``tools.selTournament(pop, 3, tournsize=3)``, replacement inside the sweep.
``DEAP_PSO``    50 particles, phi1 = phi2 = 2, speed limited to +-ub/2, gbest updated inside the sweep.
``DEAP_CMAES``  ``deap.cma.Strategy(centroid=[ub]*dim, sigma=0.5, lambda_=50)`` driven by ``eaGenerateUpdate`` -- the optimizer
                ``get_config`` always appends to ``t_optimizer_for_cp`` as the AEI 'Gap' baseline (src/config.py:104-105).

The class names are the reference's (they are looked up by name), DEAP itself is not used: DE and PSO are written out in the
reference's wrappers, CMA-ES follows the published update equations of deap 1.3.3 (requirements.txt:7).  DEAP is not part of the
reference tree, so no reference traces exist for these three: the kernels are pinned to the C oracle only (parity unpinned with
respect to the reference; DE's tournament uses Python's unseeded ``random`` there, so even the reference is not reproducible).
All arithmetic runs in metabox_amd/csrc/mbx_classic.hpp; ``run_episode`` is the B = 1 view, ``run_batch`` runs many (problem x run)
pairs in lock step.
"""
import numpy as np
import torch

from .._abi import ALGO_CMAES, ALGO_DE, ALGO_PSO
from .basic_optimizer import Basic_Optimizer


class _Classic(Basic_Optimizer):
    _ALGO = None
    _NP = 50

    def __init__(self, config):
        super().__init__(config)
        self._config = config
        self.log_interval = config.log_interval
        self.cost = None
        self.log_index = None

    def make_batch(self, suite, problem_idx, seeds, early_stop=True):
        from ..suite import Batch
        c = self._config
        return Batch(suite, self._ALGO, problem_idx, seeds, self._NP, c.maxFEs, c.log_interval, c.n_logpoint, early_stop=early_stop)

    def _n_steps(self):
        c = self._config
        evals = c.maxFEs if self._ALGO == ALGO_CMAES else c.maxFEs - self._NP        # CMA-ES evaluates nothing at construction
        return -(-evals // self._NP)

    def run_batch(self, suite, problem_idx, seeds):
        """-> dict of device tensors (cost [B, n_logpoint+1] padded, fes [B], cost_len [B], ...)."""
        batch = self.make_batch(suite, problem_idx, seeds)
        batch.reset()
        for _ in range(self._n_steps()):
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


class DEAP_DE(_Classic):
    _ALGO = ALGO_DE

    def __init__(self, config):
        super().__init__(config)
        config.NP = 50                  # deap_de.py:11-13
        config.F = 0.5
        config.Cr = 0.5


class DEAP_PSO(_Classic):
    _ALGO = ALGO_PSO

    def __init__(self, config):
        super().__init__(config)
        config.phi1 = 2.                # deap_pso.py:11-13
        config.phi2 = 2.
        config.population_size = 50


class DEAP_CMAES(_Classic):
    _ALGO = ALGO_CMAES

    def __init__(self, config):
        super().__init__(config)
        config.NP = 50                  # deap_cmaes.py:15
