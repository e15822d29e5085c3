"""GLEET backbone optimizer — host mirror of the reference class (src/optimizer/gleet_optimizer.py:6-314).

PSO with ps = 100 whose acceleration c = 4.1 is split per particle by the agent's action between pbest and gbest, inertia from
0.9 down by 0.5 / (maxFEs / ps) per generation, clipping border, reward 100 * (pre_gbest - gbest) / max_cost.  The state is
[ps, 27]: 9 features per particle plus its exploration memory and the swarm's exploitation memory.  All of it is produced by
the fused kernel in metabox_amd/csrc/mbx_gleet.hpp; this class is the reference's plugin protocol over a B = 1 batch.
"""
import numpy as np
import torch

from .._abi import ALGO_GLEET
from .learnable_optimizer import Learnable_Optimizer


class GLEET_Optimizer(Learnable_Optimizer):
    def __init__(self, config):
        super().__init__(config)
        self.__config = config
        self.dim = config.dim
        self.ps = 100                       # gleet_optimizer.py:11-30
        self.c = 4.1
        self.w_decay = True
        self.reward_scale = 100
        self.max_fes = config.maxFEs
        self.boarder_method = 'clipping'
        self.reward_func = 'direct'
        self.fes = None
        self.cost = None
        self.log_index = None
        self.log_interval = config.log_interval
        self.__batch = None

    def make_batch(self, suite, problem_idx, seeds, early_stop=True):
        from ..suite import Batch
        c = self.__config
        return Batch(suite, ALGO_GLEET, problem_idx, seeds, self.ps, c.maxFEs, c.log_interval, c.n_logpoint, early_stop=early_stop)

    def __sync_public(self):
        NP, D = self.ps, self.dim
        sc = self.__batch.read_state(0)[3 * NP * D + 3 * NP + D + 9 * NP + 10:]
        self.fes = int(sc[1])
        self.log_index = int(sc[2])
        self.cost = [float(v) for v in sc[16:16 + int(sc[3])]]

    def init_population(self, problem):
        suite = problem._bound_suite()
        seed = int(np.random.randint(0, 2 ** 31 - 1)) * 2654435761 + int(np.random.randint(0, 2 ** 31 - 1))
        if self.__batch is not None:
            self.__batch.close()
        self.__batch = self.make_batch(suite, [problem._suite_index], [seed])
        state = self.__batch.reset()
        torch.cuda.synchronize()
        self.__sync_public()
        return state[0].cpu().numpy().reshape(self.ps, 27)

    def update(self, action, problem):
        a = torch.as_tensor(np.ascontiguousarray(action, dtype=np.float32).reshape(1, self.ps)).cuda()
        state, reward, done = self.__batch.step(a)
        torch.cuda.synchronize()
        self.__sync_public()
        return state[0].cpu().numpy().reshape(self.ps, 27), float(reward[0].item()), bool(done[0].item())
