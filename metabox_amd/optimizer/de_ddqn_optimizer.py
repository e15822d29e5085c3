"""DE-DDQN backbone optimizer — host mirror of the reference class (src/optimizer/de_ddqn_optimizer.py:7-220).

One env step evaluates ONE trial vector built with the mutation operator the agent chose (rand/1, rand/2,
rand-to-best/2, cur-to-rand/1); the arithmetic, the operator-credit records and the 99-feature state are produced by
the fused kernel in metabox_amd/csrc/mbx_ddqn.hpp.
"""
import numpy as np
import torch

from .._abi import ALGO_DEDDQN
from .learnable_optimizer import Learnable_Optimizer


class DE_DDQN_Optimizer(Learnable_Optimizer):
    def __init__(self, config):
        super().__init__(config)
        config.F = 0.5                  # de_ddqn_optimizer.py:10-14
        config.Cr = 1.0
        config.NP = 100
        config.gen_max = 10
        config.W = 50
        self.__config = config
        self.fes = None
        self.cost = None
        self.log_index = None
        self.log_interval = config.log_interval
        self.__batch = None

    def make_batch(self, suite, problem_idx, seeds, early_stop=True):
        from ..suite import Batch
        c = self.__config
        return Batch(suite, ALGO_DEDDQN, problem_idx, seeds, c.NP, c.maxFEs, c.log_interval, c.n_logpoint, early_stop=early_stop)

    def __sync_public(self):
        st = self.__batch.read_state(0)
        sc = st[self.__batch.sc_off:]
        self.fes = int(sc[1])
        self.log_index = int(sc[2])
        self.cost = [float(v) for v in sc[16:16 + int(sc[3])]]

    def init_population(self, problem):
        suite = problem._bound_suite()
        seed = int(np.random.randint(0, 2 ** 31 - 1)) * 2654435761 + int(np.random.randint(0, 2 ** 31 - 1))
        if self.__batch is not None:
            self.__batch.close()
        self.__batch = self.make_batch(suite, [problem._suite_index], [seed])
        c = self.__config
        self.__batch.sc_off = c.NP * c.dim + c.NP + 2 * c.dim + 8 + 40 + 3 * 160 + 300 + 16
        state = self.__batch.reset()
        torch.cuda.synchronize()
        self.__sync_public()
        return state[0].cpu().numpy()

    def update(self, action, problem):
        if int(action) not in (0, 1, 2, 3):
            raise ValueError('Action error')
        a = torch.tensor([int(action)], dtype=torch.int32).cuda()
        state, reward, done = self.__batch.step(a)
        torch.cuda.synchronize()
        self.__sync_public()
        return state[0].cpu().numpy(), float(reward[0].item()), bool(done[0].item())
