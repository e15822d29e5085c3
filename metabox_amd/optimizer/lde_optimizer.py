"""LDE backbone optimizer — host mirror of the reference class (src/optimizer/lde_optimizer.py:6-198).

Arithmetic runs in the fused gfx950 kernel (metabox_amd/csrc/mbx_lde.hpp).  Constructor side effects follow the
reference (``config.NP = 50, BINS = 5, P_INI = 1, P_NUM_MIN = 2, P_MIN = 2/NP``); ``config.NP_override`` lets
BASELINE.json's pop = 100 configuration be run (the shipped policy weights only fit NP = 50).
"""
import numpy as np
import torch

from .._abi import ALGO_LDE
from .learnable_optimizer import Learnable_Optimizer


class LDE_Optimizer(Learnable_Optimizer):
    def __init__(self, config):
        super().__init__(config)
        self.__config = config
        config.NP = int(getattr(config, 'NP_override', None) or 50)      # lde_optimizer.py:10
        config.BINS = 5
        config.P_INI = 1
        config.P_NUM_MIN = 2
        config.P_MIN = config.P_NUM_MIN / config.NP
        self.__BATCH_SIZE = 1
        self.fes = None
        self.cost = None
        self.log_index = None
        self.log_interval = config.log_interval
        self.gbest_cost = None
        self.__batch = None

    def make_batch(self, suite, problem_idx, seeds, early_stop=True):
        from ..suite import Batch
        c = self.__config
        return Batch(suite, ALGO_LDE, problem_idx, seeds, c.NP, c.maxFEs, c.log_interval, c.n_logpoint,
                     early_stop=early_stop)

    def get_best(self):
        return self.gbest_cost

    def __sync_public(self):
        c = self.__config
        st = self.__batch.read_state(0)
        sc = st[c.NP * c.dim + c.NP + 8:]
        self.gbest_cost = float(sc[0])
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
        return state.cpu().numpy()                      # [1, NP + 10] like the reference's input_net

    def update(self, action, problem):
        a = torch.as_tensor(np.asarray(action, dtype=np.float32).reshape(1, -1)).cuda()
        state, reward, done = self.__batch.step(a)
        torch.cuda.synchronize()
        self.__sync_public()
        return state.cpu().numpy(), reward.cpu().numpy().copy(), bool(done[0].item())
