"""RL-PSO backbone optimizer — host mirror of the reference class (src/optimizer/rl_pso_optimizer.py:7-148).

One env step moves ONE particle (round robin over NP = 100): inertia w (decremented on every call, as the reference does),
cognitive weight c * rand1, social weight c * action with c = 2.05; velocity and position clipping; one evaluation;
pbest / gbest; reward (pre_cost - new_cost) / (max_cost - gbest).  State = gbest_position | current_position[cur] (2 D
values), action = one float.  The arithmetic lives in metabox_amd/csrc/mbx_rlpso.hpp.
"""
import numpy as np
import torch

from .._abi import ALGO_RLPSO
from .learnable_optimizer import Learnable_Optimizer


class RL_PSO_Optimizer(Learnable_Optimizer):
    def __init__(self, config):
        super().__init__(config)
        config.w_decay = True           # rl_pso_optimizer.py:10-12
        config.c = 2.05
        config.NP = 100
        self.__config = config
        self.fes = None
        self.cost = None
        self.log_index = None
        self.log_interval = config.log_interval
        self.__batch = None

    def make_batch(self, suite, problem_idx, seeds, early_stop=True):
        from ..suite import Batch
        c = self.__config
        return Batch(suite, ALGO_RLPSO, problem_idx, seeds, c.NP, c.maxFEs, c.log_interval, c.n_logpoint, early_stop=early_stop)

    def __sync_public(self):
        c = self.__config
        sc = self.__batch.read_state(0)[3 * c.NP * c.dim + 2 * c.NP + c.dim:]
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
        return state[0].cpu().numpy()

    def update(self, action, problem):
        a = torch.as_tensor(np.asarray(action, dtype=np.float32).reshape(1)).cuda()
        state, reward, done = self.__batch.step(a)
        torch.cuda.synchronize()
        self.__sync_public()
        return state[0].cpu().numpy(), float(reward[0].item()), bool(done[0].item())
