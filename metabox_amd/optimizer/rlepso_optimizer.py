"""RLEPSO backbone optimizer — host mirror of the reference class (src/optimizer/rlepso_optimizer.py:6-263).

The arithmetic of ``init_population`` / ``update`` runs in the fused gfx950 kernel (metabox_amd/csrc/
mbx_rlepso.hpp).  This class keeps the reference's constructor side effects (``config.NP = 100``,
``config.w_decay = True``), its public attributes and the single-instance protocol as a B = 1 view over the
batched engine, so agents written against ``env.reset()/env.step()`` keep working.
"""
import numpy as np
import torch

from .._abi import ALGO_RLEPSO
from .learnable_optimizer import Learnable_Optimizer


class RLEPSO_Optimizer(Learnable_Optimizer):
    def __init__(self, config):
        super().__init__(config)
        config.w_decay = True
        if not getattr(config, 'NP_override', None):
            config.NP = 100                              # rlepso_optimizer.py:11
        else:
            config.NP = int(config.NP_override)
        self.__config = config
        self.__dim = config.dim
        self.__NP = config.NP
        self.__n_group = 5                               # rlepso_optimizer.py:26
        self.fes = None
        self.cost = None
        self.log_index = None
        self.log_interval = config.log_interval
        self.__max_fes = config.maxFEs
        self.name = 'EPSO'
        self.__batch = None
        self.__batch_key = None

    # ---- batched engine ---------------------------------------------------------------------------
    def make_batch(self, suite, problem_idx, seeds, early_stop=True):
        from ..suite import Batch
        c = self.__config
        return Batch(suite, ALGO_RLEPSO, problem_idx, seeds, self.__NP, c.maxFEs, c.log_interval, c.n_logpoint,
                     early_stop=early_stop, n_group=self.__n_group)

    # ---- single-instance protocol (B = 1 view) ------------------------------------------------------
    # One env step = the action's H2D copy, one generation kernel, ONE device-to-host copy of the instance's scalar block + cost list
    # (mbx_read_public: 67 doubles).  State, reward and done are derived from those scalars on the host (state = fes / maxFEs,
    # rlepso_optimizer.py:170-171; reward = change of the running return, :251-254), so nothing else crosses PCIe.  A reset for another
    # problem / run re-binds the existing one-instance batch (mbx_batch_rebind) instead of re-creating it.
    def __sync_public(self):
        sc = self.__batch.read_public(0)
        self.fes = int(sc[1])
        self.log_index = int(sc[2])
        n = int(sc[3])
        self.cost = [float(v) for v in sc[16:16 + n]]
        return sc

    def init_population(self, problem):
        suite = problem._bound_suite()
        # the Philox key is drawn from numpy's global stream so that `np.random.seed(run)` (src/tester.py:198)
        # still makes a run reproducible
        seed = int(np.random.randint(0, 2 ** 31 - 1)) * 2654435761 + int(np.random.randint(0, 2 ** 31 - 1))
        if self.__batch is not None and self.__batch_key == id(suite):
            self.__batch.rebind([problem._suite_index], [seed])
        else:
            if self.__batch is not None:
                self.__batch.close()
            self.__batch = self.make_batch(suite, [problem._suite_index], [seed])
            self.__batch_key = id(suite)
            self.__action = torch.zeros(1, self.__n_group * 7, dtype=torch.float32, device=self.__batch.device)
            self.__host_action = torch.zeros(1, self.__n_group * 7, dtype=torch.float32).pin_memory()
        self.__batch.reset()
        sc = self.__sync_public()
        self.__return = float(sc[5])
        return np.array([sc[1] / self.__max_fes])

    def update(self, action, problem):
        a = np.asarray(action, dtype=np.float32).reshape(1, -1)
        assert a.shape[-1] == self.__n_group * 7, 'actions size is not right!'
        self.__host_action.copy_(torch.from_numpy(a))
        self.__action.copy_(self.__host_action, non_blocking=True)
        self.__batch.step(self.__action)
        sc = self.__sync_public()                         # waits for the kernel: the copy is enqueued behind it on the same stream
        reward = float(sc[5]) - self.__return
        self.__return = float(sc[5])
        return np.array([sc[1] / self.__max_fes]), reward, bool(sc[4] != 0.)
