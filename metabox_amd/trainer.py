"""Training entry point (reference: src/trainer.py:59-187): epochs over the shuffled train set, one ``PBO_Env`` per
problem, ``agent.train_episode(env)``, ``.npy`` logs.  Training keeps the reference's single-environment semantics
(the environment is the B = 1 view of the fused kernels); figures are not produced."""
import os
import pickle

import numpy as np

from . import agent as _agents
from . import optimizer as _optimizers
from .environment import PBO_Env
from .tester import _lookup
from .utils import construct_problem_set


class Trainer(object):
    def __init__(self, config):
        self.config = config
        if config.resume_dir is None:
            self.agent = _lookup(_agents, config.train_agent)(config)
        else:
            with open(config.resume_dir + config.train_agent + '.pkl', 'rb') as f:
                self.agent = pickle.load(f)
            self.agent.update_setting(config)
        self.optimizer = _lookup(_optimizers, config.train_optimizer)(config)
        self.train_set, self.test_set = construct_problem_set(config)

    def save_log(self, epochs, steps, cost, returns, normalizer):
        log_dir = self.config.log_dir + f'/train/{self.agent.__class__.__name__}/{self.config.run_time}/log/'
        os.makedirs(log_dir, exist_ok=True)
        np.save(log_dir + 'return', np.stack((steps, returns), 0))
        for problem in self.train_set.data:
            name = str(problem)
            if len(cost[name]) == 0:
                continue
            while len(cost[name]) < len(epochs):
                cost[name].append(cost[name][-1])
                normalizer[name].append(normalizer[name][-1])
            np.save(log_dir + name + '_cost', np.stack((epochs, cost[name], normalizer[name]), 0))

    def train_batched(self, max_epochs=None):
        """--train_batch_size > 1: every epoch is ONE lock-step batch of (train problem x train_batch_size runs) instances
        driven by ``agent.train_batch`` (vectorised PPO; the reference accepts the flag but never batches,
        src/trainer.py:159-161).  Instances are sharded over ranks when torch.distributed is initialised and gradients are
        averaged, so every rank holds the same policy."""
        import numpy as np
        from .distributed import instance_table, partition_bounds, philox_seed, shard_range
        from .environment import BatchedPBO_Env
        from .suite import Suite
        from .tester import _world
        problems = self.train_set.data
        suite = Suite(problems)
        rank, world = _world()
        exceed, epoch, log = False, 0, []
        while not exceed:
            pidx, run = instance_table(len(problems), self.config.train_batch_size)
            # cost-weighted contiguous split like Tester's run_pairs (distributed.cost_partition): an epoch of lock-step training ends with the
            # slowest rank, and the table is function-sorted, so equal counts would leave the Gallagher / Weierstrass rank ~2x the work
            if len(pidx) < world:                                                    # decided from global quantities only: EVERY rank raises, none is left waiting in an all-reduce
                raise ValueError(f'train_batched: {world} ranks but only {len(pidx)} instances in the epoch, a rank would own none: '
                                 f'raise --train_batch_size or use fewer ranks')
            bounds = partition_bounds(problems, pidx, world)
            if bool(np.any(np.diff(bounds) <= 0)):
                # few instances per rank and cost ratios up to ~2x: the cost midpoints can leave a rank empty although there are enough instances;
                # such epochs train with the equal-count split (every rank owns at least one instance)
                bounds = np.array([shard_range(len(pidx), r, world)[0] for r in range(world)] + [len(pidx)], dtype=np.int64)
            lo, hi = int(bounds[rank]), int(bounds[rank + 1])
            seeds = philox_seed(run, np.arange(len(pidx)), epoch_salt=epoch + 1)
            env = BatchedPBO_Env(problems, self.optimizer, pidx[lo:hi], seeds[lo:hi], suite=suite)
            exceed, info = self.agent.train_batch(env)
            env.close()
            log.append(info)
            epoch += 1
            if max_epochs is not None and epoch >= max_epochs:
                break
        if rank == 0:
            log_dir = self.config.log_dir + f'/train/{self.agent.__class__.__name__}/{self.config.run_time}/log/'
            os.makedirs(log_dir, exist_ok=True)
            np.save(log_dir + 'return', np.stack(([i['learn_steps'] for i in log], [i['return'] for i in log]), 0))
        return {'epochs': epoch, 'learn_steps': [i['learn_steps'] for i in log], 'returns': [i['return'] for i in log]}

    def train(self, max_epochs=None):
        if getattr(self.config, 'train_batch_size', 1) > 1 and hasattr(self.agent, 'train_batch'):
            return self.train_batched(max_epochs)
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            # train_episode is the reference's one-instance loop (src/trainer.py:142-187): nothing synchronises the ranks' parameters there, while the agents'
            # checkpoints are written by rank 0 only (agent/utils.py: save_class, `_mbx_replicated`).  Under a process group the data-parallel path is the only one.
            if hasattr(self.agent, 'train_batch'):
                return self.train_batched(max_epochs)
            raise RuntimeError('Trainer.train: the one-instance train_episode loop is not synchronised across ranks; run it in a single process or use an agent with train_batch')
        exceed, epoch = False, 0
        cost_record = {str(p): [] for p in self.train_set.data}
        normalizer_record = {str(p): [] for p in self.train_set.data}
        return_record, learn_steps, epoch_steps = [], [], []
        while not exceed:
            learn_step = 0
            self.train_set.shuffle()
            for problem in self.train_set:
                env = PBO_Env(problem, self.optimizer)
                exceed, info = self.agent.train_episode(env)
                name = str(problem)
                learn_step = info['learn_steps']
                cost_record[name].append(info['gbest'])
                normalizer_record[name].append(info['normalizer'])
                return_record.append(info['return'])
                learn_steps.append(learn_step)
                if exceed:
                    break
            self.agent.train_epoch()
            epoch_steps.append(learn_step)
            self.save_log(epoch_steps, learn_steps, cost_record, return_record, normalizer_record)
            epoch += 1
            if max_epochs is not None and epoch >= max_epochs:
                break
        return {'epochs': epoch, 'learn_steps': learn_steps, 'returns': return_record}
