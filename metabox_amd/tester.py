"""Testing / rollout harness — batched re-statement of the reference loops (src/tester.py).

``Tester(config).test()``   51 seeded runs x test problems x (agent, optimizer) pairs  (tester.py:180-263)
``rollout(config)``         21 checkpoints x train problems x 5 runs                   (tester.py:266-352)
``test_for_random_search``  Random_search over train+test problems, 51 runs            (tester.py:355-407)
``mgd_test(config)``        zero-shot gap of two saved agents on the target suite      (tester.py:421-497)
``mte_test(config)``        transfer efficiency from two rollout.pkl files (host only)  (tester.py:500-608)

Where the reference runs one (problem, run) episode at a time, here the whole (problem x run) table of a pair is ONE
lock-step batch on the GPU (sharded over ranks when torch.distributed is initialised; rows are all-gathered at the end
of the epoch).  The result dictionaries and the pickle files keep the reference's schema:
``{'cost': {problem: {name: [runs][51]}}, 'fes': {problem: {name: [runs]}}, 'T0', 'T1': {name}, 'T2': {name}}``.

Timing fields: T0 is the reference's host calibration loop.  The fused kernel cannot separate evaluation time from the
rest, so T1 (time inside problem.eval) is reported as 0 and T2 (wall ms per run) as batch wall time / instances.
"""
import copy
import os
import pickle
import time

import numpy as np
import torch

from . import agent as _agents
from . import optimizer as _optimizers
from .distributed import gather_rows, instance_table, pack_rows, partition_bounds, philox_seed, unpack_rows
from .environment import BatchedPBO_Env
from .utils import construct_problem_set

N_COST = 51          # cost rows are padded to 51 entries whatever n_logpoint is (tester.py:204-205, 330-331)


def _lookup(module, name):
    try:
        return getattr(module, name)
    except AttributeError:
        raise NotImplementedError(f'{name} is not part of the accelerated path of this build (SURVEY.md §2)')


def cal_t0(dim, fes):
    """Host speed calibration (tester.py:59-74): 10 repetitions of `fes` tiny numpy operations, ms."""
    T0 = 0
    for _ in range(10):
        start = time.perf_counter()
        for _ in range(fes):
            x = np.random.rand(dim)
            x + x
            x / (x + 2)
            x * x
            np.sqrt(x)
            np.log(x)
            np.exp(x)
        T0 += (time.perf_counter() - start) * 1000
    return T0 / 10


def _world():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def _accel_device(config):
    """The batched harness runs on the GPU whatever `--device` says (the reference's default is 'cpu'): agents follow the batch."""
    if not torch.cuda.is_available():
        raise RuntimeError('the batched test / rollout harness needs an MI355X (torch.cuda.is_available() is False): there is no CPU fallback')
    return torch.device('cuda', torch.cuda.current_device())


def _pad51(row):
    row = list(row)
    while len(row) < N_COST:
        row.append(row[-1])
    return row


def run_pairs(problems, runner, runs, n_instances_cap=0, epoch_salt=0, n_logpoint=50):
    """Run `runner(suite_problems, problem_idx, seeds) -> results dict` over the (problem x run) table, sharded over
    ranks and optionally chunked; returns (cost [N, nlog+1], fes [N], ret [N]) as numpy arrays in table order plus the
    device time of the whole table in ms: the SUM over ranks of each rank's wall time.  Shards are cost-weighted (the rank that owns the
    expensive functions owns the fewest instances), so a per-rank `wall / local instances` depends on which functions the rank happened
    to get; the sum over ranks divided by N is the mean GPU time per instance whatever the number of ranks and however the table was cut
    (with one rank it is the wall time; with balanced shards it is world x the epoch's wall time)."""
    pidx, run = instance_table(len(problems), runs)
    n_total = len(pidx)
    rank, world = _world()
    # cost-weighted contiguous split (distributed.cost_partition): every rank gets the same predicted work, not the same number of instances
    bounds = partition_bounds(problems, pidx, world)
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    seeds = philox_seed(run, np.arange(n_total), epoch_salt)
    cap = n_instances_cap if n_instances_cap and n_instances_cap > 0 else (hi - lo)
    rows = []
    t0 = time.perf_counter()
    for a in range(lo, hi, max(cap, 1)):
        b = min(a + cap, hi)
        rows.append(pack_rows(runner(problems, pidx[a:b], seeds[a:b])))
    torch.cuda.synchronize()
    wall_ms = (time.perf_counter() - t0) * 1000
    dev = torch.device('cuda', torch.cuda.current_device())
    # an empty shard (fewer instances than ranks) still takes part in the all-gather with the right width and device
    local = torch.cat(rows, 0) if rows else torch.zeros(0, n_logpoint + 1 + 3, dtype=torch.float64, device=dev)
    full = unpack_rows(gather_rows(local, n_total, bounds=bounds))
    # the table's device time (the batched engine's T2, see the docstring): sum of the ranks' wall times
    total_ms = wall_ms
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([wall_ms], dtype=torch.float64, device=dev if dist.get_backend() == 'nccl' else 'cpu')
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        total_ms = float(t.item())
    return full['cost'].cpu().numpy(), full['fes'].cpu().numpy(), full['return'].cpu().numpy(), total_ms


def _learnable_runner(agent, optimizer, suite_cache, early_stop=True):
    def run(problems, pidx, seeds):
        from .suite import Suite
        key = id(problems)
        if key not in suite_cache:
            suite_cache[key] = Suite(problems)
        env = BatchedPBO_Env(problems, optimizer, pidx, seeds, early_stop=early_stop, suite=suite_cache[key])
        out = agent.rollout_batch(env)
        env.close()
        return out
    return run


def _random_search_runner(optimizer, suite_cache):
    def run(problems, pidx, seeds):
        from .suite import Suite
        key = id(problems)
        if key not in suite_cache:
            suite_cache[key] = Suite(problems)
        res = optimizer.run_batch(suite_cache[key], pidx, seeds)
        res['return'] = torch.zeros_like(res['fes'])
        return res
    return run


def _fill(results, problems, name, runs, cost, fes):
    for k, p in enumerate(problems):
        rows = slice(k * runs, (k + 1) * runs)
        results['cost'][str(p)][name] = [_pad51(r) for r in cost[rows]]
        results['fes'][str(p)][name] = [float(v) for v in fes[rows]]


class Tester(object):
    def __init__(self, config):
        self.config = config
        self.agent_name_list = list(config.agent_for_cp)
        self.agent_for_cp = []
        for name in config.agent_for_cp:
            with open(config.agent_load_dir + name + '.pkl', 'rb') as f:
                self.agent_for_cp.append(pickle.load(f))
        self.l_optimizer_for_cp = [_lookup(_optimizers, n)(copy.deepcopy(config)) for n in config.l_optimizer_for_cp]
        self.t_optimizer_for_cp, self.skipped = [], []
        for n in config.t_optimizer_for_cp:
            if hasattr(_optimizers, n):
                self.t_optimizer_for_cp.append(getattr(_optimizers, n)(copy.deepcopy(config)))
            else:
                self.skipped.append(n)      # BayesianOptimizer (scikit-optimize), ...: not built
        if config.agent is not None:
            with open(config.agent_load_dir + config.agent + '.pkl', 'rb') as f:
                self.agent_for_cp.append(pickle.load(f))
            self.agent_name_list.append(config.agent)
            self.l_optimizer_for_cp.append(_lookup(_optimizers, config.optimizer)(copy.deepcopy(config)))
        elif config.optimizer is not None:
            self.t_optimizer_for_cp.append(_lookup(_optimizers, config.optimizer)(copy.deepcopy(config)))
        self.log_dir = config.test_log_dir
        os.makedirs(self.log_dir, exist_ok=True)
        if config.problem[-6:] == '-torch':
            config.problem = config.problem[:-6]
        _, self.test_set = construct_problem_set(config)
        self.runs = getattr(config, 'test_runs', 51)
        self.test_results = {'cost': {}, 'fes': {}, 'T0': 0., 'T1': {}, 'T2': {}}
        names = self.agent_name_list + [type(o).__name__ for o in self.t_optimizer_for_cp]
        for n in names:
            self.test_results['T1'][n] = 0.
            self.test_results['T2'][n] = 0.
        for p in self.test_set.data:
            self.test_results['cost'][str(p)] = {n: [] for n in names}
            self.test_results['fes'][str(p)] = {n: [] for n in names}

    def test(self):
        cfg = self.config
        self.test_results['T0'] = cal_t0(cfg.dim, cfg.maxFEs)
        problems = self.test_set.data
        cache = {}
        cap = getattr(cfg, 'n_instances', 0)
        early = not getattr(cfg, 'fixed_horizon', False)
        for name, agent, optimizer in zip(self.agent_name_list, self.agent_for_cp, self.l_optimizer_for_cp):
            if hasattr(agent, 'to'):
                agent.to(_accel_device(cfg))
            cost, fes, _, wall = run_pairs(problems, _learnable_runner(agent, optimizer, cache, early), self.runs, cap, n_logpoint=cfg.n_logpoint)
            _fill(self.test_results, problems, name, self.runs, cost, fes)
            self.test_results['T1'][name] = 0.
            self.test_results['T2'][name] = wall / len(cost)
        for optimizer in self.t_optimizer_for_cp:
            name = type(optimizer).__name__
            cost, fes, _, wall = run_pairs(problems, _random_search_runner(optimizer, cache), self.runs, cap, n_logpoint=cfg.n_logpoint)
            _fill(self.test_results, problems, name, self.runs, cost, fes)
            self.test_results['T1'][name] = 0.
            self.test_results['T2'][name] = wall / len(cost)
        rank, _ = _world()
        random_search_results = test_for_random_search(cfg)
        if rank == 0:
            with open(self.log_dir + 'test.pkl', 'wb') as f:
                pickle.dump(self.test_results, f, -1)
            with open(self.log_dir + 'random_search_baseline.pkl', 'wb') as f:
                pickle.dump(random_search_results, f, -1)
        return self.test_results


def test_for_random_search(config):
    """Random_search over train + test problems with the single key 'Random_search' (tester.py:355-407)."""
    train_set, test_set = construct_problem_set(config)
    problems = (train_set + test_set).data
    optimizer = _optimizers.Random_search(copy.deepcopy(config))
    name = type(optimizer).__name__
    res = {'cost': {}, 'fes': {}, 'T0': cal_t0(config.dim, config.maxFEs), 'T1': {name: 0.}, 'T2': {name: 0.}}
    for p in problems:
        res['cost'][str(p)] = {name: []}
        res['fes'][str(p)] = {name: []}
    runs = getattr(config, 'test_runs', 51)
    cost, fes, _, wall = run_pairs(problems, _random_search_runner(optimizer, {}), runs, getattr(config, 'n_instances', 0), n_logpoint=config.n_logpoint)
    _fill(res, problems, name, runs, cost, fes)
    res['T2'][name] = wall / len(cost)
    return res


def rollout(config):
    """21 checkpoints x train problems x 5 runs -> rollout.pkl with keys cost / fes / return
    [problem][agent][checkpoint] -> list over runs (tester.py:266-352)."""
    if config.problem[-6:] == '-torch':
        config.problem = config.problem[:-6]
    train_set, _ = construct_problem_set(config)
    problems = train_set.data
    runs = getattr(config, 'rollout_runs', 5)
    n_cp = config.n_checkpoint
    results = {'cost': {}, 'fes': {}, 'return': {}}
    for p in problems:
        for k in results:
            results[k][str(p)] = {a: [[] for _ in range(n_cp + 1)] for a in config.agent_for_rollout}
    cache = {}
    for agent_name, opt_name in zip(config.agent_for_rollout, config.optimizer_for_rollout):
        optimizer = _lookup(_optimizers, opt_name)(copy.deepcopy(config))
        for cp in range(n_cp + 1):
            with open(config.agent_load_dir + agent_name + '/checkpoint' + str(cp) + '.pkl', 'rb') as f:
                agent = pickle.load(f)
            if hasattr(agent, 'to'):
                agent.to(_accel_device(config))
            cost, fes, ret, _ = run_pairs(problems, _learnable_runner(agent, optimizer, cache), runs,
                                          getattr(config, 'n_instances', 0), epoch_salt=cp, n_logpoint=config.n_logpoint)
            for k, p in enumerate(problems):
                rows = slice(k * runs, (k + 1) * runs)
                results['cost'][str(p)][agent_name][cp] = [_pad51(r) for r in cost[rows]]
                results['fes'][str(p)][agent_name][cp] = [float(v) for v in fes[rows]]
                results['return'][str(p)][agent_name][cp] = [float(v) for v in ret[rows]]
    rank, _ = _world()
    if rank == 0:
        os.makedirs(config.rollout_log_dir, exist_ok=True)
        with open(config.rollout_log_dir + 'rollout.pkl', 'wb') as f:
            pickle.dump(results, f, -1)
    return results


def name_translate(problem):
    if problem in ['bbob', 'bbob-torch']:
        return 'Synthetic'
    if problem in ['bbob-noisy', 'bbob-noisy-torch']:
        return 'Noisy-Synthetic'
    if problem in ['protein', 'protein-torch']:
        return 'Protein-Docking'
    raise ValueError(problem + ' is not defined!')


def mgd_test(config):
    """Meta-generalisation gap MGD = 100 (1 - AEI_from / AEI_to) of one agent class trained on two suites and tested on
    the target suite (tester.py:421-497).  Both agents share the learnable optimizer named by --optimizer; each
    (problem x 51 runs) table is one lock-step batch.  Returns {'aei', 'aei_std', 'mgd'} and writes the reference's
    two pickles."""
    from .logger import Logger
    _, test_set = construct_problem_set(config)
    problems = test_set.data
    agents = []
    for path in (config.model_from, config.model_to):
        with open(path, 'rb') as f:
            agents.append(pickle.load(f))
    optimizer = _lookup(_optimizers, config.optimizer)(copy.deepcopy(config))
    names = [f'{config.agent}_from', f'{config.agent}_to']
    results = {'cost': {str(p): {n: [] for n in names} for p in problems},
               'fes': {str(p): {n: [] for n in names} for p in problems},
               'T0': cal_t0(config.dim, config.maxFEs), 'T1': {n: 0. for n in names}, 'T2': {n: 0. for n in names}}
    runs = getattr(config, 'test_runs', 51)
    cap = getattr(config, 'n_instances', 0)
    cache = {}
    for name, agent in zip(names, agents):
        if hasattr(agent, 'to'):
            agent.to(_accel_device(config))
        cost, fes, _, wall = run_pairs(problems, _learnable_runner(agent, optimizer, cache), runs, cap, n_logpoint=config.n_logpoint)
        _fill(results, problems, name, runs, cost, fes)
        results['T2'][name] = wall / len(cost)
    baseline = test_for_random_search(config)
    rank, _ = _world()
    if rank == 0:
        os.makedirs(config.mgd_test_log_dir, exist_ok=True)
        with open(config.mgd_test_log_dir + 'test.pkl', 'wb') as f:
            pickle.dump(results, f, -1)
        with open(config.mgd_test_log_dir + 'random_search_baseline.pkl', 'wb') as f:
            pickle.dump(baseline, f, -1)
    aei, aei_std = Logger(config).aei_metric(results, baseline, config.maxFEs)
    mgd = 100 * (1 - aei[names[0]] / aei[names[1]])
    if rank == 0:
        print(f'AEI: {aei}')
        print(f'AEI STD: {aei_std}')
        print(f'MGD({name_translate(config.problem_from)}_{config.difficulty_from}, '
              f'{name_translate(config.problem_to)}_{config.difficulty_to}) of {config.agent}: {mgd}%')
    return {'aei': aei, 'aei_std': aei_std, 'mgd': mgd}


def _checkpoint_returns(path, agent):
    """rollout.pkl -> [checkpoint, problem * run] matrix of returns (problems side by side, tester.py:508-521)."""
    with open(path, 'rb') as f:
        returns = pickle.load(f)['return']
    return np.concatenate([np.asarray(returns[p][agent], dtype=np.float64) for p in returns], axis=1)


def _running_mean(curve):
    """The reference's smoothing loop with smooth = 1 (tester.py:553-571) is the prefix mean of the curve."""
    return np.cumsum(curve) / np.arange(1, len(curve) + 1)


def transfer_efficiency(pretrain, scratch):
    """MTE = 1 - t/T (tester.py:577-590).  T: fraction of training at which the from-scratch curve peaks; t: fraction at
    which the pre-trained curve first reaches that peak (linear interpolation between checkpoints; 1 if it never does,
    0 if it starts above)."""
    n = len(scratch)
    top = int(np.argmax(scratch))
    peak = scratch[top]
    t = 0
    if pretrain[0] < peak:
        for i in range(1, n):
            if pretrain[i - 1] < peak <= pretrain[i]:
                t = ((peak - pretrain[i - 1]) / (pretrain[i] - pretrain[i - 1]) + i - 1) / n
                break
    if pretrain[-1] < peak:
        t = 1
    with np.errstate(divide='ignore', invalid='ignore'):
        return float(1 - np.float64(t) / np.float64(top / n))


def mte_test(config):
    """Meta-transfer efficiency of fine-tuning vs training from scratch, from the two rollout.pkl files named by
    --pre_train_rollout / --scratch_rollout (tester.py:500-608).  Pure host post-processing: checkpoint-mean returns,
    Savitzky-Golay (13, 5), prefix mean, then `transfer_efficiency`.  The reference's figure is replaced by the curves
    it would draw, saved as MTE_<agent>.npz."""
    from scipy.signal import savgol_filter
    agent = config.agent
    pre = _checkpoint_returns(config.pre_train_rollout, agent)
    scr = _checkpoint_returns(config.scratch_rollout, agent)
    runs = getattr(config, 'rollout_runs', 5)
    n_problem = pre.shape[1] // runs

    def band(m):            # mean over problems of the per-problem run std, as standard error
        return m.reshape(m.shape[0], n_problem, runs).std(-1).mean(-1) / np.sqrt(runs)

    curve_pre = _running_mean(savgol_filter(pre.mean(-1), 13, 5))
    curve_scr = _running_mean(savgol_filter(scr.mean(-1), 13, 5))
    mte = transfer_efficiency(curve_pre, curve_scr)
    print(f'MTE({name_translate(config.problem_from)}_{config.difficulty_from}, '
          f'{name_translate(config.problem_to)}_{config.difficulty_to}) of {config.agent}: {mte}')
    os.makedirs(config.mte_test_log_dir, exist_ok=True)
    np.savez(os.path.join(config.mte_test_log_dir, f'MTE_{agent}.npz'), pretrain=curve_pre, scratch=curve_scr,
             pretrain_band=band(pre), scratch_band=band(scr[:, :pre.shape[1]]), mte=mte)
    return mte
