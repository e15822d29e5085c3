"""Shared helpers of the test-suite (problem lookup, tolerances)."""
import functools
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')

# Tolerance of every floating-point parity check on costs (north star: 1e-5 relative on fitness).
# Costs are (f_raw + bias) - optimum with bias up to 2500, i.e. quantised at ulp(bias) <= 4.6e-13, so a
# relative bound alone is meaningless next to the 1e-8 stop threshold: the absolute floor is 8 ulp(2500).
RTOL = 1e-5
ATOL = 8 * 4.6e-13


def close(a, b, rtol=RTOL, atol=ATOL):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.all(np.abs(a - b) <= atol + rtol * np.abs(b))


@functools.lru_cache(maxsize=None)
def problems(suite, dim):
    """{func_id: problem} for train+test of `suite` at `dim` (host-side instance generator)."""
    from metabox_amd.problem.bbob import BBOB_Dataset
    tr, te = BBOB_Dataset.get_datasets(suite, dim, 5.0)
    return {p.func_id: p for p in tr.data + te.data}


def load(name):
    return np.load(os.path.join(GOLDEN, name))


def fake_results(rs, names, problem_names):
    """Synthetic test.pkl-shaped results dict (schema of src/tester.py:123-127) drawn from RandomState `rs`; shared by
    tools/gen_golden.py (which feeds it to the reference's metric functions) and the metric tests."""
    d = {'cost': {}, 'fes': {}, 'T0': 31.25, 'T1': {}, 'T2': {}}
    for a in names:
        d['T1'][a] = float(rs.uniform(5, 20))
        d['T2'][a] = float(d['T1'][a] + rs.uniform(50, 900))
    for p in problem_names:
        d['cost'][p], d['fes'][p] = {}, {}
        for a in names:
            c = np.sort(rs.lognormal(0, 2, size=(51, 51)), axis=1)[:, ::-1] * rs.uniform(0.1, 10)
            d['cost'][p][a] = [list(r) for r in c]
            d['fes'][p][a] = [float(v) for v in rs.randint(5000, 20001, size=51)]
    return d


METRIC_PROBLEMS = ['Sphere', 'Schwefel', 'Rastrigin_F15']
METRIC_AGENTS = ['RLEPSO_Agent', 'LDE_Agent', 'DEAP_CMAES', 'Random_search']


def metric_inputs():
    rs = np.random.RandomState(99)
    test = fake_results(rs, METRIC_AGENTS, METRIC_PROBLEMS)
    rand = fake_results(rs, ['Random_search'], METRIC_PROBLEMS + ['Ellipsoidal'])
    return test, rand


def fake_rollout(seed, agent='RLEPSO_Agent', n_problems=18, n_cp=21, runs=5, trend=1.0):
    """Synthetic rollout.pkl-shaped dict (schema of src/tester.py:282-310): returns improve with the checkpoint index."""
    rs = np.random.RandomState(seed)
    out = {'cost': {}, 'fes': {}, 'return': {}}
    for p in range(n_problems):
        name = f'problem_{p}'
        base = rs.uniform(-50, 50)
        out['return'][name] = {agent: [[float(base + trend * 4 * c + rs.normal(0, 3)) for _ in range(runs)] for c in range(n_cp)]}
        out['cost'][name] = {agent: [[[1.0] * 51 for _ in range(runs)] for _ in range(n_cp)]}
        out['fes'][name] = {agent: [[20000.0] * runs for _ in range(n_cp)]}
    return out
