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


# ------------------------------------------------------------------------------------------------ branch divergences, proven
# The reference and this build compute the same float64 objective with different instruction sequences (numpy's SIMD transcendentals /
# BLAS vs libm or the kernels' own routines): costs agree to ~1e-15 at first and, because a PSO amplifies perturbations, to 1e-13 .. 1e-10
# late in a 199-generation episode.  An integer-valued output (per_no_improve, hence the re-initialisation mask and fes; reward; done)
# can therefore differ only where the comparison that produces it, `new_cost < c_cost` (rlepso_optimizer.py:225-233), is closer than that
# deviation.  Instead of granting the tests a budget of mismatching generations, the FIRST generation at which the bookkeeping of an episode
# differs is examined: for every particle whose stagnation counter differs, the reference's own margin |new_cost - c_cost| must be no larger
# than twice the deviation between the two implementations' operands (+ 4 ulp), and that deviation must itself be inside the parity
# tolerance.  Everything before that generation must be exact; afterwards the two runs are different (equally valid) trajectories and only
# the float tolerances on gbest / cost curves apply.  tests/golden/rlepso_ties.npz (tools/gen_golden.py rlepso_ties) holds the reference's
# per-generation per_no_improve for every episode and its c_cost for the episodes that need the proof.
TIE_RTOL = 1e-9


def prove_tie_arrays(rp, rn, ref_pni, prev_cc, cur_cc, cur_pni, ledger, who, case, g):
    """rp / rn: the reference side's c_cost before / after the update, ref_pni its counters after; the other arguments are this side's."""
    cur_pni, ref_pni = np.asarray(cur_pni, dtype=np.float64), np.asarray(ref_pni, dtype=np.float64)
    if np.array_equal(cur_pni, ref_pni):
        return True
    rec = []
    for i in np.nonzero(cur_pni != ref_pni)[0]:
        margin = abs(rn[i] - rp[i])
        dev = abs(cur_cc[i] - rn[i]) + abs(prev_cc[i] - rp[i])
        assert dev <= TIE_RTOL * abs(rn[i]) + ATOL, (who, case, g, int(i), 'operands outside the parity tolerance', dev, rn[i])
        assert margin <= 2 * dev + 4 * np.spacing(abs(rn[i])), (who, case, g, int(i), 'a decision differs although the reference margin exceeds the '
                                                                'deviation between the implementations', margin, dev)
        rec.append((int(i), float(margin / np.spacing(abs(rn[i]))), float(dev / np.spacing(abs(rn[i])))))
    ledger.append((who, case, g, rec))
    return False


def prove_tie(ties, case, g, prev_cc, cur_cc, cur_pni, ledger, who):
    """Compare the stagnation counters after update() number g (0-based) with the reference's.  Returns True when they are identical;
    otherwise proves that every differing particle sits on a near-tie (see above), appends a record to `ledger` and returns False."""
    ref_pni = ties[f'{case}/pni'][g + 1]
    if np.array_equal(np.asarray(cur_pni, dtype=np.float64), ref_pni.astype(np.float64)):
        return True
    key = f'{case}/ccost'
    assert key in ties.files, (f'{who}: {case}: bookkeeping differs from the reference at generation {g}, but the fixture holds no reference '
                               f'c_cost for this episode: add it to EXTRA_TIE_CASES (rlepso_ties) / HD_EXTRA_TIE_CASES (rlepso_hd) in tools/gen_golden.py and regenerate that section')
    return prove_tie_arrays(ties[key][g], ties[key][g + 1], ref_pni, prev_cc, cur_cc, cur_pni, ledger, who, case, g)


def print_ledger(ledger):
    for who, case, g, rec in ledger:
        parts = ', '.join(f'particle {i}: reference margin {m:.0f} ulp <= 2 x deviation {d:.0f} ulp' for i, m, d in rec)
        print(f'  [{who}] {case}: first branch divergence at generation {g}: {parts}')


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
