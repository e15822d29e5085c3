"""CPU: the C oracle's RLEPSO restatement replays whole reference episodes.

The numpy draws are regenerated from the seed (legacy MT19937 stream, the reference's call order) and fed
as the per-step tape; the float32 actions come from the fixture.  Expected: gbest trajectory, fes, reward,
done flags and the final cost list of the reference run.
"""
import numpy as np
import pytest

from helpers import close, load, print_ledger, problems, prove_tie
from oracle import oracle

TR = load('rlepso_traces.npz')
TIES = load('rlepso_ties.npz')
CASES = [str(c) for c in TR['cases']]
LEDGER = []


def replay(case):
    suite, fid, seed, _ = case.split('/')
    p = problems(suite, 10)[int(fid)]
    cfg = oracle.make_cfg(1, 100, 10, 20000, 400, 50)
    o = oracle.RlepsoOracle(p.desc(), p.bias, cfg)
    fd = oracle.NumpyTapeFeeder(int(seed), 100, 10, p.noise[0])
    o.reset(fd.reset_tape())
    g0 = oracle.split_rlepso_state(o.state(), 100, 10, 50)['scalars'][oracle.SC_GBEST]
    acts = TR[f'{case}/actions']
    rows = []
    exact_until = len(acts)                      # generations [0, exact_until) have bookkeeping identical to the reference's
    prev = oracle.split_rlepso_state(o.state(), 100, 10, 50)
    for g, a in enumerate(acts):
        s, r, d = o.step(a, fd.step_tape())
        cur = oracle.split_rlepso_state(o.state(), 100, 10, 50)
        sc = cur['scalars']
        fd.commit(sc[oracle.SC_REINIT] > 0)
        rows.append((sc[oracle.SC_GBEST], sc[oracle.SC_FES], r, d))
        if exact_until == len(acts) and not prove_tie(TIES, case, g, prev['ccost'], cur['ccost'], cur['pni'], LEDGER, 'oracle'):
            exact_until = g
        prev = cur
    return g0, np.array(rows), prev, exact_until


@pytest.mark.parametrize('case', CASES)
def test_oracle_replays_reference_episode(case):
    g0, rows, st, m = replay(case)
    assert close(g0, TR[f'{case}/gbest0'], rtol=1e-9)
    assert close(rows[:, 0], TR[f'{case}/gbest'], rtol=1e-9), 'gbest trajectory'
    n = int(st['scalars'][oracle.SC_COST_LEN])
    ref_cost = TR[f'{case}/cost']
    assert n == len(ref_cost)
    assert close(st['cost'][:n], ref_cost, rtol=1e-9)
    # integer-valued outputs: exact up to the first generation whose only difference is a proven near-tie (helpers.prove_tie); the
    # re-initialisation mask of generation m still derives from identical counters, so fes / reward / done are compared through m - 1
    assert np.array_equal(rows[:m, 1], TR[f'{case}/fes'][:m])
    assert np.array_equal(rows[:m, 2], TR[f'{case}/reward'][:m])
    assert np.array_equal(rows[:m, 3].astype(bool), TR[f'{case}/done'][:m])
    if m == len(rows):
        assert np.array_equal(st['pni'], TR[f'{case}/final_pni'])
        # positions after 199 generations: the matvec is an fma chain here and BLAS (its own blocking) in the reference, 1e-16 apart per
        # evaluation; a PSO amplifies that (bbob/15/2, two maps per evaluation: 2.7e-11 with every branch still identical to the reference's)
        assert np.abs(st['pos'].reshape(100, 10) - TR[f'{case}/final_pos']).max() <= 1e-9
        assert close(st['pbest'], TR[f'{case}/final_pbest'], rtol=1e-9)


def test_every_branch_divergence_is_a_proven_near_tie():
    """Runs after the parametrised replays (file order): the ledger lists every episode in which the oracle's bookkeeping leaves the
    reference's, each with the proof that the deciding comparison is closer than the implementations agree."""
    for case in CASES:
        if not any(c == case for _, c, _, _ in LEDGER) and int(TIES[f'{case}/oracle_first_divergence']) >= 0:
            replay(case)                           # (running this test alone)
    print(f'{len(LEDGER)} of {len(CASES)} episodes leave the reference bookkeeping at a proven near-tie:')
    print_ledger(LEDGER)
    assert {c for _, c, _, _ in LEDGER} == {c for c in CASES if int(TIES[f'{c}/oracle_first_divergence']) >= 0}


def test_philox_mode_is_deterministic_and_seed_dependent():
    p = problems('bbob', 10)[8]
    cfg = oracle.make_cfg(1, 100, 10, 20000, 400, 50)
    act = np.full(35, 0.5, dtype=np.float32)
    outs = []
    for seed in (1, 1, 2):
        o = oracle.RlepsoOracle(p.desc(), p.bias, cfg, seed=seed)
        o.reset()
        for _ in range(5):
            o.step(act)
        outs.append(o.state())
    assert np.array_equal(outs[0], outs[1])
    assert not np.array_equal(outs[0], outs[2])
