"""CPU: the C oracle's LDE restatement replays whole reference episodes (numpy draws regenerated from the seed,
torch.randint indices and float32 actions from the fixture)."""
import numpy as np
import pytest

from helpers import close, load, problems
from oracle import oracle

TR = load('lde_traces.npz')
CASES = [str(c) for c in TR['cases']]
# whole reference episodes at BASELINE config 3's geometry (bbob-noisy D = 30; NP = 50 as shipped and NP = 100 with the reference's one population
# literal patched in the generator): tools/gen_golden.py lde_hd, keys suite/dim/NP/fid/seed/mode
HD = load('lde_traces_hd.npz')
HD_CASES = [str(c) for c in HD['cases']]
NP = 50


def lde_case(case):
    """(fixture, suite, dim, NP, fid, seed, actions [G, 2 NP] float32) of a case key of either fixture.  'uniform' episodes of the hd fixture carry no
    actions: they are the seeded draws of tools/gen_golden.py run_lde_episode, regenerated here."""
    k = case.split('/')
    if len(k) == 5:
        return TR, k[0], int(k[1]), 50, int(k[2]), int(k[3]), TR[f'{case}/actions']
    suite, dim, np_, fid, seed = k[0], int(k[1]), int(k[2]), int(k[3]), int(k[4])
    if f'{case}/actions' in HD.files:
        acts = HD[f'{case}/actions']
    else:
        ars = np.random.RandomState(20_000 + seed)
        acts = np.stack([ars.uniform(0, 1, size=(1, 2 * np_)).astype(np.float32)[0] for _ in range(len(HD[f'{case}/r']))])
    return HD, suite, dim, np_, fid, seed, acts


def replay(case, stepper=None):
    F, suite, dim, NP, fid, seed, acts = lde_case(case)
    p = problems(suite, dim)[fid]
    maxfes = 2000 * dim
    cfg = oracle.make_cfg(2, NP, dim, maxfes, maxfes // 50, 50)
    o = oracle.LdeOracle(p.desc(), p.bias, cfg)
    fd = oracle.LdeTapeFeeder(seed, NP, dim, p.noise[0], maxfes)
    s0 = o.reset(fd.reset_tape())
    rows, states = [], {}
    for g, (a, r) in enumerate(zip(acts, F[f'{case}/r'])):
        s, rew, d = o.step(a, fd.step_tape(r))
        sc = oracle.split_lde_state(o.state(), NP, dim, 50)['scalars']
        rows.append((sc[oracle.SC_GBEST], sc[oracle.SC_FES], rew, d))
        states[g] = s
    return s0, np.array(rows), states, oracle.split_lde_state(o.state(), NP, dim, 50), dim


@pytest.mark.parametrize('case', CASES + HD_CASES)
def test_oracle_replays_reference_lde_episode(case):
    TR, NP = lde_case(case)[0], lde_case(case)[3]
    s0, rows, states, st, dim = replay(case)
    check_lde_replay(case, TR, NP, dim, s0, rows, states, st)


def check_lde_replay(case, TR, NP, dim, s0, rows, states, fin):
    """Shared by the oracle test and the HIP test.  Episodes run by the reference are reproducible only while np.argsort (lde_optimizer.py:75, default
    kind: an unstable sort whose implementation depends on the CPU's SIMD level) never meets two EXACTLY equal fitness values: `first_tie_gen` of the hd
    fixture is the first update() the reference entered with such a pair (a collapsed population; -1: never).  Generations before it must match;
    from it on the reference's own row order is unspecified and nothing is compared."""
    key = f'{case}/first_tie_gen'
    m = int(TR[key]) if key in TR.files and int(TR[key]) >= 0 else len(rows)
    assert np.abs(s0 - TR[f'{case}/state0']).max() <= 1e-9
    assert close(rows[:m, 0], TR[f'{case}/gbest'][:m])
    assert np.array_equal(rows[:m, 1], TR[f'{case}/fes'][:m])
    assert np.array_equal(rows[:m, 3].astype(bool), TR[f'{case}/done'][:m])
    ref_r = TR[f'{case}/reward'][:m]
    assert np.all(np.abs(rows[:m, 2] - ref_r) <= 1e-5 * np.abs(ref_r) + 1e-9)
    for row in TR[f'{case}/states']:                                  # LSTM input features at sampled generations
        if int(row[0]) < m:
            assert np.abs(states[int(row[0])] - row[1:]).max() <= 1e-5
    if m == len(rows):
        n = int(fin['scalars'][oracle.SC_COST_LEN])
        assert n == len(TR[f'{case}/cost']) and close(fin['cost'][:n], TR[f'{case}/cost'])
        assert np.abs(fin['pop'].reshape(NP, dim) - TR[f'{case}/final_pop']).max() <= 1e-9
        assert close(fin['fit'], TR[f'{case}/final_fit'])


def test_histogram_matches_numpy():
    rs = np.random.RandomState(0)
    import ctypes as C
    for _ in range(20):
        f = np.sort(rs.lognormal(size=NP) * 10 ** rs.uniform(-6, 6))
        norm = (f - f.min()) / (f.max() - f.min())
        want = np.histogram(norm, 5)[0]
        cfg = oracle.make_cfg(2, NP, 10, 20000, 400, 50)
        # feed through the feature path: build an oracle, overwrite nothing — use numpy restatement of the bin rule
        edges = np.linspace(0, 1, 6)
        idx = np.minimum((norm * 5).astype(int), 4)
        idx[norm < edges[idx]] -= 1
        inc = (norm >= edges[idx + 1]) & (idx != 4)
        idx[inc] += 1
        assert np.array_equal(np.bincount(idx, minlength=5), want)
