"""Protein docking: dataset split, oracle energy vs the reference's KATs (CPU); HIP energy, RLEPSO and LDE episodes on
protein problems replayed against the reference (GPU)."""
import functools

import numpy as np
import pytest

from helpers import close, load
from oracle import oracle

KAT = load('protein_kat.npz')
TR = load('protein_traces.npz')


@functools.lru_cache(maxsize=None)
def protein():
    from metabox_amd.problem.protein_docking import Protein_Docking_Dataset
    tr, te = Protein_Docking_Dataset.get_datasets('protein', difficulty='easy')
    nxt = float(np.random.rand())
    return {str(p): p for p in tr.data + te.data}, [str(p) for p in tr.data], [str(p) for p in te.data], nxt


def test_dataset_split_matches_reference():
    byid, tr, te, nxt = protein()
    assert tr == [str(x) for x in KAT['train_ids']] and te == [str(x) for x in KAT['test_ids']]
    assert nxt == float(KAT['next_rand'])
    assert len(tr) == 200 and len(te) == 80
    p = byid['1AVX_1']
    assert (p.dim, p.lb, p.ub, p.optimum, p.n_atoms) == (12, -1.5, 1.5, None, 100)
    assert p.q.shape == (100, 100) and p.basis.shape == (12, 300)


def test_oracle_energy_matches_reference():
    byid = protein()[0]
    for key in [k for k in KAT.files if k.startswith('f/')]:
        f = oracle.evaluate(byid[key[2:]].desc(), KAT['x'])
        assert np.all(np.abs(f - KAT[key]) <= 1e-10 * np.abs(KAT[key])), key


def _replay_rlepso_oracle(case):
    _, pid, seed = case.split('/')
    p = protein()[0][pid]
    cfg = oracle.make_cfg(1, 100, 12, 1000, 200, 5)
    o = oracle.RlepsoOracle(p.desc(), None, cfg)
    fd = oracle.NumpyTapeFeeder(int(seed), 100, 12, 0)
    o.reset(fd.reset_tape())
    rows = []
    for a in TR[f'{case}/actions']:
        s, r, d = o.step(a, fd.step_tape())
        sc = oracle.split_rlepso_state(o.state(), 100, 12, 5)['scalars']
        fd.commit(sc[oracle.SC_REINIT] > 0)
        rows.append((sc[0], sc[1], r, d))
    return np.array(rows), oracle.split_rlepso_state(o.state(), 100, 12, 5)


@pytest.mark.parametrize('case', [str(c) for c in TR['cases'] if str(c).startswith('rlepso')])
def test_oracle_rlepso_on_protein(case):
    rows, st = _replay_rlepso_oracle(case)
    assert close(rows[:, 0], TR[f'{case}/gbest'], rtol=1e-9)
    assert np.array_equal(rows[:, 1], TR[f'{case}/fes']) and np.array_equal(rows[:, 3].astype(bool), TR[f'{case}/done'])
    n = int(st['scalars'][3])
    assert n == len(TR[f'{case}/cost']) == 6 and close(st['cost'][:n], TR[f'{case}/cost'], rtol=1e-9)   # n_logpoint = 5


@pytest.mark.parametrize('case', [str(c) for c in TR['cases'] if str(c).startswith('lde')])
def test_oracle_lde_on_protein(case):
    _, pid, seed = case.split('/')
    p = protein()[0][pid]
    cfg = oracle.make_cfg(2, 50, 12, 1000, 200, 5)
    o = oracle.LdeOracle(p.desc(), None, cfg)
    fd = oracle.LdeTapeFeeder(int(seed), 50, 12, 0, 1000)
    s0 = o.reset(fd.reset_tape())
    assert np.abs(s0 - TR[f'{case}/state0']).max() <= 1e-9
    for g, (a, r) in enumerate(zip(TR[f'{case}/actions'], TR[f'{case}/r'])):
        s, rew, d = o.step(a, fd.step_tape(r))
        assert abs(oracle.split_lde_state(o.state(), 50, 12, 5)['scalars'][0] - TR[f'{case}/gbest'][g]) <= 1e-9 * abs(TR[f'{case}/gbest'][g])
        assert d == TR[f'{case}/done'][g]


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
def test_hip_energy_matches_reference_and_oracle():
    from metabox_amd.suite import Suite
    byid = protein()[0]
    keys = [k[2:] for k in KAT.files if k.startswith('f/')]
    s = Suite([byid[k] for k in keys])
    assert s.optimum(0) is None and byid[keys[0]].optimum is None
    X = np.random.RandomState(1).uniform(-1.5, 1.5, size=(130, 12))          # more than one block of rows
    for k, pid in enumerate(keys):
        f = s.eval(k, KAT['x'])
        assert np.all(np.abs(f - KAT[f'f/{pid}']) <= 1e-9 * np.abs(KAT[f'f/{pid}'])), pid
        g = oracle.evaluate(byid[pid].desc(), X)
        assert np.all(np.abs(s.eval(k, X) - g) <= 1e-9 * np.abs(g)), pid
    p = byid[keys[0]]
    assert np.ndim(p.eval(KAT['x'][0])) == 0 and p.eval(KAT['x']).shape == (12,)
    # The kernel stops its pair walk behind the pairs that can reach the 9 A cut-off while the candidate is inside the box (DevProblem::n_close, ordered at upload by the
    # smallest reachable distance); a candidate OUTSIDE the box (mbx_eval takes any x) must walk the whole list.  Rows far outside move atoms by several Angstrom, so pairs
    # beyond n_close do come within the cut-off: the energies must still be the oracle's.  Rows on the box's faces (|x_k| = ub exactly) take the short walk.
    rs = np.random.RandomState(2)
    Xout = np.concatenate([rs.uniform(-4., 4., size=(40, 12)), np.where(rs.uniform(size=(24, 12)) < 0.5, -1.5, 1.5), rs.uniform(-1.5, 1.5, size=(24, 12))])
    Xout[40:64:2, 3] = 1.5000000000000002                                         # one ulp outside on one coordinate
    for k, pid in enumerate(keys):
        g = oracle.evaluate(byid[pid].desc(), Xout)
        assert np.all(np.abs(s.eval(k, Xout) - g) <= 1e-9 * np.abs(g) + 1e-12), pid


@pytest.mark.gpu
def test_suite_rejects_asymmetric_protein_tables_and_odd_atom_counts_work():
    """The energy kernel evaluates the pairs i < j only (the reference's tables are symmetric by construction, protein_docking.py:175-181):
    mbx_suite_create must refuse tables that are not, and the folded pair list must also cover an odd number of atoms (the middle atom's row
    appears once) -- checked against the oracle, which sums all n^2 pairs."""
    from metabox_amd._abi import MbxError
    from metabox_amd.suite import Suite
    byid = protein()[0]
    p = byid[sorted(byid)[0]]

    class Tampered:
        def __init__(self, base, fn):
            self.base, self.fn = base, fn
            self.dim, self.lb, self.ub, self.optimum = base.dim, base.lb, base.ub, None

        def desc(self):
            return self.fn(dict(self.base.desc()))

    def asym(d):
        pw = d['pw'].copy(); n = d['n_peaks']
        pw[2 * n * n + 3 * n + 7] += 1e-3                               # r[3][7] != r[7][3]
        d['pw'] = pw
        return d
    with pytest.raises(MbxError):
        Suite([Tampered(p, asym)])

    def odd(d):                                                         # drop the last atom: n = 99
        n = d['n_peaks']; m = n - 1
        t = d['pw'].reshape(3, n, n)[:, :m, :m]
        d['pw'] = np.ascontiguousarray(t).ravel(); d['n_peaks'] = m
        d['py'] = np.ascontiguousarray(d['py'].reshape(-1, n, 3)[:, :m].reshape(d['py'].shape[0], 3 * m))
        d['pc'] = np.ascontiguousarray(d['pc'].reshape(n, 3)[:m]).ravel()
        return d
    q = Tampered(p, odd)
    s = Suite([q])
    X = np.random.RandomState(4).uniform(-1.5, 1.5, size=(9, 12))
    g = oracle.evaluate(q.desc(), X)
    assert np.all(np.abs(s.eval(0, X) - g) <= 1e-9 * np.abs(g))


@pytest.mark.gpu
def test_hip_rlepso_and_lde_on_protein_replay_reference():
    import torch
    from metabox_amd.suite import Batch, Suite
    from metabox_amd._abi import ALGO_LDE, ALGO_RLEPSO
    byid = protein()[0]
    cases = [str(c) for c in TR['cases']]
    pids = sorted({c.split('/')[1] for c in cases})
    s = Suite([byid[p] for p in pids])
    for c in cases:
        algo, pid, seed = c.split('/')
        k = pids.index(pid)
        acts = TR[f'{c}/actions']
        if algo == 'rlepso':
            b = Batch(s, ALGO_RLEPSO, [k], [0], 100, 1000, 200, 5)
            fd = oracle.NumpyTapeFeeder(int(seed), 100, 12, 0)
            sc_off = 3 * 100 * 12 + 3 * 100 + 12
        else:
            b = Batch(s, ALGO_LDE, [k], [0], 50, 1000, 200, 5)
            fd = oracle.LdeTapeFeeder(int(seed), 50, 12, 0, 1000)
            sc_off = 50 * 12 + 50 + 8
        b.set_tape(torch.from_numpy(fd.reset_tape()[None]).cuda())
        b.reset()
        for g, a in enumerate(acts):
            tape = fd.step_tape() if algo == 'rlepso' else fd.step_tape(TR[f'{c}/r'][g])
            b.set_tape(torch.from_numpy(tape[None]).cuda())
            st, r, d = b.step(torch.from_numpy(a[None].astype(np.float32)).cuda())
            sc = b.read_state(0)[sc_off:sc_off + 16]
            if algo == 'rlepso':
                fd.commit(sc[oracle.SC_REINIT] > 0)
            assert close(sc[0], TR[f'{c}/gbest'][g]), (c, g)
            assert sc[1] == TR[f'{c}/fes'][g] and bool(d[0].item()) == bool(TR[f'{c}/done'][g]), (c, g)
        res = b.results()
        n = int(res['cost_len'][0].item())
        assert n == len(TR[f'{c}/cost']) and close(res['cost'][0, :n].cpu().numpy(), TR[f'{c}/cost']), c
        assert res['cost'].shape[1] == 6
        b.close()


@pytest.mark.gpu
def test_protein_rlepso_resident_rollout_equals_per_generation_and_matches_the_oracle():
    """RLEPSO on protein docking (src/config.py:86-90: dim 12, maxFEs 1000, 5 log points; the setting of the published T2 row) takes the resident route:
    k_rlepso_run<256, 100, 12, 5> keeps the swarm on chip across the nine generations of an episode.  Whole episodes in uneven chunks, bit for bit
    against mbx_rlepso_act_step per generation (the run-time-geometry kernel), and against the C oracle on the actions the kernel drew."""
    import torch
    from metabox_amd.suite import Batch, Suite
    from metabox_amd._abi import ALGO_RLEPSO
    byid, tr, te, _ = protein()
    pids = [tr[0], tr[7], te[0], '1AVX_1', '7CEI_3', te[-1]]
    ps = [byid[p] for p in pids]
    s = Suite(ps)
    B = 12
    pidx, seeds = np.arange(B) % len(ps), np.arange(B, dtype=np.uint64) * 7919 + 3
    a = Batch(s, ALGO_RLEPSO, pidx, seeds, 100, 1000, 200, 5)
    b = Batch(s, ALGO_RLEPSO, pidx, seeds, 100, 1000, 200, 5)
    assert a.rollout_is_resident() and a.launch_info()['fixed_geometry'] == 8, a.launch_info()
    table = torch.rand(1000 + 200 + 1, 2, 35, generator=torch.Generator().manual_seed(9)).cuda()
    table[:, 1] = 0.05 + 0.3 * table[:, 1]
    table = table.contiguous()
    a.reset(); b.reset()
    acts_all = []
    for n in (4, 3, 4):                                   # the episode ends inside the third launch (9 generations), which then idles
        st, rw, dn, traj = a.rlepso_rollout(table, n, trajectory=True)
        st, rw, dn = st.clone(), rw.clone(), dn.clone()
        acts_all.append(traj['actions'].cpu().numpy().copy())
        rsum = torch.zeros(B, dtype=torch.float64, device='cuda')
        for g in range(n):
            live = (b.done == 0).clone()
            sb, rb, db, acts = b.act_step(table, want_actions=True)
            assert torch.equal(traj['state'][g], sb[:, 0]) and torch.equal(traj['reward'][g], rb) and torch.equal(traj['done'][g], db), g
            assert torch.equal(traj['actions'][g][live], acts[live]), g
            rsum += rb
        assert torch.equal(st[:, 0], sb[:, 0]) and torch.equal(dn, db) and torch.equal(rw, rsum)
        torch.cuda.synchronize()
        for k in range(B):
            assert np.array_equal(a.read_state(k), b.read_state(k), equal_nan=True), k
    ra, rb_ = a.results(), b.results()
    for key in ra:
        assert torch.equal(ra[key], rb_[key]), key
    assert torch.all(ra['steps'] == 9) and torch.all(ra['fes'] >= 1000)
    acts_all = np.concatenate(acts_all)[:9]
    cfg = oracle.make_cfg(1, 100, 12, 1000, 200, 5)
    for k in range(B):
        o = oracle.RlepsoOracle(ps[pidx[k]].desc(), None, cfg, seed=int(seeds[k]))
        o.reset()
        for g in range(9):
            _, _, d = o.step(acts_all[g, k])
        assert d
        want = oracle.split_rlepso_state(o.state(), 100, 12, 5)
        got = oracle.split_rlepso_state(a.read_state(k), 100, 12, 5)
        assert close(got['scalars'][oracle.SC_GBEST], want['scalars'][oracle.SC_GBEST]) and got['scalars'][oracle.SC_FES] == want['scalars'][oracle.SC_FES], k
        assert close(got['pbest'], want['pbest']) and close(got['cost'], want['cost']), k
    a.close(); b.close()


@pytest.mark.gpu
def test_close_pair_counts_of_host_and_library_agree():
    """The inter-rank partition weights a protein problem with its close-pair count computed on the host (Protein_Docking.close_pairs); the energy kernel walks the count
    mbx_suite_create computed (mbx_suite_close_pairs).  Same bound, two implementations: they must give the same number for all 280 problems."""
    from metabox_amd.config import get_config
    from metabox_amd.suite import Suite
    from metabox_amd.utils import construct_problem_set
    cfg = get_config(['--problem', 'protein', '--device', 'cuda'])
    tr, te = construct_problem_set(cfg)
    ps = (tr + te).data
    s = Suite(ps)
    lib_counts = [int(s.lib.mbx_suite_close_pairs(s._h, i)) for i in range(len(ps))]
    host_counts = [p.close_pairs() for p in ps]
    assert lib_counts == host_counts and min(lib_counts) == 1771 and max(lib_counts) == 3826
    s.close()
