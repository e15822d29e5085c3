"""GPU: the fused LDE generation kernel — tape replay against the reference's episodes, Philox parity with the
oracle, and size-independent properties at a large batch."""
import numpy as np
import pytest
import torch

from helpers import close, load, problems
from oracle import oracle

pytestmark = pytest.mark.gpu
NP = 50


def _suite(suite, dim):
    from metabox_amd.suite import Suite
    ps = problems(suite, dim)
    ids = sorted(ps)
    return Suite([ps[i] for i in ids]), ids


def _lde_groups():
    from test_oracle_lde import CASES, HD_CASES, lde_case
    groups = {}
    for c in CASES + HD_CASES:
        _, suite, dim, np_, _, _, _ = lde_case(c)
        groups.setdefault((suite, dim, np_), []).append(c)
    return groups


@pytest.mark.parametrize('suite,dim,NP', sorted(_lde_groups()))
def test_lde_tape_replay_matches_reference_episodes(suite, dim, NP):
    """Whole REFERENCE episodes (src/optimizer/lde_optimizer.py:159-198) replayed through mbx_set_tape + mbx_step.  D = 10 / NP = 50: the shipped setting;
    bbob-noisy D = 30 at NP = 50 and NP = 100: the geometries of BASELINE config 3 (k_lde_step's compile-time-geometry instantiations 3 and 6), one
    function per noise model + Gallaghers (tools/gen_golden.py lde_hd; NP = 100 = the reference with its one population literal patched in the generator)."""
    from metabox_amd.suite import Batch
    from metabox_amd._abi import ALGO_LDE
    from test_oracle_lde import check_lde_replay, lde_case
    mine = _lde_groups()[(suite, dim, NP)]
    s, ids = _suite(suite, dim)
    maxfes = 2000 * dim
    info = [lde_case(c) for c in mine]
    pidx = [ids.index(k[4]) for k in info]
    B = len(mine)
    batch = Batch(s, ALGO_LDE, pidx, np.arange(B), NP, maxfes, maxfes // 50, 50)
    assert batch.state_dim == NP + 10 and batch.action_dim == 2 * NP
    if dim == 30:
        assert batch.launch_info()['fixed_geometry'] == (3 if NP == 50 else 6), batch.launch_info()
    feeders = [oracle.LdeTapeFeeder(k[5], NP, dim, s.problems[j].noise[0], maxfes) for k, j in zip(info, pidx)]
    acts = [k[6] for k in info]
    rr = [k[0][f'{c}/r'] for k, c in zip(info, mine)]
    G = max(len(a) for a in acts)
    tape = np.stack([f.reset_tape() for f in feeders])
    batch.set_tape(torch.from_numpy(tape).cuda())
    st0 = batch.reset().cpu().numpy()
    rows = [[] for _ in range(B)]
    feats = [dict() for _ in range(B)]
    alive = np.ones(B, bool)
    sc_off = NP * dim + NP + 8
    for g in range(G):
        a = np.zeros((B, 2 * NP), np.float32)
        for b in range(B):
            if alive[b]:
                tape[b] = feeders[b].step_tape(rr[b][g])
                a[b] = acts[b][g]
        batch.set_tape(torch.from_numpy(tape).cuda())
        st, r, d = batch.step(torch.from_numpy(a).cuda())
        torch.cuda.synchronize()
        st = st.cpu().numpy(); r = r.cpu().numpy(); d = d.cpu().numpy()
        for b in range(B):
            if not alive[b]:
                continue
            sc = batch.read_state(b)[sc_off:sc_off + 16]
            rows[b].append((sc[0], sc[1], r[b], d[b]))
            feats[b][g] = st[b].copy()
            if d[b] or g == len(acts[b]) - 1:
                # (an episode whose reference run met exactly equal fitness values may legitimately end at another generation: first_tie_gen)
                alive[b] = False
                key = f'{mine[b]}/first_tie_gen'
                if key not in info[b][0].files or int(info[b][0][key]) < 0:
                    assert d[b] and g == len(acts[b]) - 1, (mine[b], g)
    assert not alive.any()
    res = batch.results()
    cost = res['cost'].cpu().numpy(); clen = res['cost_len'].cpu().numpy()
    for b, c in enumerate(mine):
        fin = oracle.split_lde_state(batch.read_state(b), NP, dim, 50)
        assert clen[b] == int(fin['scalars'][oracle.SC_COST_LEN]) and np.array_equal(cost[b, :clen[b]], fin['cost'][:clen[b]]), c
        check_lde_replay(c, info[b][0], NP, dim, st0[b], np.array(rows[b]), feats[b], fin)
    batch.close()


# NP = 50 is the reference's population (lde_optimizer.py:10-14, lde_agent.py:37); NP = 100 is BASELINE.json config 3 as written
# ("LDE ... dim=30 pop=100"), reachable through config.NP_override: same kernel family, run-time geometry, 512 threads, 90.8 KB of LDS.
@pytest.mark.parametrize('suite,dim,NP', [('bbob', 10, 50), ('bbob-noisy', 30, 50), ('bbob-noisy', 30, 100)])
def test_lde_philox_parity_with_oracle(suite, dim, NP):
    from metabox_amd.suite import Batch
    from metabox_amd._abi import ALGO_LDE
    s, ids = _suite(suite, dim)
    B, G = len(ids), 30
    maxfes = 2000 * dim
    rs = np.random.RandomState(5)
    actions = rs.uniform(0, 1, size=(G, B, 2 * NP)).astype(np.float32)
    seeds = np.arange(B, dtype=np.uint64) * 104729 + 3
    batch = Batch(s, ALGO_LDE, np.arange(B), seeds, NP, maxfes, maxfes // 50, 50)
    st0 = batch.reset().cpu().numpy().copy()
    hist = []
    for g in range(G):
        st, r, d = batch.step(torch.from_numpy(actions[g]).cuda())
        hist.append((st.cpu().numpy().copy(), r.cpu().numpy().copy()))
    cfg = oracle.make_cfg(2, NP, dim, maxfes, maxfes // 50, 50)
    for b in range(B):
        p = s.problems[b]
        o = oracle.LdeOracle(p.desc(), p.bias, cfg, seed=int(seeds[b]))
        f0 = o.reset()
        assert np.abs(f0 - st0[b]).max() <= 1e-7, ids[b]
        for g in range(G):
            f, rew, d = o.step(actions[g, b])
            assert np.abs(f - hist[g][0][b]).max() <= 1e-5, (ids[b], g)
            assert abs(rew - hist[g][1][b]) <= 1e-5 * abs(rew) + 1e-9, (ids[b], g)
        want = oracle.split_lde_state(o.state(), NP, dim, 50)
        got = oracle.split_lde_state(batch.read_state(b), NP, dim, 50)
        assert close(got['fit'], want['fit']), ids[b]
        assert np.abs(got['pop'] - want['pop']).max() <= 1e-9, ids[b]
        assert np.array_equal(got['hsum'][:5], want['hsum'][:5])       # (slot 5: the kernel's packed copy of the last histogram)
    batch.close()


@pytest.mark.parametrize('NP', [50, 100])
def test_lde_large_batch_properties(NP):
    """16384 instances (BASELINE.json config 3 size) on bbob-noisy d=30, at the reference's NP = 50 and at the configuration's pop = 100:
    deterministic, shard-independent, sorted; the launch geometry is the documented one."""
    from metabox_amd.suite import Batch
    from metabox_amd._abi import ALGO_LDE
    s, ids = _suite('bbob-noisy', 30)
    B = 16384
    pidx = np.arange(B) % len(ids)
    seeds = np.arange(B, dtype=np.uint64) + 99
    act = torch.rand(B, 2 * NP, generator=torch.Generator().manual_seed(1)).cuda()

    def run(sel):
        b = Batch(s, ALGO_LDE, pidx[sel], seeds[sel], NP, 60000, 1200, 50)
        info = b.launch_info()
        assert info['threads'] == (256 if NP == 50 else 512) and info['fixed_geometry'] == (3 if NP == 50 else 6), info
        # the maps stay in global memory (scalar-operand matvec): 38 848 B at NP = 50 (53 248 with them), 76 416 B at NP = 100 (90 816): two workgroups per CU
        assert info['lds_bytes'] == {50: 38848, 100: 76416}[NP] and 2 * info['lds_bytes'] <= 160 * 1024, info
        assert b.state_dim == NP + 10 and b.action_dim == 2 * NP
        b.reset()
        for _ in range(6):
            st, _, _ = b.step(act[sel].contiguous())
        out = {k: v.cpu().numpy() for k, v in b.results().items()}
        out['state'] = st.cpu().numpy().copy()
        fit0 = oracle.split_lde_state(b.read_state(0), NP, 30, 50)['fit']
        b.close()
        return out, fit0
    full, fit0 = run(np.arange(B))
    again, _ = run(np.arange(B))
    for k in full:
        assert np.array_equal(full[k], again[k], equal_nan=True), k
    part, _ = run(np.arange(B)[::4])
    for k in full:
        assert np.array_equal(full[k][::4], part[k], equal_nan=True), k
    assert np.all(np.diff(fit0) >= 0)                                  # population kept sorted by fitness
    st = full['state']
    assert np.all(st[:, 0] == 0) and np.all((st[:, NP - 1] == 1) | (st[:, NP - 1] == 0))   # min-max normalised
    assert np.all(st[:, NP:NP + 5].sum(1) == NP)                       # histogram counts every individual
    assert np.all(full['fes'] == NP * 7)


@pytest.mark.parametrize('NP', [50, 100])
def test_lde_resident_rollout_large_batch_properties(NP):
    """The route config 3 is TIMED on, at its full size: 16384 instances on bbob-noisy d=30 through mbx_lde_rollout (k_lde_run<NP, 30>: PolicyNet inside the kernel, 12 generations
    in two launches).  Deterministic, independent of the shard an instance runs in (every fourth instance alone gives the same rows, (h, c) and features), population kept
    sorted, features normalised, histogram complete, FEs billed per generation.  (the property test stepped k_lde_step only.)"""
    from metabox_amd.suite import Batch
    from metabox_amd._abi import ALGO_LDE
    s, ids = _suite('bbob-noisy', 30)
    B = 16384
    pidx = np.arange(B) % len(ids)
    seeds = np.arange(B, dtype=np.uint64) * 31 + 5
    net = _lde_net(NP)
    w, H = net.packed_weights(), net.lstm.hidden_size

    def run(sel):
        b = Batch(s, ALGO_LDE, pidx[sel], seeds[sel], NP, 60000, 1200, 50)
        assert b.lde_rollout_is_resident() and b.launch_info()['fixed_geometry'] == (3 if NP == 50 else 6)
        b.reset()
        h, c = torch.zeros(len(sel), H, device='cuda'), torch.zeros(len(sel), H, device='cuda')
        b.lde_rollout(w, H, h, c, 5)
        st, rw, dn = b.lde_rollout(w, H, h, c, 7)
        out = {k: v.cpu().numpy() for k, v in b.results().items()}
        out.update(state=st.cpu().numpy().copy(), h=h.cpu().numpy(), c=c.cpu().numpy(), reward=rw.cpu().numpy().copy())
        fits = [oracle.split_lde_state(b.read_state(k), NP, 30, 50)['fit'] for k in (0, len(sel) // 2, len(sel) - 1)]
        b.close()
        return out, fits
    full, fits = run(np.arange(B))
    again, _ = run(np.arange(B))
    for k in full:
        assert np.array_equal(full[k], again[k], equal_nan=True), k
    part, _ = run(np.arange(B)[::4])
    for k in full:
        assert np.array_equal(full[k][::4], part[k], equal_nan=True), k
    assert all(np.all(np.diff(f) >= 0) for f in fits)                  # population kept sorted by fitness
    st = full['state']
    assert np.all(st[:, 0] == 0) and np.all((st[:, NP - 1] == 1) | (st[:, NP - 1] == 0))   # min-max normalised
    assert np.all(st[:, NP:NP + 5].sum(1) == NP)                       # histogram counts every individual
    assert np.all(full['fes'] == NP * 13) and np.all(full['steps'] == 12)
    assert np.isfinite(full['h']).all() and np.abs(full['h']).max() <= 1 and np.ptp(full['h']) > 0


def test_lde_end_to_end_statistics_match_the_reference():
    """Philox-driven batched LDE rollouts (batched LSTM policy, shipped weights) vs 30 reference runs on three bbob
    problems: final-cost, FEs and return distributions must be statistically indistinguishable."""
    from scipy import stats as sps
    from metabox_amd.agent import LDE_Agent
    from metabox_amd.config import get_config
    from metabox_amd.environment import BatchedPBO_Env
    from metabox_amd.optimizer import LDE_Optimizer
    ref = load('lde_stats.npz')
    cfg = get_config(['--problem', 'bbob', '--dim', '10', '--device', 'cuda'])
    cfg.agent_save_dir = None
    agent = LDE_Agent(cfg).load_exported_weights(load('lde_policy.npz')).to('cuda')
    opt = LDE_Optimizer(cfg)
    fids = [1, 15, 20]
    ps = [problems('bbob', 10)[f] for f in fids]
    runs = 192
    pidx = np.repeat(np.arange(len(fids)), runs)
    torch.manual_seed(7)
    env = BatchedPBO_Env(ps, opt, pidx, np.arange(len(pidx), dtype=np.uint64) * 40503 + 1)
    out = agent.rollout_batch(env)
    cost = out['cost'][:, -1].cpu().numpy().reshape(len(fids), runs)
    fes = out['fes'].cpu().numpy().reshape(len(fids), runs)
    ret = out['return'].cpu().numpy().reshape(len(fids), runs)
    env.close()
    for k, f in enumerate(fids):
        rc, rf, rr = ref[f'{f}/final_cost'], ref[f'{f}/fes'], ref[f'{f}/return']
        lg, lr = np.log10(cost[k] + 1e-12), np.log10(rc + 1e-12)
        se = np.sqrt(lg.var() / runs + lr.var() / len(lr))
        assert abs(lg.mean() - lr.mean()) <= 4 * se + 1e-3, (f, lg.mean(), lr.mean(), se)
        assert sps.mannwhitneyu(cost[k], rc).pvalue > 1e-3, f
        assert sps.mannwhitneyu(ret[k], rr).pvalue > 1e-3, (f, ret[k].mean(), rr.mean())
        if f == 1:
            assert sps.mannwhitneyu(fes[k], rf).pvalue > 1e-3, (f, fes[k].mean(), rf.mean())
        else:
            assert np.all(fes[k] == 20000)


@pytest.mark.parametrize('NP', [50, 100])
def test_lde_compile_time_geometry_kernel_equals_generic_kernel(monkeypatch, NP):
    """BASELINE config 3 (D = 30; NP = 50 as in the reference, NP = 100 as written) runs k_lde_step with the geometry fixed at compile time --
    at NP = 100 also with the scalar-operand matvec (maps read from global memory instead of LDS); MBX_GENERIC_GEOMETRY=1 keeps the
    run-time-geometry kernel.  Every state word must be identical after 40 generations on the 30 noisy functions."""
    from metabox_amd.suite import Batch
    from metabox_amd._abi import ALGO_LDE
    s, ids = _suite('bbob-noisy', 30)
    B, G, maxfes = len(ids), 40, 60000
    actions = torch.rand(G, B, 2 * NP, generator=torch.Generator().manual_seed(8)).cuda()
    seeds = np.arange(B, dtype=np.uint64) * 131 + 9
    states = []
    for generic in ('0', '1'):
        monkeypatch.setenv('MBX_GENERIC_GEOMETRY', generic)
        batch = Batch(s, ALGO_LDE, np.arange(B), seeds, NP, maxfes, maxfes // 50, 50)
        batch.reset()
        for g in range(G):
            batch.step(actions[g])
        states.append(np.stack([batch.read_state(b) for b in range(B)]))
        batch.close()
    assert np.array_equal(states[0], states[1], equal_nan=True)


def test_lde_config3_population_through_the_plugin_surface():
    """BASELINE.json config 3 as written (pop = 100) through LDE_Agent / LDE_Optimizer / BatchedPBO_Env: config.NP_override = 100 sizes the
    PolicyNet (state NP + 10, action 2 NP) and the batch; a seeded fresh policy (no shipped weights fit 2 NP = 200 outputs) rolls out."""
    from metabox_amd.agent import LDE_Agent
    from metabox_amd.config import get_config
    from metabox_amd.environment import BatchedPBO_Env
    from metabox_amd.optimizer import LDE_Optimizer
    cfg = get_config(['--problem', 'bbob-noisy', '--dim', '30', '--device', 'cuda'])
    cfg.agent_save_dir = None
    cfg.NP_override = 100
    torch.manual_seed(3)
    agent = LDE_Agent(cfg).to('cuda')
    opt = LDE_Optimizer(cfg)
    assert cfg.NP == 100 and cfg.node_dim == 110 and cfg.output_dim_actor == 200
    ps = problems('bbob-noisy', 30)
    ids = sorted(ps)
    env = BatchedPBO_Env([ps[i] for i in ids], opt, np.arange(60) % len(ids), np.arange(60, dtype=np.uint64) + 5)
    assert env.state_dim == 110 and env.action_dim == 200
    out = agent.rollout_batch(env, max_steps=12)
    fes = out['fes'].cpu().numpy()
    assert np.all(fes == 100 * 13) and np.all(np.isfinite(out['cost'].cpu().numpy()))
    cost = out['cost'].cpu().numpy()
    assert np.all(cost[:, -1] <= cost[:, 0])
    env.close()


# ------------------------------------------------------------------------------------------------ resident rollout (k_lde_run, mbx_lde_rollout)
def _lde_net(NP, seed=3):
    """The PolicyNet config 3 runs at this population: the shipped bbob-easy weights at the reference's NP = 50, a seeded fresh net at NP = 100."""
    import os
    from metabox_amd.agent import LDE_Agent
    from metabox_amd.config import get_config
    cfg = get_config(['--problem', 'bbob-noisy', '--dim', '30', '--device', 'cuda'])
    cfg.agent_save_dir = None
    if NP != 50:
        cfg.NP_override = NP
    torch.manual_seed(seed)
    agent = LDE_Agent(cfg)
    if NP == 50:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        agent.load_exported_weights(np.load(os.path.join(root, 'metabox_amd', 'agent_model', 'lde_bbob_easy.npz')))
    return agent.to('cuda').net


def _lde_rollout_case(suite_name, fids, NP, B, chunks, resident=True, maxfes=None, early_stop=True, dim=30):
    """One mbx_lde_rollout launch per chunk against mbx_lde_policy + mbx_step per generation on a twin batch: per-generation actions, features,
    rewards, done flags, the LSTM's (h, c), whole state blocks and result tables must agree bit for bit."""
    from metabox_amd.suite import Batch, Suite
    from metabox_amd._abi import ALGO_LDE
    maxfes = maxfes or 2000 * dim
    ps = [problems(suite_name, dim)[f] for f in fids]
    s = Suite(ps)
    net = _lde_net(NP)
    w, H = net.packed_weights(), net.lstm.hidden_size
    pidx, seeds = np.arange(B) % len(ps), np.arange(B, dtype=np.uint64) * 7919 + 5
    a = Batch(s, ALGO_LDE, pidx, seeds, NP, maxfes, maxfes // 50, 50, early_stop=early_stop)
    b = Batch(s, ALGO_LDE, pidx, seeds, NP, maxfes, maxfes // 50, 50, early_stop=early_stop)
    assert a.lde_rollout_is_resident() == resident
    a.reset(); b.reset()
    ha, ca = torch.zeros(B, H, device='cuda'), torch.zeros(B, H, device='cuda')
    hb, cb = torch.zeros(B, H, device='cuda'), torch.zeros(B, H, device='cuda')
    for n in chunks:
        st, rw, dn, traj = a.lde_rollout(w, H, ha, ca, n, trajectory=True)
        st, rw, dn = st.clone(), rw.clone(), dn.clone()
        rsum = torch.zeros(B, dtype=torch.float64, device='cuda')
        for g in range(n):
            live = (b.done == 0).clone()
            hprev, cprev = hb.clone(), cb.clone()
            acts = b.lde_policy(w, H, hb, cb).clone()
            hb[~live] = hprev[~live]; cb[~live] = cprev[~live]       # (the per-generation policy kernel also advances finished instances; the rollout leaves them)
            sb, rb, db = b.step(acts)
            assert torch.equal(traj['actions'][g][live], acts[live]), (g, 'actions')
            assert torch.equal(traj['reward'][g], rb) and torch.equal(traj['done'][g], db), (g, 'reward / done')
            assert torch.equal(traj['state'][g][live], sb[live]), (g, 'features')
            rsum += rb
        assert torch.equal(st, sb) and torch.equal(dn, db) and torch.equal(rw, rsum)
        assert torch.equal(ha, hb) and torch.equal(ca, cb)
        torch.cuda.synchronize()
        for k in range(0, B, max(1, B // 16)):
            sa_, sb_ = a.read_state(k), b.read_state(k)
            assert np.array_equal(sa_, sb_, equal_nan=True), (k, np.flatnonzero(sa_ != sb_)[:8])
    ra, rb_ = a.results(), b.results()
    for key in ra:
        assert torch.equal(ra[key], rb_[key]), key
    out = {k: v.cpu().numpy() for k, v in ra.items()}
    a.close(); b.close()
    return out


@pytest.mark.parametrize('NP', [50, 100])
def test_lde_resident_rollout_equals_policy_plus_step(NP):
    """k_lde_run (population, order, features and the LSTM state on chip across the generations of a launch; PolicyNet inside the workgroup)
    == k_lstm_policy + k_lde_step per generation, bit for bit: all 30 noisy functions (config 3's suite) in uneven chunks, then the bbob
    functions whose kinds the resident kernel builds, then whole short episodes with the stop rule (launches that start with finished
    instances, terminations inside a launch)."""
    noisy = tuple(sorted(problems('bbob-noisy', 30)))
    _lde_rollout_case('bbob-noisy', noisy, NP, 60, (1, 6, 13, 2))
    _lde_rollout_case('bbob', (1, 2, 6, 7, 8, 9, 10, 11, 12, 13, 14, 16, 17, 18, 19, 21, 22, 23), NP, 36, (3, 8))
    r = _lde_rollout_case('bbob-noisy', (101, 107, 116, 122, 128), NP, 20, (4, 9, 30), maxfes=NP * 12)
    assert np.all(r['fes'] >= NP * 12) and np.all(r['steps'] == 11)          # every instance terminated inside the second launch


def test_lde_resident_rollout_reference_setting_dim10_all_24_kinds():
    """The reference's own LDE setting -- NP = 50 (lde_optimizer.py:10), bbob / bbob-noisy --dim 10, the shipped LDE_Agent.pkl -- takes the resident route
    (VERDICT r04 item 6): k_lde_run<50, 10> (one column tile; a second tile array carries the kinds whose row sums read two arrays or the candidate itself:
    F3, F4, F5, F15, F20, F24, src/problem/bbob.py:229-287, 585-602, 740-759, 869-890).  All 24 bbob functions and all 30 noisy ones, bit for bit against
    k_lstm_policy + k_lde_step per generation, in uneven chunks; then whole short episodes with terminations inside a launch."""
    _lde_rollout_case('bbob', tuple(range(1, 25)), 50, 48, (1, 6, 13, 2), dim=10)
    _lde_rollout_case('bbob-noisy', tuple(range(101, 131)), 50, 60, (5, 11), dim=10)
    r = _lde_rollout_case('bbob', (1, 3, 4, 5, 15, 20, 24, 21), 50, 16, (4, 9, 30), maxfes=50 * 12, dim=10)
    assert np.all(r['fes'] >= 50 * 12) and np.all(r['steps'] == 11)


def test_lde_resident_rollout_plain_bbob_dim30_all_24_kinds():
    """The reference's NP = 50 on plain bbob --dim 30 (round 6): a batch that holds F3 / F4 / F5 / F15 / F20 / F24 takes k_lde_run<50, 30, 50, true>, the instantiation with the
    second tile array (38.9 KB of LDS); batches without them keep the lean one config 3 is timed on.  All 24 functions, bit for bit against k_lstm_policy + k_lde_step per
    generation, in uneven chunks; then whole short episodes with terminations inside a launch.  src/problem/bbob.py:229-287, 585-602, 740-759, 869-890."""
    _lde_rollout_case('bbob', tuple(range(1, 25)), 50, 48, (1, 6, 13, 2))
    r = _lde_rollout_case('bbob', (1, 3, 4, 5, 15, 20, 24, 21), 50, 16, (4, 9, 30), maxfes=50 * 12)
    assert np.all(r['fes'] >= 50 * 12) and np.all(r['steps'] == 11)


def test_lde_rollout_host_loop_route(monkeypatch):
    """Behind the same entry point: objective kinds the resident kernel of a geometry does not build (NP = 100 at D = 30 has the lean instantiation only: F3, F15, F24 need the
    second tile array), and MBX_ROLLOUT_PER_GENERATION=1 take mbx_lde_policy + mbx_step per generation -- same records."""
    _lde_rollout_case('bbob', (1, 3, 15, 24), 100, 8, (2, 5), resident=False)
    # whole short episodes on this route: instances terminate inside the second call and the third call starts with every instance done -- their
    # (h, c) and action rows must stay untouched, as include/mbx.h promises for both routes
    r = _lde_rollout_case('bbob', (1, 3, 15, 24), 100, 8, (4, 9, 30), resident=False, maxfes=100 * 12)
    assert np.all(r['steps'] == 11)
    monkeypatch.setenv('MBX_ROLLOUT_PER_GENERATION', '1')
    _lde_rollout_case('bbob-noisy', (101, 128), 100, 8, (3, 4), resident=False)


@pytest.mark.parametrize('NP,suite,dim', [(50, 'bbob-noisy', 30), (100, 'bbob-noisy', 30), (50, 'bbob', 10), (50, 'bbob-noisy', 10), (50, 'bbob', 30)])
def test_lde_resident_rollout_matches_the_oracle(NP, suite, dim):
    """The resident kernel against the C oracle directly: 10 generations of all 30 noisy functions in ONE launch; the oracle, on the same Philox
    seeds, replays the actions the in-kernel PolicyNet drew and must see the same features / rewards after every generation and the same
    population at the end."""
    from metabox_amd.suite import Batch
    from metabox_amd._abi import ALGO_LDE
    s, ids = _suite(suite, dim)
    B, G, maxfes = len(ids), 10, 2000 * dim
    net = _lde_net(NP)
    w, H = net.packed_weights(), net.lstm.hidden_size
    seeds = np.arange(B, dtype=np.uint64) * 104729 + 3
    batch = Batch(s, ALGO_LDE, np.arange(B), seeds, NP, maxfes, maxfes // 50, 50)
    assert batch.lde_rollout_is_resident()
    st0 = batch.reset().cpu().numpy().copy()
    h, c = torch.zeros(B, H, device='cuda'), torch.zeros(B, H, device='cuda')
    _, _, _, traj = batch.lde_rollout(w, H, h, c, G, trajectory=True)
    torch.cuda.synchronize()
    acts, feats, rw = (traj[k].cpu().numpy() for k in ('actions', 'state', 'reward'))
    cfg = oracle.make_cfg(2, NP, dim, maxfes, maxfes // 50, 50)
    for b in range(B):
        p = s.problems[b]
        o = oracle.LdeOracle(p.desc(), p.bias, cfg, seed=int(seeds[b]))
        f0 = o.reset()
        assert np.abs(f0 - st0[b]).max() <= 1e-7, ids[b]
        for g in range(G):
            f, rew, d = o.step(acts[g, b])
            assert np.abs(f - feats[g, b]).max() <= 1e-5, (ids[b], g)
            assert abs(rew - rw[g, b]) <= 1e-5 * abs(rew) + 1e-9, (ids[b], g)
        want = oracle.split_lde_state(o.state(), NP, dim, 50)
        got = oracle.split_lde_state(batch.read_state(b), NP, dim, 50)
        assert close(got['fit'], want['fit']), ids[b]
        assert np.abs(got['pop'] - want['pop']).max() <= 1e-9, ids[b]
        assert np.array_equal(got['hsum'][:5], want['hsum'][:5])
    batch.close()
