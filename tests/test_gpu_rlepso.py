"""GPU: the fused RLEPSO generation kernel (through the C-ABI).

1. tape replay  — recorded numpy draws (regenerated from the seed) + the reference's float32 actions drive the
   HIP kernel; its gbest trajectory / cost list must match the REFERENCE run within 1e-5 relative.
2. Philox parity — HIP kernel vs the C oracle on identical Philox seeds and identical actions.
3. size-independent properties at the BASELINE.json batch size (4096 instances).
"""
import numpy as np
import pytest
import torch

from helpers import ATOL, RTOL, close, load, print_ledger, problems, prove_tie, prove_tie_arrays
from oracle import oracle

pytestmark = pytest.mark.gpu
NP, D, NLOG, MAXFES, LOGI = 100, 10, 50, 20000, 400


@pytest.fixture(scope='module')
def env():
    from metabox_amd.suite import Suite
    out = {}
    for suite in ('bbob', 'bbob-noisy'):
        ps = problems(suite, 10)
        ids = sorted(ps)
        out[suite] = (Suite([ps[i] for i in ids]), ids)
    return out


def _tape_replay_group(s, ids, TR, TIES, mine, NP, D, maxfes, ledger):
    """Replay the reference episodes `mine` (all on suite `s`, one geometry) through mbx_set_tape + mbx_step; returns (episodes whose bookkeeping
    is identical to the reference's in every generation, worst gbest relative error, launch info)."""
    from metabox_amd.suite import Batch
    from metabox_amd._abi import ALGO_RLEPSO
    fid_of = lambda c: int(c.split('/')[-3])
    seed_of = lambda c: int(c.split('/')[-2])
    pidx = [ids.index(fid_of(c)) for c in mine]
    B = len(mine)
    batch = Batch(s, ALGO_RLEPSO, pidx, np.arange(B), NP, maxfes, maxfes // NLOG, NLOG)
    info = batch.launch_info()
    feeders = [oracle.NumpyTapeFeeder(seed_of(c), NP, D, s.problems[k].noise[0]) for c, k in zip(mine, pidx)]
    acts = [TR[f'{c}/actions'] for c in mine]
    G = max(len(a) for a in acts)
    tape = np.stack([f.reset_tape() for f in feeders])
    batch.set_tape(torch.from_numpy(tape).cuda())
    batch.reset()
    torch.cuda.synchronize()
    prev = [oracle.split_rlepso_state(batch.read_state(b), NP, D, NLOG) for b in range(B)]
    for b, c in enumerate(mine):
        assert close(prev[b]['scalars'][oracle.SC_GBEST], TR[f'{c}/gbest0']), c
    gb = np.full((B, G), np.nan); fes = np.full((B, G), np.nan); rw = np.zeros((B, G)); dn = np.zeros((B, G), bool)
    alive = np.ones(B, bool)
    exact_until = [len(a) for a in acts]          # generations [0, exact_until[b]) have bookkeeping identical to the reference's
    for g in range(G):
        a = np.zeros((B, 35), np.float32)
        for b in range(B):
            if alive[b]:
                tape[b] = feeders[b].step_tape()
                a[b] = acts[b][g]
        batch.set_tape(torch.from_numpy(tape).cuda())
        st, r, d = batch.step(torch.from_numpy(a).cuda())
        torch.cuda.synchronize()
        r = r.cpu().numpy(); d = d.cpu().numpy()
        for b in range(B):
            if not alive[b]:
                continue
            cur = oracle.split_rlepso_state(batch.read_state(b), NP, D, NLOG)
            sc = cur['scalars']
            feeders[b].commit(sc[oracle.SC_REINIT] > 0)
            gb[b, g] = sc[oracle.SC_GBEST]; fes[b, g] = sc[oracle.SC_FES]; rw[b, g] = r[b]; dn[b, g] = d[b]
            if exact_until[b] == len(acts[b]) and not prove_tie(TIES, mine[b], g, prev[b]['ccost'], cur['ccost'], cur['pni'], ledger, 'hip'):
                exact_until[b] = g
            prev[b] = cur
            if d[b]:
                alive[b] = False
                assert g == len(acts[b]) - 1, (mine[b], 'episode length', g, len(acts[b]))
    assert not alive.any()
    res = batch.results()
    cost = res['cost'].cpu().numpy(); clen = res['cost_len'].cpu().numpy()
    n_exact, worst = 0, 0.0
    for b, c in enumerate(mine):
        n = len(acts[b])
        ref = TR[f'{c}/gbest']
        assert close(gb[b, :n], ref), (c, 'gbest trajectory', np.nanmax(np.abs(gb[b, :n] - ref) / np.maximum(np.abs(ref), 1e-300)))
        ref_cost = TR[f'{c}/cost']
        assert clen[b] == len(ref_cost), c
        assert close(cost[b, :clen[b]], ref_cost), c
        assert np.all(cost[b, clen[b]:] == cost[b, clen[b] - 1])          # 51-padding rule (tester.py:204-205)
        # integer-valued outputs: EXACT up to the first generation at which the bookkeeping leaves the reference's, and that
        # generation is a proven near-tie (helpers.prove_tie); no mismatch budget
        m = exact_until[b]
        assert np.array_equal(fes[b, :m], TR[f'{c}/fes'][:m]), c
        assert np.array_equal(rw[b, :m], TR[f'{c}/reward'][:m]), c
        assert np.array_equal(dn[b, :m], TR[f'{c}/done'][:m]), c
        if m == n:
            fin = oracle.split_rlepso_state(batch.read_state(b), NP, D, NLOG)
            assert np.array_equal(fin['pni'], TR[f'{c}/final_pni']), c
            assert np.abs(fin['pos'].reshape(NP, D) - TR[f'{c}/final_pos']).max() <= 1e-9, c
            assert close(fin['pbest'], TR[f'{c}/final_pbest'], rtol=1e-9), c
        n_exact += int(m == n)
        rel = np.abs(gb[b, :n] - ref) / np.maximum(np.abs(ref), 1e-12)
        worst = max(worst, float(rel.max()))
    batch.close()
    return n_exact, worst, info


def test_tape_replay_matches_reference_episodes(env):
    TR = load('rlepso_traces.npz')
    TIES = load('rlepso_ties.npz')
    cases = [str(c) for c in TR['cases']]
    ledger, worst, n_exact = [], 0.0, 0
    for suite in ('bbob', 'bbob-noisy'):
        s, ids = env[suite]
        mine = [c for c in cases if c.split('/')[0] == suite]
        ne, w, info = _tape_replay_group(s, ids, TR, TIES, mine, NP, D, MAXFES, ledger)
        assert info['fixed_geometry'] == 1, info
        n_exact += ne; worst = max(worst, w)
    print(f'tape replay: {n_exact}/{len(cases)} episodes with bookkeeping identical to the reference in every generation; worst gbest rel err '
          f'{worst:.2e}; {len(ledger)} episodes leave it at a proven near-tie:')
    print_ledger(ledger)


def test_tape_replay_with_the_fast_fdr_scan(env, monkeypatch):
    """MBX_F_FDR_FAST (include/mbx.h; here through its test override MBX_FDR_FAST=1 in the environment, read when the batch is created): the generation kernels
    WITHOUT the near-tie flag replay the reference episodes like the default (exact) kernels do -- recorded episodes hold no two candidates within an ulp."""
    from metabox_amd._abi import F_FDR_FAST
    from metabox_amd.suite import Batch
    monkeypatch.setenv('MBX_FDR_FAST', '1')
    TR = load('rlepso_traces.npz')
    TIES = load('rlepso_ties.npz')
    s, ids = env['bbob']
    probe = Batch(s, 1, np.zeros(1, int), np.zeros(1, np.uint64), NP, MAXFES, MAXFES // 50, 50)
    assert probe.flags == F_FDR_FAST                               # the override reached the batch's flags
    probe.close()
    mine = [c for c in (str(c) for c in TR['cases']) if c.split('/')[0] == 'bbob'][::3]
    ledger = []
    ne, w, info = _tape_replay_group(s, ids, TR, TIES, mine, NP, D, MAXFES, ledger)
    assert info['fixed_geometry'] == 1 and ne + len(ledger) == len(mine)
    print(f'fast-FDR kernels: {ne}/{len(mine)} episodes identical in every generation, worst gbest rel err {w:.2e}')


# (dim, NP) -> the compile-time-geometry instantiation mbx_step must take: 7 = k_rlepso_step<512, 100, 30, 5> (bbob --dim 30, the geometry config 3's
# suite runs RLEPSO at), 2 = k_rlepso_step<1024, 128, 40, 5> (BASELINE config 5), 10 = NP 100 at D 40 (the reference as shipped: resident rollout kernel of its own, mbx_step on the run-time-geometry kernel)
HD_GEOMETRIES = {(30, 100): 7, (40, 100): 10, (40, 128): 2}


@pytest.mark.parametrize('dim,np_', sorted(HD_GEOMETRIES))
def test_tape_replay_matches_reference_episodes_at_dim_30_and_40(dim, np_):
    """Whole REFERENCE episodes at the geometries of BASELINE configs 3 / 5 (tools/gen_golden.py rlepso_hd: --dim 30 / 40 as shipped, and NP = 128 with
    the reference's one population constant patched in the generator), replayed through mbx_set_tape + mbx_step: 12 / 11 / 13 episodes on 24 functions incl.
    F16 / F21 / F24, all three noise models, the Gallagher noisy kinds, 'uniform'-action episodes and F5 / F7 where __reinit fires in > 100 generations.
    src/optimizer/rlepso_optimizer.py:39-65, 173-263."""
    from metabox_amd.suite import Suite
    HD = load('rlepso_traces_hd.npz')
    cases = [str(c) for c in HD['cases'] if int(c.split('/')[1]) == dim and int(c.split('/')[2]) == np_]
    assert len(cases) >= 11
    ledger, worst, n_exact = [], 0.0, 0
    for suite in ('bbob', 'bbob-noisy'):
        ps = problems(suite, dim)
        ids = sorted(ps)
        mine = [c for c in cases if c.split('/')[0] == suite]
        ne, w, info = _tape_replay_group(Suite([ps[i] for i in ids]), ids, HD, HD, mine, np_, dim, 2000 * dim, ledger)
        assert info['fixed_geometry'] == HD_GEOMETRIES[(dim, np_)], info
        n_exact += ne; worst = max(worst, w)
    print(f'tape replay D = {dim}, NP = {np_}: {n_exact}/{len(cases)} episodes with bookkeeping identical to the reference in every generation; worst '
          f'gbest rel err {worst:.2e}; {len(ledger)} episodes leave it at a proven near-tie:')
    print_ledger(ledger)


def _hip_vs_oracle(batch, problems_, seeds, actions, dim, maxfes, logi, nlog, label):
    """Step the HIP batch and one C oracle per instance through the same Philox seeds and actions, generation by generation.  Floats
    must agree within the parity tolerance throughout.  Bookkeeping (per_no_improve, fes) must be EXACT up to the first generation at
    which an instance leaves the oracle's branch, and that generation must be a proven near-tie (helpers.prove_tie_arrays: the
    oracle's own margin |new_cost - c_cost| is below twice the deviation between the two implementations' operands).  Returns the
    ledger and the per-instance generation up to which everything was exact."""
    G, B = actions.shape[0], actions.shape[1]
    cfg = oracle.make_cfg(1, NP, dim, maxfes, logi, nlog)
    orc = [oracle.RlepsoOracle(p.desc(), p.bias, cfg, seed=int(sd)) for p, sd in zip(problems_, seeds)]
    for o in orc:
        o.reset()
    split = lambda st: oracle.split_rlepso_state(st, NP, dim, nlog)
    hp = [split(batch.read_state(b)) for b in range(B)]
    op = [split(o.state()) for o in orc]
    ledger, exact_until, odone = [], [G] * B, [False] * B
    for g in range(G):
        batch.step(torch.from_numpy(actions[g]).cuda())
        torch.cuda.synchronize()
        for b in range(B):
            if odone[b]:
                continue
            _, _, d = orc[b].step(actions[g, b])
            odone[b] = bool(d)
            hc, oc = split(batch.read_state(b)), split(orc[b].state())
            assert close(hc['scalars'][oracle.SC_GBEST], oc['scalars'][oracle.SC_GBEST]), (label, b, g, 'gbest')
            if exact_until[b] == G:
                same = prove_tie_arrays(op[b]['ccost'], oc['ccost'], oc['pni'], hp[b]['ccost'], hc['ccost'], hc['pni'], ledger,
                                        'hip vs oracle', f'{label}/instance {b}', g)
                if not same:
                    exact_until[b] = g
                else:
                    assert hc['scalars'][oracle.SC_FES] == oc['scalars'][oracle.SC_FES], (label, b, g, 'fes')
                    assert close(hc['pbest'], oc['pbest']) and np.abs(hc['pos'] - oc['pos']).max() <= 1e-9, (label, b, g)
            hp[b], op[b] = hc, oc
    return ledger, exact_until


@pytest.mark.parametrize('suite', ['bbob', 'bbob-noisy'])
def test_philox_parity_with_oracle(env, suite):
    from metabox_amd.suite import Batch
    from metabox_amd._abi import ALGO_RLEPSO
    s, ids = env[suite]
    B = len(ids)
    G = 40
    rs = np.random.RandomState(11)
    # mildly stagnating random policy so that __reinit fires in some instances
    actions = rs.uniform(0, 1, size=(G, B, 35)).astype(np.float32)
    seeds = np.arange(B, dtype=np.uint64) * 7919 + 17
    batch = Batch(s, ALGO_RLEPSO, np.arange(B), seeds, NP, MAXFES, LOGI, NLOG)
    batch.reset()
    ledger, exact_until = _hip_vs_oracle(batch, s.problems, seeds, actions, D, MAXFES, LOGI, NLOG, suite)
    print(f'{suite}: {sum(m == G for m in exact_until)}/{B} instances bit-compatible with the oracle through {G} generations; others:')
    print_ledger(ledger)
    batch.close()


def test_full_batch_properties(env):
    """BASELINE.json size (4096 instances): determinism, shard-independence, monotone gbest, done is absorbing."""
    from metabox_amd.suite import Batch
    from metabox_amd._abi import ALGO_RLEPSO
    s, ids = env['bbob']
    B = 4096
    pidx = np.arange(B) % len(ids)
    seeds = np.arange(B, dtype=np.uint64) // len(ids) + 1000
    act = torch.rand(B, 35, generator=torch.Generator().manual_seed(0)).cuda()

    def run(sel, steps=12):
        b = Batch(s, ALGO_RLEPSO, pidx[sel], seeds[sel], NP, MAXFES, LOGI, NLOG)
        b.reset()
        g = []
        for _ in range(steps):
            b.step(act[sel].contiguous())
            g.append(b.results()['cost'][:, 0].clone())
        r = {k: v.cpu().numpy() for k, v in b.results().items()}
        gbest = np.array([b.read_state(i)[3 * NP * D + 3 * NP + D] for i in (0, len(sel) - 1)])
        b.close()
        return r, gbest
    full, gfull = run(np.arange(B))
    again, _ = run(np.arange(B))
    for k in full:
        assert np.array_equal(full[k], again[k]), k                  # same seeds -> identical bits
    half = np.arange(B)[1::2]
    part, _ = run(half)
    for k in full:
        assert np.array_equal(full[k][half], part[k]), k              # results do not depend on batch position/size
    assert np.all(full['steps'] == 12) or np.all(full['steps'] <= 12)
    assert np.all(full['fes'] >= 100 + 100 * full['steps'])


def test_actor_table_equals_mlp_forward():
    """The per-fes (mu, sigma) table used by rollout_batch reproduces the actor's forward (PyTorch fp32 reference) and
    the reference's recorded (state -> mu, sigma) pairs."""
    from metabox_amd.agent.rlepso_agent import RLEPSO_Agent
    from metabox_amd.config import get_config
    cfg = get_config(['--problem', 'bbob', '--dim', '10', '--device', 'cuda'])
    cfg.agent_save_dir = None
    pol = load('rlepso_policy.npz')
    agent = RLEPSO_Agent(cfg).load_exported_weights(pol).to('cuda')
    with torch.no_grad():
        mu, sg = agent.actor.distribution(torch.from_numpy(pol['io/state']).cuda())
    assert np.allclose(mu.cpu().numpy(), pol['io/mu'], atol=2e-6) and np.allclose(sg.cpu().numpy(), pol['io/sigma'], atol=2e-6)
    tab = agent.actor_table(20000, 100, torch.device('cuda', 0))
    fes = torch.tensor([100., 200., 4321., 19999., 20000., 20043.], dtype=torch.float64, device='cuda')
    with torch.no_grad():
        want_mu, want_sg = agent.actor.distribution((fes / 20000).to(torch.float32)[:, None])
    got = tab.table.index_select(0, fes.long())
    assert torch.allclose(got[:, :35], want_mu, atol=2e-6) and torch.allclose(got[:, 35:], want_sg, atol=2e-6)
    # the fused 3-GEMM forward used per generation == the module-by-module forward
    st = torch.rand(257, 1, device='cuda')
    with torch.no_grad():
        m0, s0 = agent.actor.distribution(st)
    torch.manual_seed(5); eps = torch.randn_like(m0)
    torch.manual_seed(5); got_a = agent.actor.act_batch(st)
    assert torch.allclose(got_a, torch.clamp(m0 + s0 * eps, 0, 1), atol=3e-6)
    torch.manual_seed(0)
    a = tab.act((fes / 20000)[:, None])
    assert a.shape == (6, 35) and a.dtype == torch.float32 and float(a.min()) >= 0 and float(a.max()) <= 1
    # sampling statistics: mean of many draws approaches clamp-free mu where sigma is small
    big = tab.act((fes[:1] / 20000).repeat(20000)[:, None])
    tight = want_sg[0] < 0.05
    if tight.any():
        assert torch.allclose(big.mean(0)[tight], want_mu[0][tight].clamp(0, 1), atol=5e-3)


def test_policy_kernel_equals_actor_forward():
    """mbx_gauss_policy (one launch: both MLPs, squashing, Philox Normal draw, clamp) vs the PyTorch fp32 modules and the
    reference's recorded (state -> mu, sigma) pairs; the draws are N(0, 1), reproducible and keyed by (seed, generation)."""
    from metabox_amd.agent.rlepso_agent import RLEPSO_Agent
    from metabox_amd.config import get_config
    from metabox_amd.environment import BatchedPBO_Env
    from metabox_amd.optimizer import RLEPSO_Optimizer
    cfg = get_config(['--problem', 'bbob', '--dim', '10', '--device', 'cuda'])
    cfg.agent_save_dir = None
    pol = load('rlepso_policy.npz')
    agent = RLEPSO_Agent(cfg).load_exported_weights(pol).to('cuda')
    actor = agent.actor
    h1, h2 = actor.hidden_sizes()
    assert (h1, h2) == (64, 32) and actor.packed_weights().numel() == 2 * (64 + 64 + 64 * 32 + 32 + 32 * 35 + 35)
    ps = [problems('bbob', 10)[f] for f in (1, 8)]
    B = 3001                                                     # not a multiple of the 4 waves of a block
    env = BatchedPBO_Env(ps, RLEPSO_Optimizer(cfg), np.arange(B) % 2, np.arange(B, dtype=np.uint64) + 99)
    state = env.reset()
    io_state = torch.from_numpy(pol['io/state']).cuda().to(torch.float64)
    n_io = io_state.shape[0]
    state[:n_io] = io_state                                      # recorded states of the reference in the first rows
    state[n_io:, 0] = torch.rand(B - n_io, dtype=torch.float64, device='cuda')
    act, ms = env.batch.gauss_policy(actor.packed_weights(), h1, h2, actor.min_sigma, actor.max_sigma, want_mu_sigma=True)
    act, ms = act.clone(), ms.clone()
    with torch.no_grad():
        mu, sg = actor.distribution(state.to(torch.float32))
    assert torch.allclose(ms[:, 0], mu, atol=2e-6) and torch.allclose(ms[:, 1], sg, atol=2e-6)
    assert np.allclose(ms[:n_io, 0].cpu().numpy(), pol['io/mu'], atol=2e-6)
    assert np.allclose(ms[:n_io, 1].cpu().numpy(), pol['io/sigma'], atol=2e-6)
    assert float(act.min()) >= 0 and float(act.max()) <= 1
    inner = (act > 0) & (act < 1) & (ms[:, 1] > 0.05)
    z = ((act - ms[:, 0]) / ms[:, 1])[inner].double()            # un-clamped draws: a truncated standard normal sample
    free = (ms[:, 0] > 0.3) & (ms[:, 0] < 0.7) & (ms[:, 1] < 0.1) & inner      # >= 3 sigma from both clamps: untruncated
    zf = ((act - ms[:, 0]) / ms[:, 1])[free].double()
    assert zf.numel() > 2000 and abs(float(zf.mean())) < 5 / zf.numel() ** 0.5 and abs(float(zf.var()) - 1) < 0.1
    assert z.numel() > 10000 and float(z.abs().max()) < 7
    again = env.batch.gauss_policy(actor.packed_weights(), h1, h2, actor.min_sigma, actor.max_sigma).clone()
    assert torch.equal(again, act)                               # same (seed, generation) -> same draws
    env.step(act)
    env.batch.state.copy_(state)
    nxt = env.batch.gauss_policy(actor.packed_weights(), h1, h2, actor.min_sigma, actor.max_sigma)
    assert not torch.equal(nxt, act) and float((nxt - act).abs().max()) > 0.1      # next generation: new draws, same (mu, sigma)
    # weights are re-packed after an in-place update
    with torch.no_grad():
        next(actor.parameters()).mul_(1.5)
    _, ms2 = env.batch.gauss_policy(actor.packed_weights(), h1, h2, actor.min_sigma, actor.max_sigma, want_mu_sigma=True)
    with torch.no_grad():
        mu2, _ = actor.distribution(state.to(torch.float32))
    assert torch.allclose(ms2[:, 0], mu2, atol=2e-6) and not torch.allclose(mu2, mu, atol=1e-4)
    env.close()


@pytest.mark.parametrize('suite_name, fids', [('bbob', (1, 3, 16, 21)), ('bbob-noisy', (101, 115, 128))])
def test_fused_act_step_equals_policy_then_step(suite_name, fids):
    """mbx_rlepso_act_step (action drawn inside the generation kernel from the actor table) == mbx_gauss_policy + mbx_step,
    bit for bit, over a whole episode including early stops and re-initialisations (and, on the noisy triple, the in-kernel noise
    draws); the table rows equal the actor's forward."""
    from metabox_amd.agent.rlepso_agent import RLEPSO_Agent
    from metabox_amd.config import get_config
    from metabox_amd.environment import BatchedPBO_Env
    from metabox_amd.optimizer import RLEPSO_Optimizer
    cfg = get_config(['--problem', 'bbob', '--dim', '10', '--device', 'cuda'])
    cfg.agent_save_dir = None
    agent = RLEPSO_Agent(cfg).load_exported_weights(load('rlepso_policy.npz')).to('cuda')
    actor = agent.actor
    h1, h2 = actor.hidden_sizes()
    net = (actor.packed_weights(), h1, h2, actor.min_sigma, actor.max_sigma)
    ps = [problems(suite_name, 10)[f] for f in fids]
    B = 64
    pidx, seeds = np.arange(B) % len(ps), np.arange(B, dtype=np.uint64) * 7919 + 3
    env_a = BatchedPBO_Env(ps, RLEPSO_Optimizer(cfg), pidx, seeds)
    env_b = BatchedPBO_Env(ps, RLEPSO_Optimizer(cfg), pidx, seeds)
    table = env_a.batch.policy_table(*net)
    assert table.shape == (20000 + 200 + 1, 2, 35)
    k = torch.tensor([0, 100, 777, 19999, 20000, 20200], device='cuda')
    with torch.no_grad():
        mu, sg = actor.distribution((k.double() / 20000).to(torch.float32)[:, None])
    assert torch.allclose(table[k, 0], mu, atol=2e-6) and torch.allclose(table[k, 1], sg, atol=2e-6)
    env_a.reset(); env_b.reset()
    for g in range(199):
        live = env_a.batch.done == 0 if g else torch.ones(B, dtype=torch.bool, device='cuda')      # done instances draw no action
        sa, ra, da, acts = env_a.batch.act_step(table, want_actions=True)
        want = env_b.batch.gauss_policy(*net)
        assert torch.equal(acts[live], want[live]), g
        sb, rb, db = env_b.step(want)
        assert torch.equal(sa, sb) and torch.equal(ra, rb) and torch.equal(da, db), g
    ra_, rb_ = env_a.results(), env_b.results()
    for key in ra_:
        assert torch.equal(ra_[key], rb_[key]), key
    assert bool(ra_['fes'].max() >= 20000)
    if suite_name == 'bbob':
        assert bool((ra_['steps'] < 199).any())                                         # Sphere instances stopped early
    # rollout_batch('fused') is that loop
    for route in ('fused', 'resident'):                # 'resident' (the default) is the same episode in a single launch
        out = agent.rollout_batch(BatchedPBO_Env(ps, RLEPSO_Optimizer(cfg), pidx, seeds), policy=route)
        assert torch.equal(out['cost'], ra_['cost']) and torch.equal(out['return'], ra_['return']) and torch.equal(out['fes'], ra_['fes']), route
    env_a.close(); env_b.close()


def _rollout_case(suite_name, dim, fids, np_, B, chunks, maxfes=None, resident=True, early_stop=True):
    """One resident launch per chunk (mbx_rlepso_rollout) against one mbx_rlepso_act_step launch per generation on a twin batch:
    whole state blocks, trajectories and result tables must agree bit for bit."""
    from metabox_amd.agent.rlepso_agent import RLEPSO_Agent
    from metabox_amd.config import get_config
    from metabox_amd.suite import Batch, Suite
    from metabox_amd._abi import ALGO_RLEPSO
    cfg = get_config(['--problem', 'bbob', '--dim', str(dim), '--device', 'cuda'])
    cfg.agent_save_dir = None
    agent = RLEPSO_Agent(cfg).load_exported_weights(load('rlepso_policy.npz')).to('cuda')
    actor = agent.actor
    h1, h2 = actor.hidden_sizes()
    net = (actor.packed_weights(), h1, h2, actor.min_sigma, actor.max_sigma)
    ps = [problems(suite_name, dim)[f] for f in fids]
    s = Suite(ps)
    maxfes = maxfes or 2000 * dim
    pidx, seeds = np.arange(B) % len(ps), np.arange(B, dtype=np.uint64) * 104729 + 11
    a = Batch(s, ALGO_RLEPSO, pidx, seeds, np_, maxfes, maxfes // 50, 50, early_stop=early_stop)
    b = Batch(s, ALGO_RLEPSO, pidx, seeds, np_, maxfes, maxfes // 50, 50, early_stop=early_stop)
    table = a.policy_table(*net)
    assert a.rollout_is_resident() == resident                      # the route mbx_rlepso_rollout takes is not silent (mbx_rlepso_rollout_resident)
    with pytest.raises(ValueError):
        a.rlepso_rollout(table[:-1].contiguous(), 1)                # a table with too few rows would be read out of bounds on the device
    a.reset(); b.reset()
    for n in chunks:
        st, rw, dn, traj = a.rlepso_rollout(table, n, trajectory=True)
        st, rw, dn = st.clone(), rw.clone(), dn.clone()
        rsum = torch.zeros(B, dtype=torch.float64, device='cuda')
        for g in range(n):
            live = (b.done == 0).clone()
            sb, rb, db, acts = b.act_step(table, want_actions=True)
            assert torch.equal(traj['state'][g], sb[:, 0]) and torch.equal(traj['reward'][g], rb) and torch.equal(traj['done'][g], db), g
            assert torch.equal(traj['actions'][g][live], acts[live]), g
            rsum += rb
        assert torch.equal(st[:, 0], sb[:, 0]) and torch.equal(dn, db) and torch.equal(rw, rsum)
        torch.cuda.synchronize()
        for k in range(0, B, max(1, B // 16)):
            sa_, sb_ = a.read_state(k), b.read_state(k)
            assert np.array_equal(sa_, sb_, equal_nan=True), (k, np.flatnonzero(sa_ != sb_)[:8])
    ra, rb_ = a.results(), b.results()
    for key in ra:
        assert torch.equal(ra[key], rb_[key]), key
    out = {'steps': ra['steps'].cpu().numpy(), 'fes': ra['fes'].cpu().numpy(),
           'pbest': np.stack([oracle.split_rlepso_state(a.read_state(k), np_, dim, 50)['pbest'] for k in range(B)])}
    a.close(); b.close()
    return out


def test_resident_rollout_equals_one_launch_per_generation():
    """mbx_rlepso_rollout (state on chip across the generations of a launch) == mbx_rlepso_act_step per generation, bit for bit, over
    whole episodes in uneven chunks: early stops inside a launch, launches that start with finished instances, re-initialisations.
    Every function of both suites, so that every per-kind body of the resident kernel runs whole episodes against the one-generation kernel."""
    r = _rollout_case('bbob', 10, tuple(range(1, 25)), 100, 96, (1, 7, 60, 3, 140))        # all 24 kinds: k_rlepso_run has one body per kind
    assert (r['steps'] < 199).any() and (r['fes'] >= 20000).any() and (r['fes'] % 100 != 0).any()      # re-initialisations bill odd FEs
    r = _rollout_case('bbob-noisy', 10, tuple(range(101, 131)), 100, 60, (25, 180))          # all 30 noisy functions: the eight bodies that carry the noise models
    assert (r['fes'] >= 20000).any()


@pytest.mark.parametrize('dim,np_,chunks', [(40, 128, (1, 90, 233, 300)), (30, 100, (250, 350)), (40, 100, (3, 400, 400))])
def test_resident_rollout_equals_one_launch_per_generation_whole_episodes_config_5_and_3_geometries(dim, np_, chunks):
    """The same bit-for-bit identity over WHOLE episodes at the geometries of BASELINE config 5 (k_rlepso_run<1024, 128, 40, 5>: one out-of-line body per
    function kind) and of bbob --dim 30 (k_rlepso_run<512, 100, 30, 5>), every function of both suites.  Together with the reference episodes replayed
    through the one-generation kernels of these geometries (test_tape_replay_matches_reference_episodes_at_dim_30_and_40) this ties the resident
    kernels config 5 is timed on to the reference run, not only to the oracle."""
    r = _rollout_case('bbob', dim, tuple(range(1, 25)), np_, 24, chunks)
    assert (r['fes'] >= 2000 * dim).any()
    r = _rollout_case('bbob-noisy', dim, tuple(range(101, 131)), np_, 30, chunks)
    assert (r['fes'] >= 2000 * dim).any()


def test_resident_ranking_orders_equal_costs_like_the_one_generation_kernel():
    """k_rlepso_run ranks the particles with ONE compare per pair and falls back to the index tie-break only when two pbest costs are exactly
    equal (a contested slot of ORDER); k_rlepso_step always counts `<` and `<=`.  Linear slope without the stop rule collapses onto the optimum
    corner (cost exactly 0 for many particles) and the step ellipsoid has plateaus: long runs of both must stay bit-identical between the two
    kernels, and the final states must really hold equal costs (otherwise the fallback was never taken)."""
    r = _rollout_case('bbob', 10, (5, 7), 100, 32, (70, 50, 30), early_stop=False)
    ties = [(len(row) - len(np.unique(row))) for row in r['pbest']]
    assert max(ties) >= 5, ties


def _resident_vs_oracle(ps, np_, dim, maxfes, G, seed_mul=6151):
    """The resident kernel against the C oracle directly (not through k_rlepso_step): G generations in ONE launch of mbx_rlepso_rollout with
    the trajectory record on; the oracle, on the same Philox seeds, replays the actions the kernel drew and must see the same state / reward /
    done after every generation and the same population at the end (noisy functions included: the in-kernel noise draws are the oracle's Philox
    draws).  Returns the launch geometry the batch ran on, so that the caller can assert WHICH instantiation it has just compared."""
    from metabox_amd.suite import Batch, Suite
    from metabox_amd._abi import ALGO_RLEPSO
    s = Suite(ps)
    B, logi = len(ps), maxfes // 50
    seeds = np.arange(B, dtype=np.uint64) * seed_mul + 17
    batch = Batch(s, ALGO_RLEPSO, np.arange(B), seeds, np_, maxfes, logi, NLOG)
    assert batch.rollout_is_resident()
    table = torch.rand(maxfes + 2 * np_ + 1, 2, 35, generator=torch.Generator().manual_seed(4)).cuda()
    table[:, 1] = 0.05 + 0.3 * table[:, 1]                                      # sigma
    table = table.contiguous()
    batch.reset()
    _, _, _, traj = batch.rlepso_rollout(table, G, trajectory=True)
    torch.cuda.synchronize()
    acts, st, rw, dn = (traj[k].cpu().numpy() for k in ('actions', 'state', 'reward', 'done'))
    cfg = oracle.make_cfg(1, np_, dim, maxfes, logi, NLOG)
    for b in range(B):
        p = ps[b]
        o = oracle.RlepsoOracle(p.desc(), p.bias, cfg, seed=int(seeds[b]))
        o.reset()
        for g in range(G):
            so, ro, do = o.step(acts[g, b])
            assert so == st[g, b] and ro == rw[g, b] and bool(do) == bool(dn[g, b]), (p.func_id, g)
            assert not do, p.func_id
        want = oracle.split_rlepso_state(o.state(), np_, dim, NLOG)
        got = oracle.split_rlepso_state(batch.read_state(b), np_, dim, NLOG)
        assert close(got['scalars'][oracle.SC_GBEST], want['scalars'][oracle.SC_GBEST]), p.func_id
        assert got['scalars'][oracle.SC_FES] == want['scalars'][oracle.SC_FES], p.func_id
        assert close(got['pbest'], want['pbest']) and np.abs(got['pos'] - want['pos']).max() <= 1e-9, p.func_id
        assert close(got['ccost'], want['ccost']), p.func_id
        assert np.array_equal(got['pni'], want['pni']), p.func_id
    info = batch.launch_info()
    batch.close()
    return info


@pytest.mark.parametrize('suite_name', ['bbob', 'bbob-noisy'])
def test_resident_rollout_matches_the_oracle(env, suite_name):
    """k_rlepso_run<256, 100, 10, 5> (the headline kernel), all 24 + 30 functions, 12 generations in one launch.  (The long-horizon proof --
    exact up to proven near-ties -- is test_philox_parity_with_oracle on k_rlepso_step, to which
    test_resident_rollout_equals_one_launch_per_generation ties this kernel bit for bit; in the first dozen generations the swarm is wide and no
    comparison is anywhere near a tie.)"""
    s, ids = env[suite_name]
    _resident_vs_oracle(list(s.problems), NP, D, MAXFES, 12)


def _suite54(dim):
    return [problems('bbob', dim)[k] for k in sorted(problems('bbob', dim))] + \
           [problems('bbob-noisy', dim)[k] for k in sorted(problems('bbob-noisy', dim))]


def test_config5_resident_rollout_matches_the_oracle():
    """The kernel bench.py times for config 5, k_rlepso_run<1024, 128, 40, 5>, tied to the oracle DIRECTLY on all 54 functions (24 bbob + 30
    noisy: in-kernel noise draws, F22 / F128-F130 on the lean Gallagher route, the row-chunked scalar-operand matvec, the FDR scan unrolled by 2)
    -- the D = 40 twin of test_resident_rollout_matches_the_oracle.  Reference semantics: src/optimizer/rlepso_optimizer.py:173-263,
    src/problem/bbob.py:96-146."""
    info = _resident_vs_oracle(_suite54(40), 128, 40, 80000, 12)
    assert info['threads'] == 1024 and info['fixed_geometry'] == 2, info


def test_dim30_resident_rollout_matches_the_oracle():
    """k_rlepso_run<512, 100, 30, 5> (bbob --dim 30, the reference's other standard dimension; float64 MFMA matvec route) on 24 + 30 functions."""
    info = _resident_vs_oracle(_suite54(30), 100, 30, 60000, 12)
    assert info['threads'] == 512, info


def test_resident_rollout_config5_geometry_and_host_loop_route(monkeypatch):
    """The 1024-thread instantiation (NP 128 / D 40) of the resident kernel, and the one-launch-per-generation route other geometries
    take behind the same entry point (NP 60 / D 10 here, and MBX_ROLLOUT_PER_GENERATION=1)."""
    ps40 = (1, 8, 15, 21)
    _rollout_case('bbob', 40, ps40, 128, 8, (2, 5), maxfes=80000)
    _rollout_case('bbob', 30, (1, 7, 16, 22), 100, 8, (3, 9), maxfes=60000)          # k_rlepso_run<512, 100, 30, 5>
    _rollout_case('bbob', 10, (1, 16), 60, 8, (3, 4), resident=False)
    monkeypatch.setenv('MBX_ROLLOUT_PER_GENERATION', '1')           # read when the batch is created
    _rollout_case('bbob', 10, (1, 21), 100, 8, (5, 2), resident=False)


def test_config5_shape_np128_dim40_mixed_suites():
    """BASELINE.json config 5 geometry: RLEPSO on bbob (24) + bbob-noisy (30) at D = 40 with NP = 128 (5 groups of 25:
    particles 125..127 keep zero coefficients under the reference's NP // n_group rule, rlepso_optimizer.py:117-126)."""
    from metabox_amd.suite import Batch, Suite
    from metabox_amd._abi import ALGO_RLEPSO
    ps = [problems('bbob', 40)[k] for k in sorted(problems('bbob', 40))] + \
         [problems('bbob-noisy', 40)[k] for k in sorted(problems('bbob-noisy', 40))]
    s = Suite(ps)
    B, G, NPc, Dc = len(ps), 6, 128, 40
    maxfes, logi = 80000, 1600
    rs = np.random.RandomState(2)
    actions = rs.uniform(0, 1, size=(G, B, 35)).astype(np.float32)
    seeds = np.arange(B, dtype=np.uint64) + 77
    b = Batch(s, ALGO_RLEPSO, np.arange(B), seeds, NPc, maxfes, logi, 50)
    b.reset()
    for g in range(G):
        b.step(torch.from_numpy(actions[g]).cuda())
    torch.cuda.synchronize()
    cfg = oracle.make_cfg(1, NPc, Dc, maxfes, logi, 50)
    for k in range(B):
        p = ps[k]
        o = oracle.RlepsoOracle(p.desc(), p.bias, cfg, seed=int(seeds[k]))
        o.reset()
        for g in range(G):
            o.step(actions[g, k])
        want = oracle.split_rlepso_state(o.state(), NPc, Dc, 50)
        got = oracle.split_rlepso_state(b.read_state(k), NPc, Dc, 50)
        assert close(got['scalars'][0], want['scalars'][0]), (p.func_id, got['scalars'][0], want['scalars'][0])
        assert got['scalars'][1] == want['scalars'][1]
        if p.noise[0] == 0:
            assert np.abs(got['pos'] - want['pos']).max() <= 1e-9, p.func_id
            assert close(got['pbest'], want['pbest']), p.func_id
        # particles beyond n_group * (NP // n_group) never move on their own (w = c = 0): velocity exactly 0
        assert np.all(got['vel'].reshape(NPc, Dc)[125:] == 0) or got['scalars'][oracle.SC_REINIT] > 0
    b.close()


def test_config5_full_batch_properties():
    """BASELINE.json config 5 at its one-GPU size (8192 instances = 65 536 / 8, 54 functions round-robin, NP 128 / D 40) through the resident
    kernel bench.py times: bit-identical re-run, results independent of batch position / size (every second instance alone reproduces its rows),
    fes accounting, done absorbing across launches."""
    from metabox_amd.suite import Batch, Suite
    from metabox_amd._abi import ALGO_RLEPSO
    ps = _suite54(40)
    s = Suite(ps)
    B, NPc, maxfes = 8192, 128, 80000
    pidx = np.arange(B) % len(ps)
    seeds = np.arange(B, dtype=np.uint64) // len(ps) + 31
    table = torch.rand(maxfes + 2 * NPc + 1, 2, 35, generator=torch.Generator().manual_seed(9)).cuda()
    table[:, 1] = 0.05 + 0.3 * table[:, 1]
    table = table.contiguous()

    def run(sel, chunks=(5, 3)):
        b = Batch(s, ALGO_RLEPSO, pidx[sel], seeds[sel], NPc, maxfes, maxfes // 50, 50)
        assert b.rollout_is_resident() and b.launch_info()['threads'] == 1024
        b.reset()
        for n in chunks:
            b.rlepso_rollout(table, n)
        r = {k: v.cpu().numpy() for k, v in b.results().items()}
        st = np.stack([b.read_state(i) for i in (0, len(sel) // 2, len(sel) - 1)])
        b.close()
        return r, st
    full, sfull = run(np.arange(B))
    again, sagain = run(np.arange(B))
    for k in full:
        assert np.array_equal(full[k], again[k]), k
    assert np.array_equal(sfull, sagain, equal_nan=True)
    half = np.arange(B)[1::2]
    part, spart = run(half)
    for k in full:
        assert np.array_equal(full[k][half], part[k]), k
    assert np.array_equal(sfull[2], spart[2], equal_nan=True)                      # instance B - 1 is the last one of both batches
    one, _ = run(np.arange(B), chunks=(8,))                                        # launch boundaries do not show
    for k in full:
        assert np.array_equal(full[k], one[k]), k
    assert np.all(full['steps'] == 8)
    assert np.all(full['fes'] >= NPc + NPc * full['steps']) and np.all(full['fes'] <= NPc + 2 * NPc * full['steps'])


def test_single_instance_batch_and_error_paths():
    from metabox_amd.suite import Batch, Suite
    from metabox_amd import _abi
    from metabox_amd._abi import ALGO_RLEPSO
    ps = [problems('bbob', 10)[1]]
    s = Suite(ps)
    b = Batch(s, ALGO_RLEPSO, [0], [5], NP, MAXFES, LOGI, NLOG)
    st = b.reset()
    assert st.shape == (1, 1) and float(st[0, 0]) == NP / MAXFES
    for _ in range(3):
        st, r, d = b.step(torch.full((1, 35), 0.5, device='cuda'))
    assert float(st[0, 0]) >= 4 * NP / MAXFES and abs(float(r[0])) == 1.0 and int(d[0]) == 0
    res = b.results()
    assert int(res['steps'][0]) == 3 and int(res['cost_len'][0]) == 2 and res['cost'].shape == (1, 51)   # fes 400 >= 1*400: one log point
    with pytest.raises(_abi.MbxError):
        Batch(s, ALGO_RLEPSO, [3], [5], NP, MAXFES, LOGI, NLOG)          # problem index out of range
    with pytest.raises(_abi.MbxError):
        Batch(s, ALGO_RLEPSO, [0], [5], 1000, MAXFES, LOGI, NLOG)        # population larger than a workgroup
    # mbx_rlepso_rollout: argument checks (same error behaviour as mbx_rlepso_act_step)
    table = torch.full((MAXFES + 2 * NP + 1, 2, 35), 0.3, device='cuda')
    with pytest.raises(_abi.MbxError):
        b.rlepso_rollout(table, 0)                                        # n_gens < 1
    b.set_tape(torch.zeros(1, b.tape_stride, dtype=torch.float64, device='cuda'))
    with pytest.raises(_abi.MbxError):
        b.rlepso_rollout(table, 2)                                        # a replay tape holds one generation and no policy draws
    b.set_tape(None)
    st2, rw2, dn2 = b.rlepso_rollout(table, 2)
    assert int(b.results()['steps'][0]) == 5 and abs(float(rw2[0])) in (0.0, 2.0)   # two more generations, rewards summed
    b.close()


def test_end_to_end_statistics_match_the_reference():
    """Philox-driven batched rollouts with the shipped policy vs 40 reference runs per bbob-easy test problem
    (tests/golden/rlepso_stats.npz): the distributions of final cost and of consumed FEs must be statistically
    indistinguishable (two-sided Mann-Whitney, and mean log-cost within 4 standard errors)."""
    from scipy import stats as sps
    from metabox_amd.agent.rlepso_agent import RLEPSO_Agent
    from metabox_amd.config import get_config
    from metabox_amd.environment import BatchedPBO_Env
    from metabox_amd.optimizer import RLEPSO_Optimizer
    ref = load('rlepso_stats.npz')
    cfg = get_config(['--problem', 'bbob', '--dim', '10', '--device', 'cuda'])
    cfg.agent_save_dir = None
    agent = RLEPSO_Agent(cfg).load_exported_weights(load('rlepso_policy.npz')).to('cuda')
    opt = RLEPSO_Optimizer(cfg)
    fids = [1, 5, 6, 10, 15, 20]
    ps = [problems('bbob', 10)[f] for f in fids]
    runs = 256
    pidx = np.repeat(np.arange(len(fids)), runs)
    torch.manual_seed(123)
    for salt, policy in enumerate(('fused', 'hip', 'table', 'torch')):
        env = BatchedPBO_Env(ps, opt, pidx, np.arange(len(pidx), dtype=np.uint64) * 2654435761 + 11 + salt)
        out = agent.rollout_batch(env, policy=policy)
        cost = out['cost'][:, -1].cpu().numpy().reshape(len(fids), runs)
        fes = out['fes'].cpu().numpy().reshape(len(fids), runs)
        env.close()
        for k, f in enumerate(fids):
            rc, rf = ref[f'{f}/final_cost'], ref[f'{f}/fes']
            if f == 5:                                   # Linear_Slope: solved exactly (cost 0) in every run of both
                assert np.all(cost[k] == 0) and np.all(rc == 0)
            else:
                lg, lr = np.log10(cost[k] + 1e-12), np.log10(rc + 1e-12)
                se = np.sqrt(lg.var() / runs + lr.var() / len(lr))
                assert abs(lg.mean() - lr.mean()) <= 4 * se + 1e-3, (f, lg.mean(), lr.mean(), se)
                assert sps.mannwhitneyu(cost[k], rc).pvalue > 1e-3, f
            if f in (1, 5):                              # early-stopping problems: FEs to reach 1e-8
                se = np.sqrt(fes[k].var() / runs + rf.var() / len(rf))
                assert abs(fes[k].mean() - rf.mean()) <= 4 * se, (f, fes[k].mean(), rf.mean())
                assert sps.mannwhitneyu(fes[k], rf).pvalue > 1e-3, f
            else:
                assert fes[k].min() >= 20000 and fes[k].max() < 20200


@pytest.mark.parametrize('dim', [5, 7])
def test_odd_dimension_single_coordinate_work_items(dim):
    """Odd D takes the one-coordinate work-item path of the move phase (even D pairs coordinates): HIP == oracle under Philox for
    RLEPSO, and the other generation kernels run at the same dimension."""
    from metabox_amd.suite import Batch, Suite
    from metabox_amd._abi import ALGO_GLEET, ALGO_LDE, ALGO_RLEPSO
    ps = problems('bbob', dim)
    ids = [1, 3, 8, 15, 17, 21]
    s = Suite([ps[i] for i in ids])
    B, G = len(ids), 25
    rs = np.random.RandomState(dim)
    actions = rs.uniform(0, 1, size=(G, B, 35)).astype(np.float32)
    seeds = np.arange(B, dtype=np.uint64) * 104729 + 5
    maxfes, nlog = 2000 * dim, 50
    batch = Batch(s, ALGO_RLEPSO, np.arange(B), seeds, NP, maxfes, maxfes // nlog, nlog)
    batch.reset()
    ledger, exact_until = _hip_vs_oracle(batch, s.problems, seeds, actions, dim, maxfes, maxfes // nlog, nlog, f'bbob d={dim}')
    print_ledger(ledger)
    batch.close()
    for algo, np_, adim in ((ALGO_LDE, 50, 100), (ALGO_GLEET, 100, 100)):
        bt = Batch(s, algo, np.arange(B), seeds, np_, maxfes, maxfes // nlog, nlog)
        st0 = bt.reset().clone()
        st1, r, d = bt.step(torch.rand(B, adim, device='cuda'))
        assert torch.isfinite(st1).all() and torch.isfinite(r).all() and not torch.equal(st0, st1)
        bt.close()


def test_compile_time_geometry_kernel_equals_generic_kernel(env, monkeypatch):
    """NP = 100 / D = 10 / 5 groups (BASELINE configs 1-2) and NP = 128 / D = 40 / 5 groups (config 5) run instantiations of
    k_rlepso_step with the geometry fixed at compile time; with MBX_GENERIC_GEOMETRY=1 the batch keeps the run-time-geometry kernel.
    Same arithmetic: every state word must be identical (60 generations on bbob and bbob-noisy at D = 10, 8 at D = 40)."""
    from metabox_amd.suite import Batch, Suite
    from metabox_amd._abi import ALGO_RLEPSO
    c5 = [problems('bbob', 40)[k] for k in sorted(problems('bbob', 40))] + \
         [problems('bbob-noisy', 40)[k] for k in sorted(problems('bbob-noisy', 40))]
    c30 = [problems('bbob', 30)[k] for k in sorted(problems('bbob', 30))]            # bbob --dim 30, NP = 100: two 512-thread workgroups per CU,
    cases = [(env['bbob'][0], 24, 2, 60, NP, MAXFES, LOGI), (env['bbob-noisy'][0], 30, 2, 60, NP, MAXFES, LOGI),   # maps read from global memory
             (Suite(c5), len(c5), 1, 8, 128, 80000, 1600), (Suite(c30), len(c30), 1, 12, 100, 60000, 1200)]
    for s, n, reps, G, np_, maxfes, logi in cases:
        B = reps * n
        pidx = np.arange(B) % n
        seeds = np.arange(B, dtype=np.uint64) * 7919 + 11
        actions = torch.rand(G, B, 35, generator=torch.Generator().manual_seed(3)).cuda()
        states = []
        for generic in ('0', '1'):
            monkeypatch.setenv('MBX_GENERIC_GEOMETRY', generic)
            batch = Batch(s, ALGO_RLEPSO, pidx, seeds, np_, maxfes, logi, NLOG)
            batch.reset()
            for g in range(G):
                batch.step(actions[g])
            states.append(np.stack([batch.read_state(b) for b in range(B)]))
            batch.close()
        assert np.array_equal(states[0], states[1], equal_nan=True), (np_, n)


def test_launch_info_routes_baseline_geometries_to_their_instantiations(env):
    """mbx_batch_launch_info: the BASELINE.json geometries run the compile-time-geometry kernels, anything else the generic ones, and
    the LDS footprints are the ones DESIGN.md's resident-workgroup arithmetic uses (5 x 30.5 KB RLEPSO workgroups per 160 KB CU)."""
    from metabox_amd.suite import Batch
    from metabox_amd._abi import ALGO_RLEPSO, ALGO_GLEET
    s, ids = env['bbob']
    b = Batch(s, ALGO_RLEPSO, [0, 1], [1, 2], NP, MAXFES, LOGI, NLOG)
    info = b.launch_info()
    assert info['threads'] == 256 and info['fixed_geometry'] == 1 and 5 * info['lds_bytes'] <= 160 * 1024, info
    assert info['state_doubles'] >= len(b.read_state(0))
    b.close()
    from metabox_amd.suite import Suite
    s30 = Suite([problems('bbob', 30)[1], problems('bbob', 30)[16]])
    b = Batch(s30, ALGO_RLEPSO, [0, 1], [1, 2], NP, 60000, 1200, NLOG)          # bbob --dim 30: no maps in LDS, two 512-thread workgroups per CU
    info = b.launch_info()
    assert info['threads'] == 512 and info['fixed_geometry'] == 7 and 2 * info['lds_bytes'] <= 160 * 1024, info
    b.close()
    b = Batch(s, ALGO_RLEPSO, [0, 1], [1, 2], 60, MAXFES, LOGI, NLOG)          # not a BASELINE geometry
    assert b.launch_info()['fixed_geometry'] == 0
    b.close()
    b = Batch(s, ALGO_GLEET, [0, 1], [1, 2], NP, MAXFES, LOGI, NLOG)
    assert b.launch_info()['fixed_geometry'] == 5
    b.close()


def test_rebind_and_read_public_equal_a_fresh_batch(env):
    """mbx_batch_rebind (new problems / seeds for an existing batch, what a second PBO_Env.reset() on the same optimizer object needs)
    gives exactly the trajectory of a freshly created batch, and mbx_read_public returns the scalar block + cost list that
    mbx_debug_read_state shows."""
    from metabox_amd.suite import Batch
    from metabox_amd._abi import ALGO_RLEPSO
    s, ids = env['bbob']
    B, G = 6, 8
    acts = torch.rand(G, B, 35, generator=torch.Generator().manual_seed(4)).cuda()
    p1, s1 = np.arange(B) % len(ids), np.arange(B, dtype=np.uint64) + 77
    p2, s2 = (np.arange(B) * 5 + 3) % len(ids), np.arange(B, dtype=np.uint64) * 13 + 5

    def run(batch):
        batch.reset()
        for g in range(G):
            batch.step(acts[g])
        torch.cuda.synchronize()
        return np.stack([batch.read_state(b) for b in range(B)])
    reused = Batch(s, ALGO_RLEPSO, p1, s1, NP, MAXFES, LOGI, NLOG)
    first = run(reused)
    reused.rebind(p2, s2)
    second = run(reused)
    fresh = Batch(s, ALGO_RLEPSO, p2, s2, NP, MAXFES, LOGI, NLOG)
    want = run(fresh)
    assert not np.array_equal(first, want)
    assert np.array_equal(second, want, equal_nan=True)
    off = 3 * NP * D + 3 * NP + D
    for b in (0, B - 1):
        pub = reused.read_public(b).copy()
        full = want[b][off:off + 16 + NLOG + 1]
        n = int(pub[oracle.SC_COST_LEN])
        assert np.array_equal(pub[:16], full[:16]) and np.array_equal(pub[16:16 + n], full[16:16 + n])
    reused.close(); fresh.close()


def test_move_phase_draws_are_uniform():
    """the element-wise uniforms of the move phase are 32-bit (a / 2^32) and the oracle draws them the same way, so oracle parity cannot
    see a distribution error.  This test does not use the oracle: it reads the kernel-side conversion (mbx_debug_rlepso_draws = rl_move's own
    code path) for many (seed, generation) pairs and tests it against U(0, 1) / the discrete uniform on [0, NP) the reference draws from
    (np.random.rand / np.random.randint, src/optimizer/rlepso_optimizer.py:76-109)."""
    import ctypes as C
    from scipy import stats
    from metabox_amd import _abi
    lib = _abi.load_lib()
    out = torch.empty(NP * D, 4, dtype=torch.float64, device='cuda')
    chunks = []
    for k in range(200):                                            # 200 (seed, generation, episode) triples x 1000 elements
        _abi.check(lib.mbx_debug_rlepso_draws(C.c_uint64(1000003 * k + 17), 1 + k % 199, 1 + k // 50, NP, D, C.c_void_p(out.data_ptr()), None))
        torch.cuda.synchronize()
        chunks.append(out.cpu().numpy().copy())
    v = np.concatenate(chunks)                                      # [200000, 4]
    n = len(v)
    for col, name in ((0, 'CLPSO uniform'), (1, 'FDR weight')):
        u = v[:, col]
        assert u.min() >= 0 and u.max() < 1, name
        assert abs(u.mean() - 0.5) < 4 * np.sqrt(1 / 12 / n), (name, u.mean())                   # 4 sigma
        assert abs(u.var() - 1 / 12) < 4 * np.sqrt(1 / 180 / n), (name, u.var())                 # Var(U^2-ish estimator) = 1/180 n
        assert stats.kstest(u, 'uniform').pvalue > 1e-4, name
        assert abs(np.corrcoef(u[:-1], u[1:])[0, 1]) < 4 / np.sqrt(n), name                       # neighbours in the stream
        # the mask P(u <= pci) that CLPSO takes from it (pci in [0.05, 0.5]) is right at the small end too, where 32 vs 53 bits would show first
        for p in (0.05, 0.001):
            k = (u <= p).sum()
            assert abs(k - n * p) < 5 * np.sqrt(n * p * (1 - p)), (name, p, k)
    assert abs(np.corrcoef(v[:, 0], v[:, 1])[0, 1]) < 4 / np.sqrt(n)                             # the two uniforms of one element
    # distinct values: 32-bit draws from a 200 000-sample set collide ~ n^2 / 2^33 = 4.7 times; 24-bit or worse would collide thousands of times
    assert n - len(np.unique(v[:, 0])) < 40
    for col in (2, 3):
        t = v[:, col]
        assert np.all(t == np.floor(t)) and t.min() >= 0 and t.max() <= NP - 1
        cnt = np.bincount(t.astype(np.int64), minlength=NP)
        assert stats.chisquare(cnt).pvalue > 1e-4, (col, cnt.min(), cnt.max())
    both = v[:, 2].astype(np.int64) * NP + v[:, 3].astype(np.int64)                                # the pair is uniform on [0, NP)^2
    assert stats.chisquare(np.bincount(both, minlength=NP * NP)).pvalue > 1e-4
    # numpy's own generator passes the same battery (the thresholds are not vacuous or over-tight)
    rs = np.random.RandomState(0)
    u = rs.rand(n)
    assert stats.kstest(u, 'uniform').pvalue > 1e-4 and stats.chisquare(np.bincount(rs.randint(0, NP, n), minlength=NP)).pvalue > 1e-4


def test_snapshot_and_resume_of_instances_is_bit_exact(env):
    """mbx_debug_read_state / mbx_debug_write_state = what `copy.deepcopy(env)` or a pickled optimizer is in the reference: a run that is snapshotted after 15
    generations, continued for 20, rolled back to the snapshot and continued again repeats itself bit for bit (the Philox counters -- generation and episode --
    travel in the state block), through the one-generation route and through the resident kernel."""
    from metabox_amd.suite import Batch
    from metabox_amd._abi import ALGO_RLEPSO
    s, ids = env['bbob']
    B = 24
    seeds = np.arange(B, dtype=np.uint64) * 31 + 7
    b = Batch(s, ALGO_RLEPSO, np.arange(B), seeds, NP, MAXFES, LOGI, NLOG)
    table = torch.rand(MAXFES + 2 * NP + 1, 2, 35, generator=torch.Generator().manual_seed(1)).cuda()
    table[:, 1] = 0.05 + 0.3 * table[:, 1]
    table = table.contiguous()
    b.reset()
    b.rlepso_rollout(table, 15)
    torch.cuda.synchronize()
    snap = [b.read_state(k) for k in range(B)]
    _, _, _, t1 = b.rlepso_rollout(table, 20, trajectory=True)
    t1 = {k: v.clone() for k, v in t1.items()}
    end1 = [b.read_state(k) for k in range(B)]
    for k in range(B):
        b.write_state(k, snap[k])
    assert all(np.array_equal(b.read_state(k), snap[k], equal_nan=True) for k in range(B))
    _, _, _, t2 = b.rlepso_rollout(table, 20, trajectory=True)
    for key in t1:
        assert torch.equal(t1[key], t2[key]), key
    assert all(np.array_equal(b.read_state(k), end1[k], equal_nan=True) for k in range(B))
    for k in range(B):                                       # ... and once more, one launch per generation
        b.write_state(k, snap[k])
    for g in range(20):
        sb, rb, db, acts = b.act_step(table, want_actions=True)
        assert torch.equal(t1['reward'][g], rb) and torch.equal(t1['done'][g], db), g
    assert all(np.array_equal(b.read_state(k), end1[k], equal_nan=True) for k in range(B))
    b.close()


def test_launch_clock_slots_report_the_shader_clock(env):
    """mbx_debug_clock_slots (include/mbx.h): while a slot pair is attached, every workgroup of the resident kernel adds its own lifetime in shader cycles (s_memtime) and in
    100 MHz ticks (s_memrealtime) to the two words; their ratio is the clock the launch ran at.  Results are unaffected; detached, nothing is written."""
    import ctypes as C
    from metabox_amd.suite import Batch
    s, ids = env['bbob']
    B = 512
    pidx, seeds = np.arange(B) % len(ids), np.arange(B, dtype=np.uint64) + 9
    table = torch.rand(MAXFES + 2 * NP + 1, 2, 35, generator=torch.Generator().manual_seed(2)).cuda()
    table[:, 1] = 0.05 + 0.2 * table[:, 1]
    table = table.contiguous()
    a = Batch(s, 1, pidx, seeds, NP, MAXFES, MAXFES // 50, 50)
    b = Batch(s, 1, pidx, seeds, NP, MAXFES, MAXFES // 50, 50)
    slots = torch.zeros(2, dtype=torch.int64, device='cuda')
    a.reset(); b.reset()
    a.lib.mbx_debug_clock_slots(a._h, C.c_void_p(slots.data_ptr()))
    a.rlepso_rollout(table, 12)
    a.lib.mbx_debug_clock_slots(a._h, None)
    b.rlepso_rollout(table, 12)
    torch.cuda.synchronize()
    cyc, ticks = (int(x) for x in slots.cpu())
    ghz = cyc / (ticks * 10.)
    assert ticks > B and 1.2 < ghz <= 2.45, (cyc, ticks, ghz)                     # MI355X: <= 2.4 GHz
    before = slots.clone()
    a.rlepso_rollout(table, 3); b.rlepso_rollout(table, 3)                        # detached: the words stay as they are
    torch.cuda.synchronize()
    assert torch.equal(slots, before)
    for k in range(0, B, 37):
        assert np.array_equal(a.read_state(k), b.read_state(k), equal_nan=True), k
    a.close(); b.close()
