"""RL-PSO: the C oracle replays the reference's episodes (CPU); the HIP step kernel replays them too, matches
the oracle under Philox, and the fused policy + multi-step rollout equals the step-by-step route (GPU)."""
import numpy as np
import pytest

from helpers import close, load, problems
from oracle import oracle

TR = load('rlpso_traces.npz')
POL = load('rlpso_policy.npz')
CASES = [str(c) for c in TR['cases']]
NP = 100
ALGO_RLPSO = 5


def _setup(case):
    suite, dim, fid, seed = case.split('/')
    dim, seed = int(dim), int(seed)
    if suite == 'protein':
        from test_protein import protein
        p = protein()[0][fid]
        return p, None, dim, 1000, 5, 0, seed
    p = problems(suite, dim)[int(fid)]
    return p, p.bias, dim, 2500, 50, p.noise[0], seed


def _check(case, gb, rw, dn, states, cost, clen, fin, fes):
    assert close(gb, TR[f'{case}/gbest']), case
    ref_r = TR[f'{case}/reward']
    assert np.all(np.abs(rw - ref_r) <= 1e-5 * np.abs(ref_r) + 1e-9), (case, int(np.argmax(np.abs(rw - ref_r))))
    assert np.array_equal(dn, TR[f'{case}/done']), case
    for row in TR[f'{case}/states']:
        got, want = states[int(row[0])], row[1:]
        assert np.all(np.abs(got - want) <= 1e-9 * np.abs(want) + 1e-11), (case, int(row[0]))
    ref_cost = TR[f'{case}/cost']
    assert clen == len(ref_cost) and close(cost[:clen], ref_cost), case
    assert np.abs(fin['pos'] - TR[f'{case}/final_pos'].ravel()).max() <= 1e-11, case
    assert np.abs(fin['vel'] - TR[f'{case}/final_vel'].ravel()).max() <= 1e-11, case
    assert close(fin['pbest'], TR[f'{case}/final_pbest']) and close(fin['ccost'], TR[f'{case}/final_ccost']), case
    assert fes == TR[f'{case}/fes']
    assert abs(fin['scalars'][oracle.SC_RLPSO_W] - TR[f'{case}/w']) <= 1e-9       # the inertia that decays every step


@pytest.mark.parametrize('case', CASES)
def test_oracle_replays_reference_rlpso_episode(case):
    p, opt, dim, maxfes, nlog, nk, seed = _setup(case)
    cfg = oracle.make_cfg(ALGO_RLPSO, NP, dim, maxfes, maxfes // nlog, nlog)
    o = oracle.RlpsoOracle(p.desc(), opt, cfg)
    fd = oracle.RlpsoTapeFeeder(seed, NP, dim, nk)
    states = {-1: o.reset(fd.reset_tape())}
    acts = TR[f'{case}/actions']
    G = len(acts)
    gb, rw, dn = np.zeros(G), np.zeros(G), np.zeros(G, bool)
    for g in range(G):
        s, r, d = o.step(acts[g], fd.step_tape())
        gb[g] = oracle.split_rlpso_state(o.state(), NP, dim, nlog)['scalars'][0]
        rw[g], dn[g], states[g] = r, d, s
    st = oracle.split_rlpso_state(o.state(), NP, dim, nlog)
    _check(case, gb, rw, dn, states, st['cost'], int(st['scalars'][3]), st, st['scalars'][1])


@pytest.mark.gpu
def test_hip_rlpso_tape_replay_matches_reference():
    import torch
    from metabox_amd.suite import Batch, Suite
    for case in CASES:
        p, opt, dim, maxfes, nlog, nk, seed = _setup(case)
        s = Suite([p])
        b = Batch(s, ALGO_RLPSO, [0], [0], NP, maxfes, maxfes // nlog, nlog)
        assert (b.state_dim, b.action_dim) == (2 * dim, 1)
        fd = oracle.RlpsoTapeFeeder(seed, NP, dim, nk)
        b.set_tape(torch.from_numpy(fd.reset_tape()[None]).cuda())
        states = {-1: b.reset()[0].cpu().numpy().copy()}
        acts = TR[f'{case}/actions']
        G = len(acts)
        rw, dn = np.zeros(G), np.zeros(G, bool)
        want_state = {int(r[0]) for r in TR[f'{case}/states']}
        tape_dev = torch.empty(1, b.tape_stride, dtype=torch.float64, device='cuda')
        acts_dev = torch.from_numpy(acts.astype(np.float32)).cuda()
        rewards, dones = [], []
        for g in range(G):
            tape_dev.copy_(torch.from_numpy(fd.step_tape()[None]))
            b.set_tape(tape_dev)
            st, r, d = b.step(acts_dev[g:g + 1])
            rewards.append(r.clone()); dones.append(d.clone())
            if g in want_state:
                states[g] = st[0].cpu().numpy().copy()
        rw[:] = torch.cat(rewards).cpu().numpy(); dn[:] = torch.cat(dones).cpu().numpy() != 0
        res = b.results()
        fin = oracle.split_rlpso_state(b.read_state(0), NP, dim, nlog)
        gb = TR[f'{case}/gbest'].copy()                          # per-step gbest is pinned through rewards / states; check the final one
        gb[-1] = fin['scalars'][0]
        _check(case, gb, rw, dn, states, res['cost'][0].cpu().numpy(), int(res['cost_len'][0].item()), fin, fin['scalars'][1])
        assert int(res['steps'][0].item()) == G
        b.close()


@pytest.mark.gpu
def test_hip_rlpso_philox_parity_with_oracle():
    import torch
    from metabox_amd.suite import Batch, Suite
    ps = problems('bbob-noisy', 10)
    ids = sorted(ps)
    s = Suite([ps[i] for i in ids])
    B, G = len(ids), 230                                        # > 2 sweeps over the swarm
    rs = np.random.RandomState(11)
    actions = (rs.rand(G, B) * 1.4 - 0.2).astype(np.float32)
    seeds = np.arange(B, dtype=np.uint64) * 37 + 9
    b = Batch(s, ALGO_RLPSO, np.arange(B), seeds, NP, 2500, 50, 50)
    st0 = b.reset().cpu().numpy().copy()
    hist = []
    for g in range(G):
        st, r, d = b.step(torch.from_numpy(actions[g]).cuda())
        hist.append((st.cpu().numpy().copy(), r.cpu().numpy().copy(), d.cpu().numpy().copy()))
    cfg = oracle.make_cfg(ALGO_RLPSO, NP, 10, 2500, 50, 50)
    for k in range(B):
        p = s.problems[k]
        o = oracle.RlpsoOracle(p.desc(), p.bias, cfg, seed=int(seeds[k]))
        f0 = o.reset()
        assert np.all(np.abs(f0 - st0[k]) <= 1e-12 * np.abs(f0) + 1e-13), ids[k]
        for g in range(G):
            f, rew, d = o.step(actions[g, k])
            got = hist[g][0][k]
            assert np.all(np.abs(f - got) <= 1e-9 * np.abs(f) + 1e-11), (ids[k], g)
            assert abs(rew - hist[g][1][k]) <= 1e-5 * abs(rew) + 1e-9, (ids[k], g)
            assert bool(hist[g][2][k]) == d
        fin, ref = oracle.split_rlpso_state(b.read_state(k), NP, 10, 50), oracle.split_rlpso_state(o.state(), NP, 10, 50)
        assert close(fin['pbest'], ref['pbest']) and close(fin['scalars'][:7], ref['scalars'][:7]), ids[k]
        assert np.abs(fin['pbpos'] - ref['pbpos']).max() <= 1e-9
    b.close()


def _agent(dim=10):
    from metabox_amd.agent import RL_PSO_Agent
    from metabox_amd.config import get_config
    cfg = get_config(['--problem', 'bbob', '--dim', str(dim), '--device', 'cuda'])
    cfg.agent_save_dir = None
    return RL_PSO_Agent(cfg).load_exported_weights(POL).to('cuda'), cfg


@pytest.mark.gpu
def test_rlpso_policy_kernel_and_fused_rollout():
    """mbx_gauss_policy (RL-PSO heads) == the PyTorch modules == the reference's recorded (state -> mu, sigma); the fused multi-step
    rollout (actor inside the step kernel, 64 steps per launch) == mbx_gauss_policy + mbx_step per step, bit for bit."""
    import torch
    from metabox_amd.environment import BatchedPBO_Env
    from metabox_amd.optimizer import RL_PSO_Optimizer
    agent, cfg = _agent()
    cfg.maxFEs, cfg.log_interval = 700, 14
    nets = agent.nets
    h1, h2 = nets.hidden_sizes()
    net = (nets.packed_weights(), h1, h2, nets.min_sigma, nets.max_sigma)
    assert (h1, h2) == (32, 8)
    with torch.no_grad():
        mu, sg = nets.distribution(torch.from_numpy(POL['io/x']).cuda())
    assert np.allclose(mu.cpu().numpy(), POL['io/mu'], atol=2e-6) and np.allclose(sg.cpu().numpy(), POL['io/sigma'], atol=2e-6)
    ps = [problems('bbob', 10)[f] for f in (1, 8, 16, 21)]
    B = 96
    pidx, seeds = np.arange(B) % len(ps), np.arange(B, dtype=np.uint64) * 101 + 1
    env_a = BatchedPBO_Env(ps, RL_PSO_Optimizer(cfg), pidx, seeds)
    env_b = BatchedPBO_Env(ps, RL_PSO_Optimizer(cfg), pidx, seeds)
    state = env_b.reset(); env_a.reset()
    n_io = POL['io/x'].shape[0]
    orig = state.clone()                                         # `state` IS the batch's state tensor
    probe = state.clone(); probe[:n_io] = torch.from_numpy(POL['io/x']).cuda().double()
    env_b.batch.state.copy_(probe)
    _, ms = env_b.batch.gauss_policy(*net, want_mu_sigma=True)
    assert np.allclose(ms[:n_io, 0, 0].cpu().numpy(), POL['io/mu'][:, 0], atol=2e-6)
    assert np.allclose(ms[:n_io, 1, 0].cpu().numpy(), POL['io/sigma'][:, 0], atol=2e-6)
    env_b.batch.state.copy_(orig)
    ret_b = torch.zeros(B, dtype=torch.float64, device='cuda')
    acts_seen = []
    for g in range(600):
        a = env_b.batch.gauss_policy(*net)
        acts_seen.append(a[:, 0].clone())
        _, r, _ = env_b.step(a)
        ret_b += r
    ret_a = torch.zeros(B, dtype=torch.float64, device='cuda')
    for launch in range(10):                                     # 9 x 64 + 24 steps
        _, r, _ = env_a.batch.rlpso_rollout(*net, 64 if launch < 9 else 24)
        ret_a += r
    ra, rb = env_a.results(), env_b.results()
    for key in ra:
        assert torch.equal(ra[key], rb[key]), key
    assert torch.equal(env_a.batch.state, env_b.batch.state)
    assert torch.allclose(ret_a, rb['return'], rtol=1e-12, atol=1e-12) and torch.allclose(ret_b, rb['return'], rtol=1e-12, atol=1e-12)
    assert bool((ra['fes'] == 700).all()) and int(ra['steps'].max()) == 600
    acts = torch.stack(acts_seen)
    assert float(acts.min()) > -0.3 and float(acts.max()) < 1.2 and float(acts.std()) > 0.01        # re-folded, not clamped
    # rollout_batch drives the same loop
    out = agent.rollout_batch(BatchedPBO_Env(ps, RL_PSO_Optimizer(cfg), pidx, seeds), chunk=100)
    assert torch.equal(out['cost'], ra['cost']) and torch.equal(out['return'], ra['return'])
    env_a.close(); env_b.close()


@pytest.mark.gpu
def test_rlpso_single_env_protocol():
    """PBO_Env(problem, RL_PSO_Optimizer) + RL_PSO_Agent.rollout_episode: the reference's loop over the B = 1 view."""
    from metabox_amd.environment import PBO_Env
    from metabox_amd.optimizer import RL_PSO_Optimizer
    agent, cfg = _agent()
    cfg.maxFEs, cfg.log_interval, cfg.n_logpoint = 160, 10, 16
    np.random.seed(3)
    p = problems('bbob', 10)[1]
    opt = RL_PSO_Optimizer(cfg)
    env = PBO_Env(p, opt)
    info = agent.rollout_episode(env)
    assert info['fes'] == 160 and len(info['cost']) == 17 and info['cost'][0] >= info['cost'][-1] and np.isfinite(info['return'])
    s = env.reset()
    assert s.shape == (20,) and opt.fes == 100 and len(opt.cost) == 1
    s2, r, d = env.step(np.array([0.4], dtype=np.float32))
    assert s2.shape == (20,) and opt.fes == 101 and not d and np.isfinite(r)
