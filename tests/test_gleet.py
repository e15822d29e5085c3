"""GLEET optimizer: the C oracle replays the reference's episodes (CPU); the fused HIP generation kernel with the
9-feature epilogue replays them too and matches the oracle under Philox (GPU)."""
import numpy as np
import pytest

from helpers import close, load, problems
from oracle import oracle

TR = load('gleet_traces.npz')
CASES = [str(c) for c in TR['cases']]
NP = 100
ALGO_GLEET = 6


def _setup(case):
    suite, dim, fid, seed = case.split('/')
    dim, seed = int(dim), int(seed)
    if suite == 'protein':
        from test_protein import protein
        p = protein()[0][fid]
        return p, None, dim, 1000, 5, 0, seed
    p = problems(suite, dim)[int(fid)]
    return p, p.bias, dim, 2000 * dim, 50, p.noise[0], seed


def _actions(seed, G):
    """The float32 actions tools/gen_golden.py (gleet section) fed to the reference, regenerated from the same seed."""
    ars = np.random.RandomState(50_000 + seed)
    return [ars.rand(NP).astype(np.float32) for _ in range(G)]


def _feat_close(got, want):
    return np.all(np.abs(got - want) <= 1e-7 * np.abs(want) + 1e-9)


def _check(case, gb, rw, dn, states, cost, clen, fin, fes):
    assert close(gb, TR[f'{case}/gbest']), case
    ref_r = TR[f'{case}/reward']
    assert np.all(np.abs(rw - ref_r) <= 1e-5 * np.abs(ref_r) + 1e-9), (case, int(np.argmax(np.abs(rw - ref_r))))
    assert np.array_equal(dn, TR[f'{case}/done']), case
    for g, want in zip(TR[f'{case}/state_gen'], TR[f'{case}/states']):
        assert _feat_close(states[int(g)], want), (case, int(g))
    for g, want in zip(TR[f'{case}/sub_gen'], TR[f'{case}/sub_states']):
        assert _feat_close(states[int(g)][::11], want), (case, int(g))
    ref_cost = TR[f'{case}/cost']
    assert clen == len(ref_cost) and close(cost[:clen], ref_cost), case
    assert np.abs(fin['pos'] - TR[f'{case}/final_pos'].ravel()).max() <= 1e-10, case
    assert close(fin['pbest'], TR[f'{case}/final_pbest']) and np.array_equal(fin['pni'], TR[f'{case}/final_pni']), case
    assert fes == TR[f'{case}/fes'] and abs(fin['scalars'][oracle.SC_GLEET_W] - TR[f'{case}/w']) <= 1e-12
    assert fin['scalars'][oracle.SC_GLEET_NOIMPROVE] == TR[f'{case}/no_improve']


@pytest.mark.parametrize('case', CASES)
def test_oracle_replays_reference_gleet_episode(case):
    p, opt, dim, maxfes, nlog, nk, seed = _setup(case)
    cfg = oracle.make_cfg(ALGO_GLEET, NP, dim, maxfes, maxfes // nlog, nlog)
    o = oracle.GleetOracle(p.desc(), opt, cfg)
    fd = oracle.GleetTapeFeeder(seed, NP, dim, nk)
    states = {-1: o.reset(fd.reset_tape())}
    G = len(TR[f'{case}/gbest'])
    acts = _actions(seed, G)
    gb, rw, dn = np.zeros(G), np.zeros(G), np.zeros(G, bool)
    for g in range(G):
        s, r, d = o.step(acts[g], fd.step_tape())
        gb[g] = oracle.split_gleet_state(o.state(), NP, dim, nlog)['scalars'][0]
        rw[g], dn[g], states[g] = r, d, s
    st = oracle.split_gleet_state(o.state(), NP, dim, nlog)
    _check(case, gb, rw, dn, states, st['cost'], int(st['scalars'][3]), st, st['scalars'][1])


@pytest.mark.gpu
def test_hip_gleet_tape_replay_matches_reference():
    import torch
    from metabox_amd.suite import Batch, Suite
    for case in CASES:
        p, opt, dim, maxfes, nlog, nk, seed = _setup(case)
        s = Suite([p])
        b = Batch(s, ALGO_GLEET, [0], [0], NP, maxfes, maxfes // nlog, nlog)
        assert (b.state_dim, b.action_dim) == (27 * NP, NP)
        fd = oracle.GleetTapeFeeder(seed, NP, dim, nk)
        b.set_tape(torch.from_numpy(fd.reset_tape()[None]).cuda())
        states = {-1: b.reset()[0].cpu().numpy().reshape(NP, 27).copy()}
        G = len(TR[f'{case}/gbest'])
        acts = _actions(seed, G)
        gb, rw, dn = np.zeros(G), np.zeros(G), np.zeros(G, bool)
        tape_dev = torch.empty(1, b.tape_stride, dtype=torch.float64, device='cuda')
        sc_off = oracle.gleet_state_doubles(NP, dim, nlog) - (nlog + 1) - 16
        for g in range(G):
            tape_dev.copy_(torch.from_numpy(fd.step_tape()[None]))
            b.set_tape(tape_dev)
            st, r, d = b.step(torch.from_numpy(acts[g][None]).cuda())
            rw[g] = r[0].item(); dn[g] = bool(d[0].item())
            states[g] = st[0].cpu().numpy().reshape(NP, 27).copy()
            gb[g] = b.read_state(0)[sc_off]
        res = b.results()
        fin = oracle.split_gleet_state(b.read_state(0), NP, dim, nlog)
        _check(case, gb, rw, dn, states, res['cost'][0].cpu().numpy(), int(res['cost_len'][0].item()), fin, fin['scalars'][1])
        b.close()


@pytest.mark.gpu
def test_hip_gleet_philox_parity_with_oracle():
    import torch
    from metabox_amd.suite import Batch, Suite
    ps = problems('bbob-noisy', 10)
    ids = sorted(ps)
    s = Suite([ps[i] for i in ids])
    B, G = len(ids), 40
    rs = np.random.RandomState(21)
    actions = rs.rand(G, B, NP).astype(np.float32)
    seeds = np.arange(B, dtype=np.uint64) * 53 + 17
    b = Batch(s, ALGO_GLEET, np.arange(B), seeds, NP, 20000, 400, 50)
    st0 = b.reset().cpu().numpy().reshape(B, NP, 27).copy()
    hist = []
    for g in range(G):
        st, r, d = b.step(torch.from_numpy(actions[g]).cuda())
        hist.append((st.cpu().numpy().reshape(B, NP, 27).copy(), r.cpu().numpy().copy()))
    cfg = oracle.make_cfg(ALGO_GLEET, NP, 10, 20000, 400, 50)
    for k in range(B):
        p = s.problems[k]
        o = oracle.GleetOracle(p.desc(), p.bias, cfg, seed=int(seeds[k]))
        f0 = o.reset()
        assert _feat_close(st0[k], f0), ids[k]
        for g in range(G):
            f, rew, d = o.step(actions[g, k])
            assert _feat_close(hist[g][0][k], f), (ids[k], g)
            assert abs(rew - hist[g][1][k]) <= 1e-5 * abs(rew) + 1e-9, (ids[k], g)
        fin, ref = oracle.split_gleet_state(b.read_state(k), NP, 10, 50), oracle.split_gleet_state(o.state(), NP, 10, 50)
        assert close(fin['pbest'], ref['pbest']) and np.array_equal(fin['pni'], ref['pni']) and close(fin['scalars'][:7], ref['scalars'][:7])
        assert np.abs(fin['pfeat'] - ref['pfeat']).max() <= 1e-7 and np.abs(fin['gfeat'] - ref['gfeat']).max() <= 1e-7
    b.close()


def _agent(device):
    from metabox_amd.agent import GLEET_Agent
    from metabox_amd.config import get_config
    cfg = get_config(['--problem', 'bbob', '--dim', '10', '--device', device])
    cfg.agent_save_dir = None
    return GLEET_Agent(cfg).load_exported_weights(load('gleet_policy.npz')), cfg


def test_gleet_networks_match_reference_io():
    """The restated attention actor / critic load the reference's state_dict unchanged and reproduce its decoder output, (mu, sigma),
    joint log-probability of a fixed action and value (host logic: runs on the CPU)."""
    import torch
    pol = load('gleet_policy.npz')
    agent, cfg = _agent('cpu')
    assert (cfg.embedding_dim, cfg.n_step, cfg.K_epochs, cfg.max_grad_norm) == (16, 10, 3, 0.1)
    assert sum(p.numel() for p in agent.actor.parameters()) == 5426
    x = torch.from_numpy(pol['io/x'])
    with torch.no_grad():
        z = agent.actor(x, only_critic=True)
        mu, sigma = agent.actor.distribution(z)
        _, logp, _ = agent.actor(x, fixed_action=torch.from_numpy(pol['io/fixed']))
        value = agent.critic(z)[0]
    assert np.allclose(z.numpy(), pol['io/z'], atol=2e-5) and np.allclose(mu.numpy(), pol['io/mu'], atol=2e-6)
    assert np.allclose(sigma.numpy(), pol['io/sigma'], atol=2e-6) and np.allclose(logp.numpy(), pol['io/logp'], rtol=1e-5, atol=1e-4)
    assert np.allclose(value.numpy(), pol['io/value'], atol=2e-5)
    a, lp, zc, ent = agent.actor(x, require_entropy=True, to_critic=True)
    assert a.shape == (3, 100, 1) and lp.shape == (3, 1) and zc.shape == (3, 100, 16) and ent.shape == (3, 100, 1)
    assert float(a.min()) >= 0 and float(a.max()) <= 1


@pytest.mark.gpu
def test_gleet_protocol_rollout_and_training(tmp_path):
    """PBO_Env(problem, GLEET_Optimizer) with the reference loops: rollout_episode, a few PPO updates of train_episode, and the
    lock-step rollout_batch; the tester harness accepts the pair by name."""
    import copy
    import torch
    from metabox_amd.environment import BatchedPBO_Env, PBO_Env
    from metabox_amd.optimizer import GLEET_Optimizer
    agent, cfg = _agent('cuda')
    agent.to('cuda')
    small = copy.deepcopy(cfg)
    small.maxFEs, small.log_interval, small.n_logpoint = 1500, 100, 15
    np.random.seed(5); torch.manual_seed(5)
    p = problems('bbob', 10)[1]
    opt = GLEET_Optimizer(small)
    env = PBO_Env(p, opt)
    s = env.reset()
    assert s.shape == (100, 27) and opt.fes == 100 and np.all(s[:, :9] == s[:, 9:18])         # memories start as the features
    s2, r, d = env.step(np.full(100, 0.5, dtype=np.float32))
    assert s2.shape == (100, 27) and opt.fes == 200 and r >= 0 and not d
    info = agent.rollout_episode(env)
    assert info['fes'] == 1500 and len(info['cost']) == 16 and info['cost'][0] >= info['cost'][-1] and info['return'] >= 0
    # PPO: 2 segments x 3 epochs, weights move
    tcfg = copy.deepcopy(small)
    tcfg.max_learning_step, tcfg.save_interval, tcfg.agent_save_dir = 6, 100, str(tmp_path) + '/'
    agent.update_setting(tcfg)
    before = [q.detach().clone() for q in agent.actor.parameters()]
    with torch.enable_grad():
        exceed, tinfo = agent.train_episode(PBO_Env(p, GLEET_Optimizer(small)))
    assert exceed and tinfo['learn_steps'] == 6 and any(not torch.equal(a, b) for a, b in zip(before, agent.actor.parameters()))
    # lock-step batch on the default budget
    ps = [problems('bbob', 10)[f] for f in (1, 8, 21)]
    B = 96
    envb = BatchedPBO_Env(ps, GLEET_Optimizer(cfg), np.arange(B) % 3, np.arange(B, dtype=np.uint64) + 3)
    out = agent.rollout_batch(envb)
    cost = out['cost'].cpu().numpy()
    assert cost.shape == (B, 51) and np.all(cost[:, 0] >= cost[:, -1]) and int(out['steps'].max()) <= 199
    assert float(out['fes'].max()) == 20000 and bool((out['return'] >= 0).all())
    envb.close()


@pytest.mark.gpu
def test_gleet_policy_kernel_equals_attention_modules():
    """mbx_gleet_policy (one workgroup per swarm: embeddings, two attention layers, normalisations, heads, sampling) vs the PyTorch
    modules on the states of a running batch and on the reference's recorded I/O; float32 round-off tolerance."""
    import torch
    from metabox_amd.environment import BatchedPBO_Env
    from metabox_amd.optimizer import GLEET_Optimizer
    agent, cfg = _agent('cuda')
    agent.to('cuda')
    actor = agent.actor
    w = actor.packed_weights()
    assert w.numel() == 5426
    ps = [problems('bbob', 10)[f] for f in (1, 8, 21)]
    B = 67
    env = BatchedPBO_Env(ps, GLEET_Optimizer(cfg), np.arange(B) % 3, np.arange(B, dtype=np.uint64) + 3)
    state = env.reset()
    for g in range(6):
        act, ms = env.batch.gleet_policy(w, actor.min_sigma, actor.max_sigma, want_mu_sigma=True)
        with torch.no_grad():
            mu, sg = actor.distribution(actor.features(state.view(B, 100, 27).to(torch.float32)))
        assert torch.allclose(ms[:, 0], mu[..., 0], atol=2e-4) and torch.allclose(ms[:, 1], sg[..., 0], atol=2e-4), g
        assert float(act.min()) >= 0 and float(act.max()) <= 1
        z = ((act - ms[:, 0]) / ms[:, 1])[(act > 0) & (act < 1)]
        assert z.numel() > 1000 and float(z.abs().max()) < 6.5 and float(z.std()) > 0.3     # un-clamped draws: bounded normal deviates
        again = env.batch.gleet_policy(w, actor.min_sigma, actor.max_sigma).clone()
        assert torch.equal(again, act)
        state, _, _ = env.step(act)
    # the reference's recorded swarms
    pol = load('gleet_policy.npz')
    x = torch.from_numpy(pol['io/x']).cuda().double().reshape(3, 2700)
    env.batch.state[:3] = x
    _, ms = env.batch.gleet_policy(w, actor.min_sigma, actor.max_sigma, want_mu_sigma=True)
    assert np.allclose(ms[:3, 0].cpu().numpy(), pol['io/mu'][..., 0], atol=2e-4) and np.allclose(ms[:3, 1].cpu().numpy(), pol['io/sigma'][..., 0], atol=2e-4)
    env.close()
    out = agent.rollout_batch(BatchedPBO_Env(ps, GLEET_Optimizer(cfg), np.arange(B) % 3, np.arange(B, dtype=np.uint64) + 3))
    assert bool((out['cost'][:, 0] >= out['cost'][:, -1]).all()) and float(out['fes'].max()) == 20000


@pytest.mark.gpu
def test_gleet_batched_ppo_updates_the_policy(tmp_path):
    """train_batch: PPO over a lock-step batch of swarms (n_step = 10, K_epochs = 3) -- losses finite, parameters move, the
    learning-step / checkpoint bookkeeping of the reference is kept, and Trainer.train_batched can drive it."""
    import copy
    import torch
    from metabox_amd.environment import BatchedPBO_Env
    from metabox_amd.optimizer import GLEET_Optimizer
    agent, cfg = _agent('cuda')
    agent.to('cuda')
    small = copy.deepcopy(cfg)
    small.maxFEs, small.log_interval, small.n_logpoint = 2600, 100, 26
    tcfg = copy.deepcopy(small)
    tcfg.max_learning_step, tcfg.save_interval, tcfg.agent_save_dir = 1000, 4, str(tmp_path) + '/'
    agent.update_setting(tcfg)
    ps = [problems('bbob', 10)[f] for f in (1, 8, 15)]
    B = 48
    env = BatchedPBO_Env(ps, GLEET_Optimizer(small), np.arange(B) % 3, np.arange(B, dtype=np.uint64) + 9)
    before = [q.detach().clone() for q in agent.actor.parameters()]
    torch.manual_seed(0)
    with torch.enable_grad():
        exceed, info = agent.train_batch(env, max_updates=6)
    assert not exceed and info['learn_steps'] == 6 and np.isfinite(info['return']) and info['return'] >= 0
    assert info['normalizer'] >= info['gbest'] > 0
    assert any(not torch.equal(a, b) for a, b in zip(before, agent.actor.parameters()))
    assert all(torch.isfinite(q).all() for q in agent.actor.parameters())
    import os
    assert os.path.exists(str(tmp_path / 'checkpoint1.pkl'))            # save_interval = 4 learning steps
    env.close()


@pytest.mark.gpu
def test_gleet_compile_time_geometry_kernel_equals_generic_kernel(monkeypatch):
    """NP = 100 / D = 10 runs k_gleet_step with the geometry fixed at compile time; MBX_GENERIC_GEOMETRY=1 keeps the run-time-geometry
    kernel.  Every state word (and every feature row handed to the policy) must be identical after 40 generations on the noisy suite."""
    import torch
    from metabox_amd.suite import Batch, Suite
    ps = problems('bbob-noisy', 10)
    ids = sorted(ps)
    s = Suite([ps[i] for i in ids])
    B, G = len(ids), 40
    actions = torch.rand(G, B, NP, generator=torch.Generator().manual_seed(6)).cuda()
    seeds = np.arange(B, dtype=np.uint64) * 29 + 3
    states, feats = [], []
    for generic in ('0', '1'):
        monkeypatch.setenv('MBX_GENERIC_GEOMETRY', generic)
        b = Batch(s, ALGO_GLEET, np.arange(B), seeds, NP, 20000, 400, 50)
        b.reset()
        for g in range(G):
            st, _, _ = b.step(actions[g])
        feats.append(st.cpu().numpy().copy())
        states.append(np.stack([b.read_state(k) for k in range(B)]))
        b.close()
    assert np.array_equal(states[0], states[1], equal_nan=True) and np.array_equal(feats[0], feats[1], equal_nan=True)
