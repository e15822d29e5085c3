"""DE-DDQN: the C oracle replays the reference's episodes (CPU); the fused HIP step kernel replays them too and matches
the oracle under Philox (GPU)."""
import numpy as np
import pytest

from helpers import close, load, problems
from oracle import oracle

TR = load('ddqn_traces.npz')
CASES = [str(c) for c in TR['cases']]
NP = 100


def _setup(case):
    suite, dim, fid, seed = case.split('/')
    dim, seed = int(dim), int(seed)
    if suite == 'protein':
        from test_protein import protein
        p = protein()[0][fid]
        return p, None, dim, 1000, 5, 0, seed
    p = problems(suite, dim)[int(fid)]
    return p, p.bias, dim, 3000, 50, p.noise[0], seed


def _check(case, gb, rw, dn, feats, cost, clen, X, fes):
    assert close(gb, TR[f'{case}/gbest']), case
    ref_r = TR[f'{case}/reward']
    assert np.all(np.abs(rw - ref_r) <= 1e-5 * np.abs(ref_r) + 1e-9), case
    assert np.array_equal(dn, TR[f'{case}/done']), case
    for row in TR[f'{case}/feats']:
        got, want = feats[int(row[0])], row[1:]
        assert np.all(np.abs(got - want) <= 1e-5 * np.abs(want) + 1e-7), (case, int(row[0]), int(np.argmax(np.abs(got - want))))
    ref_cost = TR[f'{case}/cost']
    assert clen == len(ref_cost) and close(cost[:clen], ref_cost), case
    assert np.abs(X - TR[f'{case}/final_X']).max() <= 1e-12, case
    assert fes == TR[f'{case}/fes']


@pytest.mark.parametrize('case', CASES)
def test_oracle_replays_reference_ddqn_episode(case):
    p, opt, dim, maxfes, nlog, nk, seed = _setup(case)
    cfg = oracle.make_cfg(3, NP, dim, maxfes, maxfes // nlog, nlog)
    o = oracle.DqOracle(p.desc(), opt, cfg)
    fd = oracle.DqTapeFeeder(seed, NP, dim, nk)
    feats = {-1: o.reset(fd.reset_tape())}
    acts = TR[f'{case}/actions']
    G = len(acts)
    gb, rw, dn = np.zeros(G), np.zeros(G), np.zeros(G, bool)
    for g in range(G):
        s, r, d = o.step(int(acts[g]), fd.step_tape())
        gb[g] = oracle.split_dq_state(o.state(), NP, dim, nlog)['scalars'][0]
        rw[g], dn[g], feats[g] = r, d, s
    st = oracle.split_dq_state(o.state(), NP, dim, nlog)
    _check(case, gb, rw, dn, feats, st['clog'], int(st['scalars'][3]), st['X'].reshape(NP, dim), st['scalars'][1])


@pytest.mark.gpu
def test_hip_ddqn_tape_replay_matches_reference():
    import torch
    from metabox_amd.suite import Batch, Suite
    from metabox_amd._abi import ALGO_DEDDQN
    for case in CASES:
        p, opt, dim, maxfes, nlog, nk, seed = _setup(case)
        s = Suite([p])
        b = Batch(s, ALGO_DEDDQN, [0], [0], NP, maxfes, maxfes // nlog, nlog)
        assert (b.state_dim, b.action_dim) == (99, 1)
        fd = oracle.DqTapeFeeder(seed, NP, dim, nk)
        b.set_tape(torch.from_numpy(fd.reset_tape()[None]).cuda())
        feats = {-1: b.reset()[0].cpu().numpy().copy()}
        acts = TR[f'{case}/actions']
        G = len(acts)
        gb, rw, dn = np.zeros(G), np.zeros(G), np.zeros(G, bool)
        want_feat = {int(r[0]) for r in TR[f'{case}/feats']}
        tape_dev = torch.empty(1, b.tape_stride, dtype=torch.float64, device='cuda')
        for g in range(G):
            tape_dev.copy_(torch.from_numpy(fd.step_tape()[None]))
            b.set_tape(tape_dev)
            st, r, d = b.step(torch.tensor([int(acts[g])], dtype=torch.int32, device='cuda'))
            rw[g] = r[0].item(); dn[g] = bool(d[0].item())
            gb[g] = b.read_public(0)[0]                 # MBX_SC_GBEST after this step: the whole trajectory is compared with the reference's
            if g in want_feat:
                feats[g] = st[0].cpu().numpy().copy()
        res = b.results()
        fin = oracle.split_dq_state(b.read_state(0), NP, dim, nlog)
        assert gb[-1] == fin['scalars'][0]
        _check(case, gb, rw, dn, feats, res['cost'][0].cpu().numpy(), int(res['cost_len'][0].item()), fin['X'].reshape(NP, dim), fin['scalars'][1])
        assert int(res['steps'][0].item()) == G
        b.close()


@pytest.mark.gpu
def test_hip_ddqn_philox_parity_with_oracle():
    import torch
    from metabox_amd.suite import Batch, Suite
    from metabox_amd._abi import ALGO_DEDDQN
    ps = problems('bbob-noisy', 10)
    ids = sorted(ps)
    s = Suite([ps[i] for i in ids])
    B, G = len(ids), 260                    # > 2 population sweeps: rings, OM_W eviction and prebest re-binding all fire
    rs = np.random.RandomState(9)
    actions = rs.randint(0, 4, size=(G, B)).astype(np.int32)
    seeds = np.arange(B, dtype=np.uint64) * 31 + 5
    b = Batch(s, ALGO_DEDDQN, np.arange(B), seeds, NP, 3000, 60, 50)
    st0 = b.reset().cpu().numpy().copy()
    hist = []
    for g in range(G):
        st, r, d = b.step(torch.from_numpy(actions[g]).cuda())
        hist.append((st.cpu().numpy().copy(), r.cpu().numpy().copy()))
    cfg = oracle.make_cfg(3, NP, 10, 3000, 60, 50)
    for k in range(B):
        p = s.problems[k]
        o = oracle.DqOracle(p.desc(), p.bias, cfg, seed=int(seeds[k]))
        f0 = o.reset()
        assert np.all(np.abs(f0 - st0[k]) <= 1e-5 * np.abs(f0) + 1e-7), ids[k]
        for g in range(G):
            f, rew, d = o.step(int(actions[g, k]))
            got = hist[g][0][k]
            assert np.all(np.abs(f - got) <= 1e-5 * np.abs(f) + 1e-7), (ids[k], g, int(np.argmax(np.abs(f - got))))
            assert abs(rew - hist[g][1][k]) <= 1e-5 * abs(rew) + 1e-9
        want = oracle.split_dq_state(o.state(), NP, 10, 50)
        got = oracle.split_dq_state(b.read_state(k), NP, 10, 50)
        assert np.abs(got['X'] - want['X']).max() <= 1e-12 and close(got['cost'], want['cost'])
        for key in ('ntot', 'nsucc'):
            assert np.array_equal(got[key], want[key]), (ids[k], key)
        assert close(got['extra'][:2], want['extra'][:2])                       # c_gworst, c_prebest (floating point)
        assert np.array_equal(got['extra'][2:9], want['extra'][2:9]), ids[k]    # pointer, gen, stagcount, |OM_W|, aliases (slots 9 / 10: the kernel's median cache)
    b.close()


@pytest.mark.gpu
def test_ddqn_batched_replay_training(tmp_path):
    """train_batch: epsilon-greedy lock-step interaction, device-resident replay, double-DQN updates with target refresh."""
    import copy
    import os
    import torch
    from metabox_amd.agent import DE_DDQN_Agent
    from metabox_amd.config import get_config
    from metabox_amd.environment import BatchedPBO_Env
    from metabox_amd.optimizer import DE_DDQN_Optimizer
    cfg = get_config(['--problem', 'bbob', '--dim', '10', '--device', 'cuda'])
    cfg.agent_save_dir = None
    agent = DE_DDQN_Agent(cfg).load_exported_weights(load('ddqn_policy.npz')).to('cuda')
    cfg.warm_up_size, cfg.update_target_steps = 512, 8
    cfg.maxFEs, cfg.log_interval = 400, 8
    tcfg = copy.deepcopy(cfg)
    tcfg.max_learning_step, tcfg.save_interval, tcfg.agent_save_dir = 10 ** 6, 10, str(tmp_path) + '/'
    agent.update_setting(tcfg)
    ps = [problems('bbob', 10)[f] for f in (1, 15)]
    env = BatchedPBO_Env(ps, DE_DDQN_Optimizer(cfg), np.arange(64) % 2, np.arange(64, dtype=np.uint64) + 1)
    before = [q.detach().clone() for q in agent.q_net.parameters()]
    torch.manual_seed(1)
    with torch.enable_grad():
        exceed, info = agent.train_batch(env, max_updates=25)
    assert not exceed and info['learn_steps'] == 25 and np.isfinite(info['return']) and info['normalizer'] >= info['gbest']
    assert any(not torch.equal(a, b) for a, b in zip(before, agent.q_net.parameters()))
    assert all(torch.isfinite(q).all() for q in agent.q_net.parameters())
    assert os.path.exists(str(tmp_path / 'checkpoint2.pkl')) and os.path.getsize(str(tmp_path / 'checkpoint2.pkl')) < 2_000_000
    env.close()


@pytest.mark.gpu
def test_ddqn_compile_time_geometry_kernel_equals_generic_kernel(monkeypatch):
    """BASELINE config 4 (protein docking, NP = 100, D = 12) runs k_dq_step with the geometry fixed at compile time;
    MBX_GENERIC_GEOMETRY=1 keeps the run-time-geometry kernel.  Every state word must be identical after 300 steps."""
    import torch
    from metabox_amd.suite import Batch, Suite
    from metabox_amd._abi import ALGO_DEDDQN
    from test_protein import protein
    ps = list(protein()[0].values())[:6]
    s = Suite(ps)
    B, G = 2 * len(ps), 300
    actions = torch.randint(0, 4, (G, B), generator=torch.Generator().manual_seed(4), dtype=torch.int32).cuda()
    seeds = np.arange(B, dtype=np.uint64) * 17 + 2
    states = []
    for generic in ('0', '1'):
        monkeypatch.setenv('MBX_GENERIC_GEOMETRY', generic)
        b = Batch(s, ALGO_DEDDQN, np.arange(B) % len(ps), seeds, NP, 1000, 200, 5)
        b.reset()
        for g in range(G):
            b.step(actions[g])
        states.append(np.stack([b.read_state(k) for k in range(B)]))
        b.close()
    assert np.array_equal(states[0], states[1], equal_nan=True)


@pytest.mark.gpu
def test_hip_ddqn_philox_parity_with_oracle_on_protein():
    """BASELINE config 4's geometry (protein docking, NP = 100, D = 12, maxFEs 1000 -> k_dq_step<100, 12>): oracle (Philox mode) and HIP
    kernel side by side on 4 protein problems x 2 seeds, every step's 99 features and reward, final state exact in its integer parts
    (src/optimizer/de_ddqn_optimizer.py:131-220 on src/problem/protein_docking.py:9-48)."""
    import torch
    from metabox_amd.suite import Batch, Suite
    from metabox_amd._abi import ALGO_DEDDQN
    from test_protein import protein
    ps = list(protein()[0].values())
    ps = [ps[0], ps[7], ps[100], ps[279]]
    s = Suite(ps)
    B, G, D = 2 * len(ps), 240, 12                  # > 2 population sweeps
    assert s.dim == D
    actions = np.random.RandomState(3).randint(0, 4, size=(G, B)).astype(np.int32)
    seeds = np.arange(B, dtype=np.uint64) * 13 + 3
    b = Batch(s, ALGO_DEDDQN, np.arange(B) % len(ps), seeds, NP, 1000, 200, 5)
    assert b.launch_info()['fixed_geometry'] != 0
    st0 = b.reset().cpu().numpy().copy()
    hist = []
    for g in range(G):
        st, r, d = b.step(torch.from_numpy(actions[g]).cuda())
        hist.append((st.cpu().numpy().copy(), r.cpu().numpy().copy(), b.read_public(0)[0]))
    cfg = oracle.make_cfg(3, NP, D, 1000, 200, 5)
    for k in range(B):
        p = ps[k % len(ps)]
        o = oracle.DqOracle(p.desc(), None, cfg, seed=int(seeds[k]))
        f0 = o.reset()
        assert np.all(np.abs(f0 - st0[k]) <= 1e-5 * np.abs(f0) + 1e-7), k
        for g in range(G):
            f, rew, d = o.step(int(actions[g, k]))
            got = hist[g][0][k]
            assert np.all(np.abs(f - got) <= 1e-5 * np.abs(f) + 1e-7), (k, g, int(np.argmax(np.abs(f - got))))
            assert abs(rew - hist[g][1][k]) <= 1e-5 * abs(rew) + 1e-9
            if k == 0:
                want_gb = oracle.split_dq_state(o.state(), NP, D, 5)['scalars'][0]
                assert abs(hist[g][2] - want_gb) <= 1e-5 * abs(want_gb) + 1e-9, g
        want = oracle.split_dq_state(o.state(), NP, D, 5)
        got = oracle.split_dq_state(b.read_state(k), NP, D, 5)
        assert np.abs(got['X'] - want['X']).max() <= 1e-12 and np.all(np.abs(got['cost'] - want['cost']) <= 1e-5 * np.abs(want['cost']) + 1e-9)
        for key in ('ntot', 'nsucc'):
            assert np.array_equal(got[key], want[key]), (k, key)
        assert np.array_equal(got['extra'][2:9], want['extra'][2:9]), k
    b.close()


@pytest.mark.gpu
def test_ddqn_protein_batch_properties():
    """BASELINE config 4, one GPU's share: 2 240 instances = 35 protein problems x 64 runs, k_dq_step<100, 12>.  Size-independent
    properties: bit-identical re-run, the odd half of the batch reproduces its rows (results do not depend on batch position / shard),
    fes accounting (NP + one evaluation per step), monotone gbest, done after maxFEs - NP steps."""
    import torch
    from metabox_amd.suite import Batch, Suite
    from metabox_amd._abi import ALGO_DEDDQN
    from test_protein import protein
    ps = list(protein()[0].values())[:35]
    s = Suite(ps)
    B, G = 35 * 64, 150
    pidx = np.arange(B) // 64
    seeds = (np.arange(B, dtype=np.uint64) % 64) + 1
    actions = torch.randint(0, 4, (G, B), generator=torch.Generator().manual_seed(11), dtype=torch.int32).cuda()

    def run(sel, maxfes=1000, steps=G):
        b = Batch(s, ALGO_DEDDQN, pidx[sel], seeds[sel], NP, maxfes, maxfes // 5, 5)
        assert b.launch_info()['fixed_geometry'] != 0
        b.reset()
        sel_dev = torch.from_numpy(sel).cuda()
        gbs = []
        for g in range(steps):
            st, r, d = b.step(actions[g].index_select(0, sel_dev).contiguous())
            if g % 50 == 49:
                gbs.append(b.results()['cost'][:, 0].clone())
        r = {k: v.cpu().numpy() for k, v in b.results().items()}
        r['state'] = st.cpu().numpy().copy()
        r['done'] = d.cpu().numpy().copy()
        x = np.stack([b.read_state(i) for i in (0, len(sel) // 2, len(sel) - 1)])
        b.close()
        return r, x
    full, xf = run(np.arange(B))
    again, xa = run(np.arange(B))
    for k in full:
        assert np.array_equal(full[k], again[k], equal_nan=True), k
    assert np.array_equal(xf, xa, equal_nan=True)
    half = np.arange(B)[1::2]
    part, _ = run(half)
    for k in full:
        assert np.array_equal(full[k][half], part[k], equal_nan=True), k
    assert np.all(full['steps'] == G) and np.all(full['fes'] == NP + G) and not full['done'].any()
    assert np.all(np.isfinite(full['state'])) and full['state'].shape == (B, 99)
    # the 64 runs of one problem differ (seeds), the same (problem, seed) pair does not depend on where it sits
    assert len(np.unique(full['cost'][:64, 0])) > 32
    # a whole short episode: every instance terminates exactly at maxFEs (protein has no optimum: de_ddqn_optimizer.py:205-212)
    short, _ = run(np.arange(0, B, 7), maxfes=NP + 40, steps=45)
    assert np.all(short['steps'] == 40) and np.all(short['fes'] == NP + 40) and short['done'].all()


@pytest.mark.gpu
def test_ddqn_greedy_action_agrees_between_the_hip_and_torch_routes():
    """rollout_batch's default Q-network route (mbx_ddqn_qnet: float32 fma chains on the matrix cores) and the PyTorch module sum in different orders,
    so near-tied Q values can give a different argmax and from there a different trajectory.  Over a lock-step rollout in which BOTH routes are evaluated on the
    same states (the hip action drives the batch), the actions must agree except where the two largest Q values are within 1e-4 of each other, and such states
    must be rare; --ddqn_policy torch keeps the earlier route selectable."""
    import torch
    from metabox_amd.agent import DE_DDQN_Agent
    from metabox_amd.config import get_config
    from metabox_amd.environment import BatchedPBO_Env
    from metabox_amd.optimizer import DE_DDQN_Optimizer
    from test_protein import protein
    cfg = get_config(['--problem', 'protein', '--device', 'cuda', '--ddqn_policy', 'torch'])
    assert cfg.ddqn_policy == 'torch'
    cfg = get_config(['--problem', 'protein', '--device', 'cuda'])
    assert cfg.ddqn_policy == 'hip'
    cfg.agent_save_dir = None
    torch.manual_seed(0)
    agent = DE_DDQN_Agent(cfg).to('cuda')
    ps = list(protein()[0].values())[:8]
    B = 8 * 16
    env = BatchedPBO_Env(ps, DE_DDQN_Optimizer(cfg), np.arange(B) // 16, np.arange(B, dtype=np.uint64) + 1)
    state = env.reset()
    packed = agent.packed_weights()
    differ, close_calls, n = 0, 0, 0
    with torch.no_grad():
        for _ in range(120):
            a_hip = env.batch.ddqn_qnet(packed).clone()
            q = agent.q_net(state.to(torch.float32))
            a_torch = torch.argmax(q, dim=1).to(torch.int32)
            top2 = torch.topk(q, 2, dim=1).values
            tight = (top2[:, 0] - top2[:, 1]).abs() <= 1e-4 * top2[:, 0].abs().clamp_min(1.0)
            bad = (a_hip != a_torch)
            assert not bool((bad & ~tight).any()), 'the routes disagree on a state whose two best Q values are NOT close'
            differ += int(bad.sum()); close_calls += int(tight.sum()); n += B
            state, _, _ = env.step(a_hip)
    assert differ <= close_calls and differ <= 0.01 * n, (differ, close_calls, n)
    env.close()
