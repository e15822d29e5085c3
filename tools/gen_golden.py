#!/usr/bin/env python
"""Generate the golden fixtures under tests/golden/ by running the UPSTREAM REFERENCE.

Runs only in the build container (needs /root/reference); the fixtures it writes are data
(inputs + expected outputs), never reference source.  Re-run: `python tools/gen_golden.py [section ...]`
with sections in {instances, kat, noise, policy, rlepso, lde, protein, ddqn, harness, stats}.

What is recorded
  instances : per (suite, dim) the problem names, biases, optima, a sha256 over every constructor-made
              array, and the value of the next np.random.rand() after the dataset build (stream position).
  kat       : f(x) for 14 points per function (0, linspace, 8 in-bounds, 4 out-of-bounds), D = 10, 30, 40.
  noise     : problem.eval(x) of the 30 noisy functions under np.random.seed(s): expected noisy values.
  policy    : weights of the shipped RLEPSO actor/critic + (state -> mu, sigma, value) pairs.
  rlepso    : whole RLEPSO episodes (shipped actor, np.random.seed(s); torch.manual_seed(s)): the
              float32 action of every generation and gbest / fes / reward / done after every update(),
              final optimizer.cost, fes.  The numpy draws are NOT stored: the test regenerates them from
              the seed with numpy's legacy MT19937 stream in the reference's draw order.
"""
import hashlib
import json
import os
import pickle
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_import  # noqa: E402

ref_import.install()
import torch  # noqa: E402

torch.set_num_threads(1)
OUT = os.path.join(os.path.dirname(HERE), 'tests', 'golden')
os.makedirs(OUT, exist_ok=True)

ARRAY_ATTRS = ('shift', 'rotate', 'scales', 'linearTF', 'Q_rotate', 'y', 'C', 'w', 'aK', 'bK', 'f0', 'mu0')


def digest(problem):
    h = hashlib.sha256()
    for a in ARRAY_ATTRS:
        if hasattr(problem, a):
            h.update(a.encode())
            h.update(np.ascontiguousarray(np.asarray(getattr(problem, a), dtype=np.float64)).tobytes())
    return h.hexdigest()


def kat_points(dim):
    rs = np.random.RandomState(20250202 + dim)
    pts = [np.zeros(dim), np.linspace(-4, 4, dim)]
    pts += list(rs.uniform(-5, 5, size=(8, dim)))
    pts += list(rs.uniform(-8, 8, size=(4, dim)))
    return np.stack(pts)


def all_problems(suite, dim, difficulty='easy'):
    from problem.bbob import BBOB_Dataset
    tr, te = BBOB_Dataset.get_datasets(suite, dim, 5.0, difficulty=difficulty)
    nxt = float(np.random.rand())
    return tr.data, te.data, nxt


def fid_of(problem):
    return int(type(problem).__name__[1:])


def gen_instances():
    out = {}
    for suite in ('bbob', 'bbob-noisy'):
        for dim in (10, 30, 40):
            tr, te, nxt = all_problems(suite, dim)
            rec = {'next_rand': nxt, 'train': [fid_of(p) for p in tr], 'test': [fid_of(p) for p in te], 'problems': {}}
            for p in tr + te:
                rec['problems'][str(fid_of(p))] = {
                    'name': str(p), 'bias': float(p.bias), 'optimum': float(p.optimum), 'sha256': digest(p),
                    'shift0': float(p.shift[0]), 'rotate00': float(p.rotate[0, 0])}
            out[f'{suite}/{dim}'] = rec
    with open(os.path.join(OUT, 'bbob_instances.json'), 'w') as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print('instances:', len(out), 'suite/dim entries')


def gen_kat():
    data = {}
    for suite in ('bbob', 'bbob-noisy'):
        for dim in (10, 30, 40):
            tr, te, _ = all_problems(suite, dim)
            X = kat_points(dim)
            data[f'x/{dim}'] = X
            for p in tr + te:
                # F*.func is the noise-free objective (bias included) for every id
                data[f'f/{suite}/{dim}/{fid_of(p)}'] = np.asarray(p.func(X.copy()), dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, 'bbob_kat.npz'), **data)
    print('kat:', len(data), 'arrays')


def gen_noise():
    data = {}
    for dim in (10, 30):
        tr, te, _ = all_problems('bbob-noisy', dim)
        rs = np.random.RandomState(777 + dim)
        X = rs.uniform(-5, 5, size=(64, dim))
        data[f'x/{dim}'] = X
        for p in tr + te:
            p.reset()
            for seed in (0, 1):
                np.random.seed(seed)
                data[f'f/{dim}/{fid_of(p)}/{seed}'] = np.asarray(p.eval(X.copy()), dtype=np.float64)
            # a near-optimum row exercises the `ftrue_unbiased >= 1e-8` switch
            xo = np.stack([p.shift, p.shift + 1e-7])
            np.random.seed(5)
            data[f'fopt/{dim}/{fid_of(p)}'] = np.asarray(p.eval(xo.copy()), dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, 'bbob_noise.npz'), **data)
    print('noise:', len(data), 'arrays')


def load_shipped(path):
    with open(path, 'rb') as f:
        return pickle.load(f)


def gen_policy():
    agent = load_shipped(os.path.join(ref_import.REF_SRC, 'agent_model/test/bbob_easy/RLEPSO_Agent.pkl'))
    actor = agent._RLEPSO_Agent__actor
    critic = agent._RLEPSO_Agent__critic
    data = {}
    for k, v in actor.state_dict().items():
        data['actor/' + k] = v.detach().cpu().numpy()
    for k, v in critic.state_dict().items():
        data['critic/' + k] = v.detach().cpu().numpy()
    s = torch.linspace(0, 1.05, 64, dtype=torch.float32).reshape(-1, 1)
    with torch.no_grad():
        mu = (torch.tanh(actor._Actor__mu_net(s)) + 1.) / 2.
        sg = (torch.tanh(actor._Actor__sigma_net(s)) + 1.) / 2. * (0.7 - 0.01) + 0.01
        val = critic._Critic__value_head(s)
    data['io/state'] = s.numpy()
    data['io/mu'] = mu.numpy()
    data['io/sigma'] = sg.numpy()
    data['io/value'] = val.numpy()
    data['meta/max_sigma'] = np.float64(actor._Actor__max_sigma)
    data['meta/min_sigma'] = np.float64(actor._Actor__min_sigma)
    np.savez_compressed(os.path.join(OUT, 'rlepso_policy.npz'), **data)
    # weights only, shipped with the package (exported arrays of the reference's trained bbob_easy checkpoint)
    pkg = os.path.join(os.path.dirname(HERE), 'metabox_amd', 'agent_model')
    os.makedirs(pkg, exist_ok=True)
    np.savez_compressed(os.path.join(pkg, 'rlepso_bbob_easy.npz'),
                        **{k: v for k, v in data.items() if k.startswith(('actor/', 'critic/'))})
    print('policy:', {k: v.shape for k, v in data.items() if k.startswith('actor')})


def run_rlepso_episode(problem, seed, agent, config, action_mode):
    """One reference rollout, recording per-generation actions and outcomes."""
    from optimizer import RLEPSO_Optimizer
    from environment import PBO_Env
    import copy
    opt = RLEPSO_Optimizer(copy.deepcopy(config))
    env = PBO_Env(problem, opt)
    actor = agent._RLEPSO_Agent__actor
    np.random.seed(seed)
    torch.manual_seed(seed)
    ars = np.random.RandomState(10_000 + seed)      # only for action_mode == 'uniform'
    state = env.reset()
    gb0 = float(opt._RLEPSO_Optimizer__particles['gbest_val'])
    actions, gbest, fes, reward, done_l = [], [], [], [], []
    done = False
    while not done:
        if action_mode == 'actor':
            with torch.no_grad():
                action = actor(torch.FloatTensor(state))[0].cpu().numpy()
        else:
            action = ars.uniform(0, 1, size=35).astype(np.float32)
        state, r, done = env.step(action)
        actions.append(action.astype(np.float32))
        gbest.append(float(opt._RLEPSO_Optimizer__particles['gbest_val']))
        fes.append(float(opt.fes))
        reward.append(float(r))
        done_l.append(bool(done))
    part = opt._RLEPSO_Optimizer__particles
    return dict(actions=np.stack(actions), gbest=np.array(gbest), fes=np.array(fes), reward=np.array(reward),
                done=np.array(done_l), cost=np.array(opt.cost, dtype=np.float64), gbest0=np.float64(gb0),
                final_pos=np.array(part['current_position']), final_pbest=np.array(part['pbest']),
                final_pni=np.array(opt._RLEPSO_Optimizer__per_no_improve))


def gen_rlepso():
    scratch = tempfile.mkdtemp()
    data = {}
    cases = []
    agent = load_shipped(os.path.join(ref_import.REF_SRC, 'agent_model/test/bbob_easy/RLEPSO_Agent.pkl'))
    for suite, fids, seeds, mode in (
            ('bbob', (1, 5, 6, 3, 7, 16, 21, 23, 24), (0, 1), 'actor'),
            ('bbob', (2, 4, 8, 9, 10, 11, 12, 13, 14, 15, 17, 18, 19, 20, 22), (2,), 'actor'),
            ('bbob', (1, 9, 20), (3,), 'uniform'),              # random actions: __reinit fires constantly
            ('bbob-noisy', (101, 102, 103, 113, 117, 121, 124, 126, 130), (0,), 'actor'),
            ('bbob-noisy', (108, 115), (4,), 'uniform')):
        config = ref_import.ref_config(['--problem', suite, '--dim', '10'], scratch)
        tr, te, _ = all_problems(suite, 10)
        byid = {fid_of(p): p for p in tr + te}
        for fid in fids:
            for seed in seeds:
                p = byid[fid]
                p.reset()
                rec = run_rlepso_episode(p, seed, agent, config, mode)
                key = f'{suite}/{fid}/{seed}/{mode}'
                cases.append(key)
                for k, v in rec.items():
                    data[f'{key}/{k}'] = v
                n_re = int(np.sum(np.diff(np.concatenate([[100.], rec['fes']])) != 100))
                print(f'{key}: gens={len(rec["gbest"])} fes={rec["fes"][-1]:.0f} final={rec["gbest"][-1]:.6g} reinit_steps={n_re}')
    data['cases'] = np.array(cases)
    np.savez_compressed(os.path.join(OUT, 'rlepso_traces.npz'), **data)
    print('rlepso:', len(cases), 'episodes')


def _final_r(calls, NP):
    """Re-apply the rejection loop of LDE_Optimizer.__add_random (lde_optimizer.py:109-121) to the recorded
    torch.randint outputs of one update() call -> the indices r[NP,2] that were finally used."""
    r = calls[0].clone()
    k = 1
    pop_index = torch.arange(NP)
    for col in range(2):
        while True:
            rep = [torch.eq(r[:, :, col], r[:, :, i]) for i in range(col)]
            rep.append(torch.eq(r[:, :, col], pop_index))
            idx = torch.nonzero(torch.any(torch.stack(rep), dim=0))
            if idx.size(0) != 0:
                r[idx[:, 0], idx[:, 1], col] = calls[k]
                k += 1
            else:
                break
    assert k == len(calls)
    return r[0].numpy().astype(np.uint8)


def run_lde_episode(problem, seed, agent, config, action_mode, NP=50, keep_actions=True):
    from environment import PBO_Env
    import copy
    opt = lde_class_with_np(NP)(copy.deepcopy(config))
    env = PBO_Env(problem, opt)
    net = agent._LDE_Agent__net if agent is not None else None
    np.random.seed(seed)
    torch.manual_seed(seed)
    ars = np.random.RandomState(20_000 + seed)
    state = env.reset()
    state0 = np.array(state[0])
    h = torch.zeros(1, 1, 50)
    c = torch.zeros(1, 1, 50)
    real_randint = torch.randint
    rec = dict(actions=[], r=[], gbest=[], fes=[], reward=[], done=[], states=[])
    done = False
    g = 0
    first_tie = -1
    while not done:
        if action_mode == 'actor':
            with torch.no_grad():
                a, h, c = net.sampler(torch.FloatTensor(state[None, :]), h, c)
            action = np.squeeze(a.reshape(1, 1, -1).cpu().numpy(), axis=0)
        else:
            action = ars.uniform(0, 1, size=(1, 2 * NP)).astype(np.float32)
        calls = []
        if first_tie < 0 and len(np.unique(opt._LDE_Optimizer__fit[0])) < NP:
            first_tie = g          # two individuals with exactly the same fitness enter update() number g: np.argsort's order of them is unspecified

        def spy(*a_, **k_):
            out = real_randint(*a_, **k_)
            calls.append(out.clone())
            return out
        torch.randint = spy
        try:
            state, r, done = env.step(action)
        finally:
            torch.randint = real_randint
        rec['actions'].append(action[0].astype(np.float32))
        rec['r'].append(_final_r(calls, NP))
        rec['gbest'].append(float(opt.gbest_cost))
        rec['fes'].append(float(opt.fes))
        rec['reward'].append(float(np.asarray(r).reshape(-1)[0]))
        rec['done'].append(bool(done))
        if g % 40 == 0 or done:
            rec['states'].append(np.concatenate([[g], np.asarray(state[0], dtype=np.float64)]))
        g += 1
    out = {k: np.stack(v) if k in ('actions', 'r', 'states') else np.array(v) for k, v in rec.items()}
    if not keep_actions:
        assert action_mode == 'uniform'           # regenerated by the tests: RandomState(20_000 + seed).uniform(0, 1, (1, 2 NP)).astype(float32) per generation
        del out['actions']
    out['cost'] = np.array(opt.cost, dtype=np.float64)
    out['first_tie_gen'] = np.int32(first_tie)
    out['state0'] = state0
    out['final_fit'] = np.array(opt._LDE_Optimizer__fit[0])
    out['final_pop'] = np.array(opt._LDE_Optimizer__pop[0])
    return out


def gen_lde():
    scratch = tempfile.mkdtemp()
    agent = load_shipped(os.path.join(ref_import.REF_SRC, 'agent_model/test/bbob_easy/LDE_Agent.pkl'))
    net = agent._LDE_Agent__net
    pol = {}
    for k, v in net.state_dict().items():
        pol['net/' + k] = v.detach().cpu().numpy()
    torch.manual_seed(0)
    x = torch.rand(1, 8, 60)
    h = torch.randn(1, 8, 50) * 0.3
    c = torch.randn(1, 8, 50) * 0.3
    with torch.no_grad():
        mu, sg, h2, c2 = net.forward(x, h, c)
    pol.update({'io/x': x.numpy(), 'io/h': h.numpy(), 'io/c': c.numpy(), 'io/mu': mu.numpy(), 'io/sigma': sg.numpy(),
                'io/h_out': h2.numpy(), 'io/c_out': c2.numpy()})
    np.savez_compressed(os.path.join(OUT, 'lde_policy.npz'), **pol)
    pkg = os.path.join(os.path.dirname(HERE), 'metabox_amd', 'agent_model')
    np.savez_compressed(os.path.join(pkg, 'lde_bbob_easy.npz'), **{k: v for k, v in pol.items() if k.startswith('net/')})
    data, cases = {}, []
    for suite, dim, fids, seeds, mode in (
            ('bbob', 10, (1, 5, 3, 16, 21), (0,), 'actor'),
            ('bbob', 10, (8,), (1,), 'uniform'),
            ('bbob-noisy', 10, (101, 117, 124), (0,), 'actor'),
            ('bbob-noisy', 30, (102,), (2,), 'actor')):
        config = ref_import.ref_config(['--problem', suite, '--dim', str(dim)], scratch)
        tr, te, _ = all_problems(suite, dim)
        byid = {fid_of(p): p for p in tr + te}
        for fid in fids:
            for seed in seeds:
                p = byid[fid]
                p.reset()
                rec = run_lde_episode(p, seed, agent, config, mode)
                key = f'{suite}/{dim}/{fid}/{seed}/{mode}'
                cases.append(key)
                for k, v in rec.items():
                    data[f'{key}/{k}'] = v
                print(f'{key}: gens={len(rec["gbest"])} fes={rec["fes"][-1]:.0f} final={rec["gbest"][-1]:.6g}')
    data['cases'] = np.array(cases)
    np.savez_compressed(os.path.join(OUT, 'lde_traces.npz'), **data)
    print('lde:', len(cases), 'episodes')


def protein_problems():
    from problem.protein_docking import Protein_Docking_Dataset
    tr, te = Protein_Docking_Dataset.get_datasets('protein', difficulty='easy')
    return {str(p): p for p in tr.data + te.data}, [str(p) for p in tr.data], [str(p) for p in te.data]


def gen_protein():
    byid, tr, te = protein_problems()
    nxt = float(np.random.rand())
    rs = np.random.RandomState(4242)
    X = np.concatenate([np.zeros((1, 12)), rs.uniform(-1.5, 1.5, size=(9, 12)), rs.uniform(-0.05, 0.05, size=(2, 12))])
    data = {'x': X, 'train_ids': np.array(tr), 'test_ids': np.array(te), 'next_rand': np.float64(nxt)}
    for pid in ('1AVX_1', '1ATN_7', '2HRK_10', '7CEI_3', tr[0], te[-1]):
        byid[pid].reset()
        data[f'f/{pid}'] = np.asarray(byid[pid].eval(X.copy()), dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, 'protein_kat.npz'), **data)
    print('protein:', {k: v.shape for k, v in data.items() if k.startswith('f/')})
    # short RLEPSO / LDE episodes on protein problems (maxFEs = 1000, optimum None => no early stop)
    scratch = tempfile.mkdtemp()
    out, cases = {}, []
    agent = load_shipped(os.path.join(ref_import.REF_SRC, 'agent_model/test/bbob_easy/RLEPSO_Agent.pkl'))
    config = ref_import.ref_config(['--problem', 'protein'], scratch)
    for pid, seed in (('1AVX_1', 0), ('1ATN_7', 1), ('2HRK_10', 2)):
        p = byid[pid]
        p.reset()
        rec = run_rlepso_episode(p, seed, agent, config, 'actor')
        key = f'rlepso/{pid}/{seed}'
        cases.append(key)
        for k, v in rec.items():
            out[f'{key}/{k}'] = v
        print(key, len(rec['gbest']), rec['fes'][-1], rec['gbest'][-1])
    lde = load_shipped(os.path.join(ref_import.REF_SRC, 'agent_model/test/bbob_easy/LDE_Agent.pkl'))
    config = ref_import.ref_config(['--problem', 'protein'], scratch)
    for pid, seed in (('1AVX_1', 3), ('7CEI_3', 4)):
        p = byid[pid]
        p.reset()
        rec = run_lde_episode(p, seed, lde, config, 'actor')
        key = f'lde/{pid}/{seed}'
        cases.append(key)
        for k, v in rec.items():
            out[f'{key}/{k}'] = v
        print(key, len(rec['gbest']), rec['fes'][-1], rec['gbest'][-1])
    out['cases'] = np.array(cases)
    np.savez_compressed(os.path.join(OUT, 'protein_traces.npz'), **out)


def run_ddqn_episode(problem, seed, agent, config):
    from optimizer import DE_DDQN_Optimizer
    from environment import PBO_Env
    import copy
    opt = DE_DDQN_Optimizer(copy.deepcopy(config))
    env = PBO_Env(problem, opt)
    net = agent._DE_DDQN_Agent__pred_func
    np.random.seed(seed)
    ars = np.random.RandomState(30_000 + seed)
    state = env.reset()
    rec = dict(actions=[], gbest=[], reward=[], done=[], feats=[np.concatenate([[-1], state])])
    done, g = False, 0
    while not done:
        with torch.no_grad():
            a = int(torch.argmax(net(torch.Tensor(state))))
        if ars.rand() < 0.35:
            a = int(ars.randint(0, 4))
        state, r, done = env.step(a)
        rec['actions'].append(a)
        rec['gbest'].append(float(opt._DE_DDQN_Optimizer__c_gbest))
        rec['reward'].append(float(r))
        rec['done'].append(bool(done))
        if g % 97 == 0 or done or g < 3 or 99 <= g <= 102 or 199 <= g <= 201:
            rec['feats'].append(np.concatenate([[g], state]))
        g += 1
    out = {'actions': np.array(rec['actions'], dtype=np.uint8), 'gbest': np.array(rec['gbest']), 'reward': np.array(rec['reward']),
           'done': np.array(rec['done']), 'feats': np.stack(rec['feats']), 'cost': np.array(opt.cost, dtype=np.float64),
           'fes': np.float64(opt.fes), 'final_cost': np.array(opt._DE_DDQN_Optimizer__cost),
           'final_X': np.array(opt._DE_DDQN_Optimizer__X)}
    return out


def gen_ddqn():
    scratch = tempfile.mkdtemp()
    data, cases = {}, []
    config = ref_import.ref_config(['--problem', 'protein'], scratch)
    torch.manual_seed(123)
    from agent import DE_DDQN_Agent
    agent = DE_DDQN_Agent(copy_config(config))
    pol = {'net/' + k: v.detach().cpu().numpy() for k, v in agent._DE_DDQN_Agent__pred_func.state_dict().items()}
    x = torch.rand(16, 99)
    with torch.no_grad():
        pol['io/x'] = x.numpy()
        pol['io/q'] = agent._DE_DDQN_Agent__pred_func(x).numpy()
    np.savez_compressed(os.path.join(OUT, 'ddqn_policy.npz'), **pol)
    byid, _, _ = protein_problems()
    for pid, seed in (('1AVX_1', 0), ('1ATN_7', 1), ('2HRK_10', 2)):
        p = byid[pid]
        p.reset()
        rec = run_ddqn_episode(p, seed, agent, config)
        key = f'protein/12/{pid}/{seed}'
        cases.append(key)
        for k, v in rec.items():
            data[f'{key}/{k}'] = v
        print(key, len(rec['gbest']), rec['gbest'][-1], np.bincount(rec['actions'], minlength=4))
    for suite, fids, seed in (('bbob', (1, 15, 21), 3), ('bbob-noisy', (103, 117), 4)):
        config = ref_import.ref_config(['--problem', suite, '--dim', '10'], scratch)
        config.maxFEs = 3000                      # shortened budget keeps the fixture small (log_interval follows)
        config.log_interval = config.maxFEs // config.n_logpoint
        tr, te, _ = all_problems(suite, 10)
        byfid = {fid_of(p): p for p in tr + te}
        for fid in fids:
            p = byfid[fid]
            p.reset()
            rec = run_ddqn_episode(p, seed, agent, config)
            key = f'{suite}/10/{fid}/{seed}'
            cases.append(key)
            for k, v in rec.items():
                data[f'{key}/{k}'] = v
            print(key, len(rec['gbest']), rec['gbest'][-1], np.bincount(rec['actions'], minlength=4))
    data['cases'] = np.array(cases)
    np.savez_compressed(os.path.join(OUT, 'ddqn_traces.npz'), **data)


def run_rlpso_episode(problem, seed, agent, config, max_steps, action_mode):
    """One reference RL-PSO rollout (one particle per step), recording the float32 actions and the outcomes."""
    from optimizer import RL_PSO_Optimizer
    from environment import PBO_Env
    import copy
    opt = RL_PSO_Optimizer(copy.deepcopy(config))
    env = PBO_Env(problem, opt)
    nets = agent._RL_PSO_Agent__nets
    np.random.seed(seed)
    torch.manual_seed(seed)
    ars = np.random.RandomState(40_000 + seed)
    state = env.reset()
    rec = dict(actions=[], gbest=[], reward=[], done=[], states=[np.concatenate([[-1], state])])
    done, g = False, 0
    while not done and g < max_steps:
        if action_mode == 'policy':
            with torch.no_grad():
                a, _ = nets(torch.FloatTensor(state))
            a = a.cpu().numpy()                         # shape (1,), float32 -- what rollout_episode passes (rl_pso_agent.py:118-124)
        else:
            a = np.array([ars.rand() * 1.4 - 0.2], dtype=np.float32)
        state, r, done = env.step(a)
        rec['actions'].append(np.float32(a[0]))
        rec['gbest'].append(float(opt._RL_PSO_Optimizer__particles['gbest_val']))
        rec['reward'].append(float(r))
        rec['done'].append(bool(done))
        if g % 211 == 0 or done or g < 3 or 99 <= g <= 101:
            rec['states'].append(np.concatenate([[g], state]))
        g += 1
    pt = opt._RL_PSO_Optimizer__particles
    return {'actions': np.array(rec['actions'], dtype=np.float32), 'gbest': np.array(rec['gbest']), 'reward': np.array(rec['reward']),
            'done': np.array(rec['done']), 'states': np.stack(rec['states']), 'cost': np.array(opt.cost, dtype=np.float64),
            'fes': np.float64(opt.fes), 'final_pos': np.array(pt['current_position']), 'final_vel': np.array(pt['velocity']),
            'final_pbest': np.array(pt['pbest']), 'final_ccost': np.array(pt['c_cost']), 'w': np.float64(opt._RL_PSO_Optimizer__w)}


def gen_rlpso():
    """RL-PSO: the shipped bbob_easy policy (weights + I/O pairs) and seeded reference episodes."""
    scratch = tempfile.mkdtemp()
    agent = load_shipped(os.path.join(ref_import.REF_SRC, 'agent_model/test/bbob_easy/RL_PSO_Agent.pkl'))
    nets = agent._RL_PSO_Agent__nets
    pol = {'nets/' + k: v.detach().cpu().numpy() for k, v in nets.state_dict().items()}
    x = torch.rand(32, 20) * 10 - 5
    with torch.no_grad():
        _, _, mu, sigma = nets(x, require_musigma=True)
    pol['io/x'] = x.numpy(); pol['io/mu'] = mu.numpy(); pol['io/sigma'] = sigma.numpy()
    pol['meta/max_sigma'] = np.float64(nets._PolicyNetwork__max_sigma); pol['meta/min_sigma'] = np.float64(nets._PolicyNetwork__min_sigma)
    np.savez_compressed(os.path.join(OUT, 'rlpso_policy.npz'), **pol)
    pkg = os.path.join(os.path.dirname(HERE), 'metabox_amd', 'agent_model')
    np.savez_compressed(os.path.join(pkg, 'rlpso_bbob_easy.npz'), **{k: v for k, v in pol.items() if k.startswith('nets/')})
    print('rlpso policy:', {k: v.shape for k, v in pol.items() if k.startswith('nets')})
    data, cases = {}, []
    for suite, fids, seed, mode in (('bbob', (1, 8, 15, 21), 5, 'policy'), ('bbob', (3, 16), 6, 'uniform'), ('bbob-noisy', (103, 116, 128), 7, 'policy')):
        config = ref_import.ref_config(['--problem', suite, '--dim', '10'], scratch)
        config.maxFEs = 2500                       # shortened budget keeps the fixture small (log_interval follows)
        config.log_interval = config.maxFEs // config.n_logpoint
        tr, te, _ = all_problems(suite, 10)
        byfid = {fid_of(p): p for p in tr + te}
        for fid in fids:
            p = byfid[fid]
            p.reset()
            rec = run_rlpso_episode(p, seed, agent, config, 10 ** 9, mode)
            key = f'{suite}/10/{fid}/{seed}'
            cases.append(key)
            for k, v in rec.items():
                data[f'{key}/{k}'] = v
            print(key, mode, len(rec['gbest']), rec['gbest'][-1], rec['fes'], rec['w'])
    byid, _, _ = protein_problems()
    config = ref_import.ref_config(['--problem', 'protein'], scratch)
    agent12 = None
    for pid, seed in (('1AVX_1', 8),):
        p = byid[pid]
        p.reset()
        rec = run_rlpso_episode(p, seed, agent, config, 10 ** 9, 'uniform')      # the shipped net is 20-dimensional: uniform actions
        key = f'protein/12/{pid}/{seed}'
        cases.append(key)
        for k, v in rec.items():
            data[f'{key}/{k}'] = v
        print(key, len(rec['gbest']), rec['gbest'][-1], rec['fes'])
    data['cases'] = np.array(cases)
    np.savez_compressed(os.path.join(OUT, 'rlpso_traces.npz'), **data)


def run_gleet_episode(problem, seed, config):
    """One reference GLEET rollout driven by seeded float32 actions in [0, 1] (the agent's squashed output range)."""
    from optimizer import GLEET_Optimizer
    from environment import PBO_Env
    import copy
    opt = GLEET_Optimizer(copy.deepcopy(config))
    env = PBO_Env(problem, opt)
    np.random.seed(seed)
    ars = np.random.RandomState(50_000 + seed)
    state = env.reset()
    rec = dict(gbest=[], reward=[], done=[], states=[(-1, state)], sub=[])
    done, g = False, 0
    while not done:
        a = ars.rand(opt.ps).astype(np.float32)
        state, r, done = env.step(a)
        rec['gbest'].append(float(opt.particles['gbest_val']))
        rec['reward'].append(float(r))
        rec['done'].append(bool(done))
        if done:
            rec['states'].append((g, state))
        elif g % 37 == 0:
            rec['sub'].append((g, state[::11]))              # particles 0, 11, ..., 99: keeps the fixture small
        g += 1
    pt = opt.particles
    # the actions are NOT stored: the tests regenerate them from RandomState(50000 + seed) exactly as above
    return {'gbest': np.array(rec['gbest']), 'reward': np.array(rec['reward']),
            'done': np.array(rec['done']), 'state_gen': np.array([k for k, _ in rec['states']]),
            'states': np.stack([s for _, s in rec['states']]), 'sub_gen': np.array([k for k, _ in rec['sub']]),
            'sub_states': np.stack([s for _, s in rec['sub']]), 'cost': np.array(opt.cost, dtype=np.float64), 'fes': np.float64(opt.fes),
            'final_pos': np.array(pt['current_position']), 'final_pbest': np.array(pt['pbest']),
            'final_pni': np.array(opt.per_no_improve), 'w': np.float64(opt.w), 'no_improve': np.float64(opt.no_improve)}


def gen_gleet():
    """GLEET optimizer: seeded reference episodes with recorded actions."""
    scratch = tempfile.mkdtemp()
    data, cases = {}, []
    for suite, fids, seed in (('bbob', (1, 6, 15, 21), 11), ('bbob-noisy', (104, 117, 130), 12)):
        config = ref_import.ref_config(['--problem', suite, '--dim', '10'], scratch)
        tr, te, _ = all_problems(suite, 10)
        byfid = {fid_of(p): p for p in tr + te}
        for fid in fids:
            p = byfid[fid]
            p.reset()
            rec = run_gleet_episode(p, seed, config)
            key = f'{suite}/10/{fid}/{seed}'
            cases.append(key)
            for k, v in rec.items():
                data[f'{key}/{k}'] = v
            print(key, len(rec['gbest']), rec['gbest'][-1], rec['fes'], rec['w'], rec['no_improve'])
    config = ref_import.ref_config(['--problem', 'bbob', '--dim', '30'], scratch)
    tr, te, _ = all_problems('bbob', 30)
    byfid = {fid_of(p): p for p in tr + te}
    byfid[10].reset()
    rec = run_gleet_episode(byfid[10], 13, config)
    cases.append('bbob/30/10/13')
    for k, v in rec.items():
        data[f'bbob/30/10/13/{k}'] = v
    byid, _, _ = protein_problems()
    config = ref_import.ref_config(['--problem', 'protein'], scratch)
    p = byid['1ATN_7']
    p.reset()
    rec = run_gleet_episode(p, 14, config)
    cases.append('protein/12/1ATN_7/14')
    for k, v in rec.items():
        data[f'protein/12/1ATN_7/14/{k}'] = v
    print('protein', len(rec['gbest']), rec['gbest'][-1], rec['fes'])
    data['cases'] = np.array(cases)
    np.savez_compressed(os.path.join(OUT, 'gleet_traces.npz'), **data)


def gen_gleet_policy():
    """GLEET actor / critic (no checkpoint ships for this agent): weights of a seeded fresh reference agent + I/O pairs."""
    scratch = tempfile.mkdtemp()
    config = ref_import.ref_config(['--problem', 'bbob', '--dim', '10'], scratch)
    torch.manual_seed(321)
    from agent import GLEET_Agent
    agent = GLEET_Agent(copy_config(config))
    pol = {'actor/' + k: v.detach().cpu().numpy() for k, v in agent.actor.state_dict().items()}
    pol.update({'critic/' + k: v.detach().cpu().numpy() for k, v in agent.critic.state_dict().items()})
    x = torch.rand(3, 100, 27) * 2 - 0.5
    with torch.no_grad():
        z = agent.actor(x, only_critic=True)
        mu = (torch.tanh(agent.actor.mu_net(z)) + 1.) / 2.
        sigma = (torch.tanh(agent.actor.sigma_net(z)) + 1.) / 2. * (agent.actor.max_sigma - agent.actor.min_sigma) + agent.actor.min_sigma
        fixed = torch.rand(3, 100, 1)
        _, logp, _ = agent.actor(x, fixed_action=fixed)
        value = agent.critic(z)[0]
    pol.update({'io/x': x.numpy(), 'io/z': z.numpy(), 'io/mu': mu.numpy(), 'io/sigma': sigma.numpy(), 'io/fixed': fixed.numpy(),
                'io/logp': logp.numpy(), 'io/value': value.numpy()})
    np.savez_compressed(os.path.join(OUT, 'gleet_policy.npz'), **pol)
    print('gleet policy:', {k: v.shape for k, v in pol.items()})


def run_qlpso_episode(problem, seed, agent, config, mode, opt=None):
    """One reference QLPSO rollout.  mode 'policy': QLPSO_Agent.__get_action (softmax over the Q-row + np.random.choice from the
    global stream); mode 'uniform': actions from a private RandomState (the global stream then only feeds the optimizer)."""
    from optimizer import QLPSO_Optimizer
    from environment import PBO_Env
    import copy
    opt = opt or QLPSO_Optimizer(copy.deepcopy(config))
    env = PBO_Env(problem, opt)
    np.random.seed(seed)
    ars = np.random.RandomState(60_000 + seed)
    state = env.reset()
    rec = dict(actions=[], gbest=[], reward=[], done=[], states=[state])
    done = False
    while not done:
        a = agent._QLPSO_Agent__get_action(state) if mode == 'policy' else np.array([ars.randint(0, 4)])
        state, r, done = env.step(a)
        rec['actions'].append(int(a[0]))
        rec['gbest'].append(float(opt._QLPSO_Optimizer__gbest_cost))
        rec['reward'].append(float(r))
        rec['done'].append(bool(done))
        rec['states'].append(int(np.squeeze(state)))
    return opt, {'actions': np.array(rec['actions'], dtype=np.uint8), 'gbest': np.array(rec['gbest']), 'reward': np.array(rec['reward'], dtype=np.int8),
                 'done': np.array(rec['done']), 'states': np.array(rec['states'], dtype=np.uint8), 'cost': np.array(opt.cost, dtype=np.float64),
                 'fes': np.float64(opt.fes), 'final_pop': np.array(opt._QLPSO_Optimizer__population),
                 'final_cost': np.array(opt._QLPSO_Optimizer__cost), 'diversity': np.float64(opt._QLPSO_Optimizer__diversity),
                 'pointer': np.int64(opt._QLPSO_Optimizer__solution_pointer)}


def gen_qlpso():
    """QLPSO: the shipped bbob_easy Q-table and seeded reference episodes (policy- and uniform-driven); one optimizer
    object is reused for two consecutive episodes because the reference never resets its particle pointer."""
    scratch = tempfile.mkdtemp()
    agent = load_shipped(os.path.join(ref_import.REF_SRC, 'agent_model/test/bbob_easy/QLPSO_Agent.pkl'))
    q = np.array(agent._QLPSO_Agent__q_table, dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, 'qlpso_policy.npz'), q_table=q)
    pkg = os.path.join(os.path.dirname(HERE), 'metabox_amd', 'agent_model')
    np.savez_compressed(os.path.join(pkg, 'qlpso_bbob_easy.npz'), q_table=q)
    print('q_table', q)
    data, cases = {}, []
    for suite, fids, seed, mode in (('bbob', (1, 8, 17, 21), 15, 'policy'), ('bbob', (3,), 16, 'uniform'), ('bbob-noisy', (105, 118, 129), 17, 'policy')):
        config = ref_import.ref_config(['--problem', suite, '--dim', '10'], scratch)
        config.maxFEs = 2500
        config.log_interval = config.maxFEs // config.n_logpoint
        tr, te, _ = all_problems(suite, 10)
        byfid = {fid_of(p): p for p in tr + te}
        for fid in fids:
            p = byfid[fid]
            p.reset()
            opt, rec = run_qlpso_episode(p, seed, agent, config, mode)
            key = f'{suite}/10/{fid}/{seed}/{mode}'
            cases.append(key)
            for k, v in rec.items():
                data[f'{key}/{k}'] = v
            print(key, len(rec['gbest']), rec['gbest'][-1], rec['fes'], rec['pointer'], np.bincount(rec['actions'], minlength=4))
            if fid == 8:                            # second episode on the SAME optimizer object: the pointer carries over
                _, rec2 = run_qlpso_episode(p, seed + 100, agent, config, mode, opt=opt)
                for k, v in rec2.items():
                    data[f'{key}/second/{k}'] = v
                print('   second episode', rec2['gbest'][-1], rec2['pointer'])
    data['cases'] = np.array(cases)
    np.savez_compressed(os.path.join(OUT, 'qlpso_traces.npz'), **data)


def copy_config(config):
    import copy
    return copy.deepcopy(config)


def gen_harness():
    """Random_search episodes (the AEI normaliser) and the metric functions of src/logger.py on a synthetic results dict."""
    import copy
    import json
    scratch = tempfile.mkdtemp()
    from optimizer import Random_search
    data, cases = {}, []
    for suite, dim, fids in (('bbob', 10, (1, 16)), ('bbob-noisy', 10, (117,)), ('protein', 12, ('1AVX_1',))):
        argv = ['--problem', suite] + ([] if suite == 'protein' else ['--dim', str(dim)])
        config = ref_import.ref_config(argv, scratch)
        if suite == 'protein':
            byid = protein_problems()[0]
        else:
            tr, te, _ = all_problems(suite, dim)
            byid = {fid_of(p): p for p in tr + te}
        for fid in fids:
            for seed in (0, 7):
                opt = Random_search(copy.deepcopy(config))
                np.random.seed(seed)
                info = opt.run_episode(byid[fid])
                key = f'rs/{suite}/{dim}/{fid}/{seed}'
                cases.append(key)
                data[f'{key}/cost'] = np.array(info['cost'], dtype=np.float64)
                data[f'{key}/fes'] = np.float64(info['fes'])
                print(key, len(info['cost']), info['fes'], info['cost'][-1])
    data['cases'] = np.array(cases)
    np.savez_compressed(os.path.join(OUT, 'random_search.npz'), **data)
    # ---- metrics: synthetic test.pkl-shaped dicts -> get_random_baseline / aei_metric / cec_metric
    from logger import Logger, get_random_baseline
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), 'tests'))
    from helpers import metric_inputs
    test, rand = metric_inputs()
    config = ref_import.ref_config(['--problem', 'bbob', '--dim', '10'], scratch)
    lg = Logger(config)
    base = get_random_baseline(rand, 20000)
    mean, std = lg.aei_metric(copy.deepcopy(test), rand, maxFEs=20000)
    cec = lg.cec_metric(copy.deepcopy(test))
    config_p = ref_import.ref_config(['--problem', 'protein'], scratch)
    mean_p, std_p = Logger(config_p).aei_metric(copy.deepcopy(test), rand, maxFEs=1000)
    out = {'baseline': {k: float(v) for k, v in base.items()},
           'aei_mean': {k: float(v) for k, v in mean.items()}, 'aei_std': {k: float(v) for k, v in std.items()},
           'aei_mean_protein': {k: float(v) for k, v in mean_p.items()}, 'aei_std_protein': {k: float(v) for k, v in std_p.items()},
           'cec': {k: float(v) for k, v in cec.items()}}
    with open(os.path.join(OUT, 'metrics.json'), 'w') as f:
        json.dump(out, f)
    print('metrics:', out['aei_mean'], out['cec'])


def gen_stats():
    """End-to-end statistics of the reference: final cost / fes / return of the shipped RLEPSO agent over 40 seeded runs of
    each bbob-easy test problem (np.random.seed(r); torch.manual_seed(r)).  The GPU test compares the Philox-driven batched
    engine against these distributions."""
    scratch = tempfile.mkdtemp()
    from optimizer import RLEPSO_Optimizer
    from environment import PBO_Env
    import copy
    agent = load_shipped(os.path.join(ref_import.REF_SRC, 'agent_model/test/bbob_easy/RLEPSO_Agent.pkl'))
    config = ref_import.ref_config(['--problem', 'bbob', '--dim', '10'], scratch)
    tr, te, _ = all_problems('bbob', 10)
    data = {}
    for p in te:
        fc, fes, ret = [], [], []
        for r in range(40):
            opt = RLEPSO_Optimizer(copy.deepcopy(config))
            np.random.seed(r)
            torch.manual_seed(r)
            p.reset()
            with torch.no_grad():
                info = agent.rollout_episode(PBO_Env(p, opt))
            fc.append(info['cost'][-1]); fes.append(info['fes']); ret.append(info['return'])
        data[f'{fid_of(p)}/final_cost'] = np.array(fc, dtype=np.float64)
        data[f'{fid_of(p)}/fes'] = np.array(fes, dtype=np.float64)
        data[f'{fid_of(p)}/return'] = np.array(ret, dtype=np.float64)
        print(fid_of(p), str(p), np.mean(fc), np.std(fc), np.mean(fes))
    np.savez_compressed(os.path.join(OUT, 'rlepso_stats.npz'), **data)


def gen_lde_stats():
    """Reference LDE (shipped bbob_easy weights) on three bbob problems, 30 seeded runs each: final cost / fes / return."""
    scratch = tempfile.mkdtemp()
    from optimizer import LDE_Optimizer
    from environment import PBO_Env
    import copy
    agent = load_shipped(os.path.join(ref_import.REF_SRC, 'agent_model/test/bbob_easy/LDE_Agent.pkl'))
    config = ref_import.ref_config(['--problem', 'bbob', '--dim', '10'], scratch)
    tr, te, _ = all_problems('bbob', 10)
    byid = {fid_of(p): p for p in tr + te}
    data = {}
    for fid in (1, 15, 20):
        p = byid[fid]
        fc, fes, ret = [], [], []
        for r in range(30):
            opt = LDE_Optimizer(copy.deepcopy(config))
            np.random.seed(r)
            torch.manual_seed(r)
            p.reset()
            with torch.no_grad():
                info = agent.rollout_episode(PBO_Env(p, opt))
            fc.append(info['cost'][-1]); fes.append(info['fes']); ret.append(info['return'])
        data[f'{fid}/final_cost'] = np.array(fc, dtype=np.float64)
        data[f'{fid}/fes'] = np.array(fes, dtype=np.float64)
        data[f'{fid}/return'] = np.array(ret, dtype=np.float64)
        print(fid, str(p), np.mean(fc), np.std(fc), np.mean(fes), np.mean(ret))
    np.savez_compressed(os.path.join(OUT, 'lde_stats.npz'), **data)


def gen_mte():
    """MTE of the reference (src/tester.py:500-608) on two seeded synthetic rollout.pkl files; the value is parsed from the
    line the reference prints (the function only prints and plots)."""
    import contextlib
    import io
    import json
    import pickle
    os.environ['MPLBACKEND'] = 'Agg'
    import matplotlib
    matplotlib.use('Agg')
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), 'tests'))
    from helpers import fake_rollout
    import tester as ref_tester
    scratch = tempfile.mkdtemp()
    out = {}
    for tag, (s_pre, t_pre, s_scr, t_scr) in {'a': (1, 1.0, 2, 0.6), 'b': (3, 0.3, 4, 1.0), 'c': (5, 1.5, 6, 0.2)}.items():
        pre, scr = os.path.join(scratch, f'pre_{tag}.pkl'), os.path.join(scratch, f'scr_{tag}.pkl')
        with open(pre, 'wb') as f:
            pickle.dump(fake_rollout(s_pre, trend=t_pre), f)
        with open(scr, 'wb') as f:
            pickle.dump(fake_rollout(s_scr, trend=t_scr), f)
        config = ref_import.ref_config(['--mte_test', '--problem_from', 'bbob', '--problem_to', 'bbob-noisy', '--agent', 'RLEPSO_Agent',
                                        '--pre_train_rollout', pre, '--scratch_rollout', scr], scratch)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            ref_tester.mte_test(config)
        line = [l for l in buf.getvalue().splitlines() if l.startswith('MTE(')][-1]
        out[tag] = {'seeds': [s_pre, t_pre, s_scr, t_scr], 'mte': float(line.rsplit(':', 1)[1]), 'line': line}
        print(tag, line)
    with open(os.path.join(OUT, 'mte.json'), 'w') as f:
        json.dump(out, f, indent=1)



# ---------------------------------------------------------------------------------------------------- rlepso_ties
# Per-generation bookkeeping of the reference for the episodes of rlepso_traces.npz: per_no_improve [G+1, NP] (uint16) for every episode and
# c_cost [G+1, NP] (float64) for the episodes in which this build's float64 path takes a different branch of `new_cost < c_cost`
# somewhere (tests/helpers.py: prove_tie shows at the first such generation that the reference's own margin is smaller than the deviation
# between the two implementations' operands).  Episodes with c_cost: those where the C oracle diverges (found here by running it next to
# the reference) plus EXTRA_TIE_CASES (episodes where only the HIP kernel diverges; the GPU test names them when they are missing).
EXTRA_TIE_CASES = {'bbob/15/2/actor', 'bbob/3/0/actor', 'bbob/3/1/actor', 'bbob/21/1/actor', 'bbob/20/2/actor', 'bbob/22/2/actor'}   # episodes in which one of the float64 paths (oracle before / after the fma-chain matvec, HIP kernel) has left the reference's branch at a near-tie: c_cost kept for all of them


def _ties_worker(case):
    import copy as _copy
    from optimizer import RLEPSO_Optimizer
    from environment import PBO_Env
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
    from oracle import oracle
    torch.set_num_threads(1)
    suite, fid, seed, mode = case.split('/')
    fid, seed = int(fid), int(seed)
    scratch = tempfile.mkdtemp()
    agent = load_shipped(os.path.join(ref_import.REF_SRC, 'agent_model/test/bbob_easy/RLEPSO_Agent.pkl'))
    config = ref_import.ref_config(['--problem', suite, '--dim', '10'], scratch)
    tr, te, _ = all_problems(suite, 10)
    p = {fid_of(q): q for q in tr + te}[fid]
    p.reset()
    opt = RLEPSO_Optimizer(_copy.deepcopy(config))
    env = PBO_Env(p, opt)
    actor = agent._RLEPSO_Agent__actor
    np.random.seed(seed)
    torch.manual_seed(seed)
    ars = np.random.RandomState(10_000 + seed)
    state = env.reset()
    part = opt._RLEPSO_Optimizer__particles
    cc, pni, acts = [part['c_cost'].copy()], [np.array(opt._RLEPSO_Optimizer__per_no_improve).copy()], []
    done = False
    while not done:
        if mode == 'actor':
            with torch.no_grad():
                a = actor(torch.FloatTensor(state))[0].cpu().numpy()
        else:
            a = ars.uniform(0, 1, size=35).astype(np.float32)
        state, r, done = env.step(a)
        part = opt._RLEPSO_Optimizer__particles
        acts.append(a.astype(np.float32))
        cc.append(part['c_cost'].copy())
        pni.append(np.array(opt._RLEPSO_Optimizer__per_no_improve).copy())
    cc, pni = np.stack(cc), np.stack(pni)
    # the C oracle on the same tape: does it take a different branch anywhere?
    from metabox_amd.problem.bbob import BBOB_Dataset
    mtr, mte = BBOB_Dataset.get_datasets(suite, 10, 5.0)
    mp = {q.func_id: q for q in mtr.data + mte.data}[fid]
    cfg = oracle.make_cfg(1, 100, 10, 20000, 400, 50)
    o = oracle.RlepsoOracle(mp.desc(), mp.bias, cfg)
    fd = oracle.NumpyTapeFeeder(seed, 100, 10, mp.noise[0])
    o.reset(fd.reset_tape())
    first = -1
    for g, a in enumerate(acts):
        o.step(a, fd.step_tape())
        st = oracle.split_rlepso_state(o.state(), 100, 10, 50)
        fd.commit(st['scalars'][oracle.SC_REINIT] > 0)
        if not np.array_equal(st['pni'], pni[g + 1]):
            first = g
            break
    return case, cc, pni.astype(np.uint16), first


def gen_rlepso_ties():
    import multiprocessing as mp
    tr = np.load(os.path.join(OUT, 'rlepso_traces.npz'))
    cases = [str(c) for c in tr['cases']]
    data = {'cases': np.array(cases)}
    with mp.get_context('fork').Pool(6) as pool:
        for case, cc, pni, first in pool.imap_unordered(_ties_worker, cases):
            data[f'{case}/pni'] = pni
            data[f'{case}/oracle_first_divergence'] = np.int32(first)
            if first >= 0 or case in EXTRA_TIE_CASES:
                data[f'{case}/ccost'] = cc
            print(case, 'gens', len(pni) - 1, 'oracle first divergence', first, flush=True)
    np.savez_compressed(os.path.join(OUT, 'rlepso_ties.npz'), **data)



# ---------------------------------------------------------------------------------------------------- rlepso_hd
# Whole RLEPSO reference episodes at the geometries of BASELINE configs 3 / 5: `--dim 30` and `--dim 40` with the reference's own
# NP = 100, and D = 40 with NP = 128.  NP is the literal `config.NP = 100` of src/optimizer/rlepso_optimizer.py:11; for NP = 128 that ONE constant is
# patched here, in the generator: the reference module's text is read from /root/reference, the literal replaced, and the result compiled in memory
# (never written anywhere) -- every derived quantity (pci :24-25, per_no_improve :30, the NP // n_group slices of __get_coe :117-126 that leave
# particles 125-127 with zero coefficients, all draw shapes) is then computed by the reference's own code.
# Per episode: actions, gbest / fes / reward / done per generation, per_no_improve per generation (uint16), final arrays, the generation at which the C
# oracle's bookkeeping first leaves the reference's (-1: never) and, for those episodes and HD_EXTRA_TIE_CASES, the reference's c_cost per generation.
HD_EXTRA_TIE_CASES = set()
HD_CASES = (
    # (suite, dim, NP, function ids, seed, action mode)
    ('bbob', 30, 100, (1, 7, 15, 16, 21, 24), 0, 'actor'),
    ('bbob-noisy', 30, 100, (103, 115, 124, 128), 1, 'actor'),
    ('bbob', 30, 100, (8,), 3, 'uniform'),
    ('bbob-noisy', 30, 100, (108,), 4, 'uniform'),
    ('bbob', 40, 100, (3, 5, 16, 20, 21, 24), 0, 'actor'),
    ('bbob-noisy', 40, 100, (102, 119, 126, 130), 1, 'actor'),
    ('bbob', 40, 100, (9,), 3, 'uniform'),
    ('bbob', 40, 128, (1, 6, 12, 16, 21, 23, 24), 2, 'actor'),
    ('bbob-noisy', 40, 128, (101, 117, 121, 129), 2, 'actor'),
    ('bbob', 40, 128, (2,), 5, 'uniform'),
    ('bbob-noisy', 40, 128, (113,), 5, 'uniform'),
)


def rlepso_class_with_np(np_):
    """RLEPSO_Optimizer of the reference with its hard-coded population size replaced (see above); np_ == 100 returns the class as shipped."""
    from optimizer import RLEPSO_Optimizer
    if np_ == 100:
        return RLEPSO_Optimizer
    path = os.path.join(ref_import.REF_SRC, 'optimizer', 'rlepso_optimizer.py')
    with open(path) as f:
        text = f.read()
    assert text.count('config.NP = 100') == 1
    ns = {'__name__': f'optimizer.rlepso_optimizer_np{np_}'}
    exec(compile(text.replace('config.NP = 100', f'config.NP = {np_}'), path, 'exec'), ns)
    return ns['RLEPSO_Optimizer']


def _hd_worker(job):
    import copy as _copy
    from environment import PBO_Env
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
    from oracle import oracle
    torch.set_num_threads(1)
    suite, dim, np_, fid, seed, mode = job
    case = f'{suite}/{dim}/{np_}/{fid}/{seed}/{mode}'
    scratch = tempfile.mkdtemp()
    agent = load_shipped(os.path.join(ref_import.REF_SRC, 'agent_model/test/bbob_easy/RLEPSO_Agent.pkl'))
    config = ref_import.ref_config(['--problem', suite, '--dim', str(dim)], scratch)
    tr, te, _ = all_problems(suite, dim)
    p = {fid_of(q): q for q in tr + te}[fid]
    p.reset()
    cls = rlepso_class_with_np(np_)
    opt = cls(_copy.deepcopy(config))
    priv = lambda name: getattr(opt, '_RLEPSO_Optimizer__' + name)
    assert priv('NP') == np_ and len(priv('pci')) == np_
    env = PBO_Env(p, opt)
    actor = agent._RLEPSO_Agent__actor
    np.random.seed(seed)
    torch.manual_seed(seed)
    ars = np.random.RandomState(10_000 + seed)
    state = env.reset()
    part = priv('particles')
    gb0 = float(part['gbest_val'])
    cc, pni = [part['c_cost'].copy()], [np.array(priv('per_no_improve')).copy()]
    acts, gbest, fes, reward, done_l = [], [], [], [], []
    done = False
    while not done:
        if mode == 'actor':
            with torch.no_grad():
                a = actor(torch.FloatTensor(state))[0].cpu().numpy()
        else:
            a = ars.uniform(0, 1, size=35).astype(np.float32)
        state, r, done = env.step(a)
        part = priv('particles')
        acts.append(a.astype(np.float32))
        gbest.append(float(part['gbest_val'])); fes.append(float(opt.fes)); reward.append(float(r)); done_l.append(bool(done))
        cc.append(part['c_cost'].copy())
        pni.append(np.array(priv('per_no_improve')).copy())
    cc, pni = np.stack(cc), np.stack(pni)
    rec = dict(actions=np.stack(acts), gbest=np.array(gbest), fes=np.array(fes), reward=np.array(reward), done=np.array(done_l),
               cost=np.array(opt.cost, dtype=np.float64), gbest0=np.float64(gb0), final_pos=np.array(part['current_position']),
               final_pbest=np.array(part['pbest']), final_pni=np.array(priv('per_no_improve')), pni=pni.astype(np.uint16),
               pci=np.array(priv('pci'), dtype=np.float64))
    # the C oracle on the same tape: the first generation at which its bookkeeping leaves the reference's
    from metabox_amd.problem.bbob import BBOB_Dataset
    mtr, mte = BBOB_Dataset.get_datasets(suite, dim, 5.0)
    mp_ = {q.func_id: q for q in mtr.data + mte.data}[fid]
    maxfes = int(config.maxFEs)
    cfg = oracle.make_cfg(1, np_, dim, maxfes, maxfes // 50, 50)
    o = oracle.RlepsoOracle(mp_.desc(), mp_.bias, cfg)
    fd = oracle.NumpyTapeFeeder(seed, np_, dim, mp_.noise[0])
    o.reset(fd.reset_tape())
    first = -1
    for g, a in enumerate(acts):
        o.step(a, fd.step_tape())
        st = oracle.split_rlepso_state(o.state(), np_, dim, 50)
        fd.commit(st['scalars'][oracle.SC_REINIT] > 0)
        if not np.array_equal(st['pni'], pni[g + 1]):
            first = g
            break
    rec['oracle_first_divergence'] = np.int32(first)
    if first >= 0 or case in HD_EXTRA_TIE_CASES:
        rec['ccost'] = cc
    return case, rec


def gen_rlepso_hd():
    import multiprocessing as mp
    jobs = [(suite, dim, np_, fid, seed, mode) for suite, dim, np_, fids, seed, mode in HD_CASES for fid in fids]
    data, cases = {}, []
    with mp.get_context('fork').Pool(7) as pool:
        for case, rec in pool.imap(_hd_worker, jobs):
            cases.append(case)
            for k, v in rec.items():
                data[f'{case}/{k}'] = v
            np_ = int(case.split('/')[2])
            n_re = int(np.sum(np.diff(np.concatenate([[float(np_)], rec['fes']])) != np_))
            print(f'{case}: gens={len(rec["gbest"])} fes={rec["fes"][-1]:.0f} final={rec["gbest"][-1]:.6g} reinit_steps={n_re} '
                  f'oracle_first_divergence={int(rec["oracle_first_divergence"])}', flush=True)
    data['cases'] = np.array(cases)
    np.savez_compressed(os.path.join(OUT, 'rlepso_traces_hd.npz'), **data)
    print('rlepso_hd:', len(cases), 'episodes')


# ---------------------------------------------------------------------------------------------------- lde_hd
# Whole LDE reference episodes at the geometry of BASELINE config 3 (bbob-noisy, D = 30): the reference's own NP = 50 -- one function per noise model,
# two Gallaghers, two noise-free functions -- and NP = 100 ("pop=100" as config 3 is written), for which the ONE literal `self.__config.NP = 50` of
# src/optimizer/lde_optimizer.py:10 is patched here, in the generator (module text read from /root/reference, literal replaced, compiled in memory; P_MIN :14,
# the histogram prior :139 and every draw shape then follow from the reference's own code).  The shipped PolicyNet is NP = 50 only, so the NP = 100 episodes
# (and most NP = 50 ones, to keep the fixture small) are driven by seeded uniform actions, which the tests regenerate instead of loading.
LDE_HD_CASES = (
    # (suite, dim, NP, fid, seed, mode)
    ('bbob-noisy', 30, 50, 115, 3, 'actor'),          # Cauchy noise, step ellipsoid
    ('bbob-noisy', 30, 50, 128, 4, 'actor'),          # Gauss noise, Gallagher 101 peaks
    ('bbob-noisy', 30, 50, 104, 5, 'uniform'),        # Gauss, Rosenbrock
    ('bbob-noisy', 30, 50, 120, 6, 'uniform'),        # uniform noise, different powers
    ('bbob-noisy', 30, 50, 124, 7, 'uniform'),        # Cauchy, Schaffers
    ('bbob-noisy', 30, 50, 130, 8, 'uniform'),        # Cauchy, Gallagher
    ('bbob', 30, 50, 16, 9, 'uniform'),
    ('bbob', 30, 50, 21, 10, 'uniform'),
    ('bbob-noisy', 30, 100, 101, 11, 'uniform'),      # NP = 100: Gauss
    ('bbob-noisy', 30, 100, 109, 12, 'uniform'),      # Cauchy (severe)
    ('bbob-noisy', 30, 100, 117, 13, 'uniform'),      # uniform noise, ellipsoid
    ('bbob-noisy', 30, 100, 129, 14, 'uniform'),      # uniform noise, Gallagher
    ('bbob-noisy', 30, 100, 126, 15, 'uniform'),      # uniform noise, Griewank-Rosenbrock
)


def lde_class_with_np(np_):
    from optimizer import LDE_Optimizer
    if np_ == 50:
        return LDE_Optimizer
    path = os.path.join(ref_import.REF_SRC, 'optimizer', 'lde_optimizer.py')
    with open(path) as f:
        text = f.read()
    assert text.count('self.__config.NP = 50') == 1
    ns = {'__name__': f'optimizer.lde_optimizer_np{np_}'}
    exec(compile(text.replace('self.__config.NP = 50', f'self.__config.NP = {np_}'), path, 'exec'), ns)
    return ns['LDE_Optimizer']


def _lde_hd_worker(job):
    torch.set_num_threads(1)
    suite, dim, np_, fid, seed, mode = job
    scratch = tempfile.mkdtemp()
    agent = load_shipped(os.path.join(ref_import.REF_SRC, 'agent_model/test/bbob_easy/LDE_Agent.pkl')) if mode == 'actor' else None
    config = ref_import.ref_config(['--problem', suite, '--dim', str(dim)], scratch)
    tr, te, _ = all_problems(suite, dim)
    p = {fid_of(q): q for q in tr + te}[fid]
    p.reset()
    rec = run_lde_episode(p, seed, agent, config, mode, NP=np_, keep_actions=(mode == 'actor'))
    return f'{suite}/{dim}/{np_}/{fid}/{seed}/{mode}', rec


def gen_lde_hd():
    import multiprocessing as mp
    data, cases = {}, []
    with mp.get_context('fork').Pool(7) as pool:
        for case, rec in pool.imap(_lde_hd_worker, LDE_HD_CASES):
            cases.append(case)
            for k, v in rec.items():
                data[f'{case}/{k}'] = v
            print(f'{case}: gens={len(rec["gbest"])} fes={rec["fes"][-1]:.0f} final={rec["gbest"][-1]:.6g} first_tie_gen={int(rec["first_tie_gen"])}', flush=True)
    data['cases'] = np.array(cases)
    np.savez_compressed(os.path.join(OUT, 'lde_traces_hd.npz'), **data)
    print('lde_hd:', len(cases), 'episodes')



# ---------------------------------------------------------------------------------------------------- train
# One training update of each reference agent on a SCRIPTED environment (states / rewards / done are fixed sequences, so nothing but the
# agent's own arithmetic is exercised): RLEPSO's PPO segment(s) (rlepso_agent.py:113-292), LDE's REINFORCE step (lde_agent.py:85-145) and
# DE-DDQN's double-DQN updates (de_ddqn_agent.py:70-106).  Recorded: initial weights, what the agent saw and did (states, the actions it
# sampled, rewards, segment / trajectory lengths, replay mini-batches), the gradient of every parameter at every optimizer step and the
# weights afterwards.  tests/test_training_parity.py feeds the same data to this framework's update code.
def _hook_steps(opt, params_named, sink, tag):
    orig = opt.step

    def step(*a, **k):
        sink.append((tag, {n: p.grad.detach().cpu().numpy().copy() for n, p in params_named}))
        return orig(*a, **k)
    opt.step = step


def gen_train():
    import copy as _copy
    import random as _random
    import types
    from agent import DE_DDQN_Agent, LDE_Agent, RLEPSO_Agent
    scratch = tempfile.mkdtemp()
    data = {}

    # ---- RLEPSO / PPO: episodes of 10 (one full segment), 7 (short segment) and 13 (10 + 3) steps
    for tag, T in (('ppo10', 10), ('ppo7', 7), ('ppo13', 13)):
        config = ref_import.ref_config(['--problem', 'bbob', '--dim', '10', '--max_learning_step', '1000'], scratch)
        config.save_interval = 10 ** 9
        torch.manual_seed(100 + T)
        agent = RLEPSO_Agent(_copy.deepcopy(config))
        actor, critic = agent._RLEPSO_Agent__actor, agent._RLEPSO_Agent__critic
        for k, v in actor.state_dict().items():
            data[f'{tag}/init/actor/{k}'] = v.cpu().numpy().copy()
        for k, v in critic.state_dict().items():
            data[f'{tag}/init/critic/{k}'] = v.cpu().numpy().copy()
        sink = []
        _hook_steps(agent._RLEPSO_Agent__optimizer_actor, list(actor.named_parameters()), sink, 'actor')
        _hook_steps(agent._RLEPSO_Agent__optimizer_critic, list(critic.named_parameters()), sink, 'critic')
        rs = np.random.RandomState(T)

        class Env:
            optimizer = types.SimpleNamespace(cost=[1.0, 0.5])

            def __init__(self):
                self.actions, self.rewards, self.states, self.t = [], [], [], 0

            def reset(self):
                self.t = 0
                self.states.append(np.array([100. / 20000.]))
                return self.states[-1]

            def step(self, a):
                self.actions.append(np.asarray(a, dtype=np.float32).copy())
                self.t += 1
                r = float(rs.choice([-1., 1.]))
                self.rewards.append(r)
                self.states.append(np.array([(100. + 100. * self.t + rs.randint(0, 40)) / 20000.]))
                return self.states[-1], r, self.t >= T
        env = Env()
        torch.manual_seed(5)
        agent.train_episode(env)
        data[f'{tag}/states'] = np.stack(env.states)                 # [T + 1, 1]
        data[f'{tag}/actions'] = np.stack(env.actions)               # [T, 35]
        data[f'{tag}/rewards'] = np.array(env.rewards)
        n_upd = len(sink) // 2
        for u in range(n_upd):
            for which, grads in (sink[2 * u], sink[2 * u + 1]):
                for k, g in grads.items():
                    data[f'{tag}/grad{u}/{which}/{k}'] = g
        data[f'{tag}/n_updates'] = np.int32(n_upd)
        for k, v in actor.state_dict().items():
            data[f'{tag}/post/actor/{k}'] = v.cpu().numpy().copy()
        for k, v in critic.state_dict().items():
            data[f'{tag}/post/critic/{k}'] = v.cpu().numpy().copy()
        print(tag, 'updates', n_upd)

    # ---- LDE / REINFORCE: 20 trajectories; equal lengths (6 each) and ragged lengths (the reference slices total // 20 per trajectory)
    for tag, lens in (('lde_equal', [6] * 20), ('lde_ragged', [4 + (i % 5) for i in range(20)])):
        config = ref_import.ref_config(['--problem', 'bbob', '--dim', '10', '--max_learning_step', '1000'], scratch)
        config.save_interval = 10 ** 9
        torch.manual_seed(77)
        agent = LDE_Agent(_copy.deepcopy(config))
        net = agent._LDE_Agent__net
        for k, v in net.state_dict().items():
            data[f'{tag}/init/net/{k}'] = v.cpu().numpy().copy()
        sink = []
        _hook_steps(agent._LDE_Agent__optimizer, list(net.named_parameters()), sink, 'net')
        rs = np.random.RandomState(len(tag))

        class LEnv:
            optimizer = types.SimpleNamespace(cost=[1.0, 0.5])

            def __init__(self):
                self.inputs, self.actions, self.rewards, self.traj, self.t = [], [], [], -1, 0

            def reset(self):
                self.traj += 1
                self.t = 0
                self.cur = rs.uniform(0, 1, size=(1, 60))
                return self.cur

            def step(self, a):
                self.inputs.append(self.cur[0].copy())
                self.actions.append(np.asarray(a, dtype=np.float32).reshape(-1).copy())
                self.t += 1
                r = np.array([rs.uniform(0, 0.3)])
                self.rewards.append(float(r[0]))
                self.cur = rs.uniform(0, 1, size=(1, 60))
                return self.cur, r, self.t >= lens[self.traj]
        env = LEnv()
        torch.manual_seed(9)
        agent.train_episode(env)
        data[f'{tag}/inputs'] = np.stack(env.inputs)
        data[f'{tag}/actions'] = np.stack(env.actions)
        data[f'{tag}/rewards'] = np.array(env.rewards)
        data[f'{tag}/lens'] = np.array(lens, dtype=np.int32)
        for k, g in sink[0][1].items():
            data[f'{tag}/grad0/net/{k}'] = g
        for k, v in net.state_dict().items():
            data[f'{tag}/post/net/{k}'] = v.cpu().numpy().copy()
        print(tag, 'steps', len(env.inputs), 'updates', len(sink))

    # ---- DE-DDQN: three double-DQN updates on recorded replay mini-batches (warm-up shortened to 64 transitions)
    config = ref_import.ref_config(['--problem', 'protein', '--max_learning_step', '3'], scratch)
    config.save_interval = 10 ** 9
    torch.manual_seed(31)
    agent = DE_DDQN_Agent(_copy.deepcopy(config))
    agent._DE_DDQN_Agent__warm_up_size = 64
    agent._DE_DDQN_Agent__max_learning_step = 3
    pred = agent._DE_DDQN_Agent__pred_func
    for k, v in pred.state_dict().items():
        data[f'ddqn/init/net/{k}'] = v.cpu().numpy().copy()
    sink, batches = [], []
    _hook_steps(agent._DE_DDQN_Agent__optimizer, list(pred.named_parameters()), sink, 'net')
    rb = agent._DE_DDQN_Agent__replay_buffer
    orig_sample = rb.sample

    def sample(n):
        out = orig_sample(n)
        batches.append([t.cpu().numpy().copy() for t in out])
        return out
    rb.sample = sample
    rs = np.random.RandomState(4)

    class DEnv:
        optimizer = types.SimpleNamespace(cost=[1.0, 0.5])

        def __init__(self):
            self.t = 0

        def reset(self):
            self.t = 0
            return rs.uniform(0, 1, size=99)

        def step(self, a):
            self.t += 1
            return rs.uniform(0, 1, size=99), float(rs.uniform(0, 2) * (rs.rand() < 0.3)), self.t >= 200
    _random.seed(12)
    np.random.seed(12)
    agent.train_episode(DEnv())
    assert len(sink) == 3 and len(batches) == 3
    for u in range(3):
        for name, arr in zip(('obs', 'act', 'rew', 'nxt', 'done'), batches[u]):
            data[f'ddqn/batch{u}/{name}'] = arr
        for k, g in sink[u][1].items():
            data[f'ddqn/grad{u}/net/{k}'] = g
    for k, v in pred.state_dict().items():
        data[f'ddqn/post/net/{k}'] = v.cpu().numpy().copy()
    print('ddqn updates', len(sink))
    np.savez_compressed(os.path.join(OUT, 'train_updates.npz'), **data)



SECTIONS = {'train': gen_train, 'lde_hd': gen_lde_hd, 'rlepso_hd': gen_rlepso_hd, 'rlepso_ties': gen_rlepso_ties, 'qlpso': gen_qlpso, 'gleet_policy': gen_gleet_policy, 'gleet': gen_gleet, 'rlpso': gen_rlpso, 'mte': gen_mte, 'lde_stats': gen_lde_stats, 'stats': gen_stats, 'harness': gen_harness, 'ddqn': gen_ddqn, 'protein': gen_protein, 'lde': gen_lde, 'instances': gen_instances, 'kat': gen_kat, 'noise': gen_noise, 'policy': gen_policy,
            'rlepso': gen_rlepso}

if __name__ == '__main__':
    todo = sys.argv[1:] or list(SECTIONS)
    for name in todo:
        SECTIONS[name]()
