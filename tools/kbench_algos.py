#!/usr/bin/env python
"""Throughput of the other batched paths (BASELINE.json configs 3 and 4, one GPU's share), policy included.
   python tools/kbench_algos.py [lde|ddqn|rs|rlpso|gleet|qlpso] """
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metabox_amd.config import get_config
from metabox_amd.environment import BatchedPBO_Env
from metabox_amd.utils import construct_problem_set

def timed(fn, steps):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(steps); torch.cuda.synchronize(); return time.perf_counter() - t0

which = sys.argv[1:] or ['lde', 'ddqn', 'rs', 'rlpso', 'gleet', 'qlpso']
if 'lde' in which:
    from metabox_amd.agent import LDE_Agent
    from metabox_amd.optimizer import LDE_Optimizer
    cfg = get_config(['--problem', 'bbob-noisy', '--dim', '30', '--device', 'cuda']); cfg.agent_save_dir = None
    agent = LDE_Agent(cfg).load_exported_weights(np.load(os.path.join(os.path.dirname(__file__), '..', 'metabox_amd', 'agent_model', 'lde_bbob_easy.npz'))).to('cuda')
    opt = LDE_Optimizer(cfg)
    tr, te = construct_problem_set(cfg); ps = sorted(tr.data + te.data, key=lambda p: p.func_id)
    B = 16384
    env = BatchedPBO_Env(ps, opt, np.arange(B) % len(ps), np.arange(B, dtype=np.uint64) + 1)
    state = env.reset(); h = torch.zeros(1, B, 50, device='cuda'); c = torch.zeros(1, B, 50, device='cuda')
    def run(n):
        global state, h, c
        with torch.no_grad():
            for _ in range(n):
                a, h, c = agent.net.act_batch(state.to(torch.float32), h, c)
                state, _, _ = env.step(a.contiguous())
    run(5); dt = timed(run, 60)
    live = int((env.results()['steps'] > 0).sum())
    print(json.dumps({'path': 'LDE bbob-noisy d=30 NP=50 (reference population), 16384 instances, LSTM policy via PyTorch', 'ms_per_step': dt / 60 * 1e3, 'env_steps_per_s': B * 60 / dt}))
    env.close()
if 'ddqn' in which:
    from metabox_amd.agent import DE_DDQN_Agent
    from metabox_amd.optimizer import DE_DDQN_Optimizer
    cfg = get_config(['--problem', 'protein', '--device', 'cuda']); cfg.agent_save_dir = None
    torch.manual_seed(0)
    agent = DE_DDQN_Agent(cfg).to('cuda'); opt = DE_DDQN_Optimizer(cfg)
    tr, te = construct_problem_set(cfg); ps = (tr + te).data[:35]            # one GPU's share of config 4: 35 problems x 64 runs
    B = 35 * 64
    env = BatchedPBO_Env(ps, opt, np.repeat(np.arange(35), 64), np.arange(B, dtype=np.uint64) + 1)
    state = env.reset()
    print(json.dumps({'ddqn_launch_info': env.batch.launch_info()}))
    def run(n):
        global state
        with torch.no_grad():
            for _ in range(n):
                a = torch.argmax(agent.q_net(state.to(torch.float32)), dim=1).to(torch.int32)
                state, _, _ = env.step(a.contiguous())
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side), torch.no_grad():
        for _ in range(3): torch.argmax(agent.q_net(state.to(torch.float32)), dim=1).to(torch.int32)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g), torch.no_grad():
        ga = torch.argmax(agent.q_net(state.to(torch.float32)), dim=1).to(torch.int32)
    def run_graph(n):
        for _ in range(n):
            g.replay(); env.step(ga)
    run_graph(5); dg = timed(run_graph, 200)
    print(json.dumps({'path': 'DE-DDQN protein, Q-net replayed as a hipGraph', 'ms_per_step': dg / 200 * 1e3, 'env_steps_per_s': B * 200 / dg}))
    packed = agent.packed_weights()
    def run_hip(n):
        for _ in range(n):
            env.step(env.batch.ddqn_qnet(packed))
    def run_q(n):
        for _ in range(n): env.batch.ddqn_qnet(packed)
    def run_s(n):
        a = env.batch.ddqn_qnet(packed)
        for _ in range(n): env.step(a)
    run_hip(5); dh = timed(run_hip, 200)
    print(json.dumps({'path': 'DE-DDQN protein, Q-net + argmax as one MFMA launch (mbx_ddqn_qnet; rollout_batch default)', 'ms_per_step': dh / 200 * 1e3, 'env_steps_per_s': B * 200 / dh}))
    dq = timed(run_q, 200); ds = timed(run_s, 200)
    print(json.dumps({'path': 'DE-DDQN protein: the two launches alone', 'qnet_us': dq / 200 * 1e6, 'k_dq_step_us': ds / 200 * 1e6}))
    run(5); dt = timed(run, 200)
    print(json.dumps({'path': 'DE-DDQN protein d=12 NP=100, 2240 instances (35 problems x 64 runs), Q-net via PyTorch', 'ms_per_step': dt / 200 * 1e3, 'env_steps_per_s': B * 200 / dt}))
    env.close()
if 'rs' in which:
    from metabox_amd.optimizer import Random_search
    from metabox_amd.suite import Suite
    cfg = get_config(['--problem', 'bbob', '--dim', '10'])
    tr, te = construct_problem_set(cfg); ps = (tr + te).data
    s = Suite(ps); B = 24 * 51
    rs = Random_search(cfg)
    torch.cuda.synchronize(); t0 = time.perf_counter(); rs.run_batch(s, np.repeat(np.arange(24), 51), np.arange(B)); dt = time.perf_counter() - t0
    print(json.dumps({'path': 'Random_search baseline epoch: 24 bbob problems x 51 runs, 199 populations each', 'seconds': dt}))
if 'rlpso' in which:
    from metabox_amd.agent import RL_PSO_Agent
    from metabox_amd.optimizer import RL_PSO_Optimizer
    cfg = get_config(['--problem', 'bbob', '--dim', '10', '--device', 'cuda']); cfg.agent_save_dir = None
    agent = RL_PSO_Agent(cfg).load_exported_weights(np.load(os.path.join(os.path.dirname(__file__), '..', 'metabox_amd', 'agent_model', 'rlpso_bbob_easy.npz'))).to('cuda')
    tr, te = construct_problem_set(cfg); ps = sorted(tr.data + te.data, key=lambda p: p.func_id)
    B = 4096
    nets = agent.nets; h1, h2 = nets.hidden_sizes(); net = (nets.packed_weights(), h1, h2, nets.min_sigma, nets.max_sigma)
    for mode, steps in (('fused', 2048), ('fused1', 512), ('hip', 256), ('torch', 128)):
        env = BatchedPBO_Env(ps, RL_PSO_Optimizer(cfg), np.arange(B) % len(ps), np.arange(B, dtype=np.uint64) + 1, early_stop=False)
        state = env.reset()
        def run(n):
            global state
            with torch.no_grad():
                if mode == 'fused':
                    for _ in range(n // 256): env.batch.rlpso_rollout(*net, 256)
                elif mode == 'fused1':
                    for _ in range(n): env.batch.rlpso_rollout(*net, 1)
                elif mode == 'hip':
                    for _ in range(n): env.step(env.batch.gauss_policy(*net))
                else:
                    for _ in range(n):
                        a, _ = nets(state.to(torch.float32)); state, _, _ = env.step(a.contiguous())
        run(256 if mode == 'fused' else 5); dt = timed(run, steps)
        print(json.dumps({'path': f'RL-PSO bbob d=10 NP=100, 4096 instances, policy={mode}', 'us_per_step': dt / steps * 1e6, 'env_steps_per_s': B * steps / dt}))
        env.close()
if 'gleet' in which:
    from metabox_amd.agent import GLEET_Agent
    from metabox_amd.optimizer import GLEET_Optimizer
    cfg = get_config(['--problem', 'bbob', '--dim', '10', '--device', 'cuda']); cfg.agent_save_dir = None
    torch.manual_seed(0)
    agent = GLEET_Agent(cfg).to('cuda')
    tr, te = construct_problem_set(cfg); ps = sorted(tr.data + te.data, key=lambda p: p.func_id)
    B = 4096
    env = BatchedPBO_Env(ps, GLEET_Optimizer(cfg), np.arange(B) % len(ps), np.arange(B, dtype=np.uint64) + 1, early_stop=False)
    state = env.reset()
    const = torch.full((B, 100), 0.5, dtype=torch.float32, device='cuda')
    def run_kernel(n):
        for _ in range(n): env.step(const)
    def run(n):
        global state
        for _ in range(n): state, _, _ = env.step(agent.act_batch(state))
    w = agent.actor.packed_weights()
    def run_hip(n):
        for _ in range(n): env.step(env.batch.gleet_policy(w, agent.actor.min_sigma, agent.actor.max_sigma))
    run_kernel(5); dk = timed(run_kernel, 60)
    run(3); dt = timed(run, 30)
    run_hip(5); dh = timed(run_hip, 60)
    print(json.dumps({'path': 'GLEET bbob d=10 NP=100, 4096 instances', 'kernel_us_per_step': dk / 60 * 1e6,
                      'kernel_env_steps_per_s': B * 60 / dk, 'with_torch_policy_ms_per_step': dt / 30 * 1e3, 'torch_env_steps_per_s': B * 30 / dt,
                      'with_hip_policy_us_per_step': dh / 60 * 1e6, 'hip_env_steps_per_s': B * 60 / dh}))
    env.close()
if 'qlpso' in which:
    from metabox_amd.optimizer import QLPSO_Optimizer
    cfg = get_config(['--problem', 'bbob', '--dim', '10', '--device', 'cuda']); cfg.agent_save_dir = None
    q = torch.from_numpy(np.load(os.path.join(os.path.dirname(__file__), '..', 'metabox_amd', 'agent_model', 'qlpso_bbob_easy.npz'))['q_table']).cuda()
    tr, te = construct_problem_set(cfg); ps = sorted(tr.data + te.data, key=lambda p: p.func_id)
    B = 4096
    env = BatchedPBO_Env(ps, QLPSO_Optimizer(cfg), np.arange(B) % len(ps), np.arange(B, dtype=np.uint64) + 1, early_stop=False)
    env.reset()
    def run(n):
        for _ in range(n // 256): env.batch.qlpso_rollout(q, 256)
    run(256); dt = timed(run, 2048)
    print(json.dumps({'path': 'QLPSO bbob d=10 NP=30, 4096 instances, tabular policy in the kernel, 256 steps per launch', 'us_per_step': dt / 2048 * 1e6,
                      'env_steps_per_s': B * 2048 / dt}))
    env.close()
