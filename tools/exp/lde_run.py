"""Config 3 alone (LDE bbob-noisy d=30, 16 384 instances, LSTM policy included; --suite / --dim for the reference's own bbob d=10 setting), for profiling:
   python tools/exp/lde_run.py [--pop 50,100] [--steps 20] [--route step|resident] [--gens-per-launch 10]
Prints one JSON line per population with the wall time per generation."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from metabox_amd.agent import LDE_Agent
from metabox_amd.config import get_config
from metabox_amd.environment import BatchedPBO_Env
from metabox_amd.optimizer import LDE_Optimizer
from metabox_amd.utils import construct_problem_set

ap = argparse.ArgumentParser()
ap.add_argument('--pop', default='50,100')
ap.add_argument('--steps', type=int, default=20)
ap.add_argument('--instances', type=int, default=16384)
ap.add_argument('--route', default='step')
ap.add_argument('--gens-per-launch', type=int, default=10)
ap.add_argument('--functions', default='')
ap.add_argument('--suite', default='bbob-noisy')
ap.add_argument('--dim', type=int, default=30)
args = ap.parse_args()
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for np_lde in [int(x) for x in args.pop.split(',')]:
    cfg = get_config(['--problem', args.suite, '--dim', str(args.dim), '--device', 'cuda'])
    cfg.agent_save_dir = None
    if np_lde != 50:
        cfg.NP_override = np_lde
    torch.manual_seed(0)
    agent = LDE_Agent(cfg)
    if np_lde == 50:
        agent.load_exported_weights(np.load(os.path.join(ROOT, 'metabox_amd', 'agent_model', 'lde_bbob_easy.npz')))
    agent.to('cuda')
    opt = LDE_Optimizer(cfg)
    tr, te = construct_problem_set(cfg)
    ps = sorted(tr.data + te.data, key=lambda p: p.func_id)
    if args.functions:
        want = [int(x) for x in args.functions.split(',')]
        ps = [p for p in ps if p.func_id in want]
    B = args.instances
    env = BatchedPBO_Env(ps, opt, np.arange(B) % len(ps), np.arange(B, dtype=np.uint64) + 1, early_stop=False)
    st = {'s': env.reset(), 'h': torch.zeros(1, B, 50, device='cuda'), 'c': torch.zeros(1, B, 50, device='cuda')}
    net = agent.net
    with torch.no_grad():
        if args.route == 'resident':
            def run(n):
                g = 0
                while g < n:
                    k = min(args.gens_per_launch, n - g)
                    env.batch.lde_rollout(net.packed_weights(), net.lstm.hidden_size, st['h'], st['c'], k)
                    g += k
        else:
            def run(n):
                for _ in range(n):
                    st['s'], st['h'], st['c'] = agent.policy_step(env, st['s'], st['h'], st['c'])
        run(3)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        run(args.steps)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / args.steps
    print(json.dumps({'suite': args.suite, 'dim': args.dim, 'resident': bool(env.batch.lde_rollout_is_resident()), 'pop': np_lde, 'route': args.route, 'ms_per_generation': dt * 1e3, 'launch_info': env.batch.launch_info()}), flush=True)
    env.close()
