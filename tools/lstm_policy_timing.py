import os, sys, json, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from metabox_amd._abi import ALGO_LDE
from metabox_amd.suite import Batch, Suite
from metabox_amd.agent.lde_agent import LDE_Agent
from metabox_amd.config import get_config
from metabox_amd.utils import construct_problem_set
for np_ in (50, 100):
    cfg = get_config(['--problem', 'bbob-noisy', '--dim', '30', '--device', 'cuda']); cfg.agent_save_dir = None
    if np_ != 50: cfg.NP_override = np_
    torch.manual_seed(0); agent = LDE_Agent(cfg).to('cuda'); net = agent.net
    tr, te = construct_problem_set(cfg); ps = sorted(tr.data + te.data, key=lambda p: p.func_id)
    B = 16384
    b = Batch(Suite(ps), ALGO_LDE, np.arange(B) % len(ps), np.arange(B, dtype=np.uint64) + 1, np_, 60000, 1200, 50)
    b.reset()
    h = torch.zeros(1, B, 50, device='cuda'); c = torch.zeros(1, B, 50, device='cuda')
    w = net.packed_weights()
    for _ in range(5): b.lde_policy(w, 50, h, c)
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): b.lde_policy(w, 50, h, c)
    e1.record(); torch.cuda.synchronize()
    print(json.dumps({'np': np_, 'us_per_policy_launch': e0.elapsed_time(e1) / 50 * 1e3}))
    b.close()
