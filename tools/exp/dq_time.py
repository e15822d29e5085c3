"""Config 4's share (2240 instances = 35 proteins x 64 runs), per env step: mbx_ddqn_qnet + mbx_step against mbx_ddqn_rollout (100 steps per launch).  MBX_LIB=... python tools/exp/dq_time.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from metabox_amd.agent import DE_DDQN_Agent
from metabox_amd.config import get_config
from metabox_amd.environment import BatchedPBO_Env
from metabox_amd.optimizer import DE_DDQN_Optimizer
from metabox_amd.utils import construct_problem_set
cfg = get_config(['--problem', 'protein', '--device', 'cuda']); cfg.agent_save_dir = None
torch.manual_seed(0)
agent = DE_DDQN_Agent(cfg).to('cuda'); opt = DE_DDQN_Optimizer(cfg)
tr, te = construct_problem_set(cfg); ps = (tr + te).data[:35]; B = 35 * 64
env = BatchedPBO_Env(ps, opt, np.repeat(np.arange(35), 64), np.arange(B, dtype=np.uint64) + 1, early_stop=False)
env.reset(); packed = agent.packed_weights()
def t(fn, n):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(n); torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
def a(n):
    for _ in range(n): env.step(env.batch.ddqn_qnet(packed))
def r(n): env.batch.ddqn_rollout(packed, n)
a(5); r(5)
print(json.dumps({'lib': os.path.basename(os.environ.get('MBX_LIB', 'libmbx.so')), 'per_step_route_ms': round(sorted(t(a, 100) for _ in range(3))[1] * 1e3, 5),
                  'resident_ms': round(sorted(t(r, 100) for _ in range(3))[1] * 1e3, 5)}))
