"""Wall-cycle split of k_dq_step per workgroup (instrumented build: -DMBX_PHASE_TIMING; MBX_LIB=build/libmbx_dqphase64.so python tools/exp/dq_phases.py)."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from metabox_amd.config import get_config
from metabox_amd.agent import DE_DDQN_Agent
from metabox_amd.optimizer import DE_DDQN_Optimizer
from metabox_amd.environment import BatchedPBO_Env
from metabox_amd.utils import construct_problem_set
cfg = get_config(['--problem', 'protein', '--device', 'cuda']); cfg.agent_save_dir = None
torch.manual_seed(0)
agent = DE_DDQN_Agent(cfg).to('cuda'); opt = DE_DDQN_Optimizer(cfg)
tr, te = construct_problem_set(cfg); ps = (tr + te).data[:35]
B = 35 * 64
env = BatchedPBO_Env(ps, opt, np.repeat(np.arange(35), 64), np.arange(B, dtype=np.uint64) + 1)
env.reset()
packed = agent.packed_weights()
for _ in range(120): env.step(env.batch.ddqn_qnet(packed))          # past the first sweep: the OM_W window is full
torch.cuda.synchronize()
ph = (C.c_ulonglong * 16)()
try: dbg = env.batch.lib.mbx_debug_phase_cycles            # instrumented builds only
except AttributeError: dbg = lambda *a: 0
dbg(ph, 16, 1)
n = 100
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a = env.batch.ddqn_qnet(packed)
e0.record()
for _ in range(n): env.step(a)
e1.record(); torch.cuda.synchronize()
dbg(ph, 16, 1)
v = np.array(list(ph), dtype=np.float64)
names = ['staging', 'mutation', 'evaluation', 'median', 'window', 'bookkeeping', 'selection', 'features', 'write-back']
print(json.dumps({'lib': os.path.basename(os.environ.get('MBX_LIB', 'libmbx.so')), 'k_dq_step_us': e0.elapsed_time(e1) / n * 1e3,
                  'kcycles_per_block': {k: round(x / n / B / 1e3, 2) for k, x in zip(names, v)}, 'total_kcycles': round(v.sum() / n / B / 1e3, 1)}))
