"""Config 4 (DE-DDQN on protein docking, one GPU's share: 35 problems x 64 runs = 2240 instances): the whole batch on one stream against S sub-batches on S streams
(instances are independent: sub-batch s's Q-network launch overlaps the others' step kernels).   python tools/exp/dq_two_streams.py"""
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
agent = DE_DDQN_Agent(cfg).to('cuda')
NPROB = int(os.environ.get('NPROB', '35'))
tr, te = construct_problem_set(cfg); ps = (tr + te).data[:NPROB]
B = NPROB * 64
pidx = np.repeat(np.arange(NPROB), 64); seeds = np.arange(B, dtype=np.uint64) + 1
packed = agent.packed_weights()
STEPS = 300

def run(S, interleave, delay_us=0):
    order = np.arange(B)
    if interleave:                                      # every sub-batch gets every problem (cost balance)
        order = np.concatenate([np.arange(B)[s::S] for s in range(S)])
    cuts = [B * s // S for s in range(S + 1)]
    streams = [torch.cuda.Stream() for _ in range(S)]
    envs = []
    for s in range(S):
        sel = order[cuts[s]:cuts[s + 1]]
        with torch.cuda.stream(streams[s]):
            e = BatchedPBO_Env(ps, DE_DDQN_Optimizer(cfg), pidx[sel], seeds[sel]); e.reset(); envs.append(e)
    torch.cuda.synchronize()
    def go(n):
        for _ in range(n):
            for s in range(S):
                with torch.cuda.stream(streams[s]):
                    envs[s].step(envs[s].batch.ddqn_qnet(packed))
    go(20); torch.cuda.synchronize()
    if delay_us:                                        # phase offset between the streams: stream s starts s x delay_us later
        for s_ in range(1, S):
            with torch.cuda.stream(streams[s_]):
                torch.cuda._sleep(int(delay_us * s_ * 2100))
    t0 = time.perf_counter(); go(STEPS); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / STEPS
    t1 = time.perf_counter(); 
    # host-side issue cost alone: the same loop is asynchronous, so time the issue of another STEPS steps before synchronising
    go(STEPS); t_issue = (time.perf_counter() - t1) / STEPS; torch.cuda.synchronize()
    cost = float(sum(e.results()['cost'][:, -1].sum().item() for e in envs))
    for e in envs: e.close()
    return {'instances': B, 'streams': S, 'delay_us': delay_us, 'interleaved': bool(interleave), 'ms_per_step': round(dt * 1e3, 4), 'host_issue_ms_per_step': round(t_issue * 1e3, 4), 'cost_sum': cost}

MODES = ((1, False, 0),) if NPROB != 35 else ((1, False, 0), (2, False, 0), (2, True, 0), (2, True, 10), (2, True, 20), (2, True, 30), (2, True, 40), (3, True, 20), (1, False, 0))
for rep in range(2):
    for S, il, dl in MODES:
        print(json.dumps(run(S, il, dl)), flush=True)
