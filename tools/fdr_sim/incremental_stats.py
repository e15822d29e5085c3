import sys, os, numpy as np
sys.path.insert(0, '/root/repo')
import torch
torch.set_num_threads(1)
import bench
from metabox_amd.agent.rlepso_agent import ActorTable
from metabox_amd.problem.bbob import BBOB_Dataset
from oracle import oracle
NP, D, MAXFES = 100, 10, 20000
config = bench.make_config(); config.device = 'cpu'
agent = bench.load_agent(config, 'cpu')
table = ActorTable(agent.actor, MAXFES, NP, 'cpu').table.numpy()
adim = table.shape[1] // 2
mu, sigma = table[:, :adim], table[:, adim:]
tr, te = BBOB_Dataset.get_datasets('bbob', D, 5.0)
ps = sorted(tr.data + te.data, key=lambda p: p.func_id)
cfg = oracle.make_cfg(1, NP, D, MAXFES, MAXFES // 50, 50)
rng = np.random.default_rng(1)
tot_full = 0; tot_inc = 0; gens = 0
bygen_full = np.zeros(200); bygen_inc = np.zeros(200); bygen_n = np.zeros(200); bygen_p = np.zeros(200)
for p in ps:
    o = oracle.RlepsoOracle(p.desc(), p.bias, cfg, seed=1234 + p.func_id)
    s = o.reset()
    st = oracle.split_rlepso_state(o.state(), NP, D, 50)
    prev_pb = st['pbest'].copy(); prev_pos = st['pbpos'].copy().reshape(NP, D)
    prev_ex = None
    done = False; g = 0
    while not done:
        # FDR on current pbest (what the step will compute)
        f = prev_pb; x = prev_pos
        order = np.lexsort((np.arange(NP), f))
        rank = np.empty(NP, int); rank[order] = np.arange(NP)
        nless = np.array([(f < f[i]).sum() for i in range(NP)])
        full = nless.sum() * D
        # exemplar
        dist = np.abs(x[None, :, :] - x[:, None, :]) + 1e-5
        fd = (f[None, :] - f[:, None])[:, :, None] / dist
        ex = np.argmin(fd, axis=1)   # [NP, D]
        if prev_ex is None:
            inc = full
        else:
            C = changed
            nC = C.sum()
            inc = 0
            for i in range(NP):
                if C[i]:
                    inc += nless[i] * D
                else:
                    cb = (C & (f < f[i])).sum()
                    for d in range(D):
                        if C[prev_ex[i, d]]:
                            inc += nless[i]
                        else:
                            inc += cb
        k = min(int(round(s * MAXFES)), table.shape[0] - 1)
        a = np.clip(mu[k] + sigma[k] * rng.standard_normal(adim, dtype=np.float32), 0, 1)
        s, _, done = o.step(a)
        st = oracle.split_rlepso_state(o.state(), NP, D, 50)
        new_pb = st['pbest'].copy(); new_pos = st['pbpos'].copy().reshape(NP, D)
        changed = (new_pb != prev_pb) | (new_pos != prev_pos).any(axis=1)
        prev_ex = ex
        prev_pb, prev_pos = new_pb, new_pos
        bygen_full[g] += full; bygen_inc[g] += inc; bygen_n[g] += 1; bygen_p[g] += changed.mean()
        tot_full += full; tot_inc += inc; gens += 1; g += 1
    print(p.func_id, g, 'full/gen', tot_full / gens, 'inc/gen', tot_inc / gens, flush=True)
print('TOTAL full', tot_full / gens, 'inc', tot_inc / gens, 'ratio', tot_inc / tot_full)
for g in range(0, 200, 10):
    if bygen_n[g]:
        print(g, bygen_n[g], bygen_full[g] / bygen_n[g], bygen_inc[g] / bygen_n[g], bygen_p[g] / bygen_n[g])
