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
F, X, G, FID = [], [], [], []
for p in ps:
    o = oracle.RlepsoOracle(p.desc(), p.bias, cfg, seed=1234 + p.func_id)
    s = o.reset()
    done = False; g = 0
    while not done:
        if g % 6 in (0, 1):     # consecutive pairs: prev exemplar available for the second
            st = oracle.split_rlepso_state(o.state(), NP, D, 50)
            F.append(st['pbest'].copy()); X.append(st['pbpos'].copy().reshape(NP, D)); G.append(g); FID.append(p.func_id)
        k = min(int(round(s * MAXFES)), table.shape[0] - 1)
        a = np.clip(mu[k] + sigma[k] * rng.standard_normal(adim, dtype=np.float32), 0, 1)
        s, _, done = o.step(a)
        g += 1
    print(p.func_id, g, flush=True)
np.savez('/tmp/fdr_states.npz', F=np.array(F), X=np.array(X), G=np.array(G), FID=np.array(FID))
