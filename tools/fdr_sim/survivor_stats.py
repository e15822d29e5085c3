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
acc = {k: [] for k in ('mean_prev', 'wmax_prev', 'mean_first8', 'wmax_first8', 'mean_both', 'wmax_both', 'same')}
fids = [int(a) for a in sys.argv[1:]] or [2, 3, 8, 15, 16, 21]
for p in ps:
    if p.func_id not in fids: continue
    o = oracle.RlepsoOracle(p.desc(), p.bias, cfg, seed=1234 + p.func_id)
    s = o.reset()
    st = oracle.split_rlepso_state(o.state(), NP, D, 50)
    prev_ex = None
    done = False; g = 0
    while not done:
        f = st['pbest'].copy(); x = st['pbpos'].copy().reshape(NP, D)
        order = np.lexsort((np.arange(NP), f))
        dist = np.abs(x[None, :, :] - x[:, None, :]) + 1e-5       # [i, j, d]
        A = (f[:, None] - f[None, :])                              # A[i, j] = f_i - f_j > 0 for better j
        ratio = A[:, :, None] / dist                               # maximise
        better = (f[None, :] < f[:, None])                         # [i, j]
        ratio_m = np.where(better[:, :, None], ratio, -np.inf)
        ex = np.argmax(ratio_m, axis=1)                            # [i, d]
        rmax = np.max(ratio_m, axis=1)
        if prev_ex is not None and g % 5 == 0:
            ii = np.arange(NP)[:, None]; dd = np.arange(D)[None, :]
            r_prev = ratio_m[ii, prev_ex, dd]                      # -inf if not better any more
            # first 8 by cost order
            r_f8 = np.max(ratio_m[:, order[:8], :], axis=1)
            r_prev_or0 = np.maximum(r_prev, ratio_m[:, order[0], :])
            r_both = np.maximum(r_prev, r_f8)
            def surv(r_lo):
                sv = (ratio_m >= (r_lo * (1 - 1e-12))[:, None, :]) & better[:, :, None]
                cnt = sv.sum(axis=1).astype(float)                 # [i, d]
                cnt = cnt[order]                                   # rank-major
                flat = cnt.reshape(-1)
                # wave = 128 consecutive scans
                wm = [flat[k:k + 128].max() for k in range(0, flat.size, 128)]
                return flat.mean(), np.mean(wm)
            m, w = surv(r_prev_or0); acc['mean_prev'].append(m); acc['wmax_prev'].append(w)
            m, w = surv(r_f8); acc['mean_first8'].append(m); acc['wmax_first8'].append(w)
            m, w = surv(r_both); acc['mean_both'].append(m); acc['wmax_both'].append(w)
            acc['same'].append((ex == prev_ex).mean())
        k = min(int(round(s * MAXFES)), table.shape[0] - 1)
        a = np.clip(mu[k] + sigma[k] * rng.standard_normal(adim, dtype=np.float32), 0, 1)
        s, _, done = o.step(a)
        st = oracle.split_rlepso_state(o.state(), NP, D, 50)
        prev_ex = ex
        g += 1
    print(p.func_id, g, {k: round(float(np.mean(v)), 2) for k, v in acc.items()}, flush=True)
