"""How often does ANY lane of a wave take a new exemplar at step k of the FDR scan (fdr_exact's unrolled loop)?  The three conditional moves of fdr_take are issued on
every step; a wave-uniform branch around them would save 3 of the 9 vector instructions of a pair whenever no lane takes.  Real pbest states (dump_states.py), the
kernel's item -> lane mapping (items in pbest-rank order, W = 2 coordinates per item, 256 threads, odd passes reversed), candidates in ascending-cost order.
   python tools/fdr_sim/dump_states.py && python tools/fdr_sim/take_stats.py"""
import numpy as np
d = np.load('/tmp/fdr_states.npz')
F, X = d['F'], d['X']
NP, D, NT, W = 100, 10, 256, 2
HD = D // W
tot_steps = tot_notake = tot_pairs_lane = 0
tot_any2 = 0
rs = np.random.RandomState(0)
for s in rs.choice(len(F), 300, replace=False):
    f, x = F[s], X[s]
    order = np.lexsort((np.arange(NP), f))                 # (cost, index)
    fs, xs = f[order], x[order]
    nless = np.array([np.sum(fs < fs[r]) for r in range(NP)])
    # per (rank, coordinate): the step indices k (1 .. nless-1) at which the running best changes
    take = np.zeros((NP, D, NP), bool)
    for r in range(NP):
        n = nless[r]
        if n <= 1:
            continue
        a = fs[:n] - fs[r]                                  # negative
        b = np.abs(xs[:n] - xs[r]) + 1e-5                   # [n, D]
        q = a[:, None] / b
        run = np.minimum.accumulate(q, axis=0)
        take[r, :, 1:n] = (q[1:] < run[:-1]).T
    NI = NP * HD
    for base, p in zip(range(0, NI, NT), range(99)):
        lim = min(base + NT, NI)
        items = np.arange(base, lim)
        if p & 1:
            items = items[::-1]
        lanes = np.full(NT, -1); lanes[:len(items)] = items
        for w in range(NT // 64):
            it = lanes[64 * w:64 * w + 64]; it = it[it >= 0]
            if len(it) == 0:
                continue
            rk, d0 = it // HD, 2 * (it % HD)
            trips = nless[rk].max()                          # the wave runs the longest lane's trip count
            if trips <= 1:
                continue
            t0 = take[rk, d0][:, 1:trips]; t1 = take[rk, d0 + 1][:, 1:trips]
            anyq0, anyq1 = t0.any(0), t1.any(0)              # per step: does any lane take for coordinate q
            tot_steps += 2 * (trips - 1)
            tot_notake += int((~anyq0).sum() + (~anyq1).sum())
            tot_any2 += int((~(anyq0 | anyq1)).sum())        # a branch around BOTH coordinates' moves
            tot_pairs_lane += int((nless[rk] - 1).clip(0).sum()) * 2
print(f'wave-steps (per coordinate): {tot_steps}; no lane takes: {tot_notake} ({100 * tot_notake / tot_steps:.1f} %); no lane takes for either coordinate of the step: '
      f'{100 * 2 * tot_any2 / tot_steps:.1f} % of the steps')
