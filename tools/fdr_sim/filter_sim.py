import numpy as np
d = np.load('/tmp/fdr_states.npz')
F, X, G, FID = d['F'], d['X'], d['G'], d['FID']
NP, D = 100, 10
C_PAIR_OLD = 36.0    # cycles per (cand, dim) pair, current scan
C_FILT = 14.0        # sub, fma, cmpx (4 each) + masked or (2)
C_RES = 50.0         # per survivor resolve (per q) ~12 instr
C_DIV = 60.0
C_EX = 36.0          # exact eval of one warm-start candidate

def analyse(f, x, prev_ex_idx):
    order = np.lexsort((np.arange(NP), f))
    rank = np.empty(NP, int); rank[order] = np.arange(NP)
    fs = f[order]; xs = x[order]                       # rank-major
    nless = np.array([(f < fs[r]).sum() for r in range(NP)])
    dist = np.abs(xs[None, :, :] - xs[:, None, :]) + 1e-5   # [r, k, d]
    A = fs[:, None] - fs[None, :]
    ratio = A[:, :, None] / dist
    valid = (np.arange(NP)[None, :] < nless[:, None])
    ratio = np.where(valid[:, :, None], ratio, -np.inf)
    ex = np.argmax(ratio, axis=1)                      # rank of exemplar [r, d]
    return order, rank, nless, ratio, ex

def wave_lanes():
    # returns list of waves; each wave = list of (rk, hd) per lane
    NI = NP * 5
    waves = []
    for base, pas in ((0, 0), (256, 1)):
        lim = min(base + 256, NI)
        for w in range(4):
            lanes = []
            for l in range(64):
                tid = w * 64 + l
                ps = lim - 1 - tid if pas else base + tid
                if base <= ps < lim:
                    lanes.append((ps // 5, ps % 5))
            waves.append(lanes)
    return waves
WAVES = wave_lanes()

tot = {}
def add(k, v): tot[k] = tot.get(k, 0.) + v
n = 0
for s in range(len(G)):
    if G[s] % 6 != 1: continue
    if FID[s - 1] != FID[s] or G[s - 1] != G[s] - 1: continue
    o0, r0, nl0, ra0, ex0 = analyse(F[s - 1], X[s - 1], None)
    order, rank, nless, ratio, ex = analyse(F[s], X[s], None)
    # previous exemplar as particle index for particle i, dim d
    prev_ex_particle = o0[ex0]                          # [rank0, d] -> particle index
    # map to current rank-major: particle i = order[r]; its prev rank r0[i]
    pe = prev_ex_particle[r0[order]]                    # [r, d] particle index
    pe_rank = rank[pe]                                  # current rank of prev exemplar
    rr = np.arange(NP)[:, None]; dd = np.arange(D)[None, :]
    r_prev = ratio[rr, pe_rank, dd]                     # -inf when no longer strictly better
    r_0 = ratio[:, 0, :]
    surv_cnt = {}
    for name, r_lo, setup in (
        ('prev+r0', np.maximum(r_prev, r_0), 2 * C_EX + C_DIV),
        ('prev+f4', np.maximum(r_prev, ratio[:, :4, :].max(axis=1)), 5 * C_EX + C_DIV),
        ('prev+f8', np.maximum(r_prev, ratio[:, :8, :].max(axis=1)), 9 * C_EX + C_DIV),
        ('prev+f16', np.maximum(r_prev, ratio[:, :16, :].max(axis=1)), 17 * C_EX + C_DIV),
    ):
        sv = (ratio >= (r_lo * (1 - 1e-12))[:, None, :]) & np.isfinite(ratio)
        cnt = sv.sum(axis=1)                            # [r, d]
        cyc = 0.
        for lanes in WAVES:
            if not lanes: continue
            nl = max(nless[r] for r, h in lanes)
            res = max(max(cnt[r, 2 * h], cnt[r, 2 * h + 1]) for r, h in lanes)
            cyc += 2 * setup + nl * 2 * C_FILT + res * 2 * C_RES
            add(name + '_res', res); add(name + '_mean', np.mean([cnt[r, 2*h] for r, h in lanes]))
        add(name, cyc)
    cyc = 0.
    for lanes in WAVES:
        if not lanes: continue
        nl = max(nless[r] for r, h in lanes)
        cyc += nl * 2 * C_PAIR_OLD
    add('old', cyc)
    n += 1
print('states', n)
for k, v in tot.items():
    print(k, round(v / n, 1))
