"""The FDR exemplar on adversarial swarms.

Reference (src/optimizer/rlepso_optimizer.py:96-109): target_index = np.argmin((f_j - f_i) / (|p_jd - p_id| + 1e-5), axis=j) -- the quotients are ROUNDED and the
first minimal one in particle-index order wins.  The oracle evaluates exactly that (oracle/mbx_oracle.c, division form).  The HIP kernels compare two candidates
by cross-multiplication, a_j b* < a* b_j (csrc/mbx_rlepso.hpp fdr_exact), which orders the exact ratios the same way but can resolve differently when two
NON-identical candidates have quotients within an ulp or two of each other.  This file builds swarms in which, for one query particle and every dimension,
the two best candidates have quotients 0, 1, 2, ... ulp apart, hands the same state block + replay tape to the oracle and to the kernel, decodes the chosen
exemplar from the velocity the step produces, and counts:
  * CPU:  the oracle agrees with the literal numpy formula of the reference on every (particle, dimension)            -> asserted, 100 %
  * GPU:  the kernel agrees with it whenever the two quotients are >= 4 ulp apart                                      -> asserted, 100 %
          and the measured disagreement rate at 0 / 1 / 2-3 ulp                                                          -> printed
  * GPU:  a batch created with MBX_FDR_EXACT=1 (near-ties flagged inside the scan and redone with the reference's divisions) agrees on EVERY pair -> asserted
  * GPU:  on natural swarms (Philox episodes of four functions) kernel and oracle pick the same exemplar everywhere    -> asserted
"""
import numpy as np
import pytest

from helpers import problems
from oracle import oracle

NP, D, NLOG, MAXFES, LOGI = 100, 10, 50, 20000, 400
# groups read actions[5 g : 5 g + 7] (rlepso_optimizer.py:119): [c_mutation, w, scale, c1, c2, c3, c4] -> period-5 pattern with c_mutation = 0, w = 0.1,
# c1 = c3 = c4 = 0 and only the FDR term alive: new velocity = c2 * u_fdr * (pbest_pos[target, d] - pbest_pos[i, d])
ACTION = np.tile(np.array([0, 0, 1, 0, 1], np.float32), 7)
U_FDR = 0.5


def c2_of_action():
    a = ACTION[:7]
    den = np.float32(a[3] + a[4]); den = np.float32(den + a[5]); den = np.float32(den + a[6]); den = np.float32(den + np.float32(1e-5))
    scale = np.float32(np.float32(1.) / den); scale = np.float32(scale * a[2]); scale = np.float32(scale * np.float32(8.))
    return float(np.float32(scale * a[4]))


def ulps_apart(x, y):
    """|distance in representable doubles| between two same-signed float64 values."""
    return int(abs(int(np.float64(x).view(np.int64)) - int(np.float64(y).view(np.int64))))


def quotient(f, P, i, m, d):
    return (f[m] - f[i]) / (np.abs(P[m, d] - P[i, d]) + 1e-5)


def craft_swarm(rs, target_ulps):
    """pbest costs f [NP], pbest positions P [NP, D], query particle q and the pair (j, k) whose quotients towards q are `target_ulps[d]` ulp apart in
    dimension d (as close as the +-24-ulp neighbourhood of the two coordinates allows); every other candidate's quotient is far weaker."""
    q, j, k = rs.choice(NP, 3, replace=False)
    f = 1000. - rs.uniform(1., 40., NP)
    f[q] = 1000.
    f[j] = 1000. - 100. * (1 + rs.uniform())
    f[k] = 1000. - 50. * (1 + rs.uniform())
    P = rs.uniform(-5, 5, (NP, D))
    P[q] = rs.uniform(-3, 3, D)
    for m in range(NP):                                   # everybody else stays >= 1 away from the query particle in every dimension
        if m in (q, j, k):
            continue
        bad = np.abs(P[m] - P[q]) < 1.
        P[m, bad] = P[q, bad] + np.where(rs.uniform(size=bad.sum()) < 0.5, -1, 1) * rs.uniform(1., 1.9, bad.sum())
    got = np.zeros(D, int)
    steps = np.arange(-24, 25)
    for d in range(D):
        bj = rs.uniform(0.01, 0.1)
        sj, sk = rs.choice([-1., 1.], 2)
        xj0 = P[q, d] + sj * bj
        aj, ak = f[j] - f[q], f[k] - f[q]
        bj_act = np.abs(xj0 - P[q, d]) + 1e-5
        xk0 = P[q, d] + sk * (ak * bj_act / aj - 1e-5)
        # neighbourhoods of both coordinates, all combinations: pick the one whose quotient distance is closest to the target
        xj = xj0 + steps * np.spacing(xj0)
        xk = xk0 + steps * np.spacing(xk0)
        qj = aj / (np.abs(xj - P[q, d]) + 1e-5)
        qk = ak / (np.abs(xk - P[q, d]) + 1e-5)
        dist = np.abs(qj.view(np.int64)[:, None] - qk.view(np.int64)[None, :])
        a, b = np.unravel_index(np.argmin(np.abs(dist - target_ulps[d])), dist.shape)
        P[j, d], P[k, d] = xj[a], xk[b]
        got[d] = ulps_apart(quotient(f, P, q, j, d), quotient(f, P, q, k, d))
    return f, P, (q, j, k), got


def reference_targets(f, P):
    """rlepso_optimizer.py:98-102, literally."""
    distance_per_dim = np.abs(P[None, :, :].repeat(NP, axis=0) - P[:, None, :].repeat(NP, axis=1))
    fitness_delta = f[None, :].repeat(NP, axis=0) - f[:, None].repeat(NP, axis=1)
    fdr = (fitness_delta[:, :, None]) / (distance_per_dim + 1e-5)
    return np.argmin(fdr, axis=1)


def state_block(template, f, P):
    """A reset state block with the crafted swarm as its pbest table, zero velocities and consistent gbest fields."""
    st = template.copy()
    sp = oracle.split_rlepso_state(st, NP, D, NLOG)          # views into st
    sp['vel'][:] = 0.
    sp['pbpos'][:] = P.ravel()
    sp['pbest'][:] = f
    sp['ccost'][:] = f
    sp['pni'][:] = 0.
    g = int(np.argmin(f))
    sp['gbpos'][:] = P[g]
    sp['scalars'][oracle.SC_GBEST] = f[g]
    sp['scalars'][oracle.SC_GBEST_IDX] = g
    return st


def replay_tape():
    """One update() worth of numpy draws (layout: oracle.NumpyTapeFeeder.step_tape): no CLPSO exemplar (u > pci), FDR weight 0.5, no re-initialisation."""
    t = np.zeros(9 * NP + 6 * NP * D)
    t[0:2 * NP] = 0.25                                       # rand1, rand2 (their terms carry zero coefficients)
    o = 2 * NP
    t[o:o + NP * D] = 0.999                                  # CLPSO uniforms > every pci_i: the particle's own pbest, no tournament
    o += NP * D + 2 * NP * D
    t[o:o + NP * D] = U_FDR
    o += NP * D + 3 * NP
    t[o:o + NP] = 0.999                                      # re-initialisation test: c_mutation = 0 anyway
    return t


def decode_agreement(vel_after, f, P, targets):
    """[NP, D] bool: the observed new velocity equals the one the reference's exemplar produces (bit for bit)."""
    c2 = c2_of_action()
    ii, dd = np.indices((NP, D))
    want = np.clip(c2 * (U_FDR * (P[targets, dd] - P)), -1., 1.)
    return vel_after.reshape(NP, D) == want


def crafted_cases(n, seed=0):
    rs = np.random.RandomState(seed)
    menu = np.array([0, 0, 1, 1, 2, 2, 3, 4, 8, 64])
    return [craft_swarm(rs, rs.permutation(menu)) for _ in range(n)]


def test_oracle_fdr_is_the_reference_formula_on_adversarial_swarms():
    p = problems('bbob', D)[1]
    cfg = oracle.make_cfg(1, NP, D, MAXFES, LOGI, NLOG)
    tape = replay_tape()
    hist = {}
    for f, P, (q, j, k), ulps in crafted_cases(60):
        o = oracle.RlepsoOracle(p.desc(), p.bias, cfg, seed=1)
        o.reset()
        o.set_state(state_block(o.state(), f, P))
        o.step(ACTION, tape)
        vel = oracle.split_rlepso_state(o.state(), NP, D, NLOG)['vel']
        tg = reference_targets(f, P)
        assert set(tg[q]) <= {j, k}, 'the crafted pair must decide the query particle'
        agree = decode_agreement(vel, f, P, tg)
        assert agree.all(), np.argwhere(~agree)[:4]
        for d in range(D):
            hist[min(ulps[d], 4)] = hist.get(min(ulps[d], 4), 0) + 1
    assert hist.get(0, 0) >= 60 and hist.get(1, 0) >= 60 and hist.get(2, 0) >= 30, hist      # the construction really produces 0 / 1 / 2-ulp pairs


@pytest.mark.gpu
@pytest.mark.parametrize('mode', ['default', 'exact'])
def test_hip_fdr_on_adversarial_swarms_measured_disagreement(mode, monkeypatch):
    import torch
    if mode == 'exact':
        monkeypatch.setenv('MBX_FDR_EXACT', '1')               # read when the batch is created
    from metabox_amd.suite import Batch, Suite
    from metabox_amd._abi import ALGO_RLEPSO
    p = problems('bbob', D)[1]
    cases = crafted_cases(512)
    B = len(cases)
    batch = Batch(Suite([p]), ALGO_RLEPSO, np.zeros(B, int), np.arange(B), NP, MAXFES, LOGI, NLOG)
    assert batch.launch_info()['fixed_geometry'] == (0 if mode == 'exact' else 1) and batch.rollout_is_resident() == (mode != 'exact')
    batch.reset()
    torch.cuda.synchronize()
    template = batch.read_state(0)
    for b, (f, P, _, _) in enumerate(cases):
        batch.write_state(b, state_block(template, f, P))
    batch.set_tape(torch.from_numpy(np.tile(replay_tape(), (B, 1))).cuda())
    batch.step(torch.from_numpy(np.tile(ACTION, (B, 1))).cuda())
    torch.cuda.synchronize()
    buckets = {0: [0, 0], 1: [0, 0], 2: [0, 0], 3: [0, 0], 4: [0, 0]}           # ulp distance (4 = four or more) -> [pairs, disagreements]
    others = [0, 0]
    for b, (f, P, (q, j, k), ulps) in enumerate(cases):
        vel = oracle.split_rlepso_state(batch.read_state(b), NP, D, NLOG)['vel']
        agree = decode_agreement(vel, f, P, reference_targets(f, P))
        for d in range(D):
            u = min(int(ulps[d]), 4)
            buckets[u][0] += 1
            buckets[u][1] += int(not agree[q, d])
        rest = np.delete(agree, q, axis=0)
        others[0] += rest.size
        others[1] += int((~rest).sum())
    batch.close()
    print(f'FDR exemplar, kernel ({"MBX_FDR_EXACT=1: near-ties redone with divisions" if mode == "exact" else "cross-multiplied compare"}) vs reference formula '
          f'(rounded quotients, np.argmin) on crafted near-ties:')
    for u in sorted(buckets):
        n, bad = buckets[u]
        print(f'  quotients {u}{"+" if u == 4 else ""} ulp apart: {bad} / {n} pairs resolved differently ({100. * bad / max(n, 1):.1f} %)')
    print(f'  all other (particle, dimension) items of the same swarms: {others[1]} / {others[0]}')
    assert buckets[4][1] == 0 and others[1] == 0, (buckets, others)
    assert buckets[0][0] >= 500 and buckets[1][0] >= 500
    if mode == 'exact':
        assert all(v[1] == 0 for v in buckets.values()), buckets
    else:
        assert buckets[2][1] == 0 and buckets[3][1] == 0, buckets          # measured: only quotients that round together or to neighbours resolve differently


@pytest.mark.gpu
def test_hip_fdr_equals_the_oracle_on_natural_swarms():
    """Whole Philox episodes: kernel and oracle step the same instances; at every generation the kernel's velocities must be the ones the oracle's
    (division-form) exemplars produce -- bitwise equal pbest tables in, so any differently resolved exemplar shows as a velocity difference far above rounding."""
    import torch
    from metabox_amd.suite import Batch, Suite
    from metabox_amd._abi import ALGO_RLEPSO
    fids = (1, 8, 15, 21)
    ps = [problems('bbob', D)[k] for k in fids]
    B, G = 64, 60
    rs = np.random.RandomState(3)
    seeds = np.arange(B, dtype=np.uint64) * 7919 + 5
    pidx = np.arange(B) % len(ps)
    batch = Batch(Suite(ps), ALGO_RLEPSO, pidx, seeds, NP, MAXFES, LOGI, NLOG)
    batch.reset()
    cfg = oracle.make_cfg(1, NP, D, MAXFES, LOGI, NLOG)
    acts = rs.uniform(0, 1, (G, B, 35)).astype(np.float32)
    items = mismatched = 0
    for g in range(G):
        torch.cuda.synchronize()
        blocks = [batch.read_state(b) for b in range(B)]
        batch.step(torch.from_numpy(acts[g]).cuda())
        torch.cuda.synchronize()
        for b in range(0, B, 4):
            o = oracle.RlepsoOracle(ps[pidx[b]].desc(), ps[pidx[b]].bias, cfg, seed=int(seeds[b]))
            o.reset()
            o.set_state(blocks[b])                         # the kernel's own state before the step: bitwise equal pbest tables on both sides
            o.step(acts[g, b])
            want = oracle.split_rlepso_state(o.state(), NP, D, NLOG)
            got = oracle.split_rlepso_state(batch.read_state(b), NP, D, NLOG)
            if want['scalars'][oracle.SC_REINIT] or got['scalars'][oracle.SC_REINIT]:
                continue                                    # re-initialised particles carry fresh velocities
            diff = np.abs(got['vel'] - want['vel'])
            items += diff.size
            mismatched += int((diff > 1e-9).sum())
    batch.close()
    print(f'natural swarms: {mismatched} of {items} (particle, dimension) items moved with a different exemplar')
    assert items > 500_000 and mismatched == 0
