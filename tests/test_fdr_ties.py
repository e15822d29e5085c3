"""The FDR exemplar on adversarial swarms.

Reference (src/optimizer/rlepso_optimizer.py:96-109): target_index = np.argmin((f_j - f_i) / (|p_jd - p_id| + 1e-5), axis=j) -- the quotients are ROUNDED and the
first minimal one in particle-index order wins.  The oracle evaluates exactly that (oracle/mbx_oracle.c, division form).  The HIP kernels scan by
cross-multiplication, a_j b* < a* b_j (csrc/mbx_rlepso.hpp fdr_exact), which orders the exact ratios the same way but can resolve differently when two
NON-identical candidates have quotients within an ulp or two of each other; by DEFAULT every kernel therefore flags each comparison that comes within 2^-49 of
a tie and settles the flagged items with rounded quotients (fdr_verify).  This file builds swarms in which, for one query particle and every dimension, the two
best candidates have quotients 0, 1, 2, ... ulp apart, hands the same state block to the reference formula and to the kernel, decodes the chosen exemplar from
the velocity the step produces, and checks:
  * CPU:  the oracle agrees with the literal numpy formula of the reference on every (particle, dimension)                              -> asserted, 100 %
  * GPU:  the kernels agree with it on EVERY item, whatever the distance -- route in {one launch per generation (replay tape), resident rollout (Philox)} x
          geometry in {NP 100 / D 10 (headline), NP 128 / D 40 (config 5)}                                                                -> asserted, 0 in every bucket
  * GPU:  a batch created with MBX_F_FDR_FAST (the cross-multiplied order alone) disagrees only where the quotients are <= 1 ulp apart    -> asserted; rate printed
  * GPU:  two batches with different flags coexist in one process (the option travels in mbx_algo_cfg.flags, not in the environment)
  * GPU:  on natural swarms -- wide ones and ones converged to the rounding floor of their costs, where most comparisons ARE ties -- kernel and oracle pick the
          same exemplar everywhere                                                                                                          -> asserted
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


def craft_swarm(rs, target_ulps, NP=NP, D=D, three=False):
    """pbest costs f [NP], pbest positions P [NP, D], query particle q and the pair (j, k) whose quotients towards q are `target_ulps[d]` ulp apart in
    dimension d (as close as the +-24-ulp neighbourhood of the two coordinates allows); every other candidate's quotient is far weaker.  three: a third candidate l is
    tuned against j the same way (a three-way near-tie: the scan's running best changes hands inside the band)."""
    q, j, k, l = rs.choice(NP, 4, replace=False)
    while q >= 5 * (NP // 5):                             # particles beyond the last full group move with zero coefficients (rlepso_optimizer.py:117-126): nothing to decode
        q, j, k, l = rs.choice(NP, 4, replace=False)
    f = 1000. - rs.uniform(1., 40., NP)
    f[q] = 1000.
    f[j] = 1000. - 100. * (1 + rs.uniform())
    f[k] = 1000. - 50. * (1 + rs.uniform())
    if three:
        f[l] = 1000. - 75. * (1 + rs.uniform())
    P = rs.uniform(-5, 5, (NP, D))
    P[q] = rs.uniform(-3, 3, D)
    for m in range(NP):                                   # everybody else stays >= 1 away from the query particle in every dimension
        if m in (q, j, k) or (three and m == l):
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
        if three:                                         # l against the (now fixed) j: its coordinate alone is searched, +-2000 ulp
            al = f[l] - f[q]
            xl0 = P[q, d] + rs.choice([-1., 1.]) * (al * (np.abs(P[j, d] - P[q, d]) + 1e-5) / aj - 1e-5)
            xl = xl0 + np.arange(-2000, 2001) * np.spacing(xl0)
            ql = al / (np.abs(xl - P[q, d]) + 1e-5)
            dl = np.abs(ql.view(np.int64) - np.float64(quotient(f, P, q, j, d)).view(np.int64))
            P[l, d] = xl[np.argmin(np.abs(dl - target_ulps[(d + 1) % D]))]
    return f, P, (q, j, k), got


def reference_targets(f, P):
    """rlepso_optimizer.py:98-102, literally."""
    NP = len(f)
    distance_per_dim = np.abs(P[None, :, :].repeat(NP, axis=0) - P[:, None, :].repeat(NP, axis=1))
    fitness_delta = f[None, :].repeat(NP, axis=0) - f[:, None].repeat(NP, axis=1)
    fdr = (fitness_delta[:, :, None]) / (distance_per_dim + 1e-5)
    return np.argmin(fdr, axis=1)


def state_block(template, f, P, NLOG=NLOG):
    """A reset state block with the crafted swarm as its pbest table, zero velocities and consistent gbest fields."""
    NP, D = P.shape
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


def replay_tape(NP=NP, D=D):
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


def decode_agreement(vel_after, f, P, targets, u_fdr=U_FDR):
    """[NP, D] bool: the observed new velocity equals the one the reference's exemplar produces (bit for bit).  u_fdr: the FDR weight, a scalar (replay
    tape) or the [NP, D] Philox draws of the generation (mbx_debug_rlepso_draws)."""
    NP, D = P.shape
    c2 = c2_of_action()
    ii, dd = np.indices((NP, D))
    want = np.clip(c2 * (u_fdr * (P[targets, dd] - P)), -1., 1.)
    want[5 * (NP // 5):] = 0.                             # NP = 128: particles 125-127 get w = c = 0 under the reference's NP // n_group rule
    return vel_after.reshape(NP, D) == want


def crafted_cases(n, seed=0, NP=NP, D=D, three=False):
    rs = np.random.RandomState(seed)
    menu = np.tile(np.array([0, 0, 1, 1, 2, 2, 3, 4, 8, 64]), D // 10)
    return [craft_swarm(rs, rs.permutation(menu), NP, D, three) for _ in range(n)]


def test_oracle_fdr_is_the_reference_formula_on_adversarial_swarms():
    p = problems('bbob', D)[1]
    cfg = oracle.make_cfg(1, NP, D, MAXFES, LOGI, NLOG)
    tape = replay_tape()
    hist = {}
    for f, P, (q, j, k), ulps in crafted_cases(60) + crafted_cases(30, seed=11, three=True):
        o = oracle.RlepsoOracle(p.desc(), p.bias, cfg, seed=1)
        o.reset()
        o.set_state(state_block(o.state(), f, P))
        o.step(ACTION, tape)
        vel = oracle.split_rlepso_state(o.state(), NP, D, NLOG)['vel']
        tg = reference_targets(f, P)
        assert len(set(tg[q]) - {j, k}) <= 1, 'the crafted pair (or triple) must decide the query particle'
        agree = decode_agreement(vel, f, P, tg)
        assert agree.all(), np.argwhere(~agree)[:4]
        for d in range(D):
            hist[min(ulps[d], 4)] = hist.get(min(ulps[d], 4), 0) + 1
    assert hist.get(0, 0) >= 60 and hist.get(1, 0) >= 60 and hist.get(2, 0) >= 30, hist      # the construction really produces 0 / 1 / 2-ulp pairs


def _philox_fdr_weights(seed, NP_, D_):
    """The FDR weights u_fdr [NP, D] the kernels draw for the instance with key `seed` in generation 1 of episode 0 (mbx_debug_rlepso_draws: rl_move's own code path)."""
    import ctypes as C
    import torch
    from metabox_amd import _abi
    out = torch.empty(NP_ * D_, 4, dtype=torch.float64, device='cuda')
    _abi.check(_abi.load_lib().mbx_debug_rlepso_draws(C.c_uint64(int(seed)), 1, 0, NP_, D_, C.c_void_p(out.data_ptr()), None))
    torch.cuda.synchronize()
    return out[:, 1].cpu().numpy().reshape(NP_, D_)


def _run_crafted(cases, NP_, D_, route, flags):
    """Step every crafted swarm once on the GPU -- route 'per_generation': mbx_set_tape + mbx_step (u_fdr = 0.5 from the tape); route 'resident': ONE generation of
    mbx_rlepso_rollout with an actor table whose mu is ACTION and whose sigma is 0 (Philox draws; u_fdr read back through mbx_debug_rlepso_draws) -- and return
    (buckets {ulp distance: [pairs, disagreements]}, [other items, disagreements], batch flags, launch info, resident?)."""
    import torch
    from metabox_amd.suite import Batch, Suite
    from metabox_amd._abi import ALGO_RLEPSO
    maxfes = 2000 * D_
    p = problems('bbob', D_)[1]
    B = len(cases)
    seeds = np.arange(B, dtype=np.uint64) * 7 + 3
    batch = Batch(Suite([p]), ALGO_RLEPSO, np.zeros(B, int), seeds, NP_, maxfes, maxfes // NLOG, NLOG, flags=flags)
    info, resident, got_flags = batch.launch_info(), batch.rollout_is_resident(), batch.flags
    batch.reset()
    torch.cuda.synchronize()
    template = batch.read_state(0)
    for b, (f, P, _, _) in enumerate(cases):
        batch.write_state(b, state_block(template, f, P))
    if route == 'per_generation':
        batch.set_tape(torch.from_numpy(np.tile(replay_tape(NP_, D_), (B, 1))).cuda())
        batch.step(torch.from_numpy(np.tile(ACTION, (B, 1))).cuda())
    else:
        rows = maxfes + 2 * NP_ + 1
        table = torch.zeros(rows, 2, 35, dtype=torch.float32)
        table[:, 0] = torch.from_numpy(ACTION)                       # mu = ACTION, sigma = 0: the sampled action IS the action of the tape route
        batch.rlepso_rollout(table.cuda().contiguous(), 1)
    torch.cuda.synchronize()
    buckets = {0: [0, 0], 1: [0, 0], 2: [0, 0], 3: [0, 0], 4: [0, 0]}           # ulp distance (4 = four or more) -> [pairs, disagreements]
    others = [0, 0]
    for b, (f, P, (q, j, k), ulps) in enumerate(cases):
        vel = oracle.split_rlepso_state(batch.read_state(b), NP_, D_, NLOG)['vel']
        u = U_FDR if route == 'per_generation' else _philox_fdr_weights(seeds[b], NP_, D_)
        agree = decode_agreement(vel, f, P, reference_targets(f, P), u)
        for d in range(D_):
            w = min(int(ulps[d]), 4)
            buckets[w][0] += 1
            buckets[w][1] += int(not agree[q, d])
        rest = np.delete(agree, q, axis=0)
        others[0] += rest.size
        others[1] += int((~rest).sum())
    batch.close()
    return buckets, others, got_flags, info, resident


def _print_buckets(title, buckets, others):
    print(title)
    for u in sorted(buckets):
        n, bad = buckets[u]
        print(f'  quotients {u}{"+" if u == 4 else ""} ulp apart: {bad} / {n} pairs resolved differently ({100. * bad / max(n, 1):.1f} %)')
    print(f'  all other (particle, dimension) items of the same swarms: {others[1]} / {others[0]}')


@pytest.mark.gpu
@pytest.mark.parametrize('route', ['per_generation', 'resident'])
@pytest.mark.parametrize('np_,dim,geometry', [(100, 10, 1), (128, 40, 2)])
def test_hip_fdr_is_the_reference_index_on_adversarial_swarms(np_, dim, geometry, route):
    """DEFAULT flags: the exemplar of every (particle, dimension) item equals np.argmin of the reference's rounded quotients -- 0 disagreements in every bucket, on the
    one-generation kernels AND on the resident rollout kernels the product and the bench run, at the geometries of BASELINE configs 2 and 5.
    src/optimizer/rlepso_optimizer.py:97-109."""
    cases = crafted_cases(384 if dim == 10 else 72, seed=dim, NP=np_, D=dim) + crafted_cases(128 if dim == 10 else 24, seed=dim + 1, NP=np_, D=dim, three=True)
    buckets, others, flags, info, resident = _run_crafted(cases, np_, dim, route, 0)
    assert flags == 0 and info['fixed_geometry'] == geometry and resident, (flags, info, resident)       # exact mode keeps the compile-time geometry and the resident route
    _print_buckets(f'FDR exemplar, default (exact) kernels, {route}, NP {np_} / D {dim}, vs reference formula (rounded quotients, np.argmin) on crafted near-ties:', buckets, others)
    assert buckets[0][0] >= 300 and buckets[1][0] >= 300, buckets
    assert all(v[1] == 0 for v in buckets.values()) and others[1] == 0, (buckets, others)


@pytest.mark.gpu
@pytest.mark.parametrize('route', ['per_generation', 'resident'])
def test_hip_fdr_fast_flag_measured_disagreement(route):
    """MBX_F_FDR_FAST: the cross-multiplied order alone.  Differs from the reference only where the two quotients round together or to neighbours."""
    from metabox_amd._abi import F_FDR_FAST
    cases = crafted_cases(512)
    buckets, others, flags, info, resident = _run_crafted(cases, NP, D, route, F_FDR_FAST)
    assert flags == F_FDR_FAST and info['fixed_geometry'] == 1 and resident, (flags, info)
    _print_buckets(f'FDR exemplar, MBX_F_FDR_FAST kernels, {route}, vs reference formula on crafted near-ties:', buckets, others)
    assert buckets[4][1] == 0 and others[1] == 0, (buckets, others)
    assert buckets[2][1] == 0 and buckets[3][1] == 0, buckets          # measured: only quotients that round together or to neighbours resolve differently
    assert buckets[0][1] > 0, buckets                                  # ... and there the fast form really is a different function (the flag reached the kernel)


@pytest.mark.gpu
def test_two_batches_with_different_flags_coexist_in_one_process():
    """The behaviour options travel in mbx_algo_cfg.flags (include/mbx.h), per batch: an exact and a fast batch created side by side, stepped alternately on the
    same crafted swarms, each keep their own FDR form; the generic-geometry and per-generation flags are per batch as well."""
    import torch
    from metabox_amd.suite import Batch, Suite
    from metabox_amd._abi import ALGO_RLEPSO, F_FDR_FAST, F_GENERIC_GEOMETRY, F_ROLLOUT_PER_GENERATION
    p = problems('bbob', D)[1]
    cases = crafted_cases(128, seed=5)
    B = len(cases)
    s = Suite([p])
    mk = lambda fl: Batch(s, ALGO_RLEPSO, np.zeros(B, int), np.arange(B), NP, MAXFES, LOGI, NLOG, flags=fl)
    exact, fast, generic, pergen = mk(0), mk(F_FDR_FAST), mk(F_GENERIC_GEOMETRY), mk(F_ROLLOUT_PER_GENERATION)
    assert (exact.flags, fast.flags, generic.flags, pergen.flags) == (0, F_FDR_FAST, F_GENERIC_GEOMETRY, F_ROLLOUT_PER_GENERATION)
    assert exact.launch_info()['fixed_geometry'] == 1 and generic.launch_info()['fixed_geometry'] == 0
    assert exact.rollout_is_resident() and fast.rollout_is_resident() and generic.rollout_is_resident() is False and pergen.rollout_is_resident() is False
    tape = torch.from_numpy(np.tile(replay_tape(), (B, 1))).cuda()
    acts = torch.from_numpy(np.tile(ACTION, (B, 1))).cuda()
    batches = (exact, fast, generic, pergen)
    for bt in batches:
        bt.reset()
    torch.cuda.synchronize()
    template = exact.read_state(0)
    for b, (f, P, _, _) in enumerate(cases):
        blk = state_block(template, f, P)
        for bt in batches:
            bt.write_state(b, blk)
    for bt in batches:                                                # alternate launches of the four batches on one stream
        bt.set_tape(tape)
        bt.step(acts)
    torch.cuda.synchronize()
    bad = {id(bt): 0 for bt in batches}
    for b, (f, P, (q, j, k), ulps) in enumerate(cases):
        tg = reference_targets(f, P)
        for bt in batches:
            vel = oracle.split_rlepso_state(bt.read_state(b), NP, D, NLOG)['vel']
            bad[id(bt)] += int((~decode_agreement(vel, f, P, tg)).sum())
    for bt in batches:
        bt.close()
    assert bad[id(exact)] == 0 and bad[id(generic)] == 0 and bad[id(pergen)] == 0 and bad[id(fast)] > 0, bad
    with pytest.raises(Exception):
        mk(1 << 9)                                                    # unknown flag bits are an argument error, not ignored
    # the near-tie flag bounds the denominators by the box: a block whose pbest positions leave it is refused, not silently mis-flagged
    probe = mk(0)
    probe.reset()
    torch.cuda.synchronize()
    f, P = cases[0][0], cases[0][1].copy()
    P[3, 2] = 5.5
    with pytest.raises(Exception, match='outside'):
        probe.write_state(0, state_block(probe.read_state(0), f, P))
    probe.close()


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


def _near_tie_items(f, P):
    """Number of (particle, dimension) items whose two smallest rounded quotients are <= 2 ulp apart (equal ones included) although they belong to different candidates."""
    NP_ = len(f)
    fdr = (f[None, :] - f[:, None])[:, :, None] / (np.abs(P[None, :, :] - P[:, None, :]) + 1e-5)          # [i, j, d]
    two = np.partition(fdr, 1, axis=1)[:, :2, :]
    lo, hi = two[:, 0, :], two[:, 1, :]
    same_sign = np.signbit(lo) == np.signbit(hi)
    ulp = np.abs(lo.view(np.int64) - hi.view(np.int64))
    return int((same_sign & (ulp <= 2) & (lo < 0)).sum())


@pytest.mark.gpu
@pytest.mark.parametrize('route', ['per_generation', 'resident'])
def test_hip_fdr_is_the_reference_index_on_converged_swarms(route):
    """Natural swarms at the rounding floor of their costs: RLEPSO runs whole episodes (no stop rule) on functions whose swarms collapse -- costs quantised at
    ulp(bias), coordinates clipped onto a bound, exact duplicates -- and the pbest tables of generations 40 / 100 / 160 / 199 are handed, as in the crafted test,
    to the reference formula and to the kernels.  There most items' best two quotients ARE ties; the exemplars must still be np.argmin's on every item."""
    import torch
    from metabox_amd.suite import Batch, Suite
    from metabox_amd._abi import ALGO_RLEPSO
    fids = (1, 2, 5, 7, 20, 24)
    ps = [problems('bbob', D)[k] for k in fids]
    B = 48
    seeds = np.arange(B, dtype=np.uint64) * 6007 + 1
    pidx = np.arange(B) % len(ps)
    run = Batch(Suite(ps), ALGO_RLEPSO, pidx, seeds, NP, MAXFES, LOGI, NLOG, early_stop=False)
    table = torch.rand(MAXFES + 2 * NP + 1, 2, 35, generator=torch.Generator().manual_seed(9))
    table[:, 1] = 0.02 + 0.1 * table[:, 1]
    table[:, 0, ::5] *= 0.05                                      # small c_mutation: few re-initialisations, the swarm is allowed to collapse
    table = table.cuda().contiguous()
    run.reset()
    swarms, done_gens = [], 0
    for upto in (40, 100, 160, 199):
        run.rlepso_rollout(table, upto - done_gens)
        done_gens = upto
        torch.cuda.synchronize()
        for b in range(B):
            sp = oracle.split_rlepso_state(run.read_state(b), NP, D, NLOG)
            swarms.append((sp['pbest'].copy(), sp['pbpos'].reshape(NP, D).copy(), None, None))
    run.close()
    ties = sum(_near_tie_items(f, P) for f, P, _, _ in swarms)
    dup = sum(NP - len(np.unique(f)) for f, P, _, _ in swarms)
    # the same machinery as the crafted test; the bucket bookkeeping is bypassed (no designated pair), every item counts
    p = problems('bbob', D)[1]
    n = len(swarms)
    probe = Batch(Suite([p]), ALGO_RLEPSO, np.zeros(n, int), np.arange(n, dtype=np.uint64) * 7 + 3, NP, MAXFES, LOGI, NLOG)
    assert probe.flags == 0 and probe.rollout_is_resident()
    probe.reset()
    torch.cuda.synchronize()
    template = probe.read_state(0)
    for b, (f, P, _, _) in enumerate(swarms):
        probe.write_state(b, state_block(template, f, P))
    if route == 'per_generation':
        probe.set_tape(torch.from_numpy(np.tile(replay_tape(), (n, 1))).cuda())
        probe.step(torch.from_numpy(np.tile(ACTION, (n, 1))).cuda())
    else:
        tb = torch.zeros(MAXFES + 2 * NP + 1, 2, 35, dtype=torch.float32)
        tb[:, 0] = torch.from_numpy(ACTION)
        probe.rlepso_rollout(tb.cuda().contiguous(), 1)
    torch.cuda.synchronize()
    items = bad = 0
    for b, (f, P, _, _) in enumerate(swarms):
        vel = oracle.split_rlepso_state(probe.read_state(b), NP, D, NLOG)['vel']
        u = U_FDR if route == 'per_generation' else _philox_fdr_weights(b * 7 + 3, NP, D)
        agree = decode_agreement(vel, f, P, reference_targets(f, P), u)
        items += agree.size
        bad += int((~agree).sum())
    probe.close()
    print(f'converged natural swarms ({route}): {bad} of {items} items moved with an exemplar other than np.argmin\'s; {ties} items whose two best quotients are <= 2 ulp '
          f'apart, {dup} duplicate pbest costs in {n} swarms')
    assert ties > 1000 and bad == 0


def craft_copy_pairs(rs, NP=NP, D=D):
    """A swarm in which particles 2m and 2m + 1 (in a random index order) share their pbest COST and the first half of their coordinates but not the rest: every candidate has an
    exact copy in dimensions < D / 2 -- a tie np.argmin resolves by the lower particle index -- while the rows differ, so the row-level copy marking (rl_mark_copies) cannot take
    them out of the scan.  Hundreds of flagged items per swarm: the kernels settle them lane by lane (fdr_settle_own), not wave by wave."""
    perm = rs.permutation(NP)
    f = np.empty(NP)
    P = rs.uniform(-4, 4, (NP, D))
    for m in range(NP // 2):
        a, b = perm[2 * m], perm[2 * m + 1]
        f[a] = f[b] = 1000. - 10. * m - rs.uniform(0., 5.)
        P[b, :D // 2] = P[a, :D // 2]
    if NP % 2:
        f[perm[-1]] = 2000.
    return f, P


@pytest.mark.gpu
@pytest.mark.parametrize('route', ['per_generation', 'resident'])
def test_hip_fdr_with_copies_in_half_of_the_coordinates(route):
    """Exact ties everywhere (craft_copy_pairs): the exemplar must be the LOWER-INDEXED copy in every item, on the path that settles a swarm with more than 64 flagged items."""
    import torch
    from metabox_amd.suite import Batch, Suite
    from metabox_amd._abi import ALGO_RLEPSO
    rs = np.random.RandomState(21)
    swarms = [craft_copy_pairs(rs) for _ in range(48)]
    p = problems('bbob', D)[1]
    n = len(swarms)
    seeds = np.arange(n, dtype=np.uint64) * 7 + 3
    probe = Batch(Suite([p]), ALGO_RLEPSO, np.zeros(n, int), seeds, NP, MAXFES, LOGI, NLOG)
    probe.reset()
    torch.cuda.synchronize()
    template = probe.read_state(0)
    for b, (f, P) in enumerate(swarms):
        probe.write_state(b, state_block(template, f, P))
    if route == 'per_generation':
        probe.set_tape(torch.from_numpy(np.tile(replay_tape(), (n, 1))).cuda())
        probe.step(torch.from_numpy(np.tile(ACTION, (n, 1))).cuda())
    else:
        tb = torch.zeros(MAXFES + 2 * NP + 1, 2, 35, dtype=torch.float32)
        tb[:, 0] = torch.from_numpy(ACTION)
        probe.rlepso_rollout(tb.cuda().contiguous(), 1)
    torch.cuda.synchronize()
    items = bad = tied = 0
    for b, (f, P) in enumerate(swarms):
        vel = oracle.split_rlepso_state(probe.read_state(b), NP, D, NLOG)['vel']
        u = U_FDR if route == 'per_generation' else _philox_fdr_weights(seeds[b], NP, D)
        tg = reference_targets(f, P)
        agree = decode_agreement(vel, f, P, tg, u)
        items += agree.size
        bad += int((~agree).sum())
        tied += _near_tie_items(f, P)
    probe.close()
    print(f'copy pairs ({route}): {bad} of {items} items moved with an exemplar other than np.argmin\'s; {tied} items whose two best quotients are equal or <= 2 ulp apart')
    assert tied > 5000 and bad == 0


def craft_grid_swarm(rs, NP=NP, D=D, cost_levels=12, step=0.25):
    """Low-entropy swarm: pbest costs from a dozen levels, coordinates on a 0.25 grid inside the box -- equal costs, equal denominators, whole-row copies, candidates that
    tie in one coordinate and differ in the next, query particles with nobody better: every tie-breaking rule of np.argmin is exercised in every swarm."""
    f = 1000. - 3. * rs.randint(0, cost_levels, NP)
    P = step * rs.randint(-16, 17, (NP, D)).astype(np.float64)
    for _ in range(NP // 10):                             # some whole-row copies (same cost, same position), as on a collapsed linear slope
        a, b = rs.choice(NP, 2, replace=False)
        f[b], P[b] = f[a], P[a]
    return f, P


@pytest.mark.gpu
@pytest.mark.parametrize('route', ['per_generation', 'resident'])
@pytest.mark.parametrize('np_,dim', [(100, 10), (128, 40)])
def test_hip_fdr_on_grid_swarms(np_, dim, route):
    """craft_grid_swarm through both routes and both BASELINE geometries: the velocity of every (particle, dimension) item must be the one np.argmin's exemplar gives."""
    import torch
    from metabox_amd.suite import Batch, Suite
    from metabox_amd._abi import ALGO_RLEPSO
    rs = np.random.RandomState(100 + dim)
    swarms = [craft_grid_swarm(rs, np_, dim, cost_levels=(4, 12, 40)[k % 3]) for k in range(48 if dim == 10 else 12)]
    maxfes = 2000 * dim
    p = problems('bbob', dim)[1]
    n = len(swarms)
    seeds = np.arange(n, dtype=np.uint64) * 7 + 3
    probe = Batch(Suite([p]), ALGO_RLEPSO, np.zeros(n, int), seeds, np_, maxfes, maxfes // NLOG, NLOG)
    assert probe.flags == 0 and probe.rollout_is_resident()
    probe.reset()
    torch.cuda.synchronize()
    template = probe.read_state(0)
    for b, (f, P) in enumerate(swarms):
        probe.write_state(b, state_block(template, f, P))
    if route == 'per_generation':
        probe.set_tape(torch.from_numpy(np.tile(replay_tape(np_, dim), (n, 1))).cuda())
        probe.step(torch.from_numpy(np.tile(ACTION, (n, 1))).cuda())
    else:
        tb = torch.zeros(maxfes + 2 * np_ + 1, 2, 35, dtype=torch.float32)
        tb[:, 0] = torch.from_numpy(ACTION)
        probe.rlepso_rollout(tb.cuda().contiguous(), 1)
    torch.cuda.synchronize()
    items = bad = 0
    for b, (f, P) in enumerate(swarms):
        vel = oracle.split_rlepso_state(probe.read_state(b), np_, dim, NLOG)['vel']
        u = U_FDR if route == 'per_generation' else _philox_fdr_weights(seeds[b], np_, dim)
        agree = decode_agreement(vel, f, P, reference_targets(f, P), u)
        items += agree.size
        bad += int((~agree).sum())
    probe.close()
    print(f'grid swarms NP {np_} / D {dim} ({route}): {bad} of {items} items moved with an exemplar other than np.argmin\'s')
    assert bad == 0
