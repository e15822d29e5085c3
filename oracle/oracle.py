"""ctypes front-end of oracle/liboracle.so — TEST INFRASTRUCTURE, NOT PRODUCT.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from metabox_amd._abi import AlgoCfg, ProblemDesc, pack_desc

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, 'liboracle.so')
_dp = C.POINTER(C.c_double)


def build(force=False):
    src = os.path.join(_HERE, 'mbx_oracle.c')
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', _HERE, '-s', '-B' if force else '-s'])
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB)
        L.orc_eval.argtypes = [C.POINTER(ProblemDesc), _dp, C.c_int, _dp]
        L.orc_apply_noise.argtypes = [C.POINTER(ProblemDesc), C.c_double, _dp, C.c_int, _dp, _dp]
        L.orc_eval_noisy_philox.argtypes = [C.POINTER(ProblemDesc), C.c_double, _dp, C.c_int, C.c_uint64, _dp]
        L.orc_philox.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                 C.POINTER(C.c_uint32)]
        L.orc_rlepso_new.restype = C.c_void_p
        L.orc_rlepso_new.argtypes = [C.POINTER(ProblemDesc), C.c_double, C.POINTER(AlgoCfg), C.c_uint64]
        L.orc_rlepso_free.argtypes = [C.c_void_p]
        L.orc_rlepso_reset.restype = C.c_double
        L.orc_rlepso_reset.argtypes = [C.c_void_p, _dp]
        L.orc_rlepso_step.argtypes = [C.c_void_p, C.POINTER(C.c_float), _dp, _dp]
        L.orc_rlepso_state.argtypes = [C.c_void_p, _dp]
        L.orc_rlepso_set_state.argtypes = [C.c_void_p, _dp]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(_dp)


def evaluate(desc, x):
    """F*.func: [n, dim] -> [n] (bias included)."""
    st, keep = pack_desc(desc)
    x = np.ascontiguousarray(x, dtype=np.float64)
    f = np.empty(x.shape[0])
    lib().orc_eval(C.byref(st), _p(x), x.shape[0], _p(f))
    return f


def apply_noise(desc, optimum, ftrue, draws):
    st, keep = pack_desc(desc)
    ftrue = np.ascontiguousarray(ftrue, dtype=np.float64)
    draws = np.ascontiguousarray(draws, dtype=np.float64)
    out = np.empty_like(ftrue)
    lib().orc_apply_noise(C.byref(st), float(optimum), _p(ftrue), len(ftrue), _p(draws), _p(out))
    return out


def evaluate_noisy_philox(desc, optimum, x, seed):
    st, keep = pack_desc(desc)
    x = np.ascontiguousarray(x, dtype=np.float64)
    f = np.empty(x.shape[0])
    lib().orc_eval_noisy_philox(C.byref(st), float(optimum), _p(x), x.shape[0], int(seed), _p(f))
    return f


def philox(seed, idx, site, gen, episode):
    out = (C.c_uint32 * 4)()
    lib().orc_philox(int(seed), idx, site, gen, episode, out)
    return list(out)


def make_cfg(algo, np_, dim, max_fes, log_interval, n_logpoint, early_stop=1, n_group=5):
    return AlgoCfg(algo, np_, dim, max_fes, log_interval, n_logpoint, early_stop, n_group)


def rlepso_state_doubles(NP, D, nlog):
    return 3 * NP * D + 3 * NP + D + 16 + nlog + 1


class RlepsoOracle:
    """One RLEPSO instance on the CPU (init_population / update restated in C)."""

    def __init__(self, desc, optimum, cfg, seed=0):
        self._st, self._keep = pack_desc(desc)
        self.cfg = cfg
        self._h = lib().orc_rlepso_new(C.byref(self._st), float('nan') if optimum is None else float(optimum),
                                       C.byref(cfg), int(seed))

    def __del__(self):
        if getattr(self, '_h', None):
            lib().orc_rlepso_free(self._h)
            self._h = None

    def reset(self, tape=None):
        return lib().orc_rlepso_reset(self._h, _p(tape) if tape is not None else None)

    def step(self, action, tape=None):
        a = np.ascontiguousarray(action, dtype=np.float32)
        out = np.empty(3)
        lib().orc_rlepso_step(self._h, a.ctypes.data_as(C.POINTER(C.c_float)),
                              _p(tape) if tape is not None else None, _p(out))
        return out[0], out[1], bool(out[2])

    def state(self):
        out = np.empty(rlepso_state_doubles(self.cfg.np, self.cfg.dim, self.cfg.n_logpoint))
        lib().orc_rlepso_state(self._h, _p(out))
        return out

    def set_state(self, block):
        block = np.ascontiguousarray(block, dtype=np.float64)
        assert block.shape == (rlepso_state_doubles(self.cfg.np, self.cfg.dim, self.cfg.n_logpoint),)
        lib().orc_rlepso_set_state(self._h, _p(block))


# scalar slots (include/mbx_layout.h)
SC_GBEST, SC_FES, SC_LOG_INDEX, SC_COST_LEN, SC_DONE, SC_RETURN, SC_GEN, SC_EPISODE, SC_GBEST_IDX, SC_REINIT = range(10)
NSCALAR = 16


def split_rlepso_state(st, NP, D, nlog):
    o = 0
    out = {}
    for name, n in (('pos', NP * D), ('vel', NP * D), ('pbpos', NP * D), ('ccost', NP), ('pbest', NP),
                    ('pni', NP), ('gbpos', D), ('scalars', NSCALAR), ('cost', nlog + 1)):
        out[name] = st[o:o + n]
        o += n
    return out


class NumpyTapeFeeder:
    """Regenerates, from a seed, the numpy legacy-stream draws that RLEPSO_Optimizer consumes, in the
    reference's call order, and lays them out as the per-step tape of include/mbx_layout.h.

    Draw order per update() (rlepso_optimizer.py:179-180,77,88,108,[eval noise],238 and,
    only when the re-init mask is non-empty, 137-138,[eval noise]).  The conditional draws are produced
    speculatively; ``commit(reinit_fired)`` rewinds the stream when the step did not re-initialise.
    """

    def __init__(self, seed, NP, D, noise_kind):
        self.rs = np.random.RandomState(seed)
        self.NP, self.D, self.noise = NP, D, noise_kind
        self.stride = 9 * NP + 6 * NP * D
        self._rewind = None

    def _noise_rows(self):
        NP = self.NP
        rows = np.zeros((3, NP))
        if self.noise == 1:
            rows[0] = self.rs.randn(NP)
        elif self.noise == 2:
            rows[0] = self.rs.rand(NP)
            rows[1] = self.rs.rand(NP)
        elif self.noise == 3:
            rows[0] = self.rs.rand(NP)
            rows[1] = self.rs.randn(NP)
            rows[2] = self.rs.randn(NP)
        return rows.ravel()

    def _fill_reinit(self, tape):
        NP, D = self.NP, self.D
        o = 6 * NP + 4 * NP * D
        tape[o:o + NP * D] = self.rs.random_sample((NP, D)).ravel()
        tape[o + NP * D:o + 2 * NP * D] = self.rs.random_sample((NP, D)).ravel()
        tape[o + 2 * NP * D:o + 2 * NP * D + 3 * NP] = self._noise_rows()

    def reset_tape(self):
        tape = np.zeros(self.stride)
        self._fill_reinit(tape)
        return tape

    def step_tape(self):
        NP, D = self.NP, self.D
        tape = np.zeros(self.stride)
        tape[0:NP] = self.rs.rand(NP, 1).ravel()
        tape[NP:2 * NP] = self.rs.rand(NP, 1).ravel()
        o = 2 * NP
        tape[o:o + NP * D] = self.rs.rand(NP, D).ravel()
        o += NP * D
        tape[o:o + 2 * NP * D] = self.rs.randint(low=0, high=NP, size=(NP, D, 2)).ravel()
        o += 2 * NP * D
        tape[o:o + NP * D] = self.rs.rand(NP, D).ravel()
        o += NP * D
        tape[o:o + 3 * NP] = self._noise_rows()
        o += 3 * NP
        tape[o:o + NP] = self.rs.rand(NP)
        self._rewind = self.rs.get_state()
        self._fill_reinit(tape)
        return tape

    def commit(self, reinit_fired):
        if not reinit_fired:
            self.rs.set_state(self._rewind)
        self._rewind = None


# ======================================================================================== LDE
def _lde_lib():
    L = lib()
    if not getattr(L, '_lde_ready', False):
        L.orc_lde_new.restype = C.c_void_p
        L.orc_lde_new.argtypes = [C.POINTER(ProblemDesc), C.c_double, C.POINTER(AlgoCfg), C.c_uint64]
        L.orc_lde_free.argtypes = [C.c_void_p]
        L.orc_lde_reset.argtypes = [C.c_void_p, _dp, _dp]
        L.orc_lde_step.argtypes = [C.c_void_p, C.POINTER(C.c_float), _dp, _dp, _dp]
        L.orc_lde_state.argtypes = [C.c_void_p, _dp]
        L._lde_ready = True
    return L


def lde_state_doubles(NP, D, nlog):
    return NP * D + NP + 8 + 16 + nlog + 1


def split_lde_state(st, NP, D, nlog):
    o, out = 0, {}
    for name, n in (('pop', NP * D), ('fit', NP), ('hsum', 8), ('scalars', NSCALAR), ('cost', nlog + 1)):
        out[name] = st[o:o + n]
        o += n
    return out


class LdeOracle:
    """One LDE instance on the CPU (lde_optimizer.py restated in C)."""

    def __init__(self, desc, optimum, cfg, seed=0):
        self._st, self._keep = pack_desc(desc)
        self.cfg = cfg
        self._h = _lde_lib().orc_lde_new(C.byref(self._st), float('nan') if optimum is None else float(optimum),
                                         C.byref(cfg), int(seed))
        self._state = np.empty(cfg.np + 10)

    def __del__(self):
        if getattr(self, '_h', None):
            _lde_lib().orc_lde_free(self._h)
            self._h = None

    def reset(self, tape=None):
        _lde_lib().orc_lde_reset(self._h, _p(tape) if tape is not None else None, _p(self._state))
        return self._state.copy()

    def step(self, action, tape=None):
        a = np.ascontiguousarray(action, dtype=np.float32).reshape(-1)
        out = np.empty(3)
        _lde_lib().orc_lde_step(self._h, a.ctypes.data_as(C.POINTER(C.c_float)), _p(tape) if tape is not None else None,
                                _p(self._state), _p(out))
        return self._state.copy(), out[0], bool(out[1])

    def state(self):
        out = np.empty(lde_state_doubles(self.cfg.np, self.cfg.dim, self.cfg.n_logpoint))
        _lde_lib().orc_lde_state(self._h, _p(out))
        return out


class LdeTapeFeeder:
    """numpy legacy-stream draws of LDE_Optimizer in the reference's call order (init: uniform(NP,D), eval noise;
    update: randint(0, ceil(NP*p), (1,NP)), uniform(1,NP,D), randint(0, D, [1,NP]), eval noise).  The torch.randint
    indices r (after the rejection loop) come from the fixture."""

    def __init__(self, seed, NP, D, noise_kind, max_fes):
        self.rs = np.random.RandomState(seed)
        self.NP, self.D, self.noise, self.max_fes = NP, D, noise_kind, max_fes
        self.stride = 7 * NP + NP * D
        self.fes = NP

    def _noise(self):
        return NumpyTapeFeeder._noise_rows(self)

    def reset_tape(self):
        NP, D = self.NP, self.D
        t = np.zeros(self.stride)
        t[7 * NP:] = self.rs.uniform(size=(NP, D)).ravel()
        t[4 * NP:7 * NP] = self._noise()
        self.fes = NP
        return t

    def step_tape(self, r):
        NP, D = self.NP, self.D
        t = np.zeros(self.stride)
        p_rate = (2 / NP - 1) * self.fes / self.max_fes + 1
        t[0:NP] = self.rs.randint(0, int(np.ceil(NP * max(0, p_rate))), size=(1, NP)).ravel()
        t[NP:2 * NP] = r[:, 0]
        t[2 * NP:3 * NP] = r[:, 1]
        t[7 * NP:] = self.rs.uniform(size=(1, NP, D)).ravel()
        t[3 * NP:4 * NP] = self.rs.randint(low=0, high=D, size=[1, NP]).ravel()
        t[4 * NP:7 * NP] = self._noise()
        self.fes += NP
        return t


# ======================================================================================== DE-DDQN
def _dq_lib():
    L = lib()
    if not getattr(L, '_dq_ready', False):
        L.orc_dq_new.restype = C.c_void_p
        L.orc_dq_new.argtypes = [C.POINTER(ProblemDesc), C.c_double, C.POINTER(AlgoCfg), C.c_uint64]
        L.orc_dq_free.argtypes = [C.c_void_p]
        L.orc_dq_reset.argtypes = [C.c_void_p, _dp, _dp]
        L.orc_dq_step.argtypes = [C.c_void_p, C.c_int, _dp, _dp, _dp]
        L.orc_dq_state.argtypes = [C.c_void_p, _dp]
        L._dq_ready = True
    return L


def dq_state_doubles(NP, D, nlog):
    return NP * D + NP + 2 * D + 8 + 40 + 3 * 160 + 300 + 16 + 16 + nlog + 1


def split_dq_state(st, NP, D, nlog):
    o, out = 0, {}
    for name, n in (('X', NP * D), ('cost', NP), ('gbpos', D), ('prepos', D), ('r', 8), ('ntot', 40), ('nsucc', 160),
                    ('omsum', 160), ('ommax', 160), ('omw', 300), ('extra', 16), ('scalars', NSCALAR), ('clog', nlog + 1)):
        out[name] = st[o:o + n]
        o += n
    return out


class DqOracle:
    """One DE-DDQN instance on the CPU (de_ddqn_optimizer.py restated in C)."""

    def __init__(self, desc, optimum, cfg, seed=0):
        self._st, self._keep = pack_desc(desc)
        self.cfg = cfg
        self._h = _dq_lib().orc_dq_new(C.byref(self._st), float('nan') if optimum is None else float(optimum), C.byref(cfg), int(seed))
        self._state = np.empty(99)

    def __del__(self):
        if getattr(self, '_h', None):
            _dq_lib().orc_dq_free(self._h)
            self._h = None

    def reset(self, tape=None):
        _dq_lib().orc_dq_reset(self._h, _p(tape) if tape is not None else None, _p(self._state))
        return self._state.copy()

    def step(self, action, tape=None):
        out = np.empty(3)
        _dq_lib().orc_dq_step(self._h, int(action), _p(tape) if tape is not None else None, _p(self._state), _p(out))
        return self._state.copy(), out[0], bool(out[1])

    def state(self):
        out = np.empty(dq_state_doubles(self.cfg.np, self.cfg.dim, self.cfg.n_logpoint))
        _dq_lib().orc_dq_state(self._h, _p(out))
        return out


class DqTapeFeeder:
    """numpy legacy-stream draws of DE_DDQN_Optimizer in the reference's call order (init: rand(NP,D), eval noise,
    randint(0,NP,5); update: randint(D,size=1), rand(1,D), eval noise of one value, randint(0,NP,5))."""

    def __init__(self, seed, NP, D, noise_kind):
        self.rs = np.random.RandomState(seed)
        self.NP, self.D, self.noise = NP, D, noise_kind
        self.stride = 16 + NP * D + 3 * NP

    def _noise(self, n):
        rows = np.zeros((3, n))
        if self.noise == 1:
            rows[0] = self.rs.randn(n) if n > 1 else self.rs.randn()
        elif self.noise == 2:
            rows[0] = self.rs.rand(n) if n > 1 else self.rs.rand()
            rows[1] = self.rs.rand(n) if n > 1 else self.rs.rand()
        elif self.noise == 3:
            rows[0] = self.rs.rand(n) if n > 1 else self.rs.rand()
            rows[1] = self.rs.randn(n) if n > 1 else self.rs.randn()
            rows[2] = self.rs.randn(n) if n > 1 else self.rs.randn()
        return rows

    def reset_tape(self):
        NP, D = self.NP, self.D
        t = np.zeros(self.stride)
        t[16:16 + NP * D] = self.rs.rand(NP, D).ravel()
        t[16 + NP * D:16 + NP * D + 3 * NP] = self._noise(NP).ravel()
        t[0:5] = self.rs.randint(0, NP, 5)
        return t

    def step_tape(self):
        NP, D = self.NP, self.D
        t = np.zeros(self.stride)
        t[5] = self.rs.randint(D, size=1)[0]
        t[16:16 + D] = self.rs.rand(1, D).ravel()
        t[8:11] = self._noise(1).ravel()
        t[0:5] = self.rs.randint(0, NP, 5)
        return t


# ======================================================================================== RL-PSO
def _rlpso_lib():
    L = lib()
    if not getattr(L, '_rlpso_ready', False):
        L.orc_rlpso_new.restype = C.c_void_p
        L.orc_rlpso_new.argtypes = [C.POINTER(ProblemDesc), C.c_double, C.POINTER(AlgoCfg), C.c_uint64]
        L.orc_rlpso_free.argtypes = [C.c_void_p]
        L.orc_rlpso_reset.argtypes = [C.c_void_p, _dp, _dp]
        L.orc_rlpso_step.argtypes = [C.c_void_p, C.c_float, _dp, _dp, _dp]
        L.orc_rlpso_state.argtypes = [C.c_void_p, _dp]
        L._rlpso_ready = True
    return L


SC_RLPSO_W, SC_RLPSO_MAXCOST, SC_RLPSO_CUR = 10, 11, 12


def rlpso_state_doubles(NP, D, nlog):
    return 3 * NP * D + 2 * NP + D + 16 + nlog + 1


def split_rlpso_state(st, NP, D, nlog):
    o, out = 0, {}
    for name, n in (('pos', NP * D), ('vel', NP * D), ('pbpos', NP * D), ('ccost', NP), ('pbest', NP), ('gbpos', D),
                    ('scalars', NSCALAR), ('cost', nlog + 1)):
        out[name] = st[o:o + n]
        o += n
    return out


class RlpsoOracle:
    """One RL-PSO instance on the CPU (rl_pso_optimizer.py restated in C)."""

    def __init__(self, desc, optimum, cfg, seed=0):
        self._st, self._keep = pack_desc(desc)
        self.cfg = cfg
        self._h = _rlpso_lib().orc_rlpso_new(C.byref(self._st), float('nan') if optimum is None else float(optimum), C.byref(cfg),
                                             int(seed))
        self._state = np.empty(2 * cfg.dim)

    def __del__(self):
        if getattr(self, '_h', None):
            _rlpso_lib().orc_rlpso_free(self._h)
            self._h = None

    def reset(self, tape=None):
        _rlpso_lib().orc_rlpso_reset(self._h, _p(tape) if tape is not None else None, _p(self._state))
        return self._state.copy()

    def step(self, action, tape=None):
        out = np.empty(2)
        _rlpso_lib().orc_rlpso_step(self._h, float(np.float32(action)), _p(tape) if tape is not None else None, _p(self._state), _p(out))
        return self._state.copy(), out[0], bool(out[1])

    def state(self):
        out = np.empty(rlpso_state_doubles(self.cfg.np, self.cfg.dim, self.cfg.n_logpoint))
        _rlpso_lib().orc_rlpso_state(self._h, _p(out))
        return out


class RlpsoTapeFeeder:
    """numpy legacy-stream draws of RL_PSO_Optimizer in the reference's call order (init_population: uniform(NP, D) positions,
    uniform(NP, D) velocities, evaluation noise of NP values; update: rand(), evaluation noise of one value)."""

    def __init__(self, seed, NP, D, noise_kind):
        self.rs = np.random.RandomState(seed)
        self.NP, self.D, self.noise = NP, D, noise_kind
        self.stride = 2 * NP * D + 3 * NP

    def _noise(self, n):
        rows = np.zeros((3, n))
        if self.noise == 1:
            rows[0] = self.rs.randn(n) if n > 1 else self.rs.randn()
        elif self.noise == 2:
            rows[0] = self.rs.rand(n) if n > 1 else self.rs.rand()
            rows[1] = self.rs.rand(n) if n > 1 else self.rs.rand()
        elif self.noise == 3:
            rows[0] = self.rs.rand(n) if n > 1 else self.rs.rand()
            rows[1] = self.rs.randn(n) if n > 1 else self.rs.randn()
            rows[2] = self.rs.randn(n) if n > 1 else self.rs.randn()
        return rows

    def reset_tape(self):
        NP, D = self.NP, self.D
        t = np.zeros(self.stride)
        t[0:NP * D] = self.rs.random_sample((NP, D)).ravel()
        t[NP * D:2 * NP * D] = self.rs.random_sample((NP, D)).ravel()
        t[2 * NP * D:2 * NP * D + 3 * NP] = self._noise(NP).ravel()
        return t

    def step_tape(self):
        t = np.zeros(self.stride)
        t[0] = self.rs.rand()
        t[1:4] = self._noise(1).ravel()
        return t


# ======================================================================================== GLEET
def _gleet_lib():
    L = lib()
    if not getattr(L, '_gleet_ready', False):
        L.orc_gleet_new.restype = C.c_void_p
        L.orc_gleet_new.argtypes = [C.POINTER(ProblemDesc), C.c_double, C.POINTER(AlgoCfg), C.c_uint64]
        L.orc_gleet_free.argtypes = [C.c_void_p]
        L.orc_gleet_reset.argtypes = [C.c_void_p, _dp, _dp]
        L.orc_gleet_step.argtypes = [C.c_void_p, C.POINTER(C.c_float), _dp, _dp, _dp]
        L.orc_gleet_state.argtypes = [C.c_void_p, _dp]
        L._gleet_ready = True
    return L


SC_GLEET_W, SC_GLEET_MAXCOST, SC_GLEET_NOIMPROVE = 10, 11, 12


def gleet_state_doubles(NP, D, nlog):
    return 3 * NP * D + 3 * NP + D + 9 * NP + 10 + 16 + nlog + 1


def split_gleet_state(st, NP, D, nlog):
    o, out = 0, {}
    for name, n in (('pos', NP * D), ('vel', NP * D), ('pbpos', NP * D), ('ccost', NP), ('pbest', NP), ('pni', NP), ('gbpos', D),
                    ('pfeat', 9 * NP), ('gfeat', 10), ('scalars', NSCALAR), ('cost', nlog + 1)):
        out[name] = st[o:o + n]
        o += n
    return out


class GleetOracle:
    """One GLEET instance on the CPU (gleet_optimizer.py restated in C).  State = [NP, 27]."""

    def __init__(self, desc, optimum, cfg, seed=0):
        self._st, self._keep = pack_desc(desc)
        self.cfg = cfg
        self._h = _gleet_lib().orc_gleet_new(C.byref(self._st), float('nan') if optimum is None else float(optimum), C.byref(cfg),
                                             int(seed))
        self._state = np.empty((cfg.np, 27))

    def __del__(self):
        if getattr(self, '_h', None):
            _gleet_lib().orc_gleet_free(self._h)
            self._h = None

    def reset(self, tape=None):
        _gleet_lib().orc_gleet_reset(self._h, _p(tape) if tape is not None else None, _p(self._state))
        return self._state.copy()

    def step(self, action, tape=None):
        a = np.ascontiguousarray(action, dtype=np.float32)
        out = np.empty(2)
        _gleet_lib().orc_gleet_step(self._h, a.ctypes.data_as(C.POINTER(C.c_float)), _p(tape) if tape is not None else None,
                                    _p(self._state), _p(out))
        return self._state.copy(), out[0], bool(out[1])

    def state(self):
        out = np.empty(gleet_state_doubles(self.cfg.np, self.cfg.dim, self.cfg.n_logpoint))
        _gleet_lib().orc_gleet_state(self._h, _p(out))
        return out


class GleetTapeFeeder:
    """numpy legacy-stream draws of GLEET_Optimizer in the reference's call order (init: uniform(NP, D) positions, uniform(NP, D)
    velocities, evaluation noise; update: rand(NP, 1), rand(NP, 1), evaluation noise)."""

    def __init__(self, seed, NP, D, noise_kind):
        self.rs = np.random.RandomState(seed)
        self.NP, self.D, self.noise = NP, D, noise_kind
        self.stride = 2 * NP * D + 3 * NP

    def _noise(self):
        NP = self.NP
        rows = np.zeros((3, NP))
        if self.noise == 1:
            rows[0] = self.rs.randn(NP)
        elif self.noise == 2:
            rows[0] = self.rs.rand(NP)
            rows[1] = self.rs.rand(NP)
        elif self.noise == 3:
            rows[0] = self.rs.rand(NP)
            rows[1] = self.rs.randn(NP)
            rows[2] = self.rs.randn(NP)
        return rows.ravel()

    def reset_tape(self):
        NP, D = self.NP, self.D
        t = np.zeros(self.stride)
        t[0:NP * D] = self.rs.random_sample((NP, D)).ravel()
        t[NP * D:2 * NP * D] = self.rs.random_sample((NP, D)).ravel()
        t[2 * NP * D:2 * NP * D + 3 * NP] = self._noise()
        return t

    def step_tape(self):
        NP = self.NP
        t = np.zeros(self.stride)
        t[0:NP] = self.rs.rand(NP, 1).ravel()
        t[NP:2 * NP] = self.rs.rand(NP, 1).ravel()
        t[2 * NP:5 * NP] = self._noise()
        return t


# ======================================================================================== QLPSO
def _qlpso_lib():
    L = lib()
    if not getattr(L, '_qlpso_ready', False):
        L.orc_qlpso_new.restype = C.c_void_p
        L.orc_qlpso_new.argtypes = [C.POINTER(ProblemDesc), C.c_double, C.POINTER(AlgoCfg), C.c_uint64]
        L.orc_qlpso_free.argtypes = [C.c_void_p]
        L.orc_qlpso_reset.restype = C.c_double
        L.orc_qlpso_reset.argtypes = [C.c_void_p, _dp]
        L.orc_qlpso_step.restype = C.c_double
        L.orc_qlpso_step.argtypes = [C.c_void_p, C.c_int, _dp, _dp]
        L.orc_qlpso_choose.restype = C.c_int
        L.orc_qlpso_choose.argtypes = [_dp, C.c_double]
        L.orc_qlpso_state.argtypes = [C.c_void_p, _dp]
        L._qlpso_ready = True
    return L


SC_QLPSO_DIVERSITY, SC_QLPSO_POINTER = 10, 11


def qlpso_state_doubles(NP, D, nlog):
    return 3 * NP * D + 2 * NP + 16 + nlog + 1


def split_qlpso_state(st, NP, D, nlog):
    o, out = 0, {}
    for name, n in (('pop', NP * D), ('vel', NP * D), ('pbpos', NP * D), ('cost', NP), ('sstate', NP), ('scalars', NSCALAR),
                    ('clog', nlog + 1)):
        out[name] = st[o:o + n]
        o += n
    return out


def qlpso_choose(q_row, u):
    """QLPSO_Agent.__get_action for one state row of the Q-table and one uniform (C restatement)."""
    q = np.ascontiguousarray(q_row, dtype=np.float64)
    return int(_qlpso_lib().orc_qlpso_choose(_p(q), float(u)))


class QlpsoOracle:
    """One QLPSO instance on the CPU (qlpso_optimizer.py restated in C).  Like the reference object, it keeps its particle pointer
    across resets."""

    def __init__(self, desc, optimum, cfg, seed=0):
        self._st, self._keep = pack_desc(desc)
        self.cfg = cfg
        self._h = _qlpso_lib().orc_qlpso_new(C.byref(self._st), float('nan') if optimum is None else float(optimum), C.byref(cfg),
                                             int(seed))

    def __del__(self):
        if getattr(self, '_h', None):
            _qlpso_lib().orc_qlpso_free(self._h)
            self._h = None

    def reset(self, tape=None):
        return int(_qlpso_lib().orc_qlpso_reset(self._h, _p(tape) if tape is not None else None))

    def step(self, action, tape=None):
        out = np.empty(2)
        s = _qlpso_lib().orc_qlpso_step(self._h, int(action), _p(tape) if tape is not None else None, _p(out))
        return int(s), out[0], bool(out[1])

    def state(self):
        out = np.empty(qlpso_state_doubles(self.cfg.np, self.cfg.dim, self.cfg.n_logpoint))
        _qlpso_lib().orc_qlpso_state(self._h, _p(out))
        return out


class QlpsoTapeFeeder:
    """numpy legacy-stream draws of a QLPSO rollout in the reference's call order.  init_population: rand(NP, D), evaluation noise,
    randint(0, 4, NP).  Per step: (agent) the uniform of np.random.choice when `policy_draws`, then (optimizer) rand(), rand(),
    evaluation noise of one value."""

    def __init__(self, seed, NP, D, noise_kind, policy_draws=True):
        self.rs = np.random.RandomState(seed)
        self.NP, self.D, self.noise, self.policy_draws = NP, D, noise_kind, policy_draws
        self.stride = NP * D + 4 * NP + 8

    def _noise(self, n):
        rows = np.zeros((3, n))
        if self.noise == 1:
            rows[0] = self.rs.randn(n) if n > 1 else self.rs.randn()
        elif self.noise == 2:
            rows[0] = self.rs.rand(n) if n > 1 else self.rs.rand()
            rows[1] = self.rs.rand(n) if n > 1 else self.rs.rand()
        elif self.noise == 3:
            rows[0] = self.rs.rand(n) if n > 1 else self.rs.rand()
            rows[1] = self.rs.randn(n) if n > 1 else self.rs.randn()
            rows[2] = self.rs.randn(n) if n > 1 else self.rs.randn()
        return rows

    def reset_tape(self):
        NP, D = self.NP, self.D
        t = np.zeros(self.stride)
        t[0:NP * D] = self.rs.rand(NP, D).ravel()
        t[NP * D:NP * D + 3 * NP] = self._noise(NP).ravel()
        t[NP * D + 3 * NP:NP * D + 4 * NP] = self.rs.randint(low=0, high=4, size=NP)
        return t

    def choice_uniform(self):
        """The agent's draw, consumed BEFORE the optimizer's draws of the same step."""
        return float(self.rs.random_sample(1)[0]) if self.policy_draws else 0.

    def step_tape(self, choice_u=0.):
        t = np.zeros(self.stride)
        t[0] = self.rs.rand()
        t[1] = self.rs.rand()
        t[2:5] = self._noise(1).ravel()
        t[5] = choice_u
        return t


# ======================================================================================== classic baselines (DE / PSO / CMA-ES)
def _classic_lib():
    L = lib()
    if not getattr(L, '_classic_ready', False):
        L.orc_classic_new.restype = C.c_void_p
        L.orc_classic_new.argtypes = [C.POINTER(ProblemDesc), C.c_double, C.POINTER(AlgoCfg), C.c_uint64]
        L.orc_classic_free.argtypes = [C.c_void_p]
        L.orc_classic_reset.argtypes = [C.c_void_p]
        L.orc_classic_step.restype = C.c_int
        L.orc_classic_step.argtypes = [C.c_void_p]
        L.orc_classic_result.argtypes = [C.c_void_p, _dp, _dp, C.POINTER(C.c_int), _dp, _dp]
        L.orc_classic_population.argtypes = [C.c_void_p, _dp, _dp]
        L._classic_ready = True
    return L


class ClassicOracle:
    """DEAP_DE (algo 8) / DEAP_PSO (9) / DEAP_CMAES (10) on the CPU, Philox draws only (no reference traces exist for these)."""

    def __init__(self, desc, optimum, cfg, seed=0):
        self._st, self._keep = pack_desc(desc)
        self.cfg = cfg
        self._h = _classic_lib().orc_classic_new(C.byref(self._st), float('nan') if optimum is None else float(optimum), C.byref(cfg),
                                                 int(seed))

    def __del__(self):
        if getattr(self, '_h', None):
            _classic_lib().orc_classic_free(self._h)
            self._h = None

    def reset(self):
        _classic_lib().orc_classic_reset(self._h)

    def step(self):
        return bool(_classic_lib().orc_classic_step(self._h))

    def result(self):
        cost = np.empty(self.cfg.n_logpoint + 1)
        fes, gb, sg, n = C.c_double(), C.c_double(), C.c_double(), C.c_int()
        _classic_lib().orc_classic_result(self._h, _p(cost), C.byref(fes), C.byref(n), C.byref(gb), C.byref(sg))
        return {'cost': cost, 'fes': fes.value, 'cost_len': n.value, 'gbest': gb.value, 'sigma': sg.value}

    def population(self):
        X, c = np.empty((self.cfg.np, self.cfg.dim)), np.empty(self.cfg.np)
        _classic_lib().orc_classic_population(self._h, _p(X), _p(c))
        return X, c
