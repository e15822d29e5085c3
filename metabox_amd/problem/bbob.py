"""BBOB (F1-F24) and noisy-BBOB (F101-F130) problem instances — host side.

Host logic only: this module draws the per-instance parameters (shift, rotation,
bias and the per-function extras) exactly as the reference's dataset builder
does, so that the same ``instance_seed`` yields the same problem instances
(reference: src/problem/bbob.py:904-952 for the dataset loop and the per-class
constructors cited in ``_build_*`` below).  The objective itself is **not**
computed here: ``func`` packs the instance into an ``mbx_problem_desc`` and
evaluates it on the GPU through the C-ABI (``mbx_eval``, include/mbx.h).

Layout of one instance for the device (see ``BBOB_Problem.desc``): a base
function kind 1..24, up to two D x D linear maps (``m1`` applied to ``x - dshift``,
``m2`` applied after the element-wise transforms), up to three per-dimension
constant vectors (pre-evaluated with numpy so the device only multiplies), the
Gallagher peak tables, a handful of scalars, and the noise model.
"""
import numpy as np

from .basic_problem import Basic_Problem

# --------------------------------------------------------------------------------------
# catalogue: function id -> (base kind, name, options)
# --------------------------------------------------------------------------------------
NOISE_NONE, NOISE_GAUSS, NOISE_UNIFORM, NOISE_CAUCHY = 0, 1, 2, 3

_BASE_NAMES = {
    1: 'Sphere', 2: 'Ellipsoidal', 3: 'Rastrigin', 4: 'Buche_Rastrigin', 5: 'Linear_Slope',
    6: 'Attractive_Sector', 7: 'Step_Ellipsoidal', 8: 'Rosenbrock_original', 9: 'Rosenbrock_rotated',
    10: 'Ellipsoidal_high_cond', 11: 'Discus', 12: 'Bent_Cigar', 13: 'Sharp_Ridge', 14: 'Different_Powers',
    15: 'Rastrigin_F15', 16: 'Weierstrass', 17: 'Schaffers', 18: 'Schaffers_high_cond',
    19: 'Composite_Grie_rosen', 20: 'Schwefel', 21: 'Gallagher_101Peaks', 22: 'Gallagher_21Peaks',
    23: 'Katsuura', 24: 'Lunacek_bi_Rastrigin',
}

# boundaryHandling coefficient of the noise-free classes that have one
# (reference: bbob.py F1:164, F7:326, F8:371, F10:441, F14:563, F17:648, F18:656, F19:713, F21:801, F22:809)
_PEN_COEF = {1: 0.0, 7: 1.0, 8: 0.0, 10: 0.0, 14: 0.0, 17: 10.0, 18: 10.0, 19: 0.0, 21: 1.0, 22: 1.0}


def _noisy_family(first_id, kind, stem, severities):
    """Three consecutive ids (gauss, uniform, cauchy) per severity (reference: bbob.py:170-207 etc.)."""
    out = {}
    fid = first_id
    for sev in severities:
        mod = sev == 'moderate'
        tag = 'moderate_' if mod else ''
        out[fid] = (kind, f'{stem}_{tag}gauss', (NOISE_GAUSS, 0.01 if mod else 1.0, 0.0))
        out[fid + 1] = (kind, f'{stem}_{tag}uniform', (NOISE_UNIFORM, 0.01 if mod else 1.0, 0.01 if mod else 1.0))
        out[fid + 2] = (kind, f'{stem}_{tag}cauchy', (NOISE_CAUCHY, 0.01 if mod else 1.0, 0.05 if mod else 0.2))
        fid += 3
    return out


_NOISY = {}
_NOISY.update(_noisy_family(101, 1, 'Sphere', ['moderate']))
_NOISY.update(_noisy_family(104, 8, 'Rosenbrock', ['moderate']))
_NOISY.update(_noisy_family(107, 1, 'Sphere', ['severe']))
_NOISY.update(_noisy_family(110, 8, 'Rosenbrock', ['severe']))
_NOISY.update(_noisy_family(113, 7, 'Step_Ellipsoidal', ['severe']))
_NOISY.update(_noisy_family(116, 10, 'Ellipsoidal', ['severe']))
_NOISY.update(_noisy_family(119, 14, 'Different_Powers', ['severe']))
_NOISY.update(_noisy_family(122, 17, 'Schaffers', ['severe']))
_NOISY.update(_noisy_family(125, 19, 'Composite_Grie_rosen', ['severe']))
_NOISY.update(_noisy_family(128, 21, 'Gallagher_101Peaks', ['severe']))

BBOB_IDS = list(range(1, 25))
NOISY_IDS = list(range(101, 131))
# the small split (reference: bbob.py:917-921); "easy" => the small split is the TEST set (bbob.py:948)
SMALL_SPLIT = {'bbob': [1, 5, 6, 10, 15, 20],
               'bbob-noisy': [101, 105, 115, 116, 117, 119, 120, 125]}


def function_spec(func_id):
    """-> (kind, name, (noise_kind, a, b))."""
    if func_id in _BASE_NAMES:
        return func_id, _BASE_NAMES[func_id], (NOISE_NONE, 0.0, 0.0)
    if func_id in _NOISY:
        return _NOISY[func_id]
    raise ValueError(f'F{func_id} is not a BBOB / noisy-BBOB function id.')


# --------------------------------------------------------------------------------------
# random rotation (reference: bbob.py:11-28) — Householder product with det = +1
# --------------------------------------------------------------------------------------
def rotate_gen(dim):
    rng = np.random
    acc = np.eye(dim)
    signs = np.ones((dim,))
    for n in range(1, dim):
        v = rng.normal(size=(dim - n + 1,))
        signs[n - 1] = np.sign(v[0])
        v[0] -= signs[n - 1] * np.sqrt((v * v).sum())
        reflect = np.eye(dim)
        reflect[n - 1:, n - 1:] = np.eye(dim - n + 1) - 2. * np.outer(v, v) / (v * v).sum()
        acc = np.dot(acc, reflect)
    signs[-1] = (-1) ** (1 - (dim % 2)) * signs.prod()
    return (signs * acc.T).T


def _cond_scales(base, dim):
    return (base ** 0.5) ** np.linspace(0, 1, dim)


class BBOB_Problem(Basic_Problem):
    """One BBOB / noisy-BBOB instance.

    Public attributes follow the reference (``dim, shift, rotate, bias, lb, ub, FES, opt,
    optimum`` plus the per-function extras ``scales, linearTF, Q_rotate, y, C, w, aK, bK,
    f0, mu0``).  ``optimum`` is f(shift) evaluated by the device at suite upload
    (reference computes it in the constructor, bbob.py:42).
    """

    def __init__(self, func_id, dim, shift, rotate, bias, lb, ub):
        self.func_id = int(func_id)
        self.kind, self.name, self.noise = function_spec(func_id)
        self.dim = dim
        self.bias = bias
        self.lb = lb
        self.ub = ub
        self.FES = 0
        self.T1 = 0
        self._suite = None          # set when the instance is uploaded (metabox_amd.suite.Suite)
        self._suite_index = None
        self._optimum = None
        # constructor recipe of the base function; may draw from np.random and rewrite shift/rotate
        shift, rotate = getattr(self, f'_build_{self.kind}', self._build_plain)(dim, shift, rotate, lb, ub)
        self.shift = shift
        self.rotate = rotate
        self.opt = self.shift

    # -- constructor recipes -----------------------------------------------------------------
    def _build_plain(self, dim, shift, rotate, lb, ub):        # F1,F2,F10,F11,F12,F14 (bbob.py:149,210,438,488,505,543)
        return shift, rotate

    def _build_3(self, dim, shift, rotate, lb, ub):            # bbob.py:233-235
        self.scales = _cond_scales(10., dim)
        return shift, rotate

    def _build_4(self, dim, shift, rotate, lb, ub):            # bbob.py:250-253 (rewrites the caller's shift in place)
        shift[::2] = np.abs(shift[::2])
        self.scales = _cond_scales(10., dim)
        return shift, rotate

    def _build_5(self, dim, shift, rotate, lb, ub):            # bbob.py:272-276
        shift = np.sign(shift)
        zero = shift == 0.
        shift[zero] = np.random.choice([-1., 1.], size=zero.sum())
        return shift * ub, rotate

    def _second_rotation_left(self, dim, rotate, base):        # Q . diag(scales) . R   (bbob.py:294-296, 529-531, 847-849, 873-877)
        return np.matmul(np.matmul(rotate_gen(dim), np.diag(_cond_scales(base, dim))), rotate)

    def _build_6(self, dim, shift, rotate, lb, ub):
        return shift, self._second_rotation_left(dim, rotate, 10.)

    def _build_7(self, dim, shift, rotate, lb, ub):            # bbob.py:314-318
        rotate = np.matmul(np.diag(_cond_scales(10., dim)), rotate)
        self.Q_rotate = rotate_gen(dim)
        return shift, rotate

    def _build_8(self, dim, shift, rotate, lb, ub):            # bbob.py:360-363 (in-place scaling of the caller's shift)
        shift *= 0.75
        return shift, np.eye(dim)

    def _build_9(self, dim, shift, rotate, lb, ub):            # bbob.py:423-427
        scale = max(1., dim ** 0.5 / 8.)
        self.linearTF = scale * rotate
        return np.matmul(0.5 * np.ones(dim), self.linearTF) / (scale ** 2), rotate

    def _build_13(self, dim, shift, rotate, lb, ub):
        return shift, self._second_rotation_left(dim, rotate, 10)

    def _build_15(self, dim, shift, rotate, lb, ub):           # bbob.py:589-592
        self.linearTF = np.matmul(np.matmul(rotate, np.diag(_cond_scales(10., dim))), rotate_gen(dim))
        return shift, rotate

    def _build_16(self, dim, shift, rotate, lb, ub):           # bbob.py:609-615
        self.linearTF = np.matmul(np.matmul(rotate, np.diag(_cond_scales(0.01, dim))), rotate_gen(dim))
        self.aK = 0.5 ** np.arange(12)
        self.bK = 3.0 ** np.arange(12)
        self.f0 = np.sum(self.aK * np.cos(np.pi * self.bK))
        return shift, rotate

    def _schaffer(self, dim, shift, rotate, cond):             # bbob.py:634-637
        self.condition = cond
        self.linearTF = np.matmul(np.diag(_cond_scales(cond, dim)), rotate_gen(dim))
        return shift, rotate

    def _build_17(self, dim, shift, rotate, lb, ub):
        return self._schaffer(dim, shift, rotate, 10.)

    def _build_18(self, dim, shift, rotate, lb, ub):
        return self._schaffer(dim, shift, rotate, 1000.)

    def _build_19(self, dim, shift, rotate, lb, ub):           # bbob.py:694-698
        scale = max(1., dim ** 0.5 / 8.)
        self.linearTF = scale * rotate
        return np.matmul(0.5 * np.ones(dim) / (scale ** 2.), self.linearTF), rotate

    def _build_20(self, dim, shift, rotate, lb, ub):           # bbob.py:744-746
        return 0.5 * 4.2096874633 * np.random.choice([-1., 1.], size=dim), rotate

    def _gallagher(self, dim, shift, rotate, lb, ub, n_peaks):  # bbob.py:768-794
        shrink, alpha0 = {101: (1., 1e3), 21: (0.98, 1e6)}[n_peaks]
        self.n_peaks = n_peaks
        self.y = shrink * (np.random.rand(n_peaks, dim) * (ub - lb) + lb)
        self.y[0] = shift * shrink
        root_alpha = 1000 ** np.random.permutation(np.linspace(0, 1, n_peaks - 1))
        root_alpha = np.insert(root_alpha, obj=0, values=np.sqrt(alpha0))
        self.C = np.vstack([np.random.permutation(root_alpha[k] ** np.linspace(-0.5, 0.5, dim))
                            for k in range(n_peaks)])
        self.w = np.insert(np.linspace(1.1, 9.1, n_peaks - 1), 0, 10.)
        return self.y[0], rotate

    def _build_21(self, dim, shift, rotate, lb, ub):
        return self._gallagher(dim, shift, rotate, lb, ub, 101)

    def _build_22(self, dim, shift, rotate, lb, ub):
        return self._gallagher(dim, shift, rotate, lb, ub, 21)

    def _build_23(self, dim, shift, rotate, lb, ub):
        return shift, self._second_rotation_left(dim, rotate, 100.)

    def _build_24(self, dim, shift, rotate, lb, ub):           # bbob.py:873-877
        self.mu0 = 2.5 / 5 * ub
        shift = np.random.choice([-1., 1.], size=dim) * self.mu0 / 2
        return shift, self._second_rotation_left(dim, rotate, 100)

    # -- plugin surface ------------------------------------------------------------------------
    def __str__(self):
        return self.name

    def get_optimal(self):
        return self.opt

    @property
    def condition_number(self):
        """Conditioning of the ellipsoid family: 1e6 for F10, 1e4 for F116-F118 (bbob.py:450,457-475)."""
        return 1e6 if self.func_id == 10 else 1e4

    @property
    def pen_coef(self):
        """Coefficient of the boundary penalty added by ``boundaryHandling``.

        Noisy classes inherit NoisyProblem.boundaryHandling = 100 * pen (bbob.py:104-105) ahead of the
        base class in the MRO; the noise-free ones define their own.  Kinds with an in-line penalty term
        (F4, F16, F20, F23, F24) handle it inside the kernel and report 0 here.
        """
        if self.noise[0] != NOISE_NONE:
            return 100.0
        return _PEN_COEF.get(self.kind, 0.0)

    def desc(self):
        """Flat description of the instance for ``mbx_suite_create`` (include/mbx.h: mbx_problem_desc)."""
        D = self.dim
        k = self.kind
        lin = np.linspace(0, 1, D)
        idx = np.arange(D)
        zeros = np.zeros(D)
        d = dict(func_id=self.func_id, kind=k, dim=D, n_peaks=0, bias=float(self.bias), lb=float(self.lb),
                 ub=float(self.ub), pen_coef=float(self.pen_coef), s=[0.0, 0.0, 0.0, 0.0],
                 noise_kind=int(self.noise[0]), noise_a=float(self.noise[1]), noise_b=float(self.noise[2]),
                 dshift=np.asarray(self.shift, dtype=np.float64), m1=np.asarray(self.rotate, dtype=np.float64),
                 m2=None, v0=None, v1=None, v2=None, py=None, pc=None, pw=None)
        if k == 2:
            d['v0'] = np.power(10, 6 * idx / (D - 1))
        elif k == 3:
            d['v0'] = self.scales
            d['v1'] = 0.2 * lin
        elif k == 4:
            d['v0'] = self.scales
        elif k == 5:
            s = np.sign(self.shift) * (10 ** lin)
            d['v0'] = s
            d['v1'] = self.ub * np.abs(s)
        elif k == 7:
            d['m2'] = self.Q_rotate
            d['v0'] = 100 ** lin
        elif k == 8:
            d['s'][0] = max(1., D ** 0.5 / 8.)
        elif k in (9, 19):
            d['m1'] = self.linearTF
            d['dshift'] = zeros
            if k == 19:
                d['s'][0] = 10.0 if self.func_id == 19 else 1.0      # factor (bbob.py:708,717-735)
        elif k == 10:
            d['v0'] = self.condition_number ** (idx / (D - 1))
        elif k == 12:
            d['v1'] = 0.5 * lin
        elif k == 14:
            d['v0'] = 2 + 4 * idx / max(1, D - 1)
        elif k == 15:
            d['m2'] = self.linearTF
            d['v1'] = 0.2 * lin
        elif k == 16:
            d['m2'] = self.linearTF
            d['s'][0] = float(self.f0)
        elif k in (17, 18):
            d['m2'] = self.linearTF
            d['v1'] = 0.5 * lin
        elif k == 20:
            d['v0'] = _cond_scales(10, D)
            d['v1'] = 2 * np.abs(self.shift)
            d['v2'] = 2 * np.sign(self.shift)
        elif k in (21, 22):
            d['n_peaks'] = self.n_peaks
            d['py'], d['pc'], d['pw'] = self.y, self.C, self.w
        elif k == 24:
            s = 1. - 1. / (2. * np.sqrt(D + 20.) - 8.2)
            d['v0'] = 2. * np.sign(self.shift)
            d['s'][0] = float(self.mu0)
            d['s'][1] = float(s)
            d['s'][2] = float(-np.sqrt((self.mu0 ** 2 - 1) / s))
        for key in ('dshift', 'm1', 'm2', 'v0', 'v1', 'v2', 'py', 'pc', 'pw'):
            if d[key] is not None:
                d[key] = np.ascontiguousarray(d[key], dtype=np.float64)
        return d

    # evaluation goes through the device suite -------------------------------------------------
    def _bound_suite(self):
        if self._suite is None:
            from ..suite import Suite
            Suite([self])           # binds itself to the problem
        return self._suite

    @property
    def optimum(self):
        if self._optimum is None:
            self._optimum = self._bound_suite().optimum(self._suite_index)
        return self._optimum

    def func(self, x):
        """Noise-free objective on the device (``F*.func``, bias included)."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        self.FES += x.shape[0]
        return self._bound_suite().eval(self._suite_index, x, noisy=False)

    def eval(self, x):
        """``Basic_Problem.eval`` wrapped by the noise model for F101-F130 (bbob.py:100-102)."""
        if self.noise[0] == NOISE_NONE:
            return super().eval(x)
        import time
        t0 = time.perf_counter()
        x = np.asarray(x, dtype=np.float64)
        single = x.ndim == 1
        x2 = x.reshape(1, -1) if single else x.reshape(-1, x.shape[-1])
        self.FES += x2.shape[0]
        y = self._bound_suite().eval(self._suite_index, np.ascontiguousarray(x2), noisy=True)
        self.T1 += (time.perf_counter() - t0) * 1000
        return y[0] if single else y


def _make_class(fid):
    def __init__(self, dim, shift, rotate, bias, lb, ub):
        BBOB_Problem.__init__(self, fid, dim, shift, rotate, bias, lb, ub)
    return type(f'F{fid}', (BBOB_Problem,), {'__init__': __init__, '__doc__': function_spec(fid)[1]})


# F1..F24, F101..F130 are importable by name like in the reference (the reference's registry is eval(f'F{id}'))
for _fid in BBOB_IDS + NOISY_IDS:
    globals()[f'F{_fid}'] = _make_class(_fid)
del _fid


class BBOB_Dataset:
    """List-like problem set with the reference's ``N / data / shuffle / __getitem__ / __add__`` surface
    (reference: bbob.py:893-989)."""

    def __init__(self, data, batch_size=1):
        self.data = data
        self.batch_size = batch_size
        self.N = len(self.data)
        self.ptr = list(range(0, self.N, batch_size))
        self.index = np.arange(self.N)

    @staticmethod
    def get_datasets(suit, dim, upperbound, shifted=True, rotated=True, biased=True,
                     train_batch_size=1, test_batch_size=1, difficulty='easy', instance_seed=3849):
        if suit == 'bbob':
            ids = BBOB_IDS
        elif suit == 'bbob-noisy':
            ids = NOISY_IDS
        else:
            raise ValueError(f'{suit} function suit is invalid or is not supported yet.')
        if difficulty not in ('easy', 'difficult'):
            raise ValueError(f'{difficulty} difficulty is invalid.')
        small = SMALL_SPLIT[suit]
        if instance_seed > 0:
            np.random.seed(instance_seed)
        assert upperbound >= 5., f'Argument upperbound must be at least 5, but got {upperbound}.'
        ub, lb = upperbound, -upperbound
        train, test = [], []
        for fid in ids:
            shift = 0.8 * (np.random.random(dim) * (ub - lb) + lb) if shifted else np.zeros(dim)
            rot = rotate_gen(dim) if rotated else np.eye(dim)
            bias = np.random.randint(1, 26) * 100 if biased else 0
            inst = globals()[f'F{fid}'](dim=dim, shift=shift, rotate=rot, bias=bias, lb=lb, ub=ub)
            in_small = fid in small
            (train if (difficulty == 'easy') != in_small else test).append(inst)
        return BBOB_Dataset(train, train_batch_size), BBOB_Dataset(test, test_batch_size)

    def __getitem__(self, item):
        if self.batch_size < 2:
            return self.data[self.index[item]]
        lo = self.ptr[item]
        return [self.data[j] for j in self.index[lo: min(lo + self.batch_size, self.N)]]

    def __len__(self):
        return self.N

    def __add__(self, other):
        return BBOB_Dataset(self.data + other.data, self.batch_size)

    def shuffle(self):
        self.index = np.random.permutation(self.N)
