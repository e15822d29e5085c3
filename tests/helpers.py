"""Shared helpers of the test-suite (problem lookup, tolerances)."""
import functools
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')

# Tolerance of every floating-point parity check on costs (north star: 1e-5 relative on fitness).
# Costs are (f_raw + bias) - optimum with bias up to 2500, i.e. quantised at ulp(bias) <= 4.6e-13, so a
# relative bound alone is meaningless next to the 1e-8 stop threshold: the absolute floor is 8 ulp(2500).
RTOL = 1e-5
ATOL = 8 * 4.6e-13


def close(a, b, rtol=RTOL, atol=ATOL):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.all(np.abs(a - b) <= atol + rtol * np.abs(b))


@functools.lru_cache(maxsize=None)
def problems(suite, dim):
    """{func_id: problem} for train+test of `suite` at `dim` (host-side instance generator)."""
    from metabox_amd.problem.bbob import BBOB_Dataset
    tr, te = BBOB_Dataset.get_datasets(suite, dim, 5.0)
    return {p.func_id: p for p in tr.data + te.data}


def load(name):
    return np.load(os.path.join(GOLDEN, name))
