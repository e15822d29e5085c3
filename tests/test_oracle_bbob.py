"""CPU: the C oracle's BBOB restatement against the golden vectors generated from the reference."""
import numpy as np
import pytest

from helpers import load, problems
from oracle import oracle


@pytest.mark.parametrize('suite', ['bbob', 'bbob-noisy'])
@pytest.mark.parametrize('dim', [10, 30, 40])
def test_oracle_func_matches_reference_kat(suite, dim):
    kat = load('bbob_kat.npz')
    X = kat[f'x/{dim}']
    for fid, p in problems(suite, dim).items():
        f = oracle.evaluate(p.desc(), X)
        g = kat[f'f/{suite}/{dim}/{fid}']
        cost = np.abs(g - p.bias)
        assert np.all(np.abs(f - g) <= 1e-11 * np.maximum(cost, 1.0)), (suite, dim, fid)


def test_oracle_optimum_is_bias():
    for suite in ('bbob', 'bbob-noisy'):
        for fid, p in problems(suite, 10).items():
            f = oracle.evaluate(p.desc(), np.asarray(p.opt, dtype=np.float64).reshape(1, -1))
            assert f[0] == p.bias, (fid, f[0] - p.bias)


@pytest.mark.parametrize('dim', [10, 30])
def test_oracle_noise_models_match_reference(dim):
    nz = load('bbob_noise.npz')
    X = nz[f'x/{dim}']
    for fid, p in problems('bbob-noisy', dim).items():
        d = p.desc()
        for seed in (0, 1):
            draws = oracle.NumpyTapeFeeder(seed, len(X), dim, p.noise[0])._noise_rows().reshape(3, -1)
            f = oracle.apply_noise(d, p.bias, oracle.evaluate(d, X), draws)
            g = nz[f'f/{dim}/{fid}/{seed}']
            assert np.all(np.abs(f - g) <= 1e-11 * np.maximum(np.abs(g - p.bias), 1.0)), (dim, fid, seed)
        # rows at / next to the optimum take the `ftrue - optimum < 1e-8` branch: returned unchanged
        xo = np.stack([p.opt, p.opt + 1e-7])
        draws = oracle.NumpyTapeFeeder(5, 2, dim, p.noise[0])._noise_rows().reshape(3, -1)
        f = oracle.apply_noise(d, p.bias, oracle.evaluate(d, xo), draws)
        assert np.allclose(f, nz[f'fopt/{dim}/{fid}'], rtol=1e-12, atol=0), (dim, fid)


def test_philox_known_answer():
    # Random123 known-answer vectors for philox4x32-10
    assert oracle.philox(0, 0, 0, 0, 0) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    w = oracle.lib()
    import ctypes as C
    out = (C.c_uint32 * 4)()
    # counter = ffffffff x4, key = ffffffff x2
    w.orc_philox(0xffffffffffffffff, 0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff, out)
    assert list(out) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
