"""CPU: host-side logic — instance generator vs reference fixtures, config, C-ABI exports, sharding."""
import ctypes
import hashlib
import json
import os
import re

import numpy as np
import pytest

from helpers import GOLDEN, problems

ROOT = os.path.dirname(GOLDEN.rstrip('/')).rsplit('/tests', 1)[0]
ARRAY_ATTRS = ('shift', 'rotate', 'scales', 'linearTF', 'Q_rotate', 'y', 'C', 'w', 'aK', 'bK', 'f0', 'mu0')


def _digest(p):
    h = hashlib.sha256()
    for a in ARRAY_ATTRS:
        if hasattr(p, a):
            h.update(a.encode())
            h.update(np.ascontiguousarray(np.asarray(getattr(p, a), dtype=np.float64)).tobytes())
    return h.hexdigest()


@pytest.mark.parametrize('suite', ['bbob', 'bbob-noisy'])
@pytest.mark.parametrize('dim', [10, 30, 40])
def test_instance_generator_is_bit_identical_to_reference(suite, dim):
    from metabox_amd.problem.bbob import BBOB_Dataset
    ref = json.load(open(os.path.join(GOLDEN, 'bbob_instances.json')))[f'{suite}/{dim}']
    tr, te = BBOB_Dataset.get_datasets(suite, dim, 5.0)
    assert float(np.random.rand()) == ref['next_rand']          # same position in numpy's global stream
    assert [p.func_id for p in tr.data] == ref['train'] and [p.func_id for p in te.data] == ref['test']
    for p in tr.data + te.data:
        r = ref['problems'][str(p.func_id)]
        assert str(p) == r['name'] and float(p.bias) == r['bias'] and type(p).__name__ == f'F{p.func_id}'
        assert _digest(p) == r['sha256'], p.func_id


def test_difficult_split_and_dataset_surface():
    from metabox_amd.problem.bbob import BBOB_Dataset
    tr, te = BBOB_Dataset.get_datasets('bbob', 10, 5.0, difficulty='difficult')
    assert sorted(p.func_id for p in tr.data) == [1, 5, 6, 10, 15, 20] and te.N == 18
    both = tr + te
    assert both.N == 24 and len(both) == 24 and both[0] is tr.data[0]
    np.random.seed(0)
    both.shuffle()
    assert sorted(both.index.tolist()) == list(range(24))
    with pytest.raises(ValueError):
        BBOB_Dataset.get_datasets('cec', 10, 5.0)
    with pytest.raises(ValueError):
        BBOB_Dataset.get_datasets('bbob', 10, 5.0, difficulty='medium')
    with pytest.raises(AssertionError):
        BBOB_Dataset.get_datasets('bbob', 10, 4.0)


def test_config_derived_fields():
    from metabox_amd.config import get_config
    c = get_config(['--problem', 'bbob', '--dim', '10'])
    assert (c.maxFEs, c.n_logpoint, c.log_interval, c.save_interval) == (20000, 50, 400, 75000)
    assert c.t_optimizer_for_cp == ['DEAP_CMAES', 'Random_search']
    p = get_config(['--problem', 'protein'])
    assert (p.dim, p.maxFEs, p.n_logpoint, p.log_interval) == (12, 1000, 5, 200)
    t = get_config(['--train', '--train_agent', 'RLEPSO_Agent', '--train_optimizer', 'RLEPSO_Optimizer'])
    assert '/RLEPSO_Agent/' in t.agent_save_dir and t.run_time.endswith('_bbob_easy_10D')


def test_library_exports_every_symbol_declared_in_the_header():
    """No compute calls: only that libmbx.so loads and exports what include/mbx.h declares."""
    from metabox_amd import _abi
    header = open(os.path.join(ROOT, 'include', 'mbx.h')).read()
    declared = set(re.findall(r'\b(mbx_[a-z_]+)\s*\(', header))
    assert declared == set(_abi.EXPORTED_SYMBOLS), declared ^ set(_abi.EXPORTED_SYMBOLS)
    lib = _abi.load_lib()
    for name in declared:
        assert hasattr(lib, name), name
    assert b'gfx950' in lib.mbx_version()
    assert ctypes.sizeof(_abi.AlgoCfg) == 36 and _abi.AlgoCfg.flags.offset == 32 and ctypes.sizeof(_abi.ProblemDesc) == 24 + 10 * 8 + 9 * 8


def test_product_does_not_import_the_oracle():
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'metabox_amd')):
        for f in files:
            if f.endswith(('.py', '.hip', '.hpp', '.cpp', '.h')):
                src = open(os.path.join(dirpath, f)).read()
                if re.search(r'^\s*(from|import)\s+oracle\b|oracle/|liboracle', src, re.M):
                    bad.append(f)
    assert not bad, bad


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from metabox_amd import _abi
    from metabox_amd.suite import Suite
    with pytest.raises(_abi.MbxError):
        Suite(list(problems('bbob', 10).values()))


def test_shard_ranges_and_seeds():
    from metabox_amd.distributed import instance_table, philox_seed, shard_range
    for n, w in ((17920, 8), (4096, 3), (5, 8)):
        spans = [shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
    p, r = instance_table(280, 64)
    assert len(p) == 17920 and p[63] == 0 and p[64] == 1 and r[64] == 0
    s = philox_seed(r, np.arange(len(p)))
    assert len(np.unique(s)) == len(s) and s.dtype == np.uint64
