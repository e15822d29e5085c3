"""Wall-cycle split per workgroup of config 3 (k_lde_step, pop 50 / 100) and config 5 (k_rlepso_step<1024, 128, 40, 5>, one launch per generation)
with an instrumented build:  MBX_LIB=build/libmbx_dqphase64.so python tools/exp/c35_phases.py"""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from metabox_amd.problem.bbob import BBOB_Dataset
from metabox_amd.suite import Suite, Batch
from metabox_amd._abi import ALGO_LDE, ALGO_RLEPSO

def phases(b, run, n, names):
    ph = (C.c_ulonglong * 16)()
    run(3); torch.cuda.synchronize(); b.lib.mbx_debug_phase_cycles(ph, 16, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(n); e1.record(); torch.cuda.synchronize()
    b.lib.mbx_debug_phase_cycles(ph, 16, 1); v = np.array(list(ph), dtype=np.float64)
    return {'us_per_step': round(e0.elapsed_time(e1) / n * 1e3, 1), 'kcycles_per_block': {k: round(x / n / b.B / 1e3, 2) for k, x in zip(names, v)}, 'total': round(v.sum() / n / b.B / 1e3, 1)}

tr, te = BBOB_Dataset.get_datasets('bbob-noisy', 30, 5.0)
ps = sorted(tr.data + te.data, key=lambda p: p.func_id)
suite = Suite(ps)
for NP in (50, 100):
    B = 16384
    act = torch.rand(B, 2 * NP, generator=torch.Generator().manual_seed(0)).cuda()
    b = Batch(suite, ALGO_LDE, np.arange(B) % len(ps), np.arange(B, dtype=np.uint64) + 7, NP, 60000, 1200, 50, early_stop=False)
    b.reset()
    print(json.dumps({'config 3 LDE pop': NP, **phases(b, lambda n: [b.step(act) for _ in range(n)], 20, ['staging+draws', 'histogram', 'mutation', 'evaluation', 'selection', 'sort+emit'])}), flush=True)
    b.close()
ps = []
for s_ in ('bbob', 'bbob-noisy'):
    tr, te = BBOB_Dataset.get_datasets(s_, 40, 5.0)
    ps += sorted(tr.data + te.data, key=lambda p: p.func_id)
suite = Suite(ps)
B = 8192
act = torch.rand(B, 35, generator=torch.Generator().manual_seed(0)).cuda()
names = ['staging', 'ranking+draws', 'pbest staging', 'move (FDR + velocity)', 'evaluation', 'commit', 'reinit draw', 'reinit', 'write-back']
for sel, tag in ((None, 'all 54'), ([5], 'F5'), ([1], 'F1'), ([15], 'F15'), ([21], 'F21')):
    idx = [k for k, p in enumerate(ps) if sel is None or p.func_id in sel]
    b = Batch(suite, ALGO_RLEPSO, np.array([idx[i % len(idx)] for i in range(B)]), np.arange(B, dtype=np.uint64) + 3, 128, 80000, 1600, 50, early_stop=False)
    b.reset()
    print(json.dumps({'config 5 functions': tag, **phases(b, lambda n: [b.step(act) for _ in range(n)], 10, names)}), flush=True)
    b.close()
