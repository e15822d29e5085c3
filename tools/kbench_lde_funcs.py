#!/usr/bin/env python
"""k_lde_step alone (constant actions, no policy) per noisy-function group at D = 30, NP = 50:  python tools/kbench_lde_funcs.py"""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metabox_amd.problem.bbob import BBOB_Dataset
from metabox_amd.suite import Suite, Batch
from metabox_amd._abi import ALGO_LDE

tr, te = BBOB_Dataset.get_datasets('bbob-noisy', 30, 5.0)
ps = sorted(tr.data + te.data, key=lambda p: p.func_id)
suite = Suite(ps)
ids = [p.func_id for p in ps]
B, NP = 16384, 50
act = torch.rand(B, 2 * NP, generator=torch.Generator().manual_seed(0)).cuda()
import ctypes as C
PH = '--phases' in sys.argv
groups = {'all 30': ids, 'F128-130 Gallagher': [128, 129, 130], 'without Gallagher': [i for i in ids if i < 128]}
for k in range(101, 128, 3):
    groups[f'F{k}-{k + 2} {ps[ids.index(k)]}'] = [k, k + 1, k + 2]
only = [a.split('=', 1)[1] for a in sys.argv if a.startswith('--only=')]
for name, grp in groups.items():
    if only and not any(o in name for o in only): continue
    sel = [ids.index(f) for f in grp]
    pidx = np.array([sel[i % len(sel)] for i in range(B)], dtype=np.int32)
    b = Batch(suite, ALGO_LDE, pidx, np.arange(B, dtype=np.uint64) + 7, NP, 60000, 1200, 50, early_stop=False)
    b.reset()
    for _ in range(3): b.step(act)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if PH:
        ph = (C.c_ulonglong * 16)(); b.lib.mbx_debug_phase_cycles(ph, 16, 1)
    e0.record()
    for _ in range(20): b.step(act)
    e1.record(); torch.cuda.synchronize()
    print(json.dumps({'functions': name, 'us_per_step': round(e0.elapsed_time(e1) / 20 * 1e3, 1)}))
    if PH:
        b.lib.mbx_debug_phase_cycles(ph, 16, 1); v = np.array(list(ph), dtype=np.float64)
        print(json.dumps({'phase_kcycles_per_block': [round(x / 20 / B / 1e3, 2) for x in v[:8]], 'share': [round(x / v.sum(), 3) for x in v[:8]]}))
    b.close()
