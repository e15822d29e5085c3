#!/usr/bin/env python
"""Kernel micro-benchmark: time k_rlepso_step alone (constant actions, no policy) per function subset.
   python tools/kbench.py [--fids 1,2,3] [--B 4096] [--steps 40]      (MBX_LIB selects an ablation build)"""
import argparse, os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metabox_amd.problem.bbob import BBOB_Dataset
from metabox_amd.suite import Suite, Batch
from metabox_amd._abi import ALGO_RLEPSO

ap = argparse.ArgumentParser()
ap.add_argument('--fids', default='all'); ap.add_argument('--B', type=int, default=4096); ap.add_argument('--steps', type=int, default=40)
ap.add_argument('--suite', default='bbob'); ap.add_argument('--dim', type=int, default=10); ap.add_argument('--each', action='store_true'); ap.add_argument('--phases', action='store_true')
a = ap.parse_args()
tr, te = BBOB_Dataset.get_datasets(a.suite, a.dim, 5.0)
ps = sorted(tr.data + te.data, key=lambda p: p.func_id)
suite = Suite(ps)
ids = [p.func_id for p in ps]
groups = [[f] for f in ids] if a.each else [ids if a.fids == 'all' else [int(x) for x in a.fids.split(',')]]
act = torch.rand(a.B, 35, generator=torch.Generator().manual_seed(0)).cuda()
for grp in groups:
    sel = [ids.index(f) for f in grp]
    pidx = np.array([sel[i % len(sel)] for i in range(a.B)], dtype=np.int32)
    b = Batch(suite, ALGO_RLEPSO, pidx, np.arange(a.B, dtype=np.uint64) + 7, 100, 2000 * a.dim, 40 * a.dim, 50, early_stop=False)
    b.reset()
    for _ in range(3): b.step(act)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if a.phases:
        import ctypes as C
        ph = (C.c_ulonglong * 16)(); b.lib.mbx_debug_phase_cycles(ph, 16, 1)
    e0.record()
    for _ in range(a.steps): b.step(act)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / a.steps * 1e3
    print(json.dumps({'lib': os.path.basename(os.environ.get('MBX_LIB', 'libmbx.so')), 'fids': grp if len(grp) < 5 else 'all', 'B': a.B, 'us_per_step': round(us, 1), 'env_steps_per_s': round(a.B / us * 1e6)}))
    if a.phases:
        b.lib.mbx_debug_phase_cycles(ph, 16, 1); v = np.array(list(ph), dtype=np.float64)
        print(json.dumps({'phase_kcycles_per_block': [round(x / a.steps / a.B / 1e3, 2) for x in v[:12]], 'share': [round(x / v.sum(), 3) for x in v[:12]]}))
    b.close()
