import os, sys, json, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from metabox_amd._abi import ALGO_RLEPSO
from metabox_amd.problem.bbob import BBOB_Dataset
from metabox_amd.suite import Batch, Suite
ps = []
for suite in ('bbob', 'bbob-noisy'):
    tr, te = BBOB_Dataset.get_datasets(suite, 40, 5.0)
    ps += sorted(tr.data + te.data, key=lambda p: p.func_id)
s = Suite(ps)
B = 8192
b = Batch(s, ALGO_RLEPSO, np.arange(B) % len(ps), np.arange(B, dtype=np.uint64) + 3, 128, 80000, 1600, 50, early_stop=False)
act = torch.rand(B, 35, generator=torch.Generator().manual_seed(0)).cuda()
b.reset()
for _ in range(2): b.step(act)
torch.cuda.synchronize()
ph = (C.c_ulonglong * 16)(); b.lib.mbx_debug_phase_cycles(ph, 16, 1)
for _ in range(8): b.step(act)
torch.cuda.synchronize()
b.lib.mbx_debug_phase_cycles(ph, 16, 1); v = np.array(list(ph), dtype=np.float64)
print(json.dumps({'phase_kcycles_per_block': [round(x / 8 / B / 1e3, 2) for x in v[:10]], 'share': [round(x / v.sum(), 3) for x in v[:10]]}))
