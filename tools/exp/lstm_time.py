"""k_lstm_policy alone (mbx_lde_policy) for config 3's two geometries:  python tools/exp/lstm_time.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from metabox_amd.problem.bbob import BBOB_Dataset
from metabox_amd.suite import Suite, Batch
from metabox_amd._abi import ALGO_LDE
tr, te = BBOB_Dataset.get_datasets('bbob-noisy', 30, 5.0)
ps = sorted(tr.data + te.data, key=lambda p: p.func_id)
suite = Suite(ps)
B = 16384
for NP in (50, 100):
    b = Batch(suite, ALGO_LDE, np.arange(B) % len(ps), np.arange(B, dtype=np.uint64) + 7, NP, 60000, 1200, 50, early_stop=False)
    b.reset()
    IN, H, A = NP + 10, 50, 2 * NP
    n = (IN + H) * 4 * H + 4 * H + 2 * H * A + 2 * A
    w = (torch.randn(n, generator=torch.Generator().manual_seed(1)) * 0.1).cuda()
    h = torch.zeros(B, H, device='cuda'); c = torch.zeros(B, H, device='cuda')
    for _ in range(5): b.lde_policy(w, H, h, c)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): b.lde_policy(w, H, h, c)
    e1.record(); torch.cuda.synchronize()
    t_all = e0.elapsed_time(e1) / 200 * 1e3
    e0.record()
    for _ in range(200): b.lde_policy(w, H, h, c, sample=False)
    e1.record(); torch.cuda.synchronize()
    print(json.dumps({'lib': os.path.basename(os.environ.get('MBX_LIB', 'libmbx.so')), 'pop': NP, 'k_lstm_policy_us': round(t_all, 1), 'without_sampling_us': round(e0.elapsed_time(e1) / 200 * 1e3, 1)}))
    b.close()
