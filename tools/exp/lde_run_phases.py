"""Wall-cycle split per workgroup-generation of k_lde_run (instrumented build):  MBX_LIB=build/libmbx_phase.so python tools/exp/lde_run_phases.py [--functions 101,...]"""
import argparse, ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from metabox_amd.agent import LDE_Agent
from metabox_amd.config import get_config
from metabox_amd.problem.bbob import BBOB_Dataset
from metabox_amd.suite import Suite, Batch
from metabox_amd._abi import ALGO_LDE

ap = argparse.ArgumentParser(); ap.add_argument('--functions', default=''); ap.add_argument('--gens', type=int, default=20); ap.add_argument('--instances', type=int, default=16384)
a = ap.parse_args()
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tr, te = BBOB_Dataset.get_datasets('bbob-noisy', 30, 5.0)
ps = sorted(tr.data + te.data, key=lambda p: p.func_id)
if a.functions:
    ps = [p for p in ps if p.func_id in [int(x) for x in a.functions.split(',')]]
suite = Suite(ps)
names = ['LSTM gates', 'cell + heads', 'tile (mutation, maps, transforms)', 'Gallagher search', 'row sums + noise', 'survivors + ranking', 'features', '(tile: draws + mutation)', '(tile: first map)']
for NP in (50, 100):
    cfg = get_config(['--problem', 'bbob-noisy', '--dim', '30', '--device', 'cuda']); cfg.agent_save_dir = None
    if NP != 50: cfg.NP_override = NP
    torch.manual_seed(0)
    net = LDE_Agent(cfg).to('cuda').net
    B = a.instances
    b = Batch(suite, ALGO_LDE, np.arange(B) % len(ps), np.arange(B, dtype=np.uint64) + 7, NP, 60000, 1200, 50, early_stop=False)
    b.reset()
    h, c = torch.zeros(B, 50, device='cuda'), torch.zeros(B, 50, device='cuda')
    w = net.packed_weights()
    ph = (C.c_ulonglong * 16)()
    b.lde_rollout(w, 50, h, c, 3); torch.cuda.synchronize(); b.lib.mbx_debug_phase_cycles(ph, 16, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); b.lde_rollout(w, 50, h, c, a.gens); e1.record(); torch.cuda.synchronize()
    b.lib.mbx_debug_phase_cycles(ph, 16, 1); v = np.array(list(ph), dtype=np.float64)
    # (g_phase_cycles has 8192 block slots: with more blocks several share one, the sum is still the total)
    print(json.dumps({'LDE pop': NP, 'us_per_generation': round(e0.elapsed_time(e1) / a.gens * 1e3, 1),
                      'kcycles_per_block_generation': {k: round(x / a.gens / B / 1e3, 2) for k, x in zip(names, v)}, 'total': round(v.sum() / a.gens / B / 1e3, 1)}), flush=True)
    b.close()
