"""How many (rank, coordinate pair) items does the exact FDR scan flag for its second pass?  Needs a library built with -DMBX_FDR_COUNT (the count travels in words 2 / 3 of the
mbx_debug_clock_slots block):  MBX_LIB=build/libmbx_count.so python tools/exp/fdr_flag_rate.py"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from metabox_amd._abi import ALGO_RLEPSO                      # noqa: E402
from metabox_amd.agent import RLEPSO_Agent                    # noqa: E402
from metabox_amd.config import get_config                     # noqa: E402
from metabox_amd.problem.bbob import BBOB_Dataset              # noqa: E402
from metabox_amd.suite import Batch, Suite                    # noqa: E402

tr, te = BBOB_Dataset.get_datasets('bbob', 10, 5.0)
ps = sorted(tr.data + te.data, key=lambda p: p.func_id)
cfg = get_config(['--problem', 'bbob', '--dim', '10', '--device', 'cuda'])
cfg.agent_save_dir = None
agent = RLEPSO_Agent(cfg).load_exported_weights(np.load(os.path.join(ROOT, 'metabox_amd', 'agent_model', 'rlepso_bbob_easy.npz'))).to('cuda')
actor = agent.actor
h1, h2 = actor.hidden_sizes()
for fids in (None, (3,), (5,), (12,), (21,), (22,)):
    sel = ps if fids is None else [p for p in ps if p.func_id in fids]
    B = 4096
    b = Batch(Suite(sel), ALGO_RLEPSO, np.arange(B) % len(sel), np.arange(B, dtype=np.uint64) + 1, 100, 20000, 400, 50, early_stop=False)
    table = b.policy_table(actor.packed_weights(), h1, h2, actor.min_sigma, actor.max_sigma)
    b.reset()
    done = 0
    for n in (5, 20, 25, 50, 50, 49):
        slots = torch.zeros(8, dtype=torch.int64, device='cuda')
        b.lib.mbx_debug_clock_slots(b._h, C.c_void_p(slots.data_ptr()))
        b.rlepso_rollout(table, n)
        torch.cuda.synchronize()
        v = slots.cpu().numpy()
        done += n
        print(f'functions {fids or "all 24"}: generations {done - n + 1}-{done}: {v[2]} of {v[3]} items flagged ({100. * v[2] / max(v[3], 1):.3f} %); flagged coordinates: {v[4]} copies of the winner only, {v[5]} with another candidate in the band, {v[6]} nothing near the winner; clock {v[0] / max(v[1], 1) / 10:.3f} GHz', flush=True)
    b.close()
