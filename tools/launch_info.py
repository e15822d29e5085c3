#!/usr/bin/env python
"""Print how each algorithm's generation kernel is launched (threads, LDS bytes, compile-time-geometry id) -- the inputs of the
resident-workgroups-per-CU arithmetic in DESIGN.md.   python tools/launch_info.py   (needs a GPU: batches are created)"""
import sys, numpy as np
sys.path.insert(0, '.')
from metabox_amd.problem.bbob import BBOB_Dataset
from metabox_amd.suite import Suite, Batch
from metabox_amd import _abi
tr, te = BBOB_Dataset.get_datasets('bbob', 10, 5.0); ps = sorted(tr.data + te.data, key=lambda p: p.func_id); s = Suite(ps)
for name, algo, np_ in [('RLEPSO', _abi.ALGO_RLEPSO, 100), ('LDE', _abi.ALGO_LDE, 50), ('DEDDQN', _abi.ALGO_DEDDQN, 100), ('RLPSO', _abi.ALGO_RLPSO, 100), ('GLEET', _abi.ALGO_GLEET, 100), ('QLPSO', _abi.ALGO_QLPSO, 30), ('DE', _abi.ALGO_DE, 50), ('PSO', _abi.ALGO_PSO, 50), ('CMAES', _abi.ALGO_CMAES, 50)]:
    b = Batch(s, algo, [0, 1], [1, 2], np_, 20000, 400, 50); print(name, 'd=10', b.launch_info()); b.close()
tr, te = BBOB_Dataset.get_datasets('bbob-noisy', 30, 5.0); ps = sorted(tr.data + te.data, key=lambda p: p.func_id); s = Suite(ps)
b = Batch(s, _abi.ALGO_LDE, [0, 1], [1, 2], 50, 60000, 1200, 50); print('LDE d=30', b.launch_info()); b.close()
