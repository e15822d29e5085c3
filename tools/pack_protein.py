#!/usr/bin/env python
"""Pack MetaBox's protein-docking input tensors (280 complexes x {coor_init, q, e, r, basis, eigval}, plain-text
files under src/problem/protein_docking_data/) into one float64 .npz that ships with the package.

These files are benchmark INPUT DATA (atom coordinates, charges, Lennard-Jones parameters, normal-mode basis), not
code.  Usage: python tools/pack_protein.py [/path/to/protein_docking_data] [out.npz]
"""
import os
import sys

import numpy as np

src = sys.argv[1] if len(sys.argv) > 1 else '/root/reference/src/problem/protein_docking_data'
out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                         'metabox_amd', 'problem', 'protein_docking_data.npz')
ids = sorted(d for d in os.listdir(src) if os.path.isdir(os.path.join(src, d)))
fields = {k: [] for k in ('coor_init', 'q', 'e', 'r', 'basis', 'eigval')}
for pid in ids:
    for k in fields:
        fields[k].append(np.loadtxt(os.path.join(src, pid, k)))
data = {k: np.stack(v) for k, v in fields.items()}
data['ids'] = np.array(ids)
np.savez_compressed(out, **data)
print(out, {k: v.shape for k, v in data.items()}, os.path.getsize(out) / 1e6, 'MB')
