"""BASELINE config 5's target: one fixed random SU(64), committed as tests/golden/su64_seed0.npy.

    python tools/make_su64.py        # scipy.stats.unitary_group.rvs(64, random_state=0), determinant divided out

Run where scipy exists (the build container); the GPU box only reads the committed file."""
import os

import numpy as np
from scipy.stats import unitary_group

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
u = unitary_group.rvs(64, random_state=0)
u = u / np.linalg.det(u) ** (1.0 / 64)
assert np.abs(u.conj().T @ u - np.eye(64)).max() < 1e-12 and abs(np.linalg.det(u) - 1) < 1e-10
np.save(os.path.join(ROOT, 'tests', 'golden', 'su64_seed0.npy'), u.astype(np.complex128))
print('su64_seed0.npy', u.shape, 'sha-like checksum', float(np.abs(u).sum()), complex(u[0, 0]))
