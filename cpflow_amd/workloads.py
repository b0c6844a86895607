"""The five BASELINE.json configurations as data: (layer, num_cp_gates, target, samples per GPU, GPUs).

Shared by bench.py (--config), the GPU tests and tools/gpu.py so that every measured or tested launch of a
"config N" is the same workload.  All use rotation_gates='xyz', num_gd_iterations=2000, learning_rate=0.1,
r=0.00055, random_seed=0 (StaticOptions defaults, cpflow/main.py:338-380 and the README example).
"""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CCZ = np.diag([1, 1, 1, 1, 1, 1, 1, -1]).astype(np.complex128)
TOFF3 = np.eye(8)[[0, 1, 2, 3, 4, 5, 7, 6]].astype(np.complex128)


def qft(n):
    N = 1 << n
    w = np.exp(2j * np.pi / N)
    return np.array([[w ** (j * k) for k in range(N)] for j in range(N)], dtype=np.complex128) / np.sqrt(N)


def su64():
    """Config 5's target: scipy.stats.unitary_group.rvs(64, random_state=0) with the determinant divided out,
    committed as a fixture (tools/make_su64.py)."""
    return np.load(os.path.join(ROOT, 'tests', 'golden', 'su64_seed0.npy'))


class Workload:
    def __init__(self, name, title, layer, num_cp_gates, target, samples_per_gpu, gpus):
        self.name, self.title, self.layer, self.num_cp_gates = name, title, layer, num_cp_gates
        self._target, self.samples_per_gpu, self.gpus = target, samples_per_gpu, gpus
        self.rotation_gates, self.num_gd_iterations, self.r, self.seed = 'xyz', 2000, 0.00055, 0

    @property
    def target(self):
        return self._target() if callable(self._target) else self._target

    @property
    def num_qubits(self):
        return max(max(p) for p in self.layer) + 1


CONFIGS = {
    'cfg1': Workload('cfg1', 'CCZ 3q, chain [[0,1],[1,2]]', [[0, 1], [1, 2]], 12, CCZ, 10, 1),
    'cfg2': Workload('cfg2', 'CCZ 3q, chain [[0,1],[1,2]]', [[0, 1], [1, 2]], 12, CCZ, 1024, 1),
    'cfg3': Workload('cfg3', 'Toffoli 3q, all-to-all [[0,1],[0,2],[1,2]]', [[0, 1], [0, 2], [1, 2]], 16, TOFF3, 512, 8),
    'cfg4': Workload('cfg4', 'QFT 4q, chain [[0,1],[1,2],[2,3]]', [[0, 1], [1, 2], [2, 3]], 24, lambda: qft(4), 2048, 1),
    'cfg5': Workload('cfg5', 'random SU(64) 6q (tests/golden/su64_seed0.npy), ring', [[i, (i + 1) % 6] for i in range(6)], 60,
                     su64, 1024, 8),
}
