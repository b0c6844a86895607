"""CPU-only timing of the stage after the raw batch in Synthesize.static(): building one
Decomposition per verified sample (circuit from angles -> CP->CZ -> ZXZ merge -> unitary -> loss).
Usage: python tools/micro/post_stage_cpu.py [num_qubits] [num_cp_gates] [count]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
from cpflow_amd.ansatz import Ansatz
from cpflow_amd.synthesis import Decomposition, HSTLoss
from cpflow_amd.topology import connected_layer, fill_layers

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 4
K = int(sys.argv[2]) if len(sys.argv) > 2 else 14
count = int(sys.argv[3]) if len(sys.argv) > 3 else 61
anz = Ansatz(nq, 'cp', placements=fill_layers(connected_layer(nq), K))
rng = np.random.default_rng(0)
angles = rng.uniform(0, 2 * np.pi, size=(count, anz.num_angles))
loss = HSTLoss(np.eye(2 ** nq))
t0 = time.perf_counter()
for a in angles:
    Decomposition._from_cp_circuit(loss, anz.unitary, anz.circuit, a, 'x')
dt = time.perf_counter() - t0
print(f'{nq} qubits, {K} CP gates: {count} decompositions in {dt * 1e3:.1f} ms ({dt / count * 1e3:.2f} ms each)')
