"""CPU post-processing of a decomposition's 1q angles (the qiskit-free part of `refine`,
cpflow/exact_decompositions.py:76-118, 193-258, 293-344).  This stage "stays on the CPU" in the
north star; it is host logic on circuits of a few dozen gates, not part of the device hot path.

Steps, as in the reference: CP -> CZ (threshold cp_threshold), merge 1q runs to Rz Rx Rz, greedily
zero / merge angles while the loss stays below reduce_threshold, round angles to rational multiples of
pi with denominator <= max_denominator when within angle_threshold.  The Clifford+T step needs
qiskit's Solovay-Kitaev pass and is not provided: the result type is 'Approximate' or 'Rational'.
"""
import math
from fractions import Fraction

import numpy as np

from .circuit import Circuit, _bracket


def _with_angles(circuit, angles):
    out, i = Circuit(circuit.num_qubits), 0
    for name, qubits, params in circuit.data:
        if name in ('rx', 'ry', 'rz'):
            out.data.append((name, qubits, (float(angles[i]),)))
            i += 1
        else:
            out.data.append((name, qubits, params))
    return out


def _angles_and_wires(circuit):
    ang = [float(p[0]) for n, _, p in circuit.data if n in ('rx', 'ry', 'rz')]
    wires = [q[0] for n, q, _ in circuit.data if n in ('rx', 'ry', 'rz')]
    return np.array(ang), wires


def reduce_angles(circuit, loss_func, reduce_threshold=1e-5, cp_threshold=0.01):
    qc = circuit.cp_to_cz(cp_threshold=cp_threshold).to_zxz()
    angles, wires = _angles_and_wires(qc)
    loss = lambda a: loss_func(_with_angles(qc, a).unitary())  # noqa: E731
    if not loss(angles) < reduce_threshold:
        raise ValueError(f'Circuit loss {loss(angles)} is above threshold {reduce_threshold}.')
    for i in range(len(angles)):                       # reduce_all_1q_angles / reduce_first_1q_angle
        trial = angles.copy()
        trial[i] = 0
        if loss(trial) < reduce_threshold:
            angles = trial
            continue
        for j in range(i + 1, len(angles)):            # can_reduce_two_angles: same wire, a_j <- a_j +- a_i
            if wires[j] != wires[i]:
                continue
            done = False
            for sign in (-1, 1):
                trial = angles.copy()
                trial[j] = angles[j] + sign * angles[i]
                trial[i] = 0
                if loss(trial) < reduce_threshold:
                    angles, done = trial, True
                    break
            if done:
                break
    return _with_angles(qc, [_bracket(a) for a in angles])


def remove_zero_rgates(circuit, tol=1e-12):
    out = Circuit(circuit.num_qubits)
    out.data = [(n, q, p) for n, q, p in circuit.data if not (n in ('rx', 'ry', 'rz') and abs(p[0]) < tol)]
    return out


def rationalize_all_rgates(circuit, max_denominator=32, angle_threshold=1e-3):
    out = Circuit(circuit.num_qubits)
    for n, q, p in circuit.data:
        if n in ('rx', 'ry', 'rz'):
            a = float(p[0])
            rat = math.pi * Fraction.from_float(a / math.pi).limit_denominator(max_denominator)
            out.data.append((n, q, (float(rat) if abs(float(rat) - a) < angle_threshold else a,)))
        else:
            out.data.append((n, q, p))
    return out


def all_rgates_are_rational(circuit, power):
    for n, _, p in circuit.data:
        if n in ('rx', 'ry', 'rz'):
            a = float(p[0])
            f = Fraction(a / math.pi).limit_denominator(2 ** power)
            if not (abs(math.pi * f - a) < 1e-6 and math.log2(f.denominator).is_integer()):
                return False
    return True


def refine_decomposition(d, max_denominator=32, angle_threshold=0.01, cp_threshold=0.01, reduce_threshold=1e-5):
    try:
        qc = remove_zero_rgates(reduce_angles(d.circuit, d.unitary_loss_func, reduce_threshold, cp_threshold))
        kind = 'Approximate'
        rat = remove_zero_rgates(rationalize_all_rgates(qc, max_denominator, angle_threshold))
        if d.unitary_loss_func(rat.unitary()) < 1e-5:
            qc = rat
            if all_rgates_are_rational(qc, int(math.log2(max_denominator))):
                kind = 'Rational'
    except ValueError:
        return f'Refined to {d.type}'
    d.circuit, d.type = qc, kind
    d.unitary = qc.unitary()
    d.loss = d.unitary_loss_func(d.unitary)
    d.cz_count = qc.count_ops().get('cz', 0)
    d.cz_depth = qc.depth(['cz'])
    return f'Refined to {kind}'
